from .criteria import RMILoss, CrossEntropyLoss2d, get_loss  # noqa: F401
