from .criteria import RMILoss, CrossEntropyLoss2d, get_loss  # noqa: F401
from .optimizer import FusedSGD, get_optimizer  # noqa: F401,E402
