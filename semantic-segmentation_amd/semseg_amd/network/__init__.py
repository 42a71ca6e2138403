"""Model factory with the reference's contract (network/__init__.py:12-54)."""
import importlib

import torch


def get_model(network, num_classes, criterion):
    """`network` is '<module>.<Factory>' as in the reference's --arch flag
    (an optional leading 'network.' is accepted)."""
    network = network[len("network."):] if network.startswith("network.") else network
    module, _, model = network.rpartition(".")
    mod = importlib.import_module(__name__ + "." + module)
    return getattr(mod, model)(num_classes=num_classes, criterion=criterion)


def get_net(args, criterion):
    """network/__init__.py:12-23"""
    from ..config import cfg
    net = get_model(network=args.arch, num_classes=cfg.DATASET.NUM_CLASSES, criterion=criterion)
    return net.cuda()


def wrap_network_in_dataparallel(net, use_apex_data_parallel=False):
    """network/__init__.py:33-42: the multi-process path wraps in our DDP."""
    if use_apex_data_parallel:
        from ..parallel import DistributedDataParallel
        return DistributedDataParallel(net)
    return torch.nn.DataParallel(net)
