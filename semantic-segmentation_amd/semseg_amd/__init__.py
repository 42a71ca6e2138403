"""semseg_amd -- MI355X-native hot path of NVIDIA/semantic-segmentation
(HRNet-OCR-MScale forward/backward, losses, SyncBN, DDP) behind the reference's
own Python entry points.  Arithmetic lives in csrc/*.hip (libsemseg_hip.so)."""
__version__ = "0.1.0"


def graph_training(net, optim, warmup=2):
    """Proxies of (net, optim) under which the reference's train() loop replays ONE captured hipGraph per iteration
    (semseg_amd/graphed.py)."""
    from .graphed import graph_training as _g
    return _g(net, optim, warmup)


def GraphedTrainStep(net, optim, warmup=2):
    from .graphed import GraphedTrainStep as _G
    return _G(net, optim, warmup)
