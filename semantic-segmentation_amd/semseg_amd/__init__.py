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


def graph_eval(net, max_graphs=4, clone_outputs=True):
    """Proxy of `net` whose evaluation-mode calls under torch.no_grad() replay one captured forward per input signature
    (semseg_amd/graphed.py GraphedEval): what the reference's validate() loop calls once per image."""
    from .graphed import graph_eval as _g
    return _g(net, max_graphs, clone_outputs)
