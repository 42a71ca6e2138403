"""The training step of the reference's loop (train.py:480-533) as ONE replayed hipGraph.

    optim.zero_grad(); loss = net(inputs); loss.mean().backward(); optim.step()

is ~840 kernel launches through ctypes plus Python autograd when it runs eagerly -- host bound (about four times the
GPU time of the step at batch 1).  `GraphedTrainStep` captures exactly that sequence once per input shape (static
input buffers, the batch is copied in) and replays it; `graph_training(net, optim)` wraps the two objects the
reference's loop holds so that the UNMODIFIED loop runs the captured step:

    net, optim = semseg_amd.graph_training(net, optim)      # e.g. from the amp.initialize shim, train.py:381
    ...
    optim.zero_grad()            # no-op (inside the graph)
    main_loss = net(inputs)      # copy-in + replay: forward, backward AND optimizer step; returns the step's loss
    main_loss.mean().backward()  # a leaf: nothing left to do
    optim.step()                 # no-op (inside the graph); learning-rate changes reach the graph through the
                                 # optimizer's device scalar (FusedSGD.sync_lr)

Everything the step needs is capture safe by construction (no host synchronisation, torch's caching allocator,
kernels on the current stream; at N > 1 the SyncBN and gradient exchanges are direct RCCL nodes).

Evaluation (`net.eval()` under `torch.no_grad()`: what utils/trnval_utils.py:134-141 calls once per image, flip and scale)
is the same host-bound story -- ~830 ctypes launches per pass -- and goes through `GraphedEval`: one captured forward per
input signature, the least recently used signature evicted beyond `max_graphs` (validation images of a dataset like
Mapillary come in many sizes; a captured forward owns its activation pool).  `graph_eval(net)` gives an evaluation-only
run the same proxy; `graph_training` includes it.  SSA_GRAPHED_EVAL=0: evaluation goes straight to the module.

Guards: at most `max_graphs` input signatures are captured (a run with ragged shapes would otherwise multiply the
activation pools); further signatures, and every step after a capture that raised (logged once), run the SAME
sequence eagerly.  With N > 1 every rank must see the same shapes in the same iteration (the warm-up steps of a
capture issue collectives).  Reloading the optimizer's state (`optim.load_state_dict`) drops the captured graphs:
they hold the addresses of the old momentum buffers.
"""
import collections
import os
import sys

import torch
from torch import nn


class GraphedTrainStep:
    """Captured `zero_grad -> net(inputs) -> backward -> optim.step` for dict inputs of device tensors."""

    def __init__(self, net, optim, warmup=2, max_graphs=4):
        self.net, self.optim, self.warmup, self.max_graphs = net, optim, warmup, max_graphs
        self._graphs = {}          # input signature -> (graph, static inputs, static loss)
        self.eager_only = False    # set after a capture that raised
        self.replays = 0

    def invalidate(self):
        """Forget the captured graphs (the optimizer's state tensors were replaced)."""
        self._graphs = {}

    def _signature(self, inputs):
        return tuple((k, tuple(v.shape), v.dtype, v.device) for k, v in sorted(inputs.items()) if torch.is_tensor(v))

    def _eager(self, static, loss_out):
        self.optim.zero_grad(set_to_none=True)
        loss = self.net(static)
        loss = loss.mean()
        scaler = getattr(self.optim, "loss_scaler", None)       # fp16 training (semseg_amd/amp.py): backward on loss * S
        (scaler.scale(loss) if scaler is not None else loss).backward()
        self.optim.step()                                       # (un-scales, skips on overflow, updates S)
        loss_out.copy_(loss.detach())

    def _capture(self, inputs):
        static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in inputs.items()}
        dev = next(v.device for v in static.values() if torch.is_tensor(v))
        loss_out = torch.zeros((), device=dev)
        # the warm-up passes below are real steps; what they change is put back after the capture, so that a new
        # input shape costs the training run nothing but time
        params = [p for g in self.optim.param_groups for p in g["params"]]
        snap_p = [p.detach().clone() for p in params]
        snap_m = [self.optim.state[p]["momentum_buffer"].clone() if "momentum_buffer" in self.optim.state.get(p, {}) else None
                  for p in params]
        bufs = list(self.net.buffers())
        snap_b = [b.detach().clone() for b in bufs]
        scaler = getattr(self.optim, "loss_scaler", None)
        snap_s = scaler.state.clone() if scaler is not None else None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(self.warmup):        # allocator, momentum buffers, packed-filter cache
                    self._eager(static, loss_out)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            self.optim.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                self._eager(static, loss_out)
        finally:
            # also when the capture raised: the warm-up steps must not have trained the network
            torch.cuda.synchronize()
            with torch.no_grad():
                for p, s0, m0 in zip(params, snap_p, snap_m):
                    p.copy_(s0)                     # (bumps the version counter: the captured step re-packs the filters)
                    buf = self.optim.state.get(p, {}).get("momentum_buffer")
                    if buf is not None:
                        if m0 is not None:
                            buf.copy_(m0)
                        else:
                            buf.zero_()             # a zero buffer is the optimizer's first-step state
                for b, s0 in zip(bufs, snap_b):
                    b.copy_(s0)
                if snap_s is not None:
                    scaler.state.copy_(snap_s)
        return graph, static, loss_out

    def _run_eager(self, inputs):
        dev = next(v.device for v in inputs.values() if torch.is_tensor(v))
        loss_out = torch.zeros((), device=dev)
        if hasattr(self.optim, "sync_lr"):
            self.optim.sync_lr()
        self._eager(inputs, loss_out)
        return loss_out

    def __call__(self, inputs):
        """One training step on `inputs`; returns the step's (mean) loss as a detached 0-dim tensor."""
        sig = self._signature(inputs)
        ent = self._graphs.get(sig)
        fresh = ent is None
        if fresh:
            if self.eager_only or len(self._graphs) >= self.max_graphs:
                return self._run_eager(inputs)
            try:
                ent = self._capture(inputs)
            except Exception as e:          # noqa: BLE001 -- whatever refused the capture: run the same step eagerly
                # ... on ONE rank only.  With several ranks a capture that raised on this rank (possibly half way through
                # the warm-up steps' collectives) has already left the ranks' collective sequences out of step: falling
                # back to eager launches here while the others replay a graph would hang the job instead of failing it.
                if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                        torch.distributed.get_world_size() > 1:
                    raise
                self.eager_only = True
                print("semseg_amd.graphed: hipGraph capture of the training step failed (%s: %s); the loop goes on with "
                      "eager launches of the same step" % (type(e).__name__, str(e)[:300]), file=sys.stderr, flush=True)
                return self._run_eager(inputs)
            self._graphs[sig] = ent
        graph, static, loss_out = ent
        if hasattr(self.optim, "sync_lr"):
            self.optim.sync_lr()                # the scheduler's current learning rate -> the device scalar
        if not fresh:
            for k, v in inputs.items():
                if torch.is_tensor(v):
                    static[k].copy_(v, non_blocking=True)
        graph.replay()
        self.replays += 1
        return loss_out.clone()


def _map_tensors(obj, fn):
    """fn over every tensor of a (possibly nested) dict / list / tuple / tensor result; everything else as it is."""
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return type(obj)((k, _map_tensors(v, fn)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(v, fn) for v in obj)
    return obj


class GraphedEval:
    """Captured `with torch.no_grad(): net(inputs)` of an evaluation-mode network, one hipGraph per input signature.

    The reference's validation loop (utils/trnval_utils.py:115-160) calls `net(inputs)` once per image, flip and scale
    and reads `output_dict['pred']` (plus the `pred_*` / `attn_*` assets of the last call).  A call here copies the batch
    into the signature's static inputs, re-packs the filters of parameters that changed since the last call (training
    between two validations; one batched launch, outside the graph), replays, and returns CLONES of the static outputs
    (`clone_outputs=False` hands out the static buffers themselves: valid until the next call with that signature).
    At most `max_graphs` signatures stay captured, least recently used evicted first.  A signature is captured when it
    comes the `capture_after`-th time (default: the second) and runs with eager launches before that: a capture costs
    two more host-side passes than an eager call, which a validation set whose images all differ in size (Mapillary)
    would pay on every image and never get back; a fixed-size set (Cityscapes) replays from its third image on."""

    def __init__(self, net, warmup=1, max_graphs=4, clone_outputs=True, capture_after=2):
        self.net, self.warmup, self.max_graphs, self.clone_outputs = net, warmup, max(1, int(max_graphs)), clone_outputs
        self.capture_after = max(1, int(capture_after))
        self._graphs = collections.OrderedDict()       # signature -> (graph, static inputs, static outputs)
        self._seen = collections.Counter()             # signature -> calls so far (signatures not captured yet)
        self.eager_only = False
        self.replays = self.captures = self.evictions = self.eager_calls = 0

    def invalidate(self):
        self._graphs.clear()

    @staticmethod
    def _signature(inputs):
        return tuple((k, tuple(v.shape), v.dtype, v.device) for k, v in sorted(inputs.items()) if torch.is_tensor(v))

    def _forward(self, inputs):
        with torch.no_grad():
            return self.net(inputs)

    def _capture(self, inputs):
        static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in inputs.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):            # allocator, packed-filter cache (the capture must not pack)
                self._forward(static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self._forward(static)
        return graph, static, out

    def __call__(self, inputs):
        from . import hip_backend
        sig = self._signature(inputs)
        ent = self._graphs.get(sig)
        if ent is None:
            if self.eager_only:
                return self._forward(inputs)
            self._seen[sig] += 1
            if self._seen[sig] < self.capture_after:
                if len(self._seen) > 4096:          # a long run over ever-new sizes: forget the counts, not the graphs
                    self._seen.clear()
                self.eager_calls += 1
                return self._forward(inputs)
            del self._seen[sig]
            while len(self._graphs) >= self.max_graphs:      # free the evicted forward's pool BEFORE capturing the next
                self._graphs.popitem(last=False)
                self.evictions += 1
            try:
                ent = self._capture(inputs)
            except Exception as e:          # noqa: BLE001 -- whatever refused the capture: evaluation runs eagerly
                self.eager_only = True
                print("semseg_amd.graphed: hipGraph capture of the evaluation forward failed (%s: %s); evaluation goes "
                      "on with eager launches" % (type(e).__name__, str(e)[:300]), file=sys.stderr, flush=True)
                torch.cuda.synchronize()
                return self._forward(inputs)
            self._graphs[sig] = ent
            self.captures += 1
        else:
            self._graphs.move_to_end(sig)
            graph, static, _ = ent
            for k, v in inputs.items():
                if torch.is_tensor(v):
                    static[k].copy_(v, non_blocking=True)
            # parameters trained since this forward was captured: their packed 16-bit forms are what the graph reads
            hip_backend.refresh_packed_filters()
        graph, static, out = ent
        graph.replay()
        self.replays += 1
        return _map_tensors(out, lambda t: t.clone()) if self.clone_outputs else _map_tensors(out, lambda t: t)


class _GraphedNet(nn.Module):
    """What the reference's loop calls as `net(inputs)`: in training mode the whole captured step.

    Transparent to the module tree: the proxy SHARES the wrapped net's `_modules` / `_parameters` / `_buffers`
    dictionaries instead of registering the net as a child, so `state_dict()` / `load_state_dict()` / `parameters()` /
    `.cuda()` / `.train()` see exactly the wrapped net's names -- also through an outer wrapper
    (`DistributedDataParallel(_GraphedNet(net)).state_dict()` has `module.X` keys and loads them back, which is what
    the reference's `restore_net` -> `forgiving_state_restore`, train.py:396, does with --snapshot / --resume)."""

    def __init__(self, net, stepper, eval_stepper=None):
        super().__init__()
        object.__setattr__(self, "_net", net)          # not a registered submodule
        object.__setattr__(self, "_stepper", stepper)  # GraphedTrainStep, or None (graph_eval: training runs eagerly)
        object.__setattr__(self, "_eval_stepper", eval_stepper)
        self._modules = net._modules
        self._parameters = net._parameters
        self._buffers = net._buffers
        # ... and the rest of the state the (load_)state_dict machinery consults: which buffers are non-persistent, the
        # hooks registered on the wrapped net
        for name in ("_non_persistent_buffers_set", "_state_dict_hooks", "_state_dict_pre_hooks",
                     "_load_state_dict_pre_hooks", "_load_state_dict_post_hooks"):
            if hasattr(net, name):
                object.__setattr__(self, name, getattr(net, name))
        self.training = net.training

    @property
    def wrapped(self):
        return self._net

    def forward(self, inputs):
        if self._net.training and torch.is_grad_enabled() and self._stepper is not None:
            loss = self._stepper(inputs)
            return loss.requires_grad_(True)   # a leaf: the loop's own .backward() has nothing to do
        ev = self._eval_stepper
        if ev is not None and not self._net.training and not torch.is_grad_enabled() and isinstance(inputs, dict) and \
                any(torch.is_tensor(v) and v.is_cuda for v in inputs.values()):
            return ev(inputs)                  # validation: one replayed forward per image (GraphedEval)
        return self._net(inputs)

    def train(self, mode=True):
        self._net.train(mode)
        self.training = mode
        return self

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(object.__getattribute__(self, "_net"), name)     # .module of the data-parallel wrapper etc.

    def __setattr__(self, name, value):
        # plain attributes set through the proxy (net.foo = x) belong to the wrapped net; tensors / modules go through
        # nn.Module's registration, which lands in the SHARED dictionaries
        if name == "training" or isinstance(value, (torch.Tensor, nn.Module)) or name.startswith("_"):
            super().__setattr__(name, value)
        else:
            setattr(object.__getattribute__(self, "_net"), name, value)


class _GraphedOptim:
    """The optimizer as the loop sees it: zero_grad / step are inside the captured step."""

    def __init__(self, optim, stepper=None):
        self.__dict__["_optim"] = optim
        self.__dict__["_stepper"] = stepper

    def zero_grad(self, *a, **k):
        pass

    def step(self, *a, **k):
        pass

    def load_state_dict(self, *a, **k):
        r = self.__dict__["_optim"].load_state_dict(*a, **k)
        st = self.__dict__.get("_stepper")
        if st is not None:
            st.invalidate()                 # the graphs point at the momentum buffers that were just replaced
        return r

    def __getattr__(self, name):
        return getattr(self.__dict__["_optim"], name)

    def __setattr__(self, name, value):
        setattr(self.__dict__["_optim"], name, value)


def _eval_stepper(net, max_eval_graphs, clone_outputs=True, capture_after=2):
    if os.environ.get("SSA_GRAPHED_EVAL", "1") == "0" or max_eval_graphs <= 0:
        return None
    return GraphedEval(net, max_graphs=max_eval_graphs, clone_outputs=clone_outputs, capture_after=capture_after)


def graph_training(net, optim, warmup=2, max_graphs=4, max_eval_graphs=4):
    """(net, optim) -> proxies under which the reference's unmodified train() loop runs one hipGraph replay per
    iteration and its validate() loop one replayed forward per image (see the module docstring)."""
    stepper = GraphedTrainStep(net, optim, warmup, max_graphs)
    return _GraphedNet(net, stepper, _eval_stepper(net, max_eval_graphs)), _GraphedOptim(optim, stepper)


def graph_eval(net, max_graphs=4, clone_outputs=True, capture_after=2):
    """net -> proxy whose evaluation-mode calls under torch.no_grad() replay a captured forward per input signature
    (GraphedEval); training-mode calls go to the module unchanged.  For evaluation-only runs (train.py --eval val/folder,
    utils/trnval_utils.py:134-141): `net = semseg_amd.graph_eval(net)`."""
    return _GraphedNet(net, None, _eval_stepper(net, max_graphs, clone_outputs, capture_after))
