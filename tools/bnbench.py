#!/usr/bin/env python
"""Micro-benchmark of the BatchNorm passes on the device, through the C ABI (no autograd): the three training passes
(apply, backward reduce, backward apply) of a stage-4 trunk level as ONE grouped launch each, replayed from a hipGraph.
python tools/bnbench.py [reps]      (rows per thread: SSA_BN_ROWS_APPLY / _BWD / _REDUCE = 2 | 4 | 8)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from semseg_amd import hip_backend as hb  # noqa: E402

L = hb.lib()
DEV = "cuda"
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731
DT = hb.ACT_DTYPE


class Prob:
    def __init__(self, C, H, W, seed):
        g = torch.Generator().manual_seed(seed)
        self.C, self.P = C, H * W
        mk = lambda: torch.randn(1, H, W, C, generator=g).to(DEV).to(DT)  # noqa: E731
        self.x, self.res, self.dz = mk(), mk(), mk()
        self.z = torch.empty_like(self.x)
        self.dx = torch.empty_like(self.x)
        self.dres = torch.empty_like(self.x)
        nrep = hb.stat_replicas()
        self.nrep = nrep
        xf = self.x.float().view(-1, C)
        st = torch.zeros(nrep, 2, C, dtype=torch.float64, device=DEV)
        st[0, 0] = xf.sum(0).double()
        st[0, 1] = (xf * xf).sum(0).double()
        self.stats = st
        self.bsums = torch.zeros(nrep, 2, C, dtype=torch.float64, device=DEV)
        self.gamma = torch.rand(C, device=DEV) + 0.5
        self.beta = torch.rand(C, device=DEV) - 0.5
        self.coef = torch.empty(4, C, device=DEV)
        self.pg = torch.zeros(2, C, device=DEV)
        self.mask = torch.zeros(self.P * (C // 8), dtype=torch.uint8, device=DEV)      # sign bytes of z (bn2 levels)
        self.use_mask = os.environ.get("SSA_BN_SIGN_MASK", "1") != "0"

    def apply(self, res):
        hb.check(L.ssa_bn_apply_train(P(self.x), self.C, P(self.res) if res else None, self.C, P(self.z), self.C, self.P, self.C,
                                      P(self.stats), self.nrep, float(self.P), P(self.gamma), P(self.beta), None, None, None,
                                      0.1, 1e-5, P(self.coef), None, 1, None, self.P, P(self.mask) if (res and self.use_mask) else None, hb._s()), "apply")

    def apply_eval(self, res):
        """the evaluation form: scale / shift given (no statistics prologue, no LDS, no barrier)"""
        hb.check(L.ssa_bn_apply(P(self.x), self.C, P(self.res) if res else None, self.C, P(self.z), self.C, self.P, self.C,
                                P(self.coef[0]), P(self.coef[1]), 1, None, self.P, hb._s()), "apply_eval")

    def reduce(self, from_x):
        msc, msh = (self.coef[0], self.coef[1]) if from_x else (None, None)
        hb.check(L.ssa_bn_bwd_reduce(P(self.x), self.C, P(self.dz), self.C, None if from_x else P(self.z), self.C, self.P, self.C,
                                     P(self.coef[2]), P(self.coef[3]), 1, None, self.P, P(self.bsums), self.nrep, 0,
                                     P(msc), P(msh), P(self.mask) if (self.use_mask and not from_x) else None, hb._s()), "reduce")

    def bapply(self, from_x, dres):
        msc, msh = (self.coef[0], self.coef[1]) if from_x else (None, None)
        hb.check(L.ssa_bn_bwd_apply(P(self.x), self.C, P(self.dz), self.C, None if from_x else P(self.z), self.C, P(self.dx), self.C,
                                    P(self.dres) if dres else None, self.C, self.P, self.C, P(self.gamma), P(self.coef[2]),
                                    P(self.coef[3]), P(self.bsums), self.nrep, float(self.P), 1, None, self.P, P(self.pg[0]),
                                    P(self.pg[1]), 1.0, P(msc), P(msh), 1, P(self.mask) if (self.use_mask and not from_x) else None,
                                    hb._s()), "bapply")


    def fused(self, from_x, dres, ticket):
        """reduce + rendezvous + apply in one launch (csrc/bn.hip bn_bwd_fused_body); ticket: a zeroed 32-bit word"""
        msc, msh = (self.coef[0], self.coef[1]) if from_x else (None, None)
        hb.check(L.ssa_bn_bwd_fused(P(self.x), self.C, P(self.dz), self.C, None if from_x else P(self.z), self.C, P(self.dx), self.C,
                                    P(self.dres) if dres else None, self.C, self.P, self.C, P(self.gamma), P(self.coef[2]),
                                    P(self.coef[3]), P(self.bsums), self.nrep, float(self.P), 1, None, self.P, P(self.pg[0]),
                                    P(self.pg[1]), 1.0, P(msc), P(msh), 1, P(self.mask) if (self.use_mask and not from_x) else None,
                                    ctypes.c_void_p(ticket), hb._s()), "fused")


TICKETS = None      # one zeroed word per fused call of a captured graph, cleared by a memset node at the graph's start


def timeit(fn, reps, pre=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            if pre is not None:
                pre()
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    level = [(48, 256, 256), (48, 128, 128), (96, 128, 128), (96, 64, 64), (192, 64, 64), (192, 32, 32), (384, 32, 32), (384, 16, 16)]
    probs = [Prob(C, H, W, i) for i, (C, H, W) in enumerate(level)]
    elems = sum(p.P * p.C for p in probs)

    def lv(fn):
        def run():
            with hb.group():
                for p in probs:
                    fn(p)
        return run
    rows = {k: os.environ.get(k, "-") for k in ("SSA_BN_ROWS_APPLY", "SSA_BN_ROWS_BWD", "SSA_BN_ROWS_REDUCE")}
    print("level of 8 problems, %.2f M elements; rows %s" % (elems / 1e6, rows))
    for name, fn, bpe in (("apply (bn1: no residual)", lambda p: p.apply(False), 4), ("apply (bn2: + residual)", lambda p: p.apply(True), 6),
                          ("apply, coefficients given (bn1)", lambda p: p.apply_eval(False), 4),
                          ("apply, coefficients given (bn2)", lambda p: p.apply_eval(True), 6),
                          ("bwd reduce (mask from z)", lambda p: p.reduce(False), 6), ("bwd reduce (mask from x)", lambda p: p.reduce(True), 4),
                          ("bwd apply (bn2: z mask, dres)", lambda p: p.bapply(False, True), 10),
                          ("bwd apply (bn1: x mask)", lambda p: p.bapply(True, False), 6)):
        t = timeit(lv(fn), reps)
        print("  %-32s %7.2f us   %5.2f TB/s algorithmic (%d B/element)" % (name, t, elems * bpe / t / 1e6, bpe))
    # the one-launch backward: every call its own ticket word (8 problems x reps + warm-up calls), cleared per replay
    tickets = torch.zeros(8 * (reps + 8) * 8, dtype=torch.int32, device=DEV)
    state = {"i": 0}

    def fused_level(from_x, dres):
        def run():
            with hb.group():
                for p in probs:
                    p.fused(from_x, dres, tickets.data_ptr() + 4 * state["i"])
                    state["i"] += 1
        return run

    def pre():
        state["i"] = 0
        tickets.zero_()
    for name, fn, bpe in (("bwd fused (bn2: z mask, dres)", fused_level(False, True), 10), ("bwd fused (bn1: x mask)", fused_level(True, False), 6)):
        pre()
        torch.cuda.synchronize()
        t = timeit(fn, reps, pre)
        print("  %-32s %7.2f us   %5.2f TB/s algorithmic (%d B/element; the two launches it replaces read 6 / 4 more)" % (name, t, elems * bpe / t / 1e6, bpe))
    to = ctypes.c_uint(0)
    L.ssa_bn_bwd_fused_timeouts(ctypes.byref(to))
    print("  rendezvous timeouts: %d" % to.value)
    p0 = probs[0]
    for name, fn in (("apply 48@256^2 alone", lambda: p0.apply(True)), ("bwd apply 48@256^2 alone", lambda: p0.bapply(False, True))):
        print("  %-32s %7.2f us" % (name, timeit(fn, reps)))


if __name__ == "__main__":
    main()
