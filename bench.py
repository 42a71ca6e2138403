#!/usr/bin/env python
"""Headline benchmark: training images/s of HRNet-OCR-MScale at 1024x1024 crop
(BASELINE.json metric), synthetic Cityscapes-shaped data, random-init weights.

    python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU,
RCCL); RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment.

One "step" = zero_grad + forward (0.5x and 1.0x passes, attention fusion,
RMI + BCE losses as in scripts/train_cityscapes_sota.yml) + backward (+ DDP
gradient all-reduce and SyncBN when N > 1) + SGD step, on one 1x3x1024x1024
batch per GPU that is already resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)



def _early_args(argv):
    """--dtype and --gpus, read before torch / the package are imported: the storage format selects which build of the
    library the process loads (semseg_amd/_lib.py reads SSA_ACT_DTYPE at import), and a bare `python bench.py --gpus N`
    (no torch.distributed.run around it) re-executes itself under the launcher."""
    dtype, gpus = None, 1
    for i, a in enumerate(argv):
        if a == "--dtype" and i + 1 < len(argv):
            dtype = argv[i + 1]
        elif a.startswith("--dtype="):
            dtype = a.split("=", 1)[1]
        elif a == "--gpus" and i + 1 < len(argv):
            gpus = int(argv[i + 1])
        elif a.startswith("--gpus="):
            gpus = int(a.split("=", 1)[1])
    return dtype, gpus


_DTYPE, _GPUS = _early_args(sys.argv[1:])
if _GPUS > 1 and "WORLD_SIZE" not in os.environ and "--cpu-baseline-only" not in sys.argv:
    # the contract's N > 1 launch is `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`; a bare
    # `python bench.py --gpus N` (the shape of the driver's N = 1 command) becomes exactly that
    import socket
    with socket.socket() as _sk:
        _sk.bind(("127.0.0.1", 0))
        _port = _sk.getsockname()[1]
    os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(_GPUS),
                              "--master-addr", "127.0.0.1", "--master-port", str(_port), os.path.abspath(__file__)]
             + sys.argv[1:])
# fp16 storage with dynamic loss scaling is the format of record: it is the reference's own arithmetic (every recipe sets
# `fp16: true`, scripts/*.yml; train.py:380-381 apex O1), gfx950 runs fp16 MFMA at the bf16 rate, and it sits an order of
# magnitude closer to the fp32 oracle than bf16 storage.  --dtype bf16 (or SSA_ACT_DTYPE) selects
# the other build; the default run reports the bf16 step beside the headline (value_bf16).
if _DTYPE is not None:
    os.environ["SSA_ACT_DTYPE"] = _DTYPE
os.environ.setdefault("SSA_ACT_DTYPE", "fp16")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# SURVEY.md section 8d / BASELINE.md section 2: conv FLOPs per training image at
# 1024x1024 (two scales, 638 convs), measured from the reference on `meta`.
FLOP_FWD_BWD_PER_IMAGE = 5.7274e12
PEAK_BF16_MFMA = 2.5e15          # dense, MI355X_MICROARCH.md
PEAK_HBM = 8.0e12                # bytes/s, MI355X_MICROARCH.md
def _lib_sha():
    """Hash of the kernel sources the loaded library was built from (embedded by csrc/Makefile, checked against the
    sources on disk by semseg_amd._lib.lib())."""
    from semseg_amd import _lib
    return _lib.built_sha()


CONV_FAMILIES = ("ConvTile", "ConvHaloGemm", "ConvGemmWide", "ConvIgemm", "ConvWgradTile", "ConvWgradHead", "ConvWgradTr")


def source_sha():
    """Hash of the kernel sources: profiles/*_pmc_traffic.json records it, and a traffic file measured
    on other kernels is refused (the GPU box has no .git to compare heads with)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "semantic-segmentation_amd", "csrc", "*.h*"))):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def family(kernel):
    f = kernel.split("<")[0]
    # the trunk's 3x3 kernels: conv_tile.hip (ConvTile, ConvTileAny) and its persistent form conv_tile_p.hip
    if f in ("ConvTileAny", "ConvTilePAny", "ConvTilePK", "ConvTileP"):
        return "ConvTile"
    # the head convs: conv_halo_gemm.hip (3x3: ConvHaloGemm3, 1x1: ConvHaloGemm1), conv_wgrad_head.hip (3x3: ...Head3)
    # conv_gemm_wide.hip: the large 1x1 convs on the 256 x 256 tile
    return {"ConvHaloGemm3": "ConvHaloGemm", "ConvHaloReg3": "ConvHaloGemm", "ConvHaloGemm1": "ConvHaloGemm", "ConvWgradHead3": "ConvWgradHead",
            "ConvGemmWide1": "ConvGemmWide", "ConvWgradTileA": "ConvWgradTile",
            # the BatchNorm backward passes, whichever way the ReLU mask arrives (csrc/bn.hip MODE)
            "BnBwdApplyXK": "BnBwdApplyK", "BnBwdFusedXK": "BnBwdFusedK", "BnBwdFusedZK": "BnBwdFusedK", "BnBwdApplyZK": "BnBwdApplyK", "BnBwdReduceXK": "BnBwdReduceK",
            "BnBwdReduceZK": "BnBwdReduceK"}.get(f, f)


def synth_batch(B, H, W, rank, device):
    """N(0,1) image, 64x64 blocks of uniform class ids,
    ~10% ignore (255); seed 1234 + rank."""
    g = torch.Generator().manual_seed(1234 + rank)
    images = torch.randn(B, 3, H, W, generator=g)
    bs = 64
    blocks = torch.randint(0, 19, (B, (H + bs - 1) // bs, (W + bs - 1) // bs), generator=g)
    gts = blocks.repeat_interleave(bs, 1).repeat_interleave(bs, 2)[:, :H, :W].clone()
    gts[torch.rand(B, H, W, generator=g) < 0.1] = 255
    return images.to(device), gts.long().to(device)


def build_model(world):
    from semseg_amd.config import cfg
    from semseg_amd.loss import RMILoss
    from semseg_amd.network import ocrnet
    from semseg_amd import nn as snn
    cfg.LOSS.SUPERVISED_MSCALE_WT = 0.05        # scripts/train_cityscapes_sota.yml
    cfg.LOSS.OCR_AUX_RMI = False
    cfg.MODEL.N_SCALES = None
    cfg.MODEL.BNFUNC = snn.SyncBatchNorm if (world > 1 or FORCE_DIST) else None
    torch.manual_seed(0)
    net = ocrnet.HRNet_Mscale(19, RMILoss(num_classes=19, ignore_index=255))
    for m in net.modules():                     # random init at a realistic scale
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
    return net.cuda().train()


# SSA_FORCE_DIST=1: run the N > 1 code path (SyncBN exchanges, DDP hooks, RCCL) in a world of one
# rank, so that a one-GPU box can exercise it -- in particular inside hipGraph capture.
FORCE_DIST = os.environ.get("SSA_FORCE_DIST", "0") == "1"


def secondary_run(args, dtype, timeout_s=420):
    """The identical benchmark step on the other storage build, in a child process (no roofline leg, no CPU baseline)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--dtype", dtype, "--gpus", "1", "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--crop", str(args.crop), "--crop-w", str(args.crop_w), "--batch", str(args.batch),
           "--no-cpu-baseline", "--no-roofline", "--no-secondary", "--eager-steps", "0"]
    if args.no_graph:
        cmd.append("--no-graph")
    env = {k: v for k, v in os.environ.items() if k not in ("SSA_ACT_DTYPE", "RANK", "LOCAL_RANK", "WORLD_SIZE")}
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                j = json.loads(line)
                return {"value": j["value"], "ms_per_step": j["ms_per_step"], "dtype": j["dtype"], "loss": j["config"]["loss"]}
        return {"value": None, "error": (out.stderr or out.stdout)[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "error": "secondary %s run exceeded %d s" % (dtype, timeout_s)}


def cpu_baseline(timeout_s=240):
    """Run the CPU baseline in a child process under a hard time limit so that a
    misbehaving host (oversubscribed cores) can never stall the benchmark."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"],
                             capture_output=True, text=True, timeout=timeout_s,
                             env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "error": (out.stderr or out.stdout)[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "error": "cpu baseline exceeded %d s" % timeout_s}


def _cpu_baseline_impl(crop=1024, timed=3):
    """The oracle (CPU restatement of the reference's modules) timed on this
    host's cores on a bounded sample of the same workload.  The thread count is
    calibrated (one 128x128 iteration per candidate): forcing one thread per
    visible CPU oversubscribes a cgroup-limited container (measured on the GPU
    box: 128 threads -> 19.5 s/iter at 256x256, 14x slower than 8 threads)."""
    from oracle.model import Net, seeded_state_dict
    from semseg_amd.network import ocrnet
    from semseg_amd.loss import RMILoss
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    net = ocrnet.HRNet_Mscale(19, RMILoss(num_classes=19))
    shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    del net
    sd = seeded_state_dict(shapes, seed=0)
    for k, v in sd.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)

    def one_iter(c):
        images, gts = synth_batch(1, c, c, 0, "cpu")
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        loss = Net(sd, 19, training=True, mscale_wt=0.05).two_scale_forward(images, gts)
        loss.backward()
        return time.perf_counter() - t0

    best_n, best_t = None, None
    for n in sorted({min(avail, c) for c in (16, 32, 64, 128)}):
        torch.set_num_threads(n)
        one_iter(256)
        t = one_iter(256)
        if best_t is None or t < best_t:
            best_n, best_t = n, t
    ncores = best_n
    torch.set_num_threads(ncores)
    times = [one_iter(crop) for _ in range(1 + timed)]
    per_iter = sum(times[1:]) / timed
    return {"value": 1.0 / per_iter, "unit": "images/s", "cores": ncores, "kind": "port",
            "sample": "oracle (CPU restatement of the reference's modules -- /root/reference does not exist on the GPU box; "
                      "bit-identical to them on fresh inputs, tests/test_oracle_golden.py) two-scale train step fwd+bwd, "
                      "fp32, the benchmarked workload itself: %d timed iters after 1 warm-up at %dx%d crop, batch 1 "
                      "(%.2f s/iter) on %d torch threads -- the fastest of 16/32/64/128 probed at 256x256, NOT all of "
                      "the %d CPUs visible (more threads are slower for this graph); the pool's hosts differ: the same "
                      "code measured 0.151-0.166 images/s between rounds"
                      % (timed, crop, crop, per_iter, ncores, avail)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--crop", type=int, default=1024)
    ap.add_argument("--crop-w", type=int, default=0, help="crop width if not square (the reference's sota recipe "
                    "trains on 1024x2048: scripts/train_cityscapes_sota.yml:14)")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step in a hipGraph")
    ap.add_argument("--eager-steps", type=int, default=3, help="eager steps timed after the graph run (0: none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=("fp16", "bf16"), default=None, help="storage format = library build "
                    "(default fp16 with dynamic loss scaling: the reference's --fp16; SSA_ACT_DTYPE is honoured too)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the bf16 step timed beside the fp16 headline")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(_cpu_baseline_impl()))
        return

    if os.environ.get("SSA_DEBUG_HANG"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SSA_DEBUG_HANG"]), repeat=True, file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d: launch with torch.distributed.run --nproc-per-node %d" % (
        world, args.gpus, args.gpus)
    # SSA_BENCH_ONE_DEVICE / SSA_DIST_BACKEND: self-test of the N > 1 code path on a one-GPU box
    # (all ranks on cuda:0, gloo collectives, eager); the driver's multi-GPU runs use neither.
    torch.cuda.set_device(0 if os.environ.get("SSA_BENCH_ONE_DEVICE") else local_rank)
    dist_on = world > 1 or FORCE_DIST
    backend = os.environ.get("SSA_DIST_BACKEND", "nccl")
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend=backend, init_method="env://")

    from semseg_amd import hip_backend as hb, rccl
    from semseg_amd._lib import lib
    net = build_model(world)
    model = net
    if dist_on:
        from semseg_amd.parallel import DistributedDataParallel
        model = DistributedDataParallel(net)
    # SGD + momentum + weight decay as in loss/optimizer.py:47-53: the one-pass HIP step
    # (ssa_sgd_momentum_step); SSA_FUSED_SGD=0: torch's multi-tensor SGD
    fused_sgd = os.environ.get("SSA_FUSED_SGD", "1") != "0"
    if fused_sgd:
        from semseg_amd.loss.optimizer import FusedSGD
        optim = FusedSGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    else:
        optim = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    crop_w = args.crop_w or args.crop
    images, gts = synth_batch(args.batch, args.crop, crop_w, rank, "cuda")
    inputs = {"images": images, "gts": gts}
    static_loss = torch.zeros((), device="cuda")
    # fp16 storage (SSA_ACT_DTYPE=fp16, the reference's --fp16): apex's dynamic loss scaling on the device, inside the
    # captured step (semseg_amd/amp.py); bf16 storage: scaler is None, scale_loss the identity
    from semseg_amd import amp as samp
    scaler = samp.attach_scaler(optim, torch.device("cuda")) if (samp.fp16_storage() and fused_sgd) else None

    def step():
        optim.zero_grad(set_to_none=True)
        loss = model(inputs)
        (scaler.scale(loss) if scaler is not None else loss).backward()
        optim.step()
        static_loss.copy_(loss.detach())

    # N > 1 is the same program as N = 1: the SyncBN exchanges and the gradient all-reduce are direct
    # RCCL calls on the compute stream, captured as nodes of the step's hipGraph.  (c10d collectives
    # cannot be captured -- its watchdog thread queries events during capture -- so a non-RCCL backend
    # runs eager.)
    use_graph = (not args.no_graph) and (not dist_on or (backend == "nccl" and rccl.ENABLED))
    graph = None
    launches_per_step = collectives_per_step = None
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):                      # eager steps: allocator warm-up, momentum buffers, filter cache
            step()
        torch.cuda.synchronize()
        lib().ssa_launch_count(1)
        c0 = rccl.total_calls() if (dist_on and backend == "nccl" and rccl.ENABLED) else 0
        step()
        launches_per_step = int(lib().ssa_launch_count(0))
        if dist_on and backend == "nccl" and rccl.ENABLED:
            collectives_per_step = rccl.total_calls() - c0
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    capture_error = None
    if use_graph:
        # A failed capture is an error (non-zero exit) at every N, never a silent fall-back to eager launches
        # (--no-graph asks for those): an eager N > 1 figure (~90 ms/step, host bound) printed as the scaling
        # curve's point would be worse than no point.  SSA_BENCH_EAGER_FALLBACK=1 (debugging on a multi-GPU node)
        # lets an N > 1 run go on with eager launches of the SAME program and flags it in the JSON line
        # (config.hipgraph = false, config.capture_error = the exception).
        try:
            graph = torch.cuda.CUDAGraph()
            optim.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                step()
            torch.cuda.synchronize()
        except Exception as e:          # noqa: BLE001
            if world == 1 or os.environ.get("SSA_BENCH_EAGER_FALLBACK", "0") != "1":
                raise
            capture_error = ("%s: %s" % (type(e).__name__, e))[:400]
            print("bench.py: rank %d: hipGraph capture of the N > 1 step failed, running eager: %s"
                  % (rank, capture_error), file=sys.stderr, flush=True)
            graph = None
            torch.cuda.synchronize()

    run = graph.replay if graph is not None else step
    for _ in range(args.warmup):
        run()

    def sync():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(static_loss.item())
    ms = dt / args.steps * 1e3
    # the same program without the hipGraph: what a loop that does not go through semseg_amd.graph_training pays
    # (~840 ctypes launches + Python autograd per step, host bound)
    eager_ms = None
    if graph is not None and args.eager_steps > 0:
        step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.eager_steps):
            step()
        sync()
        eager_ms = (time.perf_counter() - t0) / args.eager_steps * 1e3
    ips = args.batch * world * args.steps / dt
    flop_scale = (args.crop / 1024.0) * (crop_w / 1024.0)

    # What one SyncBN exchange costs on THIS job's communicator: 200 back-to-back in-place all-reduces of a level-sized
    # fp64 record (8 replicas x 2 x 720 channels) on the compute stream, all ranks together; x the step's collective
    # count = the time the step spends in exchanges no compute hides (they sit on the critical path of the level chain).
    coll_us = None
    if dist_on and backend == "nccl" and rccl.ENABLED:
        rec = torch.zeros(8 * 2 * 720, dtype=torch.float64, device="cuda")
        c_ = rccl.comm(0)
        for _ in range(20):
            c_.all_reduce_(rec)
        sync()
        t0 = time.perf_counter()
        for _ in range(200):
            c_.all_reduce_(rec)
        sync()
        coll_us = (time.perf_counter() - t0) / 200 * 1e6
        c_.calls -= 220

    roof = None
    recs = []
    if not args.no_roofline:
        # Per-launch timing, live: every launch of the group-aware kernels (all conv-class kernels,
        # BatchNorm, sums, resampling) of two eager steps is bracketed by HIP events on its stream
        # inside the library (ssa_profile_begin/_end), keyed by kernel instantiation, with the
        # algorithmic flops/bytes of the problems it carries.  Every rank runs the steps (they
        # contain collectives when N > 1); rank 0 reports.
        hb.profile_begin()
        for _ in range(2):
            # eager launches are host bound: park the GPU behind a ~150 ms spin kernel first, so the
            # whole step is queued when it starts executing and the two events around a launch
            # bracket the kernel, not the host's launch gap
            if hasattr(torch.cuda, "_sleep"):
                torch.cuda._sleep(int(3.0e8))
            step()
        recs = hb.profile_end()
    if rank == 0 and recs:
        # an (event, event) bracket costs GPU time by itself (two marker packets); calibrate it on
        # empty brackets queued behind the same kind of spin kernel and take it off every launch
        if hasattr(torch.cuda, "_sleep"):
            torch.cuda._sleep(int(3.0e7))
        cal = []
        for _ in range(200):
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            c1.record()
            cal.append((c0, c1))
        torch.cuda.synchronize()
        empty_us = sorted(a.elapsed_time(b) for a, b in cal)[len(cal) // 2] * 1e3
        fam = {}
        for r in recs:
            t_us = max(r["total_us"] - empty_us * r["launches"], 0.05 * r["launches"])
            a = fam.setdefault(family(r["kernel"]), {"us": 0.0, "launches": 0, "jobs": 0, "flops": 0.0, "bytes": 0.0})
            a["us"] += t_us
            a["launches"] += r["launches"]
            a["jobs"] += r["jobs"]
            a["flops"] += r["flops"]
            a["bytes"] += r["bytes"]
        conv = {k: v for k, v in fam.items() if k in CONV_FAMILIES}
        name, d = max(conv.items(), key=lambda kv: kv[1]["us"])        # conv-class family with the most time
        aname, ad = max(fam.items(), key=lambda kv: kv[1]["us"])        # any family with the most time
        # HBM bytes per launch of that kernel family: PMC counters (FETCH_SIZE, WRITE_SIZE in separate
        # rocprofv3 passes over this same command, corrected as MI355X_MICROARCH.md prescribes),
        # collected offline (rocprofv3 cannot run inside this process) by tools/pmc_traffic.py; refused
        # unless it was measured on these very kernel sources
        traffic, traffic_src, mfma_busy, pmc_fam = None, None, None, {}
        # newest profiles/rNN_pmc.json first
        import glob
        for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")), reverse=True):
            rel = os.path.relpath(pmc, ROOT)
            with open(pmc) as f:
                pj = json.load(f)
            if pj.get("source_sha") != source_sha():
                traffic_src = traffic_src or "%s refused: measured on other kernel sources" % rel
                continue
            ent = pj["kernels"].get(name)
            if ent:
                traffic, traffic_src = ent.get("hbm_bytes_per_launch"), rel
                mfma_busy = ent.get("mfma_busy_frac")
            # MFMA-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles), fabric bytes, LDS conflicts per family
            pmc_fam = {k: {"mfma_busy_frac": v.get("mfma_busy_frac"), "hbm_bytes_per_launch": v.get("hbm_bytes_per_launch"),
                           "lds_conflict_frac": v.get("lds_conflict_frac"), "waves_per_simd": v.get("waves_per_simd")}
                       for k, v in pj["kernels"].items() if k in CONV_FAMILIES}
            break
        # north_star's own criterion: MFMA fraction over the 3x3 convolutions (forward, data and weight gradients) --
        # the kernels that run ONLY 3x3 stride-1 layers: the trunk tile kernels, the head's halo GEMM, the two
        # weight-gradient kernels; algorithmic FLOPs of their jobs / the sum of their launch durations / 2.5 PF.
        # (4.962 TFLOP per image are 3x3; the stride-2 3x3 layers run on ConvIgemm / ConvWgradTr beside
        # 1x1 layers and are left out of both sums.)  Durations are this eager leg's (each launch alone on the chip);
        # inside the replayed step the weight-gradient stream shares the chip and the same kernels run 10-15 % longer.
        k3 = {"us": 0.0, "flops": 0.0, "kernels": {}}
        for r in recs:
            base = r["kernel"].split("<")[0]
            if family(r["kernel"]) == "ConvTile" or base in ("ConvHaloGemm3", "ConvHaloReg3", "ConvWgradHead3", "ConvWgradTile", "ConvWgradTileA"):
                t_us = max(r["total_us"] - empty_us * r["launches"], 0.05 * r["launches"])
                k3["us"] += t_us
                k3["flops"] += r["flops"]
                e = k3["kernels"].setdefault(base, {"ms_per_step": 0.0, "tflop_per_step": 0.0})
                e["ms_per_step"] += t_us / 2e3
                e["tflop_per_step"] += r["flops"] / 2e12
        for e in k3["kernels"].values():
            e["mfma_frac"] = e["tflop_per_step"] * 1e12 / max(e["ms_per_step"] * 1e-3, 1e-9) / PEAK_BF16_MFMA
        mfma_3x3 = {"frac": k3["flops"] / max(k3["us"] * 1e-6, 1e-9) / PEAK_BF16_MFMA if k3["us"] else None,
                    "tflop_per_step": k3["flops"] / 2e12, "ms_per_step": k3["us"] / 2e3, "kernels": k3["kernels"],
                    "target": 0.7}
        sec = d["us"] * 1e-6
        mfma_bound = d["flops"] / max(d["bytes"], 1.0) > PEAK_BF16_MFMA / PEAK_HBM
        roof = {"bound": "mfma" if mfma_bound else "hbm", "kernel": name,
                "achieved": d["flops"] / sec / 1e12 if mfma_bound else d["bytes"] / sec / 1e9,
                "peak": PEAK_BF16_MFMA / 1e12 if mfma_bound else PEAK_HBM / 1e9,
                "unit": "TFLOP/s" if mfma_bound else "GB/s",
                "frac": d["flops"] / sec / PEAK_BF16_MFMA if mfma_bound else d["bytes"] / sec / PEAK_HBM,
                "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                "mfma_busy_frac": mfma_busy, "pmc_conv_families": pmc_fam, "mfma_3x3": mfma_3x3,
                "launches_per_step": d["launches"] // 2, "problems_per_launch": d["jobs"] / d["launches"],
                "avg_launch_us": d["us"] / d["launches"], "flop_per_launch": d["flops"] / d["launches"],
                "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                "mfma_tflops": d["flops"] / sec / 1e12, "mfma_frac": d["flops"] / sec / PEAK_BF16_MFMA,
                "hbm_GBps": d["bytes"] / sec / 1e9, "hbm_frac": d["bytes"] / sec / PEAK_HBM,
                "event_bracket_overhead_us": empty_us,
                "note": "dominant = the conv-class kernel family with the most time in an eager pass (per-launch "
                        "HIP events on the launch stream, grouped launches carry several layers' problems); bound = "
                        "which roof its aggregate arithmetic intensity falls under; both fractions are given",
                "dominant_any_kernel": {"kernel": aname, "ms_per_step": ad["us"] / 2e3, "launches_per_step": ad["launches"] // 2,
                                        "hbm_GBps": ad["bytes"] / (ad["us"] * 1e-6) / 1e9,
                                        "hbm_frac": ad["bytes"] / (ad["us"] * 1e-6) / PEAK_HBM},
                "timed_kernel_share_of_step": sum(v["us"] for v in fam.values()) / 2e3 / ms,
                "families": {k: {"ms_per_step": v["us"] / 2e3, "launches_per_step": v["launches"] // 2,
                                 "problems_per_launch": v["jobs"] / v["launches"],
                                 "tflops": v["flops"] / (v["us"] * 1e-6) / 1e12,
                                 "hbm_GBps": v["bytes"] / (v["us"] * 1e-6) / 1e9}
                             for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["us"])}}
        if os.environ.get("SSA_DUMP_KERNELS"):
            for r in sorted(recs, key=lambda r: -r["total_us"])[:60]:
                print("KERNEL %-70s n/step %4d jobs %5d avg %7.1f us  %6.1f TF/s %6.0f GB/s" % (
                    r["kernel"][:70], r["launches"] // 2, r["jobs"] // 2, r["total_us"] / r["launches"],
                    r["flops"] / max(r["total_us"], 1e-3) / 1e6, r["bytes"] / max(r["total_us"], 1e-3) / 1e3), file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
    # the same step on the other storage build (bf16, no loss scaling), timed in a child process at N = 1: one library
    # build per process.  Reported beside the headline, never as `value`.
    other = None
    if rank == 0 and world == 1 and not args.no_secondary and not dist_on and samp.fp16_storage():
        other = secondary_run(args, "bf16")

    if rank == 0:
        out = {
            "metric": "train images/sec HRNet-OCR-MScale 1024x1024 crop",
            "value": ips, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp16" if samp.fp16_storage() else "bf16", "data": "synthetic",
            # the bf16-storage build of the same step (child process, N = 1 only); `value` is the fp16 step with the
            # dynamic loss scaler inside the captured graph -- the reference's arithmetic (README.md "which number")
            "value_bf16": other["value"] if other else None,
            "ms_per_step_bf16": other.get("ms_per_step") if other else None,
            "secondary_error": other.get("error") if other else None,
            "config": {"workload": "train_cityscapes_sota: HRNet-OCR-MScale two-scale train step, RMI+BCE loss, "
                                   "crop %dx%d, batch %d/GPU, SGD, synthetic Cityscapes-shaped batch, random init"
                                   % (args.crop, crop_w, args.batch),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "hipgraph": graph is not None, "capture_error": capture_error, "loss": loss_val,
                       "eager_ms_per_step": eager_ms,
                       "optimizer": "ssa_sgd_momentum_step" if fused_sgd else "torch.optim.SGD(foreach)",
                       "loss_scale": scaler.loss_scale() if scaler is not None else None,
                       "library_launches_per_step": launches_per_step,
                       "lib_sha": _lib_sha(),
                       # deferred weight gradients: flushed every N layers onto a side stream (a parallel branch of
                       # the captured step); None = on the compute stream at the end of backward
                       "wgrad_side_stream_flush_at": hb._WGRAD_FLUSH_AT if hb._WGRAD_SIDE else None,
                       "collectives_per_step": collectives_per_step,
                       # measured on this job's ranks (above): one level-sized fp64 all-reduce, and that x the count
                       "syncbn_collective_us": coll_us,
                       "collective_ms_per_step": (coll_us * collectives_per_step / 1e3)
                       if (coll_us is not None and collectives_per_step) else None,
                       # gradient exchange: ranges of the arena are all-reduced on a communication stream while backward
                       # runs; what no compute can hide is the LAST range -- estimate = its bytes x 2 (ring all-reduce
                       # traffic per rank) / 300 GB/s of xGMI per GPU
                       "grad_exchanges_per_step": getattr(model, "exchanges", None) if dist_on else None,
                       "exposed_comm_ms_estimate": (getattr(model, "tail_elements", 0) * 4 * 2 / 300e9 * 1e3) if dist_on else None,
                       "logit_tolerance": "north_star asks 1e-3 relative; 16-bit STORAGE of ~450 layers gives ~1.5e-2 "
                                          "end to end in fp16 (this line's format; gradient cosine vs the fp32 oracle 0.99) "
                                          "and ~1e-1 in bf16 on random weights -- the fp32-oracle-with-16-bit-storage "
                                          "emulation gives the same; every op holds one-rounding tolerance teacher-forced "
                                          "at this config (tests/test_parity_1024_gpu.py, DESIGN.md section 4)"},
            "model_flops_util": ips / world * FLOP_FWD_BWD_PER_IMAGE * flop_scale / PEAK_BF16_MFMA,
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist_on:
        rccl.shutdown()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
