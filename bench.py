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

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# SURVEY.md section 8d / BASELINE.md section 2: conv FLOPs per training image at
# 1024x1024 (two scales, 638 convs), measured from the reference on `meta`.
FLOP_FWD_BWD_PER_IMAGE = 5.7274e12
PEAK_BF16_MFMA = 2.5e15          # dense, MI355X_MICROARCH.md
TILE_NAMES = {0: "conv_igemm_kernel<2,2,2,2> (128x128)", 1: "conv_igemm_kernel<4,1,2,2> (256x64)",
              2: "conv_igemm_kernel<4,1,1,3> (128x96)", 3: "conv_igemm_kernel<4,1,2,1> (256x32)",
              4: "conv_igemm_kernel<2,2,1,1> (64x64)", 5: "conv_igemm_kernel<2,2,2,1> (128x64)",
              100: "conv_tile_kernel (halo tile, 128/256 px x Cout)",
              101: "conv_halo_gemm_kernel (256 px x 128 ch, halo chunks)", -1: "conv_wgrad_tr_kernel",
              102: "conv_wgrad_head_kernel (128 co x 128 ci x 3 kw, persistent)"}


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


def cpu_baseline(timeout_s=150):
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


def _cpu_baseline_impl(crop=256, timed=2):
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
    for n in sorted({min(avail, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(n)
        one_iter(128)
        t = one_iter(128)
        if best_t is None or t < best_t:
            best_n, best_t = n, t
    ncores = best_n
    torch.set_num_threads(ncores)
    times = [one_iter(crop) for _ in range(1 + timed)]
    per_iter = sum(times[1:]) / timed
    scale = (1024.0 / crop) ** 2
    return {"value": 1.0 / (per_iter * scale), "unit": "images/s", "cores": ncores, "kind": "port",
            "sample": "oracle (CPU port of the reference modules) fwd+bwd, fp32, %d timed iters after 1 warm-up at "
                      "%dx%d crop (%.3f s/iter) on %d threads (best of 8/16/32/64; %d CPUs visible); value = that "
                      "rate / %.0f (pixel-count ratio to 1024x1024)"
                      % (timed, crop, crop, per_iter, ncores, avail, scale)}


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    # SSA_BENCH_ONE_DEVICE / SSA_DIST_BACKEND: self-test of the N > 1 code path on a one-GPU box
    # (all ranks on cuda:0, gloo collectives); the driver's multi-GPU runs use neither.
    torch.cuda.set_device(0 if os.environ.get("SSA_BENCH_ONE_DEVICE") else local_rank)
    dist_on = world > 1 or FORCE_DIST
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend=os.environ.get("SSA_DIST_BACKEND", "nccl"), init_method="env://")

    net = build_model(world)
    model = net
    if dist_on:
        from semseg_amd.parallel import DistributedDataParallel
        from semseg_amd import ops as sops
        # N > 1 runs eager (no hipGraph around RCCL calls): the step is host-launch bound, where
        # the second stream buys nothing -- keep every collective on one stream
        sops.backend().concurrency = 0
        model = DistributedDataParallel(net)
    # SGD + momentum + weight decay as in loss/optimizer.py:47-53.  SSA_FUSED_SGD=1: the one-pass
    # HIP step (ssa_sgd_momentum_step); default: torch's multi-tensor SGD
    fused_sgd = os.environ.get("SSA_FUSED_SGD", "0") == "1"
    if fused_sgd:
        from semseg_amd.loss.optimizer import FusedSGD
        optim = FusedSGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    else:
        optim = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    crop_w = args.crop_w or args.crop
    images, gts = synth_batch(args.batch, args.crop, crop_w, rank, "cuda")
    inputs = {"images": images, "gts": gts}
    static_loss = torch.zeros((), device="cuda")

    def step():
        optim.zero_grad(set_to_none=True)
        loss = model(inputs)
        loss.backward()
        optim.step()
        static_loss.copy_(loss.detach())

    graph = None
    # N > 1: the captured graph would contain ~1,270 RCCL collectives (SyncBN exchanges, gradient
    # buckets); opt-in (SSA_DDP_GRAPH=1) until that has run on a multi-GPU node
    use_graph = (not args.no_graph) and (not dist_on or os.environ.get("SSA_DDP_GRAPH", "0") == "1")
    graph_error = None
    if use_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            optim.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                step()
            torch.cuda.synchronize()
        except Exception as e:  # fall back to eager launches
            graph_error = repr(e)[:200]
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
    ips = args.batch * world * args.steps / dt
    flop_scale = (args.crop / 1024.0) * (crop_w / 1024.0)

    roof = None
    store = []
    if not args.no_roofline:
        # every rank runs the two profiled steps (they contain collectives when N > 1); only rank 0
        # attaches per-launch HIP events
        from semseg_amd import hip_backend as hb, ops as sops
        be = sops.backend()
        saved_conc, be.concurrency = be.concurrency, 0   # one stream: a launch's events bracket only that launch
        if rank == 0:
            hb.set_profile(store)
        for _ in range(2):
            # eager launches are host bound (~17 us of Python per launch): park the GPU behind a
            # ~150 ms spin kernel first, so the whole step is queued when it starts executing and
            # the two events around a launch bracket the kernel, not the host's launch gap
            if hasattr(torch.cuda, "_sleep"):
                torch.cuda._sleep(int(3.0e8))
            step()
        torch.cuda.synchronize()
        hb.set_profile(None)
        be.concurrency = saved_conc
    if rank == 0 and store:
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
        empty = sorted(a.elapsed_time(b) for a, b in cal)[len(cal) // 2] * 1e-3
        agg = {}
        for kind, tile, flops, e0, e1, shape in store:
            key = (kind, tile)
            a = agg.setdefault(key, [0.0, 0.0, 0, 0.0])
            a[0] += flops
            a[1] += max(e0.elapsed_time(e1) * 1e-3 - empty, 1e-7)
            a[2] += 1
            # algorithmic HBM bytes of the launch: bf16 input + output once,
            # the filter once (fp32 dW for a weight gradient); batch = args.batch
            kk, st, ci, co, ho, wo = shape
            pix = args.batch * ho * wo
            a[3] += 2.0 * (pix * st * st * ci + pix * co) + (4.0 if kind.startswith("wgrad") else 2.0) * co * ci * kk * kk
        if os.environ.get("SSA_DUMP_SHAPES"):
            per = {}
            for kind, tile, flops, e0, e1, shape in store:
                a = per.setdefault((kind, tile) + tuple(shape), [0.0, 0.0, 0])
                a[0] += flops
                a[1] += e0.elapsed_time(e1) * 1e-3
                a[2] += 1
            for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:60]:
                print("SHAPE %-6s tile %4d k%d s%d cin %4d cout %4d out %4dx%-4d  n/step %3d  avg %7.1f us  %6.1f TF/s  %.3f ms/step"
                      % (k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], v[2] // 2, v[1] / v[2] * 1e6,
                         v[0] / v[1] / 1e12, v[1] / 2 * 1e3), file=sys.stderr)
        tot_t = sum(a[1] for a in agg.values())
        dom = max(agg.items(), key=lambda kv: kv[1][1])      # most time
        (kind, tile), (fl, tt, n, by) = dom
        name = TILE_NAMES.get(tile, kind)
        # HBM bytes per launch of that kernel family: PMC counters (FETCH_SIZE, WRITE_SIZE in separate
        # rocprofv3 passes over this same command, corrected as MI355X_MICROARCH.md prescribes) --
        # collected offline into profiles/ (rocprofv3 cannot run inside this process)
        traffic, traffic_src = None, None
        fam = {"wgrad": "conv_wgrad_tr_kernel", "tile": "conv_tile_kernel", "halo": "conv_halo_gemm_kernel",
               "igemm": "conv_igemm_kernel", "wgrad_head": "conv_wgrad_head_kernel"}.get(kind)
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if fam and os.path.exists(pmc):
            with open(pmc) as f:
                ent = json.load(f)["kernels"].get(fam)
            if ent:
                traffic, traffic_src = ent["hbm_bytes_per_launch"], "profiles/r01_pmc_traffic.json"
        roof = {"bound": "mfma", "kernel": name, "achieved": fl / tt / 1e12, "peak": PEAK_BF16_MFMA / 1e12,
                "unit": "TFLOP/s", "frac": fl / tt / PEAK_BF16_MFMA, "traffic": traffic,
                "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                "launches_per_step": n // 2, "avg_launch_us": tt / n * 1e6,
                "flop_per_launch": fl / n, "algorithmic_bytes_per_launch": by / n,
                "hbm_achieved_GBps": by / tt / 1e9, "hbm_frac_of_8TBps": by / tt / 8e12,
                "event_bracket_overhead_us": empty * 1e6,
                "note": "dominant = the conv-class kernel family with the most time in an eager, single-stream "
                        "pass (per-launch HIP events on the launch stream); its layers are a mix of HBM-bound "
                        "(48 ch: 216 FLOP/B) and MFMA-bound shapes, both fractions are given",
                "gemm_time_share_of_step": tot_t / 2 / (ms * 1e-3),
                "all_gemm_kernels": {("%s/%s" % (k[0], TILE_NAMES.get(k[1], "-"))): {
                    "tflops": v[0] / v[1] / 1e12, "hbm_GBps": v[3] / v[1] / 1e9, "ms_per_step": v[1] / 2 * 1e3,
                    "launches_per_step": v[2] // 2}
                    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        out = {
            "metric": "train images/sec HRNet-OCR-MScale 1024x1024 crop",
            "value": ips, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "train_cityscapes_sota: HRNet-OCR-MScale two-scale train step, RMI+BCE loss, "
                                   "crop %dx%d, batch %d/GPU, SGD, synthetic Cityscapes-shaped batch, random init"
                                   % (args.crop, crop_w, args.batch),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "hipgraph": graph is not None, "loss": loss_val,
                       "optimizer": "ssa_sgd_momentum_step" if fused_sgd else "torch.optim.SGD(foreach)"},
            "model_flops_util": ips / world * FLOP_FWD_BWD_PER_IMAGE * flop_scale / PEAK_BF16_MFMA,
            "roofline": roof, "cpu_baseline": cpu,
        }
        if graph_error:
            out["config"]["hipgraph_error"] = graph_error
        print(json.dumps(out))
    if dist_on:
        from semseg_amd import rccl
        rccl.shutdown()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
