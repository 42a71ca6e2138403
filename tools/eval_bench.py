#!/usr/bin/env python
"""Secondary measurement (not the headline metric): inference throughput of the three eval
configurations of BASELINE.json on one MI355X, synthetic input, random weights -- through `semseg_amd.graph_eval` (the
captured forward a drop-in user's validate() loop replays per image) with the eager call of the bare module beside it:
  configs[1]  HRNet-OCR single scale, 1024x2048                     (ocrnet.HRNet)
  configs[2]  HRNet-OCR-MScale hierarchical attention {0.5,1.0,2.0}  (ocrnet.HRNet_Mscale, N_SCALES)
  configs[4]  Mapillary: 65 classes, {0.5,1.0,2.0} on a 1536x2048 image (the 2.0x pass is 3072x4096), one GPU --
              BASELINE.json asks for fp16: run with SSA_ACT_DTYPE=fp16 (the row says which storage format ran)
usage: python tools/eval_bench.py [iters] [mapillary H W]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "semantic-segmentation_amd")]
import torch  # noqa: E402


def run(name, arch, n_scales, H, W, iters, classes=19, use_graph=True):
    """One row: the forward the reference's validate() loop calls (`net(inputs)` under no_grad) through
    `semseg_amd.graph_eval` -- what a drop-in user gets -- and the same call on the bare module (eager launches)."""
    from semseg_amd.config import cfg
    from semseg_amd.loss import CrossEntropyLoss2d
    from semseg_amd.network import get_model
    from semseg_amd.graphed import graph_eval
    cfg.MODEL.N_SCALES = n_scales
    cfg.MODEL.BNFUNC = None
    torch.manual_seed(0)
    cfg.DATASET.NUM_CLASSES = classes
    net = get_model(arch, classes, CrossEntropyLoss2d(ignore_index=255)).cuda().eval()
    images = torch.randn(1, 3, H, W, device="cuda")
    inputs = {"images": images}
    gnet = graph_eval(net, max_graphs=2, clone_outputs=True)
    out_buf = {}

    def timed(model, n):
        with torch.no_grad():
            for _ in range(2):
                out_buf["pred"] = model(inputs)["pred"]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                out_buf["pred"] = model(inputs)["pred"]
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    eager = timed(net, max(2, iters // 3))
    dt = timed(gnet if use_graph else net, iters)
    ev = gnet._eval_stepper
    pred = out_buf["pred"]
    from semseg_amd import _lib
    res = {"config": name, "arch": arch, "storage": _lib.ACT, "scales": n_scales or [1.0], "input": [H, W], "classes": classes,
           "ms_per_image": dt * 1e3, "eager_ms_per_image": eager * 1e3,
           "images_per_s": 1.0 / dt, "hipgraph": bool(use_graph and ev is not None and not ev.eager_only and ev.replays > 0),
           "api": "semseg_amd.graph_eval (clone_outputs=True)", "pred_shape": list(pred.shape),
           "finite": bool(torch.isfinite(pred).all()), "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}
    print(json.dumps(res))
    del net, gnet
    torch.cuda.empty_cache()


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    if len(sys.argv) > 2 and sys.argv[2] == "mapillary":
        # BASELINE configs[4]: Mapillary-sized full-resolution {0.5,1.0,2.0} inference, 65 classes,
        # on ONE GPU (the reference needs amp O3 to fit); H W from argv
        H, W = int(sys.argv[3]), int(sys.argv[4])
        run("configs[4] Mapillary-sized {0.5,1.0,2.0} eval", "ocrnet.HRNet_Mscale", [0.5, 1.0, 2.0], H, W, iters,
            classes=65)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "mapillary-ref":
        # the reference's own recipe, scripts/eval_mapillary.yml:13-18: pre_size 2177 (a 4:3 Mapillary image becomes
        # 1632 x 2177), n_scales 0.25,0.5,1.0,2.0 (the 2.0x pass is 3264 x 4354 = 14.2 Mpixel), fp16 -- on ONE GPU
        run("configs[4] eval_mapillary.yml: pre_size 2177, {0.25,0.5,1.0,2.0}", "ocrnet.HRNet_Mscale",
            [0.25, 0.5, 1.0, 2.0], 1632, 2177, iters, classes=65)
        sys.exit(0)
    run("configs[1] HRNet-OCR single-scale eval", "ocrnet.HRNet", None, 1024, 2048, iters)
    if len(sys.argv) > 2 and sys.argv[2] == "c1":          # (profiling runs: the single-scale row only)
        sys.exit(0)
    run("configs[2] HRNet-OCR-MScale {0.5,1.0,2.0} eval", "ocrnet.HRNet_Mscale", [0.5, 1.0, 2.0], 1024, 2048, iters)
    run("configs[4] Mapillary 65 classes {0.5,1.0,2.0} eval", "ocrnet.HRNet_Mscale", [0.5, 1.0, 2.0], 1536, 2048, iters,
        classes=65)
