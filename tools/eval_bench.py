#!/usr/bin/env python
"""Secondary measurement (not the headline metric): inference throughput of the three eval
configurations of BASELINE.json on one MI355X, synthetic input, random weights, hipGraph replay:
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
    from semseg_amd.config import cfg
    from semseg_amd.loss import CrossEntropyLoss2d
    from semseg_amd.network import get_model
    cfg.MODEL.N_SCALES = n_scales
    cfg.MODEL.BNFUNC = None
    torch.manual_seed(0)
    cfg.DATASET.NUM_CLASSES = classes
    net = get_model(arch, classes, CrossEntropyLoss2d(ignore_index=255)).cuda().eval()
    images = torch.randn(1, 3, H, W, device="cuda")
    inputs = {"images": images}
    out_buf = {}

    def step():
        with torch.no_grad():
            o = net(inputs)
        out_buf["pred"] = o["pred"]

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = None
    try:
        if not use_graph:
            raise RuntimeError("eager requested")
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            step()
        torch.cuda.synchronize()
    except Exception as e:
        print("graph capture failed, eager:", repr(e)[:120], file=sys.stderr)
        graph = None
    f = graph.replay if graph is not None else step
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    pred = out_buf["pred"]
    from semseg_amd import _lib
    res = {"config": name, "arch": arch, "storage": _lib.ACT, "scales": n_scales or [1.0], "input": [H, W], "classes": classes,
           "ms_per_image": dt * 1e3,
           "images_per_s": 1.0 / dt, "hipgraph": graph is not None, "pred_shape": list(pred.shape),
           "finite": bool(torch.isfinite(pred).all()), "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}
    print(json.dumps(res))
    del net, graph
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
    run("configs[2] HRNet-OCR-MScale {0.5,1.0,2.0} eval", "ocrnet.HRNet_Mscale", [0.5, 1.0, 2.0], 1024, 2048, iters)
    run("configs[4] Mapillary 65 classes {0.5,1.0,2.0} eval", "ocrnet.HRNet_Mscale", [0.5, 1.0, 2.0], 1536, 2048, iters,
        classes=65)
