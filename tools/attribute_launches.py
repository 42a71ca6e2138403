"""Which Python call sites issue the torch-native kernels and device-to-device copies of one
training step?  (profiles/r01_bench_graph_summary.txt: ~350 aten add launches and ~300
hipMemcpyDtoD per step that are not ours.)  Runs one eager step under torch.profiler with Python
stacks and prints, per aten operator that launched device work, the innermost frames inside this
repository.
    python tools/attribute_launches.py [crop] > gpurun_out/attribute_launches.txt"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "semantic-segmentation_amd")]
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402


def main():
    crop = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    net = bench.build_model(1)
    optim = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    images, gts = bench.synth_batch(1, crop, crop, 0, "cuda")
    inputs = {"images": images, "gts": gts}

    def step():
        optim.zero_grad(set_to_none=True)
        loss = net(inputs)
        loss.backward()
        optim.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    sites = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or not ev.kernels:      # only operators that launched device work
            continue
        if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::") and ev.cpu_parent.kernels:
            continue                                                # count the outermost aten op once
        frames = [f for f in (ev.stack or []) if ROOT in f or "autograd" in f][:3]
        kinds = ",".join(sorted({k.name.split("<")[0][:40] for k in ev.kernels}))
        sites[(ev.name, kinds, " <- ".join(f.replace(ROOT + "/", "") for f in frames) or "(engine / no python frame)")] += 1
    total = sum(sites.values())
    print("aten operators that launched device work in one eager step at crop %d: %d" % (crop, total))
    for (name, kinds, where), n in sites.most_common(60):
        print("%5d  %-28s %-60s %s" % (n, name, kinds, where))


if __name__ == "__main__":
    main()
