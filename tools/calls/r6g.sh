#!/bin/bash
# Round 6 call G: the all-taps weight-gradient kernel (ConvWgradTileA: 14 / 13 n-blocks x 3 m-blocks per 4-wave workgroup,
# LDS-DMA double buffer, one barrier per tile) and the strip fitting at the true residency (one workgroup per CU).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6g}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -x -m gpu > gpurun_out/${T}_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -5 gpurun_out/${T}_kernel_tests.log
for cfg in "1 256" "0 256" "0 512" "1 256" "0 512"; do
  set -- $cfg
  SSA_WGRAD_ALL=$1 SSA_WGRAD_SLOTS=$2 timeout 400 python bench.py --dtype bf16 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_all$1_slots$2.json 2> gpurun_out/${T}_bench.err
  python - <<PY
import json
j = json.load(open("gpurun_out/${T}_bench_all$1_slots$2.json"))
f = j["roofline"]["families"]
print("all=$1 slots=$2: ms", round(j["ms_per_step"], 3), "loss", round(j["config"]["loss"], 3), " ".join("%s %.3f(%d)" % (k, v["ms_per_step"], v["launches_per_step"]) for k, v in f.items() if "Wgrad" in k))
PY
done
timeout 300 python -m pytest tests/test_parity_1024_gpu.py -q -x -m gpu > gpurun_out/${T}_parity1024.log 2>&1; echo "parity1024 rc=$?"; tail -3 gpurun_out/${T}_parity1024.log
