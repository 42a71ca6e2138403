#!/bin/bash
# Round 6 call F: BatchNorm backward reduce restructured (centring at the end, mask source compile-time): 4 vs 8 rows.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6f}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -x -m gpu > gpurun_out/${T}_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -3 gpurun_out/${T}_kernel_tests.log
for r in 8 4; do
echo "== bnbench reduce rows $r"; SSA_BN_ROWS_REDUCE=$r timeout 300 python tools/bnbench.py 30 2>&1 | tee gpurun_out/${T}_bnbench_r$r.txt | grep "reduce"
done
for r in 8 4 8 4; do
  SSA_BN_ROWS_REDUCE=$r timeout 400 python bench.py --dtype bf16 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_bf16_r$r.json 2> gpurun_out/${T}_bench.err
  python - <<PY
import json
j = json.load(open("gpurun_out/${T}_bench_bf16_r$r.json"))
f = j["roofline"]["families"]
print("reduce rows $r: ms", round(j["ms_per_step"], 3), "loss", j["config"]["loss"], " ".join("%s %.3f" % (k, v["ms_per_step"]) for k, v in f.items() if k.startswith("Bn")))
PY
done
