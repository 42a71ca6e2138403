#!/bin/bash
# Round 6 call C: the wide 1x1 GEMM with 64-channel stages (whole 128-byte lines, double buffer) against the 32-channel
# ring (experiment build w32) and the 256 x 128 kernel; BatchNorm apply with / without the statistics prologue; the new
# evaluation-graph test; bench lines.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6c}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "gemm_wide or conv_fwd_bwd" > gpurun_out/${T}_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -3 gpurun_out/${T}_kernel_tests.log
timeout 600 python -m pytest tests/test_graphed_step_gpu.py -q -x -m gpu -k "graph_eval" > gpurun_out/${T}_graph_eval_test.log 2>&1
echo "graph_eval test rc=$?"; tail -12 gpurun_out/${T}_graph_eval_test.log
echo "== headbench, wide CK=64 RING=2"; timeout 300 python tools/headbench.py 20 --only-1x1 --fwd-only 2>&1 | tee gpurun_out/${T}_headbench_w64.txt | grep 1x1
echo "== headbench, wide CK=32 RING=4"; timeout 300 python tools/headbench.py 20 --only-1x1 --fwd-only --lib w32 2>&1 | tee gpurun_out/${T}_headbench_w32.txt | grep 1x1
echo "== headbench, wide off"; SSA_GEMM_WIDE=0 timeout 300 python tools/headbench.py 20 --only-1x1 --fwd-only 2>&1 | tee gpurun_out/${T}_headbench_old.txt | grep 1x1
echo "== bnbench"; timeout 300 python tools/bnbench.py 30 2>&1 | tee gpurun_out/${T}_bnbench.txt | tail -12
for wide in 1 0 1 0; do
  SSA_GEMM_WIDE=$wide timeout 400 python bench.py --dtype bf16 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_bf16_wide${wide}.json 2> gpurun_out/${T}_bench.err
  python - <<PY
import json
j = json.load(open("gpurun_out/${T}_bench_bf16_wide${wide}.json"))
f = j["roofline"]["families"]
print("wide=${wide}: ms", round(j["ms_per_step"], 3), "ConvHaloGemm", round(f.get("ConvHaloGemm", {}).get("ms_per_step", 0), 3),
      "ConvGemmWide", round(f.get("ConvGemmWide", {}).get("ms_per_step", 0), 3),
      "BN", round(sum(v["ms_per_step"] for k, v in f.items() if k.startswith("Bn")), 3))
PY
done
echo "== eval bench (graph_eval vs eager)"; timeout 600 python tools/eval_bench.py 10 2>gpurun_out/${T}_eval.err | tee gpurun_out/${T}_eval_bench.json | cut -c1-330
