#!/bin/bash
# Round 6 call B: first device run of the wide 1x1 GEMM (csrc/conv_gemm_wide.hip) and of the two-round-trip problem
# search of the grouped launches (csrc/group.h): kernel tests, head micro-benchmark with the wide kernel on / off,
# BatchNorm micro-benchmark, a bf16 bench line with the wide kernel on / off.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6b}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py -q -x -m gpu > gpurun_out/${T}_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -4 gpurun_out/${T}_kernel_tests.log
echo "== headbench, wide on"; timeout 300 python tools/headbench.py 20 --only-1x1 --fwd-only 2>&1 | tee gpurun_out/${T}_headbench_wide.txt | tail -12
echo "== headbench, wide off"; SSA_GEMM_WIDE=0 timeout 300 python tools/headbench.py 20 --only-1x1 --fwd-only 2>&1 | tee gpurun_out/${T}_headbench_old.txt | tail -12
echo "== bnbench"; timeout 300 python tools/bnbench.py 30 2>&1 | tee gpurun_out/${T}_bnbench.txt | tail -12
for wide in 1 0; do
  SSA_GEMM_WIDE=$wide timeout 400 python bench.py --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_bf16_wide${wide}.json 2> gpurun_out/${T}_bench.err
  python - <<PY
import json
j = json.load(open("gpurun_out/${T}_bench_bf16_wide${wide}.json"))
f = j["roofline"]["families"]
print("wide=${wide}: ms", round(j["ms_per_step"], 3), "ConvHaloGemm", round(f.get("ConvHaloGemm", {}).get("ms_per_step", 0), 3),
      "ConvGemmWide", round(f.get("ConvGemmWide", {}).get("ms_per_step", 0), 3),
      "BN", round(sum(v["ms_per_step"] for k, v in f.items() if k.startswith("Bn")), 3))
PY
done
