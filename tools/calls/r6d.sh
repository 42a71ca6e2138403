#!/bin/bash
# Round 6 call D: BatchNorm prologues without fp64 division / square root (the apply with the coefficients GIVEN ran
# 6.5 us against 12.8 with the statistics prologue, call C): kernel tests, bnbench, step A/B against the experiment
# build `bnold` (HEAD's bn.hip), the wide GEMM's per-Cin dispatch.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6d}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -x -m gpu > gpurun_out/${T}_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -3 gpurun_out/${T}_kernel_tests.log
echo "== bnbench new"; timeout 300 python tools/bnbench.py 30 2>&1 | tee gpurun_out/${T}_bnbench.txt | grep -v amdgpu.ids
echo "== bnbench old"; timeout 300 python tools/libvariant.py bnold tools/bnbench.py 30 2>&1 | tee gpurun_out/${T}_bnbench_old.txt | grep -v amdgpu.ids
echo "== headbench 1x1"; timeout 300 python tools/headbench.py 20 --only-1x1 --fwd-only 2>&1 | tee gpurun_out/${T}_headbench.txt | grep 1x1
for v in new old new old; do
  if [ $v = new ]; then
    timeout 400 python bench.py --dtype bf16 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_bf16_$v.json 2> gpurun_out/${T}_bench.err
  else
    timeout 400 python tools/libvariant.py bnold bench.py --dtype bf16 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${T}_bench_bf16_$v.json 2> gpurun_out/${T}_bench.err
  fi
  python - <<PY
import json
j = json.load(open("gpurun_out/${T}_bench_bf16_$v.json"))
f = j["roofline"]["families"]
print("$v: ms", round(j["ms_per_step"], 3), "loss", j["config"]["loss"], " ".join("%s %.3f" % (k, v["ms_per_step"]) for k, v in f.items() if k.startswith("Bn")))
PY
done
timeout 400 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_bench_fp16.json 2>> gpurun_out/${T}_bench.err
python -c "
import json; j=json.load(open('gpurun_out/${T}_bench_fp16.json')); print('fp16 headline', round(j['ms_per_step'],3), 'bf16 child', j.get('ms_per_step_bf16'))"
