#!/bin/bash
# Round 5, call E: fp16 training (dynamic loss scaling on the device, semseg_amd/amp.py + csrc/optim.hip) on the device:
# the scaler tests, the fp16-build training parity tests, a bench line of the fp16 build.    bash tools/calls/r5e.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r5e}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_amp_fp16_gpu.py tests/test_optim_gpu.py -q -m gpu --durations=10 > gpurun_out/${T}_tests.log 2>&1
echo "amp / optimizer tests rc=$?: $(tail -1 gpurun_out/${T}_tests.log)"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/${T}_tests.log | head
grep -hE "train loss|grad cosine|LOSSES|comparisons|operand ranges" gpurun_out/fp16_e2e_train.log gpurun_out/fp16_amp_loop.log gpurun_out/fp16_teacher_train.log 2>/dev/null | head
timeout 300 env SSA_ACT_DTYPE=fp16 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --eager-steps 0 > gpurun_out/${T}_bench_fp16.log 2>&1
grep -h '^{' gpurun_out/${T}_bench_fp16.log > gpurun_out/${T}_bench_fp16.json
python -c 'import json; d=json.load(open("gpurun_out/'${T}'_bench_fp16.json")); print("fp16 build:", round(d["ms_per_step"],2), "ms", d["dtype"], "loss", round(d["config"]["loss"],4), "scale", d["config"]["loss_scale"])' 2>&1 | tail -1
tail -3 gpurun_out/${T}_bench_fp16.log | cut -c1-300
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0 2>&1 | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("bf16 build:", round(d["ms_per_step"],2), "ms", d["dtype"])'
