#!/bin/bash
# Round 5, call G: the OCR concatenation without a copy (ops.cat_slots) and the LDS-staged separable backward of the
# few-channel fp32 resizes on the device: op-level tests, the teacher-forced 1024^2 step, end-to-end evaluation, A/B of
# the resize backward, a one-step trace.        bash tools/calls/r5g.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r5g}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py -q -m gpu -x -k "bilinear or cat_slots or twenty or bn_ or ocr" > gpurun_out/${T}_tests.log 2>&1
echo "op-level tests rc=$?: $(tail -1 gpurun_out/${T}_tests.log)"
timeout 600 python -m pytest tests/test_parity_1024_gpu.py::test_teacher_forced_ops_at_1024 tests/test_e2e_gpu.py tests/test_attnscale_gpu.py \
  "tests/test_amp_fp16_gpu.py::test_reference_loop_with_amp_on_the_fp16_build" -q -m gpu -x > gpurun_out/${T}_parity.log 2>&1
echo "teacher-forced 1024 + e2e + attnscale + amp loop rc=$?: $(tail -1 gpurun_out/${T}_parity.log)"
grep -E "^FAILED|^ERROR" gpurun_out/${T}_parity.log gpurun_out/${T}_tests.log | head
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0"
line() { grep -h '^{' "$1" | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", round(d["config"]["loss"],4))' 2>&1 | tail -1; }
run() { name=$1; shift; timeout 200 env "$@" $B > gpurun_out/${T}_bench_$name.log 2>&1; echo "$name [$*]: $(line gpurun_out/${T}_bench_$name.log)"; }
run default A=1
run gather SSA_BILINEAR_ROWS=0
run default2 A=1
run gather2 SSA_BILINEAR_ROWS=0
cd /tmp && rm -rf /tmp/prof_$T && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$T -- python $OLDPWD/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --eager-steps 0 > $OLDPWD/gpurun_out/${T}_trace_run.log 2>&1
cd $OLDPWD
CSV=$(find /tmp/prof_$T -name "*kernel_trace.csv" | head -1)
python tools/trace_step.py "$CSV" 70 > gpurun_out/${T}_trace_step.txt 2>&1
head -40 gpurun_out/${T}_trace_step.txt
