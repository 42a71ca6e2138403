#!/bin/bash
# Round 5, call C: first device run of the fused object-attention kernels (csrc/ocr_attn.hip): op-level tests on both
# storage builds, the teacher-forced 1024^2 training step (attention forward + backward at its real shapes), the
# hierarchical evaluation end to end, the step with and without the fused kernels, a one-step trace.
#                                                                                   bash tools/calls/r5c.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r5c}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "ocr or probe or softmax" -s > gpurun_out/${T}_kernels.log 2>&1
echo "attention kernel tests rc=$?: $(tail -1 gpurun_out/${T}_kernels.log)"; grep -h "attn_" gpurun_out/${T}_kernels.log | head -30
timeout 300 env SSA_ACT_DTYPE=fp16 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "ocr" > gpurun_out/${T}_kernels_f16.log 2>&1
echo "fp16 build rc=$?: $(tail -1 gpurun_out/${T}_kernels_f16.log)"
timeout 400 python -m pytest tests/test_parity_1024_gpu.py::test_teacher_forced_ops_at_1024 tests/test_e2e_gpu.py::test_eval_nscale -q -m gpu -s > gpurun_out/${T}_parity.log 2>&1
echo "teacher-forced 1024 + nscale rc=$?: $(tail -1 gpurun_out/${T}_parity.log)"
grep -E "operand ranges|comparisons|ocr_attention|nscale " gpurun_out/${T}_parity.log | head -30
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0"
line() { grep -h '^{' "$1" | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", round(d["config"]["loss"],4))' 2>&1 | tail -1; }
run() { name=$1; shift; timeout 200 env "$@" $B > gpurun_out/${T}_bench_$name.log 2>&1; echo "$name [$*]: $(line gpurun_out/${T}_bench_$name.log)"; }
run default A=1
run unfused SSA_OCR_ATTN_FUSED=0
run default2 A=1
cd /tmp && rm -rf /tmp/prof_$T && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$T -- python $OLDPWD/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --eager-steps 0 > $OLDPWD/gpurun_out/${T}_trace_run.log 2>&1
cd $OLDPWD
CSV=$(find /tmp/prof_$T -name "*kernel_trace.csv" | head -1)
python tools/trace_step.py "$CSV" 60 > gpurun_out/${T}_trace_step.txt 2>&1
head -45 gpurun_out/${T}_trace_step.txt
