#!/bin/bash
# Round 5, call H: the one test the last full suite (call S2: 181 passed, 1 failed) was red on, at its calibrated crop;
# a second bench line + one-step trace of the final library (the evidence call drew a slow box).   bash tools/calls/r5h.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r5h}
mkdir -p gpurun_out
timeout 400 python -m pytest "tests/test_amp_fp16_gpu.py::test_teacher_forced_training_ops_on_the_fp16_build" -q -m gpu > gpurun_out/${T}_test.log 2>&1
echo "fp16 teacher-forced training at 1024 rc=$?: $(tail -1 gpurun_out/${T}_test.log)"
grep -hE "comparisons|operand ranges" gpurun_out/fp16_teacher_train.log | head -3
timeout 400 python bench.py > gpurun_out/${T}_bench.log 2>&1
grep -h '^{' gpurun_out/${T}_bench.log > gpurun_out/${T}_bench_line.json
python -c 'import sys,json; d=json.load(open(sys.argv[1])); print(round(d["ms_per_step"],2), "ms", round(d["value"],2), "img/s; roofline frac", round(d["roofline"]["frac"],4), "traffic", d["roofline"]["traffic"], "sha", d["config"]["lib_sha"])' gpurun_out/${T}_bench_line.json
mkdir -p gpurun_out/${T}_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -o $T -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0 > gpurun_out/${T}_rocprof.log 2>&1
f=$(ls gpurun_out/${T}_prof/*/*kernel_trace.csv gpurun_out/${T}_prof/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_step.py "$f" 110 gpurun_out/${T}_launches.txt > gpurun_out/${T}_trace_step.txt 2>&1
cp $(ls gpurun_out/${T}_prof/*/*kernel_stats.csv gpurun_out/${T}_prof/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/${T}_prof
head -16 gpurun_out/${T}_trace_step.txt
