#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r4i}
mkdir -p gpurun_out/${T}_prof
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu -x > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?: $(tail -1 gpurun_out/${T}_tests.log)"
timeout 600 env SSA_ACT_DTYPE=fp16 python -m pytest tests/test_parity_eval_gpu.py -q -m gpu -x -s -k "three_scales_small" > gpurun_out/${T}_f16_small.log 2>&1
echo "fp16 small teacher-forced rc=$?"; grep -v "amdgpu.ids\|^$" gpurun_out/${T}_f16_small.log | tail -22
timeout 600 env SSA_ACT_DTYPE=fp16 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "test_conv_fwd_bwd and (case34 or case35 or case36)" > gpurun_out/${T}_f16_conv.log 2>&1
echo "fp16 attn conv cases rc=$?: $(tail -1 gpurun_out/${T}_f16_conv.log)"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
line() { grep -h '^{' "$1" | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", round(d["config"]["loss"],4))' 2>&1 | tail -1; }
timeout 200 $B > gpurun_out/${T}_bench_default.log 2>&1; echo "default: $(line gpurun_out/${T}_bench_default.log)"
timeout 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_prof -o $T -- $B > gpurun_out/${T}_rocprof.log 2>&1
f=$(ls gpurun_out/${T}_prof/*/*kernel_trace.csv gpurun_out/${T}_prof/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_step.py "$f" 70 gpurun_out/${T}_launches.txt > gpurun_out/${T}_trace_step.txt 2>&1
rm -rf gpurun_out/${T}_prof
head -24 gpurun_out/${T}_trace_step.txt
timeout 200 $B > gpurun_out/${T}_bench_default2.log 2>&1; echo "default again: $(line gpurun_out/${T}_bench_default2.log)"
