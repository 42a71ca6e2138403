#!/bin/bash
# Round 5, call F: 32 layers per grouped weight-gradient launch (csrc/group.h MAXJOBS) against 16 on one box, plus the
# tests touched since the full suite of call S1.        bash tools/calls/r5f.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r5f}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py "tests/test_amp_fp16_gpu.py::test_reference_loop_with_amp_on_the_fp16_build" \
  "tests/test_attnscale_gpu.py" -q -m gpu -x -k "wgrad or twenty or group or amp or attnscale" > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?: $(tail -1 gpurun_out/${T}_tests.log)"
grep -E "^FAILED|^ERROR" gpurun_out/${T}_tests.log | head
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0"
line() { grep -h '^{' "$1" | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", round(d["config"]["loss"],4))' 2>&1 | tail -1; }
run() { name=$1; shift; timeout 200 env "$@" $B > gpurun_out/${T}_bench_$name.log 2>&1; echo "$name [$*]: $(line gpurun_out/${T}_bench_$name.log)"; }
run jobs32 A=1
run jobs16 SSA_GROUP_JOBS=16
run jobs32b A=1
run jobs16b SSA_GROUP_JOBS=16
run jobs32_strip12 SSA_WGRAD_STRIP=12
