#!/bin/bash
# Round 5, call B: does the weight-gradient branch overlap when it leaves room?  Step-level A/B on one box:
# one weight-gradient workgroup per CU (SSA_WGRAD_SLOTS=256), the main branch captured on a high-priority stream
# (SSA_MAIN_PRIO=-1), both; plus a one-step trace of the default.      bash tools/calls/r5b.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r5b}
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0"
line() { grep -h '^{' "$1" | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", round(d["config"]["loss"],4))' 2>&1 | tail -1; }
run() { name=$1; shift; timeout 200 env "$@" $B > gpurun_out/${T}_bench_$name.log 2>&1; echo "$name [$*]: $(line gpurun_out/${T}_bench_$name.log)"; }
run default A=1
run slots256 SSA_WGRAD_SLOTS=256
run prio SSA_MAIN_PRIO=-1
run slots256prio SSA_WGRAD_SLOTS=256 SSA_MAIN_PRIO=-1
run slots384 SSA_WGRAD_SLOTS=384
run nostream SSA_WGRAD_STREAM=0
run default2 A=1
