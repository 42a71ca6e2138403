#!/bin/bash
# Weight gradients on a side stream: step-level tests (eager, graphed, DDP-in-graph), then bench.py per flush interval.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-side}
log=gpurun_out/$T.log
: > "$log"
[ -n "$NO_TESTS" ] || timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_graphed_step_gpu.py tests/test_ddp_graph_gpu.py tests/test_ddp_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu -x > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?: $(tail -1 gpurun_out/${T}_tests.log)" >> "$log"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0"
line() { grep -h '^{' "$1" | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", round(d["config"]["loss"],4))' 2>&1 | tail -1; }
for spec in ${SPECS:-default= off=SSA_WGRAD_STREAM=0 default2=}; do
  name=${spec%%=*}; envs=${spec#*=}
  timeout 200 env ${envs//,/ } $B > gpurun_out/${T}_bench_$name.log 2>&1; echo "$name [$envs]: $(line gpurun_out/${T}_bench_$name.log)" >> "$log"
done
cat "$log"
