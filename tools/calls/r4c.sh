#!/bin/bash
# Round 4: trunk-conv experiments -- phase stamps (timing build) and the level micro-benchmark in the default and the
# small (one n-block per workgroup, 3 workgroups per CU) mode, then bench.py in both.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r4c}
mkdir -p gpurun_out
log=gpurun_out/$T.log
: > "$log"
for m in 0 1; do
  echo "==== SSA_TILE_P_SMALL=$m" >> "$log"
  timeout 200 env SSA_TILE_P_SMALL=$m python tools/tilebench.py --timing 2>&1 | grep -v "amdgpu.ids" >> "$log"
  timeout 200 env SSA_TILE_P_SMALL=$m python tools/tilebench.py 20 2>&1 | grep -v "amdgpu.ids" >> "$log"
done
timeout 300 env SSA_TILE_P_SMALL=1 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py -q -m gpu -x -k "conv or block" > gpurun_out/${T}_tests.log 2>&1
echo "small-mode tests rc=$?: $(tail -1 gpurun_out/${T}_tests.log)" >> "$log"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
line() { grep -h '^{' "$1" | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", round(d["config"]["loss"],4))' 2>&1 | tail -1; }
timeout 200 $B > gpurun_out/${T}_bench_default.log 2>&1; echo "default: $(line gpurun_out/${T}_bench_default.log)" >> "$log"
for w in 600 750 900; do
  timeout 200 env SSA_TILE_P_SMALL=1 SSA_TILE_P_WGS=$w $B > gpurun_out/${T}_bench_small$w.log 2>&1; echo "small wgs $w: $(line gpurun_out/${T}_bench_small$w.log)" >> "$log"
done
cat "$log"
