#!/bin/bash
# Round 4: fragment-ring depth of the trunk conv (experiment builds rd3 / rd6 against the product's 4), small mode.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
export SSA_TILE_P_SMALL=1
T=${1:-r4e}
mkdir -p gpurun_out
log=gpurun_out/$T.log
: > "$log"
B="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
line() { grep -h '^{' "$1" | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", round(d["config"]["loss"],4))' 2>&1 | tail -1; }
for v in rd3 rd6; do
  echo "==== $v" >> "$log"
  timeout 200 python tools/libvariant.py $v tools/tilebench.py 20 2>&1 | grep -v "amdgpu.ids" | grep "384 @  32\|192 @  64\|96 @ 128\|48 @ 256\|units [3-5]:\|stage-2" >> "$log"
  timeout 200 python tools/libvariant.py $v $B > gpurun_out/${T}_bench_$v.log 2>&1; echo "$v: $(line gpurun_out/${T}_bench_$v.log)" >> "$log"
done
timeout 200 python $B > gpurun_out/${T}_bench_rd4.log 2>&1; echo "rd4 (product): $(line gpurun_out/${T}_bench_rd4.log)" >> "$log"
cat "$log"
