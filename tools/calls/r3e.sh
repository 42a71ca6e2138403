#!/bin/bash
# Round 3, call E: bench A/B after the strip-length rule (one round of workgroups) and the leaner fetch / MFMA ring.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=r3e
log=gpurun_out/$T.log
: > "$log"
run() { local name=$1 t=$2; shift 2; timeout "$t" "$@" > "gpurun_out/${T}_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
run old 150 env SSA_TILE_P=0 SSA_BLOCK_FOLD=0 $B
run p 150 env SSA_BLOCK_FOLD=0 $B
run fold 150 $B
run p_w448 150 env SSA_BLOCK_FOLD=0 SSA_TILE_P_WGS=448 $B
run p_w480 150 env SSA_BLOCK_FOLD=0 SSA_TILE_P_WGS=480 $B
run p_w512 150 env SSA_BLOCK_FOLD=0 SSA_TILE_P_WGS=512 $B
run p_w384 150 env SSA_BLOCK_FOLD=0 SSA_TILE_P_WGS=384 $B
for n in old p fold p_w448 p_w480 p_w512 p_w384; do
  echo "$n: $(grep -h '^{' gpurun_out/${T}_$n.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", d["config"]["loss"])' 2>&1 | tail -1)" >> "$log"
done
cat "$log"
