#!/bin/bash
# Round 2, call J: side streams for the buckets of a bracket (parallel branches of the captured graph).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2j.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2j_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
b() { local name=$1; shift; run "bench_$name" 120 env "$@" python bench.py --no-cpu-baseline --no-roofline; grep -h '^{' "gpurun_out/r2j_bench_$name.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['config']['library_launches_per_step'], d['config']['loss'])" >> "$log" 2>&1; }
b s1 SSA_GROUP_STREAMS=1
b s2 SSA_GROUP_STREAMS=2
b s4 SSA_GROUP_STREAMS=4
b s1b SSA_GROUP_STREAMS=1
run group_tests 300 env SSA_GROUP_STREAMS=4 python -m pytest tests/test_group_gpu.py -q -m gpu
cat "$log"
