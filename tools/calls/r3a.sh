#!/bin/bash
# Round 3, call A: first contact of the persistent trunk conv (conv_tile_p.hip) and the folded bn1 with hardware:
# op-level + grouped + fused-backward tests, A/B of old / persistent / persistent+fold, strip-length sweep, trace.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=r3a
mkdir -p gpurun_out/${T}_prof
log=gpurun_out/$T.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/${T}_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
run tests 420 python -m pytest tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py tests/test_kernels_gpu.py -q -m gpu -x
tail -3 gpurun_out/${T}_tests.log >> "$log"
run bench_old 150 env SSA_TILE_P=0 SSA_BLOCK_FOLD=0 $B
run bench_p 150 env SSA_BLOCK_FOLD=0 $B
run bench_fold 150 $B
run bench_fold_w384 150 env SSA_TILE_P_WGS=384 $B
run bench_fold_w768 150 env SSA_TILE_P_WGS=768 $B
run bench_fold_w1024 150 env SSA_TILE_P_WGS=1024 $B
run rocprof 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_prof -o $T -- $B
f=$(ls gpurun_out/${T}_prof/*/*kernel_trace.csv gpurun_out/${T}_prof/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_step.py "$f" 60 > gpurun_out/${T}_trace_step.txt 2>&1
rm -rf gpurun_out/${T}_prof
for n in old p fold fold_w384 fold_w768 fold_w1024; do
  echo "$n: $(grep -h '^{' gpurun_out/${T}_bench_$n.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", d["config"]["loss"])' 2>&1 | tail -1)" >> "$log"
done
head -40 gpurun_out/${T}_trace_step.txt >> "$log"
cat "$log"
