#!/bin/bash
# Round-end evidence on the final build: smoke, PMC passes (-> profiles/r03_pmc.*), bench line with roofline + CPU baseline,
# rocprofv3 kernel trace + stats of the same command, trunk-conv micro-benchmark + phase timing, secondary rows, eval bench,
# the full `-m gpu` suite.  bash tools/calls/evidence.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r3ev}
log=gpurun_out/$T.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/${T}_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
run smoke 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
# ---- PMC passes first: bench.py's roofline block reads profiles/r03_pmc.json (refused unless measured on these sources)
bash tools/calls/pmc.sh ${T}_pmc > gpurun_out/${T}_pmc_call.log 2>&1
cp gpurun_out/${T}_pmc_pmc.json profiles/r03_pmc.json
cp gpurun_out/${T}_pmc_pmc.txt profiles/r03_pmc.txt
run bench 500 python bench.py
grep -h '^{' gpurun_out/${T}_bench.log > gpurun_out/${T}_bench_line.json
mkdir -p gpurun_out/${T}_prof
run rocprof 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -o $T -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0
f=$(ls gpurun_out/${T}_prof/*/*kernel_trace.csv gpurun_out/${T}_prof/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_step.py "$f" 110 gpurun_out/${T}_launches.txt > gpurun_out/${T}_trace_step.txt 2>&1
cp $(ls gpurun_out/${T}_prof/*/*kernel_stats.csv gpurun_out/${T}_prof/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/${T}_prof
run tilebench 300 python tools/tilebench.py 20
run tile_timing 300 python tools/tilebench.py --timing
run bench_1024x2048 150 python bench.py --crop-w 2048 --no-cpu-baseline --no-roofline
run bench_batch2 150 python bench.py --batch 2 --no-cpu-baseline --no-roofline
run bench_dist1 200 env SSA_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-roofline
run bench_fold 150 env SSA_BLOCK_FOLD=1 python bench.py --no-cpu-baseline --no-roofline
run bench_old_tile 150 env SSA_TILE_P=0 python bench.py --no-cpu-baseline --no-roofline
run eval_bench 300 python tools/eval_bench.py
run gpu_suite 2400 python -m pytest tests -q -m gpu
tail -15 gpurun_out/${T}_gpu_suite.log >> "$log"
for n in bench bench_1024x2048 bench_batch2 bench_dist1 bench_fold bench_old_tile; do
  echo "$n: $(grep -h '^{' gpurun_out/${T}_$n.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", round(d["value"],2), "img/s; eager", d["config"].get("eager_ms_per_step"))' 2>&1 | tail -1)" >> "$log"
done
cat "$log"
