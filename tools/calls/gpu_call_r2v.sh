#!/bin/bash
# Round 2, call V (evidence): smoke, the full GPU suite, the bench line with roofline + cpu baseline, rocprofv3
# kernel trace + stats of the same command, PMC traffic passes (FETCH_SIZE / WRITE_SIZE separately), secondary rows.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out/r2v_prof gpurun_out/r2v_pmc_fetch gpurun_out/r2v_pmc_write
export TMPDIR=/tmp
log=gpurun_out/r2v.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2v_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
run smoke 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
run bench 500 python bench.py
run rocprof 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2v_prof -o r2v -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run pmc_fetch 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2v_pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
run pmc_write 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2v_pmc_write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
ls gpurun_out/r2v_pmc_fetch gpurun_out/r2v_pmc_write >> "$log" 2>&1
run pmc_table 120 python tools/pmc_traffic.py gpurun_out/r2v_pmc_fetch/f_counter_collection.csv gpurun_out/r2v_pmc_write/w_counter_collection.csv gpurun_out/r2v_pmc_traffic.json
# keep the merged output small: the per-dispatch counter CSVs are large
rm -f gpurun_out/r2v_pmc_fetch/*kernel_trace.csv gpurun_out/r2v_pmc_write/*kernel_trace.csv
gzip -f gpurun_out/r2v_pmc_fetch/f_counter_collection.csv gpurun_out/r2v_pmc_write/w_counter_collection.csv 2>/dev/null
run bench_1024x2048 150 python bench.py --crop-w 2048 --no-cpu-baseline --no-roofline
run bench_batch2 150 python bench.py --batch 2 --no-cpu-baseline --no-roofline
run bench_dist1 150 env SSA_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-roofline
run eval_bench 300 python tools/eval_bench.py
run gpu_suite 1500 python -m pytest tests -q -m gpu
tail -15 gpurun_out/r2v_gpu_suite.log >> "$log"
cat "$log"
