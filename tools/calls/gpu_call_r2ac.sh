#!/bin/bash
# Round 2, call AC (evidence refresh after the last kernel-source edits: 32-bit offset guard in conv_tile, eight
# splits in flight in the weight-gradient reduce): tests, PMC passes, bench line, trace.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out/r2ac_prof gpurun_out/r2ac_pmc_fetch gpurun_out/r2ac_pmc_write
export TMPDIR=/tmp
log=gpurun_out/r2ac.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2ac_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
run tests 200 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py -q -m gpu
tail -2 gpurun_out/r2ac_tests.log >> "$log"
run pmc_fetch 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r2ac_pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
run pmc_write 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r2ac_pmc_write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
run pmc_table 60 python tools/pmc_traffic.py gpurun_out/r2ac_pmc_fetch/f_counter_collection.csv gpurun_out/r2ac_pmc_write/w_counter_collection.csv gpurun_out/r2ac_pmc_traffic.json
rm -f gpurun_out/r2ac_pmc_fetch/* gpurun_out/r2ac_pmc_write/*
cp gpurun_out/r2ac_pmc_traffic.json profiles/r02_pmc_traffic.json
run bench 400 python bench.py
run rocprof 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2ac_prof -o r2ac -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
cat "$log"
grep -h '^{' gpurun_out/r2ac_bench.log | cut -c1-200
