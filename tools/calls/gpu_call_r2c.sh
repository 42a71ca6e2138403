#!/bin/bash
# Round 2, call C: the full GPU suite (incl. the 1024^2 parity tests), the bench line, a rocprofv3
# kernel trace of the same command, the secondary rows.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2c.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2c_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
run parity1024 900 python -m pytest tests/test_parity_1024_gpu.py -q -s -m gpu
run gpu_suite 1100 python -m pytest tests -q -m gpu --deselect tests/test_parity_1024_gpu.py
run bench 400 python bench.py
run bench_1024x2048 150 python bench.py --crop-w 2048 --no-cpu-baseline --no-roofline
run bench_batch2 150 python bench.py --batch 2 --no-cpu-baseline --no-roofline
export TMPDIR=/tmp
run rocprof 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2c_prof -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline
cat "$log"
