#!/bin/bash
# Round 2, call R: BatchNorm prologues load the statistics replicas together; 16-byte loads in the tiled repack; trace.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out/r2r_prof
export TMPDIR=/tmp
log=gpurun_out/r2r.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2r_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
b() { local name=$1; shift; run "bench_$name" 120 env "$@" python bench.py --no-cpu-baseline --no-roofline; grep -h '^{' "gpurun_out/r2r_bench_$name.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['config']['library_launches_per_step'], d['config']['loss'])" >> "$log" 2>&1; }
run tests 600 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu
tail -4 gpurun_out/r2r_tests.log >> "$log"
b default SSA_X=0
b default2 SSA_X=0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2r_prof -o r2r -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r2r_prof_bench.log 2>&1
echo "prof rc=$?" >> "$log"
cat "$log"
