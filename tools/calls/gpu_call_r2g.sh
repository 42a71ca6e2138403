#!/bin/bash
# Round 2, call G: attnscale teacher diagnostics, device BICUBIC + prefetcher tests.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2g.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2g_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
run attnscale 400 python -m pytest tests/test_attnscale_gpu.py -q -s -m gpu
run data 200 python -m pytest tests/test_data_gpu.py -q -m gpu
cat "$log"
