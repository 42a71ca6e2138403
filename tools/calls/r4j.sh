#!/bin/bash
# Round 4: step-level A/B of the host-side changes (fan-out handles in the head, fitted weight-gradient strips) and a
# sweep of the side-stream knobs on the new kernels.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r4j}
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu -x > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?: $(tail -1 gpurun_out/${T}_tests.log)"
timeout 300 python -m pytest tests/test_e2e_gpu.py -q -m gpu -x -k "train_step or smoke" > gpurun_out/${T}_e2e.log 2>&1
echo "e2e train rc=$?: $(tail -1 gpurun_out/${T}_e2e.log)"
timeout 200 env SSA_ACT_DTYPE=fp16 python tools/debug_ocr_attn.py 2>&1 | grep -v amdgpu.ids
timeout 200 python tools/debug_ocr_attn.py 2>&1 | grep -v amdgpu.ids
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
line() { grep -h '^{' "$1" | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", round(d["config"]["loss"],4))' 2>&1 | tail -1; }
run() { name=$1; shift; timeout 200 env "$@" $B > gpurun_out/${T}_bench_$name.log 2>&1; echo "$name [$*]: $(line gpurun_out/${T}_bench_$name.log)"; }
run default A=1
run nofit SSA_WGRAD_FIT=0
run flush128 SSA_WGRAD_FLUSH_AT=128
run flush192 SSA_WGRAD_FLUSH_AT=192
run flush384 SSA_WGRAD_FLUSH_AT=384
run strip6 SSA_WGRAD_STRIP=6
run strip10 SSA_WGRAD_STRIP=10
run nostream SSA_WGRAD_STREAM=0
run default2 A=1
