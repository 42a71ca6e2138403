#!/bin/bash
# First GPU call of the next round: everything that was written after round 1's GPU budget ran
# out, in order of value, each leg under its own timeout and with its own log under gpurun_out/.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/next_gpu_call.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/next_gpu_call.log
: > "$log"
run() {  # name, timeout, command...
  local name=$1 t=$2; shift 2
  echo "== $name" >> "$log"
  timeout "$t" "$@" > "gpurun_out/$name.log" 2>&1
  echo "$name rc=$?" >> "$log"
}
# 1. tests that have not run on hardware yet (fused SGD variants, filter-cache refresh, OCRNetASPP,
#    sibling training steps, RCCL inside a captured graph with a one-rank communicator)
SSA_TEST_UNVERIFIED=1 run unverified_tests 420 python -m pytest tests/test_optim_gpu.py tests/test_siblings_gpu.py \
    tests/test_ddp_graph_gpu.py tests/test_deferred_reduce_gpu.py tests/test_fuse_bwd_gpu.py tests/test_rccl_direct_gpu.py tests/test_data_gpu.py -q -s -m gpu
# 2. fused SGD in the bench (valid now that the version counters are bumped): compare with the default
SSA_FUSED_SGD=1 run bench_fused_sgd 120 python bench.py --no-cpu-baseline
run bench_torch_sgd 120 python bench.py --no-cpu-baseline
# 2a. the reference's actual sota crop, 1024x2048 (secondary row, exactly 2x the work)
run bench_1024x2048 120 python bench.py --crop 1024 --crop-w 2048 --no-cpu-baseline
# 2a'. two images per GPU (scripts/train_mapillary.yml trains 1024x1024 crops with bs_trn 2): same launch count, twice the work
run bench_batch2 120 python bench.py --batch 2 --no-cpu-baseline
# 2b. deferred + batched weight-gradient reduces (641 -> 9 launches per step)
SSA_DEFER_WGRAD_REDUCE=1 run bench_defer_reduce 120 python bench.py --no-cpu-baseline
SSA_DEFER_WGRAD_REDUCE=1 SSA_FUSED_SGD=1 run bench_defer_reduce_fused_sgd 120 python bench.py --no-cpu-baseline
# 2c. backward fusions (BN backward sums / residual add in the data-gradient epilogue)
SSA_FUSE_BWD=1 run bench_fuse_bwd 120 python bench.py --no-cpu-baseline
SSA_FUSE_BWD=1 SSA_DEFER_WGRAD_REDUCE=1 SSA_FUSED_SGD=1 run bench_all_three 120 python bench.py --no-cpu-baseline
# 2d. the N > 1 code path over a one-rank communicator: eager, captured, captured with direct RCCL calls
SSA_FORCE_DIST=1 run bench_dist1_eager 180 python bench.py --no-cpu-baseline --no-roofline
SSA_FORCE_DIST=1 SSA_DDP_GRAPH=1 run bench_dist1_graph 180 python bench.py --no-cpu-baseline --no-roofline
SSA_FORCE_DIST=1 SSA_DDP_GRAPH=1 SSA_RCCL_DIRECT=1 run bench_dist1_graph_direct 180 python bench.py --no-cpu-baseline --no-roofline
# 2e. per-launch cost of the fused data-gradient epilogues next to the plain tile kernel
run convbench 240 tools/bin/convbench 20
# 3. who launches the ~350 aten adds and ~300 D2D copies per step
run attribute_launches 180 python tools/attribute_launches.py 512
cat "$log"
