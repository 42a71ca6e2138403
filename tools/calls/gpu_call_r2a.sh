#!/bin/bash
# Round 2, call A: what round 1 left unverified and is kept (fused SGD, conv_tile_aux epilogues,
# sibling archs, input-pipeline tail, RCCL inside capture over a one-rank communicator).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
mkdir -p gpurun_out
log=gpurun_out/r2a.log
: > "$log"
run() { local name=$1 t=$2; shift 2; echo "== $name" >> "$log"; timeout "$t" "$@" > "gpurun_out/r2a_$name.log" 2>&1; echo "$name rc=$?" >> "$log"; }
SSA_TEST_UNVERIFIED=1 run unverified_tests 400 python -m pytest tests/test_optim_gpu.py tests/test_siblings_gpu.py \
    tests/test_fuse_bwd_gpu.py tests/test_data_gpu.py tests/test_ddp_graph_gpu.py tests/test_rccl_direct_gpu.py -q -s -m gpu
SSA_FUSED_SGD=1 run bench_fused_sgd 100 python bench.py --no-cpu-baseline
SSA_FUSE_BWD=1 SSA_FUSED_SGD=1 run bench_fuse_bwd 100 python bench.py --no-cpu-baseline --no-roofline
SSA_FORCE_DIST=1 run bench_dist1_eager 120 python bench.py --no-cpu-baseline --no-roofline
SSA_FORCE_DIST=1 SSA_DDP_GRAPH=1 SSA_RCCL_DIRECT=1 run bench_dist1_graph_direct 150 python bench.py --no-cpu-baseline --no-roofline
SSA_FORCE_DIST=1 SSA_DDP_GRAPH=1 run bench_dist1_graph 150 python bench.py --no-cpu-baseline --no-roofline
run convbench 200 tools/bin/convbench 20
nproc >> "$log"; free -g >> "$log"
cat "$log"
