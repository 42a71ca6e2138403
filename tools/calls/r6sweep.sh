#!/bin/bash
# Round 6: environment sweeps on the final library (no code change): weight-gradient flush interval, trunk-conv
# workgroup budget, BatchNorm reduce grid.  One bench line each, same box.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6sweep}
mkdir -p gpurun_out
run() {
  env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --eager-steps 0 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
  python -c 'import sys,json; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%-40s %.3f ms" % (sys.argv[2], d["ms_per_step"]))' gpurun_out/${T}_bench.json "$*" 2>/dev/null || { echo "$* FAILED"; tail -5 gpurun_out/${T}_bench.err; }
}
run SSA_NOP=1
run SSA_WGRAD_FLUSH_AT=128
run SSA_WGRAD_FLUSH_AT=192
run SSA_WGRAD_FLUSH_AT=384
run SSA_TILE_P_WGS=512
run SSA_TILE_P_WGS=640
run SSA_BN_REDUCE_BLOCKS=1024
run SSA_BN_WIDE_CHUNKS=4
run SSA_BN_WIDE_CHUNKS=8
run SSA_NOP=1
