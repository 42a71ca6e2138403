#!/bin/bash
# Round 6 call N: the register-fed 3x3 head kernel with the filter ring left in flight across the chunk barrier
# (default) against the draining form (experiment build drain0); PMC counters of the kernel alone (tools/headbench.py).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6n}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "halo_reg or conv_fwd_bwd" > gpurun_out/${T}_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -2 gpurun_out/${T}_kernel_tests.log
echo "== headbench 3x3, ring in flight across the barrier"; timeout 300 python tools/headbench.py 20 --only-3x3 --fwd-only 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_headbench_reg.txt
echo "== headbench 3x3, drained at the barrier"; timeout 300 python tools/headbench.py 20 --only-3x3 --fwd-only --lib drain0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${T}_headbench_drain0.txt
B="python tools/headbench.py 3 --only-3x3 --fwd-only"
pass() { local n=$1; shift; mkdir -p gpurun_out/${T}_$n; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/${T}_$n -o p -- $B > gpurun_out/${T}_$n.log 2>&1; echo "pass $n ($*) rc=$?"; }
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
files=$(ls gpurun_out/${T}_*/*/*counter_collection.csv gpurun_out/${T}_*/*counter_collection.csv 2>/dev/null)
python tools/pmc_counters.py gpurun_out/${T}_pmc.json $files > gpurun_out/${T}_pmc.txt 2>&1
for d in fetch write sq lds; do rm -rf gpurun_out/${T}_$d; done
head -30 gpurun_out/${T}_pmc.txt
