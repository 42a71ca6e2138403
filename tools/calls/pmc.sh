#!/bin/bash
# PMC passes over bench.py (2 timed steps each): HBM traffic, MFMA busy, LDS conflicts, occupancy -> gpurun_out/<tag>_pmc.json
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-pmc}
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --eager-steps 0"
pass() { local n=$1; shift; mkdir -p gpurun_out/${T}_$n; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/${T}_$n -o p -- $B > gpurun_out/${T}_$n.log 2>&1; echo "pass $n ($*) rc=$?"; }
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
files=$(ls gpurun_out/${T}_*/*/*counter_collection.csv gpurun_out/${T}_*/*counter_collection.csv 2>/dev/null)
python tools/pmc_counters.py gpurun_out/${T}_pmc.json $files > gpurun_out/${T}_pmc.txt 2>&1
for d in fetch write sq lds; do rm -rf gpurun_out/${T}_$d; done
head -45 gpurun_out/${T}_pmc.txt
