#!/bin/bash
# Last call of the round: PMC passes on the final sources (-> profiles/r04_pmc.*), then the first device run of the
# three-workgroups-per-CU form of conv_tile_q.hip: its op-level tests, the micro-benchmark, the step.   bash tools/calls/r4v.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r4v}
mkdir -p gpurun_out
bash tools/calls/pmc.sh ${T} > gpurun_out/${T}_pmc_call.log 2>&1
grep -c . gpurun_out/${T}_pmc.txt
timeout 60 env SSA_TILE_Q_THREE=1 python -m pytest tests/test_conv_tile_q_gpu.py -q -m gpu > gpurun_out/${T}_tests_three.log 2>&1
echo "three-per-CU tests: $(tail -1 gpurun_out/${T}_tests_three.log)"
grep -E "^FAILED|^ERROR" gpurun_out/${T}_tests_three.log | head -5
timeout 60 python tools/tilebench.py 10 --q --three 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_tilebench_three.txt
cat gpurun_out/${T}_tilebench_three.txt
B="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0"
timeout 60 env SSA_TILE_Q=3 python $B > gpurun_out/${T}_bench_q3.log 2>&1
timeout 60 python $B > gpurun_out/${T}_bench_default.log 2>&1
for f in q3 default; do grep -h '^{' gpurun_out/${T}_bench_$f.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["ms_per_step"],2), "ms", d["config"]["loss"])' $f || tail -3 gpurun_out/${T}_bench_$f.log; done
