#!/bin/bash
# Round 3, call J: the N > 1 program (overlapped gradient exchange on a communication stream, second RCCL communicator)
# over a one-rank communicator: DDP tests, captured step, bench with SSA_FORCE_DIST=1, rocprofv3 trace showing the
# RCCL kernels next to the weight-gradient kernels.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=r3j
mkdir -p gpurun_out/${T}_prof
timeout 400 python -m pytest tests/test_ddp_gpu.py tests/test_ddp_graph_gpu.py tests/test_rccl_direct_gpu.py -q -m gpu -x > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?: $(tail -1 gpurun_out/${T}_tests.log)"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
timeout 200 env SSA_FORCE_DIST=1 $B > gpurun_out/${T}_bench_dist1.log 2>&1
grep -h '^{' gpurun_out/${T}_bench_dist1.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); c=d["config"]; print("dist1:", round(d["ms_per_step"],2), "ms, hipgraph", c["hipgraph"], "collectives", c["collectives_per_step"], "grad exchanges", c["grad_exchanges_per_step"], "exposed est ms", c["exposed_comm_ms_estimate"], c["capture_error"])'
tail -3 gpurun_out/${T}_bench_dist1.log | cut -c1-300
timeout 200 env SSA_FORCE_DIST=1 SSA_DDP_OVERLAP=0 $B > gpurun_out/${T}_bench_dist1_nooverlap.log 2>&1
grep -h '^{' gpurun_out/${T}_bench_dist1_nooverlap.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); c=d["config"]; print("dist1 no overlap:", round(d["ms_per_step"],2), "ms")'
timeout 200 $B > gpurun_out/${T}_bench_plain.log 2>&1
grep -h '^{' gpurun_out/${T}_bench_plain.log | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print("plain:", round(d["ms_per_step"],2), "ms")'
timeout 240 env SSA_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_prof -o $T -- $B > gpurun_out/${T}_rocprof.log 2>&1
f=$(ls gpurun_out/${T}_prof/*/*kernel_trace.csv gpurun_out/${T}_prof/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_step.py "$f" 20 gpurun_out/${T}_launches.txt > gpurun_out/${T}_trace_step.txt 2>&1
rm -rf gpurun_out/${T}_prof
grep -n -i "nccl\|rccl" gpurun_out/${T}_launches.txt | head -20
