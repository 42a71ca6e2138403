#!/bin/bash
# Round 6 call X: the filter re-pack reads every parameter once (operand forms chained per source tile), the BCE gradient scaling
# folded into rmi_bwd_logits, graph_eval capture policy, p2p stream probe: tests, two bench lines,
# one replayed step under rocprofv3 (per-kernel durations).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6x}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py tests/test_p2p_gpu.py tests/test_graphed_step_gpu.py tests/test_ddp_graph_gpu.py tests/test_optim_gpu.py -q -x -m gpu > gpurun_out/${T}_kernel_tests.log 2>&1
echo "kernel tests rc=$?"; tail -3 gpurun_out/${T}_kernel_tests.log
for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --eager-steps 0 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
  python -c 'import sys,json; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%.3f ms  %.2f img/s" % (d["ms_per_step"], d["value"]))' gpurun_out/${T}_bench.json || tail -20 gpurun_out/${T}_bench.err
done
mkdir -p gpurun_out/${T}_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -o $T -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0 > gpurun_out/${T}_rocprof.log 2>&1
f=$(ls gpurun_out/${T}_prof/*/*kernel_trace.csv gpurun_out/${T}_prof/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_step.py "$f" 110 gpurun_out/${T}_launches.txt > gpurun_out/${T}_trace_step.txt 2>&1
rm -rf gpurun_out/${T}_prof
head -64 gpurun_out/${T}_trace_step.txt
timeout 300 python -m pytest tests/test_e2e_gpu.py tests/test_parity_1024_gpu.py tests/test_amp_fp16_gpu.py -q -x -m gpu > gpurun_out/${T}_parity.log 2>&1
echo "parity rc=$?"; tail -3 gpurun_out/${T}_parity.log
