#!/bin/bash
# Round-6 evidence on the final build: smoke, kernel tests, PMC passes (-> profiles/r06_pmc.*), the bench line with
# roofline + CPU baseline, rocprofv3 stats + one-step trace, micro-benchmarks, eval bench (bf16 + fp16 rows, the
# reference's full Mapillary recipe in fp16), the fp16-training bench line, secondary rows.   bash tools/calls/final6.sh [tag]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6fin}
mkdir -p gpurun_out profiles
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1
echo "smoke rc=$?: $(grep '^smoke' gpurun_out/${T}_smoke.log | tail -1)"
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu > gpurun_out/${T}_tests.log 2>&1
echo "kernel tests: $(tail -1 gpurun_out/${T}_tests.log)"
bash tools/calls/pmc.sh ${T} > gpurun_out/${T}_pmc_call.log 2>&1
cp gpurun_out/${T}_pmc.json profiles/r06_pmc.json; cp gpurun_out/${T}_pmc.txt profiles/r06_pmc.txt
timeout 600 python bench.py > gpurun_out/${T}_bench.log 2>&1
grep -h '^{' gpurun_out/${T}_bench.log > gpurun_out/${T}_bench_line.json
mkdir -p gpurun_out/${T}_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -o $T -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0 > gpurun_out/${T}_rocprof.log 2>&1
f=$(ls gpurun_out/${T}_prof/*/*kernel_trace.csv gpurun_out/${T}_prof/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_step.py "$f" 110 gpurun_out/${T}_launches.txt > gpurun_out/${T}_trace_step.txt 2>&1
cp $(ls gpurun_out/${T}_prof/*/*kernel_stats.csv gpurun_out/${T}_prof/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/${T}_prof
python -c 'import sys,json; d=json.load(open(sys.argv[1])); print(round(d["ms_per_step"],2), "ms", round(d["value"],2), "img/s; roofline frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "mfma_busy", d["roofline"].get("mfma_busy_frac"), "cpu", d["cpu_baseline"])' gpurun_out/${T}_bench_line.json
head -14 gpurun_out/${T}_trace_step.txt
timeout 200 python tools/tilebench.py 20 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_tilebench.txt
timeout 100 python tools/bnbench.py 30 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_bnbench.txt
timeout 300 python tools/eval_bench.py 10 2>&1 | grep '^{' > gpurun_out/${T}_eval_bench.json
timeout 300 env SSA_ACT_DTYPE=fp16 python tools/eval_bench.py 10 2>&1 | grep '^{' >> gpurun_out/${T}_eval_bench.json
timeout 300 env SSA_ACT_DTYPE=fp16 python tools/eval_bench.py 3 mapillary-ref 2>&1 | grep '^{' >> gpurun_out/${T}_eval_bench.json
cat gpurun_out/${T}_eval_bench.json | python -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["config"], d["storage"], round(d["ms_per_image"],2), "ms", round(d["peak_mem_GB"],1), "GB")'
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --eager-steps 0"
timeout 200 $B --crop 1024 --crop-w 2048 > gpurun_out/${T}_bench_1024x2048.log 2>&1; grep -h '^{' gpurun_out/${T}_bench_1024x2048.log > gpurun_out/${T}_bench_1024x2048.json
timeout 200 $B --batch 2 > gpurun_out/${T}_bench_batch2.log 2>&1; grep -h '^{' gpurun_out/${T}_bench_batch2.log > gpurun_out/${T}_bench_batch2.json
timeout 200 env SSA_FORCE_DIST=1 $B > gpurun_out/${T}_bench_dist1.log 2>&1; grep -h '^{' gpurun_out/${T}_bench_dist1.log > gpurun_out/${T}_bench_dist1.json
timeout 200 env SSA_ACT_DTYPE=fp16 $B > gpurun_out/${T}_bench_fp16.log 2>&1; grep -h '^{' gpurun_out/${T}_bench_fp16.log > gpurun_out/${T}_bench_fp16.json
python -c 'import sys,json; d=json.load(open(sys.argv[1])); c=d["config"]; print("dist1: collectives/step", c["collectives_per_step"], "one exchange", c.get("syncbn_collective_us"), "us ->", c.get("collective_ms_per_step"), "ms/step")' gpurun_out/${T}_bench_dist1.json
for f in 1024x2048 batch2 dist1 fp16; do python -c 'import sys,json; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["ms_per_step"],2), "ms", round(d["value"],2), "img/s")' gpurun_out/${T}_bench_$f.json $f; done
