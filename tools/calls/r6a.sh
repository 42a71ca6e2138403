#!/bin/bash
# Round 6 call A: the full `-m gpu` suite with the new parity tests (four-scale 65-class chain on both builds, per-output
# fp16 e2e bound, OCRNet training sibling, smoke entry, scaler counters), then the default bench line (fp16 headline + the
# bf16 child) and a bare `bench.py --gpus 1` sanity of the self-launch path.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=${1:-r6a}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/${T}_gpu_suite.log 2>&1
echo "suite rc=$?"
tail -40 gpurun_out/${T}_gpu_suite.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r6a_bench_line.json")) if True else None
r = j.get("roofline") or {}
print("value", j["value"], "ms", j["ms_per_step"], "dtype", j["dtype"], "bf16", j.get("value_bf16"), j.get("ms_per_step_bf16"), j.get("secondary_error"))
print("frac", r.get("frac"), "mfma_3x3", json.dumps(r.get("mfma_3x3"))[:900])
print("cpu", j.get("cpu_baseline"))
fam = r.get("families", {})
for k, v in list(fam.items())[:24]:
    print("  %-22s %.3f ms  n=%d  %.0f TF/s %.0f GB/s" % (k, v["ms_per_step"], v["launches_per_step"], v["tflops"], v["hbm_GBps"]))
PY
tail -5 gpurun_out/${T}_bench.err
