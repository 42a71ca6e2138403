#!/bin/bash
# Round 4: quick A/B call -- kernel tests, tilebench, bench.py for the default and every named environment variant.
# bash tools/calls/r4b.sh <tag> [name=ENV1=v,ENV2=v ...]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
export TMPDIR=/tmp
T=$1; shift
mkdir -p gpurun_out
log=gpurun_out/$T.log
: > "$log"
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_group_gpu.py tests/test_fuse_bwd_gpu.py -q -m gpu -x > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?: $(tail -1 gpurun_out/${T}_tests.log)" >> "$log"
if [ -z "$NO_TILEBENCH" ]; then
  timeout 200 python tools/tilebench.py 20 > gpurun_out/${T}_tilebench.txt 2>&1
  echo "tilebench rc=$?" >> "$log"; grep -v amdgpu.ids gpurun_out/${T}_tilebench.txt >> "$log"
fi
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline"
line() { grep -h '^{' "$1" | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],2), "ms", d["config"]["library_launches_per_step"], "launches, loss", round(d["config"]["loss"],4))' 2>&1 | tail -1; }
timeout 200 $B > gpurun_out/${T}_bench_default.log 2>&1; echo "default: $(line gpurun_out/${T}_bench_default.log)" >> "$log"
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  timeout 150 env ${envs//,/ } $B > gpurun_out/${T}_bench_$name.log 2>&1; echo "$name [$envs]: $(line gpurun_out/${T}_bench_$name.log)" >> "$log"
done
timeout 200 $B > gpurun_out/${T}_bench_default2.log 2>&1; echo "default again: $(line gpurun_out/${T}_bench_default2.log)" >> "$log"
cat "$log"
