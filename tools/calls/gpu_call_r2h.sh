#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD/semantic-segmentation_amd:$PWD
timeout 300 python tools/debug_teacher.py attnscale.DeepV3R50 0.5,1.0,2.0 > gpurun_out/r2h_dbg3.log 2>&1; echo "dbg3 rc=$?"
