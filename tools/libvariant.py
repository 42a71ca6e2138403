"""Run a tool / bench.py against an experiment build of the library (tools/expbuild.sh):
python tools/libvariant.py <name> <script.py> [args...]   -> lib/libsemseg_hip_<name>.so"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
    sys.path.insert(0, p)
from semseg_amd import _lib  # noqa: E402

name, script = sys.argv[1], sys.argv[2]
_lib.LIB_PATH = _lib.LIB_PATH.replace("libsemseg_hip.so", "libsemseg_hip_%s.so" % name)
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
