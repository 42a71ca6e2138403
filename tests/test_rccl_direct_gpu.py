"""semseg_amd.rccl.DirectComm: ncclAllReduce enqueued directly on the compute stream, over a
ONE-rank communicator (all a one-GPU box can host): bootstrap through torch.distributed, the call
itself, and its capture in a hipGraph.  The multi-GPU behaviour is covered only by the driver's
multi-GPU bench runs.  Ran on an MI355X in round 2 (profiles/r02_call_a.log): RCCL prints its
version banner to stdout after the script's last line, hence the line-wise check."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


CODE = r"""
import os, sys, socket, torch, torch.distributed as dist
sys.path[:0] = [%r, %r]
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from semseg_amd import rccl
c = rccl.comm()
assert (c.rank, c.world) == (0, 1)
x = torch.arange(1441, dtype=torch.float64, device="cuda") * 0.5
want = x.clone()
c.all_reduce_sum_(x)                       # one rank: the sum is the input
c.all_reduce_(x, average=True)             # ... and so is the mean (ncclAvg)
torch.cuda.synchronize()
assert torch.equal(x, want)
y = torch.randn(1 << 20, device="cuda"); wy = y.clone()
c.all_reduce_sum_(y); torch.cuda.synchronize()
assert torch.equal(y, wy)
# captured in a graph, on a side stream, replayed
g = torch.cuda.CUDAGraph()
z = torch.ones(512, dtype=torch.float64, device="cuda")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    c.all_reduce_sum_(z)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    z.mul_(2.0)
    c.all_reduce_sum_(z)
    z.add_(1.0)
torch.cuda.synchronize()
z.fill_(1.0)
g.replay(); g.replay()
torch.cuda.synchronize()
assert torch.equal(z, torch.full_like(z, 7.0)), z[:4]      # ((1*2+1)*2+1)
rccl.shutdown()
dist.destroy_process_group()
print("ok")
"""


def test_direct_comm_one_rank():
    code = CODE % (ROOT, os.path.join(ROOT, "semantic-segmentation_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout.split(), (r.stdout + r.stderr)[-2000:]
