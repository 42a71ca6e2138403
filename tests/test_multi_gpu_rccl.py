"""Two ranks on two GPUs of one node over RCCL/xGMI -- the launch line the driver uses for its scaling runs
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N`):
the direct RCCL communicator across real devices, and the captured N > 1 training step (SyncBN exchanges and the
in-place gradient all-reduce as nodes of the hipGraph).  Needs two GPUs: deselected on one-GPU boxes
(tests/conftest.py), where tests/test_ddp_graph_gpu.py and tests/test_rccl_direct_gpu.py cover the same program over
a one-rank communicator.  Has NOT run on hardware yet (the development boxes have one GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.multigpu]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _torchrun(script_args, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


COMM = r"""
import os, sys, torch, torch.distributed as dist
sys.path[:0] = [%r, %r]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dist.init_process_group("nccl")
from semseg_amd import rccl
c = rccl.comm()
assert (c.rank, c.world) == (rank, world)
x = torch.full((1441,), rank + 1.0, dtype=torch.float64, device="cuda")
c.all_reduce_sum_(x); torch.cuda.synchronize()
assert torch.equal(x, torch.full_like(x, world * (world + 1) / 2.0)), x[:3]
y = torch.full((1 << 20,), float(rank), device="cuda")
c.all_reduce_(y, average=True); torch.cuda.synchronize()
assert torch.allclose(y, torch.full_like(y, (world - 1) / 2.0)), y[:3]
# the same collective as a node of a captured graph, replayed twice: z -> world * 2z
z = torch.ones(512, dtype=torch.float64, device="cuda")
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    c.all_reduce_sum_(z)                       # connections are set up outside the capture
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    z.mul_(2.0)
    c.all_reduce_sum_(z)
torch.cuda.synchronize()
z.fill_(1.0)
g.replay(); g.replay(); torch.cuda.synchronize()
assert torch.equal(z, torch.full_like(z, (2.0 * world) ** 2)), z[:3]
dist.barrier()
rccl.shutdown(); dist.destroy_process_group()
if rank == 0:
    print("comm ok")
"""


def test_direct_rccl_across_two_gpus(tmp_path):
    script = tmp_path / "comm.py"
    script.write_text(COMM % (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")))
    r = _torchrun([str(script)], timeout=300)
    assert r.returncode == 0 and "comm ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_bench_two_gpus_is_the_captured_program():
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--crop", "256", "--steps", "4", "--warmup", "1",
                   "--no-cpu-baseline", "--no-roofline"])
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    print("2 GPUs: %.2f ms/step, %.1f img/s, %s collectives/step, loss %.4f" % (
        line["ms_per_step"], line["value"], cfg["collectives_per_step"], cfg["loss"]))
    assert line["n_gpus"] == 2 and cfg["global_batch"] == 2 and cfg["parallelism"] == "dp2"
    assert cfg["hipgraph"] is True and cfg["capture_error"] is None       # the N > 1 step IS a replayed graph
    assert 0 < cfg["collectives_per_step"] < 700
    assert cfg["loss"] == cfg["loss"] and line["value"] > 0
