import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _limit_cpu_threads():
    """The CPU sides of the parity tests (the oracle and the storage emulation, fp32 torch on the host) run on a bounded
    number of intra-op threads: torch's default is one per visible CPU, and on the GPU boxes (256 visible, cgroup-limited)
    that is an order of magnitude SLOWER than 8-16 threads for this graph of small convs (bench.py's CPU baseline probes
    it: 128 threads 19.5 s per 256 x 256 iteration, 14x slower than 8).  SSA_TEST_THREADS overrides."""
    try:
        import torch
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        torch.set_num_threads(max(1, min(int(os.environ.get("SSA_TEST_THREADS", "16")), avail)))
    except Exception:       # noqa: BLE001
        pass


def pytest_configure(config):
    _limit_cpu_threads()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "multigpu: needs >= 2 MI355X in one node; DESELECTED (not skipped) elsewhere")


# Order of the `-m gpu` suite (the driver runs it with -x): op-level kernels, grouped launches, data, optimizer, the
# data-parallel path and the headline-configuration parity first; the long full-size evaluation tests last, so that a
# late failure hides nothing that is cheap to run (round-4 review: a failure at test 172 of 184 hid 12 tests).
_ORDER = ["test_kernels_gpu", "test_group_gpu", "test_fuse_bwd_gpu", "test_data_gpu", "test_optim_gpu",
          "test_rccl_direct_gpu", "test_ddp_gpu", "test_ddp_graph_gpu", "test_multi_gpu_rccl", "test_graphed_step_gpu",
          "test_amp_fp16_gpu", "test_parity_1024_gpu", "test_e2e_gpu", "test_deepv3_gpu", "test_attnscale_gpu",
          "test_siblings_gpu", "test_fp16_storage_gpu", "test_parity_eval_gpu"]


def _file_rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    return _ORDER.index(name) if name in _ORDER else -1        # (CPU test files keep their place in front)


def pytest_collection_modifyitems(config, items):
    """Tests that need two GPUs of one node are deselected on smaller boxes: a one-GPU run reports no skips."""
    items.sort(key=_file_rank)                                  # stable: the order inside a file is the file's
    multi = [it for it in items if it.get_closest_marker("multigpu")]
    if not multi:
        return
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:       # noqa: BLE001
        n = 0
    if n < 2:
        config.hook.pytest_deselected(items=multi)
        items[:] = [it for it in items if not it.get_closest_marker("multigpu")]


@pytest.fixture(scope="session")
def out_dir():
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    return d


# ---- SSA_EMU=1: run the `-m gpu` kernel tests against the CPU emulation build of the kernel sources
# (tests/emu_util.py, tools/emu) with CPU tensors as device memory -- a development aid for machines without
# a GPU (small shapes only: a workgroup is 256 fibers).  Never set by the driver; the product is untouched.
_EMU = bool(os.environ.get("SSA_EMU"))


@pytest.fixture(autouse=True)
def _emu_mode(request):
    if not _EMU:
        yield
        return
    import torch
    from emu_util import emu_backend
    mod = request.module
    saved_dev = getattr(mod, "DEV", None)
    if saved_dev is not None:
        mod.DEV = "cpu"
    saved_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        with emu_backend():
            yield
    finally:
        torch.cuda.synchronize = saved_sync
        if saved_dev is not None:
            mod.DEV = saved_dev
