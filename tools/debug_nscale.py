#!/usr/bin/env python
"""Debug aid: the hierarchical {0.5, 1.0, 2.0} evaluation of tests/test_e2e_gpu.py::test_eval_nscale traced op by op --
fp32 oracle, storage-emulation backend and the HIP path -- printing every op whose HIP error exceeds 2x the
emulation's (which op a discrepancy of the outputs comes from).  SSA_ACT_DTYPE selects the storage format."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from semseg_amd import ops
    from semseg_amd.config import cfg
    from semseg_amd.loss import RMILoss
    from semseg_amd.network import ocrnet
    from oracle_backend import OracleBackend
    from bf16_emu_backend import Bf16EmuBackend, traced
    from test_e2e_gpu import parity_state_dict, _synth, _rel
    from oracle.model import Net
    cfg.LOSS.SUPERVISED_MSCALE_WT = 0.05
    cfg.MODEL.BNFUNC = None
    net0 = ocrnet.HRNet_Mscale(19, RMILoss(num_classes=19, ignore_index=255))
    sd = parity_state_dict([(k, tuple(v.shape)) for k, v in net0.state_dict().items()], seed=0)
    images, gts = _synth(2, 256, 256, seed=77)
    with torch.no_grad():
        Net(sd, 19, training=True, bn_momentum=1.0, criterion="ce").two_scale_forward(images, gts)
    images = images[:1, :, :128, :192].contiguous()
    gts = gts[:1, :128, :192].contiguous()

    def run(backend, device):
        prev = ops._BACKEND
        ops._set_backend_for_tests(backend)
        try:
            cfg.MODEL.N_SCALES = [0.5, 1.0, 2.0]
            net = ocrnet.HRNet_Mscale(19, RMILoss(num_classes=19, ignore_index=255))
            net.load_state_dict(sd)
            net = net.to(device).eval()
            with torch.no_grad():
                o = net({"images": images.to(device), "gts": gts.to(device)})
            return {k: v.float().cpu() for k, v in o.items()}
        finally:
            cfg.MODEL.N_SCALES = None
            ops._set_backend_for_tests(prev)
    ref_log, names, emu_err, hip_err = [], [], [], []
    ref = run(traced(OracleBackend(), lambda i, n, y: (ref_log.append(y.detach()), names.append(n))), "cpu")
    emu = run(traced(Bf16EmuBackend(), lambda i, n, y: emu_err.append(_rel(y.detach(), ref_log[i]))), "cpu")
    hip = run(traced(ops.HipBackend(), lambda i, n, y: hip_err.append(_rel(y.detach().float().cpu(), ref_log[i]))), "cuda")
    print("ops", len(ref_log), len(emu_err), len(hip_err))
    shown = 0
    for i in range(len(hip_err)):
        if hip_err[i] > 2.0 * emu_err[i] + 2e-3 and shown < 40:
            print("op %4d %-16s %-22s hip %.4f emu %.4f" % (i, names[i], tuple(ref_log[i].shape), hip_err[i], emu_err[i]))
            shown += 1
    for i in range(max(0, len(hip_err) - 70), len(hip_err)):
        print("   %4d %-16s %-22s hip %.4f emu %.4f" % (i, names[i], tuple(ref_log[i].shape), hip_err[i], emu_err[i]))
    for k in sorted(ref):
        print("out %-10s hip %.4f emu %.4f" % (k, _rel(hip[k], ref[k]), _rel(emu[k], ref[k])))


if __name__ == "__main__":
    main()
