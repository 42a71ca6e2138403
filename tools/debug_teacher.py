"""Diagnostic for the teacher-forced harness on the GPU: for selected layers print the norms of the
reference / HIP parameter gradients and compare both with a direct torch evaluation on the device."""
import copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "semantic-segmentation_amd"), ROOT]
from semseg_amd import ops, hip_backend as hb
from teacher_backend import TeacherBackend
from test_attnscale_cpu import build
from test_e2e_gpu import _synth


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "attnscale.DeepV3R50"
    scales = [float(s) for s in (sys.argv[2] if len(sys.argv) > 2 else "0.5,1.0,2.0").split(",")]
    gold = {"scales": scales, "wt": 0.05 if name == "attnscale.DeepV3R50" else 0, "seed": 3}
    cpu_net = build(name, gold, True).float()
    sd = cpu_net.state_dict()
    for k in sd:
        if k.endswith("aspp.img_conv.1.weight"):
            sd[k].mul_(0.05)
    images, gts = _synth(2, 128, 192, seed=17)
    hip_net = copy.deepcopy(cpu_net).cuda().train()
    tb = TeacherBackend(cpu_net, hip_net)
    seen = {}

    def debug(idx, opname, p, q, r, h, hip_in, hip_out, dys, from_arena):
        shape = tuple(p.shape)
        if opname != "conv2d" or shape not in ((19, 256, 1, 1), (48, 256, 1, 1)):
            if p.dim() == 1 and idx % 7 == 0:
                print("  op %3d %-12s %-18s |r| %.3e |h| %.3e cos %.4f arena=%s" % (
                    idx, opname, shape, float(r.norm()), float(h.norm()), cos(h.cpu(), r), from_arena), flush=True)
            return
        x = hip_in[0].float()
        dy = dys[0].to("cuda").float()
        if dy.shape[1:3] != x.shape[1:3]:
            print("  op %d: dy %s x %s (padded conv)" % (idx, tuple(dy.shape), tuple(x.shape)))
            return
        direct = torch.einsum("bhwo,bhwi->oi", dy, x).reshape(shape)
        direct_bf = torch.einsum("bhwo,bhwi->oi", dy.bfloat16().float(), x).reshape(shape)
        print("op %3d %s %s P=%d |dy| %.3e |x| %.3e |r| %.3e |h| %.3e |direct| %.3e | cos(h,r) %.4f cos(h,direct) %.4f "
              "cos(r,direct) %.4f cos(direct_bf,direct) %.4f arena=%s" % (
                  idx, opname, shape, x.shape[0] * x.shape[1] * x.shape[2], float(dy.norm()), float(x.norm()),
                  float(r.norm()), float(h.norm()), float(direct.norm()), cos(h.cpu(), r), cos(h, direct),
                  cos(r.cuda(), direct), cos(direct_bf, direct), from_arena), flush=True)
        for k0, (i0, d0) in seen.items():
            if k0[0] == shape:
                print("      vs op %d: cos(h, that direct) %.4f" % (i0, cos(h, d0)))
        seen[(shape, idx)] = (idx, direct)

    tb.debug = debug
    import threading
    G = hb._GRADS
    o_slot, o_pub, o_run = G.slot, G.publish, hb._run_wgrad_jobs
    watch = {(19, 256, 1, 1)}

    def slot(p):
        had = id(p) in G.slots
        v = o_slot(p)
        if tuple(p.shape) in watch:
            print("    [slot] thr %d had=%s ptr %x chunks %d armed %s grad_is_none %s" % (
                threading.get_ident() % 10000, had, v.data_ptr(), len(G.chunks), G.armed, p.grad is None), flush=True)
        return v

    def publish():
        ent = [(tuple(p.shape), v.data_ptr(), p.grad is None) for p, v in G.slots.values() if tuple(p.shape) in watch]
        print("    [publish] thr %d slots %d queued %d watch %s" % (threading.get_ident() % 10000, len(G.slots),
                                                                  len(hb._WGRAD_Q), ent), flush=True)
        vs = [v for p, v in G.slots.values() if tuple(p.shape) in watch]
        o_pub()
        torch.cuda.synchronize()
        for v in vs:
            print("    [publish] after: |v| %.4e" % float(v.norm()), flush=True)

    def run(jobs, strip):
        for j in jobs:
            if tuple(j.target.shape) in watch:
                print("    [wgrad job] thr %d target %x P %s" % (threading.get_ident() % 10000, j.target.data_ptr(),
                                                                j.geom_in), flush=True)
        return o_run(jobs, strip)

    G.slot, G.publish, hb._run_wgrad_jobs = slot, publish, run
    print("main thread %d" % (threading.get_ident() % 10000))
    ops._set_backend_for_tests(tb)
    hb.clear_pack_cache()
    out = cpu_net({"images": images, "gts": gts})
    loss = out["pred"] if isinstance(out, dict) else out
    loss.backward()
    torch.cuda.synchronize()
    fails = tb.rec.failures()
    print("%d ops, %d failures; failing ops: %s" % (tb.rec.n_ops, len(fails), sorted({r[0] for r in fails})))


if __name__ == "__main__":
    main()
