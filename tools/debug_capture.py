"""Debug helper: capture one training step (concurrent streams) in a hipGraph at a small crop."""
import faulthandler, os, sys, time
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "semantic-segmentation_amd")]
import torch
import bench
crop = int(sys.argv[1]) if len(sys.argv) > 1 else 256
same_stream = (sys.argv[2] == "same") if len(sys.argv) > 2 else True
net = bench.build_model(1)
optim = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
images, gts = bench.synth_batch(1, crop, crop, 0, "cuda")
inputs = {"images": images, "gts": gts}
static_loss = torch.zeros((), device="cuda")
fwd_only = os.environ.get("FWD_ONLY", "") != ""
def step():
    optim.zero_grad(set_to_none=True)
    if fwd_only:
        with torch.no_grad():
            loss = net(inputs)
    else:
        loss = net(inputs)
        loss.backward()
        optim.step()
    static_loss.copy_(loss.detach())
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("warmup ok, loss", float(static_loss), flush=True)
g = torch.cuda.CUDAGraph()
optim.zero_grad(set_to_none=True)
kw = {"stream": side} if same_stream else {}
with torch.cuda.graph(g, **kw):
    step()
torch.cuda.synchronize()
print("capture ok", flush=True)
for i in range(3):
    g.replay()
torch.cuda.synchronize()
print("replay ok, loss", float(static_loss), flush=True)
t0 = time.perf_counter()
for i in range(5):
    g.replay()
torch.cuda.synchronize()
print("ms/step %.2f" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
