"""tools/dev/host_profile.py [classifier|segmenter] [steps]: where the HOST time of an EAGER training step goes — cProfile over a few
eager steps (no synchronisation inside the steps), the top functions by own time and by cumulative time; the wall per step beside
the device time of the same step as one HIP graph.  Eager steps of the zoo models are host-bound (classifier 31.5 ms eager vs
14.9 ms graphed): this is the list to shorten."""
import cProfile, io, os, pstats, sys, time
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloud_transformers_amd.layers.pointwise import convert_pointwise          # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "classifier"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
if which == "classifier":
    from tests.test_zoo_gpu import Classifier
    B, N = 8, 2048
    net = convert_pointwise(Classifier().cuda()).train()
    cloud = torch.rand(B, 3, 1, N, device="cuda") * 2 - 1
    labels = torch.randint(15, (B,), device="cuda")
    fg = (torch.rand(B, 1, 1, N, device="cuda") > 0.4).float()
    ce, bce = nn.CrossEntropyLoss(), nn.BCEWithLogitsLoss()

    def loss_fn():
        logits, mask = net(cloud)
        return ce(logits, labels) + bce(mask, fg)
else:
    from tests.test_zoo_gpu import Segmenter
    B, N = 8, 4096
    net = convert_pointwise(Segmenter().cuda()).train()
    cloud = torch.rand(B, 6, 1, N, device="cuda") * 2 - 1
    labels = torch.randint(13, (B, N), device="cuda")
    ce = nn.CrossEntropyLoss()

    def loss_fn():
        out = net(cloud)
        out = out[0] if isinstance(out, (tuple, list)) else out
        return ce(out.reshape(B, 13, N), labels)
opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9)


def step():
    opt.zero_grad(set_to_none=True)
    loss_fn().backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
t_wall = (time.perf_counter() - t0) / steps
print("%s: host time per eager step %.2f ms (launches returned), wall %.2f ms" % (which, t_host * 1e3, t_wall * 1e3))
torch.autograd.set_multithreading_enabled(False)      # backward on this thread: its Python functions show in the profile
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(60)
    print("==== by %s (all %d steps)" % (key, steps))
    print("\n".join(l[:200] for l in s.getvalue().splitlines()[4:75]))
