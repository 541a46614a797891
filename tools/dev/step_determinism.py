"""Is a model's forward + loss + backward bitwise repeatable?  Two passes from the same state: outputs and every parameter
gradient compared with torch.equal; prints the tensors that differ.   python3 tools/dev/step_determinism.py [inpainter|segmenter|classifier]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
which = sys.argv[1] if len(sys.argv) > 1 else "inpainter"
torch.manual_seed(0)
if which == "inpainter":
    from cloud_transformers_amd.chamfer import loss_chamfer
    from cloud_transformers_amd.emd import emdModule
    from cloud_transformers_amd.metrics import sphere_noise
    from tests.test_zoo_gpu import Inpainter
    net = Inpainter().cuda().train()
    gen = torch.Generator(device="cuda").manual_seed(1)
    partial = torch.rand(2, 3, 1, 2048, device="cuda", generator=gen) - 0.5
    gt = torch.nn.functional.normalize(torch.randn(2, 16384, 3, device="cuda", generator=gen), dim=2) * 0.4
    gt4 = gt.transpose(1, 2).unsqueeze(2).contiguous()
    noise = torch.cat([sphere_noise(2, 16384, "cuda", gen), torch.zeros(2, 1, 16384, device="cuda")], dim=1)
    emd = emdModule()
    def run():
        rec4 = net(noise, partial)
        rec = rec4.squeeze(2).transpose(1, 2).contiguous()
        dist, _ = emd(rec, gt, 0.005, 50)
        loss = torch.sqrt(dist).mean(1).mean() + loss_chamfer(rec4, gt4)
        return rec4, loss
elif which == "segmenter":
    from tools.segmenter_step_bench import Segmenter
    net = Segmenter().cuda().train()
    cloud = torch.rand(8, 6, 4096, device="cuda") * 2 - 1
    def run():
        y = net(cloud)
        return y, y.square().mean()
else:
    from tests.test_zoo_gpu import Classifier
    net = Classifier().cuda().train()
    cloud = torch.rand(8, 3, 1, 2048, device="cuda") * 2 - 1
    def run():
        logits, mask = net(cloud)
        return logits, logits.square().mean() + mask.square().mean()
res = []
for rep in range(2):
    net.zero_grad(set_to_none=True)
    torch.manual_seed(123)                      # the classifier's heads hold Dropout layers
    out, loss = run()
    loss.backward()
    torch.cuda.synchronize()
    res.append((out.detach().clone(), float(loss), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
print(which, "loss", res[0][1], res[1][1], " output equal:", bool(torch.equal(res[0][0], res[1][0])))
diff = [(n, float((res[0][2][n] - res[1][2][n]).abs().max()), float(res[0][2][n].abs().max())) for n in res[0][2] if not torch.equal(res[0][2][n], res[1][2][n])]
print("parameter gradients that differ: %d of %d" % (len(diff), len(res[0][2])))
for n, d, m in diff[:40]:
    print("   %-60s max diff %.2e (max %.2e)" % (n, d, m))
