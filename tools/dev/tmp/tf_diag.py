import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_zoo_gpu as T
gold = np.load("/root/repo/tests/golden/zoo_segmenter_blocks.npz")
net = T._model(int(gold["seed"])).train()
xyz = torch.from_numpy(gold["cloud"]).cuda().squeeze(2)[:, :3].contiguous()
def stats(a, b):
    a = a.detach().cpu().double().numpy(); b = np.asarray(b, dtype=np.float64)
    e = np.abs(a - b) / np.abs(b).max()
    return "max %.2e  frac>1e-4 %.5f  n>1e-4 %d  median %.1e" % (e.max(), (e > 1e-4).mean(), (e > 1e-4).sum(), np.median(e))
for i in range(12):
    x = T._from_bf16_bits(gold["x_in_%d" % i]).cuda().requires_grad_(True)
    out, _ = net.attentions_encoder[i](x, xyz)
    (out * T._cot_for(i, out.shape).cuda()).sum().backward()
    print("seg", i, "out:", stats(out[:, :64], gold["out_%d" % i]), "| g_in:", stats(x.grad[:, :64], gold["g_in_%d" % i]))
gold = np.load("/root/repo/tests/golden/zoo_inpainter_decoder_blocks.npz")
torch.manual_seed(int(gold["seed"])); net = T.Inpainter(); T._perturb(net, int(gold["seed"]) + 2); net = net.cuda().eval()
noise = torch.from_numpy(gold["noise"]).cuda()
for i in range(12):
    z = torch.from_numpy(gold["z"]).cuda().requires_grad_(True)
    x = T._from_bf16_bits(gold["x_in_%d" % i]).cuda().requires_grad_(True)
    out, _ = net.attentions_decoder[i](x, z, noise[:, :3].contiguous())
    (out * T._cot_for(100 + i, out.shape).cuda()).sum().backward()
    print("dec", i, "out:", stats(out[:, :64], gold["out_%d" % i]), "| g_in:", stats(x.grad[:, :64], gold["g_in_%d" % i]), "| g_z:", stats(z.grad, gold["g_z_%d" % i]))
