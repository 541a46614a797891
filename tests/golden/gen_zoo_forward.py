#!/usr/bin/env python3
"""Whole-model golden: the reference's S3DIS segmenter (model_zoo/s3dis/segmenter.py on the reference's own layers, CPU,
the two third-party stand-ins of gen_golden.py) evaluated on a small seeded cloud.  Weights are NOT stored: they are
the ones `torch.manual_seed(SEED); Model()` produces, and this package's modules draw their initial parameters in the
same order (checked by tests/test_zoo_cpu.py::test_same_seed_same_weights), so the GPU test rebuilds them from the seed.

Saved to tests/golden/zoo_segmenter_forward.npz: the input cloud, the logits in eval mode and in training mode (batch
statistics), and d(sum of logits * cot)/d(cloud) in eval mode.  tests/golden/zoo_classifier_forward.npz: the same for the
ScanObjectNN classifier (model_zoo/scanobject/classifier.py: MultiHeadPool, Res2D/3D blocks, class + mask heads), eval mode;
tests/golden/zoo_inpainter_forward.npz: the completion inpainter (encoder, style mapping, twelve AdaIN decoder blocks), eval mode;
tests/golden/zoo_reconstructor_decoder.npz: the What3D reconstruction decoder from a given style vector + its Chamfer loss.

usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_zoo_forward.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G          # noqa: E402

SEED, B, N = 0, 2, 512


def perturb(model, seed):
    """Move every parameter off its initial value, deterministically (tests/test_zoo_gpu.py applies the same function):
    at initialisation the blocks' key BatchNorm weights and the AdaIN residual scales are exactly zero, which would leave
    the learned-key paths out of the comparison."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(torch.randn(p.shape, generator=g) * (0.01 if p.dim() >= 2 else 0.05))


def main():
    G._install_shims()
    if G.REF not in sys.path:
        sys.path.insert(0, G.REF)
    ns = {"__name__": "zoo_model"}
    with open(os.path.join(G.REF, "model_zoo/s3dis/segmenter.py")) as f:
        exec(compile(f.read(), "segmenter.py", "exec"), ns)
    torch.manual_seed(SEED)
    model = ns["Model"]()
    perturb(model, SEED + 2)
    g = torch.Generator().manual_seed(SEED + 1)
    xyz = torch.rand(B, 3, N, generator=g) * 2 - 1
    rgb = torch.rand(B, 3, N, generator=g)
    cloud = torch.cat([xyz, rgb], dim=1)[:, :, None].contiguous()          # [B, 6, 1, N] as the loaders deliver it
    cot = torch.randn(B, 13, 1, N, generator=g)
    model.eval()
    x = cloud.clone().requires_grad_(True)
    out_eval, _ = model(x)
    (out_eval * cot).sum().backward()
    g_cloud = x.grad.clone()
    # training mode (batch statistics): the twelve re-normalising blocks amplify rounding differences between any two
    # implementations (block by block they agree to 1e-5), so the golden also keeps the first 64 channels after each of
    # the first three blocks, where a tight comparison is meaningful
    model.train()
    taps = []
    hooks = [model.attentions_encoder[i].register_forward_hook(lambda m, a, o: taps.append(o[0][:, :64].detach().clone()))
             for i in range(3)]
    with torch.no_grad():
        out_train, _ = model(cloud)
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(HERE, "zoo_segmenter_forward.npz"), seed=SEED, cloud=cloud.numpy(), cot=cot.numpy(),
                        out_eval=out_eval.detach().numpy(), g_cloud=g_cloud.numpy(), out_train=out_train.numpy(),
                        train_block1=taps[0].numpy(), train_block2=taps[1].numpy(), train_block3=taps[2].numpy())
    print("saved", out_eval.shape, float(out_eval.abs().max()), float(g_cloud.abs().max()))

    # ScanObjectNN classifier, eval mode (dropout off)
    ns = {"__name__": "zoo_model"}
    with open(os.path.join(G.REF, "model_zoo/scanobject/classifier.py")) as f:
        exec(compile(f.read(), "classifier.py", "exec"), ns)
    torch.manual_seed(SEED)
    model = ns["Model"]()
    perturb(model, SEED + 2)
    model.eval()
    cloud = (torch.rand(B, 3, N, generator=g) * 2 - 1)[:, :, None].contiguous()
    cot_c = torch.randn(B, 15, generator=g)
    cot_m = torch.randn(B, 1, 1, N, generator=g)
    x = cloud.clone().requires_grad_(True)
    cls, mask, _ = model(x)
    ((cls * cot_c).sum() + (mask * cot_m).sum()).backward()
    np.savez_compressed(os.path.join(HERE, "zoo_classifier_forward.npz"), seed=SEED, cloud=cloud.numpy(), cot_cls=cot_c.numpy(),
                        cot_mask=cot_m.numpy(), cls=cls.detach().numpy(), mask=mask.detach().numpy(), g_cloud=x.grad.numpy())
    print("saved", cls.shape, mask.shape, float(cls.abs().max()), float(mask.abs().max()), float(x.grad.abs().max()))

    # completion inpainter (model_zoo/completion/inpainter.py): encoder -> style vector -> AdaIN decoder over a noise cloud
    ns = {"__name__": "zoo_model"}
    with open(os.path.join(G.REF, "model_zoo/completion/inpainter.py")) as f:
        exec(compile(f.read(), "inpainter.py", "exec"), ns)
    torch.manual_seed(SEED)
    model = ns["Model"]()
    perturb(model, SEED + 2)
    model.eval()
    part = (torch.rand(B, 3, 256, generator=g) * 2 - 1)[:, :, None].contiguous()            # partial cloud to the encoder
    sphere = torch.nn.functional.normalize(torch.randn(B, 3, N, generator=g), dim=1)          # noise on the unit sphere
    noise = torch.cat([sphere, (torch.rand(B, 1, N, generator=g) > 0.5).float()], dim=1)      # + the "is a real point" label
    cot_r = torch.randn(B, 3, 1, N, generator=g)
    nz = noise.clone().requires_grad_(True)
    pt = part.clone().requires_grad_(True)
    # every decoder layer re-normalises per cloud (instance norm), which amplifies rounding differences layer by layer like
    # training-mode BatchNorm does: keep the style vector and the first 64 channels after the stem and the first two
    # decoder blocks for tight comparisons
    taps = {}
    hooks = [model.mapping.register_forward_hook(lambda m, a, o: taps.__setitem__("z", o.detach().clone())),
             model.attentions_decoder[0].register_forward_hook(lambda m, a, o: taps.__setitem__("dec1", o[0][:, :64].detach().clone())),
             model.attentions_decoder[1].register_forward_hook(lambda m, a, o: taps.__setitem__("dec2", o[0][:, :64].detach().clone()))]
    # ... and, for the backward, the cotangent arriving at the inputs of the last two decoder blocks
    hooks += [model.attentions_decoder[11].register_full_backward_hook(lambda m, gi, go: taps.__setitem__("g_dec12", gi[0][:, :64].detach().clone())),
              model.attentions_decoder[10].register_full_backward_hook(lambda m, gi, go: taps.__setitem__("g_dec11", gi[0][:, :64].detach().clone()))]
    rec, _ = model(nz, pt)
    (rec * cot_r).sum().backward()
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(HERE, "zoo_inpainter_forward.npz"), seed=SEED, part=part.numpy(), noise=noise.numpy(),
                        cot=cot_r.numpy(), rec=rec.detach().numpy(), g_noise=nz.grad.numpy(), g_part=pt.grad.numpy(),
                        z=taps["z"].numpy(), dec1=taps["dec1"].numpy(), dec2=taps["dec2"].numpy(),
                        g_dec12=taps["g_dec12"].numpy(), g_dec11=taps["g_dec11"].numpy())
    print("saved", rec.shape, float(rec.abs().max()), float(nz.grad.abs().max()), float(pt.grad.abs().max()))
    gen_reconstructor_decoder(g)


def gen_reconstructor_decoder(g):
    """What3D single-view reconstruction (model_zoo/image_reconstruction/reconstructor.py:26-92), DECODER ONLY: the image
    encoder is torchvision's pretrained ResNet-50 (not in this image, no network), so the fixture starts from a given
    style vector z — what `mapping` hands to the decoder (:84) — and runs the reference's own start / twelve
    MultiHeadUnionAdaIn / final stack on its own layers, then the training loss's Chamfer term (PCN-style
    loss_chamfer_adj, dist_chamfer.py:80-89, with the reference's pure-torch distances chamfer_pytorch.py:4-14 since the
    CUDA extension cannot run here).  `torchvision.models` is a stand-in that only lets the class construct: nothing of
    it is executed.  Besides the outputs the fixture keeps the INPUT of the last decoder block and the cotangent at its
    output, so that the block's gradients can be compared on identical inputs (whole-model gradients drift, see
    tests/test_zoo_gpu.py)."""
    import types
    tv, tvm = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")
    tvm.resnet50 = lambda pretrained=False: torch.nn.Sequential(torch.nn.Identity(), torch.nn.Identity(), torch.nn.Identity())
    tv.models = tvm
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tvm
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_chamfer_pytorch", os.path.join(G.REF, "chamfer_extension", "chamfer_pytorch.py"))
    cp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cp)
    ns = {"__name__": "zoo_model"}
    with open(os.path.join(G.REF, "model_zoo/image_reconstruction/reconstructor.py")) as f:
        exec(compile(f.read(), "reconstructor.py", "exec"), ns)
    torch.manual_seed(SEED)
    model = ns["Model"]()
    perturb(model, SEED + 2)
    model.eval()
    Bd, Nd = 2, 256
    z = torch.relu(torch.randn(Bd, 512, generator=g)).requires_grad_(True)                  # post-ReLU style vector
    noise = torch.nn.functional.normalize(torch.randn(Bd, 3, Nd, generator=g), dim=1).requires_grad_(True)   # utils/pcd_utils.py:5-13
    target = torch.rand(Bd, 3, 1, Nd, generator=g)
    fs = ns["forward_style"]
    taps = {}
    last = model.attentions_decoder[11]
    hooks = [model.attentions_decoder[0].register_forward_hook(lambda m, a, o: taps.__setitem__("dec1", o[0][:, :64].detach().clone())),
             last.register_forward_pre_hook(lambda m, a: taps.__setitem__("x11", a[0].detach().clone())),
             last.register_forward_hook(lambda m, a, o: taps.__setitem__("x12", o[0])),
             last.register_full_backward_hook(lambda m, gi, go: (taps.__setitem__("g_x11", gi[0].detach().clone()),
                                                                  taps.__setitem__("g_x12", go[0].detach().clone())) and None)]
    x = fs(model.start, noise, z)
    for blk in model.attentions_decoder:
        x, _ = blk(x, z, noise)
    rec = fs(model.final, x, z).unsqueeze(2)                                                 # [B,3,1,N], sigmoid output
    d0, d1 = cp.dist_chamfer(rec[:, :, 0].permute(0, 2, 1), target[:, :, 0].permute(0, 2, 1))
    loss = (torch.sqrt(d0.clamp_min(0)).mean() + torch.sqrt(d1.clamp_min(0)).mean()) / 2    # loss_chamfer_adj
    loss.backward()
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(HERE, "zoo_reconstructor_decoder.npz"), seed=SEED, z=z.detach().numpy(), noise=noise.detach().numpy(),
                        target=target.numpy(), rec=rec.detach().numpy(), loss=float(loss), dec1=taps["dec1"].numpy(),
                        # (AdaIN blocks normalise per cloud: cloud 0 alone is a self-contained block-level case)
                        x11=taps["x11"][:1].numpy(), g_x12=taps["g_x12"][:1].numpy(), g_x11=taps["g_x11"][:1].numpy(),
                        g_z=z.grad.numpy(), g_noise=noise.grad.numpy())
    print("saved reconstructor decoder", rec.shape, float(loss), float(z.grad.abs().max()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "reconstructor":        # this fixture alone
        G._install_shims()
        if G.REF not in sys.path:
            sys.path.insert(0, G.REF)
        gen_reconstructor_decoder(torch.Generator().manual_seed(SEED + 7))
    else:
        main()
