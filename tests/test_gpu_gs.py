"""GPU: the `*_GS` model family (joint_model.py:17-33,54-99,140-202,307-346; instantiated nowhere in the reference) on the native kernels
— vs_gsnorm_*, vs_upsample_trilinear_*, vs_softmax2_fwd, conv + activation without normalisation — against goldens the reference's own
classes produced (tests/golden/gs.npz, oracle/make_golden.py) and per-op against torch on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from tests import golden_util as G
from tests.test_oracle_golden import GS_BLOCKS, run_gs_block

pytestmark = [pytest.mark.gpu, pytest.mark.filterwarnings("ignore:.*align_corners.*")]

DT = {"fp32": (torch.float32, 2e-5), "bf16": (torch.bfloat16, 1.5e-2), "fp16": (torch.float16, 2e-3)}


def _mods():
    import joint_model
    from oracle import ref_cpu as O
    from vae_segmentation_amd import ops
    return joint_model, O, ops


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("dt", sorted(DT))
@pytest.mark.parametrize("scale,side,c", [(2, 6, 16), (4, 5, 32), (8, 3, 64), (2, 7, 8)])
def test_upsample_trilinear_vs_torch(dt, scale, side, c):
    M, O, ops = _mods()
    dtype, tol = DT[dt]
    gen = torch.Generator().manual_seed(scale)
    x = torch.randn(2, c, side, side + 1, side + 2, generator=gen).to(dtype).float()
    gy = torch.randn(2, c, side * scale, (side + 1) * scale, (side + 2) * scale, generator=gen).to(dtype).float()
    xr = x.clone().requires_grad_(True)
    yr = nn.Upsample(scale_factor=scale, mode="trilinear")(xr)
    yr.backward(gy)
    xd = x.permute(0, 2, 3, 4, 1).contiguous().to(dtype).cuda().requires_grad_(True)
    yd = ops.UpsampleTrilinear.apply(xd, scale)
    yd.backward(gy.permute(0, 2, 3, 4, 1).contiguous().to(dtype).cuda())
    torch.cuda.synchronize()
    assert _rel(yd.permute(0, 4, 1, 2, 3), yr) < tol
    assert _rel(xd.grad.permute(0, 4, 1, 2, 3), xr.grad) < tol


@pytest.mark.parametrize("dt", sorted(DT))
@pytest.mark.parametrize("c,groups", [(8, 2), (16, 4), (32, 8), (64, 8), (16, 1)])
def test_gsnorm_and_softmax2_vs_torch(dt, c, groups):
    M, O, ops = _mods()
    dtype, tol = DT[dt]
    gen = torch.Generator().manual_seed(c)
    x = (torch.rand(2, c, 5, 6, 7, generator=gen) + 0.05).to(dtype).float()
    gy = torch.randn(2, c, 5, 6, 7, generator=gen).to(dtype).float()
    xr = x.clone().requires_grad_(True)
    yr = O.GSNorm3d(c, groups)(xr)
    yr.backward(gy)
    xd = x.permute(0, 2, 3, 4, 1).contiguous().to(dtype).cuda().requires_grad_(True)
    yd = ops.GSNorm.apply(xd, groups)
    yd.backward(gy.permute(0, 2, 3, 4, 1).contiguous().to(dtype).cuda())
    assert _rel(yd.permute(0, 4, 1, 2, 3), yr) < tol
    assert _rel(xd.grad.permute(0, 4, 1, 2, 3), xr.grad) < 4 * tol
    # softmax over the first two channels, planar fp32 out
    lr = x[:, :2].clone().requires_grad_(True)
    pr = torch.softmax(lr * 3, 1)
    gp = torch.randn(pr.shape, generator=gen)
    pr.backward(gp)
    ld = (x * 3).permute(0, 2, 3, 4, 1).contiguous().to(dtype).cuda().requires_grad_(True)
    lq = ld.detach().float().cpu().permute(0, 4, 1, 2, 3)[:, :2].clone().requires_grad_(True)       # what the kernel sees after rounding
    pq = torch.softmax(lq, 1)
    pq.backward(gp)
    pd = ops.Softmax2.apply(ld)
    pd.backward(gp.cuda())
    assert _rel(pd, pq) < 1e-5
    assert _rel(ld.grad[..., :2].permute(0, 4, 1, 2, 3), lq.grad) < max(tol, 1e-4)
    assert float(ld.grad[..., 2:].abs().max()) == 0.0


@pytest.mark.parametrize("tag", sorted(GS_BLOCKS))
def test_gs_blocks_vs_reference_golden(tag):
    M, O, ops = _mods()
    g = G.load("gs")
    mod = O.deterministic_fill_(GS_BLOCKS[tag](M), seed=int(g[tag + ".seed"])).cuda()
    y, gin = run_gs_block(mod, g, tag, device="cuda")
    G.check_tensor(g, tag + ".out", y, rtol=1e-3, what=tag)
    G.check_tensor(g, tag + ".gin", gin, rtol=1e-3, what=tag)
    G.check_grads(g, tag, [(n, p.grad) for n, p in mod.named_parameters()], rtol=1e-3, what=tag)


def test_segmentation_gs_vs_reference_golden():
    M, O, ops = _mods()
    from vae_segmentation_amd.evaluation import avg_dsc
    g = G.load("gs")
    seg, oseg = M.Segmentation_GS(1, 2), O.Segmentation_GS(1, 2)
    assert list(seg.state_dict().keys()) == list(oseg.state_dict().keys())
    assert all(a.shape == b.shape for a, b in zip(seg.state_dict().values(), oseg.state_dict().values()))
    seg = O.deterministic_fill_(seg, seed=0).cuda()
    img, lab = O.synthetic_image(2, 32, 2).cuda(), O.synthetic_label(2, 32, 3).cuda()
    batch = seg({"img": img, "gt": ops.onehot(lab, 2)}, "img", "pred")
    loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=2, eps=1e-4)
    loss.backward()
    G.scalar_close(g, "seg.dice_loss", loss.item(), 1e-3)
    G.check_tensor_f64(g, "seg.pred", batch["pred"], k=256, floor=1e-3)
    rep = G.check_grads_f64(g, "seg", [(n, p.grad) for n, p in seg.named_parameters()], floor=2e-3)
    G.vacuity(rep, "seg_gs32")
