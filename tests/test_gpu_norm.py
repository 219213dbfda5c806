"""GPU: the block settings no reference entry point passes — norm_type=2 (BatchNorm3d, joint_model.py:12-13) and soft=True (Softplus,
joint_model.py:38,104) — through the native general normalisation path (ops.NormAct, vaeseg.h vs_norm_*).

  * the op itself against torch.nn.BatchNorm3d / InstanceNorm3d + ReLU / Softplus autograd on the CPU (fp32 2e-5; bf16 / fp16 looser),
    incl. negative scales, padded channels, running statistics and eval mode;
  * the reference's blocks and Segmentation(norm_type=2) against goldens from the unmodified reference (tests/golden/blocks_norm.npz,
    seg32_bn.npz, oracle/make_golden.py);
  * a HIP-graph replayed train step whose BatchNorm buffers advance on every replay."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from tests import golden_util as G
from tests.test_oracle_golden import BLOCKS_NORM, check_bn_buffers, run_block_norm

pytestmark = pytest.mark.gpu

DT = {"fp32": (torch.float32, 2e-5), "bf16": (torch.bfloat16, 1.5e-2), "fp16": (torch.float16, 2e-3)}


def _mods():
    import joint_model
    from oracle import ref_cpu as O
    from vae_segmentation_amd import ops, optim
    from vae_segmentation_amd import train as T
    return joint_model, O, T, ops, optim


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("dt", sorted(DT))
@pytest.mark.parametrize("norm", ["instance", "batch", "batch_eval"])
@pytest.mark.parametrize("act", ["relu", "softplus"])
@pytest.mark.parametrize("c_real,c", [(16, 16), (3, 8), (64, 64)])
def test_norm_act_op_vs_torch(dt, norm, act, c_real, c):
    """y = act(norm(x) * gamma + beta) and its backward on a channels-last tensor with padded channels, against torch on the CPU in
    fp32 on the same (storage-rounded) operands."""
    M, O, T, ops, optim = _mods()
    dtype, tol = DT[dt]
    n, side = 2, 10
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, c_real, side, side, side, generator=gen) * 1.5 + 0.3
    gy = torch.randn(n, c_real, side, side, side, generator=gen)
    xq, gq = x.to(dtype).float(), gy.to(dtype).float()
    if norm == "instance":
        ref_norm, bn = nn.InstanceNorm3d(c_real), None
    else:
        ref_norm = nn.BatchNorm3d(c_real, momentum=0.1)
        with torch.no_grad():
            ref_norm.weight.copy_(torch.linspace(-1.2, 1.4, c_real))          # negative scales too
            ref_norm.bias.copy_(torch.linspace(-0.3, 0.2, c_real))
            ref_norm.running_mean.copy_(torch.linspace(-0.5, 0.5, c_real))
            ref_norm.running_var.copy_(torch.linspace(0.5, 2.0, c_real))
        bn = nn.BatchNorm3d(c_real, momentum=0.1)
        bn.load_state_dict(ref_norm.state_dict())
        bn = bn.cuda()
        if norm == "batch_eval":
            ref_norm.eval(); bn.eval()
    ref_act = nn.ReLU() if act == "relu" else nn.Softplus()
    xr = xq.clone().requires_grad_(True)
    yr = ref_act(ref_norm(xr))
    yr.backward(gq)
    # channels-last, padded
    xcl = torch.zeros(n, side, side, side, c); xcl[..., :c_real] = xq.permute(0, 2, 3, 4, 1)
    gcl = torch.zeros(n, side, side, side, c); gcl[..., :c_real] = gq.permute(0, 2, 3, 4, 1)
    xd = xcl.to(dtype).cuda().requires_grad_(True)
    xs = ops.instnorm_stats(xd.detach())
    yd = ops.NormAct.apply(xd, xs, bn.weight if bn is not None else None, bn.bias if bn is not None else None, bn,
                           ops.VS_ACT_SOFTPLUS if act == "softplus" else ops.VS_ACT_RELU, c_real)
    yd.backward(gcl.to(dtype).cuda())
    torch.cuda.synchronize()
    if c > c_real:
        assert float(yd.detach()[..., c_real:].abs().max()) == 0.0
    assert _rel(yd[..., :c_real].permute(0, 4, 1, 2, 3), yr) < tol
    assert _rel(xd.grad[..., :c_real].permute(0, 4, 1, 2, 3), xr.grad) < 4 * tol
    if c > c_real:
        assert float(xd.grad[..., c_real:].abs().max()) == 0.0
    if bn is not None:
        assert _rel(bn.weight.grad, ref_norm.weight.grad) < 4 * tol
        assert _rel(bn.bias.grad, ref_norm.bias.grad) < 4 * tol
        assert _rel(bn.running_mean, ref_norm.running_mean) < 1e-5
        assert _rel(bn.running_var, ref_norm.running_var) < 1e-5
        assert int(bn.num_batches_tracked) == int(ref_norm.num_batches_tracked)


@pytest.mark.parametrize("tag", sorted(BLOCKS_NORM))
def test_blocks_norm_vs_reference_golden(tag):
    M, O, T, ops, optim = _mods()
    g = G.load("blocks_norm")
    native = {"conv": M.Conv, "dconv": M.DoubleConv, "down": M.Down, "up": M.Up}[tag.split("_")[0]]
    a, b = [int(v) for v in tag.split("_")[-2:]]
    mod = native(a, b, norm_type=2 if "_bn" in tag else (3 if "_gs_" in tag else 1), soft="_soft" in tag)
    mod = O.bn_fill_(O.deterministic_fill_(mod, seed=int(g[tag + ".seed"])))
    if "_gs_" in tag:                      # GSNorm3d blocks (norm_type=3): positive weights on positive inputs, as the fixture was made
        O.positive_fill_(mod)
    mod = mod.cuda()
    y, gin = run_block_norm(mod, g, tag, device="cuda")
    G.check_tensor(g, tag + ".out", y, rtol=1e-3, what=tag)
    G.check_tensor(g, tag + ".gin", gin, rtol=1e-3, what=tag)
    # the bias of a conv whose output is normalised with batch / instance statistics has an exactly zero gradient (rounding noise in the
    # reference, as under InstanceNorm: SURVEY F10); in eval mode the running statistics do not cancel it and it is live
    dead = set()
    for name, m in mod.named_modules():
        if isinstance(m, nn.Sequential):
            for i in range(len(m) - 1):
                if isinstance(m[i], nn.Conv3d) and isinstance(m[i + 1], (nn.BatchNorm3d, nn.InstanceNorm3d)) and "_eval_" not in tag:
                    dead.add((name + "." if name else "") + "%d.bias" % i)
    G.check_grads(g, tag, [(n, p.grad) for n, p in mod.named_parameters()], rtol=1e-3, what=tag, dead=lambda n: n in dead)
    check_bn_buffers(g, tag, mod, 1e-4)


def test_state_dict_contract_norm_type_2():
    M, O, T, ops, optim = _mods()
    seg, oseg = M.Segmentation(1, 2), O.Segmentation(1, 2)                       # the constructors' default: BatchNorm3d
    sa, sb = seg.state_dict(), oseg.state_dict()
    assert list(sa.keys()) == list(sb.keys()) and all(sa[k].shape == sb[k].shape for k in sa)
    assert any(k.endswith("running_var") for k in sa)
    d = M.DoubleConv(8, 16, norm_type=1, soft=True)
    assert isinstance(d.conv[2], nn.Softplus) and d.conv[2] is d.conv[5] is d.conv[8]


def test_seg32_bn_vs_reference_golden_and_train_steps():
    """Segmentation(norm_type=2): loss, prediction, gradients (incl. the BatchNorm affine pairs) and running statistics against the
    reference; then HIP-graph replayed SGD steps: the buffers advance on every replay and the loss goes down."""
    M, O, T, ops, optim = _mods()
    g = G.load("seg32_bn")
    seg = O.bn_fill_(O.deterministic_fill_(M.Segmentation(1, 2, norm_type=2), seed=0)).cuda()
    img, lab = O.synthetic_image(2, 32, 2).cuda(), O.synthetic_label(2, 32, 3).cuda()
    loss, aux = T.seg_train_losses(seg, img, lab)
    loss.backward()
    G.scalar_close(g, "dice_loss", loss.item(), 1e-3)
    G.check_tensor_f64(g, "pred", aux["batch"]["pred"], k=256, floor=1e-3)
    rep = G.check_grads_f64(g, "seg", [(n, p.grad) for n, p in seg.named_parameters()], floor=2e-3)
    G.vacuity(rep, "seg32_bn")
    check_bn_buffers(g, "seg", seg, 1e-4)
    params = list(seg.parameters())
    for p in params:
        p.grad = None
    opt = optim.SGD(params, lr=1e-2, momentum=0.9)
    # an eager pass whose graph is still alive must be refused with an explanation (its gradient accumulators live on the default stream;
    # capturing over them used to crash inside hipStreamEndCapture), and accepted once the graph is gone
    with pytest.raises(RuntimeError, match="earlier eager pass"):
        T.GraphedStep(lambda: T.seg_train_losses(seg, img, lab), params, opt, warmup=1)
    del loss, aux
    gs = T.GraphedStep(lambda: T.seg_train_losses(seg, img, lab), params, opt, warmup=1)
    tracked0 = int(seg.in_block.conv[1].num_batches_tracked)
    rm0 = seg.in_block.conv[1].running_mean.clone()
    losses = []
    for _ in range(6):
        gs.step()
        losses.append(float(gs.loss.item()))
    assert int(seg.in_block.conv[1].num_batches_tracked) == tracked0 + 6         # one per replay (the capture itself runs nothing)
    assert not torch.equal(rm0, seg.in_block.conv[1].running_mean)
    assert losses[-1] < losses[0], losses
    assert all(torch.isfinite(p).all() for p in params)


def test_bn_eval_mode_uses_running_statistics():
    """model.eval(): a batch-1 forward equals the CPU oracle's with the same running statistics (validation path of a norm_type=2 model)"""
    M, O, T, ops, optim = _mods()
    seg = O.bn_fill_(O.deterministic_fill_(M.Segmentation(1, 2, norm_type=2), seed=0)).cuda()
    oseg = O.bn_fill_(O.deterministic_fill_(O.Segmentation(1, 2, norm_type=2), seed=0))
    img = O.synthetic_image(2, 32, 2)
    with torch.no_grad():
        seg({"img": img.cuda()}, "img", "pred"); oseg({"img": img}, "img", "pred")          # one training pass each
        seg.eval(); oseg.eval()
        a = seg({"img": img[:1].cuda()}, "img", "pred")["pred"]
        b = oseg({"img": img[:1]}, "img", "pred")["pred"]
    assert _rel(a, b) < 1e-3
