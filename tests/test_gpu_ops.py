"""GPU parity, op level: every libvaeseg kernel against the stock fp32 PyTorch op it replaces, run on the CPU.

Tolerances: fp32 kernels (exact-f32 MFMA, fp32/fp64 reductions) 2e-5 relative to the tensor's max magnitude;
bf16 / fp16 kernels are compared with the same fp32 reference evaluated on operands rounded to that format, 1.5e-2 / 2e-3 (one
rounding of the stored result is 2^-9 = 2e-3 in bf16, 2^-12 in fp16; the lazy InstanceNorm input adds another rounding)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16, torch.float16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 1.5e-2, torch.float16: 2e-3}      # fp16: 11 significand bits against bf16's 8


def _ops():
    from vae_segmentation_amd import ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def q(t, dtype):
    """round to the kernel dtype and back (what the kernel sees)."""
    return t.detach().to(dtype).float().clone()


def to_cl(x, cp, dtype):
    n, c = x.shape[:2]
    out = torch.zeros((n,) + tuple(x.shape[2:]) + (cp,), dtype=torch.float32)
    out[..., :c] = x.detach().permute(0, 2, 3, 4, 1)
    return out.to(dtype).cuda().contiguous()


def from_cl(y, c):
    return y.float().cpu()[..., :c].permute(0, 4, 1, 2, 3).contiguous()


def relerr(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def in_relu(x):
    return torch.relu(F.instance_norm(x))


CONV_CASES = [  # (N, Cin, Cout, D, H, W)
    (2, 8, 8, 8, 8, 16), (1, 8, 16, 5, 6, 20), (1, 16, 8, 4, 4, 16), (1, 16, 16, 6, 5, 7), (1, 32, 16, 4, 4, 8),
    (1, 16, 32, 4, 8, 16), (1, 32, 32, 3, 3, 3), (1, 64, 32, 4, 4, 4), (1, 32, 64, 6, 6, 6), (1, 64, 128, 3, 3, 3),
    (1, 128, 64, 3, 3, 3), (2, 256, 256, 3, 3, 3), (1, 2, 8, 8, 8, 8), (1, 1, 8, 4, 4, 16),
    (1, 128, 128, 8, 8, 8), (1, 64, 64, 16, 16, 16), (2, 128, 64, 8, 8, 8),
    # small-volume kernel (igemm_k3s.h, padded volume <= 512 voxels): ragged box, three samples; two voxels;
    # the largest cube it takes (6^3) next to volumes it leaves to k3b
    (3, 64, 32, 5, 6, 6), (2, 64, 64, 1, 1, 2), (1, 32, 64, 6, 6, 6), (1, 32, 64, 8, 8, 8), (1, 32, 32, 7, 6, 6),
]


# shapes that make the persistent 3x3x3 kernels walk several tiles per workgroup (more tiles than the grid), cross a sample
# boundary mid-walk (statistics flush), use 32-row weight blocks, two channel chunks, and ragged edges in all three axes
# the last two take the tall-tile (4x8x16) variant of the bf16 kernel (one wave of 256..1024 workgroups), one with a ragged y edge
# the 8 -> 8 channel cases run the Toeplitz kernel (igemm_k3t.h, 4x8x32 tiles): ragged in all three axes with three samples; more tiles
# (576, four samples) than the persistent grid (512); 2 real input channels of 8
CONV_CASES_LARGE = [(2, 8, 8, 48, 48, 64), (2, 32, 32, 16, 32, 64), (1, 16, 32, 37, 30, 50), (3, 64, 32, 9, 10, 21),
                    (2, 16, 16, 32, 64, 64), (1, 8, 8, 32, 100, 48), (3, 8, 8, 7, 9, 37), (4, 8, 8, 24, 48, 128), (2, 2, 8, 12, 20, 70)]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", CONV_CASES_LARGE)
def test_conv_k3_fwd_bwd_large(case, dtype):
    # the channel sum is a near-cancelling sum of ~150k outputs: its bf16-staging noise grows like sqrt(voxels) against the
    # sqrt(sum of squares) scale used below, hence the wider statistic tolerance at these sizes
    test_conv_k3_fwd_bwd(case, True, dtype, stat_tol=2.0)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("lazy", [False, True])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_k3_fwd_bwd(case, lazy, dtype, stat_tol=1.0):
    ops = _ops()
    n, cin, cout, d, h, w = case
    x = rnd(n, cin, d, h, w, seed=1)
    wt = rnd(cout, cin, 3, 3, 3, seed=2, scale=(3.0 / (27 * cin)) ** 0.5)
    gy = rnd(n, cout, d, h, w, seed=3)
    xq, wq, gq = q(x, dtype).requires_grad_(True), q(wt, dtype).requires_grad_(True), q(gy, dtype)
    a = in_relu(xq) if lazy else xq
    if lazy and dtype == torch.bfloat16:
        pass  # the kernel normalises in fp32 and rounds the activation to bf16 when staging: covered by TOL
    y_ref = F.conv3d(a, wq, None, padding=1)
    (y_ref * gq).sum().backward()

    x_cl = to_cl(x, ops.cpad(cin), dtype).requires_grad_(True)
    xs = ops.instnorm_stats(x_cl.detach()) if lazy else None
    w_gpu = wt.cuda().requires_grad_(True)
    if dtype == torch.bfloat16:
        w_gpu = q(wt, dtype).cuda().requires_grad_(True)
    y, ys = ops.ConvK3.apply(x_cl, xs, w_gpu, None)
    torch.cuda.synchronize()
    tol = TOL[dtype]
    assert relerr(from_cl(y, cout), y_ref.detach()) < tol
    yr = q(y_ref.detach(), dtype).double()
    st = ops.stats_total(ys).cpu()[:, :cout]
    ref_sum, ref_sq = yr.sum((2, 3, 4)), (yr * yr).sum((2, 3, 4))
    assert float((st[..., 0] - ref_sum).abs().max() / ref_sq.sqrt().max()) < tol * stat_tol
    assert float((st[..., 1] - ref_sq).abs().max() / ref_sq.max()) < tol * stat_tol
    if ops.cpad(cout) > cout:
        assert float(y.float()[..., cout:].abs().max()) == 0.0
    y.backward(to_cl(gy, ops.cpad(cout), dtype))
    torch.cuda.synchronize()
    gtol = tol * (4 if lazy else 1)
    assert relerr(from_cl(x_cl.grad, cin), xq.grad) < gtol
    assert relerr(w_gpu.grad.cpu(), wq.grad) < gtol


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [(3, 96, 32, 5, 6, 6), (2, 160, 96, 2, 3, 4)])
def test_conv_k3_small_volume_odd_chunk_counts(case, dtype):
    """3 and 5 chunks of 32 channels: the small-volume kernel keeps two stages in flight, the last stage of an odd count runs alone.
    (Materialised input: the statistics kernels only take the model's channel counts.)"""
    test_conv_k3_fwd_bwd(case, False, dtype)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("lazy", [False, True])
@pytest.mark.parametrize("case", [(1, 32, 32, 1, 1, 54), (2, 32, 64, 30, 2, 2), (1, 64, 32, 1, 7, 16), (2, 32, 32, 2, 30, 2)])
def test_conv_k3_small_volume_extreme_aspects(case, lazy, dtype):
    """The small-volume kernel stages the REAL voxels of the padded sample (igemm_k3s.h, end of round 6): a thread's fragment is voxel rv -> (z, y, x) by
    reciprocal multiplication.  Volumes far from cubic — one long axis up to the 512-voxel padded limit — put the largest divisors and the most padding there."""
    test_conv_k3_fwd_bwd(case, lazy, dtype)


K2_CASES = [(2, 8, 8, 8, 16), (1, 16, 4, 6, 10), (1, 32, 4, 4, 4), (1, 64, 2, 2, 6), (1, 128, 2, 2, 2), (2, 256, 2, 2, 2)]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("lazy", [False, True])
@pytest.mark.parametrize("case", K2_CASES)
def test_conv_k2s2_fwd_bwd(case, lazy, dtype):
    ops = _ops()
    n, c, d, h, w = case
    x = rnd(n, c, d, h, w, seed=4)
    wt = rnd(c, c, 2, 2, 2, seed=5, scale=(3.0 / (8 * c)) ** 0.5)
    b = rnd(c, seed=6, scale=0.1)
    gy = rnd(n, c, d // 2, h // 2, w // 2, seed=7)
    xq, wq, bq, gq = q(x, dtype).requires_grad_(True), q(wt, dtype).requires_grad_(True), b.clone().requires_grad_(True), q(gy, dtype)
    a = in_relu(xq) if lazy else xq
    y_ref = F.conv3d(a, wq, bq, stride=2)
    (y_ref * gq).sum().backward()
    x_cl = to_cl(x, c, dtype).requires_grad_(True)
    xs = ops.instnorm_stats(x_cl.detach()) if lazy else None
    w_gpu, b_gpu = q(wt, dtype).cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.ConvK2S2.apply(x_cl, xs, w_gpu, b_gpu)
    tol = TOL[dtype]
    assert relerr(from_cl(y, c), y_ref.detach()) < tol
    y.backward(to_cl(gy, c, dtype))
    torch.cuda.synchronize()
    gtol = tol * (4 if lazy else 1)
    assert relerr(from_cl(x_cl.grad, c), xq.grad) < gtol
    assert relerr(w_gpu.grad.cpu(), wq.grad) < gtol
    assert relerr(b_gpu.grad.cpu(), bq.grad) < gtol


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("lazy", [False, True])
@pytest.mark.parametrize("case", [(2, 16, 4, 4, 8), (1, 32, 3, 5, 7), (1, 64, 2, 2, 3), (1, 128, 3, 3, 3), (2, 256, 2, 2, 2)])
def test_conv_transpose_fwd_bwd(case, lazy, dtype):
    ops = _ops()
    n, c, d, h, w = case
    x = rnd(n, c, d, h, w, seed=8)
    wt = rnd(c, c, 2, 2, 2, seed=9, scale=(3.0 / c) ** 0.5)
    b = rnd(c, seed=10, scale=0.1)
    gy = rnd(n, c, 2 * d, 2 * h, 2 * w, seed=11)
    xq, wq, bq, gq = q(x, dtype).requires_grad_(True), q(wt, dtype).requires_grad_(True), b.clone().requires_grad_(True), q(gy, dtype)
    a = in_relu(xq) if lazy else xq
    y_ref = F.conv_transpose3d(a, wq, bq, stride=2)
    (y_ref * gq).sum().backward()
    x_cl = to_cl(x, c, dtype).requires_grad_(True)
    xs = ops.instnorm_stats(x_cl.detach()) if lazy else None
    w_gpu, b_gpu = q(wt, dtype).cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.ConvT2S2.apply(x_cl, xs, w_gpu, b_gpu)
    tol = TOL[dtype]
    assert relerr(from_cl(y, c), y_ref.detach()) < tol
    y.backward(to_cl(gy, c, dtype))
    torch.cuda.synchronize()
    gtol = tol * (4 if lazy else 1)
    assert relerr(from_cl(x_cl.grad, c), xq.grad) < gtol
    assert relerr(w_gpu.grad.cpu(), wq.grad) < gtol
    assert relerr(b_gpu.grad.cpu(), bq.grad) < gtol


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("nc", [2, 1, 3, 4, 8])
def test_out_block_softmax(dtype, nc):
    """nc = 2: the fused epilogue; other class counts (main_source.py:92-93): the plain conv + vs_softmax_cl_fwd / _bwd."""
    ops = _ops()
    n, d, h, w = 2, 6, 8, 20
    x = rnd(n, 8, d, h, w, seed=12)
    wt = rnd(nc, 8, 3, 3, 3, seed=13, scale=0.3)
    b = rnd(nc, seed=14, scale=0.2)
    gp = rnd(n, nc, d, h, w, seed=15)
    xq, wq, bq = q(x, dtype).requires_grad_(True), q(wt, dtype).requires_grad_(True), b.clone().requires_grad_(True)
    p_ref = torch.softmax(F.conv3d(in_relu(xq), wq, bq, padding=1), dim=1)
    (p_ref * gp).sum().backward()
    x_cl = to_cl(x, 8, dtype).requires_grad_(True)
    xs = ops.instnorm_stats(x_cl.detach())
    w_gpu, b_gpu = q(wt, dtype).cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    p = ops.ConvK3Softmax.apply(x_cl, xs, w_gpu, b_gpu)
    tol = TOL[dtype]
    assert relerr(p.cpu(), p_ref.detach()) < tol
    p.backward(gp.cuda())
    torch.cuda.synchronize()
    assert relerr(from_cl(x_cl.grad, 8), xq.grad) < tol * 4
    assert relerr(w_gpu.grad.cpu(), wq.grad) < tol * 4
    assert relerr(b_gpu.grad.cpu(), bq.grad) < tol * 4


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("nc,cp", [(3, 8), (5, 8), (8, 8), (3, 16)])
def test_softmax_pass_with_logit_dropout(dtype, nc, cp):
    """vs_softmax_cl_fwd / _bwd with the logit dropout of joint_model.py:386-387: against torch on the same logits times the mask the library
    exports for the same seed (vs_dropout_mask, element index = planar index)."""
    ops = _ops()
    from vae_segmentation_amd._lib import check, lib
    n, d, h, w = 2, 5, 6, 7
    vox = d * h * w
    logits = rnd(n, nc, d, h, w, seed=70, scale=3.0)
    gp = rnd(n, nc, d, h, w, seed=71)
    lq = q(logits, dtype)
    l_cl = to_cl(logits, cp, dtype)
    if cp > nc:
        l_cl[..., nc:] = 7.0                      # whatever sits in the padded channels must not reach the probabilities
    seed, pdrop = 4242, 0.3
    mask = torch.empty(n * nc * vox, dtype=torch.float32, device="cuda")
    check(lib.vs_dropout_mask(mask.data_ptr(), mask.numel(), pdrop, seed, None), "dropout_mask")
    mask = mask.view(n, nc, d, h, w).cpu()
    assert 0.5 < (mask > 0).float().mean() < 0.9
    for p_use, m in ((0.0, torch.ones_like(mask)), (pdrop, mask)):
        lr = lq.clone().requires_grad_(True)
        p_ref = torch.softmax(lr * m, dim=1)
        (p_ref * gp).sum().backward()
        prob = torch.empty((n, nc, d, h, w), dtype=torch.float32, device="cuda")
        check(lib.vs_softmax_cl_fwd(l_cl.data_ptr(), prob.data_ptr(), n, vox, cp, nc, ops.vs_dtype(l_cl), p_use, seed, None), "softmax_cl_fwd")
        gl = torch.full((n, d, h, w, cp), 9.0, dtype=dtype, device="cuda")
        check(lib.vs_softmax_cl_bwd(prob.data_ptr(), gp.cuda().contiguous().data_ptr(), gl.data_ptr(), n, vox, cp, nc, ops.vs_dtype(l_cl), p_use, seed, None),
              "softmax_cl_bwd")
        torch.cuda.synchronize()
        assert relerr(prob.cpu(), p_ref.detach()) < 1e-5
        assert float((prob.sum(1) - 1).abs().max()) < 1e-5
        assert relerr(from_cl(gl, nc), lr.grad) < TOL[dtype]
        if cp > nc:
            assert float(gl[..., nc:].float().abs().max()) == 0.0
    # refused arguments
    assert lib.vs_softmax_cl_fwd(l_cl.data_ptr(), prob.data_ptr(), n, vox, cp, 9, ops.vs_dtype(l_cl), 0.0, 0, None) != 0
    assert lib.vs_softmax_cl_fwd(l_cl.data_ptr(), prob.data_ptr(), n, vox, 12, nc, ops.vs_dtype(l_cl), 0.0, 0, None) != 0
    assert lib.vs_softmax_cl_bwd(prob.data_ptr(), None, gl.data_ptr(), n, vox, cp, nc, ops.vs_dtype(l_cl), 0.0, 0, None) != 0


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c", [8, 16, 32, 64, 128, 256])
def test_materialize_skip_add(c, dtype):
    ops = _ops()
    n, d, h, w = 2, 3, 5, 4
    x1, x2 = rnd(n, c, d, h, w, seed=16), rnd(n, c, d, h, w, seed=17) * 2 + 0.3
    g = rnd(n, c, d, h, w, seed=18)
    a1, a2 = q(x1, dtype).requires_grad_(True), q(x2, dtype).requires_grad_(True)
    ref = in_relu(a1) + in_relu(a2)
    (ref * q(g, dtype)).sum().backward()
    c1, c2 = to_cl(x1, c, dtype).requires_grad_(True), to_cl(x2, c, dtype).requires_grad_(True)
    out = ops.Materialize.apply(c1, ops.instnorm_stats(c1.detach()), c2, ops.instnorm_stats(c2.detach()))
    tol = TOL[dtype]
    assert relerr(from_cl(out, c), ref.detach()) < tol
    out.backward(to_cl(g, c, dtype))
    torch.cuda.synchronize()
    assert relerr(from_cl(c1.grad, c), a1.grad) < tol * 2
    assert relerr(from_cl(c2.grad, c), a2.grad) < tol * 2


def test_pack_unpack_planar_roundtrip():
    ops = _ops()
    for dtype in DT:
        x = rnd(2, 2, 4, 4, 8, seed=19)
        cl = ops.PackPlanar.apply(x.cuda(), dtype)
        assert cl.shape == (2, 4, 4, 8, 8)
        assert torch.equal(from_cl(cl, 2), q(x, dtype))
        assert float(cl.float()[..., 2:].abs().max()) == 0.0
        back = ops.UnpackPlanar.apply(cl, 2)
        assert torch.equal(back.cpu(), q(x, dtype))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c,s", [(256, 3), (32, 4), (128, 6)])
def test_linear_layers(dtype, c, s):
    """(256, 3): one k per lane (27 voxels per channel); (32, 4): four k per lane with the activation staged through LDS; (128, 6): the 96^3 bottleneck's
    27648-wide layers — four k per lane, x gathered from memory (its fp32 image does not fit the LDS); the frozen fc2's transposed copy takes the unpermuted path."""
    ops = _ops()
    b, dim = 2, 128
    feat = rnd(b, c, s, s, s, seed=20)
    w1, b1 = rnd(dim, c * s ** 3, seed=21, scale=0.02), rnd(dim, seed=22, scale=0.1)
    fq, w1r, b1r = q(feat, dtype).requires_grad_(True), w1.clone().requires_grad_(True), b1.clone().requires_grad_(True)
    for relu in (False, True):
        fq.grad = w1r.grad = b1r.grad = None
        y_ref = F.linear(fq.reshape(b, -1), w1r, b1r)
        if relu:
            y_ref = torch.relu(y_ref)
        gy = rnd(b, dim, seed=23)
        (y_ref * gy).sum().backward()
        x_cl = to_cl(feat, c, dtype).requires_grad_(True)
        wg, bg = w1.cuda().requires_grad_(True), b1.cuda().requires_grad_(True)
        y = ops.LinearCL.apply(x_cl, wg, bg, relu)
        assert relerr(y.cpu(), y_ref.detach()) < 2e-5
        y.backward(gy.cuda())
        torch.cuda.synchronize()
        assert relerr(from_cl(x_cl.grad, c), fq.grad) < TOL[dtype]
        assert relerr(wg.grad.cpu(), w1r.grad) < 2e-5
        assert relerr(bg.grad.cpu(), b1r.grad) < 2e-5
    # fc2: latent -> channels-last
    z = rnd(b, dim, seed=24).requires_grad_(True)
    w2, b2 = rnd(c * s ** 3, dim, seed=25, scale=0.1).requires_grad_(True), rnd(c * s ** 3, seed=26, scale=0.1).requires_grad_(True)
    h_ref = F.linear(z, w2, b2).view(b, c, s, s, s)
    gh = rnd(b, c, s, s, s, seed=27)
    (h_ref * q(gh, dtype)).sum().backward()
    zg, w2g, b2g = z.detach().cuda().requires_grad_(True), w2.detach().cuda().requires_grad_(True), b2.detach().cuda().requires_grad_(True)
    hh = ops.LinearToCL.apply(zg, w2g, b2g, c, s, dtype)
    assert relerr(from_cl(hh, c), h_ref.detach()) < TOL[dtype]
    hh.backward(to_cl(gh, c, dtype))
    torch.cuda.synchronize()
    assert relerr(zg.grad.cpu(), z.grad) < 2e-5
    assert relerr(w2g.grad.cpu(), w2.grad) < 2e-5
    assert relerr(b2g.grad.cpu(), b2.grad) < 2e-5


@pytest.mark.parametrize("b", [1, 3, 4, 7, 16])
def test_linear_layers_other_batch_sizes(b):
    """vs_linear_fwd / vs_linear_bwd at the batch sizes behind the kernels' other NB instantiations (1 | 2 | 4 | 8 | 16 rows per workgroup; test_linear_layers runs b = 2):
    the four-k-per-lane path with a gathered, an LDS-staged and an unpermuted activation, and the one-k path."""
    ops = _ops()
    dtype, dim = torch.bfloat16, 64
    for c, s in ((32, 4), (256, 3), (64, 6)):
        feat = rnd(b, c, s, s, s, seed=300 + c)
        w1, b1 = rnd(dim, c * s ** 3, seed=301, scale=0.02), rnd(dim, seed=302, scale=0.1)
        fq, w1r, b1r = q(feat, dtype).requires_grad_(True), w1.clone().requires_grad_(True), b1.clone().requires_grad_(True)
        y_ref = torch.relu(F.linear(fq.reshape(b, -1), w1r, b1r))
        gy = rnd(b, dim, seed=303)
        (y_ref * gy).sum().backward()
        x_cl = to_cl(feat, c, dtype).requires_grad_(True)
        wg, bg = w1.cuda().requires_grad_(True), b1.cuda().requires_grad_(True)
        y = ops.LinearCL.apply(x_cl, wg, bg, True)
        assert relerr(y.cpu(), y_ref.detach()) < 2e-5, (b, c, s)
        y.backward(gy.cuda())
        torch.cuda.synchronize()
        assert relerr(from_cl(x_cl.grad, c), fq.grad) < TOL[dtype], (b, c, s)
        assert relerr(wg.grad.cpu(), w1r.grad) < 2e-5 and relerr(bg.grad.cpu(), b1r.grad) < 2e-5, (b, c, s)


def test_reparam_kl_dice_bce_label_ops():
    ops = _ops()
    from oracle import ref_cpu as O
    from vae_segmentation_amd import evaluation as E
    mean, std, noise = rnd(2, 128, seed=28).requires_grad_(True), rnd(2, 128, seed=29).abs().requires_grad_(True), rnd(2, 128, seed=30)
    z_ref = mean + noise * std * 0.35
    kl_ref = O.KLloss({"mean": mean, "std": std})
    (z_ref.sum() * 0.5 + kl_ref).backward()
    mg, sg = mean.detach().cuda().requires_grad_(True), std.detach().cuda().requires_grad_(True)
    z = ops.Reparam.apply(mg, sg, noise.cuda(), 0.35)
    kl = E.KLloss({"mean": mg, "std": sg})
    (z.sum() * 0.5 + kl).backward()
    assert relerr(z.cpu(), z_ref.detach()) < 1e-6
    assert abs(kl.item() - kl_ref.item()) / abs(kl_ref.item()) < 1e-5
    assert relerr(mg.grad.cpu(), mean.grad) < 1e-5 and relerr(sg.grad.cpu(), std.grad) < 1e-5
    # KL with a zero std (log(1e-5) branch, SURVEY KAT KL-2)
    b = {"mean": torch.tensor([[1.0, -2.0, 0.5]]).cuda(), "std": torch.tensor([[0.0, 2.0, 0.5]]).cuda()}
    assert abs(E.KLloss(b).item() - 16.26289939880371) < 1e-4
    # Dice, both eps, channel slices, mean / per-sample, gradients to both operands
    s = torch.softmax(rnd(2, 2, 8, 8, 8, seed=31) * 3, 1).requires_grad_(True)
    t = torch.softmax(rnd(2, 2, 8, 8, 8, seed=32) * 3, 1).requires_grad_(True)
    for eps in (1e-6, 1e-4):
        for bot, top in ((1, 2), (0, 2)):
            for rm in (True, False):
                s.grad = t.grad = None
                ref = O.avg_dsc({"s": s, "t": t}, "s", "t", botindex=bot, topindex=top, return_mean=rm, eps=eps)
                wgt = torch.tensor([0.3, 0.7]) if not rm else torch.tensor(1.0)
                (ref * wgt).sum().backward()
                sg_, tg_ = s.detach().cuda().requires_grad_(True), t.detach().cuda().requires_grad_(True)
                got = E.avg_dsc({"s": sg_, "t": tg_}, "s", "t", botindex=bot, topindex=top, return_mean=rm, eps=eps)
                (got * wgt.cuda()).sum().backward()
                assert relerr(got.detach().cpu(), ref.detach()) < 1e-6
                assert relerr(sg_.grad.cpu(), s.grad) < 1e-5 and relerr(tg_.grad.cpu(), t.grad) < 1e-5
    # hard dice + reference KATs (tests/golden/kats.npz values)
    from tests import golden_util as G
    g = G.load("kats")
    s1 = torch.tensor([.9, .1, .8, .2, .7, .3, .6, .4]).view(1, 1, 2, 2, 2)
    t1 = torch.tensor([1., 0, 1, 0, 0, 1, 1, 0]).view(1, 1, 2, 2, 2)
    bb = {"s": torch.cat((1 - s1, s1), 1).cuda(), "t": torch.cat((1 - t1, t1), 1).cuda()}
    assert abs(E.avg_dsc(bb, "s", "t", botindex=1, topindex=2).item() - float(g["dice1"])) < 1e-6
    assert abs(E.avg_dsc(bb, "s", "t", botindex=1, topindex=2, binary=True).item() - float(g["dice2_binary"])) < 1e-6
    assert abs(E.avg_dsc(bb, "s", "t", botindex=1, topindex=2, eps=1e-4).item() - float(g["dice3_eps1e4"])) < 1e-6
    assert abs(E.dice(bb["s"], bb["t"]).item() - float(g["dice_fn"])) < 1e-6
    # hard Dice with FOUR classes (it used to raise for n_class != 2): the reference's values, ties included
    s4, t4 = O.kat_scores(1), O.kat_scores(2)
    hard = ops.hard_onehot(s4.cuda()).cpu()
    assert torch.equal(hard.argmax(1).to(torch.int8), torch.from_numpy(g["dice4_argmax_s"])) and bool((hard.sum(1) == 1).all())
    b4 = {"s": s4.cuda(), "t": t4.cuda()}
    assert abs(E.avg_dsc(b4, "s", "t", binary=True, botindex=1, topindex=4).item() - float(g["dice4_binary"])) < 1e-6
    assert abs(E.avg_dsc(b4, "s", "t", binary=True, botindex=0, topindex=4).item() - float(g["dice4_binary_all"])) < 1e-6
    assert np.allclose(E.avg_dsc(b4, "s", "t", binary=True, botindex=1, topindex=4, return_mean=False).cpu().numpy(), g["dice4_binary_nomean"], atol=1e-6)
    one = torch.rand(2, 1, 3, 4, 5)
    assert bool((ops.hard_onehot(one.cuda()) == 1).all())                       # a single channel: argmax is 0 everywhere
    assert np.array_equal(E.binarize(torch.tensor([.49, .5, .81]).cuda()).cpu().numpy(), g["bin"])
    assert np.array_equal(E.confident_binarize(torch.tensor([.1, .2, .5, .8, .81]).cuda()).cpu().numpy(), g["cbin"])
    # BCE
    p = torch.sigmoid(rnd(2, 1, 4, 4, 4, seed=33)).requires_grad_(True)
    tt = (rnd(2, 1, 4, 4, 4, seed=34) > 0).float()
    ref = O.avg_ce({"a": p, "b": tt}, "a", "b")
    ref.backward()
    pg = p.detach().cuda().requires_grad_(True)
    got = E.avg_ce({"a": pg, "b": tt.cuda()}, "a", "b")
    got.backward()
    assert abs(got.item() - ref.item()) < 1e-6 and relerr(pg.grad.cpu(), p.grad) < 1e-5
    # one-hot
    lab = (rnd(2, 1, 4, 4, 4, seed=35) > 0.5).float()
    assert torch.equal(ops.onehot(lab.cuda(), 2).cpu(), O.one_hot(lab, 2))


def test_cpu_tensor_is_rejected_loudly():
    import joint_model
    seg = joint_model.Segmentation(1, 2, norm_type=1)
    with pytest.raises(RuntimeError):
        seg({"x": torch.zeros(1, 1, 16, 16, 16)}, "x", "y")


@pytest.mark.parametrize("dtype", DT)
def test_dropout_forward_backward_share_the_mask(dtype):
    """F.dropout(training=True) semantics with a counter-based mask: kept fraction ~ 1-p, kept values scaled by 1/(1-p),
    the backward applies the SAME mask, a different seed gives a different mask."""
    ops = _ops()
    torch.manual_seed(1234)                       # the mask seed derives from torch's seed: fixed here, so the 2.8-sigma bound below cannot flake
    x = (rnd(2, 16, 8, 8, 8, seed=40).abs() + 0.5)
    x_cl = to_cl(x, 16, dtype).requires_grad_(True)
    p, seed = 0.3, ops.next_dropout_seed()
    y = ops.Dropout.apply(x_cl, p, seed)
    keep = (y != 0)
    frac = keep.float().mean().item()
    assert abs(frac - (1 - p)) < 0.015             # 16384 draws: sigma = 0.0036
    ratio = (y.float()[keep] / x_cl.detach().float()[keep])
    assert float((ratio - 1 / (1 - p)).abs().max()) < {torch.bfloat16: 2e-2, torch.float16: 2e-3, torch.float32: 1e-5}[dtype]
    assert torch.equal(ops.dropout_mask(x_cl.numel(), p, seed).view_as(y) != 0, keep)      # the exported mask is the applied one
    g = to_cl(rnd(2, 16, 8, 8, 8, seed=41).abs() + 0.5, 16, dtype)
    y.backward(g)
    assert torch.equal(x_cl.grad != 0, keep)
    y2 = ops.Dropout.apply(x_cl.detach(), p, ops.next_dropout_seed())
    assert not torch.equal(y2 != 0, keep)


def test_segmentation_and_vae_with_dropout_run_and_stay_normalised():
    """dropout > 0 at every site of the reference (joint_model.py:256-264, 379-388): probabilities still sum to 1,
    gradients are finite, and two calls differ (fresh masks)."""
    import joint_model as M
    from oracle import ref_cpu as O
    seg = O.deterministic_fill_(M.Segmentation(1, 2, norm_type=1), seed=0).cuda()
    img = O.synthetic_image(1, 32, 2).cuda()
    a = seg({"x": img}, "x", "p", dropout=0.2)["p"]
    b = seg({"x": img}, "x", "p", dropout=0.2)["p"]
    assert float((a.sum(1) - 1).abs().max()) < 1e-5
    assert float((a - b).abs().max()) > 1e-4
    a[:, 1].mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in seg.parameters())
    vae = O.deterministic_fill_(M.VAE(2, 2, norm_type=1, dim=128, spatial=64), seed=0).cuda()
    r, m, s = vae(O.one_hot(O.synthetic_label(1, 64, 3)).cuda(), dropout=0.1)
    assert float((r.sum(1) - 1).abs().max()) < 1e-5 and torch.isfinite(r).all()


GROUP_LAYERS = [  # (kind, N, Cin, Cout, D, H, W) — one backward pass over all of them: every (CB, KIND) bucket of the grouped weight-gradient
    # launch gets several layers of different sizes, more than G3_GROUP_MAX (16) in the largest bucket
    ("k3", 2, 16, 16, 12, 12, 20), ("k3", 2, 32, 32, 6, 6, 6), ("k3", 2, 64, 32, 5, 4, 7), ("k3", 2, 32, 64, 3, 3, 3),
    ("k3", 2, 128, 128, 3, 3, 3), ("k3", 2, 16, 8, 9, 16, 33), ("k3", 2, 8, 8, 16, 16, 32), ("k3", 2, 8, 16, 7, 9, 18),
    ("k3", 2, 1, 8, 8, 8, 16), ("k3", 2, 16, 32, 8, 8, 8), ("k3", 2, 32, 16, 8, 8, 8), ("k3", 2, 64, 64, 4, 4, 4),
    ("k3", 2, 16, 16, 4, 4, 4), ("k3", 2, 16, 16, 5, 5, 5), ("k3", 2, 16, 16, 6, 6, 6), ("k3", 2, 32, 32, 4, 4, 4),
    ("k3", 2, 32, 32, 5, 5, 5), ("k3", 2, 64, 64, 3, 3, 3), ("k3", 2, 64, 64, 5, 5, 5), ("k3", 2, 128, 64, 3, 3, 3),
    ("k3", 2, 256, 256, 2, 2, 2), ("k3", 2, 16, 16, 20, 24, 40),
    ("k3", 2, 8, 8, 7, 9, 21), ("k3", 2, 8, 2, 6, 5, 17), ("k3", 1, 3, 8, 5, 6, 33), ("k3", 2, 32, 8, 5, 7, 19),      # 8-channel layers at ragged sizes: the M-packed form's shifted rows at every border
    ("k2", 2, 8, 8, 8, 8, 16), ("k2", 2, 16, 16, 4, 6, 10), ("k2", 2, 64, 64, 2, 2, 6), ("k2", 2, 32, 32, 6, 6, 6),
    ("t2", 2, 16, 16, 4, 4, 8), ("t2", 2, 32, 32, 3, 5, 7), ("t2", 2, 128, 128, 3, 3, 3), ("t2", 2, 8, 8, 4, 4, 4),
]


def _group_layers_backward(ops, dtype):
    """-> (list of (weight grad, bias grad or None) on the GPU, list of CPU references)"""
    outs, refs, total = [], [], None
    for i, (kind, n, cin, cout, d, h, w) in enumerate(GROUP_LAYERS):
        x = rnd(n, cin, d, h, w, seed=10 + i)
        xq = q(x, dtype)
        a = in_relu(xq)
        x_cl = to_cl(x, ops.cpad(cin), dtype)
        xs = ops.instnorm_stats(x_cl)
        if kind == "k3":
            wt = q(rnd(cout, cin, 3, 3, 3, seed=40 + i, scale=(3.0 / (27 * cin)) ** 0.5), dtype).requires_grad_(True)
            y_ref = F.conv3d(a, wt, None, padding=1)
            gy = rnd(*y_ref.shape, seed=70 + i)
            w_gpu, b_gpu = wt.detach().cuda().requires_grad_(True), None
            y, _ = ops.ConvK3.apply(x_cl, xs, w_gpu, None)
        elif kind == "k2":
            wt = q(rnd(cout, cin, 2, 2, 2, seed=40 + i, scale=(3.0 / (8 * cin)) ** 0.5), dtype).requires_grad_(True)
            bt = rnd(cout, seed=90 + i).requires_grad_(True)
            y_ref = F.conv3d(a, wt, bt, stride=2)
            gy = rnd(*y_ref.shape, seed=70 + i)
            w_gpu, b_gpu = wt.detach().cuda().requires_grad_(True), bt.detach().cuda().requires_grad_(True)
            y = ops.ConvK2S2.apply(x_cl, xs, w_gpu, b_gpu)
        else:
            wt = q(rnd(cin, cout, 2, 2, 2, seed=40 + i, scale=(3.0 / (8 * cin)) ** 0.5), dtype).requires_grad_(True)
            bt = rnd(cout, seed=90 + i).requires_grad_(True)
            y_ref = F.conv_transpose3d(a, wt, bt, stride=2)
            gy = rnd(*y_ref.shape, seed=70 + i)
            w_gpu, b_gpu = wt.detach().cuda().requires_grad_(True), bt.detach().cuda().requires_grad_(True)
            y = ops.ConvT2S2.apply(x_cl, xs, w_gpu, b_gpu)
        gq = q(gy, dtype)
        (y_ref * gq).sum().backward()
        refs.append((wt.grad.clone(), None if kind == "k3" else bt.grad.clone()))
        term = (y.float() * to_cl(gy, ops.cpad(cout), dtype).float()).sum()
        total = term if total is None else total + term
        outs.append((w_gpu, b_gpu))
    total.backward()                              # ONE pass: the engine callback at its end issues the grouped launches
    torch.cuda.synchronize()
    return [(wg.grad.clone(), None if bg is None else bg.grad.clone()) for wg, bg in outs], refs


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("mpack,uber,swap,big,fold", [("1", "1", "1", "0", "1"), ("0", "1", "1", "0", "1"), ("1", "0", "1", "0", "1"), ("1", "1", "0", "0", "1"),
                                                      ("1", "1", "1", "1", "1"), ("1", "1", "0", "1", "1"), ("0", "1", "1", "1", "1"), ("1", "1", "1", "0", "0")])
def test_grouped_weight_gradients_many_layers(dtype, mpack, uber, swap, big, fold, monkeypatch):
    """vs_conv_wgrad_multi (main_source.py:660: the gradients the optimiser step reads): 34 layers of all conv kinds deferred to the end of ONE
    backward pass and issued as grouped launches — each against F.conv3d / F.conv_transpose3d autograd on the CPU, against the
    per-layer launches (vs_conv_wgrad), and bitwise reproducible.  mpack: the M-packed form of the layers with 8 stored output channels
    (csrc/wgrad.hip g3b_body) on / off; uber: all buckets in one grid / one grid per bucket; swap: operand exchange on / off; big: big-tile kernel."""
    ops = _ops()
    # uber 1: every bucket in one grid (g3b_uber_kernel, the default); 0: one grid per bucket.  swap 1 (default): operands of the lazy-input 3x3x3 layers exchanged
    # (csrc/wgrad.hip multi_plan); 0: as submitted.  big: the 8 x 8 x 16-tile kernel (g3c_body) takes every layer it supports — by default only tensors of
    # >= 400 k voxels (none of this list) get it.  Switched through the library's one configuration entry point (vs_set_config), restored afterwards.
    # fold 1 (default): a transposed conv's bias gradient is summed by its own weight-gradient workgroups (their Q operand is that tensor); 0: a pass of its own.
    with ops.config(wgrad_mpack=int(mpack), wgrad_uber=int(uber), wgrad_swap=int(swap), wgrad_big_min_voxels=1 if big == "1" else 1000000000,
                    wgrad_bias_fold=int(fold), wgrad_xcd=2 if fold == "1" else 1):
        assert ops._GROUP["enabled"]
        got, refs = _group_layers_backward(ops, dtype)
        tol = TOL[dtype] * 4
        for (gw, gb), (rw, rb), case in zip(got, refs, GROUP_LAYERS):
            assert relerr(gw.cpu(), rw) < tol, case
            if rb is not None:
                assert relerr(gb.cpu(), rb) < tol, case
        again, _ = _group_layers_backward(ops, dtype)
        for (gw, gb), (aw, ab) in zip(got, again):
            assert torch.equal(gw, aw)
            if gb is not None and dtype != torch.float32:      # fp32 mode keeps vs_bias_grad (float atomics)
                assert torch.equal(gb, ab)
        ops.set_wgrad_grouping(False)
        try:
            single, _ = _group_layers_backward(ops, dtype)
        finally:
            ops.set_wgrad_grouping(True)
        for (gw, gb), (sw, sb), case in zip(got, single, GROUP_LAYERS):
            assert relerr(gw, sw) < 2e-6, case
            if gb is not None:
                assert relerr(gb, sb) < 1e-5, case


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_transposed_conv_bias_gradient_folded_into_its_weight_gradient(dtype):
    """wgrad.hip g3b_body, stride-2 kind (round 6): db of nn.ConvTranspose3d (joint_model.py:118) = the per-channel sum of the fine gradient its weight-gradient
    workgroups stage as Q — summed there.  With FEW workgroups per layer (wgrad_group_wgs) every workgroup walks several tiles, ragged ones included; against the
    CPU autograd and against the separate pass (wgrad_bias_fold=0)."""
    ops = _ops()
    res = {}
    for fold in (1, 0):
        with ops.config(wgrad_bias_fold=fold, wgrad_group_wgs=6):
            outs, refs = [], []
            total = None
            for i, (n, c, d, h, w) in enumerate([(2, 16, 9, 7, 19), (1, 32, 6, 10, 17), (2, 8, 5, 9, 33), (2, 64, 3, 4, 5)]):
                x = rnd(n, c, d, h, w, seed=200 + i)
                a = in_relu(q(x, dtype))
                x_cl = to_cl(x, ops.cpad(c), dtype)
                xs = ops.instnorm_stats(x_cl)
                wt = q(rnd(c, c, 2, 2, 2, seed=210 + i, scale=(3.0 / (8 * c)) ** 0.5), dtype).requires_grad_(True)
                bt = rnd(c, seed=220 + i).requires_grad_(True)
                y_ref = F.conv_transpose3d(a, wt, bt, stride=2)
                gy = rnd(*y_ref.shape, seed=230 + i)
                (y_ref * q(gy, dtype)).sum().backward()
                refs.append((wt.grad.clone(), bt.grad.clone()))
                w_gpu, b_gpu = wt.detach().cuda().requires_grad_(True), bt.detach().cuda().requires_grad_(True)
                y = ops.ConvT2S2.apply(x_cl, xs, w_gpu, b_gpu)
                term = (y.float() * to_cl(gy, ops.cpad(c), dtype).float()).sum()
                total = term if total is None else total + term
                outs.append((w_gpu, b_gpu))
            total.backward()
            torch.cuda.synchronize()
            res[fold] = [(wg.grad.clone(), bg.grad.clone()) for wg, bg in outs]
            for (gw, gb), (rw, rb) in zip(res[fold], refs):
                assert relerr(gw.cpu(), rw) < TOL[dtype] * 4 and relerr(gb.cpu(), rb) < TOL[dtype] * 4, (fold, relerr(gb.cpu(), rb))
    for (fw, fb), (sw, sb) in zip(res[1], res[0]):
        assert torch.equal(fw, sw)                               # the weight gradient itself is untouched by the fold
        assert relerr(fb, sb) < 1e-5


def test_dice_loss_sum_label_target_equals_materialised_one_hot():
    """ops.LabelTarget (the one-hot of main_source.py:449-451 evaluated inside the loss kernels) against the same loss on the materialised
    one-hot: loss, terms and the gradients of the source and of the other target, bit for bit; also through the unfused spelling."""
    ops = _ops()
    b, c, side = 2, 2, 16
    g = torch.Generator().manual_seed(11)
    src = torch.softmax(torch.randn(b, c, side, side, side, generator=g), 1)
    other = torch.softmax(torch.randn(b, c, side, side, side, generator=g), 1)
    label = (torch.rand(b, 1, side, side, side, generator=g) > 0.6).float()
    res = {}
    for mode in ("label", "onehot", "label_unfused"):
        ops.FUSED_LOSS[0] = mode != "label_unfused"
        try:
            s_g, o_g = src.cuda().requires_grad_(True), other.cuda().requires_grad_(True)
            gt = ops.onehot(label.cuda(), c) if mode == "onehot" else ops.LabelTarget(label.cuda())
            final, terms = ops.dice_loss_sum(s_g, [(o_g, 0.1), (gt, 1.0)], botindex=1, topindex=2, eps=1e-4)
            final.backward()
            res[mode] = (final.detach().cpu(), torch.stack([t.detach().cpu() for t in terms]), s_g.grad.cpu(), o_g.grad.cpu())
        finally:
            ops.FUSED_LOSS[0] = True
    for a, b_ in zip(res["label"], res["onehot"]):
        assert torch.equal(a, b_)
    for a, b_ in zip(res["label_unfused"], res["onehot"]):
        assert relerr(a, b_) < 1e-5
    with pytest.raises(ValueError):                          # a label volume of the wrong size is refused, not mis-indexed
        ops.dice_loss_sum(src.cuda(), [(ops.LabelTarget(label[:, :, :8].cuda()), 1.0)], botindex=1, topindex=2, eps=1e-4)


@pytest.mark.parametrize("k", [1, 2, 3])
def test_dice_loss_sum_matches_reference_spelling(k):
    """ops.dice_loss_sum (one launch each way) against the reference's spelling — 1 - avg_dsc per term, torch scalar arithmetic
    (main_source.py:469-471) — on the CPU, and against the unfused GPU path: value, terms, gradients of source and targets."""
    ops = _ops()
    b, c, side = 2, 2, 12
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(b, c, side, side, side, generator=g)
    src = torch.softmax(logits, 1)
    tgts = [torch.softmax(torch.randn(b, c, side, side, side, generator=g), 1) for _ in range(k)]
    ws = [0.1, 1.0, 0.37][:k]
    eps = 1e-4

    def ref_dice(s, t):
        s1, t1 = s[:, 1:2].reshape(b, 1, -1), t[:, 1:2].reshape(b, 1, -1)
        return (2 * (s1 * t1).sum(2) / (s1.sum(2) + t1.sum(2) + eps)).mean()

    s_ref = src.clone().requires_grad_(True)
    t_ref = [t.clone().requires_grad_(True) for t in tgts]
    terms_ref = [1 - ref_dice(s_ref, t) for t in t_ref]
    final_ref = sum(w * tr for w, tr in zip(ws, terms_ref))
    final_ref.backward()

    outs = {}
    for fused in (True, False):
        ops.FUSED_LOSS[0] = fused
        try:
            s_g = src.cuda().requires_grad_(True)
            t_g = [t.cuda().requires_grad_(True) for t in tgts]
            ops.stats_arena_begin(s_g.device)
            final, terms = ops.dice_loss_sum(s_g, list(zip(t_g, ws)), botindex=1, topindex=2, eps=eps)
            final.backward()
            torch.cuda.synchronize()
            outs[fused] = (final.item(), [float(t) for t in terms], s_g.grad.cpu(), [t.grad.cpu() for t in t_g])
        finally:
            ops.FUSED_LOSS[0] = True
    for fused, (fv, tv, gs, gts) in outs.items():
        assert abs(fv - final_ref.item()) < 2e-6, fused
        for a, r in zip(tv, terms_ref):
            assert abs(a - r.item()) < 2e-6
        assert relerr(gs, s_ref.grad) < 2e-5
        for a, r in zip(gts, t_ref):
            assert relerr(a, r.grad) < 2e-5
    assert abs(outs[True][0] - outs[False][0]) < 1e-6


@pytest.mark.parametrize("dtype", DT)
def test_weight_used_several_times_in_one_backward(dtype):
    """A network applied more than once per forward (the VAE inside Embed runs three times, joint_model.py:469-500) uses each weight
    several times in one backward pass: the deferred grouped launches must SUM the uses (descriptors sharing their destination are
    reduced together) — all three conv kinds, live biases included, against autograd on the CPU."""
    ops = _ops()
    n, c = 2, 16
    xs_ = [rnd(n, c, 8, 8, 16, seed=50), rnd(n, c, 4, 6, 8, seed=51), rnd(n, c, 12, 4, 8, seed=52)]         # three uses, three sizes
    w3 = q(rnd(c, c, 3, 3, 3, seed=53, scale=0.05), dtype)
    w2, b2 = q(rnd(c, c, 2, 2, 2, seed=54, scale=0.1), dtype), rnd(c, seed=55, scale=0.1)
    wt, bt = q(rnd(c, c, 2, 2, 2, seed=56, scale=0.1), dtype), rnd(c, seed=57, scale=0.1)
    ref = [t.clone().requires_grad_(True) for t in (w3, w2, b2, wt, bt)]
    gpu = [t.clone().cuda().requires_grad_(True) for t in (w3, w2, b2, wt, bt)]
    total_ref, total = 0.0, None
    ops.stats_arena_begin(torch.device("cuda", 0))
    for i, x in enumerate(xs_):
        xq = q(x, dtype)
        a = in_relu(xq)
        y3 = F.conv3d(a, ref[0], None, padding=1)
        y2 = F.conv3d(a, ref[1], ref[2], stride=2)
        yt = F.conv_transpose3d(a, ref[3], ref[4], stride=2)
        g3, g2, gt_ = rnd(*y3.shape, seed=60 + i), rnd(*y2.shape, seed=70 + i), rnd(*yt.shape, seed=80 + i)
        total_ref = total_ref + (y3 * q(g3, dtype)).sum() + (y2 * q(g2, dtype)).sum() + (yt * q(gt_, dtype)).sum()
        x_cl = to_cl(x, c, dtype)
        st = ops.instnorm_stats(x_cl)
        o3, _ = ops.ConvK3.apply(x_cl, st, gpu[0], None)
        o2 = ops.ConvK2S2.apply(x_cl, st, gpu[1], gpu[2])
        ot = ops.ConvT2S2.apply(x_cl, st, gpu[3], gpu[4])
        term = (o3.float() * to_cl(g3, c, dtype).float()).sum() + (o2.float() * to_cl(g2, c, dtype).float()).sum() + (ot.float() * to_cl(gt_, c, dtype).float()).sum()
        total = term if total is None else total + term
    total_ref.backward()
    total.backward()
    torch.cuda.synchronize()
    tol = TOL[dtype] * 4
    for name, r, gq_ in zip(("k3 weight", "k2s2 weight", "k2s2 bias", "convT weight", "convT bias"), ref, gpu):
        assert relerr(gq_.grad.cpu(), r.grad) < tol, name


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_weight_gradient_fused_into_backward_data_matches_the_grouped_launch(dtype):
    """ConvK3.backward hands the weight gradient of an 8 -> 8 layer to its backward-data launch (ops._wgrad_fusable, csrc/igemm_k3tw.h) when the weight is used
    once in the pass; a weight used twice keeps the grouped launch (its uses are summed there).  Same gradients either way."""
    ops = _ops()
    n, c = 2, 8
    x = rnd(n, c, 12, 16, 40, seed=90)
    w1, w2 = q(rnd(c, c, 3, 3, 3, seed=91, scale=0.08), dtype), q(rnd(c, c, 3, 3, 3, seed=92, scale=0.08), dtype)
    gsum = to_cl(rnd(n, c, 12, 16, 40, seed=93), c, dtype).float()

    def run(fuse, reuse):
        ops.FUSE_WGRAD = fuse
        taken = []
        orig = ops._group_submit_slabs
        ops._group_submit_slabs = lambda *a, **k: (taken.append(1), orig(*a, **k))[1]
        try:
            ws = [t.clone().cuda().requires_grad_(True) for t in (w1, w2)]
            ops.stats_arena_begin(torch.device("cuda", 0))
            x_cl = to_cl(x, c, dtype).requires_grad_(True)
            st = ops.instnorm_stats(x_cl)
            y1, s1 = ops.ConvK3.apply(x_cl, st, ws[0], None)
            y2, s2 = ops.ConvK3.apply(y1, s1, ws[1], None)
            if reuse:
                y2, s2 = ops.ConvK3.apply(y2, s2, ws[1], None)
            (y2.float() * gsum).sum().backward()
            torch.cuda.synchronize()
            return [t.grad.clone() for t in ws] + [x_cl.grad.float().clone()], len(taken)
        finally:
            ops._group_submit_slabs = orig
            ops.FUSE_WGRAD = True

    for reuse in (False, True):
        ref, n_ref = run(False, reuse)
        got, n_got = run(True, reuse)
        assert n_ref == 0
        assert n_got == (1 if reuse else 2)          # the weight applied twice stays with the grouped launch
        for a, b in zip(got, ref):
            assert bool(torch.isfinite(a).all())
            assert relerr(a.cpu(), b.cpu()) < 2e-5
    # a second pass right away: the use counts start over when a backward pass ends
    _, n_again = run(True, False)
    assert n_again == 2
    # forward(W), an UNRELATED backward pass ends, forward(W) again, then ONE backward through both uses — the forward-time use count reads 1 with two
    # live uses; the first use's gradient rides in its backward-data launch (slabs), the second must get a destination of its own and autograd must add the two
    def interleaved(fuse):
        ops.FUSE_WGRAD = fuse
        try:
            wt = w1.clone().cuda().requires_grad_(True)
            other = w2.clone().cuda().requires_grad_(True)
            ops.stats_arena_begin(torch.device("cuda", 0))
            x_cl = to_cl(x, c, dtype)
            st = ops.instnorm_stats(x_cl)
            ya, _ = ops.ConvK3.apply(x_cl, st, wt, None)                       # first forward through W
            yo, _ = ops.ConvK3.apply(x_cl, st, other, None)
            (yo.float() * gsum).sum().backward()                               # an unrelated backward pass ends in between
            yb, _ = ops.ConvK3.apply(x_cl, st, wt, None)                       # second forward through W
            ((ya.float() + 2.0 * yb.float()) * gsum).sum().backward()
            torch.cuda.synchronize()
            return wt.grad.clone()
        finally:
            ops.FUSE_WGRAD = True
    g_ref, g_got = interleaved(False), interleaved(True)
    assert bool(torch.isfinite(g_got).all()) and relerr(g_got.cpu(), g_ref.cpu()) < 2e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("frozen", [False, True])
@pytest.mark.parametrize("cl", [True, False])
def test_out_block_backward_one_launch_matches_the_three_launch_path(dtype, frozen, cl):
    """ConvK3SoftmaxCL.backward through vs_conv_k3_softmax2_bwd_data (ops._out_block_bwd_fused) against its softmax-backward + backward-data + grouped-gradient
    form: both gradient parts of the probabilities (planar and channels-last), logit dropout, trainable and frozen (the VAE inside Joint) out_block."""
    ops = _ops()
    n, d, h, w = 2, 12, 16, 40
    x = rnd(n, 8, d, h, w, seed=21)
    wt, b = q(rnd(2, 8, 3, 3, 3, seed=22, scale=0.3), dtype), rnd(2, seed=23, scale=0.2)
    gp = rnd(n, 2, d, h, w, seed=24).cuda()
    gc = to_cl(rnd(n, 8, d, h, w, seed=25), 8, dtype)

    def run(fuse):
        ops.FUSE_SOFTMAX_BWD = fuse
        taken = []
        orig = ops._out_block_bwd_fused
        ops._out_block_bwd_fused = lambda *a, **k: (lambda r: (taken.append(r is not None), r)[1])(orig(*a, **k))
        try:
            wg, bg = wt.clone().cuda().requires_grad_(not frozen), b.clone().cuda().requires_grad_(not frozen)
            ops.stats_arena_begin(torch.device("cuda", 0))
            x_cl = to_cl(x, 8, dtype).requires_grad_(True)
            xs = ops.instnorm_stats(x_cl.detach())
            if cl:                               # the prediction that another network reads (Joint: Segmentation -> VAE)
                prob, prob_cl = ops.ConvK3SoftmaxCL.apply(x_cl, xs, wg, bg, 0.2, 99)
                ((prob * gp).sum() + (prob_cl.float() * gc.float()).sum()).backward()
            else:                                # the VAE's own out_block
                prob = ops.ConvK3Softmax.apply(x_cl, xs, wg, bg, 0.2, 99)
                (prob * gp).sum().backward()
            torch.cuda.synchronize()
            return [x_cl.grad.float().clone()] + ([] if frozen else [wg.grad.clone(), bg.grad.clone()]), taken
        finally:
            ops._out_block_bwd_fused = orig
            ops.FUSE_SOFTMAX_BWD = True

    ref, t_ref = run(False)
    got, t_got = run(True)
    assert t_ref == [False] and t_got == [True]
    for a, r in zip(got, ref):
        assert bool(torch.isfinite(a).all())
        assert relerr(a.cpu(), r.cpu()) < 2e-5
