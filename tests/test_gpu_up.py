"""GPU parity of the composed Up block head (csrc/igemm_k4.h, ops.UpConvK3): ConvTranspose3d(C, C, 2, stride 2) -> Conv3d(C, Co, 3, padding 1)
(/root/reference/joint_model.py:116-120 + 40) as ONE operator on the coarse grid, against F.conv_transpose3d + F.conv3d autograd in fp32 on
the CPU, at every (channels, side) pair the 96^3 / 128^3 / 160^3 networks contain plus ragged and multi-tile cases.  16-bit storage only
(the fp32 parity mode keeps the two-launch form).  Tolerances: those of tests/test_gpu_ops.py (bf16 1.5e-2, fp16 2e-3; x4 behind the lazy input)."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_ops import TOL, from_cl, in_relu, q, relerr, rnd, to_cl

pytestmark = pytest.mark.gpu

DT = [torch.bfloat16, torch.float16]
# (N, C, Co, D, H, W) of the COARSE input
UP_CASES = [
    (2, 16, 8, 48, 48, 48), (2, 32, 16, 24, 24, 24), (2, 64, 32, 12, 12, 12), (2, 128, 64, 6, 6, 6), (2, 256, 128, 3, 3, 3),     # configs[1]: up5 .. up1
    (1, 16, 8, 64, 64, 64), (1, 32, 16, 32, 32, 32), (1, 256, 128, 4, 4, 4),                                                      # configs[3] (128^3)
    (2, 32, 16, 40, 40, 40), (2, 128, 64, 10, 10, 10), (2, 256, 128, 5, 5, 5),                                                    # configs[4] (160^3): odd side 5
    (1, 16, 8, 5, 6, 19), (3, 32, 16, 3, 5, 17), (1, 64, 32, 2, 9, 4), (2, 16, 8, 1, 1, 1), (1, 32, 32, 4, 4, 16), (1, 64, 16, 6, 7, 5),   # ragged; Co != C / 2
]


def _ops():
    from vae_segmentation_amd import ops
    return ops


from tests.test_gpu_layers import _last_call          # noqa: E402  (one-slot memo of the CPU reference, shared by the two lib_mode runs of a case)


@_last_call
def _up_ref(case, dtype, lazy):
    n, c, co, d, h, w = case
    x = rnd(n, c, d, h, w, seed=21)
    w2 = rnd(c, c, 2, 2, 2, seed=22, scale=(3.0 / c) ** 0.5)
    b2 = rnd(c, seed=23, scale=0.3)
    w3 = rnd(co, c, 3, 3, 3, seed=24, scale=(3.0 / (27 * c)) ** 0.5)
    gy = rnd(n, co, 2 * d, 2 * h, 2 * w, seed=25)
    xq, gq = q(x, dtype).requires_grad_(True), q(gy, dtype)
    w2q, w3q = q(w2, dtype), q(w3, dtype)
    a = in_relu(xq) if lazy else xq
    y_ref = F.conv3d(F.conv_transpose3d(a, w2q, b2, stride=2), w3q, None, padding=1)
    (y_ref * gq).sum().backward()
    return x, b2, gy, xq, w2q, w3q, y_ref.detach()


@pytest.mark.parametrize("lazy", [True, False])
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", UP_CASES)
def test_up_composed_vs_cpu_autograd(case, dtype, lazy):
    ops = _ops()
    n, c, co, d, h, w = case
    if lazy and d * h * w == 1:
        pytest.skip("InstanceNorm needs more than one voxel")
    big = n * d * h * w * c > 3_000_000
    if big and not lazy:
        pytest.skip("the large shapes run once, with the lazy input the networks use")
    x, b2, gy, xq, w2q, w3q, y_ref = _up_ref(case, dtype, lazy)

    x_cl = to_cl(x, c, dtype).requires_grad_(True)
    xs = ops.instnorm_stats(x_cl.detach()) if lazy else None
    wt, bt, wc = w2q.cuda(), b2.cuda(), w3q.cuda()            # frozen (requires_grad False): the composed path
    ops.stats_arena_begin(x_cl.device)
    y, ys = ops.UpConvK3.apply(x_cl, xs, wt, bt, wc)
    y.backward(to_cl(gy, co, dtype))
    torch.cuda.synchronize()
    tol = TOL[dtype]
    # the statistics are those of the STORED values (the consumer normalises what it loads): compared with sums over the kernel's own output —
    # against the reference's they carry the systematic part of the weight rounding times sqrt(voxels)
    yo = from_cl(y, co).double()
    st = ops.stats_total(ys).cpu()[:, :co]
    own_sum, own_sq, own_abs = yo.sum((2, 3, 4)), (yo * yo).sum((2, 3, 4)), yo.abs().sum((2, 3, 4))
    errs = {"y": relerr(from_cl(y, co), y_ref.detach()),
            "stat_sum": float(((st[..., 0] - own_sum).abs() / own_abs).max()),
            "stat_sq": float(((st[..., 1] - own_sq).abs() / own_sq).max()),
            "gx": relerr(from_cl(x_cl.grad, c), xq.grad)}
    lims = {"y": 2 * tol, "stat_sum": 1e-5, "stat_sq": 1e-5, "gx": 4 * tol}
    bad = {k: (errs[k], lims[k]) for k in errs if not errs[k] < lims[k]}
    print("\nup %s %s lazy=%s: %s" % (case, dtype, lazy, ", ".join("%s %.2e (<%.1e)" % (k, errs[k], lims[k]) for k in errs)))
    assert not bad, bad


TRAIN_CASES = [(2, 16, 8, 48, 48, 48), (1, 16, 8, 5, 6, 19), (2, 32, 16, 12, 12, 12), (1, 64, 32, 4, 6, 5), (2, 16, 8, 1, 2, 1), (1, 128, 64, 3, 3, 3)]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", TRAIN_CASES)
def test_up_composed_weight_gradients_vs_cpu_autograd(case, dtype):
    """Trainable weights: dWeff from the VS_CONV_UP weight-gradient launch + the boundary sums (vs_up_faces) through the parameter-space chain
    rule (vs_up_chain) against autograd's gradients of ConvTranspose3d.weight / .bias and Conv3d.weight."""
    ops = _ops()
    n, c, co, d, h, w = case
    x = rnd(n, c, d, h, w, seed=41)
    w2 = rnd(c, c, 2, 2, 2, seed=42, scale=(3.0 / c) ** 0.5)
    b2 = rnd(c, seed=43, scale=0.3)
    w3 = rnd(co, c, 3, 3, 3, seed=44, scale=(3.0 / (27 * c)) ** 0.5)
    # an upstream gradient with zero mean per (n, channel), as the InstanceNorm backward that always follows produces (the bias gradient uses it)
    gy = rnd(n, co, 2 * d, 2 * h, 2 * w, seed=45)
    gy = q(gy, dtype)
    gy = gy - gy.mean((2, 3, 4), keepdim=True)
    xq = q(x, dtype).requires_grad_(True)
    w2q, w3q, b2q = q(w2, dtype).requires_grad_(True), q(w3, dtype).requires_grad_(True), b2.clone().requires_grad_(True)
    y_ref = F.conv3d(F.conv_transpose3d(in_relu(xq), w2q, b2q, stride=2), w3q, None, padding=1)
    gq = q(gy, dtype)                     # what the kernel reads; its channel sums are zero only up to the rounding (the reference sees the same values)
    (y_ref * gq).sum().backward()

    x_cl = to_cl(x, c, dtype).requires_grad_(True)
    xs = ops.instnorm_stats(x_cl.detach())
    wt, bt, wc = w2q.detach().cuda().requires_grad_(True), b2.cuda().requires_grad_(True), w3q.detach().cuda().requires_grad_(True)
    ops.stats_arena_begin(x_cl.device)
    y, ys = ops.UpConvK3.apply(x_cl, xs, wt, bt, wc)
    y.backward(to_cl(gy, co, dtype))
    torch.cuda.synchronize()
    tol = TOL[dtype]
    # the kernel takes the volume sum of gy as exactly zero (what an InstanceNorm backward gives); the reference's db2 / dW3 carry the rounding
    # residue of gq's channel sums times the weights: compare against the reference minus that residue
    resid = gq.sum((0, 2, 3, 4))                                     # [co]
    gb_ref = b2q.grad - (w3q.detach().sum((2, 3, 4)) * resid[:, None]).sum(0)
    gw3_ref = w3q.grad - resid[:, None, None, None, None] * b2.view(1, -1, 1, 1, 1)
    errs = {"gx": relerr(from_cl(x_cl.grad, c), xq.grad), "gw2": relerr(wt.grad.cpu(), w2q.grad), "gb2": relerr(bt.grad.cpu(), gb_ref),
            "gw3": relerr(wc.grad.cpu(), gw3_ref)}
    lims = {"gx": 4 * tol, "gw2": 4 * tol, "gb2": 4 * tol, "gw3": 4 * tol}
    bad = {k: (errs[k], lims[k]) for k in errs if not errs[k] < lims[k]}
    print("\nup train %s %s: %s" % (case, dtype, ", ".join("%s %.2e (<%.1e)" % (k, errs[k], lims[k]) for k in errs)))
    assert not bad, bad


@pytest.mark.parametrize("dtype", DT)
def test_up_block_module_uses_composed_path_when_frozen(dtype):
    """`Up` (joint_model.py:114-124) with frozen weights runs the composed operator and matches the two-launch form of the same module."""
    import joint_model as M
    from oracle import ref_cpu as O
    ops = _ops()
    up = O.deterministic_fill_(M.Up(32, 16, norm_type=1), seed=3).cuda()
    M.set_kernel_dtype(up, dtype)
    for p_ in up.parameters():
        p_.requires_grad = False
    x = rnd(2, 32, 12, 12, 12, seed=31).cuda().requires_grad_(True)
    outs = []
    for fuse in (True, False):
        ops.FUSE_UP, keep = fuse, ops.FUSE_UP_MIN_VOXELS
        ops.FUSE_UP_MIN_VOXELS = 0
        try:
            xx = x.detach().clone().requires_grad_(True)
            ops.stats_arena_begin(xx.device)
            y = up(xx)
            (y * y).sum().backward()
            outs.append((y.detach().float().cpu(), xx.grad.float().cpu()))
        finally:
            ops.FUSE_UP, ops.FUSE_UP_MIN_VOXELS = True, keep
    rl2 = lambda a, b: float((a - b).norm() / b.norm())
    ey, eg = rl2(outs[0][0], outs[1][0]), rl2(outs[0][1], outs[1][1])
    print("\nUp module composed vs two-launch (%s): y %.2e, grad %.2e (relative L2)" % (dtype, ey, eg))
    # two 16-bit evaluations of the same block: each rounds differently, and three InstanceNorm/ReLU layers follow (a flipped ReLU edge moves single gradient elements by O(0.1))
    assert ey < 4 * TOL[dtype] and eg < {torch.bfloat16: 0.15, torch.float16: 0.05}[dtype]


@pytest.mark.parametrize("dtype", DT)
def test_trainable_composed_up_multi_step_matches_two_launch_form(dtype, monkeypatch):
    """The trainable composed head across optimiser steps: `Up` (joint_model.py:114-124) trained for 4 SGD(momentum) steps with the
    composed operator forced on at a small size — eagerly and through a captured train.GraphedStep (raw-address chain-rule job, re-composition
    of the images after every step by ops.weights_changed) — against the two-launch form of the same module.  Graph and eager run the same
    kernels on the same data (tight bound); composed vs two-launch are two 16-bit evaluations of the block (the bound of the frozen test above)."""
    import joint_model as M
    from oracle import ref_cpu as O
    from vae_segmentation_amd import optim
    from vae_segmentation_amd import train as T
    ops = _ops()
    monkeypatch.setattr(ops, "FUSE_UP_TRAINABLE", True)
    monkeypatch.setattr(ops, "FUSE_UP_TRAINABLE_MIN_VOXELS", 0)
    monkeypatch.setattr(ops, "FUSE_UP_MIN_VOXELS", 0)
    x = rnd(2, 32, 12, 12, 12, seed=61).cuda()
    steps = 4

    def run(fuse, graph):
        monkeypatch.setattr(ops, "FUSE_UP", fuse)
        up = O.deterministic_fill_(M.Up(32, 16, norm_type=1), seed=5).cuda()
        M.set_kernel_dtype(up, dtype)
        params = [p for p in up.parameters()]
        opt = optim.SGD(params, lr=0.05, momentum=0.9)

        def loss_fn():
            ops.stats_arena_begin(x.device)
            y = up(x)
            return (y.float() ** 2).mean(), {}

        losses = []
        if graph:
            gs = T.GraphedStep(loss_fn, params, opt, warmup=1)
            for _ in range(steps):
                losses.append(gs.step().detach().clone())
            gs.loss = gs.aux = None
        else:
            for _ in range(steps):
                for p_ in params:
                    p_.grad = None
                loss, _ = loss_fn()
                loss.backward()
                opt.step()
                losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        mine = {id(p_) for p_ in params}
        assert any(k[0] in mine for k in ops._UP_TRAINABLE) == fuse, "composed trainable path %s" % ("was not taken" if fuse else "was taken")
        return [float(v) for v in losses], {n: p.detach().float().cpu().clone() for n, p in up.named_parameters()}

    init = {n: p.detach().float().clone() for n, p in O.deterministic_fill_(M.Up(32, 16, norm_type=1), seed=5).named_parameters()}
    l_e, w_e = run(True, False)
    l_g, w_g = run(True, True)
    l_t, w_t = run(False, False)
    rl2 = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-20))
    print("\ntrainable composed Up, %s: losses eager %s | graph %s | two-launch %s" % (dtype, ["%.5f" % v for v in l_e], ["%.5f" % v for v in l_g], ["%.5f" % v for v in l_t]))
    live = [n for n in w_e if float((w_t[n] - init[n]).norm()) > 1e-6 * float(init[n].norm()) + 1e-9]          # parameters that moved (dead biases do not)
    assert len(live) >= 4, live
    for n in live:
        # the same kernels, replayed: weights after 4 steps agree to rounding of the fp64-atomic statistics (bit-identical on the deterministic build)
        assert rl2(w_g[n], w_e[n]) < (1e-6 if ops.is_deterministic() else 1e-3), n
        # composed vs two-launch: compare the UPDATE the 4 steps made
        du_c, du_t = w_e[n] - init[n], w_t[n] - init[n]
        assert rl2(du_c, du_t) < {torch.bfloat16: 0.25, torch.float16: 0.08}[dtype], (n, rl2(du_c, du_t))
    for a, b in zip(l_e, l_t):
        assert abs(a - b) < 4 * TOL[dtype] * abs(b) + 1e-6
    for a, b in zip(l_e, l_g):
        assert abs(a - b) < 1e-4 * abs(b) + 1e-7
