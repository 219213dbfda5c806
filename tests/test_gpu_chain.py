"""GPU parity of the chain kernels (csrc/chain.h, ops.ConvK3Chain; round 6): the three 3x3x3 convolutions of a DoubleConv (joint_model.py:35-52) at the
deep levels as ONE launch each way must give what the per-layer launches give — in the deterministic build BIT FOR BIT (same per-workgroup arithmetic,
commuting integer statistics), in the benchmarked build to the rounding of the fp64 atomics — and the per-layer launches are pinned against CPU autograd
by tests/test_gpu_layers.py at the same shapes.  Every chain ends with a host check of the kernels' fault word (a bounded wait that gave up)."""
import pytest
import torch

from tests.test_gpu_ops import relerr, to_cl

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16, torch.float16]
# (N, Cin, Cout, side, lazy input): the DoubleConvs of configs[1] at 6^3 / 3^3 (Seg down4, VAE down4 / down5 / up1) and of configs[3] / [4] (4^3, 5^3), B = 3 (a ragged slot walk)
CASES = [(2, 64, 128, 6, False), (2, 256, 128, 6, False), (2, 128, 256, 3, False), (2, 128, 128, 6, True), (1, 256, 256, 4, False), (2, 128, 256, 5, False),
         (3, 64, 64, 6, True), (9, 32, 64, 3, False), (2, 128, 128, 6, "two")]


def _ops():
    from vae_segmentation_amd import ops
    return ops


def _run(ops, chain, x_cl, xs, ws, gy, two):
    """forward + backward of the block, as one chain or layer by layer; -> (y, stats, gx, weight gradients)"""
    ops.CHAIN = chain
    x = x_cl.clone().requires_grad_(True)
    params = [w.clone().requires_grad_(True) for w in ws]
    with ops.arena_scope(x.device):
        if chain:
            flat = []
            for p in params:
                flat += [p, None]
            y, ys = ops.ConvK3Chain.apply(x, xs, *flat)
        else:
            y, ys = x, xs
            for p in params:
                y, ys = ops.ConvK3.apply(y, ys, p, None)
        y.backward(gy)
    return y.detach(), ops.stats_total(ys), x.grad, [p.grad for p in params]


@pytest.mark.parametrize("lib_mode", ["det", "atomic"], indirect=True)
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", CASES)
def test_chain_equals_per_layer_launches(case, dtype, lib_mode):
    ops = _ops()
    n, cin, cout, side, lazy = case
    two = lazy == "two"
    torch.manual_seed(7)
    x = torch.randn(n, cin, side, side, side)
    x_cl = to_cl(x, cin, dtype)
    xs = ops.instnorm_stats(x_cl) if lazy else None
    chans = [(cin, cout), (cout, cout)] + ([] if two else [(cout, cout)])
    ws = [(torch.randn(co, ci, 3, 3, 3) / (27 * ci) ** 0.5).cuda() for ci, co in chans]
    gy = to_cl(torch.randn(n, cout, side, side, side), cout, dtype)
    holders = [type("Holder", (), {"weight": torch.empty(co, ci, 3, 3, 3, device="meta")}) for ci, co in chans]
    assert ops.chain_ok(x_cl, xs, holders), "the case must be one the chain kernels take"
    was = ops.CHAIN
    try:
        ref = _run(ops, False, x_cl, xs, ws, gy, two)
        got = _run(ops, True, x_cl, xs, ws, gy, two)
    finally:
        ops.CHAIN = was
    ops.chain_fault()
    names = ["y", "stats", "gx"] + ["gw%d" % i for i in range(len(ws))]
    flat_ref = [ref[0], ref[1], ref[2]] + ref[3]
    flat_got = [got[0], got[1], got[2]] + got[3]
    if lib_mode == "det":
        for nm, a, b in zip(names, flat_ref, flat_got):
            assert torch.equal(a, b), "%s differs between the chain and the per-layer launches (max %g)" % (nm, (a.double() - b.double()).abs().max().item())
    else:
        # fp64 atomics arrive in another order: statistics differ in their last bits, and a 16-bit rounding of a conv output can flip on that
        tol = {torch.float32: 2e-5, torch.bfloat16: 1.5e-2, torch.float16: 2e-3}[dtype]
        for nm, a, b in zip(names, flat_ref, flat_got):
            e = relerr(b.double().cpu(), a.double().cpu())
            assert e < tol, "%s: %g" % (nm, e)


def test_double_conv_module_takes_the_chain_and_matches(monkeypatch):
    """modules.DoubleConv routes its three convs through ONE ConvK3Chain where the library takes the shape, inside Down / Up blocks as well; the block's
    outputs and every parameter gradient equal the per-layer path's bit for bit (deterministic build: the test-suite default)."""
    ops = _ops()
    from vae_segmentation_amd import modules
    torch.manual_seed(3)
    blk = modules.Down(64, 128, norm_type=1).cuda()
    modules.set_kernel_dtype(blk, torch.bfloat16)
    x = to_cl(torch.randn(2, 64, 12, 12, 12), 64, torch.bfloat16)
    xs = ops.instnorm_stats(x)
    calls = []
    orig = ops.ConvK3Chain.apply

    def spy(*a):
        calls.append(len(a))
        return orig(*a)
    res = {}
    for chain in (False, True):
        monkeypatch.setattr(ops, "CHAIN", chain)
        monkeypatch.setattr(ops.ConvK3Chain, "apply", spy)
        for p in blk.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        with ops.arena_scope(x.device):
            out = blk(modules.Act(xi, xs))
            out.raw.float().square().sum().backward()
        res[chain] = [out.raw.detach().clone(), ops.stats_total(out.stats).clone(), xi.grad.clone()] + [p.grad.clone() for p in blk.parameters() if p.grad is not None]
    ops.chain_fault()
    assert calls == [8], "one chain of three layers (x, stats, three weight / bias pairs) when chains are on, none when off: %s" % calls
    assert len(res[False]) == len(res[True])
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)


# (N, Cin, Cout, side): the 3x3x3 layers with lazy inputs at the 24^3 / 12^3 levels of configs[1] and at the 32^3 / 16^3 / 8^3 levels of configs[3]
EA_CASES = [(2, 32, 32, 24), (2, 16, 32, 24), (2, 64, 64, 12), (2, 128, 64, 12), (2, 32, 64, 12), (1, 32, 32, 32), (1, 64, 64, 16), (1, 128, 128, 8), (3, 32, 32, 10)]


@pytest.mark.parametrize("lib_mode", ["det", "atomic"], indirect=True)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", EA_CASES)
def test_epilogue_apply_equals_backward_data_then_apply(case, dtype, lib_mode):
    """k3b_kernel<..., EA> (csrc/igemm_k3b.h; fp32 parity mode: k3x_kernel<..., EA>, csrc/igemm_k3x.h): the backward-data launch of a 3x3x3 conv on a lazy input applies the InstanceNorm+ReLU backward to its own outputs
    after a per-sample arrival counter — against the two launches it replaces (which tests/test_gpu_layers.py pins to CPU autograd at these shapes)."""
    ops = _ops()
    n, cin, cout, side = case
    torch.manual_seed(11)
    x_cl = to_cl(torch.randn(n, cin, side, side, side), cin, dtype)
    xs = ops.instnorm_stats(x_cl)
    w = (torch.randn(cout, cin, 3, 3, 3) / (27 * cin) ** 0.5).cuda()
    gy = to_cl(torch.randn(n, cout, side, side, side), cout, dtype)
    if not ops.lib.vs_conv_k3_bwd_data_applied_supported(n, side, side, side, cout, cin, ops.vs_dtype(x_cl)):
        assert dtype == torch.float32, "the 16-bit cases must be ones the kernel takes"
        pytest.skip("the parity mode's kernel runs one workgroup per CU: this launch has more than 256")
    res = {}
    was = ops.EPILOGUE_APPLY
    try:
        for ea in (False, True):
            ops.EPILOGUE_APPLY = ea
            x = x_cl.clone().requires_grad_(True)
            wt = w.clone().requires_grad_(True)
            with ops.arena_scope(x.device):
                y, _ = ops.ConvK3.apply(x, xs, wt, None)
                y.backward(gy)
            res[ea] = (x.grad.clone(), wt.grad.clone())
    finally:
        ops.EPILOGUE_APPLY = was
    ops.chain_fault()
    for nm, a, b in zip(("gx", "gw"), res[False], res[True]):
        if lib_mode == "det":
            assert torch.equal(a, b), "%s differs (max %g)" % (nm, (a.double() - b.double()).abs().max().item())
        else:
            e = relerr(b.double().cpu(), a.double().cpu())
            assert e < {torch.float32: 2e-5, torch.bfloat16: 1.5e-2, torch.float16: 2e-3}[dtype], "%s: %g" % (nm, e)


@pytest.mark.parametrize("lib_mode", ["det", "atomic"], indirect=True)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("case", [(1, 2, 64, 6, False), (1, 2, 32, 12, True), (0, 2, 64, 6, False), (0, 2, 32, 12, True), (1, 1, 128, 3, True), (0, 2, 16, 24, False),
                                  (1, 2, 16, 24, True)])
def test_stride2_epilogue_apply_equals_backward_data_then_apply(case, dtype, lib_mode):
    """g1_kernel's epilogue apply (csrc/igemm.h, vs_conv_s2_bwd_data_applied): the backward-data launch of Conv3d(k2, s2) (scatter) / ConvTranspose3d(k2, s2) (gather) on a
    lazy input applies the InstanceNorm+ReLU backward to its own outputs — with the U-Net skip's parked gradient summed in — against the two launches it replaces
    (which tests/test_gpu_layers.py pins to CPU autograd)."""
    _stride2_case(_ops(), case, dtype, lib_mode)


def _stride2_case(ops, case, dtype, lib_mode):
    scatter, n, c, side, with_add = case          # side: the COARSE grid's extent; c channels on both sides
    torch.manual_seed(5)
    fine, coarse = (side * 2,) * 3, (side,) * 3
    if scatter:      # backward of Conv3d(c, c, 2, stride 2): x fine (lazy), gy coarse
        x_cl = to_cl(torch.randn(n, c, *fine), c, dtype)
        gy = to_cl(torch.randn(n, c, *coarse), c, dtype)
        w = (torch.randn(c, c, 2, 2, 2) / (8 * c) ** 0.5).cuda()
        form = ops.VS_PACK_SCATTER_D1
    else:            # backward of ConvTranspose3d(c, c, 2, stride 2): x coarse (lazy), gy fine
        x_cl = to_cl(torch.randn(n, c, *coarse), c, dtype)
        gy = to_cl(torch.randn(n, c, *fine), c, dtype)
        w = (torch.randn(c, c, 2, 2, 2) / (8 * c) ** 0.5).cuda()
        form = ops.VS_PACK_ROWS_D0
    xs = ops.instnorm_stats(x_cl)
    add = to_cl(torch.randn(n, c, *(fine if scatter else coarse)) * 0.1, c, dtype) if with_add else None
    gn, gd, gh, gw, gc = gy.shape
    if not ops.lib.vs_conv_s2_bwd_data_applied_supported(gn, gd, gh, gw, gc, c, 1 if scatter else 0, ops.vs_dtype(x_cl)):
        assert scatter and side == 24, "a case the kernel must take"
        pytest.skip("48^3 x 16 scatter: 864 workgroups, more than the 256 that are certainly resident")
    wpb = ops.pack_weight(w, form, gc, dtype)
    res = {}
    was = ops.EPILOGUE_APPLY
    try:
        for ea in (False, True):
            ops.EPILOGUE_APPLY = ea
            with ops.arena_scope(x_cl.device):
                if add is not None:
                    ops._PENDING["grads"][(x_cl.data_ptr(), tuple(x_cl.shape))] = add
                res[ea] = ops.conv_bwd_data_lazy(gy, wpb, x_cl, xs, ops.VS_CONV_K2S2, scatter=bool(scatter)).clone()
                assert not ops._PENDING["grads"]
    finally:
        ops.EPILOGUE_APPLY = was
        ops._PENDING["grads"].clear()
    ops.chain_fault()
    a, b = res[False], res[True]
    if lib_mode == "det":
        assert torch.equal(a, b), "gx differs (max %g)" % (a.double() - b.double()).abs().max().item()
    else:
        e = relerr(b.double().cpu(), a.double().cpu())
        assert e < {torch.float32: 2e-5, torch.bfloat16: 1.5e-2, torch.float16: 2e-3}[dtype], "gx: %g" % e


@pytest.mark.parametrize("scatter", [0, 1])
def test_stride2_epilogue_apply_on_the_exact_f32_kernels(scatter):
    """the same with the parity mode's limb arithmetic switched off (vs_config.f32_limbs = 0: g1_kernel<float, ..., LIMB = false, EA>) — an A/B configuration, kept correct"""
    ops = _ops()
    with ops.config(f32_limbs=0):
        was = ops.is_deterministic()
        ops.set_deterministic(True)
        try:
            _stride2_case(ops, (scatter, 2, 32, 6, True), torch.float32, "det")
        finally:
            ops.set_deterministic(was)
