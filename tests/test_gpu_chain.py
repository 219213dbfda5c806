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
