"""Full-tensor backward parity at the benchmark's own sizes, and the determinism of the parity runs.

VERDICT r02 "What's weak" 1-2: the end-to-end gradient gates at 96^3 / 128^3 were vacuous (limits of 0.35 .. 0.9 from 8 x the reference's own
fp32-to-fp64 distance on 16 samples per tensor) and not reproducible run to run.  Here:
  * the HIP fp32 step, the CPU oracle in fp32 and the CPU oracle in fp64 run on the SAME inputs (joint96 = BASELINE configs[1], da128 =
    configs[3]); per gradient tensor the FULL-tensor relative L2 distances  HIP - f64,  oracle32 - f64  and  HIP - oracle32  are printed
    (`pytest -s`; committed as profiles/r03_parity_report.txt);
  * asserted: the HIP error is within 2 x the oracle-fp32 error (floor 2e-3) on every tensor, except tensors listed by name in the report
    (the two fp32 results are two draws of the same rounding amplification — ReLU masks flip under rounding, SURVEY F8 — so their ratio is
    heavy-tailed) which must stay within 8 x; and the MEDIAN ratio over the tensors is at most 2;
  * the library's deterministic mode makes a step bit-reproducible; the default (fp64-atomic) mode stays within the same bounds.
"""
import numpy as np
import pytest
import torch

from tests import golden_util as G

pytestmark = pytest.mark.gpu


def _mods():
    import joint_model
    from oracle import ref_cpu as O
    from vae_segmentation_amd import ops
    from vae_segmentation_amd import train as T
    return joint_model, O, T, ops


def _native_joint(M, O, side, seg_seed=None):
    seg = M.Segmentation(n_channels=1, n_class=2, norm_type=1)
    vae = M.VAE(n_channels=2, n_class=2, norm_type=1, dim=128, spatial=side)
    joint = M.Joint(models=[seg, vae])
    O.deterministic_fill_(joint, seed=0)
    if seg_seed is not None:
        O.deterministic_fill_(joint.Seg, seed=seg_seed)
    joint = joint.cuda()
    for p in joint.Vae.parameters():
        p.requires_grad = False
    joint.Vae.eval()
    return joint


def _oracle_joint(O, side, dt, seg_seed=None):
    j = O.build_joint(side)
    if seg_seed is not None:
        O.deterministic_fill_(j.Seg, seed=seg_seed)
    return j.to(dt)


def _table(name, hip, o32, o64, floor=2e-3, factor=2.0, hard=8.0):
    """hip / o32 / o64: {param name: fp64 CPU gradient tensor}.  Prints the per-tensor table, returns (ratios, outliers)."""
    rows, outliers = [], []
    print("\n==== %s: full-tensor gradient parity (relative L2) ====" % name)
    print("%-34s %10s %12s %12s %12s %7s" % ("tensor", "numel", "HIP-f64", "oracle32-f64", "HIP-oracle32", "ratio"))
    for n in o64:
        ref = o64[n]
        if G.is_dead_bias(n) or float(ref.norm()) < 1e-5 * np.sqrt(ref.numel()) * 10:
            assert float(hip[n].norm()) <= max(10 * float(o32[n].norm()), 1e-4 * np.sqrt(ref.numel())), n
            continue
        nrm = float(ref.norm())
        mine, theirs, cross = float((hip[n] - ref).norm()) / nrm, float((o32[n] - ref).norm()) / nrm, float((hip[n] - o32[n]).norm()) / nrm
        ratio = mine / max(theirs, 1e-30)
        rows.append((n, ref.numel(), mine, theirs, cross, ratio))
        flag = ""
        if mine > max(floor, factor * theirs):
            outliers.append((n, mine, theirs))
            flag = "  <-- outlier (> %gx)" % factor
        print("%-34s %10d %12.3e %12.3e %12.3e %7.2f%s" % (n, ref.numel(), mine, theirs, cross, ratio, flag))
        assert mine <= max(floor, hard * theirs), "%s %s: HIP %.3g vs oracle-fp32 %.3g (both against fp64)" % (name, n, mine, theirs)
    ratios = sorted(r[5] for r in rows)
    med = ratios[len(ratios) // 2]
    print("%s: %d tensors; HIP/oracle32 error ratio: median %.2f, 90th percentile %.2f, max %.2f; worst HIP error %.3e, worst oracle-fp32 error %.3e; "
          "%d outliers beyond %gx: %s" % (name, len(rows), med, ratios[int(0.9 * (len(ratios) - 1))], ratios[-1], max(r[2] for r in rows),
                                          max(r[3] for r in rows), len(outliers), factor, [o[0] for o in outliers]))
    return med, outliers


def test_joint96_full_tensor_gradients_vs_cpu_oracle():
    """BASELINE configs[1]: 96^3, batch 2, joint_train (main_source.py:449-471) — every Seg gradient tensor in full."""
    M, O, T, ops = _mods()
    img, lab = O.synthetic_image(2, 96, 2), O.synthetic_label(2, 96, 3)
    joint = _native_joint(M, O, 96)
    final, aux = T.joint_train_losses(joint, img.cuda(), lab.cuda())
    final.backward()
    hip = {n: p.grad.detach().double().cpu() for n, p in joint.Seg.named_parameters()}
    res = {}
    for dt in (torch.float32, torch.float64):
        oj = _oracle_joint(O, 96, dt)
        ol, _ = O.joint_train_losses(oj, img.to(dt), lab)
        ol.backward()
        res[dt] = (float(ol), {n: p.grad.detach().double() for n, p in oj.Seg.named_parameters()})
    l32, g32 = res[torch.float32]
    l64, g64 = res[torch.float64]
    print("\njoint96 loss: HIP %.9f  oracle fp32 %.9f  oracle fp64 %.9f" % (final.item(), l32, l64))
    assert abs(final.item() - l64) <= max(1e-3 * abs(l64), 3 * abs(l32 - l64))
    med, outliers = _table("joint96 (configs[1])", hip, g32, g64)
    assert med <= 2.0, med
    # The benchmarked THROUGHPUT modes on the same inputs and weights (16-bit storage of activations and packed weights, fp32 accumulation) —
    # reported, not gated (north_star's 1e-3 is an fp32 statement).  At these synthetic, randomly initialised weights the network
    # amplifies a rounding perturbation ~1e6 x (the reference's own fp32 run is 5-7 % from fp64), so ANY 16-bit storage of the conv outputs
    # decorrelates the gradient: the last table is the CPU oracle itself with its conv outputs (and their gradients) rounded to bf16 and nothing
    # else changed (golden_util.emulate_storage_rounding) — the HIP bf16 mode sits where that emulation sits.  What is asserted: the loss.
    def cosines(g):
        c = sorted(float((g[n] * g64[n]).sum() / (g[n].norm() * g64[n].norm()).clamp_min(1e-300)) for n in g64
                   if not G.is_dead_bias(n) and float(g64[n].norm()) > 0)
        return c[0], c[len(c) // 2]
    for dt, name in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        M.set_kernel_dtype(joint, dt)
        for p in joint.Seg.parameters():
            p.grad = None
        seed = torch.tensor(1024.0, device="cuda") if dt == torch.float16 else None      # fp16: Dice gradients are subnormal without a loss scale
        f16, _ = T.joint_train_losses(joint, img.cuda(), lab.cuda())
        f16.backward(gradient=seed)
        scale = 1024.0 if dt == torch.float16 else 1.0
        h16 = {n: p.grad.detach().double().cpu() / scale for n, p in joint.Seg.named_parameters()}
        print("\njoint96 loss in the %s throughput mode: HIP %.6f (fp64 %.6f, relative difference %.2e)" % (name, f16.item(), l64, abs(f16.item() - l64) / abs(l64)))
        _table("joint96, HIP %s storage mode" % name, h16, g32, g64, floor=2e-3, factor=2.0, hard=1e9)
        print("%s mode: cosine between HIP and fp64 gradient tensors: min %.3f, median %.3f" % ((name,) + cosines(h16)))
        assert abs(f16.item() - l64) <= 2e-2 * abs(l64)
    oj = _oracle_joint(O, 96, torch.float32)
    hooks = G.emulate_storage_rounding(oj, torch.bfloat16)
    ol, _ = O.joint_train_losses(oj, img, lab)
    ol.backward()
    ge = {n: p.grad.detach().double() for n, p in oj.Seg.named_parameters()}
    for h in hooks:
        h.remove()
    print("\njoint96 loss of the CPU oracle with bf16 storage of its conv outputs: %.6f" % float(ol))
    _table("joint96, CPU oracle fp32 with conv outputs rounded to bf16 (emulated storage)", ge, g32, g64, floor=2e-3, factor=2.0, hard=1e9)
    print("emulated bf16 storage: cosine between its and the fp64 gradient tensors: min %.3f, median %.3f" % cosines(ge))
    print("oracle fp32: cosine between its and the fp64 gradient tensors: min %.3f, median %.3f" % cosines(g32))
    M.set_kernel_dtype(joint, torch.float32)


def test_seg96_full_tensor_gradients_vs_cpu_oracle():
    """seg_train at 96^3, batch 2 (main_source.py:421-441): the size where the reference's own fp32 gradients are still 2e-3 .. 3e-3 from fp64
    (no VAE behind the loss), i.e. where a per-tensor limit still binds."""
    M, O, T, ops = _mods()
    img, lab = O.synthetic_image(2, 96, 2), O.synthetic_label(2, 96, 3)
    seg = O.deterministic_fill_(M.Segmentation(1, 2, norm_type=1), seed=0).cuda()
    loss, aux = T.seg_train_losses(seg, img.cuda(), lab.cuda())
    loss.backward()
    hip = {n: p.grad.detach().double().cpu() for n, p in seg.named_parameters()}
    res = {}
    for dt in (torch.float32, torch.float64):
        oseg = O.deterministic_fill_(O.Segmentation(1, 2, norm_type=1), seed=0).to(dt)
        ol, _ = O.seg_train_losses(oseg, img.to(dt), lab)
        ol.backward()
        res[dt] = (float(ol), {n: p.grad.detach().double() for n, p in oseg.named_parameters()})
    print("\nseg96 loss: HIP %.9f  oracle fp32 %.9f  oracle fp64 %.9f" % (loss.item(), res[torch.float32][0], res[torch.float64][0]))
    # no median bound here: one rounding draw decides all tensors together (the errors of a pass are perfectly correlated along the backward
    # chain).  This pass is a 4-6 x draw (HIP 1.2e-2 against the oracle's 2-3e-3 on the encoder tensors, 1.2-1.7 x on the decoder's);
    # joint96 is a 1.05 x draw, da128 a 1.4 x one.  That it is amplification and not kernel error is pinned by tests/test_gpu_backward_steps.py:
    # every single backward step of this pass agrees with an fp64 recomputation from HIP's own inputs to < 5e-6.
    _table("seg96", hip, res[torch.float32][1], res[torch.float64][1])


def test_da128_full_tensor_gradients_vs_cpu_oracle():
    """BASELINE configs[3]: 128^3 teacher-student domain-adaptation step (main_target.py:520-596, domain_loss_type 0), batch 1."""
    M, O, T, ops = _mods()
    img, lab = O.synthetic_image(1, 128, 2), O.synthetic_label(1, 128, 3)
    student, teacher = _native_joint(M, O, 128), _native_joint(M, O, 128, seg_seed=1)
    for p in teacher.parameters():
        p.requires_grad = False
    final, aux = T.domain_adaptation_losses(student, teacher, img.cuda(), lab.cuda(), lambda_vae=1.0, domain_loss_type=0)
    final.backward()
    hip = {n: p.grad.detach().double().cpu() for n, p in student.Seg.named_parameters()}
    res = {}
    for dt in (torch.float32, torch.float64):
        os_, ot = _oracle_joint(O, 128, dt), _oracle_joint(O, 128, dt, seg_seed=1)
        for p in ot.parameters():
            p.requires_grad = False
        ol, _ = O.domain_adaptation_losses(os_, ot, img.to(dt), lab, lambda_vae=1.0, domain_loss_type=0)
        ol.backward()
        res[dt] = (float(ol), {n: p.grad.detach().double() for n, p in os_.Seg.named_parameters()})
    l32, g32 = res[torch.float32]
    l64, g64 = res[torch.float64]
    print("\nda128 loss: HIP %.9f  oracle fp32 %.9f  oracle fp64 %.9f" % (final.item(), l32, l64))
    assert abs(final.item() - l64) <= max(1e-3 * abs(l64), 3 * abs(l32 - l64))
    med, outliers = _table("da128 (configs[3])", hip, g32, g64)
    assert med <= 2.0, med


def _joint_step_grads(M, O, T, side, dtype):
    joint = _native_joint(M, O, side)
    M.set_kernel_dtype(joint, dtype)
    final, aux = T.joint_train_losses(joint, O.synthetic_image(2, side, 2).cuda(), O.synthetic_label(2, side, 3).cuda())
    final.backward()
    torch.cuda.synchronize()
    return final.detach().clone(), aux["batch"]["pred"].detach().clone(), [p.grad.detach().clone() for p in joint.Seg.parameters()]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deterministic_mode_is_bit_reproducible(dtype):
    """Two runs of the same joint_train step in deterministic mode: loss, prediction and every gradient are bit-identical."""
    M, O, T, ops = _mods()
    assert ops.is_deterministic(), "the test session runs in deterministic mode (tests/conftest.py)"
    a = _joint_step_grads(M, O, T, 64, dtype)
    b = _joint_step_grads(M, O, T, 64, dtype)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for ga, gb in zip(a[2], b[2]):
        assert torch.equal(ga, gb)


def test_statistics_limbs_match_fp64_sums():
    """The four fixed-point limbs of a statistics buffer reproduce the fp64-atomic mode's (sum, sumsq) to fp64 rounding."""
    M, O, T, ops = _mods()
    x = (torch.from_numpy(2 * O.hashed_uniform(2 * 24 * 24 * 24 * 32, 7300, 1) - 1).view(2, 24, 24, 24, 32) * 3).cuda()
    det = ops.stats_total(ops.instnorm_stats(x.contiguous())).cpu()
    ops.set_deterministic(False)
    try:
        ops.stats_arena_begin(x.device)
        plain = ops.stats_total(ops.instnorm_stats(x.contiguous())).cpu()
    finally:
        ops.set_deterministic(True)
    ref = torch.stack([x.double().sum((1, 2, 3)), (x.double() ** 2).sum((1, 2, 3))], -1).cpu()
    assert float((det - ref).abs().max() / ref.abs().max()) < 1e-6           # the kernel sums fp32 partials per thread: fp32-level agreement with torch
    assert float((det - plain).abs().max() / ref.abs().max()) < 1e-12


def test_default_atomic_mode_seg32_and_joint96_vs_golden(atomic_mode):
    """The mode the benchmark runs (fp64 atomics, arrival order): the same golden checks, at the same limits, hold."""
    M, O, T, ops = _mods()
    assert not ops.is_deterministic()
    g = G.load("seg32")
    seg = O.deterministic_fill_(M.Segmentation(1, 2, norm_type=1), seed=0).cuda()
    loss, aux = T.seg_train_losses(seg, O.synthetic_image(2, 32, 2).cuda(), O.synthetic_label(2, 32, 3).cuda(), eps=1e-6)
    loss.backward()
    G.scalar_close(g, "dice_loss_eps1e6", loss.item(), 1e-3)
    G.check_tensor_f64(g, "pred", aux["batch"]["pred"], k=256, floor=1e-3)
    G.vacuity(G.check_grads_f64(g, "seg", [(n, p.grad) for n, p in seg.named_parameters()], floor=2e-3, what="seg32 (atomic mode)"), "seg32 (atomic mode)")
    g = G.load("joint96")
    joint = _native_joint(M, O, 96)
    final, aux = T.joint_train_losses(joint, O.synthetic_image(2, 96, 2).cuda(), O.synthetic_label(2, 96, 3).cuda())
    final.backward()
    G.scalar_close(g, "final", final.item(), 1e-3)
    G.check_tensor_f64(g, "pred", aux["batch"]["pred"], k=512, floor=1e-3)
    G.vacuity(G.check_grads_f64(g, "seg", [(n, p.grad) for n, p in joint.Seg.named_parameters()], floor=2e-3, what="joint96 (atomic mode)"),
              "joint96 (atomic mode)")
