"""GPU: BASELINE configs[4] — fp16 storage with dynamic loss scaling, and the 160^3 geometry.

The reference trains in fp32 only (main_source.py:117 fixes 128^3 fp32), so the oracle for the fp16 mode is fp32 / fp64 math: the
fp16 kernels are pinned per op against F.conv3d autograd on fp16-rounded operands (tests/test_gpu_ops.py, tests/test_gpu_layers.py,
2e-3), and the end-to-end mode is gated here like the bf16 mode (loss within 2 %, probabilities, direction of the near-loss gradients),
plus the loss-scaling protocol (torch.cuda.amp.GradScaler's, on the device) and the 160^3 forward in fp32 mode against the reference
golden (tests/golden/joint160_fwd.npz, oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from tests import golden_util as G

pytestmark = pytest.mark.gpu


def _mods():
    import joint_model
    from oracle import ref_cpu as O
    from vae_segmentation_amd import optim
    from vae_segmentation_amd import train as T
    return joint_model, O, T, optim


def _build_joint(M, O, side, dtype):
    seg = M.Segmentation(n_channels=1, n_class=2, norm_type=1)
    vae = M.VAE(n_channels=2, n_class=2, norm_type=1, dim=128, spatial=side)
    joint = M.Joint(models=[seg, vae])
    O.deterministic_fill_(joint, seed=0)
    joint = joint.cuda()
    for p in joint.Vae.parameters():
        p.requires_grad = False
    joint.Vae.eval()
    M.set_kernel_dtype(joint, dtype)
    return joint


def test_fp16_mode_joint96_with_loss_scaling():
    """configs[1]'s step in fp16 storage.  Without scaling the Dice gradients (O(1e-6)) sit in fp16's subnormal range; with the loss
    differentiated at scale 65536 the unscaled gradients must point the way the fp64 gradients do and the forward must be at least as
    close to fp64 as the bf16 mode's gate asks (fp16 has 3 more significand bits: tighter bounds here)."""
    M, O, T, optim = _mods()
    g = G.load("joint96")
    joint = _build_joint(M, O, 96, torch.float16)
    scaler = optim.LossScaler(init_scale=65536.0)
    final, aux = T.joint_train_losses(joint, O.synthetic_image(2, 96, 2).cuda(), O.synthetic_label(2, 96, 3).cuda())
    final.backward(gradient=scaler.seed)
    torch.cuda.synchronize()
    f64 = float(g["final@f64"])
    assert abs(final.item() - f64) / f64 < 5e-3
    assert abs(aux["recon_loss"].item() - float(g["recon_loss@f64"])) / float(g["recon_loss@f64"]) < 2e-2
    pred = G.flat64(aux["batch"]["pred"])
    perr = np.abs(pred[G.sample_idx(pred.size, 512)] - g["pred.samples@f64"])
    assert perr.mean() < 1e-2 and perr.max() < 0.1, (perr.mean(), perr.max())
    cos = {}
    for name, p in joint.Seg.named_parameters():
        assert torch.isfinite(p.grad).all(), name
        if G.is_dead_bias(name) or p.numel() < 8:
            continue
        a = G.flat64(p.grad)[G.sample_idx(p.numel(), 16)] / 65536.0
        b = g["seg.grad.%s.samples@f64" % name].astype(np.float64)
        cos[name] = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        if name.startswith("out_block"):                                       # magnitude too: the scale really was applied and removed
            assert 0.8 < np.linalg.norm(a) / np.linalg.norm(b) < 1.25, (name, np.linalg.norm(a), np.linalg.norm(b))
    print("\nfp16 gradient cosines vs fp64:", {k: round(v, 3) for k, v in cos.items()})
    near_loss = [v for k, v in cos.items() if k.startswith("out_block") or k.startswith("up5.conv.1.conv.6")]
    assert min(near_loss) > 0.95, near_loss
    # the optimiser divides by the scale: one SGD step moves the weights by lr * g, not lr * 65536 * g
    w = joint.Seg.out_block.weight
    before = w.detach().clone()
    gtrue = w.grad.detach().clone() / 65536.0
    opt = optim.SGD(joint.Seg.parameters(), lr=1e-2, momentum=0.9)
    opt.step(scaler=scaler)
    torch.cuda.synchronize()
    assert float(((before - w.detach()) - 1e-2 * gtrue).abs().max()) <= 1e-6 * float(gtrue.abs().max()) + 1e-12
    assert scaler.scale.item() == 65536.0 and scaler.found_inf.item() == 0.0 and scaler.tracker.item() == 1


def test_loss_scaler_protocol_overflow_skip_backoff_growth():
    """GradScaler semantics on the device: a step whose gradients overflow fp16 is skipped as a whole (weights and momentum untouched),
    the scale halves until the step goes through, and doubles after `growth_interval` clean steps — through HIP-graph replay, where the
    scale is read from device memory by the captured backward."""
    M, O, T, optim = _mods()
    side = 64
    joint = _build_joint(M, O, side, torch.float16)
    img, lab = O.synthetic_image(1, side, 2).cuda(), O.synthetic_label(1, side, 3).cuda()
    params = list(joint.Seg.parameters())
    opt = optim.SGD(params, lr=1e-2, momentum=0.9)
    scaler = optim.LossScaler(init_scale=2.0 ** 40, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    gs = T.GraphedStep(lambda: T.joint_train_losses(joint, img, lab), params, opt, warmup=1, scaler=scaler)
    assert gs.tail, "the scaled step's tail (finite check, scaled SGD, scale update, re-pack) must be inside the graph (VERDICT r04 item 3)"
    start = [p.detach().clone() for p in params]
    scales, moved = [], []
    for _ in range(40):
        gs.step()
        torch.cuda.synchronize()
        scales.append(scaler.scale.item())
        moved.append(any(not torch.equal(p.detach(), s) for p, s in zip(params, start)))
        if moved[-1]:
            break
    assert moved[-1], "no step ever went through: %s" % scales
    k = moved.index(True)                       # steps 0 .. k-1 overflowed and were skipped, step k was applied
    assert k >= 1, "scale 2^40 should overflow fp16 gradients"
    assert scales[:k] == [2.0 ** (40 - i - 1) for i in range(k)], scales     # halved once per skipped step
    assert scales[k] == scales[k - 1]                                          # a clean step leaves the scale alone (tracker 1 of 3)
    for p in params:
        st = opt.state[p]
        assert torch.isfinite(p).all() and torch.isfinite(st["momentum_buffer"]).all()
    # growth: the scale one notch under the overflow is a knife edge (the next steps' gradients may or may not fit), so the rule is checked well below it —
    # the scale and the tracker are device scalars the captured step reads, set here between replays
    scaler.scale.fill_(2.0 ** 12)
    scaler.tracker.zero_()
    seen = []
    for _ in range(3):
        gs.step()
        torch.cuda.synchronize()
        seen.append((scaler.scale.item(), int(scaler.tracker.item())))
    assert seen == [(2.0 ** 12, 1), (2.0 ** 12, 2), (2.0 ** 13, 0)], seen             # three clean steps in a row: growth, tracker back to zero
    assert torch.isfinite(gs.loss).all()


def test_joint160_forward_fp32_vs_reference_golden():
    """configs[4] geometry, forward: 160^3, batch 2, fp32 kernels against the reference's modules (golden joint160_fwd)."""
    M, O, T, optim = _mods()
    g = G.load("joint160_fwd")
    joint = _build_joint(M, O, 160, torch.float32)
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        final, aux = T.joint_train_losses(joint, O.synthetic_image(2, 160, 2).cuda(), O.synthetic_label(2, 160, 3).cuda())
    torch.cuda.synchronize()
    for key, val in (("final", final), ("recon_loss", aux["recon_loss"]), ("dice_loss", aux["dice_loss"])):
        G.scalar_close(g, key, val.item(), 1e-3)
    b = aux["batch"]
    G.check_tensor_f64(g, "pred", b["pred"], k=512, floor=1e-3)
    G.check_tensor_f64(g, "recon", b["recon"], k=512, floor=1e-3)
    assert G.rel_l2(b["mean"].detach().cpu(), g["mean@f64"]) < max(1e-3, 3 * G.rel_l2(g["mean"], g["mean@f64"]))
    print("\n160^3 B=2 fp32 forward: peak device memory %.2f GB" % (torch.cuda.max_memory_allocated() / 1e9))


def test_joint160_fp16_train_steps_and_memory():
    """configs[4]: 160^3, batch 2, fp16 storage + dynamic loss scale, HIP-graph replayed train steps.  Loss against the fp64 golden of
    the same forward (2 %), finite weights after the steps, and the peak device memory that backs DESIGN.md's "no activation
    recomputation needed on 288 GB" (printed; asserted to be a small fraction of the card)."""
    M, O, T, optim = _mods()
    g = G.load("joint160_fwd")
    joint = _build_joint(M, O, 160, torch.float16)
    img, lab = O.synthetic_image(2, 160, 2).cuda(), O.synthetic_label(2, 160, 3).cuda()
    params = list(joint.Seg.parameters())
    opt = optim.SGD(params, lr=1e-2, momentum=0.9)
    scaler = optim.LossScaler(init_scale=65536.0)
    torch.cuda.reset_peak_memory_stats()
    gs = T.GraphedStep(lambda: T.joint_train_losses(joint, img, lab), params, opt, warmup=1, scaler=scaler)
    first = None
    for i in range(3):
        loss = gs.step()
        torch.cuda.synchronize()
        if first is None:
            first = float(gs.aux["dice_loss"].item()), float(gs.aux["recon_loss"].item())
    peak = torch.cuda.max_memory_allocated() / 1e9
    print("\n160^3 B=2 fp16 joint_train (graph + warm-up pools): peak device memory %.2f GB; loss scale %g" % (peak, scaler.scale.item()))
    assert abs(first[0] - float(g["dice_loss@f64"])) / float(g["dice_loss@f64"]) < 2e-2
    assert abs(first[1] - float(g["recon_loss@f64"])) / float(g["recon_loss@f64"]) < 5e-2
    assert scaler.scale.item() == 65536.0, "no overflow expected at the default scale"
    assert all(torch.isfinite(p).all() for p in params)
    assert any(not torch.equal(p.detach().cpu(), q.detach()) for p, q in zip(params[:4], list(O.deterministic_fill_(M.Segmentation(1, 2, norm_type=1), seed=0).parameters())[:4]))
    assert peak < 40.0


def _sample_cosine(g, named_grads, scale=1.0):
    """cosine between the HIP gradients and joint160_bwd's stored samples (64 per tensor, the reference's own fp32 run), over all live tensors, and the worst
    per-tensor ratio of l2 norms"""
    mine, ref, ratios = [], [], {}
    for name, gr in named_grads:
        key = "seg.grad.%s" % name
        if G.is_dead_bias(name) or key + ".l2" not in g:
            continue
        a = G.flat64(gr) / scale
        mine.append(a[G.sample_idx(a.size, 64)])
        ref.append(g[key + ".samples"].astype(np.float64))
        ratios[name] = float(np.sqrt((a * a).sum()) / max(float(g[key + ".l2"]), 1e-300))
    m, r = np.concatenate(mine), np.concatenate(ref)
    return float((m * r).sum() / np.sqrt((m * m).sum() * (r * r).sum())), ratios


def test_joint160_backward_vs_reference_fp32_golden_and_fp16_direction():
    """BASELINE configs[4]'s geometry, the BACKWARD half: tests/golden/joint160_bwd.npz holds the reference's own eager fp32
    joint_train step at 160^3 (batch 1: its fp64 twin does not fit the build container, so there is no fp64 yardstick and no envelope at this size —
    two fp32 implementations are two draws of the network's rounding amplification, 1e-2 .. 1e-1 apart at 128^3: tests/golden/envelopes2.npz).
    Gates: the fp32 (parity) mode's losses to 1e-3, its probabilities to the golden's samples, every live gradient tensor's norm within a factor 1.5 of the
    reference's and the sampled whole-gradient cosine >= 0.95.  Then the fp16 mode (loss scale 65536) at a TRAINED state — 60 SGD steps of the fp32 mode on
    this very batch: a state where gradients are not rounding noise (tests/test_gpu_convergence.py says why the random-weight state proves nothing for a
    16-bit mode) — against the fp32 mode's gradient there: whole-gradient cosine >= 0.9, the link that ties configs[4]'s fp16 gradient to the pinned fp32 kernels."""
    M, O, T, optim = _mods()
    from vae_segmentation_amd import ops
    g = G.load("joint160_bwd")
    joint = _build_joint(M, O, 160, torch.float32)
    img, lab = O.synthetic_image(1, 160, 2).cuda(), O.synthetic_label(1, 160, 3).cuda()
    final, aux = T.joint_train_losses(joint, img, lab)
    final.backward()
    torch.cuda.synchronize()
    for key, val in (("final", final), ("recon_loss", aux["recon_loss"]), ("dice_loss", aux["dice_loss"])):
        assert abs(val.item() - float(g[key])) <= 1e-3 * abs(float(g[key])), (key, val.item(), float(g[key]))
    pred = G.flat64(aux["batch"]["pred"])
    perr = np.abs(pred[G.sample_idx(pred.size, 512)] - g["pred.samples"].astype(np.float64))
    assert perr.max() < 2e-2 and perr.mean() < 1e-3, (perr.max(), perr.mean())
    cos, ratios = _sample_cosine(g, [(n, p.grad) for n, p in joint.Seg.named_parameters()])
    worst = max(ratios.items(), key=lambda kv: abs(np.log(kv[1])))
    print("\njoint160 backward, fp32 mode vs the reference's fp32 run: sampled whole-gradient cosine %.4f; per-tensor norm ratio worst %s %.3f" % (cos, worst[0], worst[1]))
    assert cos >= 0.95, cos
    assert all(1 / 1.5 < r < 1.5 for r in ratios.values()), worst
    # ---- a trained state, then fp16 against the fp32 mode ----
    del final, aux, pred                        # GraphedStep refuses parameters an earlier eager pass's autograd graph still references
    params = list(joint.Seg.parameters())
    opt = optim.SGD(params, lr=5e-2, momentum=0.9)
    for p in params:
        p.grad = None
    gs = T.GraphedStep(lambda: T.joint_train_losses(joint, img, lab), params, opt, warmup=1)
    l0 = float(gs.step())
    for _ in range(59):
        l1 = float(gs.step())
    torch.cuda.synchronize()
    assert l1 < l0 - 0.05, (l0, l1)
    state = {k: v.detach().clone() for k, v in joint.Seg.state_dict().items()}
    del gs
    grads = {}
    for name, dtype in (("fp32", torch.float32), ("fp16", torch.float16)):
        j = _build_joint(M, O, 160, dtype)
        j.Seg.load_state_dict(state)
        ops.weights_changed()
        seed = torch.tensor(65536.0, device="cuda") if dtype == torch.float16 else None
        f, _ = T.joint_train_losses(j, img, lab)
        f.backward(gradient=seed)
        torch.cuda.synchronize()
        sc = 65536.0 if dtype == torch.float16 else 1.0
        grads[name] = (float(f), torch.cat([p.grad.detach().double().flatten() / sc for n, p in j.Seg.named_parameters() if not G.is_dead_bias(n)]))
        del j, f
    (lf, a), (lh, b) = grads["fp32"], grads["fp16"]
    whole = float((a * b).sum() / (a.norm() * b.norm()))
    print("joint160 trained state (60 steps, loss %.4f -> %.4f): fp16 loss %.5f vs fp32-mode %.5f, whole-gradient cosine fp16 vs fp32 mode %.4f" % (l0, l1, lh, lf, whole))
    assert abs(lh - lf) <= 2e-2 * abs(lf)
    assert torch.isfinite(b).all() and whole >= 0.9, whole


def test_activation_recomputation_at_160_cubed_is_bit_identical():
    """configs[4] as BASELINE words it — 160^3, fp16, WITH activation checkpointing: the recomputing pass gives the stored-activation
    pass's loss and gradients bit for bit at the configuration's own size (deterministic build), and its peak memory is lower."""
    M, O, T, optim = _mods()
    from vae_segmentation_amd import ops
    assert ops.is_deterministic()
    img, lab = O.synthetic_image(2, 160, 2).cuda(), O.synthetic_label(2, 160, 3).cuda()
    seed = torch.tensor(65536.0, device="cuda")
    res = {}
    for rec in (False, True):
        joint = _build_joint(M, O, 160, torch.float16)
        M.set_recompute(rec)
        ops.set_wgrad_grouping(False)
        try:
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            final, aux = T.joint_train_losses(joint, img, lab)
            final.backward(gradient=seed)
            torch.cuda.synchronize()
            res[rec] = (final.detach().clone(), [p.grad.detach().clone() for p in joint.Seg.parameters()], torch.cuda.max_memory_allocated() - base)
        finally:
            M.set_recompute(False)
            ops.set_wgrad_grouping(True)
        del joint, final, aux
    assert torch.equal(res[False][0], res[True][0])
    for a, b in zip(res[False][1], res[True][1]):
        assert torch.equal(a, b)
    print("\n160^3 B=2 fp16 joint_train pass: peak %.2f GB kept, %.2f GB with recomputation" % (res[False][2] / 1e9, res[True][2] / 1e9))
    assert res[True][2] < 0.9 * res[False][2]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_activation_recomputation_gives_identical_gradients_and_saves_memory(dtype):
    """BASELINE configs[4] names activation checkpointing: modules.set_recompute re-runs every Down / Up block in backward instead of keeping
    its interior activations.  In the deterministic build (the test session's) the recomputed statistics are bit-identical to the first pass,
    so the gradients must be too; the peak memory of the pass must drop."""
    M, O, T, optim = _mods()
    from vae_segmentation_amd import ops
    assert ops.is_deterministic()
    side = 64
    img, lab = O.synthetic_image(2, side, 2).cuda(), O.synthetic_label(2, side, 3).cuda()
    seed = torch.tensor(1024.0, device="cuda") if dtype == torch.float16 else None      # a fixed loss scale: fp16 Dice gradients are subnormal without one
    res = {}
    for rec in (False, True):
        joint = _build_joint(M, O, side, dtype)
        M.set_recompute(rec)
        ops.set_wgrad_grouping(False)          # both arms launch the weight gradients per layer (recomputation does; the grouped launches split K differently)
        try:
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            final, aux = T.joint_train_losses(joint, img, lab)
            final.backward(gradient=seed)
            torch.cuda.synchronize()
            res[rec] = (final.detach().clone(), [p.grad.detach().clone() for p in joint.Seg.parameters()], torch.cuda.max_memory_allocated() - base)
        finally:
            M.set_recompute(False)
            ops.set_wgrad_grouping(True)
        del joint, final, aux
    assert torch.equal(res[False][0], res[True][0])
    for a, b in zip(res[False][1], res[True][1]):
        assert torch.equal(a, b)
    print("\npeak memory of one %s joint_train pass at %d^3, batch 2: %.0f MB kept, %.0f MB with recomputation" % (dtype, side, res[False][2] / 2 ** 20, res[True][2] / 2 ** 20))
    # at 64^3 the fp32 planar tensors of the module boundary (image, prediction, reconstruction, one-hot, their gradients) are a large fixed share,
    # larger still next to 16-bit activations (fp16: 174 MB of 210 MB measured)
    assert res[True][2] < 0.9 * res[False][2]
