"""GPU parity, block / network / train-step level, through the joint_model.py surface.

fp32 kernel mode is the parity gate named by BASELINE.json's north_star ("within 1e-3 relative fp32"):
  * against the golden fixtures the unmodified reference produced (tests/golden, oracle/make_golden.py), and
  * against the CPU oracle on the same seeded inputs at sizes it finishes in seconds.
bf16 mode (the throughput mode) is checked with looser, separately stated tolerances.
"""
import numpy as np
import pytest
import torch

from tests import golden_util as G

pytestmark = pytest.mark.gpu

RTOL_FP32 = 1e-3          # north_star tolerance (floor of every fp64-yardstick check below)
RTOL_GRAD_FP32 = 2e-3     # floor for gradients; see tests/golden_util.py: limits are max(floor, k x the
                          # reference-fp32 run's own distance to the same code run in fp64), k = 3 for forward
                          # tensors / losses and 2 for gradients; a gradient tensor between 2x and 8x is listed by name as an outlier
                          # (profiles/r03_parity_report.txt), beyond 8x it fails.  The runs are made in the library's deterministic mode
                          # (tests/conftest.py), so every number here reproduces bit for bit.  Through ~60 InstanceNorm/ReLU layers the
                          # reference's eager fp32 gradients themselves sit 1e-2..1e-1 from the fp64 result at
                          # 64^3..128^3 (at this random init the net amplifies a 1e-7 rounding perturbation ~1e6x:
                          # ReLU masks flip), and that distance is itself a random draw per tensor.


def _mods():
    import joint_model
    from oracle import ref_cpu as O
    from vae_segmentation_amd import train as T
    return joint_model, O, T


def _fill(module, seed, O):
    O.deterministic_fill_(module, seed=seed)
    return module.cuda()


BLOCKS = {
    "conv_2_8": lambda M: M.Conv(2, 8, norm_type=1),
    "dconv_8_16": lambda M: M.DoubleConv(8, 16, norm_type=1),
    "down_8_16": lambda M: M.Down(8, 16, norm_type=1),
    "up_16_8": lambda M: M.Up(16, 8, norm_type=1),
    "down_64_128": lambda M: M.Down(64, 128, norm_type=1),
    "up_256_128": lambda M: M.Up(256, 128, norm_type=1),
}


@pytest.mark.parametrize("tag", sorted(BLOCKS))
def test_blocks_vs_reference_golden(tag):
    M, O, _ = _mods()
    g = G.load("blocks")
    seed = int(g[tag + ".seed"])
    shape = tuple(int(v) for v in g[tag + ".shape"])
    mod = _fill(BLOCKS[tag](M), seed, O)
    x = torch.from_numpy(2 * O.hashed_uniform(int(np.prod(shape)), 7001, seed) - 1).view(shape).cuda().requires_grad_(True)
    y = mod(x)
    w = torch.from_numpy(2 * O.hashed_uniform(y.numel(), 7002, seed) - 1).view_as(y).cuda()
    (y * w).sum().backward()
    G.check_tensor(g, tag + ".out", y, rtol=RTOL_FP32, what=tag)
    G.check_tensor(g, tag + ".gin", x.grad, rtol=RTOL_FP32, what=tag)
    G.check_grads(g, tag, [(n, p.grad) for n, p in mod.named_parameters()], rtol=RTOL_FP32, what=tag,
                  dead=G.is_dead_bias_in_block(tag))


def test_state_dict_contract():
    M, O, _ = _mods()
    seg = M.Segmentation(1, 2, norm_type=1)
    vae = M.VAE(2, 2, norm_type=1, dim=128)
    oseg, ovae = O.Segmentation(1, 2, norm_type=1), O.VAE(2, 2, norm_type=1, dim=128)
    for a, b in ((seg, oseg), (vae, ovae)):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys())
        assert all(sa[k].shape == sb[k].shape for k in sa)
    assert len(seg.state_dict()) == 68 and sum(p.numel() for p in seg.parameters()) == 2276018
    assert len(vae.state_dict()) == 90 and sum(p.numel() for p in vae.parameters()) == 15434378
    assert vae.fc_mean.weight.shape == (128, 16384)


def test_seg32_vs_golden_and_oracle():
    M, O, T = _mods()
    g = G.load("seg32")
    seg = _fill(M.Segmentation(1, 2, norm_type=1), 0, O)
    img, lab = O.synthetic_image(2, 32, 2), O.synthetic_label(2, 32, 3)
    loss, aux = T.seg_train_losses(seg, img.cuda(), lab.cuda(), eps=1e-6)
    loss.backward()
    G.scalar_close(g, "dice_loss_eps1e6", loss.item(), RTOL_FP32)
    G.check_tensor_f64(g, "pred", aux["batch"]["pred"], k=256, floor=RTOL_FP32)
    G.check_grads_f64(g, "seg", [(n, p.grad) for n, p in seg.named_parameters()], floor=RTOL_GRAD_FP32)
    # full-tensor comparison against the oracle on the same inputs
    oseg = O.deterministic_fill_(O.Segmentation(1, 2, norm_type=1), seed=0)
    ol, oaux = O.seg_train_losses(oseg, img, lab, eps=1e-6)
    ol.backward()
    assert G.rel_l2(aux["batch"]["pred"].detach().cpu(), oaux["batch"]["pred"].detach()) < RTOL_FP32
    for (n1, p1), (n2, p2) in zip(seg.named_parameters(), oseg.named_parameters()):
        if p2.grad.norm() > 1e-4 * np.sqrt(p2.numel()):
            assert G.rel_l2(p1.grad.cpu(), p2.grad) < 2e-2, n1      # fp32-vs-fp32 full tensors (both ~1e-2 from fp64)


def test_multiclass_steps_vs_reference_golden():
    """More than one labelled structure (n_class = 1 + the number of --pan_index entries, main_source.py:92-93): seg_train with four classes at
    32^3 and joint_train with three at 64^3 (Segmentation -> three-channel prediction -> VAE), fp32 kernels, against the reference's modules."""
    M, O, T = _mods()
    g = G.sub(G.load("multiclass"), "seg32_c4/")
    seg = _fill(M.Segmentation(1, 4, norm_type=1), 0, O)
    loss, aux = T.seg_train_losses(seg, O.synthetic_image(2, 32, 2).cuda(), O.synthetic_label(2, 32, 3, n_class=4).cuda(), n_class=4)
    loss.backward()
    assert aux["batch"]["pred"].shape == (2, 4, 32, 32, 32)
    G.scalar_close(g, "dice_loss", loss.item(), RTOL_FP32)
    G.check_tensor_f64(g, "pred", aux["batch"]["pred"], k=256, floor=RTOL_FP32)
    G.check_grads_f64(g, "seg", [(n, p.grad) for n, p in seg.named_parameters()], floor=RTOL_GRAD_FP32, what="seg32_c4")
    g = G.sub(G.load("multiclass"), "joint64_c3/")
    seg = M.Segmentation(n_channels=1, n_class=3, norm_type=1)
    vae = M.VAE(n_channels=3, n_class=3, norm_type=1, dim=128, spatial=64)
    joint = M.Joint(models=[seg, vae])
    O.deterministic_fill_(joint, seed=0)
    joint = joint.cuda()
    for p in joint.Vae.parameters():
        p.requires_grad = False
    joint.Vae.eval()
    final, aux = T.joint_train_losses(joint, O.synthetic_image(2, 64, 2).cuda(), O.synthetic_label(2, 64, 3, n_class=3).cuda(), n_class=3)
    final.backward()
    torch.cuda.synchronize()
    for key, val in (("final", final), ("recon_loss", aux["recon_loss"]), ("dice_loss", aux["dice_loss"])):
        G.scalar_close(g, key, val.item(), RTOL_FP32)
    b = aux["batch"]
    G.check_tensor_f64(g, "pred", b["pred"], k=512, floor=RTOL_FP32)
    G.check_tensor_f64(g, "recon", b["recon"], k=512, floor=RTOL_FP32)
    rep = G.check_grads_f64(g, "seg", [(n, p.grad) for n, p in joint.Seg.named_parameters()], floor=RTOL_GRAD_FP32, what="joint64_c3")
    G.vacuity(rep, "joint64_c3")
    assert all(p.grad is None for p in joint.Vae.parameters())
    with pytest.raises(NotImplementedError):
        M.Segmentation(1, 9, norm_type=1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_multiclass_joint_step_16bit_and_graph_replay(dtype):
    """The same three-class joint step in the 16-bit storage modes: losses close to the fp64 golden; captured and replayed with its optimiser
    (train.GraphedStep), two steps equal to two eager steps."""
    M, O, T = _mods()
    from vae_segmentation_amd import optim
    g = G.sub(G.load("multiclass"), "joint64_c3/")

    def build():
        seg = M.Segmentation(n_channels=1, n_class=3, norm_type=1)
        vae = M.VAE(n_channels=3, n_class=3, norm_type=1, dim=128, spatial=64)
        joint = M.Joint(models=[seg, vae])
        O.deterministic_fill_(joint, seed=0)
        joint = M.set_kernel_dtype(joint.cuda(), dtype)
        for p in joint.Vae.parameters():
            p.requires_grad = False
        joint.Vae.eval()
        return joint

    img, lab = O.synthetic_image(2, 64, 2).cuda(), O.synthetic_label(2, 64, 3, n_class=3).cuda()
    ja, jb = build(), build()
    final, aux = T.joint_train_losses(ja, img, lab, n_class=3)
    tol = 3e-2 if dtype == torch.bfloat16 else 6e-3
    for key, val in (("final", final), ("recon_loss", aux["recon_loss"]), ("dice_loss", aux["dice_loss"])):
        assert abs(val.item() - float(g[key + "@f64"])) <= tol * abs(float(g[key + "@f64"])), (key, val.item(), float(g[key + "@f64"]))
    assert aux["batch"]["recon"].shape == (2, 3, 64, 64, 64)
    pa, pb = [p for p in ja.parameters() if p.requires_grad], [p for p in jb.parameters() if p.requires_grad]
    opt_a, opt_b = optim.SGD(pa, lr=1e-2, momentum=0.9), optim.SGD(pb, lr=1e-2, momentum=0.9)
    gs = T.GraphedStep(lambda: T.joint_train_losses(jb, img, lab, n_class=3), pb, opt_b, warmup=1)
    for _ in range(2):
        opt_a.zero_grad()
        la, _ = T.joint_train_losses(ja, img, lab, n_class=3)
        la.backward()
        opt_a.step()
        lb = gs.step()
        assert abs(la.item() - lb.item()) < 2e-3 * abs(la.item())
    for p1, p2 in zip(pa, pb):
        assert G.rel_l2(p1.detach().cpu(), p2.detach().cpu()) < 2e-3


def test_seg96_vs_reference_golden():
    """seg_train at the BASELINE size (96^3, B=2): the gradient check whose 2e-3 floor binds at the real layer shapes."""
    M, O, T = _mods()
    g = G.load("seg96")
    seg = _fill(M.Segmentation(1, 2, norm_type=1), 0, O)
    loss, aux = T.seg_train_losses(seg, O.synthetic_image(2, 96, 2).cuda(), O.synthetic_label(2, 96, 3).cuda())
    loss.backward()
    G.scalar_close(g, "dice_loss", loss.item(), RTOL_FP32)
    G.check_tensor_env(g, "seg96", "pred", aux["batch"]["pred"], k=512, floor=RTOL_FP32)
    # every gradient tensor within 1.5 x the MEASURED envelope of the reference's own fp32 arithmetic (tests/golden/envelopes.npz: the reference's
    # fp32 step with its weights moved by +-1 ulp lands 2.5e-3 .. 1.8e-2 from its fp64 run at this size); no hand-set floor, no outlier list
    rep = G.check_grads_env(g, "seg96", "seg", [(n, p.grad) for n, p in seg.named_parameters()], floor=RTOL_GRAD_FP32, what="seg96")
    G.envelope_summary(rep, "seg96")


def test_vae64_train_vs_golden():
    M, O, T = _mods()
    g = G.load("vae64_train")
    vae = _fill(M.VAE(2, 2, norm_type=1, dim=128, spatial=64), 0, O)
    noise = torch.from_numpy(2 * O.hashed_uniform(2 * 128, 7100, 5) - 1).view(2, 128)
    final, aux = T.vae_train_losses(vae, O.synthetic_label(2, 64, 3).cuda(), scale=0.35, noise=noise.cuda())
    final.backward()
    G.scalar_close(g, "final", final.item(), RTOL_FP32)
    G.scalar_close(g, "kl", aux["kl_loss"].item(), RTOL_FP32)
    b = aux["batch"]
    assert G.rel_l2(b["mean"].detach().cpu(), g["mean@f64"]) < max(RTOL_FP32, 3 * G.rel_l2(g["mean"], g["mean@f64"]))
    assert G.rel_l2(b["std"].detach().cpu(), g["std@f64"]) < max(RTOL_FP32, 3 * G.rel_l2(g["std"], g["std@f64"]))
    G.check_tensor_f64(g, "recon", b["recon"], k=256, floor=RTOL_FP32)
    G.check_grads_f64(g, "vae", [(n, p.grad) for n, p in vae.named_parameters()], floor=RTOL_GRAD_FP32)


def _build_joint(M, O, side):
    seg = M.Segmentation(n_channels=1, n_class=2, norm_type=1)
    vae = M.VAE(n_channels=2, n_class=2, norm_type=1, dim=128, spatial=side)
    joint = M.Joint(models=[seg, vae])
    O.deterministic_fill_(joint, seed=0)
    joint = joint.cuda()
    for p in joint.Vae.parameters():
        p.requires_grad = False
    joint.Vae.eval()
    return joint


@pytest.mark.parametrize("side,bs,name", [(64, 2, "joint64"), (96, 2, "joint96"), (128, 1, "joint128")])
def test_joint_train_step_vs_reference_golden(side, bs, name):
    """BASELINE configs[1] (96^3, B=2) and the reference-native 128^3 case, fp32 kernels."""
    M, O, T = _mods()
    g = G.load(name)
    joint = _build_joint(M, O, side)
    final, aux = T.joint_train_losses(joint, O.synthetic_image(bs, side, 2).cuda(), O.synthetic_label(bs, side, 3).cuda())
    final.backward()
    torch.cuda.synchronize()
    for key, val in (("final", final), ("recon_loss", aux["recon_loss"]), ("dice_loss", aux["dice_loss"])):
        G.scalar_close(g, key, val.item(), RTOL_FP32)
    b = aux["batch"]
    assert G.rel_l2(b["mean"].detach().cpu(), g["mean@f64"]) < max(RTOL_FP32, 3 * G.rel_l2(g["mean"], g["mean@f64"]))
    assert G.rel_l2(b["std"].detach().cpu(), g["std@f64"]) < max(RTOL_FP32, 3 * G.rel_l2(g["std"], g["std@f64"]))
    # every size: held to the measured envelope of the reference's own fp32 arithmetic (tests/golden/envelopes.npz: joint96, round 5;
    # envelopes2.npz: joint64 and joint128, round 6) — ONE HIP run, every tensor, 1.5 x, no outlier list
    G.check_tensor_env(g, name, "pred", b["pred"], k=512, floor=RTOL_FP32)
    G.check_tensor_env(g, name, "recon", b["recon"], k=512, floor=RTOL_FP32)
    rep = G.check_grads_env(g, name, "seg", [(n, p.grad) for n, p in joint.Seg.named_parameters()], floor=RTOL_GRAD_FP32, what=name)
    G.envelope_summary(rep, name)
    assert all(p.grad is None for p in joint.Vae.parameters())


def test_domain_adaptation128_vs_reference_golden():
    M, O, T = _mods()
    g = G.load("da128")
    student, teacher = _build_joint(M, O, 128), _build_joint(M, O, 128)
    O.deterministic_fill_(teacher.Seg, seed=1)
    for p in teacher.parameters():
        p.requires_grad = False
    img, lab = O.synthetic_image(1, 128, 2).cuda(), O.synthetic_label(1, 128, 3).cuda()
    final, aux = T.domain_adaptation_losses(student, teacher, img, lab, lambda_vae=1.0, domain_loss_type=0)
    final.backward()
    G.scalar_close(g, "final0", final.item(), RTOL_FP32)
    for k_o, k_g in (("recon_loss", "recon_loss"), ("kl_loss", "kl"), ("dice_loss", "dice_loss"), ("dice_loss_fake", "fake_loss")):
        G.scalar_close(g, k_g, aux[k_o].item(), RTOL_FP32)
    # pseudo-label voxels may flip only where the teacher's soft output is within rounding of 0.5
    fake_sum = aux["batch"]["fake"].double().sum().item()
    assert abs(fake_sum - float(g["fake.sum"])) <= 4
    G.envelope_summary(G.check_grads_env(g, "da128", "seg", [(n, p.grad) for n, p in student.Seg.named_parameters()], floor=RTOL_GRAD_FP32, what="da128 type 0"), "da128 type 0")
    # domain_loss_type 8 (main_target.py:550-560): loss and its own gradient set; evaluated on the host (as the reference) and on the device
    for host in (True, False):
        for p in student.Seg.parameters():
            p.grad = None
        f8, _ = T.domain_adaptation_losses(student, teacher, img, lab, lambda_vae=1.0, domain_loss_type=8, host_schedule=host)
        f8.backward()
        G.scalar_close(g, "final8", f8.item(), RTOL_FP32)
        G.envelope_summary(G.check_grads_env(g, "da128", "seg8", [(n, p.grad) for n, p in student.Seg.named_parameters()], floor=RTOL_GRAD_FP32, what="da128 type 8"), "da128 type 8")
    # the other scalar combinations of main_target.py:571-592 against the same arithmetic on the golden's terms
    r, f = float(g["recon_loss@f64"]), float(g["fake_loss@f64"])
    for kw, want in ((dict(domain_loss_type=11), r + f + r * f), (dict(domain_loss_type=12), r + f - r * f),
                     (dict(domain_loss_type=13), max(r - 0.15, 0.0)), (dict(domain_loss_type=14), max(r - 0.1, 0.0) + f),
                     (dict(only_pseudo=True), f), (dict(turn_epoch=2, epoch=1), r), (dict(turn_epoch=2, epoch=2), r + f),
                     (dict(lambda_vae_warmup=4, epoch=1), 0.25 * r + f), (dict(lambda_vae_warmup=4, epoch=4), r + f)):
        with torch.no_grad():
            v, _ = T.domain_adaptation_losses(student, teacher, img, lab, lambda_vae=1.0, **kw)
        assert abs(v.item() - want) <= 2e-3 * abs(want), (kw, v.item(), want)


@pytest.mark.parametrize("graph", [False, True])
def test_test_time_finetune128_vs_reference_golden(graph):
    """per-case test-time training (main_target.py:809-953), eager and HIP-graph replayed, against the loop
    run on the reference's modules: per-iteration losses, the accumulated weight update (= two gradients, the second taken on
    updated weights) and the hard-Dice scores with / without finetuning."""
    M, O, T = _mods()
    from vae_segmentation_amd import ops
    g = G.load("ft128")
    img, lab = O.synthetic_image(1, 128, 2).cuda(), O.synthetic_label(1, 128, 3).cuda()
    draws = []
    # The accumulated update of two chained steps is the most amplified quantity of the suite (the reference's own fp32 run sits 1.4 from its fp64 run in the median
    # tensor: tests/golden/envelopes2.npz), and the envelope is a 12-run maximum a 13th run of the same arithmetic exceeds with probability 1/13 per tensor: as for
    # embed128 the gate takes the MEDIAN of three HIP runs, each with the base model's weights moved by +-1 ulp exactly as the envelope's runs were (seed 0: unmoved);
    # losses, scores and the prediction are gated on the unmoved run.
    for seed in (0, 1, 2):
        model, model_ft, teacher = _build_joint(M, O, 128), _build_joint(M, O, 128), _build_joint(M, O, 128)
        G.perturb_ulp_(model, seed)
        O.deterministic_fill_(teacher.Seg, seed=1)
        for p in teacher.parameters():
            p.requires_grad = False
        ops.weights_changed()
        runner = T.TestTimeFinetune(model, model_ft, teacher, 128, steps=2, lr=1e-2, lambda_vae=1.0, domain_loss_type=8, graph=graph)
        for rep in range(2 if (graph and seed == 0) else 1):            # a second case on the same runner must start from model's weights again
            log, score_noft, score, pred = runner.run(img, lab)
            ref = dict(model.Seg.named_parameters())
            upd = [(n, (p.detach() - ref[n].detach()) / 1e-2) for n, p in model_ft.Seg.named_parameters()]
            if seed == 0:
                for it, rec in enumerate(log):
                    for k_o, k_g in (("recon_loss", "recon_loss"), ("dice_loss", "dice_loss"), ("dice_loss_fake", "fake_loss"), ("final_loss", "final")):
                        G.scalar_close(g, "it%d.%s" % (it, k_g), rec[k_o].item(), RTOL_FP32)
                # hard Dice counts argmax voxels: a handful may flip where the two probabilities are within rounding of each other
                assert abs(score_noft.item() - float(g["score_noft@f64"])) < 2e-3
                assert abs(score.item() - float(g["score@f64"])) < 2e-3
                G.check_tensor_env(g, "ft128", "pred", pred, k=512, floor=RTOL_FP32)
                if rep == 0:
                    first_case = G.grads_dist(g, "upd", upd, what="ft128")
                else:
                    assert G.grads_dist(g, "upd", upd, what="ft128") == first_case, "the second case on the same runner did not start from the model's weights"
        draws.append(G.grads_dist(g, "upd", upd, what="ft128"))
        del runner, model, model_ft, teacher
    G.envelope_summary(G.check_grads_env(g, "ft128", "upd", None, floor=RTOL_GRAD_FP32, what="ft128", draws=draws), "ft128 (%s)" % ("graph" if graph else "eager"))


def test_vae128_native_shapes_vs_reference_golden():
    M, O, T = _mods()
    g = G.load("vae128_train")
    vae = _fill(M.VAE(2, 2, norm_type=1, dim=128, spatial=128), 0, O)
    final, aux = T.vae_train_losses(vae, O.synthetic_label(1, 128, 3).cuda(), scale=0.35, noise=torch.from_numpy(g["z"]).cuda())
    final.backward()
    G.scalar_close(g, "final", final.item(), RTOL_FP32)
    G.check_tensor_env(g, "vae128_train", "recon", aux["batch"]["recon"], k=512, floor=RTOL_FP32)
    G.envelope_summary(G.check_grads_env(g, "vae128_train", "vae", [(n, p.grad) for n, p in vae.named_parameters()], floor=RTOL_GRAD_FP32, what="vae128_train"), "vae128_train")


def test_bf16_mode_joint96_close_to_fp32_reference():
    """Throughput mode (bf16 storage, fp32 accumulate) on BASELINE configs[1].  Forward: loss scalars within 2 %,
    probabilities within 3e-2 mean / 0.2 max absolute of the fp64 yardstick.  Backward: at this random init the network is chaotic
    (see RTOL_GRAD_FP32 above: even fp32 rounding moves early-layer gradients by 10 %), so bf16 gradients are checked
    where the comparison is meaningful — the layers nearest the loss (out_block, up5) must point the same way as the
    fp64 gradient (cosine > 0.9) — and per-op bf16 backward accuracy is covered by tests/test_gpu_ops.py."""
    M, O, T = _mods()
    g = G.load("joint96")
    joint = _build_joint(M, O, 96)
    M.set_kernel_dtype(joint, torch.bfloat16)
    final, aux = T.joint_train_losses(joint, O.synthetic_image(2, 96, 2).cuda(), O.synthetic_label(2, 96, 3).cuda())
    final.backward()
    f64 = float(g["final@f64"])
    assert abs(final.item() - f64) / f64 < 2e-2
    assert abs(aux["recon_loss"].item() - float(g["recon_loss@f64"])) / float(g["recon_loss@f64"]) < 5e-2
    pred = G.flat64(aux["batch"]["pred"])
    ps = pred[G.sample_idx(pred.size, 512)]
    perr = np.abs(ps - g["pred.samples@f64"])
    assert perr.mean() < 3e-2 and perr.max() < 0.2, (perr.mean(), perr.max())     # 30 chaotic layers deep: see docstring
    cos = {}
    for name, p in joint.Seg.named_parameters():
        key = "seg.grad.%s" % name
        if G.is_dead_bias(name) or p.numel() < 8:
            continue
        a = G.flat64(p.grad)[G.sample_idx(p.numel(), 16)]
        b = g[key + ".samples@f64"].astype(np.float64)
        cos[name] = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
    print("\nbf16 gradient cosines vs fp64:", {k: round(v, 3) for k, v in cos.items()})
    near_loss = [v for k, v in cos.items() if k.startswith("out_block") or k.startswith("up5.conv.1.conv.6")]
    assert min(near_loss) > 0.9, near_loss
    assert all(np.isfinite(G.flat64(p.grad)).all() for p in joint.Seg.parameters())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_16bit_joint128_weight_gradients_same_with_and_without_m_packing(dtype, monkeypatch):
    """The 128^3 joint step: the weight gradients of the layers with 8 stored output channels take the M-packed form by default (csrc/wgrad.hip g3b_body);
    every gradient of the default plan equals the unpacked plan's (VS_WGRAD_MPACK=0) up to the order of the fp32 sums, and the forced-on plan is the
    default plan bit for bit."""
    M, O, T = _mods()
    from vae_segmentation_amd import ops
    # the layers under test stay in the grouped launches: no weight gradient inside a backward-data launch (csrc/igemm_k3tw.h takes in_block.conv.3, up5's second
    # conv and out_block by default)
    monkeypatch.setattr(ops, "FUSE_WGRAD", False)
    monkeypatch.setattr(ops, "FUSE_SOFTMAX_BWD", False)
    img, lab = O.synthetic_image(1, 128, 2).cuda(), O.synthetic_label(1, 128, 3).cuda()
    grads = {}
    for mode in ("default", "1", "0"):
        ops.set_config(wgrad_mpack=1 if mode == "default" else int(mode))        # (the library's default is 1; restored below)
        joint = _build_joint(M, O, 128)
        M.set_kernel_dtype(joint, dtype)
        final, _ = T.joint_train_losses(joint, img, lab)
        final.backward()
        torch.cuda.synchronize()
        grads[mode] = {n: p.grad.detach().float().cpu() for n, p in joint.Seg.named_parameters()}
    ops.set_config(wgrad_mpack=1)
    for n, g1 in grads["1"].items():
        assert torch.equal(grads["default"][n], g1), n
        g0 = grads["0"][n]
        if float(g0.norm()) > 0:
            assert G.rel_l2(g1, g0) < 2e-5, (n, G.rel_l2(g1, g0))
    packed = [n for n in grads["1"] if not torch.equal(grads["1"][n], grads["0"][n])]
    assert any(n.startswith("in_block") for n in packed) and any(n.startswith("out_block") for n in packed), packed   # the packed layers do differ in the last bits


def test_bf16_joint_step_same_with_and_without_the_channels_last_prediction(monkeypatch):
    """Joint.forward (joint_model.py:447-450) feeds Segmentation's prediction to the VAE.  In bf16 mode out_block writes the channels-last copy
    the VAE reads and the softmax backward takes the two gradient parts itself; VS_SOFTMAX_CL=0 spells it with vs_pack_planar,
    vs_unpack_planar and autograd's add.  Same arithmetic: the step must not change (statistics atomics reorder, hence a tolerance)."""
    M, O, T = _mods()
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("VS_SOFTMAX_CL", flag)
        joint = _build_joint(M, O, 64)
        M.set_kernel_dtype(joint, torch.bfloat16)
        final, aux = T.joint_train_losses(joint, O.synthetic_image(2, 64, 2).cuda(), O.synthetic_label(2, 64, 3).cuda())
        final.backward()
        torch.cuda.synchronize()
        assert (getattr(aux["batch"]["pred"], "_vs_cl", None) is not None) == (flag == "1")
        out[flag] = (final.item(), aux["recon_loss"].item(), aux["batch"]["recon"].detach().float().cpu(),
                     {n: p.grad.detach().float().cpu() for n, p in joint.Seg.named_parameters()})
    assert abs(out["1"][0] - out["0"][0]) < 1e-5 and abs(out["1"][1] - out["0"][1]) < 1e-5
    assert float((out["1"][2] - out["0"][2]).abs().max()) < 1e-3
    for n in out["1"][3]:
        a, b = out["1"][3][n], out["0"][3][n]
        if G.is_dead_bias(n):
            continue
        assert float((a - b).norm() / b.norm().clamp_min(1e-20)) < 2e-2, n


@pytest.mark.parametrize("tail", [True, False])
def test_sgd_step_and_graph_replay_match_eager(tail):
    """Three SGD(momentum) steps: native multi-tensor kernel vs torch.optim.SGD on the oracle (CPU), then a
    HIP-graph replayed step against the eager step — with the tail of the step (SGD launch, weight re-pack) inside the graph and outside it."""
    M, O, T = _mods()
    from vae_segmentation_amd import optim
    side, bs = 32, 2
    seg = _fill(M.Segmentation(1, 2, norm_type=1), 0, O)
    oseg = O.deterministic_fill_(O.Segmentation(1, 2, norm_type=1), seed=0)
    img, lab = O.synthetic_image(bs, side, 2), O.synthetic_label(bs, side, 3)
    opt = optim.SGD(seg.parameters(), lr=1e-2, momentum=0.9)
    oopt = torch.optim.SGD(oseg.parameters(), lr=1e-2, momentum=0.9)
    for _ in range(3):
        opt.zero_grad(); oopt.zero_grad()
        l, _ = T.seg_train_losses(seg, img.cuda(), lab.cuda())
        ol, _ = O.seg_train_losses(oseg, img, lab)
        l.backward(); ol.backward()
        opt.step(); oopt.step()
        assert abs(l.item() - ol.item()) / ol.item() < RTOL_FP32
    for (n1, p1), (_, p2) in zip(seg.named_parameters(), oseg.named_parameters()):
        assert G.rel_l2(p1.detach().cpu(), p2.detach()) < RTOL_FP32, n1
    # graph replay
    seg_a = _fill(M.Segmentation(1, 2, norm_type=1), 0, O)
    seg_b = _fill(M.Segmentation(1, 2, norm_type=1), 0, O)
    ig, lg = img.cuda(), lab.cuda()
    opt_a = optim.SGD(seg_a.parameters(), lr=1e-2, momentum=0.9)
    opt_b = optim.SGD(seg_b.parameters(), lr=1e-2, momentum=0.9)
    gs = T.GraphedStep(lambda: T.seg_train_losses(seg_b, ig, lg), seg_b.parameters(), opt_b, warmup=1, capture_tail=tail)
    assert gs.tail is tail
    for _ in range(2):
        opt_a.zero_grad()
        la, _ = T.seg_train_losses(seg_a, ig, lg)
        la.backward()
        opt_a.step()
        lb = gs.step()
        assert abs(la.item() - lb.item()) < 1e-5
    for (n1, p1), (_, p2) in zip(seg_a.named_parameters(), seg_b.named_parameters()):
        assert G.rel_l2(p1.detach().cpu(), p2.detach().cpu()) < 1e-4, n1


# ---- Encoder / Fusion / Joint2 / Embed on the native kernels (goldens: oracle/make_golden.py gold_rank4) ----
def _dsc_main_source(s, t, bot, top, eps=1e-4):
    d = 2 * torch.sum(s * t, (2, 3, 4)) / (torch.sum(s, (2, 3, 4)) + torch.sum(t, (2, 3, 4)) + eps)
    return torch.mean(d[:, bot:top])


def test_encoder128_vs_reference_golden():
    M, O, T = _mods()
    g = G.load("enc128")
    enc = _fill(M.Encoder(1, 1, norm_type=1), 4, O)
    assert [tuple(p.shape) for p in (enc.fc1.weight, enc.fc2.weight, enc.fc_mean.weight)] == [(1024, 16384), (128, 1024), (1, 128)]
    x = O.synthetic_image(1, 128, seed=5).abs().cuda().requires_grad_(True)
    out = enc(x)
    out.sum().backward()
    G.scalar_close(g, "out", out.item(), RTOL_FP32)
    # One voxel of channel 47 of down4's first 3x3x3 conv has a normalised pre-activation of 9.4e-7 in the fp64 run — inside fp32
    # rounding of the ReLU threshold.  The native fp32 sum order lands it on the other side (the reference's fp32 run does not),
    # so that voxel's mask differs: measured, it is the ONLY element of that 128 x 8^3 gradient off by more than 4e-8, and it moves
    # every gradient upstream of it by ~1e-2 (one of 512 voxels).  Everything between the loss and that voxel is checked at the
    # usual floor; the tensors upstream of it, and dL/dx, at 3e-2.
    named = [(n, p.grad) for n, p in enc.named_parameters()]
    tight = [(n, gr) for n, gr in named if n.startswith(("fc", "down5.", "down4.conv.1.conv.3", "down4.conv.1.conv.6"))]
    loose = [(n, gr) for n, gr in named if (n, gr) not in tight]
    G.check_grads_f64(g, "enc", tight, floor=RTOL_GRAD_FP32)
    G.check_grads_f64(g, "enc", loose, floor=3e-2)
    G.check_tensor_f64(g, "gx", x.grad, k=512, floor=3e-2, factor=8.0)


def test_fusion64_vs_reference_golden():
    M, O, T = _mods()
    from vae_segmentation_amd.evaluation import avg_dsc
    g = G.load("fusion64")
    fus = _fill(M.Fusion(1, 2, 2, norm_type=1), 6, O)
    img, gt = O.synthetic_image(1, 64, seed=2).cuda(), O.one_hot(O.synthetic_label(1, 64, seed=3)).cuda()
    mask = O.one_hot(O.synthetic_label(1, 64, seed=7)).cuda().requires_grad_(True)
    batch = fus({"img": img, "mask": mask, "gt": gt}, "img", "mask", "pred")
    loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=2, eps=1e-4)
    loss.backward()
    G.scalar_close(g, "loss", loss.item(), RTOL_FP32)
    G.check_tensor_f64(g, "pred", batch["pred"], k=512, floor=RTOL_FP32)
    G.check_tensor_f64(g, "gmask", mask.grad, k=512, floor=RTOL_GRAD_FP32, factor=8.0)
    G.check_grads_f64(g, "fus", [(n, p.grad) for n, p in fus.named_parameters()], floor=RTOL_GRAD_FP32)


def test_joint2_is_segmentation_then_discriminator():
    M, O, T = _mods()
    seg, dis = _fill(M.Segmentation(1, 2, norm_type=1), 0, O), _fill(M.Encoder(1, 1, norm_type=1), 4, O)
    j2 = M.Joint2(models=[seg, dis])
    assert set(k.split(".")[0] for k in j2.state_dict()) == {"Seg", "Dis"}
    img = O.synthetic_image(1, 128, seed=2).cuda()
    batch = j2({"img": img}, "img", "pred", "score")
    assert batch["score"].shape == (1, 1) and 0.0 < batch["score"].item() < 1.0
    ref = dis(seg({"img": img}, "img", "p")["p"][:, 1:2])
    assert abs(ref.item() - batch["score"].item()) < 1e-6


def test_embed128_vs_reference_golden():
    M, O, T = _mods()
    from vae_segmentation_amd import ops
    from vae_segmentation_amd.evaluation import avg_dsc
    g = G.load("embed128")
    img, gt = O.synthetic_image(1, 128, seed=2).cuda(), O.one_hot(O.synthetic_label(1, 128, seed=3)).cuda()
    draws = {"enc": [], "vae": [], "fus": []}
    for seed in (0, 1, 2):          # the candidate's own draws: its weights moved by +-1 ulp exactly as oracle/make_golden.py moves the reference's (seed 0: unmoved)
        emb = M.Embed(models=[M.Encoder(1, 128, norm_type=1), M.VAE(2, 2, norm_type=1, dim=128), M.Fusion(1, 2, 2, norm_type=1)])
        O.deterministic_fill_(emb, seed=8)
        G.perturb_ulp_(emb, seed)
        emb = emb.cuda()
        batch = emb({"img": img, "venous_pancreas_only": gt, "gt": gt}, "img", "pred", noise=torch.from_numpy(g["z"]).cuda())
        dsc = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=2, eps=1e-4)
        lat = torch.mean((batch["latent_code"] - batch["latent_code_gt"].detach()) ** 2)
        (dsc + lat).backward()
        if seed == 0:
            G.scalar_close(g, "dice_loss", dsc.item(), RTOL_FP32)
            G.scalar_close(g, "latent_loss", lat.item(), RTOL_FP32)
        # three networks deep (Encoder -> VAE -> Fusion, ~90 InstanceNorm/ReLU layers at 128^3).  Outputs: EVERY run within 1.5 x the measured envelope of the
        # reference's own fp32 arithmetic (tests/golden/envelopes.npz); no hand-set floor (round 4 used 2.5e-2 on the gradients and a factor 4 on `pred`)
        for k in ("pred", "gt_recon", "init_seg", "seg_recon"):
            G.check_tensor_env(g, "embed128", k, batch[k], k=512, floor=RTOL_FP32)
        for pre, mod in (("enc", emb.Encoder), ("vae", emb.Vae), ("fus", emb.Fusion)):
            draws[pre].append(G.grads_dist(g, pre, [(n, p.grad) for n, p in mod.named_parameters()], what="embed128 " + pre))
        del emb, batch, dsc, lat
        ops.drop_stale_wgrads()
    # gradients (145 live tensors): the median of the three runs, tensor by tensor, within 1.5 x the envelope (golden_util.check_grads_env says why not one run)
    for pre in ("enc", "vae", "fus"):
        G.envelope_summary(G.check_grads_env(g, "embed128", pre, None, floor=RTOL_GRAD_FP32, what="embed128 " + pre, draws=draws[pre],
                                             exceed=1 if pre == "fus" else 0), "embed128 " + pre)


def test_seg32_dropout_with_exported_masks_vs_oracle(monkeypatch):
    """The F.dropout sites of Segmentation.forward (joint_model.py:379-388) with dropout = 0.2: the masks the kernels drew (counter hash of
    seed and element index, exported by vs_dropout_mask) are fed to the oracle, whose F.dropout is replaced by a multiply with them; forward
    and every parameter gradient then have to agree like any other fp32-mode result (fp64 yardstick: the oracle also runs in fp64)."""
    M, O, T = _mods()
    from vae_segmentation_amd import ops
    seeds, orig = [], ops.next_dropout_seed

    def recording_seed():
        s = orig()
        seeds.append(s)
        return s
    monkeypatch.setattr(ops, "next_dropout_seed", recording_seed)
    # the masks are a pure function of (torch's seed, the process-wide dropout call counter): pin both, or the draw — and with it which activations
    # sit on a ReLU edge under a kept mask element — depends on how many dropout sites earlier tests of the process happened to run
    torch.manual_seed(1234)
    monkeypatch.setattr(ops, "_DROPOUT_CALLS", [1000])
    p, side, bs = 0.2, 32, 2
    seg = _fill(M.Segmentation(1, 2, norm_type=1), 0, O)
    img, lab = O.synthetic_image(bs, side, 2), O.synthetic_label(bs, side, 3)
    batch = seg({"img": img.cuda(), "gt": ops.onehot(lab.cuda(), 2)}, "img", "pred", dropout=p)
    from vae_segmentation_amd.evaluation import avg_dsc
    loss = 1 - avg_dsc(batch, "pred", "gt", botindex=1, topindex=2, eps=1e-6)
    loss.backward()
    torch.cuda.synchronize()
    assert len(seeds) == 5
    shapes = [(bs, 4, 4, 4, 64), (bs, 8, 8, 8, 32), (bs, 16, 16, 16, 16), (bs, 32, 32, 32, 8)]          # after up2, up3 + x3, up4 + x2, up5
    masks = [ops.dropout_mask(int(np.prod(s)), p, sd).view(s).permute(0, 4, 1, 2, 3).contiguous().cpu() for s, sd in zip(shapes, seeds[:4])]
    masks.append(ops.dropout_mask(bs * 2 * side ** 3, p, seeds[4]).view(bs, 2, side, side, side).cpu())       # the two logits, planar order
    for m in masks:
        vals = set(np.unique(m.numpy()).round(5).tolist())
        assert vals <= {0.0, round(1 / (1 - p), 5)}
        assert abs(float((m == 0).float().mean()) - p) < 0.03
    res = {}
    for dt in (torch.float64, torch.float32):
        queue = [m.to(dt) for m in masks]
        monkeypatch.setattr(O, "_maybe_dropout", lambda x, pp: x * queue.pop(0) if pp else x)
        oseg = O.deterministic_fill_(O.Segmentation(1, 2, norm_type=1), seed=0).to(dt)
        ob = oseg({"img": img.to(dt), "gt": O.one_hot(lab, 2).to(dt)}, "img", "pred", dropout=p)
        ol = 1 - O.avg_dsc(ob, "pred", "gt", botindex=1, topindex=2, eps=1e-6)
        ol.backward()
        assert not queue
        res[dt] = (ol.item(), ob["pred"].detach().double(), {n: q.grad.double() for n, q in oseg.named_parameters()})
    l64, p64, g64 = res[torch.float64]
    l32, p32, g32 = res[torch.float32]
    assert abs(loss.item() - l64) <= max(1e-3 * abs(l64), 3 * abs(l32 - l64))
    pred = batch["pred"].detach().double().cpu()
    assert float((pred - p64).abs().max()) <= max(1e-3, 3 * float((p32 - p64).abs().max()))
    over = 0
    for n, prm in seg.named_parameters():
        if G.is_dead_bias(n):
            assert float(prm.grad.norm()) < 1e-4
            continue
        mine = float((prm.grad.double().cpu() - g64[n]).norm() / g64[n].norm())
        theirs = float((g32[n] - g64[n]).norm() / g64[n].norm())
        # the bias of a stride-2 / transposed conv in front of an InstanceNorm'd DoubleConv acts only through the zero padding of the
        # next 3x3x3 conv: its exact gradient is a near-cancelling sum over the voxels (|sum| ~ 1e-3 of sum |.|), so fp32 rounding of the
        # terms shows up amplified ~1e3 times in the relative error — floor 1e-2 for these 8 vectors
        # weights: the 2e-3 floor.  In the default (fp64-atomic) mode this gate was unreliable — over 28 runs 21 within 2e-3, 6 between 2.5e-3 and
        # 3.4e-3, one at 6.4e-3: the statistics atomics arrive in a different order each run, and at 32^3 a single activation that sits at a
        # ReLU edge and is kept (x 1.25) by the dropout mask moves the deepest gradients by that much when it flips.  The test suite runs in the
        # library's deterministic mode (tests/conftest.py): one fixed summation, the same numbers every run.
        floor = 2e-3
        lim = max(floor, 8 * theirs)
        over += lim > 1e-2
        assert mine <= lim, (n, mine, lim, theirs)
    print("\nseg32 dropout: %d gradient tensors had a limit above 1e-2" % over)


# ---- remaining train methods on the native modules (goldens: oracle/make_golden.py gold_methods) ---------------------------------------
def _embed_native(M, O):
    emb = M.Embed(models=[M.Encoder(1, 128, norm_type=1), M.VAE(2, 2, norm_type=1, dim=128), M.Fusion(1, 2, 2, norm_type=1)])
    O.deterministic_fill_(emb, seed=8)
    return emb.cuda()


def test_embed_train_and_refine_vae128_vs_reference_golden():
    """main_source.py:546-628: the embed_train and refine_vae loss bodies (train.embed_train_losses / refine_vae_losses) on the native Embed."""
    M, O, T = _mods()
    g = G.load("embed_train128")
    emb = _embed_native(M, O)
    img, lab = O.synthetic_image(1, 128, seed=2).cuda(), O.synthetic_label(1, 128, seed=3).cuda()
    z = torch.from_numpy(g["z"]).cuda()
    final, aux = T.embed_train_losses(emb, img, lab, noise=z)
    final.backward()
    for k_o, k_g in (("dice_loss1", "dice_loss1"), ("dice_loss2", "dice_loss2"), ("recon_loss", "recon_loss"), ("inpaint_loss", "inpaint_loss"),
                     ("kl_loss", "kl"), ("mse_loss", "mse")):
        G.scalar_close(g, k_g, aux[k_o].item(), RTOL_FP32)
    G.scalar_close(g, "final", final.item(), RTOL_FP32)
    for pre, mod in (("enc", emb.Encoder), ("vae", emb.Vae), ("fus", emb.Fusion)):
        G.vacuity(G.check_grads_f64(g, pre, [(n, p.grad) for n, p in mod.named_parameters()], floor=RTOL_GRAD_FP32), "embed_train " + pre)
    emb2 = _embed_native(M, O)
    for p in emb2.Encoder.parameters():
        p.requires_grad = False
    rfinal, _ = T.refine_vae_losses(emb2, img, lab, noise=z)
    rfinal.backward()
    G.scalar_close(g, "refine_final", rfinal.item(), RTOL_FP32)
    G.check_grads_f64(g, "rvae", [(n, p.grad) for n, p in emb2.Vae.named_parameters()], floor=RTOL_GRAD_FP32)
    G.check_grads_f64(g, "rfus", [(n, p.grad) for n, p in emb2.Fusion.named_parameters()], floor=RTOL_GRAD_FP32)
    assert all(p.grad is None for p in emb2.Encoder.parameters())


def test_sep_joint_train128_vs_reference_golden():
    """main_source.py:629-659: per-sample Dice scores, the teacher's squared reconstruction score as weight."""
    M, O, T = _mods()
    g = G.load("sep_joint128")
    student, teacher = _build_joint(M, O, 128), _build_joint(M, O, 128)
    O.deterministic_fill_(teacher.Seg, seed=1)
    for p in teacher.parameters():
        p.requires_grad = False
    final, aux = T.sep_joint_train_losses(student, teacher, O.synthetic_image(1, 128, 2).cuda(), O.synthetic_label(1, 128, 3).cuda())
    final.backward()
    G.scalar_close(g, "final", final.item(), RTOL_FP32)
    G.vacuity(G.check_grads_f64(g, "seg", [(n, p.grad) for n, p in student.Seg.named_parameters()], floor=RTOL_GRAD_FP32), "sep_joint128")


def test_domain_adaptation_dis_and_discriminator_train128_vs_reference_golden():
    """main_target.py:696-732 (Joint2 student, frozen discriminator and pseudo-label teacher) and :491-501 (score regression)."""
    M, O, T = _mods()
    g = G.load("da_dis128")
    seg, dis = _fill(M.Segmentation(1, 2, norm_type=1), 0, O), _fill(M.Encoder(1, 1, norm_type=1), 4, O)
    j2 = M.Joint2(models=[seg, dis])
    for p in j2.Dis.parameters():
        p.requires_grad = False
    teacher = _fill(M.Segmentation(1, 2, norm_type=1), 1, O)
    for p in teacher.parameters():
        p.requires_grad = False
    img, lab = O.synthetic_image(1, 128, 2).cuda(), O.synthetic_label(1, 128, 3).cuda()
    final, aux = T.domain_adaptation_dis_losses(j2, teacher, img, lab, lambda_vae=1.0, epoch=1, lambda_vae_warmup=0)
    final.backward()
    G.scalar_close(g, "final", final.item(), RTOL_FP32)
    for k_o, k_g in (("dice_loss", "dice_loss"), ("dice_loss_fake", "fake_loss"), ("discriminator_loss", "dis_loss")):
        G.scalar_close(g, k_g, aux[k_o].item(), RTOL_FP32)
    G.vacuity(G.check_grads_f64(g, "seg", [(n, p.grad) for n, p in j2.Seg.named_parameters()], floor=RTOL_GRAD_FP32), "da_dis128")
    dis2 = _fill(M.Encoder(1, 1, norm_type=1), 4, O)
    dl, daux = T.discriminator_train_loss(dis2, lab.float(), torch.tensor([[0.7]]).cuda())
    dl.backward()
    G.scalar_close(g, "dtrain_loss", dl.item(), RTOL_FP32)
    tight = [(n, p.grad) for n, p in dis2.named_parameters() if n.startswith(("fc", "down5."))]
    loose = [(n, p.grad) for n, p in dis2.named_parameters() if not n.startswith(("fc", "down5."))]
    G.check_grads_f64(g, "dis", tight, floor=RTOL_GRAD_FP32)
    G.check_grads_f64(g, "dis", loose, floor=3e-2)          # see test_encoder128: one ReLU-threshold voxel can move the upstream gradients ~1e-2
