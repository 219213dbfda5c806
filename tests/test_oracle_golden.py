"""CPU: the oracle (oracle/ref_cpu.py) reproduces what the unmodified reference produced
(tests/golden/*.npz, written by oracle/make_golden.py in the build container)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from tests import golden_util as G

torch.set_num_threads(8)


def test_kats():
    g = G.load("kats")
    assert np.allclose(O.KLloss({"mean": torch.zeros(2, 3), "std": torch.ones(2, 3)}).numpy(), g["kl1"], rtol=0, atol=0)
    b = {"mean": torch.tensor([[1.0, -2.0, 0.5]]), "std": torch.tensor([[0.0, 2.0, 0.5]])}
    assert np.allclose(O.KLloss(b).numpy(), g["kl2"], rtol=1e-7)
    s1 = torch.tensor([.9, .1, .8, .2, .7, .3, .6, .4]).view(1, 1, 2, 2, 2)
    t1 = torch.tensor([1., 0, 1, 0, 0, 1, 1, 0]).view(1, 1, 2, 2, 2)
    b = {"s": torch.cat((1 - s1, s1), 1), "t": torch.cat((1 - t1, t1), 1)}
    assert O.avg_dsc(b, "s", "t", botindex=1, topindex=2).item() == pytest.approx(float(g["dice1"]), rel=1e-7)
    assert O.avg_dsc(b, "s", "t", botindex=0, topindex=2).item() == pytest.approx(float(g["dice1_all"]), rel=1e-7)
    assert np.allclose(O.avg_dsc(b, "s", "t", botindex=1, topindex=2, return_mean=False).numpy(), g["dice1_nomean"])
    assert O.avg_dsc(b, "s", "t", botindex=1, topindex=2, binary=True).item() == pytest.approx(float(g["dice2_binary"]), rel=1e-7)
    assert O.avg_dsc(b, "s", "t", botindex=1, topindex=2, eps=1e-4).item() == pytest.approx(float(g["dice3_eps1e4"]), rel=1e-7)
    assert O.dice(b["s"], b["t"]).item() == pytest.approx(float(g["dice_fn"]), rel=1e-7)
    assert np.array_equal(O.binarize(torch.tensor([.49, .5, .81])).numpy(), g["bin"])
    assert np.array_equal(O.confident_binarize(torch.tensor([.1, .2, .5, .8, .81])).numpy(), g["cbin"])
    x = torch.arange(16.).view(1, 2, 2, 2, 2)
    assert np.allclose(torch.relu(O.norm_layer(1, 2)(x)).reshape(-1).numpy(), g["in_relu"], rtol=1e-6)
    # the values SURVEY.md §4 recorded from the reference
    assert float(g["kl1"]) == pytest.approx(1.4999699592590332, rel=1e-7)
    assert float(g["dice1"]) == pytest.approx(0.6499999165534973, rel=1e-7)
    p = torch.tensor([.9, .2, .6, .4]).view(1, 1, 1, 2, 2)
    q = torch.tensor([1., 0, 1, 0]).view(1, 1, 1, 2, 2)
    assert O.avg_ce({"a": p, "b": q}, "a", "b").item() == pytest.approx(float(g["bce"]), rel=1e-6)
    # hard Dice with four classes (argmax -> one-hot, ties to the first maximal channel)
    b4 = {"s": O.kat_scores(1), "t": O.kat_scores(2)}
    assert np.array_equal(torch.argmax(b4["s"], dim=1).numpy(), g["dice4_argmax_s"]) and g["dice4_argmax_s"][0, 0, 0, 0] == 0 and g["dice4_argmax_s"][1, 1, 1, 1] == 1
    assert O.avg_dsc(b4, "s", "t", binary=True, botindex=1, topindex=4).item() == pytest.approx(float(g["dice4_binary"]), rel=1e-7)
    assert O.avg_dsc(b4, "s", "t", binary=True, botindex=0, topindex=4).item() == pytest.approx(float(g["dice4_binary_all"]), rel=1e-7)
    assert np.allclose(O.avg_dsc(b4, "s", "t", binary=True, botindex=1, topindex=4, return_mean=False).numpy(), g["dice4_binary_nomean"], rtol=1e-7)


BLOCKS = {
    "conv_2_8": lambda: O.Conv(2, 8, norm_type=1),
    "dconv_8_16": lambda: O.DoubleConv(8, 16, norm_type=1),
    "down_8_16": lambda: O.Down(8, 16, norm_type=1),
    "up_16_8": lambda: O.Up(16, 8, norm_type=1),
    "down_64_128": lambda: O.Down(64, 128, norm_type=1),
    "up_256_128": lambda: O.Up(256, 128, norm_type=1),
}


@pytest.mark.parametrize("tag", sorted(BLOCKS))
def test_blocks(tag):
    g = G.load("blocks")
    mod = BLOCKS[tag]()
    seed = int(g[tag + ".seed"])
    shape = tuple(int(v) for v in g[tag + ".shape"])
    O.deterministic_fill_(mod, seed=seed)
    x = torch.from_numpy(2 * O.hashed_uniform(int(np.prod(shape)), 7001, seed) - 1).view(shape).requires_grad_(True)
    y = mod(x)
    w = torch.from_numpy(2 * O.hashed_uniform(y.numel(), 7002, seed) - 1).view_as(y)
    (y * w).sum().backward()
    G.check_tensor(g, tag + ".out", y, rtol=1e-5, what=tag)
    G.check_tensor(g, tag + ".gin", x.grad, rtol=1e-5, what=tag)
    G.check_grads(g, tag, [(n, p.grad) for n, p in mod.named_parameters()], rtol=1e-5, what=tag)


BLOCKS_NORM = {
    "conv_bn_2_8": lambda: O.Conv(2, 8, norm_type=2),
    "dconv_bn_8_16": lambda: O.DoubleConv(8, 16, norm_type=2),
    "down_bn_8_16": lambda: O.Down(8, 16, norm_type=2),
    "up_bn_16_8": lambda: O.Up(16, 8, norm_type=2),
    "down_bn_64_128": lambda: O.Down(64, 128, norm_type=2),
    "conv_bn_eval_2_8": lambda: O.Conv(2, 8, norm_type=2),
    "dconv_soft_8_16": lambda: O.DoubleConv(8, 16, norm_type=1, soft=True),
    "conv_soft_2_8": lambda: O.Conv(2, 8, norm_type=1, soft=True),
    "dconv_bn_soft_8_16": lambda: O.DoubleConv(8, 16, norm_type=2, soft=True),
    # norm_type=3: GSNorm3d (one group) inside the blocks — positive weights on positive inputs (oracle.ref_cpu.positive_fill_)
    "conv_gs_2_8": lambda: O.Conv(2, 8, norm_type=3),
    "dconv_gs_8_16": lambda: O.DoubleConv(8, 16, norm_type=3),
    "down_gs_8_16": lambda: O.Down(8, 16, norm_type=3),
}


def run_block_norm(mod, g, tag, device="cpu"):
    """the case of oracle/make_golden.py:_blocks_norm on `mod` -> (out, input gradient)"""
    seed = int(g[tag + ".seed"])
    shape = tuple(int(v) for v in g[tag + ".shape"])
    u = O.hashed_uniform(int(np.prod(shape)), 7001, seed)
    x = torch.from_numpy(u if "_gs_" in tag else 2 * u - 1).view(shape).to(device).requires_grad_(True)
    if "_eval_" in tag:
        mod.train()
        with torch.no_grad():
            mod(x)
        mod.eval()
    y = mod(x)
    w = torch.from_numpy(2 * O.hashed_uniform(y.numel(), 7002, seed) - 1).view_as(y).to(device)
    (y * w).sum().backward()
    return y, x.grad


def check_bn_buffers(g, tag, mod, rtol):
    for name, m in mod.named_modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            for buf in ("running_mean", "running_var"):
                ref = g["%s.bn.%s.%s" % (tag, name, buf)]
                got = getattr(m, buf).detach().cpu().double().numpy()
                assert np.abs(got - ref).max() <= rtol * max(np.abs(ref).max(), 1e-3), (tag, name, buf)
            assert int(m.num_batches_tracked) == int(g["%s.bn.%s.tracked" % (tag, name)]), (tag, name)


@pytest.mark.parametrize("tag", sorted(BLOCKS_NORM))
def test_blocks_norm(tag):
    """BatchNorm3d (training / eval) and Softplus blocks: the oracle against the reference's own blocks"""
    g = G.load("blocks_norm")
    mod = O.bn_fill_(O.deterministic_fill_(BLOCKS_NORM[tag](), seed=int(g[tag + ".seed"])))
    if "_gs_" in tag:
        O.positive_fill_(mod)
    y, gin = run_block_norm(mod, g, tag)
    G.check_tensor(g, tag + ".out", y, rtol=1e-5, what=tag)
    G.check_tensor(g, tag + ".gin", gin, rtol=1e-5, what=tag)
    G.check_grads(g, tag, [(n, p.grad) for n, p in mod.named_parameters()], rtol=1e-5, what=tag)
    check_bn_buffers(g, tag, mod, 1e-6)


def test_seg32_bn():
    g = G.load("seg32_bn")
    seg = O.bn_fill_(O.deterministic_fill_(O.Segmentation(1, 2, norm_type=2), seed=0))
    img, lab = O.synthetic_image(2, 32, seed=2), O.synthetic_label(2, 32, seed=3)
    loss, aux = O.seg_train_losses(seg, img, lab)
    loss.backward()
    assert loss.item() == pytest.approx(float(g["dice_loss"]), rel=1e-5)
    G.check_tensor(g, "pred", aux["batch"]["pred"], k=256, rtol=1e-4, what="seg32_bn")
    G.check_grads(g, "seg", [(n, p.grad) for n, p in seg.named_parameters()], rtol=1e-3, what="seg32_bn")
    check_bn_buffers(g, "seg", seg, 1e-5)


GS_BLOCKS = {
    "gsnorm_16_4": lambda M: M.GSNorm3d(16, num_group=4),
    "conv_gs_2_8": lambda M: M.Conv_GS(2, 8),
    "dconv_gs_8_16": lambda M: M.DoubleConv_GS(8, 16),
    "down_gs_8_16": lambda M: M.Down_GS(8, 16),
    "up_gs_16_8": lambda M: M.Up_GS(16, 8),
    "gsconv_k3_8_16": lambda M: M.GSConv3d(8, 16, 3, num_group=2, padding=1),
    "gsconv_k2_8_8": lambda M: M.GSConv3d(8, 8, 2, num_group=2, stride=2),
    "sconv_k3_1_8": lambda M: M.SConv3d(1, 8, 3, padding=1),
    "gsconvt_k2_8_8": lambda M: M.GSConvTranspose3d(8, 8, 2, num_group=2, stride=2),
}


def run_gs_block(mod, g, tag, device="cpu"):
    """the case of oracle/make_golden.py:_gs on `mod` -> (out, input gradient)"""
    seed = int(g[tag + ".seed"])
    shape = tuple(int(v) for v in g[tag + ".shape"])
    u = O.hashed_uniform(int(np.prod(shape)), 7001, seed)
    x = torch.from_numpy(u + 0.05 if tag.startswith("gsnorm") else 2 * u - 1).view(shape).to(device).requires_grad_(True)
    y = mod(x)
    w = torch.from_numpy(2 * O.hashed_uniform(y.numel(), 7002, seed) - 1).view_as(y).to(device)
    (y * w).sum().backward()
    return y, x.grad


@pytest.mark.parametrize("tag", sorted(GS_BLOCKS))
@pytest.mark.filterwarnings("ignore:.*align_corners.*")
def test_gs_blocks(tag):
    """the *_GS family: the oracle's restatement against the reference's own classes"""
    g = G.load("gs")
    mod = O.deterministic_fill_(GS_BLOCKS[tag](O), seed=int(g[tag + ".seed"]))
    y, gin = run_gs_block(mod, g, tag)
    G.check_tensor(g, tag + ".out", y, rtol=1e-5, what=tag)
    G.check_tensor(g, tag + ".gin", gin, rtol=1e-5, what=tag)
    G.check_grads(g, tag, [(n, p.grad) for n, p in mod.named_parameters()], rtol=2e-5, what=tag)


@pytest.mark.filterwarnings("ignore:.*align_corners.*")
def test_seg_gs32():
    g = G.load("gs")
    seg = O.deterministic_fill_(O.Segmentation_GS(1, 2), seed=0)
    img, lab = O.synthetic_image(2, 32, seed=2), O.synthetic_label(2, 32, seed=3)
    batch = seg({"img": img, "gt": O.one_hot(lab)}, "img", "pred")
    loss = 1 - O.avg_dsc(batch, "pred", "gt", botindex=1, topindex=2, eps=1e-4)
    loss.backward()
    assert loss.item() == pytest.approx(float(g["seg.dice_loss"]), rel=1e-5)
    G.check_tensor(g, "seg.pred", batch["pred"], k=256, rtol=1e-4, what="seg_gs32")
    G.check_grads(g, "seg", [(n, p.grad) for n, p in seg.named_parameters()], rtol=1e-3, what="seg_gs32")


def test_seg32():
    g = G.load("seg32")
    seg = O.deterministic_fill_(O.Segmentation(1, 2, norm_type=1), seed=0)
    loss, aux = O.seg_train_losses(seg, O.synthetic_image(2, 32, 2), O.synthetic_label(2, 32, 3), eps=1e-6)
    loss.backward()
    assert loss.item() == pytest.approx(float(g["dice_loss_eps1e6"]), rel=1e-6)
    G.check_tensor(g, "pred", aux["batch"]["pred"], k=256, rtol=1e-5)
    G.check_grads(g, "seg", [(n, p.grad) for n, p in seg.named_parameters()], rtol=1e-5)
    l4, _ = O.seg_train_losses(seg, O.synthetic_image(2, 32, 2), O.synthetic_label(2, 32, 3), eps=1e-4)
    assert l4.item() == pytest.approx(float(g["dice_loss_eps1e4"]), rel=1e-6)


def test_multiclass_seg32_four_classes_and_joint64_three_classes():
    """n_class = 1 + the number of labelled structures (main_source.py:92-93): the oracle against the reference's modules for 4 and 3 classes."""
    g = G.sub(G.load("multiclass"), "seg32_c4/")
    seg = O.deterministic_fill_(O.Segmentation(1, 4, norm_type=1), seed=0)
    loss, aux = O.seg_train_losses(seg, O.synthetic_image(2, 32, 2), O.synthetic_label(2, 32, 3, n_class=4), n_class=4)
    loss.backward()
    assert loss.item() == pytest.approx(float(g["dice_loss"]), rel=1e-6)
    G.check_tensor(g, "pred", aux["batch"]["pred"], k=256, rtol=1e-5)
    G.check_grads(g, "seg", [(n, p.grad) for n, p in seg.named_parameters()], rtol=1e-5)
    g = G.sub(G.load("multiclass"), "joint64_c3/")
    joint = O.build_joint(64, n_class=3)
    final, aux = O.joint_train_losses(joint, O.synthetic_image(2, 64, 2), O.synthetic_label(2, 64, 3, n_class=3), n_class=3)
    final.backward()
    assert final.item() == pytest.approx(float(g["final"]), rel=1e-6)
    assert aux["recon_loss"].item() == pytest.approx(float(g["recon_loss"]), rel=1e-5)
    G.check_tensor(g, "pred", aux["batch"]["pred"], k=512, rtol=1e-5)
    G.check_tensor(g, "recon", aux["batch"]["recon"], k=512, rtol=1e-5)
    G.check_grads(g, "seg", [(n, p.grad) for n, p in joint.Seg.named_parameters()], rtol=1e-4)


def test_seg96():
    g = G.load("seg96")
    seg = O.deterministic_fill_(O.Segmentation(1, 2, norm_type=1), seed=0)
    loss, aux = O.seg_train_losses(seg, O.synthetic_image(2, 96, 2), O.synthetic_label(2, 96, 3))
    loss.backward()
    assert loss.item() == pytest.approx(float(g["dice_loss"]), rel=1e-6)
    G.check_tensor(g, "pred", aux["batch"]["pred"], k=512, rtol=1e-5)
    G.check_grads(g, "seg", [(n, p.grad) for n, p in seg.named_parameters()], rtol=1e-4)


def test_vae64_train():
    g = G.load("vae64_train")
    vae = O.deterministic_fill_(O.VAE(2, 2, norm_type=1, dim=128, spatial=64), seed=0)
    noise = torch.from_numpy(2 * O.hashed_uniform(2 * 128, 7100, 5) - 1).view(2, 128)
    final, aux = O.vae_train_losses(vae, O.synthetic_label(2, 64, 3), scale=0.35, noise=noise)
    final.backward()
    assert final.item() == pytest.approx(float(g["final"]), rel=1e-6)
    assert aux["kl_loss"].item() == pytest.approx(float(g["kl"]), rel=1e-5)
    assert np.allclose(aux["batch"]["mean"].detach().numpy(), g["mean"], rtol=1e-4, atol=1e-6)
    assert np.allclose(aux["batch"]["std"].detach().numpy(), g["std"], rtol=1e-4, atol=1e-6)
    G.check_tensor(g, "recon", aux["batch"]["recon"], k=256, rtol=1e-5)
    G.check_grads(g, "vae", [(n, p.grad) for n, p in vae.named_parameters()], rtol=1e-4)


@pytest.mark.parametrize("side,bs,name", [(64, 2, "joint64"), (96, 2, "joint96"), (128, 1, "joint128")])
def test_joint_train(side, bs, name):
    g = G.load(name)
    joint = O.build_joint(side)
    final, aux = O.joint_train_losses(joint, O.synthetic_image(bs, side, 2), O.synthetic_label(bs, side, 3))
    final.backward()
    assert final.item() == pytest.approx(float(g["final"]), rel=1e-6)
    assert aux["recon_loss"].item() == pytest.approx(float(g["recon_loss"]), rel=1e-5)
    assert aux["dice_loss"].item() == pytest.approx(float(g["dice_loss"]), rel=1e-6)
    b = aux["batch"]
    assert O.KLloss(b).item() == pytest.approx(float(g["kl"]), rel=1e-5)
    G.check_tensor(g, "pred", b["pred"], k=512, rtol=1e-5)
    G.check_tensor(g, "recon", b["recon"], k=512, rtol=1e-5)
    G.check_grads(g, "seg", [(n, p.grad) for n, p in joint.Seg.named_parameters()], rtol=1e-4)
    assert all(p.grad is None for p in joint.Vae.parameters()) and bool(g["vae_grads_none"])


def test_joint160_forward():
    """BASELINE configs[4] geometry (160^3, B=2), forward only: the oracle against the reference modules' golden."""
    g = G.load("joint160_fwd")
    joint = O.build_joint(160)
    with torch.no_grad():
        final, aux = O.joint_train_losses(joint, O.synthetic_image(2, 160, 2), O.synthetic_label(2, 160, 3))
    assert final.item() == pytest.approx(float(g["final"]), rel=1e-6)
    assert aux["recon_loss"].item() == pytest.approx(float(g["recon_loss"]), rel=1e-5)
    G.check_tensor(g, "pred", aux["batch"]["pred"], k=512, rtol=1e-5)
    G.check_tensor(g, "recon", aux["batch"]["recon"], k=512, rtol=1e-5)


def test_domain_adaptation128():
    g = G.load("da128")
    student, teacher = O.build_joint(128), O.build_joint(128)
    O.deterministic_fill_(teacher.Seg, seed=1)
    for p in teacher.parameters():
        p.requires_grad = False
    img, lab = O.synthetic_image(1, 128, 2), O.synthetic_label(1, 128, 3)
    final, aux = O.domain_adaptation_losses(student, teacher, img, lab, lambda_vae=1.0, domain_loss_type=0)
    final.backward()
    assert final.item() == pytest.approx(float(g["final0"]), rel=1e-6)
    for k_o, k_g in (("recon_loss", "recon_loss"), ("kl_loss", "kl"), ("dice_loss", "dice_loss"),
                     ("dice_loss_fake", "fake_loss")):
        assert aux[k_o].item() == pytest.approx(float(g[k_g]), rel=1e-5), k_o
    assert aux["batch"]["fake"].double().sum().item() == float(g["fake.sum"])
    G.check_grads(g, "seg", [(n, p.grad) for n, p in student.Seg.named_parameters()], rtol=1e-4)
    cur = O.lambda_schedule(aux["recon_loss"].detach(), 1.0)
    assert cur == pytest.approx(float(g["cur_lambda"]))
    r, f = aux["recon_loss"].item(), aux["dice_loss_fake"].item()
    f8 = (r + f / cur) if cur > 1 else (cur * r + f)
    assert f8 == pytest.approx(float(g["final8"]), rel=1e-5)
    assert (cur * r + f) / (1 + cur) == pytest.approx(float(g["final9"]), rel=1e-5)
    for p in student.Seg.parameters():
        p.grad = None
    f8t, _ = O.domain_adaptation_losses(student, teacher, img, lab, lambda_vae=1.0, domain_loss_type=8)
    f8t.backward()
    assert f8t.item() == pytest.approx(float(g["final8"]), rel=1e-5)
    G.check_grads(g, "seg8", [(n, p.grad) for n, p in student.Seg.named_parameters()], rtol=1e-4)


def test_test_time_finetune128():
    """main_target.py:809-953 (two test-time-training iterations at lr 1e-2, domain_loss_type 8, then hard-Dice validation) restated
    in the oracle against the same loop run on the reference's own modules (oracle/make_golden.py:_ft128)."""
    g = G.load("ft128")
    model, model_ft, teacher = O.build_joint(128), O.build_joint(128), O.build_joint(128)
    O.deterministic_fill_(teacher.Seg, seed=1)
    for p in teacher.parameters():
        p.requires_grad = False
    img, lab = O.synthetic_image(1, 128, 2), O.synthetic_label(1, 128, 3)
    log = O.test_time_finetune(model, model_ft, teacher, img, lab, steps=2, lr=1e-2, lambda_vae=1.0, domain_loss_type=8)
    for it, rec in enumerate(log):
        for k_o, k_g in (("recon_loss", "recon_loss"), ("dice_loss", "dice_loss"), ("dice_loss_fake", "fake_loss"), ("final_loss", "final")):
            assert rec[k_o].item() == pytest.approx(float(g["it%d.%s" % (it, k_g)]), rel=1e-5), (it, k_o)
    ref = dict(model.Seg.named_parameters())
    upd = [(n, (p.detach() - ref[n].detach()) / 1e-2) for n, p in model_ft.Seg.named_parameters()]
    # the update is a difference of fp32 weights divided by lr: its own rounding noise is ~6e-8*|w|/lr per element
    G.check_grads(g, "upd", upd, rtol=2e-3)
    with torch.no_grad():
        batch = {"img": img, "gt": O.one_hot(lab)}
        batch = model(batch, "img", "pred_noft", "recon_noft")
        batch = model_ft(batch, "img", "pred", "recon")
        assert O.avg_dsc(batch, "pred_noft", "gt", binary=True, botindex=1, topindex=2).item() == pytest.approx(float(g["score_noft"]), rel=1e-5)
        assert O.avg_dsc(batch, "pred", "gt", binary=True, botindex=1, topindex=2).item() == pytest.approx(float(g["score"]), rel=1e-5)
    G.check_tensor(g, "pred", batch["pred"], k=512, rtol=1e-4)


def test_vae128_native_with_recorded_noise():
    g = G.load("vae128_train")
    vae = O.deterministic_fill_(O.VAE(2, 2, norm_type=1, dim=128, spatial=128), seed=0)
    assert vae.fc_mean.weight.shape == (128, 16384)      # reference state_dict shape at the native size
    final, aux = O.vae_train_losses(vae, O.synthetic_label(1, 128, 3), scale=0.35, noise=torch.from_numpy(g["z"]))
    final.backward()
    assert final.item() == pytest.approx(float(g["final"]), rel=1e-6)
    G.check_tensor(g, "recon", aux["batch"]["recon"], k=512, rtol=1e-5)
    G.check_grads(g, "vae", [(n, p.grad) for n, p in vae.named_parameters()], rtol=1e-4)


# ---- Encoder / Fusion / Embed (oracle/make_golden.py: gold_rank4) ------------------------------
def _dsc_main_source(s, t, bot, top, eps=1e-4):
    d = 2 * torch.sum(s * t, (2, 3, 4)) / (torch.sum(s, (2, 3, 4)) + torch.sum(t, (2, 3, 4)) + eps)
    return torch.mean(d[:, bot:top])


def test_encoder128():
    g = G.load("enc128")
    enc = O.deterministic_fill_(O.Encoder(1, 1, norm_type=1), seed=4)
    assert enc.fc1.weight.shape == (1024, 16384)          # reference state_dict shape
    x = O.synthetic_image(1, 128, seed=5).abs().requires_grad_(True)
    out = enc(x)
    out.sum().backward()
    assert np.allclose(out.detach().numpy(), g["out"], rtol=1e-5)
    G.check_tensor(g, "gx", x.grad, k=512, rtol=1e-4)
    G.check_grads(g, "enc", [(n, p.grad) for n, p in enc.named_parameters()], rtol=1e-4)


def test_fusion64():
    g = G.load("fusion64")
    fus = O.deterministic_fill_(O.Fusion(1, 2, 2, norm_type=1), seed=6)
    img, gt = O.synthetic_image(1, 64, seed=2), O.one_hot(O.synthetic_label(1, 64, seed=3))
    mask = O.one_hot(O.synthetic_label(1, 64, seed=7)).requires_grad_(True)
    batch = fus({"img": img, "mask": mask}, "img", "mask", "pred")
    loss = 1 - _dsc_main_source(batch["pred"], gt, 1, 2)
    loss.backward()
    assert loss.item() == pytest.approx(float(g["loss"]), rel=1e-6)
    G.check_tensor(g, "pred", batch["pred"], k=512, rtol=1e-5)
    G.check_tensor(g, "gmask", mask.grad, k=512, rtol=1e-4)
    G.check_grads(g, "fus", [(n, p.grad) for n, p in fus.named_parameters()], rtol=1e-4)


def test_embed128():
    g = G.load("embed128")
    emb = O.Embed([O.Encoder(1, 128, norm_type=1), O.VAE(2, 2, norm_type=1, dim=128), O.Fusion(1, 2, 2, norm_type=1)])
    O.deterministic_fill_(emb, seed=8)
    img, gt = O.synthetic_image(1, 128, seed=2), O.one_hot(O.synthetic_label(1, 128, seed=3))
    batch = emb({"img": img, "venous_pancreas_only": gt}, "img", "pred", noise=torch.from_numpy(g["z"]))
    dsc = 1 - _dsc_main_source(batch["pred"], gt, 1, 2)
    lat = torch.mean((batch["latent_code"] - batch["latent_code_gt"].detach()) ** 2)
    (dsc + lat).backward()
    assert dsc.item() == pytest.approx(float(g["dice_loss"]), rel=1e-5)
    assert lat.item() == pytest.approx(float(g["latent_loss"]), rel=1e-5)
    for k in ("pred", "gt_recon", "init_seg", "seg_recon"):
        G.check_tensor(g, k, batch[k], k=512, rtol=1e-4)
    for pre, mod in (("enc", emb.Encoder), ("vae", emb.Vae), ("fus", emb.Fusion)):
        G.check_grads(g, pre, [(n, p.grad) for n, p in mod.named_parameters()], rtol=2e-4)


# ---- remaining train methods (oracle/make_golden.py: gold_methods) --------------------------------------------------------------------
def _embed(spatial=128):
    emb = O.Embed([O.Encoder(1, 128, norm_type=1, spatial=spatial), O.VAE(2, 2, norm_type=1, dim=128, spatial=spatial), O.Fusion(1, 2, 2, norm_type=1)])
    return O.deterministic_fill_(emb, seed=8)


def test_embed_train_and_refine_vae128():
    """main_source.py:546-628: embed_train's and refine_vae's loss bodies restated in the oracle, against the reference modules."""
    g = G.load("embed_train128")
    emb = _embed()
    img, lab = O.synthetic_image(1, 128, seed=2), O.synthetic_label(1, 128, seed=3)
    z = torch.from_numpy(g["z"])
    final, aux = O.embed_train_losses(emb, img, lab, noise=z)
    final.backward()
    for k_o, k_g in (("dice_loss1", "dice_loss1"), ("dice_loss2", "dice_loss2"), ("recon_loss", "recon_loss"), ("inpaint_loss", "inpaint_loss"),
                     ("kl_loss", "kl"), ("mse_loss", "mse")):
        assert aux[k_o].item() == pytest.approx(float(g[k_g]), rel=1e-5), k_o
    assert final.item() == pytest.approx(float(g["final"]), rel=1e-6)
    for pre, mod in (("enc", emb.Encoder), ("vae", emb.Vae), ("fus", emb.Fusion)):
        G.check_grads(g, pre, [(n, p.grad) for n, p in mod.named_parameters()], rtol=2e-4, dead=G.is_dead_bias)
    emb2 = _embed()
    for p in emb2.Encoder.parameters():
        p.requires_grad = False
    rfinal, _ = O.refine_vae_losses(emb2, img, lab, noise=z)
    rfinal.backward()
    assert rfinal.item() == pytest.approx(float(g["refine_final"]), rel=1e-6)
    G.check_grads(g, "rvae", [(n, p.grad) for n, p in emb2.Vae.named_parameters()], rtol=2e-4, dead=G.is_dead_bias)
    G.check_grads(g, "rfus", [(n, p.grad) for n, p in emb2.Fusion.named_parameters()], rtol=2e-4, dead=G.is_dead_bias)


def test_sep_joint_train128():
    g = G.load("sep_joint128")
    student, teacher = O.build_joint(128), O.build_joint(128)
    O.deterministic_fill_(teacher.Seg, seed=1)
    for p in teacher.parameters():
        p.requires_grad = False
    final, aux = O.sep_joint_train_losses(student, teacher, O.synthetic_image(1, 128, 2), O.synthetic_label(1, 128, 3))
    final.backward()
    assert final.item() == pytest.approx(float(g["final"]), rel=1e-6)
    assert aux["recon_loss"].item() == pytest.approx(1 - float(g["recon"].mean()), rel=1e-5)
    G.check_grads(g, "seg", [(n, p.grad) for n, p in student.Seg.named_parameters()], rtol=1e-4, dead=G.is_dead_bias)


def test_domain_adaptation_dis_and_discriminator_train128():
    g = G.load("da_dis128")
    seg = O.deterministic_fill_(O.Segmentation(1, 2, norm_type=1), seed=0)
    dis = O.deterministic_fill_(O.Encoder(1, 1, norm_type=1), seed=4)
    j2 = O.Joint2([seg, dis])
    for p in j2.Dis.parameters():
        p.requires_grad = False
    teacher = O.deterministic_fill_(O.Segmentation(1, 2, norm_type=1), seed=1)
    for p in teacher.parameters():
        p.requires_grad = False
    img, lab = O.synthetic_image(1, 128, 2), O.synthetic_label(1, 128, 3)
    final, aux = O.domain_adaptation_dis_losses(j2, teacher, img, lab, lambda_vae=1.0, epoch=1, lambda_vae_warmup=0)
    final.backward()
    assert final.item() == pytest.approx(float(g["final"]), rel=1e-6)
    for k_o, k_g in (("dice_loss", "dice_loss"), ("dice_loss_fake", "fake_loss"), ("discriminator_loss", "dis_loss")):
        assert aux[k_o].item() == pytest.approx(float(g[k_g]), rel=1e-5), k_o
    G.check_grads(g, "seg", [(n, p.grad) for n, p in j2.Seg.named_parameters()], rtol=1e-4, dead=G.is_dead_bias)
    dis2 = O.deterministic_fill_(O.Encoder(1, 1, norm_type=1), seed=4)
    dl, daux = O.discriminator_train_loss(dis2, lab.float(), torch.tensor([[0.7]]))
    dl.backward()
    assert dl.item() == pytest.approx(float(g["dtrain_loss"]), rel=1e-5)
    G.check_grads(g, "dis", [(n, p.grad) for n, p in dis2.named_parameters()], rtol=2e-4, dead=G.is_dead_bias)
