"""GPU parity of the step glue (SURVEY.md §8a "step glue", "EMA teacher"): the multi-tensor optimiser / EMA / gradient-staging
kernels against torch.optim and the reference's own arithmetic on the CPU, plus the two robustness cases of ADVICE.md round 1:
a HIP graph that replays a frozen network must see weights changed by EMA, and dropout > 0 must not be captured."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(8, 1, 3, 3, 3), (8,), (16, 8, 2, 2, 2), (4097,), (128, 64, 3, 3, 3), (2,), (3, 4100)]      # ragged around the 4096-element chunk


def _tensors(seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g) * scale for s in SHAPES]


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_adam_multi_matches_torch_adam(wd):
    """vs_adam_multi (main_source.py:292-294: torch.optim.Adam(betas=(0.9, 0.999), weight_decay)) over 3 steps."""
    from vae_segmentation_amd import optim
    ref = [torch.nn.Parameter(t.clone()) for t in _tensors(0)]
    mine = [torch.nn.Parameter(t.clone().cuda()) for t in _tensors(0)]
    o_ref = torch.optim.Adam(ref, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    o_mine = optim.Adam(mine, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    for step in range(3):
        for p, q, g in zip(ref, mine, _tensors(10 + step, 0.1)):
            p.grad, q.grad = g.clone(), g.clone().cuda()
        o_ref.step()
        o_mine.step()
    torch.cuda.synchronize()
    for p, q in zip(ref, mine):
        assert float((q.detach().cpu() - p.detach()).abs().max()) <= 2e-6 * float(p.detach().abs().max()) + 1e-7
        st_r, st_m = o_ref.state[p], o_mine.state[q]
        assert torch.allclose(st_m["exp_avg"].cpu(), st_r["exp_avg"], rtol=1e-5, atol=1e-8)
        assert torch.allclose(st_m["exp_avg_sq"].cpu(), st_r["exp_avg_sq"], rtol=1e-5, atol=1e-10)
        assert int(st_m["step"]) == 3


@pytest.mark.parametrize("momentum,wd", [(0.9, 0.0), (0.0, 1e-3), (0.9, 1e-3)])
def test_sgd_multi_matches_torch_sgd(momentum, wd):
    from vae_segmentation_amd import optim
    ref = [torch.nn.Parameter(t.clone()) for t in _tensors(1)]
    mine = [torch.nn.Parameter(t.clone().cuda()) for t in _tensors(1)]
    o_ref = torch.optim.SGD(ref, lr=1e-2, momentum=momentum, weight_decay=wd)
    o_mine = optim.SGD(mine, lr=1e-2, momentum=momentum, weight_decay=wd)
    for step in range(3):
        for p, q, g in zip(ref, mine, _tensors(20 + step, 0.1)):
            p.grad, q.grad = g.clone(), g.clone().cuda()
        o_ref.step()
        o_mine.step()
    for p, q in zip(ref, mine):
        assert float((q.detach().cpu() - p.detach()).abs().max()) <= 1e-6 * float(p.detach().abs().max()) + 1e-7


def test_ema_multi_matches_reference_arithmetic():
    """vs_ema_multi against main_target.py:512-516: sd_teacher[key] = alpha * sd_teacher[key] + (1 - alpha) * sd_student[key] over the
    Seg state_dict, three updates in a row; the teacher's cached packed weights follow (in place: same buffer address)."""
    import joint_model as M
    from oracle import ref_cpu as O
    from vae_segmentation_amd import ops, optim
    teacher = O.deterministic_fill_(M.Segmentation(1, 2, norm_type=1), seed=1).cuda()
    student = O.deterministic_fill_(M.Segmentation(1, 2, norm_type=1), seed=0).cuda()
    for p in teacher.parameters():
        p.requires_grad = False
    sd_t = {k: v.detach().cpu().clone() for k, v in teacher.state_dict().items()}
    img = O.synthetic_image(1, 32, 2).cuda()
    with torch.no_grad():
        teacher({"x": img}, "x", "p")                          # fills the frozen pack cache
    w = teacher.down4.conv[1].conv[0].weight
    addr = {k: ent[0].data_ptr() for k, ent in w._vs_pack_cache.items()}
    assert addr
    alpha = 0.9
    for _ in range(3):
        sd_s = {k: v.detach().cpu() for k, v in student.state_dict().items()}
        for k in sd_s:
            sd_t[k] = alpha * sd_t[k] + (1 - alpha) * sd_s[k]
        optim.ema_update(teacher, student, alpha)
    torch.cuda.synchronize()
    for k, v in teacher.state_dict().items():
        assert float((v.cpu() - sd_t[k]).abs().max()) <= 1e-6 * float(sd_t[k].abs().max()) + 1e-8, k
    assert {k: ent[0].data_ptr() for k, ent in w._vs_pack_cache.items()} == addr          # refreshed in place
    # the refreshed images are the ones a fresh pack of the new weights gives
    for (form, c_pad, dt), ent in w._vs_pack_cache.items():
        fresh = ops.pack_weight(w, form, c_pad, ops._DT_TORCH[dt])
        assert torch.equal(fresh, ent[0])
    # and the forward with them equals a forward of a network that simply holds the new weights
    other = M.Segmentation(1, 2, norm_type=1).cuda()
    other.load_state_dict({k: v.cuda() for k, v in sd_t.items()})
    with torch.no_grad():
        a = teacher({"x": img}, "x", "p")["p"]
        b = other({"x": img}, "x", "p")["p"]
    assert float((a - b).abs().max()) < 1e-5


def test_copy_scale_multi_and_scale_copy():
    """vs_copy_scale_multi (gradient staging into the flat all-reduce bucket, with the 1/world factor) and vs_scale_copy (in-place
    1/world after a sum all-reduce) against plain tensor arithmetic."""
    from vae_segmentation_amd._lib import check, lib
    from vae_segmentation_amd.optim import _Tables
    src = [t.cuda() for t in _tensors(3)]
    flat = torch.full((sum(t.numel() for t in src) + 5,), 7.0, device="cuda")
    dst, off = [], 0
    for t in src:
        dst.append(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    (sp, dp, sizes, bm), nb = _Tables().get([src, dst], flat.device)
    stream = torch.cuda.current_stream().cuda_stream
    check(lib.vs_copy_scale_multi(sp.data_ptr(), dp.data_ptr(), sizes.data_ptr(), bm.data_ptr(), nb, 0.125, stream), "copy_scale_multi")
    want = torch.cat([t.reshape(-1) for t in src]) * 0.125
    assert torch.equal(flat[:off], want) and bool((flat[off:] == 7.0).all())
    check(lib.vs_scale_copy(flat.data_ptr(), flat.data_ptr(), off, 0.5, stream), "scale_copy")
    assert torch.equal(flat[:off], want * 0.5) and bool((flat[off:] == 7.0).all())
    out = torch.empty(off, device="cuda")
    check(lib.vs_scale_copy(flat.data_ptr(), out.data_ptr(), off, 3.0, stream), "scale_copy")
    assert torch.equal(out, want * 0.5 * 3.0)


def _joint(M, O, side, seed_seg=0):
    seg = M.Segmentation(n_channels=1, n_class=2, norm_type=1)
    vae = M.VAE(n_channels=2, n_class=2, norm_type=1, dim=128, spatial=side)
    joint = M.Joint(models=[seg, vae])
    O.deterministic_fill_(joint, seed=0)
    if seed_seg:
        O.deterministic_fill_(joint.Seg, seed=seed_seg)
    joint = joint.cuda()
    for p in joint.Vae.parameters():
        p.requires_grad = False
    joint.Vae.eval()
    return joint


def test_graph_replay_sees_ema_updated_teacher():
    """the captured test-time-training graph holds raw pointers to the frozen teacher's packed weights.  After an EMA
    update of the teacher the graph-replayed runner must give what an eager runner built on the updated teacher gives."""
    import joint_model as M
    from oracle import ref_cpu as O
    from vae_segmentation_amd import optim
    from vae_segmentation_amd import train as T
    side = 64
    img, lab = O.synthetic_image(1, side, 2).cuda(), O.synthetic_label(1, side, 3).cuda()
    model, model_ft, teacher = _joint(M, O, side), _joint(M, O, side), _joint(M, O, side, seed_seg=1)
    for p in teacher.parameters():
        p.requires_grad = False
    runner = T.TestTimeFinetune(model, model_ft, teacher, side, steps=1, lr=1e-2, domain_loss_type=0, graph=True)
    log0, _, _, _ = runner.run(img, lab)
    before = float(log0[0]["dice_loss_fake"].item())
    optim.ema_update(teacher.Seg, model.Seg, 0.5)               # moves the teacher half-way to the student: pseudo-labels change
    log1, _, s1, _ = runner.run(img, lab)
    # eager runner on copies holding the updated teacher weights
    model_b, model_ft_b, teacher_b = _joint(M, O, side), _joint(M, O, side), _joint(M, O, side)
    teacher_b.load_state_dict(teacher.state_dict())
    for p in teacher_b.parameters():
        p.requires_grad = False
    eager = T.TestTimeFinetune(model_b, model_ft_b, teacher_b, side, steps=1, lr=1e-2, domain_loss_type=0, graph=False)
    log2, _, s2, _ = eager.run(img, lab)
    after_graph, after_eager = float(log1[0]["dice_loss_fake"].item()), float(log2[0]["dice_loss_fake"].item())
    assert abs(after_graph - after_eager) < 1e-5, (before, after_graph, after_eager)
    assert abs(after_graph - before) > 1e-4, "the EMA update should have changed the pseudo-label loss"
    assert abs(s1.item() - s2.item()) < 1e-4


def test_dropout_is_not_captured():
    """host-side dropout seeds must not be baked into a graph.  A direct capture raises; TestTimeFinetune falls back to
    eager launches, and two iterations then draw different masks."""
    import joint_model as M
    from oracle import ref_cpu as O
    from vae_segmentation_amd import optim
    from vae_segmentation_amd import train as T
    seg = O.deterministic_fill_(M.Segmentation(1, 2, norm_type=1), seed=0).cuda()
    img, lab = O.synthetic_image(1, 32, 2).cuda(), O.synthetic_label(1, 32, 3).cuda()

    def loss_fn():
        batch = seg({"x": img}, "x", "p", dropout=0.2)
        return batch["p"][:, 1].mean(), {}
    with pytest.raises(RuntimeError, match="dropout"):
        T.GraphedStep(loss_fn, seg.parameters(), optim.SGD(seg.parameters(), lr=1e-2), warmup=1)
    torch.cuda.synchronize()
    from vae_segmentation_amd import ops
    ops.drop_stale_wgrads()
    side = 64
    model, model_ft, teacher = _joint(M, O, side), _joint(M, O, side), _joint(M, O, side, seed_seg=1)
    model_ft.seg_dropout = 0.2
    runner = T.TestTimeFinetune(model, model_ft, teacher, side, steps=2, lr=0.0, graph=True)
    assert runner.stepper is None and runner.graph is False
    log, _, _, _ = runner.run(O.synthetic_image(1, side, 2).cuda(), O.synthetic_label(1, side, 3).cuda())
    a, b = float(log[0]["dice_loss"].item()), float(log[1]["dice_loss"].item())
    assert a != b                                               # lr = 0: same weights, so only fresh masks can make the two iterations differ


def _seg_step_pair(side=32, seed=0):
    import joint_model as M
    from oracle import ref_cpu as O
    seg = O.deterministic_fill_(M.Segmentation(1, 2, norm_type=1), seed=seed).cuda()
    img, lab = O.synthetic_image(1, side, 2 + seed).cuda(), O.synthetic_label(1, side, 3 + seed).cuda()
    return seg, img, lab


def test_two_graphed_steps_first_replays_after_second_was_built():
    """the captured tail's re-pack launch reads a descriptor table by address.  Building a second GraphedStep on another
    model registers more trainable images and REPLACES ops' table; the first graph must keep its own alive and go on training its model
    exactly as an eager loop does."""
    from vae_segmentation_amd import ops, optim
    from vae_segmentation_amd import train as T
    seg_a, img_a, lab_a = _seg_step_pair(seed=0)
    seg_b, img_b, lab_b = _seg_step_pair(seed=1)
    opt_a = optim.SGD(seg_a.parameters(), lr=1e-2, momentum=0.9)
    gs_a = T.GraphedStep(lambda: T.seg_train_losses(seg_a, img_a, lab_a), list(seg_a.parameters()), opt_a, warmup=1)
    assert gs_a.tail
    table_a = gs_a._repack_table
    assert table_a is not None and table_a.data_ptr() == ops.repack_table().data_ptr()
    opt_b = optim.SGD(seg_b.parameters(), lr=1e-2, momentum=0.9)
    gs_b = T.GraphedStep(lambda: T.seg_train_losses(seg_b, img_b, lab_b), list(seg_b.parameters()), opt_b, warmup=1)
    assert gs_b.tail and ops.repack_table().data_ptr() != table_a.data_ptr()        # replaced, not rewritten
    assert gs_a._repack_table is table_a                                           # and still alive under the first graph
    torch.cuda.empty_cache()                                                       # a freed table would be released to the driver here
    junk = [torch.full((table_a.numel(),), 0xFF, dtype=torch.uint8, device="cuda") for _ in range(64)]    # ... or recycled by these
    for _ in range(3):
        gs_a.step()
        gs_b.step()
    torch.cuda.synchronize()
    del junk
    # eager twin of model A: same fill, same three steps
    seg_c, _, _ = _seg_step_pair(seed=0)
    opt_c = optim.SGD(seg_c.parameters(), lr=1e-2, momentum=0.9)
    for _ in range(3):
        opt_c.zero_grad()
        loss, _ = T.seg_train_losses(seg_c, img_a, lab_a)
        loss.backward()
        opt_c.step()
    torch.cuda.synchronize()
    for (n, p), q in zip(seg_a.named_parameters(), seg_c.parameters()):
        assert torch.equal(p.detach(), q.detach()), n
    # the images the first graph re-packed are the images of its CURRENT weights
    w = seg_a.down1.conv[1].conv[0].weight
    for (form, c_pad, dt), ent in w._vs_pack_plan.items():
        assert torch.equal(ops.pack_weight(w, form, c_pad, ops._DT_TORCH[dt]), ent[0])


def test_lr_schedule_needs_no_recapture_and_matches_eager():
    """the captured SGD launch reads lr / momentum / weight decay from device memory (vs_sgd_momentum_dev_multi); a scheduler
    that moves them between steps is followed WITHOUT re-capture, and the trajectory equals the eager loop's bit for bit."""
    from vae_segmentation_amd import optim
    from vae_segmentation_amd import train as T
    seg_a, img, lab = _seg_step_pair(seed=0)
    seg_c, _, _ = _seg_step_pair(seed=0)
    opt_a = optim.SGD(seg_a.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    opt_c = optim.SGD(seg_c.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    gs = T.GraphedStep(lambda: T.seg_train_losses(seg_a, img, lab), list(seg_a.parameters()), opt_a, warmup=1)
    assert gs.tail
    lrs = [1e-2, 5e-3, 5e-3, 2e-2, 1e-3]
    for i, lr in enumerate(lrs):
        for o in (opt_a, opt_c):
            o.param_groups[0]["lr"] = lr
            if i == 3:
                o.param_groups[0]["momentum"] = 0.5
        gs.step()
        opt_c.zero_grad()
        loss, _ = T.seg_train_losses(seg_c, img, lab)
        loss.backward()
        opt_c.step()
    torch.cuda.synchronize()
    assert gs.recaptures == 0
    for (n, p), q in zip(seg_a.named_parameters(), seg_c.parameters()):
        assert torch.equal(p.detach(), q.detach()), n


def test_requires_grad_toggle_recaptures_the_tail():
    """The live parameter set IS baked into the capture: freezing a block between steps (embed_train does it by epoch parity,
    main_source.py:550-554) re-captures, the frozen block stops moving, and un-freezing re-captures again."""
    from vae_segmentation_amd import optim
    from vae_segmentation_amd import train as T
    seg, img, lab = _seg_step_pair(seed=0)
    params = list(seg.parameters())
    opt = optim.SGD(params, lr=1e-2, momentum=0.0)
    gs = T.GraphedStep(lambda: T.seg_train_losses(seg, img, lab), params, opt, warmup=1)
    assert gs.tail
    gs.step()
    frozen = list(seg.down4.parameters())
    for p in frozen:
        p.requires_grad = False
    snap = [p.detach().clone() for p in frozen]
    other = seg.out_block.weight.detach().clone()
    gs.step()
    gs.step()
    torch.cuda.synchronize()
    assert gs.recaptures == 1 and gs.tail
    assert all(torch.equal(p.detach(), s) for p, s in zip(frozen, snap))
    assert not torch.equal(seg.out_block.weight.detach(), other)
    for p in frozen:
        p.requires_grad = True
    gs.step()
    torch.cuda.synchronize()
    assert gs.recaptures == 2
    assert any(not torch.equal(p.detach(), s) for p, s in zip(frozen, snap))
