"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/vaeseg.h declares,
argument validation rejects bad calls without touching a GPU, the host-side helpers, and the world_size-2
gradient exchange (gloo) used by the data-parallel step."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vae_segmentation_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 85
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(raw, name), name
    for must in ("vs_conv_gather_fwd", "vs_conv_scatter_fwd", "vs_conv_k3_softmax2_fwd", "vs_conv_wgrad", "vs_pack_weight",
                 "vs_instnorm_relu_fwd", "vs_instnorm_relu_bwd_reduce", "vs_instnorm_relu_bwd_apply", "vs_softmax2_bwd",
                 "vs_dice_fwd", "vs_dice_bwd", "vs_kl_fwd", "vs_kl_bwd", "vs_reparam_fwd", "vs_linear_fwd", "vs_onehot",
                 "vs_binarize", "vs_bce_fwd", "vs_sgd_momentum_multi", "vs_adam_multi", "vs_ema_multi", "vs_strerror",
                 "vs_version", "vs_conv_wgrad_workspace_bytes", "vs_copy_scale_multi", "vs_conv_gather_bwd_data",
                 "vs_conv_scatter_bwd_data", "vs_pack_weight_multi", "vs_dropout", "vs_softmax2_dropout_bwd",
                 "vs_conv_k3_softmax2_dropout_fwd", "vs_softmax_cl_fwd", "vs_softmax_cl_bwd", "vs_conv_wgrad_multi", "vs_conv_wgrad_multi_workspace_bytes",
                 "vs_dice_loss_multi_fwd", "vs_dice_loss_multi_bwd", "vs_dice_loss_multi_scratch_doubles"):
        assert must in protos, must
    assert _lib.lib.vs_version() == 500
    assert b"dtype" in _lib.lib.vs_strerror(-3)


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before any launch (no compute call is made here)."""
    from vae_segmentation_amd._lib import lib
    assert lib.vs_conv_gather_fwd(None, None, None, None, None, None, 1, 4, 4, 4, 8, 8, 0, 0, 1e-5, None) == -1
    assert lib.vs_pack_weight(None, None, 8, 8, 27, 8, 0, 0, None) == -1
    assert lib.vs_dice_fwd(None, None, None, None, None, 1, 2, 64, 1, 2, 1e-4, None) == -1
    # packed-weight sizing is pure host arithmetic: 16 rows x (27 taps x 8 ch -> 7 k-groups of 32) bf16 fragments ...
    assert lib.vs_packed_weight_bytes(16, 8, 27, 1) == 1 * 1 * 7 * 64 * 8 * 2
    # ... except the 8-channel 3x3x3 layers (<= 8 rows, bf16), packed as 9 Toeplitz k-groups (tz, ty) for k3t_kernel
    assert lib.vs_packed_weight_bytes(8, 8, 27, 1) == 9 * 64 * 8 * 2
    assert lib.vs_packed_weight_bytes(8, 8, 27, 0) == 18 * 64 * 4 * 4                # fp32: 18 y-Toeplitz k-groups (36 window taps (dz, wy, dx), two per k-group) for k3_kernel<.., TY>
    assert lib.vs_packed_weight_bytes(16, 8, 27, 0) == 1 * 1 * 14 * 64 * 4 * 4       # more than 8 rows: the standard order
    assert lib.vs_packed_weight_bytes(64, 64, 27, 0) == 4 * 2 * 54 * 64 * 4 * 4
    assert lib.vs_conv_wgrad_workspace_bytes(2, 96, 96, 96, 8, 8, 0) > 0
    # maximum sizes: shapes the 32-bit offsets of the kernels cannot address are refused (VS_ESHAPE = -2), never mis-computed.
    # (fake, aligned, never dereferenced pointers: the check precedes every launch)
    fake = 0x10000
    assert lib.vs_conv_gather_fwd(fake, None, fake, None, fake, None, 1, 2048, 1024, 1024, 8, 8, 0, 1, 1e-5, None) == -2     # >= 2^31 elements
    assert lib.vs_conv_gather_fwd(fake, None, fake, None, fake, None, 1, 512, 512, 512, 8, 8, 0, 1, 1e-5, None) == -2        # bf16 3x3x3: >= 2 GiB
    assert lib.vs_conv_gather_fwd(fake, None, fake, None, fake, None, 1, 5, 5, 5, 8, 8, 1, 1, 1e-5, None) == -2              # stride-2 conv on odd sizes
    assert lib.vs_conv_gather_fwd(fake, None, fake, None, fake, None, 1, 4, 4, 4, 24, 8, 0, 1, 1e-5, None) == -2             # channel count not 8 / 16 / 32k
    assert lib.vs_conv_gather_fwd(fake + 4, None, fake, None, fake, None, 1, 4, 4, 4, 8, 8, 0, 1, 1e-5, None) == -5          # VS_EALIGN
    assert lib.vs_conv_gather_fwd(fake, None, fake, None, fake, None, 0, 4, 4, 4, 8, 8, 0, 1, 1e-5, None) == -2              # empty batch


def test_grouped_weight_gradient_planning_without_gpu():
    """vs_conv_wgrad_multi's host side: descriptor layout (ctypes mirror == C struct), workspace planning and rejection of bad
    descriptors happen before any launch."""
    from vae_segmentation_amd import ops
    from vae_segmentation_amd._lib import lib, VS_BF16, VS_F32, VS_CONV_K3, VS_CONV_K2S2
    assert ctypes.sizeof(ops.WgradDesc) == 112
    fake = 0x10000

    counter = [0]

    def desc(n, d, h, w, m_ch, c_ch, kind, m_real=None, c_real=None, bias=False, dw=None):
        counter[0] += 1
        dst = dw if dw is not None else fake + 0x100000 * counter[0]          # a destination of its own: descriptors sharing dw are parts of one gradient
        x = ops.WgradDesc(fake, None, fake, None, dst, None, None, 0, 0, 0, n, d, h, w, m_ch, c_ch, m_real or m_ch, c_real or c_ch, kind, 0)
        if bias:
            x.bias_g, x.db, x.bias_rows, x.bias_c_ch, x.bias_c_real = fake, dst + 0x80000, n * d * h * w, m_ch, m_real or m_ch
        return x

    layers = [desc(2, 96, 96, 96, 8, 8, VS_CONV_K3), desc(2, 48, 48, 48, 16, 16, VS_CONV_K3), desc(2, 6, 6, 6, 128, 128, VS_CONV_K3),
              desc(2, 24, 24, 24, 32, 32, VS_CONV_K2S2, bias=True), desc(2, 48, 48, 48, 8, 8, VS_CONV_K2S2, bias=True)]
    arr = (ops.WgradDesc * len(layers))(*layers)
    grouped = lib.vs_conv_wgrad_multi_workspace_bytes(ctypes.addressof(arr), len(layers), VS_BF16)
    single = [lib.vs_conv_wgrad_workspace_bytes(x.n, x.dp, x.hp, x.wp, x.m_ch, x.c_ch, x.kind) for x in layers]
    # grouped: every layer owns a slab region (they run in one grid); fp32 mode: the 3x3x3 layers likewise (grouped limb launches, csrc/wgrad.hip
    # g3x_group_kernel) and, since round 4, the stride-2 layers (g3_group_kernel: one grid per channel-block width, <= 8 tiles per workgroup)
    assert grouped > 0 and grouped % 16 == 0
    assert grouped >= 128 * 128 * 27 * 4                       # at least one slab of the 128x128 layer
    f32 = lib.vs_conv_wgrad_multi_workspace_bytes(ctypes.addressof(arr), len(layers), VS_F32)
    k2 = (ops.WgradDesc * 2)(*layers[3:])
    k2_f32 = lib.vs_conv_wgrad_multi_workspace_bytes(ctypes.addressof(k2), 2, VS_F32)
    nobias = (ops.WgradDesc * 2)(desc(2, 24, 24, 24, 32, 32, VS_CONV_K2S2), desc(2, 48, 48, 48, 8, 8, VS_CONV_K2S2))
    k2_nobias = lib.vs_conv_wgrad_multi_workspace_bytes(ctypes.addressof(nobias), 2, VS_F32)
    # 32 -> 32 at 24^3: 2 x 6 x 6 x 2 = 144 tiles x 4 block pairs, <= 8 tiles per workgroup -> >= 18 slabs of 4 x 8 taps x 256 floats; plus the 8 -> 8 layer's
    assert k2_nobias >= 18 * 4 * 8 * 256 * 4 and k2_nobias % 16 == 0
    assert k2_nobias < k2_f32 <= k2_nobias + 2 * 256 * 32 * 8 + 512                    # + the two bias layers' partial sums
    assert f32 >= k2_f32 + 128 * 128 * 27 * 4 and f32 % 16 == 0
    assert lib.vs_conv_wgrad_multi(ctypes.addressof(arr), len(layers), None, 0, VS_BF16, 1e-5, None) == -1      # no workspace
    assert lib.vs_conv_wgrad_multi(ctypes.addressof(arr), len(layers), fake, 16, VS_BF16, 1e-5, None) == -4      # VS_EWORKSPACE
    # a weight used several times in one pass: descriptors with the same destination are summed — their slabs lie side by side
    one = (ops.WgradDesc * 1)(desc(1, 24, 24, 24, 32, 32, VS_CONV_K3, dw=fake))
    two = (ops.WgradDesc * 2)(desc(1, 24, 24, 24, 32, 32, VS_CONV_K3, dw=fake), desc(1, 12, 12, 12, 32, 32, VS_CONV_K3, dw=fake))
    for dt in (VS_BF16, VS_F32):
        assert lib.vs_conv_wgrad_multi_workspace_bytes(ctypes.addressof(two), 2, dt) > lib.vs_conv_wgrad_multi_workspace_bytes(ctypes.addressof(one), 1, dt) > 0
    mixed = (ops.WgradDesc * 2)(desc(1, 8, 8, 8, 32, 32, VS_CONV_K3, dw=fake), desc(1, 8, 8, 8, 32, 16, VS_CONV_K3, dw=fake))   # same destination, other layer
    assert lib.vs_conv_wgrad_multi_workspace_bytes(ctypes.addressof(mixed), 2, VS_BF16) == 0
    assert lib.vs_conv_wgrad_multi(ctypes.addressof(mixed), 2, fake, 1 << 24, VS_BF16, 1e-5, None) == -1
    bad = (ops.WgradDesc * 1)(desc(2, 8, 8, 8, 12, 8, VS_CONV_K3))                                               # channels not a multiple of 8
    assert lib.vs_conv_wgrad_multi_workspace_bytes(ctypes.addressof(bad), 1, VS_BF16) == 0
    assert lib.vs_conv_wgrad_multi(ctypes.addressof(bad), 1, fake, 1 << 20, VS_BF16, 1e-5, None) == -2
    # fused Dice-loss sum: scratch sizing and argument checks
    assert lib.vs_dice_loss_multi_scratch_doubles(2, 2, 2) == 3 * 2 * 2 * 2 + 2 * 2 * 5 * 256
    assert lib.vs_dice_loss_multi_fwd(None, None, None, 2, None, None, None, 2, 2, 64, 1, 2, 1e-4, None) == -1
    tp = (ctypes.c_void_p * 5)(*[fake] * 5)
    wp = (ctypes.c_float * 5)(*[1.0] * 5)
    assert lib.vs_dice_loss_multi_fwd(fake, ctypes.addressof(tp), ctypes.addressof(wp), 5, fake, fake, fake, 2, 2, 64, 1, 2, 1e-4, None) == -1   # k > 4
    assert lib.vs_dice_loss_multi_fwd(fake, ctypes.addressof(tp), ctypes.addressof(wp), 2, fake, fake, fake, 2, 2, 63, 1, 2, 1e-4, None) == -5   # voxels % 4


def test_host_helpers():
    from vae_segmentation_amd import ops
    assert [ops.cpad(c) for c in (1, 2, 8, 9, 16, 17, 32, 33, 256)] == [8, 8, 8, 16, 16, 32, 32, 64, 256]
    from tests import golden_util as G
    assert G.is_dead_bias("in_block.conv.0.bias") and G.is_dead_bias("up5.conv.1.conv.6.bias")
    assert not G.is_dead_bias("down1.conv.0.bias") and not G.is_dead_bias("out_block.bias") and not G.is_dead_bias("fc2.bias")
    with pytest.raises(RuntimeError):
        ops._require_cuda(torch.zeros(1))


def test_module_surface_on_cpu_is_constructible_but_not_runnable():
    import joint_model as M
    seg = M.Segmentation(1, 2, norm_type=1)
    vae = M.VAE(2, 2, norm_type=1, dim=128, spatial=96)
    assert vae.fc_mean.weight.shape == (128, 6912) and vae.fc2.weight.shape == (6912, 128)
    joint = M.Joint(models=[seg, vae])
    assert [n.split(".")[0] for n, _ in joint.named_parameters()][0] == "Seg"
    with pytest.raises(RuntimeError):
        joint({"x": torch.zeros(1, 1, 96, 96, 96)}, "x", "p", "r")
    bn = M.Segmentation(1, 2)                                # the constructors' default norm_type=2: BatchNorm3d holders, reference state_dict keys
    assert "in_block.conv.1.running_var" in bn.state_dict() and isinstance(M.DoubleConv(8, 8, norm_type=1, soft=True).conv[2], torch.nn.Softplus)
    gs = M.Segmentation(1, 2, norm_type=3)                   # GSNorm3d holders (joint_model.py:14-15): parameter-less, same state_dict keys as norm_type=1
    assert list(gs.state_dict().keys()) == list(M.Segmentation(1, 2, norm_type=1).state_dict().keys())
    with pytest.raises(ValueError):
        M.Segmentation(1, 2, norm_type=4)
    assert M.Segmentation(1, 3, norm_type=1).out_block.weight.shape[0] == 3 and M.VAE(5, 5, norm_type=1, dim=128).in_block.conv[0].weight.shape[1] == 5
    with pytest.raises(NotImplementedError):                 # the probabilities travel in one 8-channel fragment
        M.Segmentation(1, 9, norm_type=1)


WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from vae_segmentation_amd.ddp import FlatGradSync
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
params = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(7))]
if rank == 1:
    for p in params: p.data.add_(1.0)          # replicas start different; broadcast must fix that
sync = FlatGradSync(params)
sync.broadcast_parameters(0)
ref0 = [torch.randn(3, 4, generator=torch.Generator().manual_seed(10)), torch.randn(7, generator=torch.Generator().manual_seed(11))]
ref1 = [torch.randn(3, 4, generator=torch.Generator().manual_seed(20)), torch.randn(7, generator=torch.Generator().manual_seed(21))]
mine = ref0 if rank == 0 else ref1
views = sync([g.clone() for g in mine])
for v, a, b in zip(views, ref0, ref1):
    assert torch.allclose(v, (a + b) / 2, atol=1e-6), "rank %%d: averaged gradient mismatch" %% rank
chk = torch.cat([p.data.reshape(-1) for p in params])
gathered = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(gathered, chk)
assert torch.equal(gathered[0], gathered[1]), "parameters differ across ranks after broadcast"
dist.destroy_process_group()
print("rank %%d ok" %% rank)
"""


def test_ddp_flat_grad_sync_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % REPO)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert all("ok" in o for o in outs)


def test_capture_safe_accumulators_detects_graphs_of_earlier_passes():
    """train.capture_safe_accumulators (GraphedStep's guard): gradient accumulators nobody owns are re-created under the caller's stream,
    ones it tagged earlier are accepted, ones an older autograd graph keeps alive are reported instead of being captured over."""
    import torch
    from vae_segmentation_amd import train as T
    ps = [torch.nn.Parameter(torch.randn(3)) for _ in range(3)] + [torch.nn.Parameter(torch.randn(2), requires_grad=False)]
    a = T.capture_safe_accumulators(ps)
    assert len(a) == 3
    b = T.capture_safe_accumulators(ps)                      # the first call's nodes are alive and tagged
    assert all(x is y for x, y in zip(a, b))
    del a, b
    loss = sum((p * 2).sum() for p in ps[:3])
    with pytest.raises(RuntimeError, match="earlier eager pass"):
        T.capture_safe_accumulators(ps)
    del loss
    assert len(T.capture_safe_accumulators(ps)) == 3


def test_main_target_parses_every_reference_flag_with_the_reference_defaults(capsys):
    """VERDICT r04 (missing 7): a reference command line must not abort in argparse.  The table is the reference's parser
    (main_target.py:29-81: option strings, defaults) as data."""
    import main_target
    ref_defaults = {
        "target_phase": "arterial", "GPU": "0,1,2,3", "batch_size": 4, "max_epoch": 1600, "save_epoch": 50, "eval_epoch": 50, "turn_epoch": -1,
        "softrelu": 0, "method": "vae_train", "data_root": "../nih_data/numpy_data/", "val_data_root": "../nih_data/numpy_data/",
        "pseudo_data_root": "../nih_data/numpy_data/", "data_path": "Multi_all.json", "train_list": "NIH_train", "val_list": "NIH_val",
        "pseudo_list": None, "load_prefix": None, "checkpoint_name": "best_model.ckpt", "load_prefix_vae": None, "load_prefix_encoder": None,
        "load_prefix_joint": None, "pan_index": "1", "pseudo_pan_index": "1", "lambda_vae": 0.1, "lambda_vae_warmup": 0, "lr_seg": 1e-2, "lr_vae": 0,
        "test_only": False, "resume": False, "save_more_reference": False, "save_eval_result": False, "no_aug": False, "only_pseudo": False,
        "fix_layer": False, "use_confident_binarize": False, "analysis_figure_name": None, "pseudo_save_epoch": 0, "domain_loss_type": 0,
        "vae_mont_number": 1, "vae_forward_scale": 0.0, "vae_decoder_dropout": 0.0, "seg_dropout": 0.0, "val_finetune": 0, "lr_finetune": 1e-2,
        "tag": False, "from_scratch": False, "adam": False, "kl": False, "alpha": 0.995, "update_every_iteration": False,
        "generate_bounding_boxes": False, "shift": 0}
    a = main_target.parse(["run0"])
    assert a.prefix == "run0"
    for k, v in ref_defaults.items():
        assert getattr(a, k) == v, (k, getattr(a, k), v)
    # a command line that uses every remaining option string of the reference at once (pseudo_list aside: see below)
    a = main_target.parse(["run1", "-P", "venous", "-G", "0", "-b", "2", "-E", "10", "--save_epoch", "5", "--eval_epoch", "5", "--turn_epoch", "2", "-S", "1",
                           "-M", "domain_adaptation", "--data_root", "d", "--val_data_root", "v", "--pseudo_data_root", "p", "-l", "x.json",
                           "--train_list", "T", "--val_list", "V", "--load_prefix", "a", "--checkpoint_name", "c.ckpt", "--load_prefix_vae", "b",
                           "--load_prefix_encoder", "e", "--load_prefix_joint", "j", "--pan_index", "1,2", "--pseudo_pan_index", "1", "--lambda_vae", "1.0",
                           "--lambda_vae_warmup", "3", "--lr_seg", "1e-3", "--lr_vae", "0", "--resume", "--save_more_reference", "--save_eval_result",
                           "--no_aug", "--fix_layer", "--use_confident_binarize", "--analysis_figure_name", "fig", "--pseudo_save_epoch", "1",
                           "--domain_loss_type", "8", "--vae_mont_number", "2", "--vae_forward_scale", "0.35", "--vae_decoder_dropout", "0.1",
                           "--seg_dropout", "0.1", "--val_finetune", "1", "--lr_finetune", "1e-3", "--tag", "--from_scratch", "--adam", "--kl",
                           "--alpha", "0.99", "--update_every_iteration", "--generate_bounding_boxes", "--shift", "4"])
    assert a.fix_layer and a.from_scratch and a.vae_mont_number == 2 and a.vae_forward_scale == 0.35 and a.shift == 4 and a.load_prefix_encoder == "e"
    assert "accepted and ignored" in capsys.readouterr().err                       # the dump / figure flags say what happens to them
    with pytest.raises(SystemExit, match="inconsistent flags"):                                            # main_target.py:145: more than one pass needs a forward scale
        main_target.parse(["r", "--vae_mont_number", "2"])
    a = main_target.parse(["r", "-M", "domain_adaptation", "--pseudo_list", "NIH_pseudo", "--pseudo_pan_index", "1"])     # main_target.py:228-307,615-692
    assert a.pseudo_list == "NIH_pseudo"
    with pytest.raises(SystemExit, match="pseudo_list"):                           # the reference's other methods never read the second loader (:689)
        main_target.parse(["r", "-M", "domain_adaptation_dis", "--pseudo_list", "NIH_pseudo"])


def test_synthetic_matches_the_oracle_generators():
    """bench.py builds its workload from vae_segmentation_amd/synthetic.py (product code, no oracle import in the benchmark's set-up); the values are
    the oracle's fixtures' (oracle/ref_cpu.py), so `final_loss` in the bench line stays comparable with the goldens' inputs."""
    import torch
    from oracle import ref_cpu as O
    from vae_segmentation_amd import synthetic as S
    assert torch.equal(S.synthetic_image(2, 12, 7), O.synthetic_image(2, 12, 7))
    assert torch.equal(S.synthetic_label(2, 12, 9), O.synthetic_label(2, 12, 9))
    a, b = O.Segmentation(1, 2, norm_type=1), O.Segmentation(1, 2, norm_type=1)
    S.deterministic_fill_(a, seed=3)
    O.deterministic_fill_(b, seed=3)
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))
    src = open(os.path.join(REPO, "bench.py")).read()
    build_src = src[src.index("def build("):src.index("def usable_cores")]
    assert "oracle" not in build_src.replace("the oracle is imported by cpu_baseline only", "")


def test_config_struct_layout_and_single_entry_point():
    """include/vaeseg.h vs_config: the Python binding's struct is the library's, the environment seeds it once, vs_set_config is the only writer and
    validates what it is given (no GPU needed)."""
    from vae_segmentation_amd import ops
    import ctypes
    assert ops.lib.vs_config_bytes() == ctypes.sizeof(ops.VsConfig)
    cfg = ops.get_config()
    assert cfg["chain"] in (0, 1) and cfg["wgrad_wgs"] >= 1 and cfg["k3x_ck"] in (8, 16)
    old = ops.set_config(wgrad_mpack=0, wgrad_big_min_voxels=123)
    try:
        now = ops.get_config()
        assert now["wgrad_mpack"] == 0 and now["wgrad_big_min_voxels"] == 123 and now["chain"] == cfg["chain"]
        assert ops.lib.vs_conv_k3_chain_supported(2, 6, 6, 6, 128, 1) == (1 if cfg["chain"] else 0)
        ops.set_config(chain=0)
        assert ops.lib.vs_conv_k3_chain_supported(2, 6, 6, 6, 128, 1) == 0
        ops.set_config(chain=cfg["chain"])
        with pytest.raises(KeyError):
            ops.set_config(no_such_switch=1)
        with pytest.raises(RuntimeError):
            ops.set_config(k3x_ck=12)                     # rejected by the library, nothing installed
        assert ops.get_config()["k3x_ck"] == cfg["k3x_ck"]
    finally:
        ops.set_config(**old)
    assert ops.get_config() == cfg
    src = "".join(open(os.path.join(REPO, "vae_segmentation_amd", "csrc", f)).read() for f in os.listdir(os.path.join(REPO, "vae_segmentation_amd", "csrc"))
                  if f.endswith((".h", ".hip", ".inc")) and f != "config.hip")
    assert src.count("getenv(") == 2, "launchers read vs_cfg(), not the environment (the two left sit inside the VS_G3B_ABLATE diagnostic build)"


def test_wgrad_xcd_rank_model_is_a_bijection():
    """csrc/wgrad.hip g3_xcd_rank (round 6): a layer's workgroups [b0, b0 + count) of the grouped weight-gradient grid are re-ranked so that XCD x (= workgroup
    id mod 8) owns one contiguous run of ranks.  The same arithmetic in Python: every rank is hit exactly once for any start and count, and the ranks of one XCD
    are consecutive (what puts the channel-block pairs of a tile, rank = k-split * pairs + pair, into one L2)."""
    import random

    def rank(b0, local, count):
        x = (b0 + local) & 7
        start = 0
        for xx in range(8):
            first = (xx - b0) & 7
            cnt = (count - first + 7) >> 3 if first < count else 0
            if xx < x:
                start += cnt
        return start + ((local - ((x - b0) & 7)) >> 3)

    src = open(os.path.join(REPO, "vae_segmentation_amd", "csrc", "wgrad.hip")).read()
    assert "const int first = (xx - b0) & 7;" in src and "return start + ((local - ((x - b0) & 7)) >> 3);" in src      # the model is the kernel's formula
    rng = random.Random(1)
    for _ in range(300):
        b0, count = rng.randrange(0, 5000), rng.randrange(16, 1500)
        ranks = [rank(b0, l, count) for l in range(count)]
        assert sorted(ranks) == list(range(count)), (b0, count)
        for x in range(8):
            mine = sorted(r for l, r in enumerate(ranks) if (b0 + l) & 7 == x)
            assert mine == list(range(mine[0], mine[0] + len(mine))) if mine else True


def test_isa_phase_counts_finds_the_tile_loop_of_k3t():
    """tools/isa_phase_counts.py (no GPU: device-only compile to assembly): the persistent loop of the 8-channel full-resolution kernel holds the tile's 72 MFMAs
    (9 k-groups x 8 output rows), two barriers per tile plus the statistics flush's, and a vector instruction count in the range DESIGN.md section 9 reasons with."""
    import re
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "isa_phase_counts.py"), os.path.join(REPO, "vae_segmentation_amd", "csrc", "igemm_k3_bf16.hip"),
                          "k3t_kernel<0, false, 8, true, unsigned short, false>"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"loop: (\d+) instructions, (\d+) barriers", out.stdout)
    assert m and int(m.group(2)) == 3, out.stdout
    tot = [ln for ln in out.stdout.splitlines() if ln.startswith("loop total")][0].split()
    valu, packed, mfma = int(tot[2]), int(tot[3]), int(tot[4])
    assert mfma == 72 and 300 <= valu + packed <= 800, out.stdout
