"""GPU: the main_source.py / main_target.py entry points run end to end on synthetic volumes (tiny budgets)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, cwd):
    out = subprocess.run([sys.executable] + args, cwd=cwd, env=dict(os.environ, PYTHONPATH=REPO), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return out.stdout


def test_main_source_joint_train_then_main_target_domain_adaptation(tmp_path):
    common = ["--size", "64", "-b", "1", "-E", "1", "--eval_epoch", "1", "--save_epoch", "1", "--synthetic_train", "2",
              "--synthetic_val", "1", "--max_iters", "2", "--display_freq", "1"]
    out = _run([os.path.join(REPO, "main_source.py"), "src", "-M", "joint_train"] + common, str(tmp_path))
    assert "Finished Training" in out and "validation result" in out
    ck = tmp_path / "3dmodel" / "src" / "model_epoch1.ckpt"
    assert ck.exists()
    import torch
    blob = torch.load(str(ck), map_location="cpu")
    assert set(blob) == {"epoch", "model_state_dict", "optimizer_state_dict"}
    assert any(k.startswith("Seg.in_block.conv.0.") for k in blob["model_state_dict"])
    assert json.load(open(tmp_path / "tensorboard" / "src" / "score_0.json"))
    assert "graph replay" in out and "train step captured as a HIP graph" in out          # the entry point runs the benchmarked (graph-replayed) path
    out = _run([os.path.join(REPO, "main_target.py"), "tgt", "-M", "domain_adaptation", "--load_prefix_joint", "src",
                "--checkpoint_name", "model_epoch1.ckpt", "--domain_loss_type", "8", "--train_first_epoch", "--pseudo_save_epoch", "1",
                "--update_every_iteration"] + common, str(tmp_path))
    assert "Finished Training" in out and "graph replay" in out
    # --pseudo_list (main_target.py:615-692): its own loss ladder, teacher re-loaded from the student, one logged-only pseudo-labelled batch per iteration
    for extra in (["--domain_loss_type", "8"], ["--lambda_vae", "2000"], ["--tag", "--no_graph"]):
        out = _run([os.path.join(REPO, "main_target.py"), "tgt_ps", "-M", "domain_adaptation", "--load_prefix_joint", "src",
                    "--checkpoint_name", "model_epoch1.ckpt", "--train_first_epoch", "--pseudo_save_epoch", "1", "--pseudo_list", "synthetic_pseudo"] + extra + common,
                   str(tmp_path))
        assert "Finished Training" in out and "dice_loss_pseudo" in out and "final_loss_pseudo" in out and "recon_loss_pseudo" in out
        assert ("graph replay" in out) == ("--no_graph" not in extra)
    # epoch 0 of domain_adaptation only validates in the reference (main_target.py:506); dropout > 0 falls back to eager launches
    out = _run([os.path.join(REPO, "main_target.py"), "tgt_do", "-M", "domain_adaptation", "--load_prefix_joint", "src",
                "--checkpoint_name", "model_epoch1.ckpt", "--seg_dropout", "0.1", "--train_first_epoch"] + common, str(tmp_path))
    assert "Finished Training" in out and "(eager)" in out
    # test-time training of each validation case (main_target.py --val_finetune, scripts/target/domain_msd_dh_ft1.bash)
    out = _run([os.path.join(REPO, "main_target.py"), "tgt_ft", "-M", "domain_adaptation", "--load_prefix_joint", "src",
                "--checkpoint_name", "model_epoch1.ckpt", "--domain_loss_type", "8", "--val_finetune", "1", "--test_only"] + common,
               str(tmp_path))
    assert "validation result without finetuning" in out and "Finished Training" in out
    assert json.load(open(tmp_path / "tensorboard" / "tgt_ft" / "score_0.json"))


def test_remaining_methods_run_end_to_end(tmp_path):
    """sep_joint_train / embed_train / refine_vae (main_source.py) and discriminator_train / domain_adaptation_dis
    (main_target.py) through the native entry points on synthetic volumes."""
    common = ["--size", "64", "-b", "1", "-E", "1", "--eval_epoch", "1", "--save_epoch", "1", "--synthetic_train", "2",
              "--synthetic_val", "1", "--max_iters", "2", "--display_freq", "1"]
    for script, method, extra in (("main_source.py", "sep_joint_train", []), ("main_source.py", "embed_train", []),
                                  ("main_source.py", "refine_vae", []), ("main_target.py", "discriminator_train", []),
                                  ("main_target.py", "domain_adaptation_dis", ["--train_first_epoch"])):
        out = _run([os.path.join(REPO, script), "m_" + method, "-M", method] + common + extra, str(tmp_path))
        assert "Finished Training" in out and "loss:" in out, (method, out[-1500:])


def test_real_data_cases_through_the_device_pipeline(tmp_path):
    """--real_data: merge.npy cases listed in lists/<data_path> (main_source.py:123-131,186-243) are loaded, relabelled, cropped / resized,
    augmented, clipped and centred on the device (data_gpu.py) and fed to the graph-replayed seg_train step; validation uses the
    un-augmented chain."""
    import numpy as np
    rng = np.random.RandomState(0)
    (tmp_path / "data").mkdir(); (tmp_path / "lists").mkdir()
    names = []
    for i, shape in enumerate([(40, 48, 44), (52, 40, 46), (44, 44, 60), (48, 50, 42)]):
        merge = np.zeros(shape + (2,), np.float32)
        merge[..., 0] = rng.randn(*shape) * 250 + 40
        merge[10:30, 12:34, 8:30, 1] = 1
        np.save(tmp_path / "data" / ("case%d_merge.npy" % i), merge)
        names.append("case%d_merge.npy" % i)
    json.dump({"NIH_train": names[:3], "NIH_val": names[3:]}, open(tmp_path / "lists" / "Multi_all.json", "w"))
    out = _run([os.path.join(REPO, "main_source.py"), "real", "-M", "seg_train", "--real_data", "-R", str(tmp_path / "data"), "-V", str(tmp_path / "data"),
                "--size", "32", "-b", "1", "-E", "2", "--eval_epoch", "1", "--save_epoch", "1", "--display_freq", "1"], str(tmp_path))
    assert "Finished Training" in out and "graph replay" in out and "validation result" in out
    assert out.count("loss:") >= 6                                   # 3 cases x 2 epochs
    assert json.load(open(tmp_path / "tensorboard" / "real" / "score_0.json"))
    # main_target.py:228-307: the second, pseudo-labelled case list of a --pseudo_list run through the same device pipeline (its own root and structure index)
    json.dump({"NIH_train": names[:3], "NIH_val": names[3:], "NIH_pseudo": names[1:3]}, open(tmp_path / "lists" / "Multi_all.json", "w"))
    out = _run([os.path.join(REPO, "main_target.py"), "realda", "-M", "domain_adaptation", "--real_data", "-R", str(tmp_path / "data"), "-V", str(tmp_path / "data"),
                "--pseudo_list", "NIH_pseudo", "--pseudo_data_root", str(tmp_path / "data"), "--pseudo_pan_index", "1", "--train_first_epoch",
                "--size", "64", "-b", "1", "-E", "1", "--eval_epoch", "1", "--save_epoch", "1", "--display_freq", "1"], str(tmp_path))
    assert "Finished Training" in out and out.count("dice_loss_pseudo") >= 3      # 3 training cases, the 2 pseudo cases cycled


def test_pseudo_list_step_losses_follow_the_reference_ladder():
    """main_target.py:636-651 — the final loss of a --pseudo_list iteration: type 8 with the stepped lambda (both of its arms), lambda_vae >= 1000
    (recon * lambda / 10000), else lambda * recon + fake; host and device forms of the type-8 schedule agree; the pseudo-labelled batch's terms
    (main_target.py:677-684) come without an autograd graph."""
    import torch
    sys.path.insert(0, REPO)
    import bench
    from vae_segmentation_amd import train as T
    joint, img, lab, teacher = bench.build(64, "fp32", 0, batch=1, teacher=True)

    def expect(r, f, dlt, lam):
        if dlt == 8:
            cur = lam * (0.6 if r < 0.15 else 1.2 if r < 0.225 else 2.0 if r < 0.3 else 3.0)
            return r + f / cur if cur > 1 else cur * r + f
        return r * lam / 10000 if lam >= 1000 else lam * r + f

    for dlt, lam in ((8, 1.0), (8, 0.2), (0, 2000.0), (0, 0.5)):
        for host in (True, False) if dlt == 8 else (True,):
            final, aux = T.domain_adaptation_pseudo_losses(joint, teacher, img, lab, lambda_vae=lam, domain_loss_type=dlt, host_schedule=host)
            r, f = aux["recon_loss"].item(), aux["dice_loss_fake"].item()
            assert abs(final.item() - expect(r, f, dlt, lam)) < 1e-5 * max(1.0, abs(final.item())), (dlt, lam, host)
            assert final.requires_grad
    out = T.pseudo_batch_losses(joint, img, lab)
    assert set(out) == {"recon_loss_pseudo", "dice_loss_pseudo", "final_loss_pseudo"} and not any(v.requires_grad for v in out.values())
    assert out["final_loss_pseudo"].item() == out["dice_loss_pseudo"].item()
