#!/usr/bin/env python3
"""Target-domain trainer — every flag of the reference's main_target.py:29-81 parses (same names, short options and defaults), native step.
Flags that only drive outputs this entry point does not produce (figures, extra reference dumps: --save_more_reference, --save_eval_result,
--analysis_figure_name, --generate_bounding_boxes, -P) are accepted with a warning.  --pseudo_list (with --pseudo_data_root / --pseudo_pan_index, main_target.py:228-307) adds the second, pseudo-labelled
loader and switches domain_adaptation to the step of main_target.py:615-692 (its own loss ladder, teacher re-loaded from the student, and a logged-only
forward on one pseudo-labelled batch per iteration).  Methods (all native): vae_train, domain_adaptation (student/teacher Joint nets, binarised or confident pseudo-labels,
domain_loss_type 0 / 8 / 9 / 11-16, --only_pseudo, --turn_epoch, --lambda_vae_warmup, optional KL term, optional EMA teacher, test-time
training with --val_finetune), discriminator_train, domain_adaptation_dis.  Uses the
utils/evaluation.py epsilon (1e-6), as main_target.py does (it imports avg_dsc from there, main_target.py:23)."""
import argparse

from vae_segmentation_amd import driver


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("prefix", help="prefix")
    p.add_argument("-P", "--target_phase", default="arterial", help="accepted (main_target.py:30,130: read, never used by the reference's loop)")
    p.add_argument("-G", "--GPU", default="0,1,2,3", help="kept for CLI compatibility; ranks come from torchrun")
    p.add_argument("-b", "--batch_size", type=int, default=4)
    p.add_argument("-E", "--max_epoch", type=int, default=1600)
    p.add_argument("--save_epoch", type=int, default=50)
    p.add_argument("--eval_epoch", type=int, default=50)
    p.add_argument("--turn_epoch", type=int, default=-1)
    p.add_argument("-S", "--softrelu", type=int, default=0)
    p.add_argument("-M", "--method", default="vae_train")
    p.add_argument("-R", "--data_root", default="../nih_data/numpy_data/")
    p.add_argument("-V", "--val_data_root", default="../nih_data/numpy_data/")
    p.add_argument("-l", "--data_path", default="Multi_all.json")
    p.add_argument("--pseudo_data_root", default="../nih_data/numpy_data/")
    p.add_argument("-t", "--train_list", default="NIH_train")
    p.add_argument("-v", "--val_list", default="NIH_val")
    p.add_argument("--pseudo_list", default=None, help="case list of the second, pseudo-labelled loader: domain_adaptation then runs the step of main_target.py:615-692")
    p.add_argument("--load_prefix", default=None)
    p.add_argument("--checkpoint_name", default="best_model.ckpt")
    p.add_argument("--load_prefix_vae", default=None)
    p.add_argument("--load_prefix_encoder", default=None, help="discriminator checkpoint: the whole model of discriminator_train, else model.Dis (main_target.py:384-390)")
    p.add_argument("--load_prefix_joint", default=None)
    p.add_argument("--pan_index", default="1")
    p.add_argument("--pseudo_pan_index", default="1")
    p.add_argument("--lambda_vae", type=float, default=0.1)
    p.add_argument("--lambda_vae_warmup", type=int, default=0)
    p.add_argument("--lr_seg", type=float, default=1e-2)
    p.add_argument("--lr_vae", type=float, default=0)
    p.add_argument("--domain_loss_type", type=int, default=0)
    p.add_argument("--kl", action="store_true")
    p.add_argument("--use_confident_binarize", action="store_true")
    p.add_argument("--pseudo_save_epoch", type=int, default=0)
    p.add_argument("--update_every_iteration", action="store_true")
    p.add_argument("--alpha", type=float, default=0.995)
    p.add_argument("--seg_dropout", type=float, default=0.0)
    p.add_argument("--vae_decoder_dropout", type=float, default=0.0)
    p.add_argument("--val_finetune", type=int, default=0, help="test-time training iterations per validation case (main_target.py:72)")
    p.add_argument("--lr_finetune", type=float, default=1e-2)
    p.add_argument("--only_pseudo", action="store_true")
    p.add_argument("--test_only", action="store_true")
    p.add_argument("--adam", action="store_true")
    p.add_argument("--resume", action="store_true", help="accepted (main_target.py:58,134: read, never used by the reference either)")
    p.add_argument("--save_more_reference", action="store_true", help="accepted; the extra reference volumes are not written")
    p.add_argument("--save_eval_result", action="store_true", help="accepted; per-case result volumes are not written")
    p.add_argument("--no_aug", action="store_true", help="no spatial augmentation of the training samples (main_target.py:61,207)")
    p.add_argument("--fix_layer", action="store_true", help="joint_train / domain_adaptation: train only Seg.up5 and Seg.out_block (main_target.py:400-406)")
    p.add_argument("--analysis_figure_name", default=None, help="accepted; scatter plots are not drawn")
    p.add_argument("--vae_mont_number", type=int, default=1, help="domain_adaptation: forward passes averaged per step (main_target.py:530-603)")
    p.add_argument("--vae_forward_scale", type=float, default=0.0, help="Joint(vae_forward_scale=...) (main_target.py:324)")
    p.add_argument("--tag", action="store_true", help="lambda_vae /= 10 at every pseudo-label refresh (main_target.py:635,712)")
    p.add_argument("--from_scratch", action="store_true", help="domain_adaptation: the loaded weights go to the frozen teacher, the student "
                   "starts from its initialisation (main_target.py:75,360-370,427)")
    p.add_argument("--generate_bounding_boxes", action="store_true", help="accepted (main_target.py:80,169: asserted, never used)")
    p.add_argument("--shift", type=int, default=0, help="CropResize(shift=...) of the training crops (main_target.py:81,204)")
    driver.add_native_flags(p)
    a = p.parse_args(argv)
    driver.check_target_flags(a)
    return a


if __name__ == "__main__":
    driver.run(parse(), side="target")
