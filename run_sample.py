#!/usr/bin/env python3
"""CLI boundary — same flags and step order as reference run_sample.py:8-137, for the three
label-generation steps this repository implements (make_cam, make_ins_seg, make_sem_seg).

The training / CRF / evaluation steps of the reference (train_cam, eval_cam, cam_to_ir_label,
train_irn, eval_ins_seg, eval_sem_seg) are outside the hot-path scope; their
`--*_pass` flags are accepted so existing command lines keep working, and asking for one of them
is an error rather than a silent skip.  Weights are inputs: --cam_weights_name / --irn_weights_name
must point at checkpoints written by the reference's training steps (or any state dict with the
same keys).
"""
import argparse
import os

from irn_amd.misc import pyutils


def _flag(v):
    """The reference declares pass flags without type= (a CLI value arrives as str and fails its
    `is True` test); parse the usual spellings instead."""
    if isinstance(v, bool):
        return v
    return str(v).strip().lower() in ("1", "true", "yes", "y")


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--num_workers", default=(os.cpu_count() or 2) // 2, type=int)
    p.add_argument("--voc12_root", required=True, type=str)
    p.add_argument("--train_list", default="voc12/train_aug.txt", type=str)
    p.add_argument("--val_list", default="voc12/val.txt", type=str)
    p.add_argument("--infer_list", default="voc12/train.txt", type=str)
    p.add_argument("--chainer_eval_set", default="train", type=str)
    p.add_argument("--cam_network", default="net.resnet50_cam", type=str)
    p.add_argument("--cam_scales", default=(1.0, 0.5, 1.5, 2.0), type=float, nargs="+")
    p.add_argument("--irn_network", default="net.resnet50_irn", type=str)
    p.add_argument("--beta", default=10, type=float)
    p.add_argument("--exp_times", default=8, type=int)
    p.add_argument("--ins_seg_bg_thres", default=0.25, type=float)
    p.add_argument("--sem_seg_bg_thres", default=0.25, type=float)
    p.add_argument("--radius", default=5, type=int,
                   help="random-walk radius of the label steps (not a flag of the reference, which hard-codes 5 at "
                        "step/make_sem_seg_labels.py:41 and step/make_ins_seg_labels.py:135; 10 = BASELINE configs[2])")
    # training / CRF / evaluation hyper-parameters of the reference (run_sample.py:25-40): accepted so that an existing
    # command line keeps parsing; the steps that read them are not part of this build
    for name, default, typ in (("cam_crop_size", 512, int), ("cam_batch_size", 16, int), ("cam_num_epoches", 5, int),
                               ("cam_learning_rate", 0.1, float), ("cam_weight_decay", 1e-4, float),
                               ("cam_eval_thres", 0.15, float), ("conf_fg_thres", 0.30, float), ("conf_bg_thres", 0.05, float),
                               ("irn_crop_size", 512, int), ("irn_batch_size", 32, int), ("irn_num_epoches", 3, int),
                               ("irn_learning_rate", 0.1, float), ("irn_weight_decay", 1e-4, float)):
        p.add_argument("--" + name, default=default, type=typ, help="accepted and ignored (training / CRF side of the reference)")
    p.add_argument("--worker_devices", default="", type=str,
                   help="device ordinal of every worker process, e.g. 0,1,2,3 (default: one per visible GPU like the reference; "
                        "0,0 = two workers sharing GPU 0)")
    p.add_argument("--cam_batch", default=0, type=int, help="images of one size per CAM trunk pass (0 = 8)")
    p.add_argument("--irn_batch", default=0, type=int, help="images per IRNet trunk pass (0 = 8)")
    p.add_argument("--keep_cams_on_device", default=True, type=_flag,
                   help="hand the CAMs of make_cam to the label steps in device memory when they run in the same process "
                        "(the .npy files are written all the same)")
    p.add_argument("--keep_edges_on_device", default=True, type=_flag,
                   help="the label step that runs first leaves every image's boundary / displacement maps on the device and the "
                        "other one skips its IRNet forward (the reference runs EdgeDisplacement in both steps)")
    p.add_argument("--walk_batch", default=0, type=int,
                   help="images per random-walk launch (not in the reference); 0 = the step's default (64 sem-seg, 32 ins-seg)")
    p.add_argument("--walk_accel", default=None, type=int, choices=(0, 1),
                   help="random-walk schedule (not in the reference): 1 = x.T^(2^exp_times) as a truncated Chebyshev series of the "
                        "operator (84 applications at exp_times 8; the default), 0 = the reference's own 2^exp_times applications "
                        "(misc/indexing.py:136-137).  Unset: the environment variable IRN_WALK_ACCEL, else 1")
    p.add_argument("--walk_accel_tol_exp", default=0, type=int,
                   help="truncation bound 10^-e of the series (0 = the library default, e = 7; 6 = 78 applications: +7 %%, may flip an argmax at an exact tie)")
    p.add_argument("--deterministic", default=None, type=int, choices=(0, 1),
                   help="1 (the default) = bit-reproducible backbones: any worker layout writes identical files (tuned shapes on a "
                        "find database without split-K solvers, MIOpen's deterministic attribute elsewhere); 0 = the last 2-4 %% of "
                        "speed, outputs then move by ~1e-5 from run to run (not in the reference).  Unset: the environment "
                        "variable IRN_DETERMINISTIC, else 1")
    p.add_argument("--split_gemm", default=None, type=int, choices=(0, 1),
                   help="1 (default): the trunk's 1x1 convolutions as fp16 hi/lo split products with fp32 accumulation (fp32-level "
                        "accuracy, ~15 %% faster end to end); 0: plain fp32 GEMMs (also IRN_SPLIT_GEMM)")
    p.add_argument("--step_timeout", default=0.0, type=float,
                   help="seconds a step may take in its worker processes before the pool is stopped and the step raises "
                        "(0 = no limit; also IRN_STEP_TIMEOUT_S)")
    p.add_argument("--log_name", default="sample_train_eval", type=str)
    p.add_argument("--cam_weights_name", default="sess/res50_cam.pth", type=str)
    p.add_argument("--irn_weights_name", default="sess/res50_irn.pth", type=str)
    p.add_argument("--cam_out_dir", default="result/cam", type=str)
    p.add_argument("--ir_label_out_dir", default="result/ir_label", type=str)
    p.add_argument("--sem_seg_out_dir", default="result/sem_seg", type=str)
    p.add_argument("--ins_seg_out_dir", default="result/ins_seg", type=str)
    p.add_argument("--edge_out_dir", default=None, type=str,
                   help="verification aid (not in the reference): also write every image's boundary / displacement maps "
                        "(<name>.npy = {'edge' [1,h,w], 'dp' [2,h,w]}) as the label steps used them")
    for name, default in (("train_cam_pass", False), ("make_cam_pass", True), ("eval_cam_pass", False),
                          ("cam_to_ir_label_pass", False), ("train_irn_pass", False), ("make_ins_seg_pass", True),
                          ("eval_ins_seg_pass", False), ("make_sem_seg_pass", True), ("eval_sem_seg_pass", False)):
        p.add_argument("--" + name, default=default, type=_flag)
    return p


OUT_OF_SCOPE = ("train_cam_pass", "eval_cam_pass", "cam_to_ir_label_pass", "train_irn_pass", "eval_ins_seg_pass",
                "eval_sem_seg_pass")


def main(argv=None):
    args = build_parser().parse_args(argv)
    for name in OUT_OF_SCOPE:
        if getattr(args, name):
            raise SystemExit("--%s: this step is not part of the MI355X hot-path build; run it with the reference" % name)
    for d in (args.cam_out_dir, args.sem_seg_out_dir, args.ins_seg_out_dir):
        os.makedirs(d, exist_ok=True)
    if args.split_gemm is not None:
        os.environ["IRN_SPLIT_GEMM"] = str(int(args.split_gemm))           # workers read it when they import the trunk
        from irn_amd.net import resnet50 as _r50
        _r50.SPLIT_GEMM = bool(args.split_gemm)
    if args.deterministic is not None:
        os.environ["IRN_DETERMINISTIC"] = str(int(args.deterministic))      # read by every process that sets MIOpen up (workers inherit it)
    pyutils.Logger(args.log_name + ".log")
    print(vars(args))
    if args.make_cam_pass is True:
        from irn_amd.step import make_cam
        timer = pyutils.Timer("step.make_cam:")
        make_cam.run(args)
    if args.make_ins_seg_pass is True:
        from irn_amd.step import make_ins_seg_labels
        timer = pyutils.Timer("step.make_ins_seg_labels:")
        make_ins_seg_labels.run(args)
    if args.make_sem_seg_pass is True:
        from irn_amd.step import make_sem_seg_labels
        timer = pyutils.Timer("step.make_sem_seg_labels:")  # noqa: F841
        make_sem_seg_labels.run(args)
    from irn_amd.step import _common
    _common.shutdown_workers()           # the per-GPU workers served every pass above


if __name__ == "__main__":
    main()
