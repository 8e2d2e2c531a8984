"""The reference's command-line surface (reference main.py:30-103, 55 tf.app.flags) as a table-driven
argparse front end that accepts the identical spellings, including absl's `--flag` / `--noflag`
booleans (reference runGan.py:149,201,271), plus the two training recipes of runGan.py."""
import argparse
from types import SimpleNamespace

# (name, type, default, help)
FLAG_TABLE = [
    ("rand_seed", int, 1, "random seed"),
    # directories
    ("input_dir_LR", str, None, "LR input directory (inference)"),
    ("input_dir_len", int, -1, "number of input frames for inference, -1 = all"),
    ("input_dir_HR", str, None, "HR input directory (inference, down-sampled on the fly)"),
    ("mode", str, "inference", "train | inference"),
    ("output_dir", str, None, "output directory (checkpoints / images)"),
    ("output_pre", str, "", "sub-folder for the images"),
    ("output_name", str, "output", "prefix of the output images"),
    ("output_ext", str, "jpg", "output image format"),
    ("summary_dir", str, None, "directory for logs"),
    # models
    ("checkpoint", str, None, "checkpoint to restore"),
    ("num_resblock", int, 16, "residual blocks in the generator"),
    ("pre_trained_model", bool, False, "True: load weights only; False: resume the whole training state"),
    ("vgg_ckpt", str, None, "vgg19 checkpoint"),
    # machine
    ("cudaID", str, "0", "visible device id(s)"),
    ("queue_thread", int, 6, "loader threads"),
    ("name_video_queue_capacity", int, 512, "filename queue capacity"),
    ("video_queue_capacity", int, 256, "video queue capacity"),
    ("video_queue_batch", int, 2, "shuffle_batch queue capacity"),
    # training data
    ("RNN_N", int, 10, "recurrent length"),
    ("batch_size", int, 4, "batch size"),
    ("flip", bool, True, "random flip augmentation"),
    ("random_crop", bool, True, "random crop"),
    ("movingFirstFrame", bool, True, "constant-moving first frame augmentation"),
    ("crop_size", int, 32, "LR crop size"),
    ("input_video_dir", str, "", "training video directory"),
    ("input_video_pre", str, "scene", "prefix of the scene directories"),
    ("str_dir", int, 1000, "first scene index"),
    ("end_dir", int, 2000, "last training scene index"),
    ("end_dir_val", int, 2050, "last validation scene index"),
    ("max_frm", int, 119, "last frame index in a scene"),
    # losses
    ("vgg_scaling", float, -0.002, "VGG perceptual loss weight (<0 disables)"),
    ("warp_scaling", float, 1.0, "warp loss weight"),
    ("pingpang", bool, False, "bi-directional (ping-pong) recurrence"),
    ("pp_scaling", float, 1.0, "ping-pong loss weight"),
    # optimisation
    ("EPS", float, 1e-12, "log epsilon"),
    ("learning_rate", float, 0.0001, "learning rate"),
    ("decay_step", int, 500000, "lr decay steps"),
    ("decay_rate", float, 0.5, "lr decay rate"),
    ("stair", bool, False, "staircase decay"),
    ("beta", float, 0.9, "Adam beta1"),
    ("adameps", float, 1e-8, "Adam epsilon"),
    ("max_epoch", int, None, "max epochs"),
    ("max_iter", int, 1000000, "max iterations"),
    ("display_freq", int, 20, "display frequency"),
    ("summary_freq", int, 100, "summary frequency"),
    ("save_freq", int, 10000, "checkpoint frequency"),
    # Dst
    ("ratio", float, 0.01, "adversarial loss weight (<=0: FRVSR)"),
    ("Dt_mergeDs", bool, True, "spatio-temporal (merged) discriminator"),
    ("Dt_ratio_0", float, 1.0, "initial fade-in ratio"),
    ("Dt_ratio_add", float, 0.0, "fade-in increment per step"),
    ("Dt_ratio_max", float, 1.0, "max fade-in ratio"),
    ("Dbalance", float, 0.4, "adaptive D/G balance threshold"),
    ("crop_dt", float, 0.75, "temporal-discriminator crop factor"),
    ("D_LAYERLOSS", bool, True, "discriminator feature (layer) loss"),
]


EXTENSIONS = {"act_dtype": "bf16", "synthetic": False}       # flags of this implementation, not in the reference


def defaults(**kw):
    f = {name: default for name, _, default, _ in FLAG_TABLE}
    f.update(EXTENSIONS)
    unknown = set(kw) - set(f)
    if unknown:
        raise ValueError("unknown flag(s): %s" % sorted(unknown))
    f.update(kw)
    return SimpleNamespace(**f)


def _to_bool(s):
    if isinstance(s, bool):
        return s
    if s.lower() in ("1", "true", "t", "yes", "y"):
        return True
    if s.lower() in ("0", "false", "f", "no", "n"):
        return False
    raise argparse.ArgumentTypeError("boolean expected, got %r" % s)


def build_parser():
    ap = argparse.ArgumentParser(description="TecoGAN / FRVSR on MI355X (flags of the reference main.py)")
    for name, typ, default, helps in FLAG_TABLE:
        if typ is bool:
            ap.add_argument("--" + name, dest=name, nargs="?", const=True, default=default, type=_to_bool, help=helps)
            ap.add_argument("--no" + name, dest=name, action="store_false", help=argparse.SUPPRESS)
        else:
            ap.add_argument("--" + name, dest=name, type=typ, default=default, help=helps)
    # extensions of this implementation (not in the reference)
    ap.add_argument("--act_dtype", choices=["f32", "bf16"], default="bf16", help="activation dtype of the HIP path")
    ap.add_argument("--synthetic", action="store_true", help="train on synthetic sequences (no video directory)")
    return ap


def parse(argv=None):
    return build_parser().parse_args(argv)


def frvsr_flags(**kw):
    """reference runGan.py:247-286 (case 4): FRVSR training, no Dst, no ping-pong, 10 res blocks."""
    base = dict(mode="train", batch_size=4, RNN_N=10, crop_size=32, learning_rate=0.00005, decay_step=500000,
                decay_rate=1.0, stair=True, beta=0.9, max_iter=500000, save_freq=10000, num_resblock=10,
                ratio=-0.01, pingpang=False)
    base.update(kw)
    return defaults(**base)


def tecogan_flags(**kw):
    """reference runGan.py:107-234 (case 3): full TecoGAN (G + spatio-temporal D + VGG + ping-pong)."""
    base = dict(mode="train", batch_size=4, RNN_N=10, crop_size=32, learning_rate=0.00005, decay_step=500000,
                decay_rate=1.0, stair=True, beta=0.9, max_iter=500000, save_freq=10000, num_resblock=16,
                vgg_scaling=0.2, ratio=0.01, Dt_mergeDs=True, Dt_ratio_max=1.0, Dt_ratio_0=1.0, Dt_ratio_add=0.0,
                pingpang=True, pp_scaling=0.5, D_LAYERLOSS=True)
    base.update(kw)
    return defaults(**base)
