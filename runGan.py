#!/usr/bin/env python
"""Recipe launcher with the reference's case numbers (reference runGan.py): `python runGan.py <case>`.

  1  inference of ./LR/calendar (or --lr-dir) with a trained model        (reference runGan.py:67-90)
  3  TecoGAN training: G + spatio-temporal D + VGG + ping-pong            (reference runGan.py:107-244)
  4  FRVSR training: l2 content + l2 warp                                 (reference runGan.py:247-296)
  2  metrics of ./results/<scene> against ./HR/<scene> -> ./results/metric_log/   (reference runGan.py:91-105; metrics.py)
  0  (dataset / model download) is outside the MI355X hot path and only prints a notice.

Each case spawns `main.py` as a child process with the reference's flag list (same spellings and values).
Extras of this implementation go after `--`:   python runGan.py 4 -- --synthetic --max_iter 200
`--gpus N` launches training with one process per GPU (torch.distributed over RCCL).
"""
import datetime
import os
import signal
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
COMMON_TRAIN = [("batch_size", 4), ("RNN_N", 10), ("movingFirstFrame", None), ("random_crop", None),
                ("crop_size", 32), ("learning_rate", 0.00005), ("decay_step", 500000), ("decay_rate", 1.0),
                ("stair", None), ("beta", 0.9), ("max_iter", 500000), ("save_freq", 10000)]
DATA = [("input_video_dir", "/mnt/netdisk/video_data/"), ("input_video_pre", "scene"), ("str_dir", 2000),
        ("end_dir", 2250), ("end_dir_val", 2290), ("max_frm", 119), ("queue_thread", 12),
        ("name_video_queue_capacity", 1024), ("video_queue_capacity", 1024)]
RECIPES = {
    3: [("num_resblock", 16), ("vgg_scaling", 0.2), ("vgg_ckpt", "model/vgg_19.ckpt"), ("pre_trained_model", None),
        ("checkpoint", "model/ourFRVSR"), ("ratio", 0.01), ("Dt_mergeDs", None), ("Dt_ratio_max", 1.0),
        ("Dt_ratio_0", 1.0), ("Dt_ratio_add", 0.0), ("pingpang", None), ("pp_scaling", 0.5), ("D_LAYERLOSS", None)],
    4: [("num_resblock", 10), ("ratio", -0.01), ("nopingpang", None)],
}


def to_argv(pairs):
    out = []
    for k, v in pairs:
        out.append("--" + k)
        if v is not None:
            out.append(str(v))
    return out


def launch(cmd, wait_sigint=True):
    print(" ".join(cmd))
    child = subprocess.Popen(cmd, preexec_fn=os.setpgrp)
    try:
        child.communicate()
    except KeyboardInterrupt:                      # forward Ctrl+C so the child saves a last checkpoint
        print("runGAN.py: sending SIGINT signal to the sub process...")
        child.send_signal(signal.SIGINT)
        child.communicate()
    return child.returncode


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        extra = argv[argv.index("--") + 1:]
        argv = argv[:argv.index("--")]
    gpus = 1
    if "--gpus" in argv:
        gpus = int(argv[argv.index("--gpus") + 1])
    if not argv or not argv[0].isdigit():
        raise SystemExit(__doc__)
    case = int(argv[0])
    py = [sys.executable]
    if case == 0:
        print("runGan.py 0 (dataset / model download) is outside the MI355X hot path of this repository.")
        return 0
    if case == 2:           # reference runGan.py:91-105
        testpre, dirstr, tarstr = ["calendar"], "./results/", "./HR/"
        cmd = py + [os.path.join(HERE, "metrics.py"), "--output", dirstr + "metric_log/",
                    "--results", ",".join(dirstr + s_ for s_ in testpre), "--targets", ",".join(tarstr + s_ for s_ in testpre)] + extra
        return subprocess.call(cmd)
    if case == 1:
        out = "./results/"
        os.makedirs(out, exist_ok=True)
        rc = 0
        for scene in ["calendar"]:
            cmd = py + [os.path.join(HERE, "main.py")] + to_argv([
                ("cudaID", "0"), ("output_dir", out), ("summary_dir", os.path.join(out, "log/")),
                ("mode", "inference"), ("input_dir_LR", os.path.join("./LR/", scene)), ("output_pre", scene),
                ("num_resblock", 16), ("checkpoint", "./model/TecoGAN"), ("output_ext", "png")]) + extra
            rc |= launch(cmd)
        return rc
    if case in RECIPES:
        stamp = datetime.datetime.now().strftime("%m-%d-%H")
        train_dir = "ex_%s%s/" % ("TecoGAN" if case == 3 else "FRVSR", stamp)
        os.makedirs(train_dir, exist_ok=True)
        head = [("cudaID", "0"), ("output_dir", train_dir), ("summary_dir", os.path.join(train_dir, "log/")),
                ("mode", "train")]
        pairs = head + COMMON_TRAIN + RECIPES[case] + DATA
        if "--synthetic" in extra:                 # no dataset / pre-trained models needed
            pairs = [(k, v) for k, v in pairs if k not in ("checkpoint", "pre_trained_model", "vgg_ckpt")]
        if gpus > 1:
            py = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
                  "--master-addr", "127.0.0.1", "--master-port", "29533"]
        return launch(py + [os.path.join(HERE, "main.py")] + to_argv(pairs) + extra)
    raise SystemExit("unknown case %d\n%s" % (case, __doc__))


if __name__ == "__main__":
    sys.exit(main())
