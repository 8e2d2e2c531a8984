#!/usr/bin/env python
"""Evaluation of result folders against target folders -- the command line of the reference's `metrics.py`
(`--output`, `--results`, `--targets`; runGan.py case 2), SURVEY.md section 8 row f4.

    python metrics.py --output <dir> --results r0,r1 --targets t0,t1 [--keys PSNR,SSIM,LPIPS,tOF,tLP100]
                      [--lpips_alexnet <torchvision AlexNet state_dict>] [--lpips_lin <LPIPSmodels/v0.1/alex.pth>]

Same protocol as the reference (metrics.py:120-239): per folder pair the PNGs are sorted by their digits, the first and
last `cutfr = 2` frames are skipped, results larger than the target are cropped, PSNR / SSIM are taken on the Y channel of
the centre crop `crop_8x8` leaves, per-folder / per-frame / per-folder-average tables go to `<output>/metrics.csv`, the log
to `<output>/metricsfile.txt`.

What this port computes itself, and what it needs from outside (nothing is silently substituted):
  * PSNR, SSIM: numpy / scipy restatements -- `skimage.measure.compare_ssim` with its defaults (7x7 uniform window, K1 0.01,
    K2 0.03, sample covariance) is restated in `ssim_y` (skimage is not installable offline).
  * LPIPS, tLP100: the AlexNet-linear LPIPS v0.1 network of the reference's LPIPSmodels/ restated in torch (`Lpips`).  Its
    linear heads ship with the reference (`LPIPSmodels/v0.1/alex.pth`, pass with --lpips_lin); the AlexNet backbone is
    torchvision's ImageNet checkpoint, which the reference downloads (pretrained_networks.py:60) and which does not exist
    offline: pass its state_dict with --lpips_alexnet, otherwise the two keys are SKIPPED with a message.
  * tOF: needs OpenCV's Farneback optical flow (metrics.py:159-160); computed only when `cv2` is importable, else skipped.
This is an offline CPU tool: it is not part of the hot path and uses no HIP kernel.
"""
import argparse
import os
import sys

import numpy as np

CUTFR = 2
ALL_KEYS = ["PSNR", "SSIM", "LPIPS", "tOF", "tLP100"]


def list_png_in_dir(dirpath):
    """reference metrics.py:29-36: *.png without the 'IB' prefix, ordered by the digits in the name."""
    names = [f for f in os.listdir(dirpath) if f.endswith(".png") and not f.startswith("IB")]
    names = sorted(names)
    names.sort(key=lambda f: int("".join(ch for ch in f if ch.isdigit()) or -1))
    return [os.path.join(dirpath, f) for f in names]


def rgb2y(img_u8):
    """Y of the BT.601 transform the reference uses (metrics.py:38-57, maxVal 255), input already rounded to [0,255]."""
    t = np.asarray(img_u8, dtype=np.float64)
    return t[..., 0] * 0.256788235294118 + t[..., 1] * 0.504129411764706 + t[..., 2] * 0.097905882352941 + 16.0


def to_uint8(x, vmin=0.0, vmax=255.0):
    x = np.asarray(x, dtype=np.float32)
    return np.clip(np.round((x - vmin) / (vmax - vmin) * 255.0), 0, 255)


def psnr_y(img_true, img_pred):
    """reference metrics.py:65-71."""
    d = rgb2y(to_uint8(img_true)) - rgb2y(to_uint8(img_pred))
    rmse = np.sqrt(np.mean(d * d))
    return 20.0 * np.log10(255.0 / rmse)


def ssim_plane(X, Y, data_range, win=7, K1=0.01, K2=0.03):
    """`skimage.measure.compare_ssim(X, Y, data_range=...)` with its defaults: uniform win x win window, sample covariance
    (normalised by NP/(NP-1)), mean over the region the window fully covers."""
    from scipy.ndimage import uniform_filter
    X, Y = np.asarray(X, np.float64), np.asarray(Y, np.float64)
    NP = win * win
    cov_norm = NP / (NP - 1.0)
    ux, uy = uniform_filter(X, size=win), uniform_filter(Y, size=win)
    uxx, uyy, uxy = uniform_filter(X * X, size=win), uniform_filter(Y * Y, size=win), uniform_filter(X * Y, size=win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
    pad = (win - 1) // 2
    return float(S[pad:S.shape[0] - pad, pad:S.shape[1] - pad].mean())


def ssim_y(img_true, img_pred):
    """reference metrics.py:73-76 (data_range = range of the PREDICTION's Y plane, as there)."""
    yt, yp = rgb2y(to_uint8(img_true)), rgb2y(to_uint8(img_pred))
    return ssim_plane(yt, yp, data_range=yp.max() - yp.min())


def crop_8x8(img):
    """reference metrics.py:78-93: the largest multiple-of-32 window that leaves at least 16 pixels in total, centred."""
    ori_h, ori_w = img.shape[0], img.shape[1]
    h, w = (ori_h // 32) * 32, (ori_w // 32) * 32
    while h > ori_h - 16:
        h -= 32
    while w > ori_w - 16:
        w -= 32
    y, x = (ori_h - h) // 2, (ori_w - w) // 2
    return img[y:y + h, x:x + w], y, x


class Lpips:
    """LPIPS v0.1, AlexNet + linear heads (reference LPIPSmodels/networks_basic.py:PNetLin, pretrained_networks.py:alexnet):
    inputs in [-1,1] NCHW; (x - shift) / scale; the five post-ReLU AlexNet feature maps, unit-normalised over channels,
    squared difference, 1x1 linear head, spatial mean, summed over the five taps."""
    SLICES = ((0,), (3,), (6,), (8,), (10,))                 # conv layers of torchvision's AlexNet.features ending each tap

    def __init__(self, alexnet_state, lin_state, device="cpu"):
        import torch
        self.t, self.dev = torch, device
        g = lambda k: alexnet_state[k].to(device).float()
        self.convs = [(g("features.%d.weight" % i), g("features.%d.bias" % i), s, p) for i, s, p in
                      ((0, 4, 2), (3, 1, 2), (6, 1, 1), (8, 1, 1), (10, 1, 1))]
        self.lins = [lin_state["lin%d.model.1.weight" % k].to(device).float() for k in range(5)]
        self.shift = torch.tensor([-.030, -.088, -.188], device=device).view(1, 3, 1, 1)
        self.scale = torch.tensor([.458, .448, .450], device=device).view(1, 3, 1, 1)

    def features(self, x):
        F = self.t.nn.functional
        outs, h = [], (x - self.shift) / self.scale
        for k, (w, b, s, p) in enumerate(self.convs):
            if k in (1, 2):                                   # max-pool 3x3 s2 precedes conv2 and conv3
                h = F.max_pool2d(h, 3, 2)
            h = F.relu(F.conv2d(h, w, b, stride=s, padding=p))
            outs.append(h)
        return outs

    def __call__(self, a, b):
        t = self.t
        with t.no_grad():
            val = 0.0
            for f0, f1, lin in zip(self.features(a.to(self.dev)), self.features(b.to(self.dev)), self.lins):
                n0 = f0 / (f0.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
                n1 = f1 / (f1.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
                val = val + t.nn.functional.conv2d((n0 - n1) ** 2, lin).mean(dim=(2, 3))
        return float(val.reshape(-1)[0])


def im2tensor(image):
    """reference LPIPSmodels/util.py:142-145: uint8 HWC RGB -> [-1,1] NCHW."""
    import torch
    return torch.from_numpy((np.asarray(image, np.float32) / (255.0 / 2.0) - 1.0).transpose(2, 0, 1)[None].copy())


def read_rgb(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def evaluate_pair(result_dir, target_dir, keys, lpips=None, flow=None, log=print):
    """One (result, target) folder pair -> dict key -> list of per-frame values (reference metrics.py:131-201)."""
    result, target = list_png_in_dir(result_dir), list_png_in_dir(target_dir)
    vals = {k: [] for k in keys}
    pre = {}
    for i in range(CUTFR, len(target) - CUTFR):
        out_img, tar_img = read_rgb(result[i]), read_rgb(target[i])
        msg = "frame %d, tar %s, out %s, " % (i, str(tar_img.shape), str(out_img.shape))
        if tar_img.shape[0] < out_img.shape[0] or tar_img.shape[1] < out_img.shape[1]:      # target not divisible by 4
            out_img = out_img[:tar_img.shape[0], :tar_img.shape[1]]
        log(result[i])
        if "tOF" in keys and flow is not None:
            og, tg = flow.grey(out_img), flow.grey(tar_img)
            if i > CUTFR:
                t_of, _, _ = crop_8x8(flow(pre["tg"], tg))
                o_of, _, _ = crop_8x8(flow(pre["og"], og))
                d = np.abs(t_of - o_of)
                vals["tOF"].append(float(np.sqrt((d * d).sum(-1)).mean()))
                msg += "tOF %02.2f, " % vals["tOF"][-1]
            pre["og"], pre["tg"] = og, tg
        tar_img, ofy, ofx = crop_8x8(tar_img)
        out_img, ofy, ofx = crop_8x8(out_img)
        if "PSNR" in keys:
            vals["PSNR"].append(psnr_y(tar_img, out_img))
            msg += "psnr %02.2f" % vals["PSNR"][-1]
        if "SSIM" in keys:
            vals["SSIM"].append(ssim_y(tar_img, out_img))
            msg += ", ssim %02.2f" % vals["SSIM"][-1]
        if lpips is not None and ("LPIPS" in keys or "tLP100" in keys):
            img0, img1 = im2tensor(tar_img), im2tensor(out_img)
            if "LPIPS" in keys:
                vals["LPIPS"].append(lpips(img0, img1))
                msg += ", lpips %02.2f" % vals["LPIPS"][-1]
            if "tLP100" in keys and i > CUTFR:
                vals["tLP100"].append(abs(lpips(pre["img0"], img0) - lpips(pre["img1"], img1)) * 100.0)
                msg += ", tLPx100 %02.2f" % vals["tLP100"][-1]
            pre["img0"], pre["img1"] = img0, img1
        log(msg + ", crop (%d, %d)" % (ofy, ofx))
    return vals


class FarnebackFlow:
    """OpenCV's Farneback flow with the reference's parameters (metrics.py:159-160); only constructed when cv2 imports."""

    def __init__(self):
        import cv2
        self.cv2 = cv2

    def grey(self, img):
        return self.cv2.cvtColor(img, self.cv2.COLOR_RGB2GRAY)

    def __call__(self, a, b):
        return self.cv2.calcOpticalFlowFarneback(a, b, None, 0.5, 3, 15, 3, 5, 1.2, 0)


class Tee(object):
    def __init__(self, path):
        self.terminal, self.log = sys.stdout, open(path, "a")

    def write(self, m):
        self.terminal.write(m)
        self.log.write(m)

    def flush(self):
        self.log.flush()


def main(argv=None, flow=None):
    """flow: optical-flow provider for tOF (an object with grey(img) and __call__(prev, next)); default: OpenCV's Farneback
    flow when cv2 imports -- tests pass a stand-in to exercise the tOF bookkeeping without OpenCV."""
    import pandas as pd
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--output", required=True, help="the path of output directory")
    ap.add_argument("--results", required=True, help="the list of paths of result directory")
    ap.add_argument("--targets", required=True, help="the list of paths of target directory")
    ap.add_argument("--keys", default=",".join(ALL_KEYS))
    ap.add_argument("--lpips_alexnet", default=None, help="torchvision AlexNet (ImageNet) state_dict file")
    ap.add_argument("--lpips_lin", default=None, help="LPIPS v0.1 linear heads (the reference's LPIPSmodels/v0.1/alex.pth)")
    a = ap.parse_args(argv)
    os.makedirs(a.output, exist_ok=True)
    sys.stdout = Tee(os.path.join(a.output, "metricsfile.txt"))
    keys = [k for k in a.keys.split(",") if k in ALL_KEYS]
    lpips = None
    if "LPIPS" in keys or "tLP100" in keys:
        if a.lpips_alexnet and a.lpips_lin and os.path.exists(a.lpips_alexnet) and os.path.exists(a.lpips_lin):
            import torch
            lpips = Lpips(torch.load(a.lpips_alexnet, map_location="cpu"), torch.load(a.lpips_lin, map_location="cpu"))
        else:
            print("[metrics] LPIPS / tLP100 skipped: pass --lpips_alexnet (torchvision's AlexNet ImageNet state_dict, which the "
                  "reference downloads) and --lpips_lin (LPIPSmodels/v0.1/alex.pth of the reference)")
            keys = [k for k in keys if k not in ("LPIPS", "tLP100")]
    if "tOF" in keys and flow is None:
        try:
            flow = FarnebackFlow()
        except ImportError:
            print("[metrics] tOF skipped: it is defined through OpenCV's Farneback optical flow and cv2 is not installed")
            keys = [k for k in keys if k != "tOF"]
    result_list, target_list = a.results.split(","), a.targets.split(",")
    sums, lens, avgs, folders = {k: 0.0 for k in keys}, {k: 0 for k in keys}, {k: [] for k in keys}, {k: 0.0 for k in keys}
    for fi, (rd, td) in enumerate(zip(result_list, target_list)):
        vals = evaluate_pair(rd, td, keys, lpips, flow)
        table = {}
        for k in keys:
            cur = np.float32(vals[k])
            table["%s_%02d" % (k, fi)] = pd.Series(cur)
            mean = cur.sum() / max(cur.shape[0], 1)
            print("%s_%02d, max %02.4f, min %02.4f, avg %02.4f" % (k, fi, cur.max(), cur.min(), mean))
            avgs[k].append(mean)
            sums[k] += float(cur.sum())
            lens[k] += int(cur.shape[0])
            folders[k] += mean
        pd.DataFrame(table).to_csv(os.path.join(a.output, "metrics.csv"), mode="w" if fi == 0 else "a")
    n = len(result_list)
    for k in keys:
        print("%s, total frame %d, total avg %02.4f, folder avg %02.4f" % (k, lens[k], sums[k] / max(lens[k], 1), folders[k] / n))
    csv = os.path.join(a.output, "metrics.csv")
    pd.DataFrame({"Avg_" + k: pd.Series(np.float32(avgs[k])) for k in keys}).to_csv(csv, mode="a")
    pd.DataFrame({"FolderAvg_" + k: pd.Series([folders[k] / n]) for k in keys}).to_csv(csv, mode="a")
    pd.DataFrame({"FrameAvg_" + k: pd.Series([sums[k] / max(lens[k], 1)]) for k in keys}).to_csv(csv, mode="a")


if __name__ == "__main__":
    main()
