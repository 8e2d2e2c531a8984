#!/usr/bin/env python
"""Generate tests/golden/metrics_reference.csv by running the REFERENCE's own evaluation script
(/root/reference/metrics.py, executed unmodified with runpy) on seeded PNG folders.

    python -m oracle.make_golden_metrics

Test infrastructure.  The script is run end to end -- folder listing and ordering, the 2-frame cut at both ends, the crop of
oversized results, crop_8x8, Y-channel PSNR, LPIPS / tLP100 through the reference's LPIPSmodels.dist_model.DistModel, the
per-folder / Avg_ / FolderAvg_ / FrameAvg_ tables of metrics.csv -- on top of stand-ins for what does not exist offline:
  * absl.flags (three string flags), cv2.imread / cvtColor (PIL + the BT.601 grey formula);
  * cv2.calcOpticalFlowFarneback -> `standin_flow` below (NOT Farneback: it pins the tOF bookkeeping, not the flow);
  * skimage.measure.compare_ssim -> this repository's metrics.ssim_plane (so the SSIM column pins how the reference CALLS
    it -- Y planes, data_range of the prediction -- not skimage's numerics, which tests/test_metrics_cpu.py holds by a
    brute-force window loop);
  * torchvision's ImageNet AlexNet -> the seeded backbone of oracle/make_golden_lpips.py; `.cuda()` -> identity.
tests/test_metrics_cpu.py regenerates the same folders and requires this repository's metrics.py to reproduce the CSV."""
import os
import runpy
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden", "metrics_reference.csv")
sys.path.insert(0, ROOT)

from oracle.make_golden_lpips import seeded_alexnet_state  # noqa: E402

FOLDERS = ((7, 96, 128, 0), (6, 70, 90, 2))       # frames, target height, width, extra rows/cols of the result frames


def write_folders(root):
    """Two (result, target) folder pairs of seeded PNG frames; the second target is not a multiple of 4 and its results are
    two pixels larger (the reference crops them, metrics.py:143-144).  Returns ([result dirs], [target dirs])."""
    from PIL import Image
    res, tar = [], []
    for fi, (n, h, w, extra) in enumerate(FOLDERS):
        r = np.random.RandomState(40 + fi)
        rd, td = os.path.join(root, "res_%d" % fi), os.path.join(root, "tar_%d" % fi)
        os.makedirs(rd)
        os.makedirs(td)
        base = np.kron(r.rand(h // 8 + 4, w // 8 + 4, 3), np.ones((8, 8, 1)))
        for f in range(n):
            big = base[f:f + h + extra, 2 * f:2 * f + w + extra] * 255
            t = np.clip(big[:h, :w] + r.randn(h, w, 3) * 3, 0, 255).astype(np.uint8)
            o = np.clip(big * 0.8 + 25 + r.randn(h + extra, w + extra, 3) * 14, 0, 255).astype(np.uint8)
            Image.fromarray(t).save(os.path.join(td, "frame_%04d.png" % f))
            Image.fromarray(o).save(os.path.join(rd, "output_%04d.png" % f))
        Image.fromarray(t).save(os.path.join(td, "IB_%04d.png" % 0))           # 'IB*' files are ignored by the listing
        res.append(rd)
        tar.append(td)
    return res, tar


def grey(img_rgb):
    """cv2.cvtColor(img, COLOR_RGB2GRAY) for uint8: BT.601 weights, rounded."""
    f = np.asarray(img_rgb, np.float64)
    return np.clip(np.round(f[..., 0] * 0.299 + f[..., 1] * 0.587 + f[..., 2] * 0.114), 0, 255).astype(np.uint8)


def standin_flow(a, b, *unused):
    """Stand-in for cv2.calcOpticalFlowFarneback(prev, next, ...): a deterministic [H,W,2] float32 field of the two frames."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.stack(((b - a) / 32.0, (np.roll(b, 1, 1) - a) / 64.0), -1).astype(np.float32)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_standins(argv):
    from PIL import Image
    from torch import nn
    import metrics as M
    flags = types.SimpleNamespace()
    values = {}

    class _Flags:
        def __call__(self, av):
            for k, v in zip(av[1::2], av[2::2]):
                values[k.lstrip("-")] = v
            return av

        def __getattr__(self, k):
            return values.get(k)

        def flag_values_dict(self):
            return dict(values)

    flags.FLAGS = _Flags()
    flags.DEFINE_string = lambda name, default, doc: values.setdefault(name, default)
    absl = _stub("absl")
    absl.flags = _stub("absl.flags", **flags.__dict__)
    _stub("cv2", imread=lambda p: np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1], COLOR_RGB2GRAY=7,
          cvtColor=lambda img, code: grey(img), calcOpticalFlowFarneback=standin_flow)
    sk = _stub("skimage")
    sk.color = _stub("skimage.color")
    sk.transform = _stub("skimage.transform")
    sk.measure = _stub("skimage.measure", compare_ssim=lambda X, Y, data_range=None: M.ssim_plane(X, Y, data_range))
    _stub("IPython", embed=lambda *a, **k: None)
    mp = _stub("matplotlib")
    mp.pyplot = _stub("matplotlib.pyplot")
    import scipy.ndimage
    _stub("scipy.ndimage.interpolation", zoom=scipy.ndimage.zoom)
    feats = nn.Sequential(
        nn.Conv2d(3, 64, 11, 4, 2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
        nn.Conv2d(64, 192, 5, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
        nn.Conv2d(192, 384, 3, padding=1), nn.ReLU(inplace=True),
        nn.Conv2d(384, 256, 3, padding=1), nn.ReLU(inplace=True),
        nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2))      # torchvision AlexNet.features
    feats.load_state_dict({k[len("features."):]: v for k, v in seeded_alexnet_state().items()})
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models", alexnet=lambda pretrained=False: types.SimpleNamespace(features=feats))
    torch.nn.Module.cuda = lambda self, *a, **k: self        # the reference asks for use_gpu=True; there is no GPU here
    torch.Tensor.cuda = lambda self, *a, **k: self
    orig_load = torch.load                                   # the heads were saved from CUDA tensors
    torch.load = lambda f, *a, **k: orig_load(f, *a, **dict(k, map_location="cpu"))
    sys.argv = argv


def main():
    tmp = tempfile.mkdtemp()
    try:
        res, tar = write_folders(tmp)
        out = os.path.join(tmp, "out")
        install_standins(["metrics.py", "--output", out, "--results", ",".join(res), "--targets", ",".join(tar)])
        sys.path.insert(0, REF)
        stdout = sys.stdout
        try:
            runpy.run_path(os.path.join(REF, "metrics.py"), run_name="__main__")
        finally:
            sys.stdout = stdout
        shutil.copy(os.path.join(out, "metrics.csv"), GOLD)
        print(open(GOLD).read())
        print("wrote", GOLD)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
