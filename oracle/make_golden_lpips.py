#!/usr/bin/env python
"""Generate tests/golden/lpips_reference.npz by running the REFERENCE's own LPIPS network
(/root/reference/LPIPSmodels/networks_basic.py:PNetLin 'alex' v0.1, imported unmodified) on seeded images.

    python -m oracle.make_golden_lpips

Test infrastructure (only tests/ use the vectors).  What this pins: the wiring of metrics.Lpips -- input scaling, the five
AlexNet taps and where the max-pools sit, channel normalisation, squared difference, 1x1 heads, spatial mean, sum -- and the
reference's TRAINED linear heads (LPIPSmodels/v0.1/alex.pth, 1152 floats, stored in the fixture).  What it does not: the
ImageNet-trained AlexNet backbone, which the reference downloads through torchvision (absent here, no network): the backbone
runs with the seeded random weights of `seeded_alexnet_state`, the same in this script and in the test.
The reference's module imports torchvision, IPython, skimage, cv2 and matplotlib at import time; none is used by the 'alex'
path beyond `torchvision.models.alexnet(...).features`, so they are provided as empty stand-ins."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden", "lpips_reference.npz")

ALEX_CONVS = ((0, 3, 64, 11), (3, 64, 192, 5), (6, 192, 384, 3), (8, 384, 256, 3), (10, 256, 256, 3))   # features.<i>: cin, cout, k


def seeded_alexnet_state(seed=7):
    """State dict of torchvision's AlexNet.features with seeded He-normal weights (a stand-in for the ImageNet weights)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for i, cin, cout, k in ALEX_CONVS:
        sd["features.%d.weight" % i] = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        sd["features.%d.bias" % i] = torch.randn(cout, generator=g) * 0.05
    return sd


def seeded_images(seed, h, w):
    """A uint8 RGB image and a perturbed copy (smooth content + noise), as metrics.py reads them from PNG files."""
    r = np.random.RandomState(seed)
    base = r.rand(h // 4 + 2, w // 4 + 2, 3)
    img = np.kron(base, np.ones((4, 4, 1)))[:h, :w]
    a = np.clip(img * 255 + r.randn(h, w, 3) * 4, 0, 255).astype(np.uint8)
    other = np.kron(r.rand(h // 8 + 2, w // 8 + 2, 3), np.ones((8, 8, 1)))[:h, :w]
    b = np.clip((0.6 * img + 0.4 * other) * 255 * 0.9 + 12 + r.randn(h, w, 3) * 10, 0, 255).astype(np.uint8)
    return a, b


def _reference_net():
    from torch import nn
    feats = nn.Sequential(
        nn.Conv2d(3, 64, 11, 4, 2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
        nn.Conv2d(64, 192, 5, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
        nn.Conv2d(192, 384, 3, padding=1), nn.ReLU(inplace=True),
        nn.Conv2d(384, 256, 3, padding=1), nn.ReLU(inplace=True),
        nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2))      # torchvision AlexNet.features
    feats.load_state_dict({k[len("features."):]: v for k, v in seeded_alexnet_state().items()})

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tv = stub("torchvision")
    tv.models = stub("torchvision.models", alexnet=lambda pretrained=False: types.SimpleNamespace(features=feats))
    stub("IPython", embed=lambda *a, **k: None)
    sk = stub("skimage")
    sk.color = stub("skimage.color")
    sk.measure = stub("skimage.measure", compare_ssim=None)
    sk.transform = stub("skimage.transform")
    stub("cv2")
    mp = stub("matplotlib")
    mp.pyplot = stub("matplotlib.pyplot")
    import scipy.ndimage
    stub("scipy.ndimage.interpolation", zoom=scipy.ndimage.zoom)
    sys.path.insert(0, REF)
    from LPIPSmodels import networks_basic as NB
    net = NB.PNetLin(pnet_type="alex", pnet_rand=True, use_dropout=True, use_gpu=False, version="0.1")
    lin = torch.load(os.path.join(REF, "LPIPSmodels", "v0.1", "alex.pth"), map_location="cpu")
    net.load_state_dict(lin)
    net.eval()
    return net, lin


def main():
    sys.path.insert(0, ROOT)
    from metrics import im2tensor
    net, lin = _reference_net()
    out = {"lin_keys": np.array(sorted(lin.keys()))}
    for k, v in lin.items():
        out["lin/" + k] = v.numpy()
    dists = []
    for case, (h, w) in enumerate(((64, 64), (52, 76), (120, 88))):
        a, b = seeded_images(100 + case, h, w)
        with torch.no_grad():
            d = net.forward(im2tensor(a), im2tensor(b))
            d_same = net.forward(im2tensor(a), im2tensor(a))
        dists.append([h, w, float(d.reshape(-1)[0]), float(d_same.reshape(-1)[0])])
        print("case %d  %dx%d  LPIPS(ref module) = %.8f   (identical images: %.3g)" % (case, h, w, dists[-1][2], dists[-1][3]))
    out["cases"] = np.array(dists, np.float64)
    np.savez_compressed(GOLD, **out)
    print("wrote", GOLD, os.path.getsize(GOLD), "bytes")


if __name__ == "__main__":
    main()
