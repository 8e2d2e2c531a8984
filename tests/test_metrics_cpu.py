"""the evaluation port (metrics.py) -- PSNR / SSIM restatements against independent brute-force forms, the
folder protocol of the reference's metrics.py:120-239, and the torch restatement of the LPIPS v0.1 AlexNet-linear network."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metrics as M  # noqa: E402


def _img(h, w, seed):
    r = np.random.RandomState(seed)
    base = r.rand(h // 8 + 2, w // 8 + 2, 3)
    big = np.kron(base, np.ones((8, 8, 1)))[:h, :w]
    return np.clip((big + 0.05 * r.randn(h, w, 3)) * 255, 0, 255).astype(np.uint8)


def test_crop_8x8_matches_the_reference_rule():
    for h, w in ((576, 720), (1080, 1920), (100, 130), (64, 64)):
        c, y, x = M.crop_8x8(np.zeros((h, w, 3)))
        assert c.shape[0] % 32 == 0 and c.shape[1] % 32 == 0
        assert c.shape[0] <= h - 16 and c.shape[1] <= w - 16 and c.shape[0] + 32 > h - 16 and c.shape[1] + 32 > w - 16
        assert y == (h - c.shape[0]) // 2 and x == (w - c.shape[1]) // 2


def test_psnr_on_the_y_channel():
    a, b = _img(64, 96, 1), _img(64, 96, 2)
    T = np.array([0.256788235294118, 0.504129411764706, 0.097905882352941])
    ya, yb = a.astype(np.float64) @ T + 16, b.astype(np.float64) @ T + 16
    want = 20 * np.log10(255.0 / np.sqrt(np.mean((ya - yb) ** 2)))
    assert abs(M.psnr_y(a, b) - want) < 1e-9
    assert M.psnr_y(a, a.astype(np.float32) + 0.4) == float("inf") or M.psnr_y(a, a.astype(np.float32) + 0.4) > 100   # rounds back


def test_ssim_equals_a_brute_force_window_loop():
    """skimage's compare_ssim defaults restated with uniform_filter == an explicit loop over every fully covered 7x7 window
    with the sample (N-1) covariance."""
    r = np.random.RandomState(0)
    X, Y = r.rand(19, 23) * 200 + 20, r.rand(19, 23) * 200 + 20
    dr = Y.max() - Y.min()
    C1, C2 = (0.01 * dr) ** 2, (0.03 * dr) ** 2
    acc = []
    for i in range(3, 19 - 3):
        for j in range(3, 23 - 3):
            wx, wy = X[i - 3:i + 4, j - 3:j + 4].ravel(), Y[i - 3:i + 4, j - 3:j + 4].ravel()
            ux, uy = wx.mean(), wy.mean()
            vx, vy = wx.var(ddof=1), wy.var(ddof=1)
            vxy = ((wx - ux) * (wy - uy)).sum() / 48.0
            acc.append(((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2)))
    assert abs(M.ssim_plane(X, Y, dr) - np.mean(acc)) < 1e-10
    a = _img(64, 64, 3)
    assert abs(M.ssim_y(a, a) - 1.0) < 1e-12
    assert M.ssim_y(a, _img(64, 64, 4)) < 0.9


def _fake_lpips():
    g = torch.Generator().manual_seed(0)
    shapes = {0: (64, 3, 11, 11), 3: (192, 64, 5, 5), 6: (384, 192, 3, 3), 8: (256, 384, 3, 3), 10: (256, 256, 3, 3)}
    alex = {}
    for i, sh in shapes.items():
        alex["features.%d.weight" % i] = torch.randn(sh, generator=g) * (2.0 / (sh[1] * sh[2] * sh[3])) ** 0.5
        alex["features.%d.bias" % i] = torch.zeros(sh[0])
    lin = {"lin%d.model.1.weight" % k: torch.rand(1, c, 1, 1, generator=g) for k, c in enumerate((64, 192, 384, 256, 256))}
    return M.Lpips(alex, lin)


def test_lpips_network_structure_and_properties():
    net = _fake_lpips()
    a, b = M.im2tensor(_img(96, 128, 5)), M.im2tensor(_img(96, 128, 6))
    feats = net.features(a)
    assert [f.shape[1] for f in feats] == [64, 192, 384, 256, 256]
    assert [tuple(f.shape[2:]) for f in feats] == [(23, 31), (11, 15), (5, 7), (5, 7), (5, 7)]     # AlexNet strides / pools
    assert net(a, a) == 0.0 and net(a, b) > 0.0
    assert abs(net(a, b) - net(b, a)) < 1e-6
    lin_path = "/root/reference/LPIPSmodels/v0.1/alex.pth"
    if os.path.exists(lin_path):                               # the reference's own linear heads have the shapes the port expects
        sd = torch.load(lin_path, map_location="cpu")
        assert [tuple(sd["lin%d.model.1.weight" % k].shape) for k in range(5)] == [(1, c, 1, 1) for c in (64, 192, 384, 256, 256)]


def test_folder_protocol_and_csv(tmp_path):
    from PIL import Image
    res, tar, out = tmp_path / "res", tmp_path / "tar", tmp_path / "out"
    res.mkdir()
    tar.mkdir()
    for i in range(7):
        t = _img(70, 90, 10 + i)
        Image.fromarray(t).save(tar / ("target_%04d.png" % i))
        r = np.clip(t.astype(np.int32) + (3 if i != 3 else 0), 0, 255).astype(np.uint8)
        Image.fromarray(np.pad(r, ((0, 2), (0, 2), (0, 0)), mode="edge")).save(res / ("output_%04d.png" % i))   # larger: cropped
    Image.fromarray(_img(70, 90, 99)).save(res / "IB_ignored_0001.png")
    old = sys.stdout
    try:
        M.main(["--output", str(out), "--results", str(res), "--targets", str(tar), "--keys", "PSNR,SSIM,LPIPS,tOF"])
    finally:
        sys.stdout = old
    log = (out / "metricsfile.txt").read_text()
    assert "LPIPS / tLP100 skipped" in log
    assert "PSNR, total frame 3, total avg" in log                      # 7 frames minus 2 x cutfr
    vals = M.evaluate_pair(str(res), str(tar), ["PSNR", "SSIM"], log=lambda *_: None)
    assert len(vals["PSNR"]) == 3 and vals["PSNR"][1] > 100 and vals["SSIM"][1] == pytest.approx(1.0)     # frame 3 is identical
    assert 30 < vals["PSNR"][0] < 45
    csv = (out / "metrics.csv").read_text()
    assert "PSNR_00" in csv and "Avg_PSNR" in csv and "FolderAvg_SSIM" in csv and "FrameAvg_PSNR" in csv


def test_lpips_restatement_matches_the_reference_module_on_seeded_backbone():
    """metrics.Lpips against the reference's own LPIPSmodels.networks_basic.PNetLin('alex', v0.1): the distances in
    tests/golden/lpips_reference.npz were produced by that module (oracle/make_golden_lpips.py, imported unmodified from the
    reference) with the reference's trained linear heads and a SEEDED random AlexNet backbone (the ImageNet weights are a
    torchvision download); the same backbone state and the stored heads must give the same numbers here."""
    import torch
    from oracle.make_golden_lpips import seeded_alexnet_state, seeded_images
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "lpips_reference.npz"))
    lin = {k: torch.from_numpy(gold["lin/" + k]) for k in gold["lin_keys"]}
    net = M.Lpips(seeded_alexnet_state(), lin)
    for case, (h, w, d_ref, d_same) in enumerate(gold["cases"]):
        a, b = seeded_images(100 + case, int(h), int(w))
        d = net(M.im2tensor(a), M.im2tensor(b))
        assert abs(d - d_ref) <= 1e-5 * max(abs(d_ref), 1e-3), (case, d, d_ref)
        assert net(M.im2tensor(a), M.im2tensor(a)) == d_same == 0.0


def test_metrics_cli_reproduces_the_reference_script_csv(tmp_path):
    """tests/golden/metrics_reference.csv is the metrics.csv the REFERENCE's own metrics.py wrote for the seeded folders of
    oracle/make_golden_metrics.py (run unmodified, on stand-ins for absl / cv2 / skimage / torchvision -- see that script).
    This repository's metrics.py, given the same folders, the same seeded LPIPS backbone, the reference's linear heads and
    the same stand-in flow, must write the same tables: listing and ordering, frame cut, crop of oversized results,
    crop_8x8, PSNR, SSIM call convention, LPIPS, tLP100, tOF bookkeeping, per-folder / Avg / FolderAvg / FrameAvg rows."""
    import torch
    from oracle import make_golden_metrics as G
    from oracle.make_golden_lpips import seeded_alexnet_state
    res, tar = G.write_folders(str(tmp_path))
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "lpips_reference.npz"))
    torch.save(seeded_alexnet_state(), str(tmp_path / "alexnet.pth"))
    torch.save({str(k): torch.from_numpy(gold["lin/" + str(k)]) for k in gold["lin_keys"]}, str(tmp_path / "lin.pth"))

    class Flow:
        grey = staticmethod(G.grey)

        def __call__(self, a, b):
            return G.standin_flow(a, b)

    stdout = sys.stdout
    try:
        M.main(["--output", str(tmp_path / "out"), "--results", ",".join(res), "--targets", ",".join(tar),
                "--lpips_alexnet", str(tmp_path / "alexnet.pth"), "--lpips_lin", str(tmp_path / "lin.pth")], flow=Flow())
    finally:
        sys.stdout = stdout
    mine = open(str(tmp_path / "out" / "metrics.csv")).read().strip().splitlines()
    ref = open(os.path.join(os.path.dirname(__file__), "golden", "metrics_reference.csv")).read().strip().splitlines()
    assert len(mine) == len(ref), (len(mine), len(ref))
    for lm, lr in zip(mine, ref):
        cm, cr = lm.split(","), lr.split(",")
        assert len(cm) == len(cr), (lm, lr)
        for a, b in zip(cm, cr):
            try:
                fb = float(b)
            except ValueError:
                assert a == b, (lm, lr)                       # header cells and empty cells
                continue
            assert abs(float(a) - fb) <= 2e-5 * max(abs(fb), 1e-3), (lm, lr)
