"""The reference's Python operator / construction API (lib/ops.py, lib/frvsr.py, lib/Teco.py, main.py) on the
HIP backend, checked against the CPU oracle with the same variables."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle.nets as ON
import oracle.ops as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rnd(*shape, seed=0):
    return torch.rand(*shape, generator=torch.Generator().manual_seed(seed)) * 2 - 1


def close(a, b, tol=1e-4):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item()), (a - b).abs().max().item()


@pytest.fixture(autouse=True)
def fresh_graph():
    import lib.frvsr
    import lib.ops
    lib.ops.reset_default_graph(seed=99)
    lib.frvsr._NETS.clear()
    yield


def cpu_vars(prefix=""):
    import lib.ops
    return {k: v.detach().cpu().clone() for k, v in lib.ops.global_variables().items() if k.startswith(prefix)}


def test_ops_conv_family_and_variable_names():
    import lib.ops as L
    x = rnd(2, 9, 11, 5, seed=1)
    with L.variable_scope("net"):
        y = L.conv2(x.cuda(), 3, 16, 1, scope="conv_1")
        y2 = L.conv2(x.cuda(), 4, 8, 2, use_bias=False, scope="conv_s2")
        y3 = L.conv2_tran(x.cuda(), 3, 12, 2, scope="conv_tran1")
        d = L.denselayer(x.cuda(), 1)
    V = cpu_vars()
    assert set(V) == {"net/conv_1/Conv/weights", "net/conv_1/Conv/biases", "net/conv_s2/Conv/weights",
                      "net/conv_tran1/Conv2d_transpose/weights", "net/conv_tran1/Conv2d_transpose/biases",
                      "net/dense/kernel", "net/dense/bias"}
    close(y, O.conv2(x, V["net/conv_1/Conv/weights"], V["net/conv_1/Conv/biases"], 1))
    close(y2, O.conv2(x, V["net/conv_s2/Conv/weights"], None, 2))
    close(y3, O.conv2_tran(x, V["net/conv_tran1/Conv2d_transpose/weights"], V["net/conv_tran1/Conv2d_transpose/biases"], 2))
    close(d, O.denselayer(x, V["net/dense/kernel"], V["net/dense/bias"]))
    with pytest.raises(ValueError):                      # TF semantics: re-creating without reuse is an error
        with L.variable_scope("net"):
            L.conv2(x.cuda(), 3, 16, 1, scope="conv_1")
    with L.variable_scope("net", reuse=True):
        close(L.conv2(x.cuda(), 3, 16, 1, scope="conv_1"), y, 0)


def test_ops_resize_pool_warp_gauss():
    import lib.ops as L
    x = rnd(2, 6, 7, 3, seed=2)
    close(L.upscale_four(x.cuda()), O.upscale_four(x), 1e-6)
    close(L.bicubic_four(x.cuda()), O.bicubic_four(x), 1e-5)
    close(L.maxpool(x.cuda()), O.maxpool(x), 0)
    close(L.lrelu(x.cuda(), 0.2), O.lrelu(x, 0.2), 0)
    fl = rnd(2, 6, 7, 2, seed=3) * 2
    close(L.dense_image_warp(x.cuda(), fl.cuda()), O.dense_image_warp(x, fl), 1e-5)
    assert torch.equal(L.space_to_depth(rnd(1, 8, 8, 3, seed=4).cuda(), 4).cpu(), O.space_to_depth4(rnd(1, 8, 8, 3, seed=4)))
    hr = torch.rand(2, 41, 45, 3, generator=torch.Generator().manual_seed(5))
    gk = torch.tensor(L.gaussian_2dkernel(9, 1.5), dtype=torch.float32)
    w = torch.zeros(9, 9, 3, 3)
    for c in range(3):
        w[:, :, c, c] = gk
    ref = torch.nn.functional.conv2d(hr.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), stride=4).permute(0, 2, 3, 1)
    close(L.tf_data_gaussDownby4(hr.cuda(), 1.5), ref, 1e-5)
    assert L.preprocess(0.25) == -0.5 and L.deprocess(-0.5) == 0.25


def test_frvsr_generator_and_fnet_drop_in():
    import lib.frvsr as Fr
    import lib.ops as L
    from tecogan_amd.flags import defaults
    FL = defaults(num_resblock=2)
    gi = torch.cat((torch.rand(2, 12, 10, 3), rnd(2, 12, 10, 48, seed=6)), -1)
    with pytest.raises(ValueError):
        Fr.generator_F(gi.cuda(), 3, reuse=False, FLAGS=None)
    with L.variable_scope("generator"):
        out = Fr.generator_F(gi.cuda(), 3, reuse=False, FLAGS=FL)
    close(out, ON.generator_F(cpu_vars("generator/"), gi, 2), 1e-4)
    with L.variable_scope("generator"):
        close(Fr.generator_F(gi.cuda(), 3, reuse=True, FLAGS=FL), out, 0)
    fi = torch.rand(3, 16, 24, 6, generator=torch.Generator().manual_seed(7))
    with L.variable_scope("fnet"):
        flow = Fr.fnet(fi.cuda(), reuse=False)
    close(flow, ON.fnet(cpu_vars("fnet/"), fi), 1e-4)
    # checkpoint-style assignment into the named variables + sync
    V = L.global_variables()
    V["generator/generator_unit/output_stage/conv/Conv/biases"].fill_(0.25)
    Fr.sync_variables()
    with L.variable_scope("generator"):
        out2 = Fr.generator_F(gi.cuda(), 3, reuse=True, FLAGS=FL)
    close(out2, ON.generator_F(cpu_vars("generator/"), gi, 2), 1e-4)


def test_generator_single_frame_raw_xavier_weights_at_baseline_size():
    """every BASELINE-size parity test and the bench run DAMPED weights (params.damp_values: the 19-frame
    recurrence is expansive with the raw init).  Without a recurrence nothing needs damping: ONE frame of generator_F
    (lib/frvsr.py:44-88) at the training shape [4,32,32,51], num_resblock = 16, seeded raw xavier weights exactly as
    lib/ops.py:40,52 draws them, fp32 mode, against the oracle in float64 with north_star's per-pixel criterion."""
    import lib.frvsr as Fr
    import lib.ops as L
    from tecogan_amd.flags import defaults
    FL = defaults(num_resblock=16)
    g = torch.Generator().manual_seed(11)
    gi = torch.cat((torch.rand(4, 32, 32, 3, generator=g), torch.rand(4, 32, 32, 48, generator=g)), -1)     # LR frame | s2d(warped HR)
    with L.variable_scope("generator"):
        out = Fr.generator_F(gi.cuda(), 3, reuse=False, FLAGS=FL)
    ref = ON.generator_F({k: v.double() for k, v in cpu_vars("generator/").items()}, gi.double(), 16)
    assert float(ref.abs().max()) > 1.0                                        # raw xavier: the output is not a small residual
    # 35 fp32 convolutions deep with values up to 12: pixels near zero are differences of large terms, and the fp32 ORACLE itself is
    # not within 1e-3 of its float64 run on every pixel -- the bound is derived from it as in the BASELINE-size step tests:
    # max(1e-3, 1.5 x the fp32 oracle's own worst pixel), measured 1.3e-3 against the oracle's 2.2e-3
    from util import per_elem_err
    o32 = ON.generator_F(cpu_vars("generator/"), gi, 16)
    worst = per_elem_err(out, ref, 1e-3).max().item()
    worst_o = per_elem_err(o32, ref, 1e-3).max().item()
    print("\n[generator_F raw xavier, [4,32,32], nres 16] worst per-pixel error %.2e (fp32 oracle: %.2e), frame maximum %.2f"
          % (worst, worst_o, float(ref.abs().max())))
    assert worst <= max(1e-3, 1.5 * worst_o), (worst, worst_o)
    assert float((per_elem_err(out, ref, 1e-3) > 1e-3).double().mean()) < 1e-3        # ... and all but a handful of pixels within 1e-3


def test_teco_discriminator_vgg_and_network_tuple():
    import lib.ops as L
    import lib.Teco as T
    from tecogan_amd.flags import tecogan_flags
    FL = tecogan_flags(batch_size=1, RNN_N=3, crop_size=16, num_resblock=1, act_dtype="f32")
    di = rnd(2, 32, 32, 27, seed=8)
    with L.variable_scope("tdiscriminator"):
        prob, layers = T.discriminator_F(di.cuda(), FLAGS=FL)
    rp, rl = ON.discriminator_F(cpu_vars("tdiscriminator/"), di)
    close(prob, rp, 1e-4)
    for a, b in zip(layers, rl):
        close(a, b, 1e-4)
    img = rnd(1, 32, 32, 3, seed=9)
    feats = T.VGG19_slim(img.cuda(), reuse=False, deep_list=list(ON.VGG_TAPS))
    Pv = {k.replace("/Conv", ""): v for k, v in cpu_vars("vgg_19/").items()}
    ref = ON.vgg19_features(Pv, img)
    assert set(feats) == set(ON.VGG_TAPS)
    for k in ON.VGG_TAPS:
        close(feats[k], ref[k], 1e-4)
    x = torch.rand(1, 3, 16, 16, 3).cuda()
    y = (torch.rand(1, 3, 64, 64, 3) * 2 - 1).cuda()
    net = T.TecoGAN(x, y, FL, GAN_Flag=True)
    assert net._fields == ('gen_output', 'train', 'learning_rate', 'update_list', 'update_list_name',
                           'update_list_avg', 'image_summary', 'global_step')
    net.train()
    net.train()
    torch.cuda.synchronize()
    assert net.global_step() == 2
    names = net.update_list_name()
    for need in ("l2_content_loss", "l2_warp_loss", "vgg_all", "PingPang", "t_adversarial_loss", "t_discrim_loss",
                 "All_loss_Gen", "t_balance", "withD_counter", "w_o_D_counter"):
        assert need in names, need
    assert len(net.update_list_avg()) == len(names)
    assert net.gen_output().shape == (5, 64, 64, 3) and torch.isfinite(net.gen_output()).all()
    fr = T.FRVSR(x, y, tecogan_flags(batch_size=1, RNN_N=3, crop_size=16, num_resblock=1, pingpang=False,
                                     vgg_scaling=-0.2, ratio=-0.01, act_dtype="f32"))
    fr.train()
    assert fr.gen_output().shape == (3, 64, 64, 3)


def test_main_inference_and_training_cli(tmp_path):
    from PIL import Image
    lr_dir = tmp_path / "LR" / "clip"
    lr_dir.mkdir(parents=True)
    rs = np.random.RandomState(0)
    for i in range(1, 9):
        Image.fromarray(rs.randint(0, 255, (20, 28, 3), dtype=np.uint8)).save(lr_dir / ("%04d.png" % i))
    out = tmp_path / "results"
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--cudaID", "0", "--output_dir", str(out),
                        "--summary_dir", str(out / "log"), "--mode", "inference", "--input_dir_LR", str(lr_dir),
                        "--output_pre", "clip", "--num_resblock", "2", "--checkpoint", "random", "--output_ext", "png"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "total time" in r.stdout and "frame number 13" in r.stdout          # 8 frames + 5 mirrored warm-up
    pngs = sorted(os.listdir(out / "clip"))
    assert len(pngs) == 8 and pngs[0] == "output_0001.png"
    assert Image.open(out / "clip" / pngs[0]).size == (112, 80)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--output_dir", str(tmp_path / "ex"), "--mode", "inference"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode != 0 and "checkpoint file is needed" in r.stderr
    tr = tmp_path / "train"
    base = [sys.executable, os.path.join(ROOT, "main.py"), "--output_dir", str(tr), "--mode", "train", "--synthetic",
            "--batch_size", "1", "--RNN_N", "3", "--crop_size", "16", "--num_resblock", "1", "--display_freq", "2",
            "--save_freq", "2", "--vgg_scaling", "0.2", "--ratio", "0.01", "--pingpang", "--pp_scaling", "0.5"]
    r = subprocess.run(base + ["--max_iter", "4"], capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "image/sec" in r.stdout and "x05" in r.stdout and "Optimization done" in r.stdout
    assert {"model-0", "model-2", "model-4"} <= set(os.listdir(tr))
    r = subprocess.run(base + ["--max_iter", "2", "--checkpoint", str(tr / "model-4"), "--nopre_trained_model"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0 and "global_step 6" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_fused_data_step_and_output_step(tmp_path):
    """Gaussian down-sampling + target crop + preprocess in one launch (reference lib/ops.py:347-367,
    lib/dataloader.py:306-332) and the uint8 output conversion with the asynchronous frame writer (lib/ops.py:521-523)."""
    import numpy as np
    from PIL import Image
    import lib.ops as ops
    from tecogan_amd.output import FrameWriter
    g = torch.Generator().manual_seed(4)
    hr = torch.rand(3, 40, 48, 3, generator=g)
    lr, tgt = ops.gauss_down_crop_preprocess(hr.cuda(), 1.5)
    k = torch.tensor(ops.gaussian_2dkernel(9, 1.5), dtype=torch.float32)
    w = torch.zeros(3, 1, 9, 9)
    w[:] = k
    ref = torch.nn.functional.conv2d(hr.permute(0, 3, 1, 2), w, stride=4, groups=3).permute(0, 2, 3, 1)
    assert tuple(lr.shape) == (3, 8, 10, 3) and tuple(tgt.shape) == (3, 32, 40, 3)
    assert (lr.cpu() - ref).abs().max().item() < 2e-6
    assert torch.equal(tgt.cpu(), hr[:, 4:36, 4:44] * 2 - 1)                     # preprocess: bit-exact
    assert (ops.tf_data_gaussDownby4(hr.cuda(), 1.5).cpu() - ref).abs().max().item() < 2e-6
    # output step: truncating uint8 conversion, RGB and BGR, incl. out-of-range values
    fr = (torch.rand(1, 36, 52, 3, generator=g) * 1.4 - 0.2)
    u8 = torch.empty(36, 52, 3, dtype=torch.uint8, device="cuda")
    from tecogan_amd import kernels as K
    K.frame_to_u8(fr[0].cuda(), u8)
    want = np.clip(fr[0].numpy() * 255.0, 0, 255).astype(np.uint8)
    assert np.array_equal(u8.cpu().numpy(), want)
    K.frame_to_u8(fr[0].cuda(), u8, bgr=True)
    assert np.array_equal(u8.cpu().numpy(), want[:, :, ::-1])
    wr = FrameWriter((36, 52, 3), slots=2)
    for i in range(5):
        wr.submit(str(tmp_path / ("f%d.png" % i)), (fr[0] * (0.5 + 0.1 * i)).cuda())
    wr.close()
    for i in range(5):
        got = np.asarray(Image.open(tmp_path / ("f%d.png" % i)))
        assert np.array_equal(got, np.clip((fr[0] * (0.5 + 0.1 * i)).numpy() * 255.0, 0, 255).astype(np.uint8)), i
