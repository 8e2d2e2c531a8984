#!/usr/bin/env python
"""Pin the CPU oracle (oracle/) to REAL TensorFlow -- to be run ONCE on any box that has TensorFlow 1.x with tf.contrib
(1.8 <= version <= 1.15, what the reference needs: lib/ops.py:1-6); this container has no TensorFlow and no network, so the
script is committed together with the tests that consume its output:

    python tools/make_tf_goldens.py [--reference /path/to/TecoGAN] [--out tests/golden]
    python -m pytest tests/test_oracle_kat.py tests/test_checkpoint_cpu.py -q          # "parity unpinned" -> pinned

It writes
  <out>/tf_ops.npz        inputs, outputs and gradients of the TensorFlow ops whose semantics live INSIDE TensorFlow at the
                          reference's own call sites: slim.conv2d (lib/ops.py:51-56, k3 s1 / k4 s2 SAME),
                          slim.conv2d_transpose (lib/ops.py:39-44, k3 s2 SAME), slim.batch_norm (lib/ops.py:89-90, training
                          mode, no scale, fused), slim.max_pool2d (lib/ops.py:93), tf.image.resize_images (lib/frvsr.py:22,
                          lib/Teco.py:244), tf.contrib.image.dense_image_warp (lib/Teco.py:120,140,224,254), tf.space_to_depth
                          (main.py:201), tf.train.AdamOptimizer / ExponentialMovingAverage / exponential_decay
                          (lib/Teco.py:95-99,415-417,425), and -- with --reference -- the reference's own upscale_four /
                          bicubic_four (lib/ops.py:126-212);
  <out>/tf_bundle/model-3.{index,data-00000-of-00001} + tf_bundle_expected.npz
                          a Saver.save of a tiny graph with the reference's variable names, one AdamOptimizer under
                          variable_scope('generator_train') and global_step (main.py:307,365): what tf_bundle.py / checkpoint.py
                          must read.
Everything is seeded; tests/test_oracle_kat.py::test_oracle_matches_tensorflow_goldens and
tests/test_checkpoint_cpu.py::test_reads_a_tensorflow_written_bundle skip while these files are absent."""
import argparse
import os
import sys

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="", help="checkout of thunil/TecoGAN (adds upscale_four / bicubic_four from its lib/ops.py)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    a = ap.parse_args()
    import tensorflow as tf
    if not tf.__version__.startswith("1."):
        raise SystemExit("TensorFlow 1.x with tf.contrib is required (found %s)" % tf.__version__)
    import tensorflow.contrib.slim as slim
    rng = np.random.RandomState(20260921)
    G = {"tf_version": np.asarray(tf.__version__)}

    def rnd(*shape, scale=1.0):
        return ((rng.rand(*shape) * 2 - 1) * scale).astype(np.float32)

    def run(fetches, feed=None):
        with tf.Session() as sess:
            sess.run(tf.global_variables_initializer())
            return sess.run(fetches, feed)

    def grads(y, xs, gy):
        return tf.gradients(y, xs, grad_ys=tf.constant(gy))

    # ---- slim.conv2d, exactly as lib/ops.py:51-56 calls it (SAME, NHWC, no activation) ---------------------------------
    for tag, (N, H, W, Ci, Co, k, s) in {"conv_k3s1": (2, 9, 11, 5, 7, 3, 1), "conv_k4s2_even": (2, 8, 12, 5, 6, 4, 2),
                                         "conv_k4s2_odd": (1, 9, 7, 3, 4, 4, 2)}.items():
        tf.reset_default_graph()
        x, w, b = rnd(N, H, W, Ci), rnd(k, k, Ci, Co, scale=0.3), rnd(Co)
        xt = tf.constant(x)
        with tf.variable_scope(tag):
            y = slim.conv2d(xt, Co, [k, k], s, 'SAME', data_format='NHWC', activation_fn=None,
                            weights_initializer=tf.constant_initializer(w), biases_initializer=tf.constant_initializer(b))
        wv, bv = [v for v in tf.global_variables() if v.name.endswith("weights:0")][0], [v for v in tf.global_variables() if v.name.endswith("biases:0")][0]
        ys = run(y)
        gy = rnd(*ys.shape)
        yv, dx, dw, db = run([y] + grads(y, [xt, wv, bv], gy))
        G.update({tag + "/x": x, tag + "/w": w, tag + "/b": b, tag + "/stride": np.asarray(s), tag + "/y": yv, tag + "/gy": gy,
                  tag + "/dx": dx, tag + "/dw": dw, tag + "/db": db})

    # ---- slim.conv2d_transpose, lib/ops.py:39-44 (k3 s2 SAME; filter [kh,kw,Cout,Cin]) -------------------------------
    tf.reset_default_graph()
    x, w, b = rnd(2, 5, 7, 4), rnd(3, 3, 6, 4, scale=0.3), rnd(6)
    xt = tf.constant(x)
    with tf.variable_scope("deconv"):
        y = slim.conv2d_transpose(xt, 6, [3, 3], 2, 'SAME', data_format='NHWC', activation_fn=None,
                                  weights_initializer=tf.constant_initializer(w), biases_initializer=tf.constant_initializer(b))
    wv, bv = [v for v in tf.global_variables() if "weights" in v.name][0], [v for v in tf.global_variables() if "biases" in v.name][0]
    ys = run(y)
    gy = rnd(*ys.shape)
    yv, dx, dw, db = run([y] + grads(y, [xt, wv, bv], gy))
    G.update({"deconv/x": x, "deconv/w": w, "deconv/b": b, "deconv/y": yv, "deconv/gy": gy, "deconv/dx": dx, "deconv/dw": dw, "deconv/db": db})

    # ---- slim.batch_norm as lib/ops.py:89-90 (training mode, scale=False, fused, eps 1e-3, decay 0.9) -----------------
    tf.reset_default_graph()
    x, beta = rnd(3, 6, 5, 8, scale=2.0) + 0.5, rnd(8)
    xt = tf.constant(x)
    y = slim.batch_norm(xt, decay=0.9, epsilon=0.001, updates_collections=tf.GraphKeys.UPDATE_OPS, scale=False, fused=True,
                        is_training=True, param_initializers={"beta": tf.constant_initializer(beta)})
    bvar = [v for v in tf.global_variables() if "beta" in v.name][0]
    mm = [v for v in tf.global_variables() if "moving_mean" in v.name][0]
    mv = [v for v in tf.global_variables() if "moving_variance" in v.name][0]
    gy = rnd(3, 6, 5, 8)
    with tf.Session() as sess:
        sess.run(tf.global_variables_initializer())
        yv, dx, dbeta = sess.run([y] + grads(y, [xt, bvar], gy))
        sess.run(tf.get_collection(tf.GraphKeys.UPDATE_OPS))
        mmv, mvv = sess.run([mm, mv])
    G.update({"bn/x": x, "bn/beta": beta, "bn/y": yv, "bn/gy": gy, "bn/dx": dx, "bn/dbeta": dbeta, "bn/moving_mean": mmv,
              "bn/moving_variance": mvv})

    # ---- slim.max_pool2d [2,2] (lib/ops.py:93) on odd sizes -----------------------------------------------------------
    tf.reset_default_graph()
    x = rnd(1, 7, 9, 3)
    xt = tf.constant(x)
    y = slim.max_pool2d(xt, [2, 2])
    gy = rnd(1, 3, 4, 3)
    yv, dx = run([y] + grads(y, [xt], gy))
    G.update({"maxpool/x": x, "maxpool/y": yv, "maxpool/gy": gy, "maxpool/dx": dx})

    # ---- tf.image.resize_images (bilinear default; lib/frvsr.py:22 x2, lib/Teco.py:244 x4) ---------------------------------
    for tag, f in (("resize2", 2), ("resize4", 4)):
        tf.reset_default_graph()
        x = rnd(2, 5, 6, 3)
        xt = tf.constant(x)
        y = tf.image.resize_images(xt, (5 * f, 6 * f))
        gy = rnd(2, 5 * f, 6 * f, 3)
        yv, dx = run([y] + grads(y, [xt], gy))
        G.update({tag + "/x": x, tag + "/y": yv, tag + "/gy": gy, tag + "/dx": dx})

    # ---- tf.contrib.image.dense_image_warp (lib/Teco.py:120,140) with its gradients -----------------------------------
    tf.reset_default_graph()
    img, flow = rnd(2, 6, 7, 3), rnd(2, 6, 7, 2, scale=2.5)
    flow[0, 0, 0] = (0.0, 0.0)
    flow[0, 1, 1] = (1.0, -2.0)                  # integer displacement: the gradient tie rule
    flow[1, 5, 6] = (-9.0, 9.0)                  # far outside: clamped
    it, ft = tf.constant(img), tf.constant(flow)
    y = tf.contrib.image.dense_image_warp(it, ft)
    gy = rnd(2, 6, 7, 3)
    yv, dimg, dflow = run([y] + grads(y, [it, ft], gy))
    G.update({"warp/img": img, "warp/flow": flow, "warp/y": yv, "warp/gy": gy, "warp/dimg": dimg, "warp/dflow": dflow})

    # ---- tf.space_to_depth(4) (main.py:201) ---------------------------------------------------------------------------------
    tf.reset_default_graph()
    x = rnd(1, 8, 12, 3)
    G.update({"s2d/x": x, "s2d/y": run(tf.space_to_depth(tf.constant(x), 4))})

    # ---- AdamOptimizer (lib/Teco.py:425: beta1 0.9, beta2 0.999, eps 1e-8), exponential_decay, EMA(0.99) ------------------
    tf.reset_default_graph()
    p0, g0 = rnd(5), rnd(5, scale=0.1)
    var = tf.Variable(p0)
    gs = tf.train.get_or_create_global_step() if hasattr(tf.train, "get_or_create_global_step") else tf.contrib.framework.get_or_create_global_step()
    lr = tf.train.exponential_decay(5e-5, gs, 2, 0.5, staircase=True)
    opt = tf.train.AdamOptimizer(lr, beta1=0.9, beta2=0.999, epsilon=1e-8)
    step = opt.apply_gradients([(tf.constant(g0), var)], global_step=gs)
    ema = tf.train.ExponentialMovingAverage(0.99)
    val = tf.Variable(0.0)
    upd = ema.apply([val])
    traj, lrs, shadows = [], [], []
    with tf.Session() as sess:
        sess.run(tf.global_variables_initializer())
        for it in range(4):
            lrs.append(sess.run(lr))
            sess.run(step)
            traj.append(sess.run(var))
            sess.run(val.assign(float(it + 1)))
            sess.run(upd)
            shadows.append(sess.run(ema.average(val)))
    G.update({"adam/p0": p0, "adam/g": g0, "adam/traj": np.stack(traj), "adam/lr": np.asarray(lrs, np.float32),
              "ema/values": np.arange(1, 5, dtype=np.float32), "ema/shadow": np.asarray(shadows, np.float32)})

    # ---- the reference's own upscale_four / bicubic_four (pure TF ops, lib/ops.py:126-212) -----------------------------------
    if a.reference:
        sys.path.insert(0, a.reference)
        from lib import ops as ref_ops                      # needs keras + cv2 as the reference does
        tf.reset_default_graph()
        x = rnd(2, 5, 6, 3)
        G.update({"ref/x": x, "ref/upscale_four": run(ref_ops.upscale_four(tf.constant(x))),
                  "ref/bicubic_four": run(ref_ops.bicubic_four(tf.constant(x)))})

    os.makedirs(a.out, exist_ok=True)
    np.savez_compressed(os.path.join(a.out, "tf_ops.npz"), **G)
    print("wrote", os.path.join(a.out, "tf_ops.npz"), "(%d arrays)" % len(G))

    # ---- a Saver.save of a tiny graph with the reference's names (main.py:307,365) --------------------------------------------
    tf.reset_default_graph()
    names = {"generator/generator_unit/input_stage/conv/Conv/weights": (3, 3, 51, 4),
             "generator/generator_unit/input_stage/conv/Conv/biases": (4,),
             "fnet/autoencode_unit/encoder_1/conv_1/Conv/weights": (3, 3, 6, 4),
             "tdiscriminator/discriminator_unit/disblock_1/BatchNorm/beta": (4,)}
    vs = {}
    for n, shp in names.items():
        vs[n] = tf.get_variable(n, initializer=tf.constant(rnd(*shp)))
    gs = tf.train.get_or_create_global_step() if hasattr(tf.train, "get_or_create_global_step") else tf.contrib.framework.get_or_create_global_step()
    loss = tf.add_n([tf.reduce_sum(v * v) for v in vs.values()])
    with tf.variable_scope("generator_train"):
        train = tf.train.AdamOptimizer(1e-3, beta1=0.9).minimize(loss, global_step=gs)
    saver = tf.train.Saver(max_to_keep=2)
    bdir = os.path.join(a.out, "tf_bundle")
    os.makedirs(bdir, exist_ok=True)
    with tf.Session() as sess:
        sess.run(tf.global_variables_initializer())
        for _ in range(3):
            sess.run(train)
        saver.save(sess, os.path.join(bdir, "model"), global_step=gs, write_meta_graph=False)
        allv = {v.op.name: sess.run(v) for v in tf.global_variables()}
    np.savez_compressed(os.path.join(a.out, "tf_bundle_expected.npz"), **{k.replace("/", "|"): v for k, v in allv.items()})
    print("wrote", bdir, "and tf_bundle_expected.npz (%d variables)" % len(allv))


if __name__ == "__main__":
    main()
