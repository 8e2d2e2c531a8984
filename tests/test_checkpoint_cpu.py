"""TensorFlow tensor-bundle reader/writer (tecogan_amd/tf_bundle.py) and the checkpoint front end -- CPU only.
The format is restated from the TensorFlow sources (no TF-written file is available offline): these tests pin the
primitive encodings with hand-built bytes and check round trips."""
import os
import struct
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tecogan_amd import tf_bundle as TB                      # noqa: E402


def test_crc32c_check_values_and_masking():
    assert TB.crc32c(b"123456789") == 0xE3069283                     # the standard CRC-32C check value
    assert TB.crc32c(b"") == 0
    assert TB.crc32c(b"\x00" * 32) == 0x8A9136AA                      # RFC 3720 B.4 test vector
    assert TB.crc32c(b"\xff" * 32) == 0x62A8AB43                      # RFC 3720 B.4 test vector
    assert TB.crc32c(b"6789", TB.crc32c(b"12345")) == 0xE3069283      # incremental
    for c in (0, 1, 0xE3069283, 0xFFFFFFFF):
        assert TB.unmask_crc(TB.mask_crc(c)) == c
    assert TB.mask_crc(0) == 0xA282EAD8


def test_varint_and_shape_proto_encoding():
    assert TB._put_varint(0) == b"\x00" and TB._put_varint(300) == b"\xac\x02"
    assert TB._get_varint(b"\xac\x02", 0) == (300, 2)
    assert TB._put_varint(-1) == b"\xff" * 9 + b"\x01"                 # int64 -1: ten-byte two's complement
    # TensorShapeProto{dim{size:3} dim{size:64}} = 12 02 08 03 12 02 08 40
    assert TB._encode_shape((3, 64)) == bytes([0x12, 2, 8, 3, 0x12, 2, 8, 64])
    assert TB._decode_shape(bytes([0x12, 2, 8, 3, 0x12, 2, 8, 64])) == (3, 64)
    assert TB._decode_shape(b"") == ()


def test_block_prefix_compression_hand_built():
    # two entries "apple"->"1", "apply"->"22" (shares "appl"), one restart at 0
    block = bytes([0, 5, 1]) + b"apple" + b"1" + bytes([4, 1, 2]) + b"y" + b"22" + struct.pack("<II", 0, 1)
    assert list(TB._block_entries(block)) == [(b"apple", b"1"), (b"apply", b"22")]
    assert TB._build_block([(b"apple", b"1"), (b"apply", b"22")], 16) == block


def test_snappy_raw_decompress_hand_built():
    # 20 x 'a': varint length 20, literal 'a', copy(2-byte offset form) of 19 bytes from offset 1
    assert TB._snappy_decompress(bytes([20, 0x00, ord("a"), (18 << 2) | 2, 1, 0])) == b"a" * 20
    # literal with a one-byte length field (tag 60<<2): 61 bytes
    payload = bytes(range(61))
    assert TB._snappy_decompress(bytes([61, 60 << 2, 60]) + payload) == payload


def test_bundle_round_trip_many_blocks(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {"generator/generator_unit/resblock_%d/conv_1/Conv/weights" % i: rng.standard_normal((3, 3, 4, 5)).astype(np.float32)
               for i in range(40)}
    tensors["global_step"] = np.asarray(1234, dtype=np.int64)
    tensors["fnet/autoencode_unit/encoder_1/conv_1/Conv/biases"] = np.zeros(32, np.float32)
    tensors["flags/some_int32"] = np.arange(6, dtype=np.int32).reshape(2, 3)
    prefix = str(tmp_path / "model-7")
    TB.write_bundle(prefix, tensors, block_size=256)                   # small blocks: many data blocks + a real index block
    assert TB.is_bundle(prefix)
    r = TB.BundleReader(prefix)
    assert r.header["num_shards"] == 1 and r.keys() == sorted(tensors, key=lambda s: s.encode())
    assert r.shape("global_step") == () and r.shape("flags/some_int32") == (2, 3)
    back = TB.read_bundle(prefix, verify=True)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k
    # footer: 48 bytes ending in the table magic
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xdb4775248b80fb57


def test_bundle_detects_corruption(tmp_path):
    prefix = str(tmp_path / "m")
    TB.write_bundle(prefix, {"a": np.ones(4, np.float32), "b": np.zeros((2, 2), np.float32)})
    raw = bytearray(open(prefix + ".index", "rb").read())
    raw[3] ^= 0x40
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        TB.BundleReader(prefix)
    TB.write_bundle(prefix, {"a": np.ones(4, np.float32)})
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[0] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError):
        TB.read_bundle(prefix, verify=True)


def test_bfloat16_entries_are_widened(tmp_path):
    prefix = str(tmp_path / "bf")
    x = np.array([1.0, -2.5, 3.140625], np.float32)
    TB.write_bundle(prefix, {"w": (x.view(np.uint32) >> 16).astype(np.uint16)})
    r = TB.BundleReader(prefix)
    r.entries["w"]["dtype"] = TB.DT_BFLOAT16                           # as a TF bfloat16 variable would be tagged
    assert np.array_equal(r.get("w"), x)


def test_checkpoint_front_end_reads_both_forms(tmp_path):
    from tecogan_amd.checkpoint import load_variables
    vals = {"generator/generator_unit/input_stage/conv/Conv/weights": torch.randn(3, 3, 51, 64),
            "generator/generator_unit/input_stage/conv/Conv/biases": torch.zeros(64)}
    tpath = str(tmp_path / "model-5")
    torch.save({"variables": vals, "global_step": 5}, tpath)
    got, extra = load_variables(tpath)
    assert extra["global_step"] == 5 and all(torch.equal(got[k], v) for k, v in vals.items())
    prefix = str(tmp_path / "TecoGAN")
    b = {k: v.numpy() for k, v in vals.items()}
    k0 = "generator/generator_unit/input_stage/conv/Conv/weights"
    b["generator_train/%s/Adam" % k0] = np.full((3, 3, 51, 64), 0.5, np.float32)
    b["generator_train/%s/Adam_1" % k0] = np.full((3, 3, 51, 64), 0.25, np.float32)
    b["generator_train/beta1_power"] = np.asarray(0.9, np.float32)
    b["global_step"] = np.asarray(77, np.int64)
    TB.write_bundle(prefix, b)
    got, extra = load_variables(prefix)
    assert set(got) == set(vals) and all(torch.equal(got[k], v) for k, v in vals.items())
    assert extra["global_step"] == 77 and float(extra["adam_m"][k0].mean()) == 0.5 and float(extra["adam_v"][k0].mean()) == 0.25
    with pytest.raises(ValueError):
        load_variables(str(tmp_path / "missing"))


def test_crc32c_large_input_path_matches_bytewise():
    rng = np.random.default_rng(1)
    for n in (65536, 65537, 200003):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        ref = TB._crc_raw_small(d, 0xFFFFFFFF) ^ 0xFFFFFFFF
        assert TB.crc32c(d) == ref
        assert TB.crc32c(d[n // 3:], TB.crc32c(d[:n // 3])) == ref


def test_parameter_store_round_trips_through_a_bundle(tmp_path):
    """save_bundle() writes variables + TF-named Adam slots + global_step; load_variables() returns them."""
    from collections import OrderedDict
    from tecogan_amd import params as P
    from tecogan_amd.checkpoint import load_variables, save_bundle
    specs = OrderedDict(generator=P.generator_spec(1), fnet=P.fnet_spec())
    ps = P.ParamStore(specs, "cpu")
    g = torch.Generator().manual_seed(3)
    ps.flat.copy_(torch.randn(ps.numel, generator=g))
    ps.m.copy_(torch.randn(ps.numel, generator=g))
    ps.v.copy_(torch.rand(ps.numel, generator=g))
    prefix = str(tmp_path / "model-12")
    save_bundle(prefix, ps, 12)
    r = TB.BundleReader(prefix)
    name = "fnet/autoencode_unit/decoder_2/conv_1/Conv/weights"
    assert r.shape(name) == (3, 3, 256, 128) and r.shape("global_step") == ()
    assert "generator_train/%s/Adam_1" % name in r.keys() and "generator_train/beta1_power" in r.keys()
    got, extra = load_variables(prefix)
    assert set(got) == set(ps.entries) and extra["global_step"] == 12
    for n in ps.entries:
        assert torch.equal(got[n], ps.view(n)), n
        assert torch.equal(extra["adam_m"][n], ps.view(n, ps.m)) and torch.equal(extra["adam_v"][n], ps.view(n, ps.v)), n
    assert abs(float(r.get("generator_train/beta2_power")) - 0.999 ** 13) < 1e-7
    assert extra["adam_steps"] == {"generator": 12, "fnet": 12}


def test_bundle_resume_carries_adam_counters_and_balance_ema(tmp_path):
    """A full resume needs each optimiser's Adam step count (the gated discriminator lags global_step) and the
    t_balance EMA (reference lib/Teco.py:415-417,425,493-494): written as TF's beta1_power accumulators + one own key."""
    from collections import OrderedDict
    from tecogan_amd import params as P
    from tecogan_amd.checkpoint import load_variables, save_bundle
    specs = OrderedDict(generator=P.generator_spec(1), fnet=P.fnet_spec(), tdiscriminator=P.discriminator_spec())
    ps = P.ParamStore(specs, "cpu")
    prefix = str(tmp_path / "model-40")
    save_bundle(prefix, ps, 40, beta1=0.9, adam_steps={"tdiscriminator": 17, "generator": 40, "fnet": 40}, tb_ema=0.3125)
    _, extra = load_variables(prefix)
    assert extra["global_step"] == 40
    assert extra["adam_steps"] == {"tdiscriminator": 17, "generator": 40, "fnet": 40}
    assert abs(extra["tb_ema"] - 0.3125) < 1e-7
    # optimiser-state names as the reference graph creates them (lib/Teco.py:438,463-468): every slot under 'generator_train',
    # bias-correction accumulators uniquified in creation order D, G, FNet
    from tecogan_amd.tf_bundle import BundleReader
    keys = set(BundleReader(prefix).keys())
    d0 = "tdiscriminator/discriminator_unit/input_stage/conv/Conv/weights"
    assert "generator_train/" + d0 + "/Adam" in keys and "generator_train/" + d0 + "/Adam_1" in keys
    assert not [k for k in keys if k.startswith("tdicriminator_train")]
    assert abs(float(BundleReader(prefix).get("generator_train/beta1_power")) - 0.9 ** 18) < 1e-7        # D: 17 updates
    assert abs(float(BundleReader(prefix).get("generator_train/beta1_power_2")) - 0.9 ** 41) < 1e-7      # FNet: 40
    assert torch.equal(extra["adam_m"][d0], ps.view(d0, ps.m))


def test_bundle_resume_recovers_long_run_adam_counts(tmp_path):
    """float32 beta1_power = 0.9^(t+1) underflows near t ~ 980, so the count must come from the exact integer key
    this backend writes, or -- for a file without it -- from beta2_power = 0.999^(t+1), which stays normal up to t ~ 8e4."""
    from collections import OrderedDict
    from tecogan_amd import params as P, tf_bundle
    from tecogan_amd.checkpoint import STEPS_KEY, load_variables, save_bundle
    specs = OrderedDict(generator=P.generator_spec(1), fnet=P.fnet_spec(), tdiscriminator=P.discriminator_spec())
    ps = P.ParamStore(specs, "cpu")
    prefix = str(tmp_path / "model-50000")
    want = {"tdiscriminator": 31234, "generator": 50000, "fnet": 50000}
    save_bundle(prefix, ps, 50000, adam_steps=want)
    _, extra = load_variables(prefix)
    assert extra["adam_steps"] == want
    # the same file without the backend's own keys (what a TF-written checkpoint would look like)
    r = tf_bundle.BundleReader(prefix)
    stripped = OrderedDict((k, r.get(k)) for k in r.keys() if not k.startswith(STEPS_KEY))
    assert float(stripped["generator_train/beta1_power"]) == 0.0             # underflowed: useless
    p2 = str(tmp_path / "tf-like")
    tf_bundle.write_bundle(p2, stripped)
    _, extra2 = load_variables(p2)
    for sc, t in want.items():
        assert abs(extra2["adam_steps"][sc] - t) <= max(2, int(2e-4 * t)), (sc, extra2["adam_steps"][sc], t)


def test_round2_style_bundle_maps_the_adam_counts_to_the_right_optimisers(tmp_path):
    """a bundle in the round-2 layout (D's accumulators under `tdicriminator_train/`, generator / FNet as
    beta1_power / beta1_power_1) must not be read with today's mapping, which would hand the generator's count to the
    (gated, lagging) discriminator."""
    from collections import OrderedDict
    from tecogan_amd import params as P, tf_bundle
    from tecogan_amd.checkpoint import load_variables
    specs = OrderedDict(generator=P.generator_spec(1), fnet=P.fnet_spec(), tdiscriminator=P.discriminator_spec())
    ps = P.ParamStore(specs, "cpu")
    b = OrderedDict((n, ps.view(n).numpy()) for n in ps.entries)
    want = {"tdiscriminator": 17, "generator": 40, "fnet": 39}
    b["generator_train/beta1_power"] = np.asarray(0.9 ** (want["generator"] + 1), np.float32)
    b["generator_train/beta1_power_1"] = np.asarray(0.9 ** (want["fnet"] + 1), np.float32)
    b["tdicriminator_train/beta1_power"] = np.asarray(0.9 ** (want["tdiscriminator"] + 1), np.float32)
    b["global_step"] = np.asarray(40, np.int64)
    prefix = str(tmp_path / "model-40")
    tf_bundle.write_bundle(prefix, b)
    _, extra = load_variables(prefix)
    assert extra["adam_steps"] == want


def test_loader_cache_budget_is_shared_by_the_ranks_of_a_node(tmp_path, monkeypatch):
    """every rank builds a training and a validation loader; their decoded-frame caches together must stay
    within RAM/4 per NODE (it was RAM/4 per loader: 8 ranks x 2 loaders = 4x the machine)."""
    from PIL import Image
    import lib.dataloader as DL
    from tecogan_amd.flags import tecogan_flags
    root = tmp_path / "scenes"
    for sc in (2000, 2001):
        d = root / ("scene_%04d" % sc)
        d.mkdir(parents=True)
        for f in range(12):
            Image.fromarray(np.zeros((160, 176, 3), np.uint8)).save(str(d / ("col_high_%04d.png" % f)))
    F = tecogan_flags(input_video_dir=str(root), input_video_pre="scene", str_dir=2000, end_dir=2000, end_dir_val=2001, max_frm=11,
                      batch_size=1, RNN_N=3, crop_size=32)
    monkeypatch.delenv("TG_LOADER_CACHE_GB", raising=False)
    caps = {}
    for ranks in (1, 8):
        monkeypatch.setenv("LOCAL_WORLD_SIZE", str(ranks))
        tr = DL.SceneSequences(F, "cpu", 2000, 2000, cache_share=15.0 / 16)
        va = DL.SceneSequences(F, "cpu", 2001, 2001, cache_share=1.0 / 16)
        caps[ranks] = (tr._cache_cap, va._cache_cap)
    ram = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
    assert abs(sum(caps[1]) - ram / 4) < 0.01 * ram and abs(8 * sum(caps[8]) - ram / 4) < 0.01 * ram
    assert caps[8][1] < caps[8][0] / 8


def test_data_parallel_ranks_draw_different_batches():
    """Every rank of a data-parallel run must see its own data stream (otherwise the averaged gradient equals a
    single-GPU step): the loader's seed is offset by the rank."""
    from lib.dataloader import SyntheticSequences
    from tecogan_amd.flags import frvsr_flags
    F = frvsr_flags(batch_size=1, RNN_N=2, crop_size=8)
    a = SyntheticSequences(F, "cpu", seed=1234 + 0).next_batch()
    b = SyntheticSequences(F, "cpu", seed=1234 + 1).next_batch()
    assert not torch.equal(a[0], b[0]) and not torch.equal(a[1], b[1])


def _v1_file(path, tensors):
    """Minimal writer of the V1 tensor-slice layout (test helper): meta entry under key "", one SavedSlice per tensor with
    packed float_val / tensor_content, keys in the ordered-code form TF uses (0 prefix, escaped name, dims, full extents)."""
    def lenfield(f, payload):
        return TB._field(f, 2, TB._put_varint(len(payload)) + payload)
    metas, items = b"", []
    for name, a in sorted(tensors.items()):
        shp = TB._encode_shape(a.shape)
        sm = lenfield(1, name.encode()) + lenfield(2, shp) + TB._field(3, 0, TB._put_varint(TB.NP_TO_DT[a.dtype]))
        sm += lenfield(4, b"".join(lenfield(1, b"") for _ in a.shape))
        metas += lenfield(1, sm)
        tp = TB._field(1, 0, TB._put_varint(TB.NP_TO_DT[a.dtype])) + lenfield(2, shp)
        tp += lenfield(5, a.astype("<f4").tobytes()) if a.dtype == np.float32 else lenfield(4, a.tobytes())
        sl = lenfield(1, name.encode()) + lenfield(2, b"".join(lenfield(1, b"") for _ in a.shape)) + lenfield(3, tp)
        key = b"\x00" + name.encode().replace(b"\x00", b"\x00\xff") + b"\x00\x01" + bytes([1, a.ndim]) + b"\x80\x7f" * a.ndim
        items.append((key, lenfield(2, sl)))
    items.insert(0, (b"", lenfield(1, metas)))
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block + b"\x00" + struct.pack("<I", TB.mask_crc(TB.crc32c(block + b"\x00"))))
        return off, len(block)
    index = []
    for kv in items:                                                   # one entry per block, like big tensors in real files
        off, size = emit(TB._build_block([kv], 16))
        index.append((kv[0], TB._handle(off, size)))
    moff, msize = emit(TB._build_block([], 1))
    ioff, isize = emit(TB._build_block(index, 1))
    foot = TB._handle(moff, msize) + TB._handle(ioff, isize)
    out.extend(foot + b"\x00" * (40 - len(foot)) + struct.pack("<Q", TB.MAGIC))
    open(path, "wb").write(bytes(out))


def test_v1_tensor_slice_checkpoint_reader(tmp_path):
    from tecogan_amd.checkpoint import load_variables
    rng = np.random.default_rng(5)
    tensors = {"vgg_19/conv1/conv1_1/weights": rng.standard_normal((3, 3, 3, 64)).astype(np.float32),
               "vgg_19/conv1/conv1_1/biases": rng.standard_normal(64).astype(np.float32),
               "global_step": np.asarray([7], np.int64).reshape(())}
    path = str(tmp_path / "vgg_19.ckpt")
    _v1_file(path, tensors)
    assert TB.is_v1_checkpoint(path)
    got = TB.read_v1_checkpoint(path)
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].shape == v.shape and np.array_equal(got[k], v), k
    only = TB.read_v1_checkpoint(path, names=["vgg_19/conv1/conv1_1/biases"])
    assert list(only) == ["vgg_19/conv1/conv1_1/biases"]
    variables, _ = load_variables(path)                                # the front end keeps the float variables
    assert set(variables) == {"vgg_19/conv1/conv1_1/weights", "vgg_19/conv1/conv1_1/biases"}
    assert torch.equal(variables["vgg_19/conv1/conv1_1/weights"], torch.from_numpy(tensors["vgg_19/conv1/conv1_1/weights"]))


TF_BUNDLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_bundle")
TF_BUNDLE_EXPECTED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_bundle_expected.npz")


@pytest.mark.skipif(not os.path.exists(TF_BUNDLE_EXPECTED),
                    reason="tests/golden/tf_bundle* absent: run tools/make_tf_goldens.py on a box with TensorFlow 1.x")
def test_reads_a_tensorflow_written_bundle():
    """a checkpoint written by TensorFlow's own Saver (tools/make_tf_goldens.py: the reference's variable names,
    an AdamOptimizer under variable_scope('generator_train'), global_step = 3) through tf_bundle.BundleReader and the
    checkpoint front end -- every variable bit for bit, the Adam slots under their TF names, the step counts recovered."""
    import glob
    from tecogan_amd.checkpoint import load_variables
    idx = glob.glob(os.path.join(TF_BUNDLE, "model-*.index"))
    assert idx, "no bundle under " + TF_BUNDLE
    prefix = idx[0][:-len(".index")]
    want = {k.replace("|", "/"): v for k, v in np.load(TF_BUNDLE_EXPECTED).items()}
    r = TB.BundleReader(prefix)
    assert set(want) <= set(r.keys()), sorted(set(want) - set(r.keys()))
    for k, v in want.items():
        got = r.get(k)
        assert got.shape == v.shape and np.array_equal(got, v), k
    variables, extra = load_variables(prefix)
    assert extra["global_step"] == 3
    w = "generator/generator_unit/input_stage/conv/Conv/weights"
    assert torch.equal(variables[w], torch.from_numpy(want[w]))
    assert torch.equal(extra["adam_m"][w], torch.from_numpy(want["generator_train/" + w + "/Adam"]))
    assert torch.equal(extra["adam_v"][w], torch.from_numpy(want["generator_train/" + w + "/Adam_1"]))
    assert extra["adam_steps"]["generator"] == 3
