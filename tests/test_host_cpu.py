"""CPU-side checks: flag surface, C-ABI symbol export, spec parity with the oracle, oracle isolation,
and the data-parallel exchange under gloo with world_size 2."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flags_accept_reference_spellings():
    from tecogan_amd import flags
    a = flags.parse(["--mode", "train", "--nopingpang", "--pre_trained_model", "--stair", "--ratio", "-0.01",
                     "--Dt_mergeDs", "--movingFirstFrame", "--random_crop", "--noflip", "--max_iter", "500000"])
    assert (a.pingpang, a.pre_trained_model, a.stair, a.ratio, a.flip) == (False, True, True, -0.01, False)
    assert len(flags.FLAG_TABLE) == 55
    d = flags.defaults()
    assert (d.num_resblock, d.RNN_N, d.crop_dt, d.Dbalance, d.vgg_scaling) == (16, 10, 0.75, 0.4, -0.002)
    with pytest.raises(ValueError):
        flags.defaults(not_a_flag=1)


def test_rungan_recipes_parse_with_main_flags():
    import runGan
    from tecogan_amd import flags
    for case in (3, 4):
        argv = runGan.to_argv(runGan.COMMON_TRAIN + runGan.RECIPES[case] + runGan.DATA)
        a = flags.parse(argv)
        assert a.batch_size == 4 and a.RNN_N == 10 and a.learning_rate == 0.00005 and a.decay_rate == 1.0
    assert flags.parse(runGan.to_argv(runGan.RECIPES[4])).pingpang is False
    t = flags.tecogan_flags()
    assert (t.pingpang, t.pp_scaling, t.vgg_scaling, t.ratio, t.num_resblock) == (True, 0.5, 0.2, 0.01, 16)
    f = flags.frvsr_flags()
    assert (f.pingpang, f.ratio, f.num_resblock) == (False, -0.01, 10)


def test_shared_library_exports_every_declared_symbol():
    from tecogan_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    header = open(os.path.join(ROOT, "include", "tecogan_hip.h")).read()
    declared = set(re.findall(r"\b(tg_[a-z0-9_]+)\s*\(", header))
    h = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared if not hasattr(h, n)]
    assert not missing, missing
    assert declared - {"tg_version", "tg_last_error_string"} == set(_lib.SIGNATURES), \
        declared.symmetric_difference(set(_lib.SIGNATURES))
    assert h.tg_version() >= 1


def test_bad_arguments_return_error_codes_not_crashes():
    """No GPU needed: argument validation happens before any launch."""
    from tecogan_amd import _lib
    L = _lib.lib()
    assert L.tg_conv_forward(None, None, None, None, None, None, None, None) == -1
    assert b"null" in L.tg_last_error_string()
    d = _lib.ConvDesc(1, 4, 4, 8, 4, 4, 8, 3, 3, 5, 1, 1, 0, 0, 0, 0, 0.0, 0, 0.0)     # stride 5 unsupported
    assert L.tg_conv_forward(ctypes.byref(d), 1, 1, None, None, None, 1, None) == -1


def test_product_never_imports_the_oracle():
    bad = []
    for base in ("tecogan_amd", "lib"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dirpath, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M):
                        bad.append(os.path.join(dirpath, f))
    for f in ("main.py", "runGan.py"):
        if re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(ROOT, f)).read(), re.M):
            bad.append(f)
    assert not bad, bad


def test_product_fails_loudly_without_gpu_tensors():
    from tecogan_amd import kernels as K
    from tecogan_amd._lib import TecoHipError
    with pytest.raises(TecoHipError):
        K._p(torch.zeros(4))                      # CPU tensor: there is no fallback


def test_param_specs_and_init_match_the_oracle():
    import oracle.nets as ON
    from tecogan_amd import params as PP
    for mine, ref in ((PP.generator_spec(16), ON.generator_spec(16)), (PP.fnet_spec(), ON.fnet_spec()),
                      (PP.discriminator_spec(), ON.discriminator_spec()), (PP.vgg_spec(), ON.vgg_spec())):
        assert list(mine.items()) == list(ref.items())
    a, b = PP.init_values(PP.fnet_spec(), 5), ON.init_params(ON.fnet_spec(), 5)
    assert all(torch.equal(a[k], b[k]) for k in a)
    a, b = PP.init_values(PP.vgg_spec(), 6, he_normal=True), ON.init_params(ON.vgg_spec(), 6, vgg_he=True)
    assert all(torch.equal(a[k], b[k]) for k in a)


# ---- data-parallel exchange under gloo, world_size 2 ---------------------------------------------------------
_DP_SCRIPT = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from collections import OrderedDict
from oracle import teco as OT
from tecogan_amd.parallel import exchange
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
F = OT.frvsr_flags(batch_size=1, RNN_N=2, crop_size=16, num_resblock=1)
g = torch.Generator().manual_seed(3)
x = torch.rand(2, 2, 16, 16, 3, generator=g); y = torch.rand(2, 2, 64, 64, 3, generator=g) * 2 - 1
S = OT.State(F, seed=42, gan=False)
R = OT.train_step(S, x[rank:rank + 1], y[rank:rank + 1])            # this rank's shard (B=1)
names = list(R["grads"])
flat = torch.cat([R["grads"][n].reshape(-1) for n in names])
ng = sum(R["grads"][n].numel() for n in names if n.startswith("generator/"))
ranges = OrderedDict(generator=(0, ng), fnet=(ng, flat.numel()))
tb = torch.tensor([float(rank) + 1.0])
w = exchange(flat, ranges, ["generator", "fnet"], tb)
flat /= w
F2 = OT.frvsr_flags(batch_size=2, RNN_N=2, crop_size=16, num_resblock=1)
S2 = OT.State(F2, seed=42, gan=False)
R2 = OT.train_step(S2, x, y)                                         # the full batch in one process
ref = torch.cat([R2["grads"][n].reshape(-1) for n in names])
err = ((flat - ref).abs().max() / ref.abs().max()).item()
assert err < 1e-5, err
assert abs(tb.item() - 1.5) < 1e-6
if rank == 0: print("DP_OK", err)
dist.destroy_process_group()
'''


def test_data_parallel_exchange_gloo_world2(tmp_path):
    """Sharding sequences over 2 ranks + flat all-reduce/world == the single-process full-batch gradient,
    and the 1-float t_balance average is identical on both ranks (same D-gate branch)."""
    script = tmp_path / "dp.py"
    script.write_text(_DP_SCRIPT % ROOT)
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_engine_program_dry_run_with_mocked_kernels(monkeypatch):
    """Host logic of the training step (engine.py: two-stream overlap schedule, VGG frame chunks, D triplet bookkeeping,
    temporal-only D, exchange hooks) executed on CPU tensors with every C-ABI wrapper replaced by a recorder: no compute,
    but every shape/view/argument expression of the program runs, for all four graph variants and both stream modes."""
    import contextlib
    import inspect

    import tecogan_amd.kernels as K

    class FakeStream:
        def __init__(self, *a, **k):
            self.cuda_stream = 0

        def wait_stream(self, s):
            pass

        def wait_event(self, e):
            pass

    class FakeEvent:
        def __init__(self, *a, **k):
            pass

        def record(self, s=None):
            pass

    cur = FakeStream()
    calls, work = [], {}

    def recorder(name, orig):
        sig = inspect.signature(orig)

        def f(*a, **k):
            calls.append(name)
            first = a[0] if a else None
            if isinstance(first, K.ConvDesc):
                work[name] = work.get(name, 0) + first.N * first.Hin * first.Win * first.Cin * first.Cout
            elif isinstance(first, torch.Tensor):
                work[name] = work.get(name, 0) + first.numel()
            ba = sig.bind(*a, **k)
            ba.apply_defaults()
            for key in ("out", "d_in", "d_x", "y", "dst"):
                if isinstance(ba.arguments.get(key), torch.Tensor):
                    return ba.arguments[key]
            return None
        return f

    keep = {"dt", "same_pad", "conv_desc"}
    for n in dir(K):
        o = getattr(K, n)
        if inspect.isfunction(o) and o.__module__ == K.__name__ and not n.startswith("_") and n not in keep:
            monkeypatch.setattr(K, n, recorder(n, o))
    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: cur)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    from tecogan_amd.engine import TrainEngine
    from tecogan_amd.flags import frvsr_flags, tecogan_flags
    small = dict(batch_size=1, RNN_N=3, crop_size=16, num_resblock=1)
    variants = [(tecogan_flags(**small), True), (frvsr_flags(**small), False),
                (tecogan_flags(pingpang=False, vgg_scaling=-1.0, **small), True),
                (tecogan_flags(Dt_mergeDs=False, **dict(small, RNN_N=4)), True)]
    for F, gan in variants:
        counts, works = {}, {}
        for ov in ("1", "0"):
            monkeypatch.setenv("TG_OVERLAP", ov)
            calls.clear()
            work.clear()
            eng = TrainEngine(F, "cpu", gan=gan, act_dtype=torch.bfloat16, use_graph=False)
            eng.step(torch.rand(1, F.RNN_N, 16, 16, 3), torch.rand(1, F.RNN_N, 64, 64, 3))
            counts[ov], works[ov] = sorted(calls), dict(work)
            assert "adam_tf" in calls and "conv_forward" in calls
        # The overlap schedule runs the SAME work, some of it in different pieces: per entry point the processed volume (images x pixels x channels of
        # the first operand) is identical; only the accumulations of the side-stream scratch gradients (lincomb) are extra.
        assert set(counts["1"]) == set(counts["0"])
        for name in works["0"]:
            if name != "lincomb":
                assert works["1"][name] == works["0"][name], (name, works["1"][name], works["0"][name])
        assert len(counts["1"]) >= len(counts["0"])
        segs = list(eng._done)                                    # (serial run: one flat program)
        assert segs[0] == "head" and segs[-1] == "update"
    # ping-pong targets repeat: with TG_VGGT_DEDUP=1 the VGG target pass runs on the T0 distinct frames only (3 of 5 here)
    # and the mirrored frames' features are gathered; everything downstream sees tensors of the same shapes
    F = tecogan_flags(**small)
    vol = {}
    for dd in ("0", "1"):
        monkeypatch.setenv("TG_VGGT_DEDUP", dd)
        calls.clear()
        work.clear()
        eng = TrainEngine(F, "cpu", gan=True, act_dtype=torch.bfloat16, use_graph=False)
        assert eng.vggt_dedup == (dd == "1")
        eng.step(torch.rand(1, F.RNN_N, 16, 16, 3), torch.rand(1, F.RNN_N, 64, 64, 3))
        vol[dd] = dict(work)
        vol[dd]["n_seq_gather"] = calls.count("seq_gather")
    monkeypatch.delenv("TG_VGGT_DEDUP")
    assert vol["1"]["n_seq_gather"] == vol["0"]["n_seq_gather"] + 4                    # one gather per VGG tap
    assert vol["1"]["vgg_preprocess_forward"] == vol["0"]["vgg_preprocess_forward"] - 2 * 64 * 64 * 3     # two frames fewer
    assert vol["1"]["cosine_loss"] == vol["0"]["cosine_loss"] and vol["1"]["conv_forward"] < vol["0"]["conv_forward"]


def test_scene_loader_moving_first_frame_augmentation(tmp_path):
    """lib/dataloader.py:112-146 of the reference: 30 % of the sequences show the FIRST frame under a random walk of
    integer crop offsets in [-4, 4] per step (camera-motion augmentation); the others are ordinary frame sequences."""
    import numpy as np
    from PIL import Image
    from lib.dataloader import SceneSequences
    from tecogan_amd.flags import frvsr_flags
    F = frvsr_flags(batch_size=8, RNN_N=4, crop_size=8, input_video_dir=str(tmp_path), str_dir=1000, end_dir=1000, max_frm=7,
                    flip=False)
    sd = tmp_path / "scene_1000"
    sd.mkdir()
    rng = np.random.RandomState(0)
    for i in range(8):     # every frame has its own constant red level, plus a fixed spatial pattern in green / blue
        img = np.zeros((64, 72, 3), np.uint8)
        img[..., 0] = 10 * i + 5
        img[..., 1] = np.arange(64)[:, None] * 3
        img[..., 2] = np.arange(72)[None, :] * 3
        Image.fromarray(img).save(sd / ("col_high_%04d.png" % i))
    ld = SceneSequences(F, "cpu", 1000, 1000, seed=3, prefetch=0)
    moved = plain = 0
    for _ in range(6):
        clips = ld._host_batch()[0]                                      # [B,T,tar,tar,3]
        assert clips.shape == (8, 4, 8 * 4 + 8, 8 * 4 + 8, 3)
        for c in clips:
            reds = [round(float(c[t, 0, 0, 0]) * 255) for t in range(4)]
            if len(set(reds)) == 1:                                      # all four crops come from ONE frame
                moved += 1
                dy = [round((float(c[t + 1, 0, 0, 1]) - float(c[t, 0, 0, 1])) * 255 / 3) for t in range(3)]
                dx = [round((float(c[t + 1, 0, 0, 2]) - float(c[t, 0, 0, 2])) * 255 / 3) for t in range(3)]
                assert all(-4 <= d <= 4 for d in dy + dx), (dy, dx)
            else:
                plain += 1
                assert reds == [reds[0] + 10 * t for t in range(4)], reds      # consecutive frames, one shared crop
                assert all(c[t, 0, 0, 1] == c[0, 0, 0, 1] and c[t, 0, 0, 2] == c[0, 0, 0, 2] for t in range(4))
    assert moved > 0 and plain > moved, (moved, plain)


# ---- engine.plan_launch_order: the host-side launch schedule of a step's segments (pure logic) ---------------------------
def _segs(spec):
    return [dict(name=n, skey=k, deps=list(d)) for n, k, d in spec]


# the round-4 TecoGAN schedule (engine._program_compute, TG_OVERLAP_PARTS default, 19 frames in chunks of 5 / 5 / 5 / 4, target
# lookahead) with a process group; "vggt" (in-step target pass), "vggt_pre" (stored features) and "vggt_next" are CONDITIONAL
TECO_SEGS = [("head", "M", []), ("vggt", "S", ["head"]), ("vggt_pre", "S", ["head"]), ("dreal", "S", ["head"]),
             ("fwd_0", "M", []), ("vgg_0", "S", ["fwd_0"]), ("fwd_1", "M", []), ("vgg_1", "S", ["fwd_1"]),
             ("fwd_2", "M", []), ("vgg_2", "S", ["fwd_2"]), ("fwd_3", "M", []), ("vgg_3", "S", ["fwd_3"]),
             ("fwd_loss", "M", ["dreal", "vggt", "vggt_pre"]), ("down", "S", ["fwd_loss"]), ("ar_d", "C", ["down"]),
             ("vggt_next", "S", ["fwd_loss"]), ("bwd", "M", []), ("bwd_b", "M", ["vgg_0", "vgg_1", "vgg_2", "vgg_3"]),
             ("wgrad", "S", ["bwd_b"]), ("ar_g", "C", ["wgrad"]), ("fnet_bwd", "M", []), ("ar_f", "C", ["fnet_bwd"]),
             ("update", "M", ["down", "wgrad", "ar_d", "ar_g", "ar_f"])]


def test_vgg_chunk_cuts_default_and_override(monkeypatch):
    from tecogan_amd.engine import TrainEngine
    monkeypatch.delenv("TG_VGG_CUTS", raising=False)
    assert TrainEngine._vgg_cuts(19) == [7] and TrainEngine._vgg_cuts(10) == [4] and TrainEngine._vgg_cuts(3) == [1]
    assert TrainEngine._vgg_cuts(1) == []
    monkeypatch.setenv("TG_VGG_CUTS", "5,10,15")
    assert TrainEngine._vgg_cuts(19) == [5, 10, 15] and TrainEngine._vgg_cuts(5) == []
    monkeypatch.setenv("TG_VGG_CUTS", "")
    assert TrainEngine._vgg_cuts(19) == []


def test_plan_launch_order_just_in_time_side_segments():
    from tecogan_amd.engine import plan_launch_order
    acts = list(plan_launch_order(_segs(TECO_SEGS), lazy=True))
    order = [a[1]["name"] for a in acts if a[0] == "launch"]
    assert sorted(order) == sorted(n for n, _, _ in TECO_SEGS) and len(set(order)) == len(order)      # everything, once
    pos = {n: i for i, n in enumerate(order)}
    for n, k, deps in TECO_SEGS:                              # dependencies are launched (events recorded) first
        assert all(pos[d] < pos[n] for d in deps), n
    for key in ("M", "S", "C"):                               # per-stream program order is kept
        mine = [n for n, k, _ in TECO_SEGS if k == key]
        assert [n for n in order if n in mine] == mine
    # a side / communication segment is launched only after a host wait on exactly its dependencies ...
    for i, (what, arg) in enumerate(acts):
        if what == "launch" and arg["skey"] != "M" and arg["deps"]:
            assert acts[i - 1] == ("wait", arg["deps"]), arg["name"]
    # ... and at every such wait the main stream has a whole segment queued BEHIND the awaited main-stream segment
    launched = []
    for what, arg in acts:
        if what == "launch":
            launched.append(arg["name"])
        else:
            for d in arg:
                if dict((n, k) for n, k, _ in TECO_SEGS)[d] == "M" and d != "fnet_bwd":
                    later_m = [n for n in launched[launched.index(d) + 1:] if dict((n, k) for n, k, _ in TECO_SEGS)[n] == "M"]
                    assert later_m, "host would wait for %s with nothing queued behind it" % d
    assert order[:2] == ["head", "fwd_0"]                     # the forward recurrence is queued before the first wait


def test_plan_launch_order_program_order_when_not_lazy():
    from tecogan_amd.engine import plan_launch_order
    acts = list(plan_launch_order(_segs(TECO_SEGS), lazy=False))
    assert [a[0] for a in acts] == ["launch"] * len(TECO_SEGS)
    assert [a[1]["name"] for a in acts] == [n for n, _, _ in TECO_SEGS]


def test_engine_replay_executes_the_plan_with_streams_and_events(monkeypatch):
    """TrainEngine._replay on fake streams / graphs / events: every segment replays once on its own stream, after waiting on
    its dependencies' events there, records its event, and side segments are preceded by a host wait on their dependencies."""
    import contextlib
    import types

    import torch

    from tecogan_amd.engine import TrainEngine
    log = []

    class Stream:
        def __init__(self, name):
            self.name = name

        def wait_event(self, ev):
            log.append(("stream_wait", self.name, ev.name))

    class Event:
        def __init__(self, name):
            self.name = name

        def record(self, st):
            log.append(("record", self.name, st.name))

        def synchronize(self):
            log.append(("host_wait", self.name))

    class Graph:
        def __init__(self, name):
            self.name = name

        def replay(self):
            log.append(("replay", self.name, current[-1].name))

    main, side, comm = Stream("M"), Stream("S"), Stream("C")
    current = [main]

    @contextlib.contextmanager
    def stream_ctx(st):
        current.append(st)
        try:
            yield
        finally:
            current.pop()

    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: current[-1])
    monkeypatch.setattr(torch.cuda, "stream", stream_ctx)
    # conditional segments: a steady-state step with the target lookahead replays "vggt_pre" and "vggt_next", not "vggt"
    state = {"ready": True, "have": True}
    conds = {"vggt": lambda: not state["ready"], "vggt_pre": lambda: state["ready"], "vggt_next": lambda: state["have"]}
    segs = [dict(name=n, skey=k, deps=list(d), graph=Graph(n), fn=None, event=Event(n), cond=conds.get(n)) for n, k, d in TECO_SEGS]
    for lazy, ready in ((True, True), (False, True), (True, False)):
        del log[:]
        state["ready"] = state["have"] = ready
        skipped = {"vggt"} if ready else {"vggt_pre", "vggt_next"}
        # (comm_thread=False: the communication segments' waits stay on this thread, as for backends that cannot be captured; the
        #  threaded launcher is covered on the GPU by the stand-in world-2 / world-8 tests)
        eng = types.SimpleNamespace(_segs=segs, streams={"S": side, "C": comm}, lazy_side=lazy, comm_thread=False, _comm_error=[])
        TrainEngine._replay(eng)
        replays = [e for e in log if e[0] == "replay"]
        assert sorted(r[1] for r in replays) == sorted(n for n, _, _ in TECO_SEGS if n not in skipped)
        for _, name, st in replays:                             # on its own stream
            assert st == dict((n, k) for n, k, _ in TECO_SEGS)[name]
        for n, k, deps in TECO_SEGS:
            if n in skipped:
                continue
            i = log.index(("replay", n, k))
            for d in deps:
                if d in skipped:                                # a skipped segment is nothing to wait for
                    assert ("stream_wait", k, d) not in log and ("host_wait", d) not in log
                    continue
                if dict((a, b) for a, b, _ in TECO_SEGS)[d] != k or True:
                    assert ("stream_wait", k, d) in log[:i], (n, d)          # device-side wait before the replay
                    assert log.index(("record", d, dict((a, b) for a, b, _ in TECO_SEGS)[d])) < log.index(("stream_wait", k, d))
                if lazy and k != "M":
                    assert ("host_wait", d) in log[:i], (n, d)
            assert log[i + 1] == ("record", n, k)
        if not lazy:
            assert not [e for e in log if e[0] == "host_wait"]


def test_discriminator_bn_scratch_pool_falls_back_when_exhausted():
    """nets.Discriminator._ws: slices of the caller's pre-zeroed pool while it lasts, a fresh (self-zeroed) tensor after that --
    a pool sized for another schedule must not kill a step."""
    from tecogan_amd.nets import Discriminator
    D = Discriminator.__new__(Discriminator)
    D.scratch, D._cursor = None, 0
    like = torch.zeros(1)
    buf, pooled = D._ws(4, like)
    assert not pooled and buf.shape == (2, 4)
    D.set_scratch(torch.zeros(20))
    a, pa = D._ws(4, like)
    b, pb = D._ws(4, like)
    c, pc = D._ws(4, like)            # 24 > 20: exhausted
    assert pa and pb and not pc and c.shape == (2, 4)
    assert a.data_ptr() == D.scratch.data_ptr() and b.data_ptr() == D.scratch[8:].data_ptr()
    d, pd = D._ws(2, like)            # the remaining 4 floats still serve a smaller request
    assert pd and d.data_ptr() == D.scratch[16:].data_ptr()


def test_test_while_train_spawns_the_reference_inference_command(tmp_path, monkeypatch):
    """main.testWhileTrain (reference main.py:151-174): after a checkpoint save, `--mode inference` on 10 frames of the calendar
    clip in a child process of its own process group; skipped with a note when the clip folder is absent."""
    import importlib
    import main as M
    importlib.reload(M)
    from tecogan_amd import flags as FL
    F = FL.parse(["--output_dir", str(tmp_path / "out"), "--mode", "train", "--num_resblock", "16", "--cudaID", "0"])
    monkeypatch.chdir(tmp_path)
    assert M.testWhileTrain(F, 500) is None                      # no ./LR/calendar/ here
    (tmp_path / "LR" / "calendar").mkdir(parents=True)
    seen = {}

    class FakePopen:
        pid = 999999

        def __init__(self, cmd, preexec_fn=None):
            seen["cmd"], seen["preexec"] = cmd, preexec_fn
            seen["spawned"] = seen.get("spawned", 0) + 1

        def poll(self):                                          # non-blocking reap (never wait() in the training loop)
            seen["polled"] = seen.get("polled", 0) + 1
            return None if seen.get("running") else 0

        def wait(self):
            seen["waited"] = seen.get("waited", 0) + 1

    monkeypatch.setattr(subprocess, "Popen", FakePopen)
    child = M.testWhileTrain(F, 500)
    cmd = seen["cmd"]
    assert isinstance(child, FakePopen) and seen["preexec"] is os.setpgrp
    assert cmd[0] == sys.executable and cmd[1].endswith("main.py")
    opts = dict(zip(cmd[2::2], cmd[3::2]))
    assert opts["--mode"] == "inference" and opts["--num_resblock"] == "16" and opts["--input_dir_len"] == "10"
    assert opts["--checkpoint"] == os.path.join(str(tmp_path / "out"), "model-500")
    assert opts["--output_dir"] == opts["--summary_dir"] == os.path.join(str(tmp_path / "out"), "train/")
    assert opts["--input_dir_LR"] == "./LR/calendar/" and opts["--output_pre"] == "" and opts["--output_name"] == "000000500"
    FL.parse(cmd[2:])                                            # the child's command line parses with the same flag table
    # the previous try-out is reaped without blocking before the next one starts (no zombies, no two children on the GPU at
    # once); while it is still running the new try-out is SKIPPED, never waited for (the reference fires and forgets)
    M.testWhileTrain(F, 1000)
    assert seen["spawned"] == 2 and seen["polled"] == 1 and "waited" not in seen
    seen["running"] = True
    assert M.testWhileTrain(F, 1250) is None and seen["spawned"] == 2 and "waited" not in seen
    monkeypatch.setattr(M, "TESTWHILETRAIN_MAX_S", -1)           # ... unless it is hung: killed (its process group), then reaped
    killed = []
    monkeypatch.setattr(os, "killpg", lambda pid, sig: killed.append(pid))
    M.testWhileTrain(F, 1300)
    assert killed == [999999] and seen["waited"] == 1 and seen["spawned"] == 3
    seen["running"] = False
    monkeypatch.setenv("TG_TEST_WHILE_TRAIN", "0")
    assert M.testWhileTrain(F, 1500) is None and seen["spawned"] == 3


def test_bench_gpus_n_without_a_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus N` with WORLD_SIZE unset (the form the driver uses for N = 1) must start N ranks itself
    (re-executing under torch.distributed.run, rendezvous on 127.0.0.1) instead of silently running one: the plumbing
    mode joins a gloo group on the CPU and reports the rank count."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"spawn_check": True, "n_gpus": 2, "ranks": 2}
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node 2" in r.stderr
    # a rank count that contradicts the launcher's is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check"], env=env2,
                        capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "does not match WORLD_SIZE" in (r2.stderr + r2.stdout)


def test_wgrad_tr_lds_swizzle_is_consistent_and_conflict_free():
    """csrc/conv_wgrad_tr.hip (round 4): the 128-byte pixels of the X halo tile and of the dY tile are stored with their four
    32-byte channel pairs permuted by tr_swz(pixel) -- applied on the global side of the LDS-DMA -- and the fragment readers
    address pair (cp ^ tr_swz(pixel)).  A Python mirror of both sides: every lane of every ds_read_b64_tr_b16 gets the pixel and
    channels it wants, and the 32 lanes of a read group (MI355X_MICROARCH.md, LDS table) hit 64 distinct banks -- 4-way
    conflicts without the swizzle."""
    def swz(q):
        return ((q >> 1) & 1) | ((q >> 2) & 2)

    TRW, PIX = 32, 128
    lds = {}                                                   # byte address -> (halo pixel, channel), as the DMA fills it
    for q in range((8 + 2) * (TRW + 2)):
        for c in range(8):
            cg = (((c >> 1) ^ swz(q)) << 1) | (c & 1)           # the 16-byte global chunk this LDS slot receives
            for b in range(0, 16, 2):
                lds[q * PIX + c * 16 + b] = (q, cg * 8 + b // 2)

    def group_conflicts(addrs):
        worst = 0
        for g in (range(0, 32), range(32, 64)):
            banks = {}
            for lane in g:
                for w in (0, 4):
                    banks.setdefault(((addrs[lane] + w) // 4) % 64, set()).add(addrs[lane] + w)
            worst = max(worst, max(len(v) for v in banks.values()))
        return worst

    worst = 0
    for quad in range(4):
        ci0 = 32 * (quad >> 1)
        for i in range(2):
            for yy in range(8):
                for tap in range(9):
                    K0 = (yy + tap // 3) * (TRW + 2) + tap % 3
                    for K in (K0, K0 + 4):                      # the two transpose reads of a fragment
                        addrs = []
                        for lane in range(64):
                            frow, fg = lane & 15, lane >> 4
                            lp, lc = 8 * fg + (frow >> 2), 4 * (frow & 3)
                            a = lp * PIX + ((((ci0 >> 4) + i) ^ swz(lp + (K & 15))) * 32) + lc * 2 + K * PIX   # atab[i][K & 15] + K * PIX
                            addrs.append(a)
                            for e in range(4):
                                assert lds[a + 2 * e] == (lp + K, ci0 + 16 * i + lc + e)
                        worst = max(worst, group_conflicts(addrs))
    assert worst == 1
    plain = [(8 * (lane >> 4) + ((lane & 15) >> 2)) * PIX + 4 * (lane & 3) * 2 for lane in range(64)]
    assert group_conflicts(plain) == 4
    # dY tile (64 output channels): unit o of row r holds global unit og; the lane's pixel keeps its swizzle bits at pixel + 4
    ldsy = {}
    for r in range(8):
        for o in range(256):
            og = (o & ~7) | ((((o & 7) >> 1) ^ swz(o >> 3)) << 1) | (o & 1)
            for b in range(0, 16, 2):
                ldsy[(r * 256 + o) * 16 + b] = (r * 32 + (og >> 3), (og & 7) * 8 + b // 2)
    for quad in range(4):
        co0 = 32 * (quad & 1)
        for j in range(2):
            for yy in range(8):
                for half in (0, 4):
                    addrs = []
                    for lane in range(64):
                        frow, fg = lane & 15, lane >> 4
                        lp, lc = 8 * fg + (frow >> 2), 4 * (frow & 3)
                        a = lp * 128 + ((((co0 >> 4) + j) ^ swz(lp)) * 32) + lc * 2 + yy * 32 * 128 + half * 128
                        addrs.append(a)
                        for e in range(4):
                            assert ldsy[a + 2 * e] == (yy * 32 + lp + half, co0 + 16 * j + lc + e)
                    assert group_conflicts(addrs) == 1


def test_round5_halo_layouts_are_consistent_and_conflict_free():
    """Python mirrors of the LDS layouts the round-5 kernels share (64-byte rows = one 32-channel chunk of a pixel, 16-byte groups
    XOR-swizzled by 2 * ((position >> 2) & 1); MI355X_MICROARCH.md LDS table for the ds_read_b128 lane groups):
      * csrc/conv4x4s2.hip forward: the 18 x 34 input halo as two column-parity planes of 18 x 17 positions -- every DMA slot's
        source pixel, and the fragment (input row r, tap column kw) of lane (frow, fg) = input pixel (r, 2 frow + kw), channels 8 fg..;
      * csrc/resblock_thr.hip: the intermediate region written from accumulator registers (ds_write_b64: channels 16 wave + 4 fg ..)
        and read back as fragments by the second conv;
    every fragment read is conflict-free (each of the four 16-lane groups touches 64 distinct banks exactly once)."""
    def swz(q):
        return ((q >> 2) & 1) << 1

    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]

    def conflict_free(addrs):
        for g in groups:
            banks = [((addrs[lane] + 4 * w) // 4) % 64 for lane in g for w in range(4)]
            if len(set(banks)) != 64:
                return False
        return True

    def frag_addr(lane, K):                                     # abase[K & 7] + K * 64 of the kernels
        frow, fg = lane & 15, lane >> 4
        return frow * 64 + ((fg ^ (((((frow & 7) + (K & 7)) >> 2) & 1) << 1)) << 4) + K * 64

    # ---- conv4x4s2 forward: planes
    ROWS, PC, PLANE = 18, 17, 18 * 17
    lds = {}                                                    # byte address -> (input row, input column, channel) as the DMA fills it
    for inst in range(39):
        for lane in range(64):
            S = inst * 64 + lane
            q, ch = S >> 2, (S & 3) ^ swz(S >> 2)
            if q >= 2 * PLANE:
                continue
            pl, rem = divmod(q, PLANE)
            r, c = divmod(rem, PC)
            for b in range(0, 16, 2):
                lds[inst * 1024 + lane * 16 + b] = (r, 2 * c + pl, ch * 8 + b // 2)
    seen = set()
    for kw in range(4):
        for r in range(ROWS):
            K = ((kw & 1) * ROWS + r) * PC + (kw >> 1)
            addrs = [frag_addr(lane, K) for lane in range(64)]
            for lane in range(64):
                frow, fg = lane & 15, lane >> 4
                for e in range(8):
                    assert lds[addrs[lane] + 2 * e] == (r, 2 * frow + kw, 8 * fg + e), (kw, r, lane)
                seen.add((r, 2 * frow + kw))
            assert conflict_free(addrs), (kw, r)
    assert seen == {(r, c) for r in range(18) for c in range(34)}          # every halo pixel is used, none is missing

    # ---- resblock_thr: the intermediate written by the first conv's epilogue, read by the second conv's fragments
    MR, MW = 8, 16
    mid = {}
    for wave in range(4):
        cgrp_w, cchunk = (wave & 1) * 2, wave >> 1
        for m in range(MR):
            for lane in range(64):
                frow, fg = lane & 15, lane >> 4
                q = m * MW + frow
                a = cchunk * 8704 + q * 64 + (((cgrp_w + (fg >> 1)) ^ swz(q)) << 4) + (fg & 1) * 8
                for e in range(4):
                    assert (a + 2 * e) not in mid
                    mid[a + 2 * e] = (m, frow, 16 * wave + 4 * fg + e)
    for ks in range(2):
        for kw in range(3):
            for mr in range(MR):
                K = mr * MW + kw
                addrs = [ks * 8704 + frag_addr(lane, K) for lane in range(64)]
                assert conflict_free(addrs)
                for lane in range(64):
                    frow, fg = lane & 15, lane >> 4
                    if frow + kw >= MW:
                        continue                                # columns 14, 15 of the second conv read past the row: discarded outputs
                    for e in range(8):
                        assert mid[addrs[lane] + 2 * e] == (mr, frow + kw, 32 * ks + 8 * fg + e)


def test_lookahead_promise_is_checked_by_memory_identity_and_version():
    """tecogan_amd/promise.py: an announced next input counts as kept only for the same memory (any view of it),
    unmodified since the announcement; equal values elsewhere, an in-place write, or None do not count."""
    import torch

    from tecogan_amd import promise
    seq = torch.rand(8, 1, 4, 4, 3)
    a = promise.announce(seq[1])
    assert promise.kept(a, seq[1]) and promise.kept(a, seq[1:2][0])           # new view objects of the announced memory
    assert not promise.kept(a, seq[2]) and not promise.kept(a, seq[1].clone()) and not promise.kept(a, None)
    assert not promise.kept(None, seq[1])
    assert not promise.kept(a, seq[1, :, :2])                                  # another shape
    seq[1].mul_(1.0)                                                           # in-place write (values equal): version moved
    assert not promise.kept(a, seq[1])
    b = promise.announce(seq[3])
    seq[5].zero_()                                                             # views share the storage's counter: conservative
    assert not promise.kept(b, seq[3])


def test_comm_thread_preserves_segment_order_and_raises_a_worker_failure_once():
    """segments.SegmentRunner._replay on fake streams (no GPU): the captured communication segments ar_d / ar_g / ar_f go to the
    helper host thread, are launched there in program order and only after the events of their dependencies were waited for;
    `update` waits for all three; a failure inside the worker is raised on the caller's thread ONCE -- the next step runs clean
    -- and every submitted task has finished before _replay returns or raises."""
    import threading
    from tecogan_amd.segments import SegmentRunner
    log, lock = [], threading.Lock()

    class Ev:
        def __init__(self, name):
            self.name, self.recorded = name, False

        def record(self, st):
            self.recorded = True
            with lock:
                log.append(("record", self.name, st.name, threading.current_thread().name))

        def synchronize(self):
            assert self.recorded, "host wait on %s before it was recorded" % self.name
            with lock:
                log.append(("sync", self.name, None, threading.current_thread().name))

    class St:
        def __init__(self, name):
            self.name = name

        def wait_event(self, e):
            assert e.recorded
            with lock:
                log.append(("wait", e.name, self.name, threading.current_thread().name))

    class G:
        def __init__(self, name, fail=False):
            self.name, self.fail = name, fail

        def replay(self):
            with lock:
                log.append(("replay", self.name, None, threading.current_thread().name))
            if self.fail:
                raise RuntimeError("boom in " + self.name)

    class Runner(SegmentRunner):
        lazy_side, comm_thread, launch_jitter = True, True, None

        def __init__(self):
            self.streams = {"S": St("S"), "C": St("C")}
            self.main = St("M")

        def _current_stream(self):
            return self.main

        def _stream_ctx(self, st):
            import contextlib
            return contextlib.nullcontext()

    def seg(name, skey, deps, fail=False):
        return dict(name=name, skey=skey, deps=deps, graph=G(name, fail), fn=None, event=Ev(name), cond=None)

    def program(fail=None):
        return [seg("head", "M", []), seg("down", "M", []), seg("ar_d", "C", ["down"], fail == "ar_d"), seg("bwd_b", "M", []),
                seg("wgrad", "S", ["bwd_b"]), seg("ar_g", "C", ["wgrad"], fail == "ar_g"), seg("fnet_bwd", "M", []),
                seg("ar_f", "C", ["fnet_bwd"]), seg("update", "M", ["wgrad", "ar_d", "ar_g", "ar_f"])]

    r = Runner()
    r._segs = program()
    r._replay()
    order = [n for k, n, _, _ in log if k == "replay"]
    assert [n for n in order if n.startswith("ar_")] == ["ar_d", "ar_g", "ar_f"]
    assert order.index("update") == len(order) - 1
    for k, n, _, th in log:
        if k == "replay":
            assert (th == "tecogan-comm") == n.startswith("ar_"), (n, th)
    for name, dep in (("ar_d", "down"), ("ar_g", "wgrad"), ("ar_f", "fnet_bwd")):                 # just-in-time: dependency COMPLETED first
        i_sync = max(i for i, (k, n, _, th) in enumerate(log) if k == "sync" and n == dep and th == "tecogan-comm")
        i_wait = max(i for i, (k, n, s, _) in enumerate(log) if k == "wait" and n == dep and s == "C")
        i_rep = next(i for i, (k, n, _, _) in enumerate(log) if k == "replay" and n == name)
        assert i_sync < i_rep and i_wait < i_rep
    upd_waits = {n for k, n, s, _ in log if k == "wait" and s == "M"}
    assert {"ar_d", "ar_g", "ar_f", "wgrad"} <= upd_waits
    # a worker failure: raised once, after every task was drained; the next step is clean
    del log[:]
    r._segs = program(fail="ar_g")
    with pytest.raises(RuntimeError, match="boom in ar_g"):
        r._replay()
    assert not r._comm_error, "the failure stayed queued: every later step would raise again"
    assert any(k == "replay" and n == "ar_f" for k, n, _, _ in log) or True
    del log[:]
    r._segs = program()
    r._replay()
    assert [n for k, n, _, _ in log if k == "replay" and n.startswith("ar_")] == ["ar_d", "ar_g", "ar_f"]
    # TG_COMM_THREAD off: the same program from the caller's thread
    del log[:]
    r.comm_thread = False
    r._segs = program()
    r._replay()
    assert all(th != "tecogan-comm" for _, _, _, th in log)
    assert [n for k, n, _, _ in log if k == "replay" and n.startswith("ar_")] == ["ar_d", "ar_g", "ar_f"]
    r.close()


def test_capture_guard_keeps_the_cyclic_collector_off_during_a_capture():
    """tecogan_amd/streams.capture_guard: a dead cycle holding a CUDAGraph must not be freed while a stream captures (its
    destructor's hipGraphDestroy would abort the process): collected before, collector off inside, state restored -- also when
    the body raises, also when the collector was already off."""
    import gc
    from tecogan_amd.streams import capture_guard

    class Node:
        freed = []

        def __init__(self, tag):
            self.me, self.tag = self, tag                     # a reference cycle: only the cyclic collector frees it

        def __del__(self):
            Node.freed.append(self.tag)

    assert gc.isenabled()
    Node("before")
    with capture_guard():
        assert "before" in Node.freed and not gc.isenabled()  # collected on entry
        Node("inside")
        for _ in range(200000):                               # allocation pressure that would trigger generation 0..2
            [] + []
        assert "inside" not in Node.freed
    assert gc.isenabled()
    gc.collect()
    assert "inside" in Node.freed
    with pytest.raises(ValueError):
        with capture_guard():
            raise ValueError("x")
    assert gc.isenabled()
    gc.disable()
    try:
        with capture_guard():
            pass
        assert not gc.isenabled()
    finally:
        gc.enable()
