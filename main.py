#!/usr/bin/env python
"""Driver with the reference's command line (reference main.py): `--mode inference` streams a PNG folder
through the hipGraph recurrent step, `--mode train` runs the FRVSR / TecoGAN training program on MI355X.

Kept from the reference: every flag spelling (tecogan_amd/flags.py), the stdout lines
"total time ... frame number ..." (main.py:270) and "progress ... image/sec ..." (main.py:409), the logfile
tee, checkpoints `<output_dir>/model-<step>` written initially, every save_freq steps and on Ctrl+C, each followed by
the reference's inference try-out of the new checkpoint in a child process (testWhileTrain).
Different by design: no TF session/graph; checkpoints are torch files keyed by the TF variable names
(SURVEY.md Appendix B); `--checkpoint random` runs with seeded random weights (no trained model offline);
multi-GPU training = launch with `python -m torch.distributed.run --nproc-per-node N main.py ...`.
"""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tecogan_amd import flags as _flags  # noqa: E402


class Logger(object):
    """stdout tee into <summary_dir>/logfile.txt (reference main.py:126-136)."""

    def __init__(self, path):
        self.terminal, self.log = sys.stdout, open(path, "a")

    def write(self, message):
        self.terminal.write(message)
        self.log.write(message)

    def flush(self):
        self.terminal.flush()
        self.log.flush()


def save_checkpoint(eng, output_dir, step, avg=None):
    """`<output_dir>/model-<step>` (torch file: exact resume) and, beside it, the same state as a TensorFlow tensor
    bundle `model-<step>.index/.data-00000-of-00001` -- the files `saver.save` writes (reference main.py:365,420).
    avg = (running averages of the loss slots, their update count): the reference's EMA shadow variables are graph variables
    and travel with its checkpoints, so the displayed averages continue after a resume instead of restarting from 0."""
    from tecogan_amd.checkpoint import save_bundle
    eng.check_handoffs()                       # (never write weights trained through a trunk launch that lost a workgroup)
    path = os.path.join(output_dir, "model-%d" % step)
    torch.save({"variables": eng.ps.state_dict(), "adam_m": eng.ps.m.cpu(), "adam_v": eng.ps.v.cpu(),
                "sched": eng.sched.cpu(), "global_step": step, "avg_raw": None if avg is None else avg[0].cpu(),
                "n_avg": 0 if avg is None else avg[1]}, path)
    steps = {scope: int(eng.sched[8 + 2 * k].item()) for k, scope in enumerate(eng.opt_scopes)}
    save_bundle(path, eng.ps, step, beta1=eng.F.beta, adam_steps=steps, tb_ema=float(eng.sched[1].item()))
    return path


TESTWHILETRAIN_MAX_S = 600          # a try-out (10 frames of the calendar clip) that takes longer than this is hung


def testWhileTrain(FLAGS, testno=0, lr_dir="./LR/calendar/"):
    """reference main.py:151-174: whenever a checkpoint has been saved, try it in `--mode inference` in a child process --
    the first 10 frames of the calendar clip, written as `<output_dir>/train/<step>_*.png`.  The reference hard-codes the
    clip folder and the interpreter; here the child runs under the same interpreter and is skipped (with a note) when the
    clip folder does not exist.  The child is its own process group so that Ctrl+C in the trainer does not reach it.
    Returns the Popen object (None when skipped).  The previous try-out is reaped WITHOUT blocking (`poll()`): while it is
    still running this try-out is skipped and logged -- the reference fires and forgets (main.py:151-180); waiting here would
    stall rank 0 inside the training loop and, with several ranks, every other rank in its next all-reduce.  A
    child that outlives TESTWHILETRAIN_MAX_S is killed (its process group) so that a hung one cannot block every later
    try-out.  TG_TEST_WHILE_TRAIN=0 disables the try-outs (benchmark runs: the child competes for the GPU)."""
    import signal
    import subprocess
    prev = getattr(testWhileTrain, "_child", None)
    if prev is not None:
        if prev.poll() is None:
            age = time.time() - getattr(testWhileTrain, "_started", time.time())
            if age < TESTWHILETRAIN_MAX_S:
                print('[testWhileTrain] step %d: the previous try-out (pid %d) is still running, this one is skipped' % (testno, prev.pid))
                return None
            print('[testWhileTrain] the previous try-out (pid %d) exceeded %d s: killed' % (prev.pid, TESTWHILETRAIN_MAX_S))
            try:
                os.killpg(prev.pid, signal.SIGKILL)              # the child is its own process group (setpgrp below)
            except OSError:
                pass
            prev.wait()
        testWhileTrain._child = None
    if os.environ.get("TG_TEST_WHILE_TRAIN", "1") == "0":
        return None
    desstr = os.path.join(FLAGS.output_dir, 'train/')
    cmd1 = [sys.executable, os.path.join(ROOT, "main.py"),
            "--output_dir", desstr, "--summary_dir", desstr, "--mode", "inference",
            "--num_resblock", "%d" % FLAGS.num_resblock,
            "--checkpoint", os.path.join(FLAGS.output_dir, 'model-%d' % testno),
            "--cudaID", FLAGS.cudaID,
            "--input_dir_LR", lr_dir, "--output_pre", "", "--output_name", "%09d" % testno, "--input_dir_len", "10"]
    print('[testWhileTrain] step %d:' % testno)
    if not os.path.isdir(lr_dir):
        print('[testWhileTrain] %s not found, inference test skipped' % lr_dir)
        return None
    print(' '.join(cmd1))
    testWhileTrain._child = subprocess.Popen(cmd1, preexec_fn=os.setpgrp)
    testWhileTrain._started = time.time()
    return testWhileTrain._child


def restore_training(eng, FLAGS):
    """reference main.py:312-320,345-352: full resume vs weights-only ("pre-trained") restore."""
    from tecogan_amd.checkpoint import load_variables
    saved, ck = load_variables(FLAGS.checkpoint)                      # torch file or TensorFlow bundle prefix
    if not FLAGS.pre_trained_model:
        print('Loading everything from the checkpoint to continue the training...')
        eng.ps.load({k: v for k, v in saved.items() if k in eng.ps.entries})
        if isinstance(ck.get("adam_m"), dict):                        # bundle: per-variable Adam slots
            for name in eng.ps.entries:
                if name in ck["adam_m"] and name in ck["adam_v"]:
                    eng.ps.view(name, eng.ps.m).copy_(ck["adam_m"][name])
                    eng.ps.view(name, eng.ps.v).copy_(ck["adam_v"][name])
            if "global_step" in ck:
                eng.sched[0] = float(ck["global_step"])
            # Adam bias-correction counters (from <scope>/beta1_power) and the t_balance EMA: without them the first
            # resumed steps would run at lr*sqrt(1-b2)/(1-b1) ~ 0.32 lr against converged moments and the D-gate
            # would restart from tb = 0 (always open)
            for k, scope in enumerate(eng.opt_scopes):
                t = ck.get("adam_steps", {}).get(scope, ck.get("global_step", 0))
                eng.sched[8 + 2 * k] = float(t)
            if "tb_ema" in ck:
                eng.sched[1] = float(ck["tb_ema"])
        else:
            eng.ps.m.copy_(ck["adam_m"])
            eng.ps.v.copy_(ck["adam_v"])
            eng.sched.copy_(ck["sched"])
        eng.host_step = int(eng.sched[0].item())
        if ck.get("avg_raw") is not None:
            return ck["avg_raw"], int(ck.get("n_avg", 0))
        return None
    print('Loading weights from the pre-trained model to start a new training...')
    vals, zero = {}, 0
    for name, e in eng.ps.entries.items():
        if name in saved:
            if tuple(saved[name].shape) != e["shape"]:
                raise ValueError('Shape mismatch for var %s' % name)
            vals[name] = saved[name]
        elif e["scope"] in ("generator", "fnet"):        # rest_zero=True for G/fnet (main.py:314)
            vals[name] = torch.zeros(e["shape"])
            zero += 1
    eng.ps.load(vals)
    print('Prepare to load %d weights from the pre-trained model (%d zero-filled)' % (len(vals), zero))


def run_inference(FLAGS):
    from lib.dataloader import inference_data_loader
    from lib.ops import save_img
    from tecogan_amd.infer import InferenceEngine
    if FLAGS.checkpoint is None:
        raise ValueError('The checkpoint file is needed to performing the test.')
    data = inference_data_loader(FLAGS)
    h, w = data.inputs[0].shape[:2]
    print("input shape:", [1, h, w, 3])
    print("output shape:", [1, h * 4, w * 4, 3])
    tdt = torch.bfloat16 if FLAGS.act_dtype == "bf16" else torch.float32
    eng = InferenceEngine(FLAGS.num_resblock, h, w, "cuda", tdt, seed=FLAGS.rand_seed + 41)
    print('Finish building the network')
    if FLAGS.checkpoint != "random":
        from tecogan_amd.checkpoint import load_variables
        print('Loading weights from ckpt model')
        saved, _ = load_variables(FLAGS.checkpoint)                   # torch file or TensorFlow bundle prefix (ValueError if absent)
        eng.load({k: v for k, v in saved.items() if k in eng.ps.entries})
    image_dir = FLAGS.output_dir if FLAGS.output_pre == "" else os.path.join(FLAGS.output_dir, FLAGS.output_pre)
    os.makedirs(image_dir, exist_ok=True)
    max_iter = len(data.inputs)
    from tecogan_amd.output import FrameWriter
    writer = FrameWriter((4 * h, 4 * w, 3))                  # uint8 conversion on the GPU, async D2H, background encoding
    print('Frame evaluation starts!!')
    try:
        _inference_loop(FLAGS, data, eng, writer, image_dir, max_iter)
    finally:
        writer.close()                                        # in-flight frames are written even if the loop raises


def _inference_loop(FLAGS, data, eng, writer, image_dir, max_iter):
    srtime = 0.0

    dev = {}

    def upload(i):
        if i < max_iter and i not in dev:
            dev[i] = torch.from_numpy(data.inputs[i].copy()).float()[None].cuda()
        return dev.get(i)

    for i in range(max_iter):
        # the clip is known up front: the frames after i are announced, the engine runs FNet on a window of them as one batch
        frame = upload(i)
        ahead = [upload(j) for j in range(i + 1, min(i + 1 + eng.window, max_iter))]
        dev.pop(i - 1, None)
        torch.cuda.synchronize()
        t0 = time.time()
        out = eng.step(frame, upcoming=ahead)
        torch.cuda.synchronize()
        srtime += time.time() - t0
        if i >= 5:
            name = os.path.splitext(os.path.basename(str(data.paths_LR[i])))[0]
            filename = FLAGS.output_name + '_' + name
            print('saving image %s' % filename)
            writer.submit(os.path.join(image_dir, "%s.%s" % (filename, FLAGS.output_ext)), out[0])
        else:   # first 5 frames: mirrored warm-up, timed but not saved (reference main.py:268-269)
            print("Warming up %d" % (5 - i))
    eng.check_handoffs()                       # (a trunk launch that lost a workgroup would have written garbage frames: fail loudly)
    print("total time " + str(srtime) + ", frame number " + str(max_iter))


def run_training(FLAGS):
    # The process drives a GPU: torch's intra-op OpenMP team (one thread per host CPU, 256 on the GPU box) is never useful here
    # and, woken by any stray CPU tensor op, delays the thread that launches the step's graph segments (step() 12.7 -> 31 ms,
    # tools/mb_mainloop.py).  The loader threads do their own parallel PNG decoding.
    torch.set_num_threads(min(4, torch.get_num_threads()))
    from lib.dataloader import frvsr_gpu_data_loader
    from tecogan_amd.engine import TrainEngine
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev, pg = torch.device("cuda", local), None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
        pg = dist.group.WORLD
    gan = FLAGS.ratio > 0                                       # reference main.py:283-286
    tdt = torch.bfloat16 if FLAGS.act_dtype == "bf16" else torch.float32
    rdata = frvsr_gpu_data_loader(FLAGS, device=dev, synthetic=FLAGS.synthetic, rank=rank)
    eng = TrainEngine(FLAGS, dev, gan=gan, act_dtype=tdt, seed=FLAGS.rand_seed + 41, process_group=pg)
    print('Finish building the network.')
    restored_avg = None
    if FLAGS.checkpoint is not None:
        restored_avg = restore_training(eng, FLAGS)
    if FLAGS.vgg_scaling > 0.0 and FLAGS.vgg_ckpt and os.path.exists(FLAGS.vgg_ckpt):
        from tecogan_amd.checkpoint import load_variables
        vgg_vars, _ = load_variables(FLAGS.vgg_ckpt)
        eng.vps.load({k: v for k, v in vgg_vars.items() if k in eng.vps.entries})
        print('VGG19 restored successfully!!')
    if rank == 0:
        print('Save initial checkpoint, before any training')
        save_checkpoint(eng, FLAGS.output_dir, eng.global_step())
    frame_len = (FLAGS.RNN_N * 2 - 1) if FLAGS.pingpang else FLAGS.RNN_N
    max_iter = FLAGS.max_iter
    if max_iter is None:
        if FLAGS.max_epoch is None:
            raise ValueError('one of max_epoch or max_iter should be provided')
        max_iter = FLAGS.max_epoch * rdata.steps_per_epoch
    step, run_step, start = 0, eng.global_step(), time.time()
    # running averages of the loss slots as the reference prints them (update_list_avg, lib/Teco.py: EMA 0.99 updated every
    # step, zero-initialised): one tiny device-side kernel per step, read only on display steps
    from tecogan_amd import kernels as K
    avg_raw, n_avg = torch.zeros_like(eng.loss), 0
    if restored_avg is not None and restored_avg[0].numel() == avg_raw.numel():
        avg_raw.copy_(restored_avg[0])
        n_avg = restored_avg[1]
    x, y = rdata.s_inputs, rdata.s_targets
    try:
        for step in range(max_iter):
            run_step = eng.global_step() + 1 if step == 0 else run_step + 1
            # one batch of lookahead: the NEXT batch's targets go through VGG-19 during this step's backward phase
            # (TrainEngine target lookahead); the loader's prefetch thread is ahead of the GPU, so this does not wait
            nx, ny = rdata.loader.next_batch()
            eng.step(x, y, next_targets=ny if eng.lookahead else None)
            K.lincomb(avg_raw, eng.loss, avg_raw, 0.99, 0.01)
            n_avg += 1
            x, y = nx, ny
            if step == 0 and rank == 0:
                print('Optimization starts!!!(Ctrl+C to stop, will try saving the last model...)')
            if rank == 0 and (run_step % FLAGS.display_freq) == 0:
                L = eng.losses(avg_raw, 1.0 - 0.99 ** n_avg)
                rate = world * (step + 1) * FLAGS.batch_size / (time.time() - start)
                remaining = (max_iter - step) * world * FLAGS.batch_size / rate
                print("progress  epoch %d  step %d  image/sec %0.1fx%02d  remaining %dh%dm" %
                      (math.ceil(run_step / rdata.steps_per_epoch), (run_step - 1) % rdata.steps_per_epoch + 1, rate,
                       frame_len, remaining // 3600, (remaining % 3600) // 60))
                print("global_step", run_step)
                print("learning_rate", float(eng.hyper[-1, 5].item()))
                for name, val in L.items():
                    print(name, val)
            if rank == 0 and (run_step % FLAGS.summary_freq) == 0:
                # reference main.py:391-402: the raw loss scalars on a VALIDATION batch (the TensorBoard summaries themselves
                # are out of scope); an eager pass of the step's program without the update segment AND without any exchange
                # segment (TrainEngine._exchange_seg), so rank 0 alone may run it: no collective is issued
                print('Run and Recording summary!!')
                vx, vy = rdata.val_loader.next_batch()
                V = eng.eval_losses(vx, vy)
                print('-----------Validation data scalars-----------')
                for name, val in V.items():
                    if name not in ("t_balance", "t_balance_now"):
                        print('val_' + name, val)
            if rank == 0 and (run_step % FLAGS.save_freq) == 0:
                print('Save the checkpoint')
                save_checkpoint(eng, FLAGS.output_dir, run_step, (avg_raw, n_avg))
                testWhileTrain(FLAGS, run_step)
    except KeyboardInterrupt:
        if step > 1 and rank == 0:
            print('main.py: KeyboardInterrupt->saving the checkpoint')
            save_checkpoint(eng, FLAGS.output_dir, run_step, (avg_raw, n_avg))
            child = testWhileTrain(FLAGS, run_step)
            if child is not None:
                child.communicate()
        print('main.py: quit')
        sys.exit(0)
    torch.cuda.synchronize()
    if rank == 0:
        save_checkpoint(eng, FLAGS.output_dir, run_step, (avg_raw, n_avg))
    print('Optimization done!!!!!!!!!!!!')


def main(argv=None):
    FLAGS = _flags.parse(argv)
    os.environ.setdefault("HIP_VISIBLE_DEVICES", FLAGS.cudaID) if "LOCAL_RANK" not in os.environ else None
    torch.manual_seed(FLAGS.rand_seed)
    if FLAGS.output_dir is None:
        raise ValueError('The output directory is needed')
    os.makedirs(FLAGS.output_dir, exist_ok=True)
    if FLAGS.summary_dir is None:
        FLAGS.summary_dir = os.path.join(FLAGS.output_dir, "log/")
    os.makedirs(FLAGS.summary_dir, exist_ok=True)
    sys.stdout = Logger(os.path.join(FLAGS.summary_dir, "logfile.txt"))
    if FLAGS.mode == 'inference':
        run_inference(FLAGS)
    elif FLAGS.mode == 'train':
        run_training(FLAGS)
    else:
        raise ValueError('mode must be train or inference')


if __name__ == "__main__":
    main()
