"""Drop-in for the loaders of the reference's `lib/dataloader.py` that the hot path consumes.

`inference_data_loader` keeps the reference semantics (sorted PNGs, optional Gaussian x4 down-sampling of an
HR folder, 5 mirrored warm-up frames).  The TF queue-runner training loaders are replaced by a torch loader
over the same directory layout (`<input_video_dir>/<pre>_<dir>/col_high_%04d.png`) with the same shared random
crop / flip and the GPU Gaussian down-sampling + 4-pixel border crop, or by seeded synthetic sequences.
"""
import collections
import os

import numpy as np
import torch

from lib.ops import *  # noqa: F401,F403
from lib import ops as _ops


def _read_png(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)


def inference_data_loader(FLAGS):
    """reference lib/dataloader.py:11-50."""
    filedir, downSP = FLAGS.input_dir_LR, False
    if (FLAGS.input_dir_LR is None) or (not os.path.exists(FLAGS.input_dir_LR)):
        if (FLAGS.input_dir_HR is None) or (not os.path.exists(FLAGS.input_dir_HR)):
            raise ValueError('Input directory not found')
        filedir, downSP = FLAGS.input_dir_HR, True
    names = sorted(f for f in os.listdir(filedir) if f.endswith(".png"))
    names.sort(key=lambda f: int(''.join(ch for ch in f if ch.isdigit()) or -1))
    if FLAGS.input_dir_len > 0:
        names = names[:FLAGS.input_dir_len]
    paths = [os.path.join(filedir, n) for n in names]

    def load(p):
        im = _read_png(p)
        if downSP:      # cv.GaussianBlur(sigma 1.5) + [::4] of the reference, done by the 9x9 HIP Gaussian conv
            t = torch.from_numpy(im / 255.0)[None].cuda()
            t = torch.nn.functional.pad(t.permute(0, 3, 1, 2), (4, 4, 4, 4), mode="reflect").permute(0, 2, 3, 1)
            return _ops.tf_data_gaussDownby4(t.contiguous(), 1.5)[0].cpu().numpy()
        return im / 255.0

    images = [load(p) for p in paths]
    paths = paths[5:0:-1] + paths            # hard-coded symmetric frame padding (reference :42-44)
    images = images[5:0:-1] + images
    Data = collections.namedtuple('Data', 'paths_LR, inputs')
    return Data(paths_LR=paths, inputs=images)


class _Prefetched:
    """Background production of host batches into a fixed RING of pinned buffers (allocated before the thread starts: no
    hipHostMalloc while the engine captures its hipGraphs), `prefetch` batches ahead.  Subclasses provide
    `_host_batch() -> tuple of numpy arrays`.  A failure in the thread is handed to the consumer and re-raised by `_next_host`:
    an exception there must not leave the training loop blocked on an empty queue.  The thread touches the buffers through
    NUMPY views only: a torch CPU op here (`Tensor.copy_`) wakes torch's intra-op OpenMP team -- 256 threads on the GPU host --
    and that alone stretched the launching thread's step() from 12.7 to 31 ms (tools/mb_mainloop.py, profiles/r03g_mainloop.txt)."""
    _prefetch, _thread, _q = 0, None, None

    def _worker(self):
        slot = 0
        try:
            while True:
                ev = self._ring_events[slot]
                if ev is not None:
                    ev.synchronize()                    # the H2D copies that last read this slot have finished
                bufs = self._ring[slot]
                for v, a in zip(self._ring_np[slot], self._host_batch()):
                    np.copyto(v, a)
                self._q.put((slot, bufs))
                slot = (slot + 1) % len(self._ring)
        except BaseException as e:                      # noqa: BLE001 -- re-raised in _next_host()
            self._q.put(("error", e))

    def _start_worker(self, shapes):
        import queue
        import threading
        nslot = self._prefetch + 2                      # queue depth + one being filled + one being copied to the device
        pin = torch.cuda.is_available()
        self._ring = [[torch.empty(*sh, pin_memory=pin) for sh in shapes] for _ in range(nslot)]
        self._ring_np = [[b.numpy() for b in bufs] for bufs in self._ring]
        self._ring_events = [None] * nslot
        self._q = queue.Queue(maxsize=self._prefetch)
        self._thread = threading.Thread(target=self._worker, daemon=True)
        self._thread.start()

    def _next_host(self):
        import queue
        while True:
            try:
                item = self._q.get(timeout=5.0)
                break
            except queue.Empty:
                if not self._thread.is_alive():
                    raise RuntimeError('the loader thread died without reporting an error')
        if item[0] == "error":
            raise RuntimeError('the loader thread failed: %r' % (item[1],)) from item[1]
        return item

    def _to_device(self, slot, bufs):
        out = [b.to(self.dev, non_blocking=True) for b in bufs]
        if out and out[0].is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._ring_events[slot] = ev                # the ring slot is free once these copies are done
        return out


class SyntheticSequences:
    """Seeded synthetic training sequences of the reference's shapes (no dataset offline).  On a GPU the batch is generated
    ON THE DEVICE (a B=4 batch is 2 M random floats: ~10 ms of one host core, most of a TecoGAN step, against microseconds
    of device time); on the CPU (tests) by the host generator."""

    def __init__(self, FLAGS, device, seed=1234):
        self.F, self.dev = FLAGS, torch.device(device)
        self.g = torch.Generator(device=self.dev if self.dev.type == "cuda" else "cpu").manual_seed(seed)
        self.image_count, self.steps_per_epoch = 10 ** 9, 10 ** 9 // FLAGS.batch_size

    def next_batch(self):
        F = self.F
        gen_dev = self.dev if self.dev.type == "cuda" else "cpu"
        x = torch.rand(F.batch_size, F.RNN_N, F.crop_size, F.crop_size, 3, generator=self.g, device=gen_dev)
        y = torch.rand(F.batch_size, F.RNN_N, 4 * F.crop_size, 4 * F.crop_size, 3, generator=self.g, device=gen_dev) * 2 - 1
        return x.to(self.dev, non_blocking=True), y.to(self.dev, non_blocking=True)


class SceneSequences(_Prefetched):
    """Directory loader (reference lib/dataloader.py:53-167,276-348): shared random crop with a Gaussian margin, random
    flip, the moving-first-frame augmentation, GPU down-sampling + target crop + preprocess in one HIP launch.  PNG
    decoding and augmentation run in a background thread that keeps `prefetch` batches ahead (the reference uses
    FLAGS.queue_thread TF queue runners), so a training step of a few milliseconds is not input-bound."""

    def __init__(self, FLAGS, device, first_dir, last_dir, seed=1, prefetch=2, cache_share=1.0):
        if FLAGS.input_video_dir == '':
            raise ValueError('Video input directory input_video_dir is not provided')
        if not os.path.exists(FLAGS.input_video_dir):
            raise ValueError('Video input directory not found')
        if not FLAGS.random_crop:
            raise Exception('Not implemented')          # as the reference (lib/dataloader.py:107): crops are required
        self.F, self.dev, self.rng = FLAGS, device, np.random.RandomState(seed)
        self.scenes = []
        for d in range(first_dir, last_dir + 1):
            sd = os.path.join(FLAGS.input_video_dir, '%s_%04d' % (FLAGS.input_video_pre, d))
            if os.path.exists(os.path.join(sd, 'col_high_%04d.png' % FLAGS.max_frm)):
                self.scenes.append(sd)
        if not self.scenes:
            raise ValueError('No scene with %d frames under %s' % (FLAGS.max_frm + 1, FLAGS.input_video_dir))
        self.image_count = len(self.scenes) * (FLAGS.max_frm - FLAGS.RNN_N + 1)
        self.steps_per_epoch = self.image_count // FLAGS.batch_size
        self._q, self._thread, self._prefetch = None, None, prefetch
        # PNG decoding is the loader's cost (B x RNN_N files per batch; a TecoGAN step is ~12 ms): decode in a thread pool
        # (PIL releases the GIL while it inflates), FLAGS.queue_thread workers as the reference's queue runners
        # (lib/dataloader.py:163-165), TG_LOADER_THREADS overrides
        nthr = int(os.environ.get("TG_LOADER_THREADS", str(max(1, int(getattr(FLAGS, "queue_thread", 6))))))
        self._pool, self._sizes = None, {}
        # decoded-frame cache (uint8): a 352x288 PNG costs ~20 ms of one core to inflate and a TecoGAN step consumes 40 of them
        # every ~12 ms, i.e. ~60 cores of pure decoding; scenes are revisited every epoch, so decoded frames are kept up to
        # TG_LOADER_CACHE_GB (0 switches the cache off).  Default: 1/4 of the host RAM divided by the ranks of this node
        # (every rank builds its own loaders: 8 ranks x RAM/4 would be twice the machine), of which a validation loader
        # (cache_share < 1) takes its share -- the budgets of all loaders of a node add up to RAM/4.  First-epoch batches are
        # decode-bound.
        try:
            ram = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
        except (ValueError, OSError):
            ram = 16 << 30
        import threading
        self._cache, self._cache_bytes, self._cache_lock = {}, 0, threading.Lock()
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
        self._cache_cap = int(float(os.environ.get("TG_LOADER_CACHE_GB", str(ram / 4 / ranks / (1 << 30)))) * (1 << 30)
                              * cache_share)
        self.cache_hits = self.cache_misses = 0
        if nthr > 1:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=nthr)

    def _decode_many(self, paths):
        """uint8 [H,W,3] arrays (the crops are converted to float afterwards: 1/30 of the pixels at 720p)."""
        def dec(p):
            a = self._cache.get(p)
            if a is not None:
                self.cache_hits += 1
                return a
            from PIL import Image
            with Image.open(p) as im:
                a = np.asarray(im.convert("RGB"))
            self.cache_misses += 1
            with self._cache_lock:                                  # the byte count is shared by the pool's threads
                if p not in self._cache and self._cache_bytes + a.nbytes <= self._cache_cap:
                    self._cache[p] = a
                    self._cache_bytes += a.nbytes
            return a
        if self._pool is None or len(paths) < 2 or all(p in self._cache for p in paths):
            return [dec(p) for p in paths]              # (all cached: no pool round trip, the fewer Python-level thread switches the better)
        return list(self._pool.map(dec, paths))

    def _frame_size(self, sd):
        """(H, W) of a scene's frames from the PNG header (no decode), cached per scene."""
        if sd not in self._sizes:
            from PIL import Image
            with Image.open(os.path.join(sd, 'col_high_%04d.png' % 0)) as im:
                self._sizes[sd] = (im.size[1], im.size[0])
        return self._sizes[sd]

    def _host_batch(self):
        """One batch of HR crops [B,T,tar,tar,3] in [0,1] on the host: all random decisions first (one RNG stream, fixed draw
        order), then every PNG the batch needs decoded by the thread pool in one go, then the crops."""
        F = self.F
        border = int(1.5 * 3.0)
        tar = F.crop_size * 4 + 2 * border
        plans, files = [], []
        for _ in range(F.batch_size):
            sd = self.scenes[self.rng.randint(len(self.scenes))]
            t0 = self.rng.randint(F.max_frm - F.RNN_N + 2)
            moving = bool(F.movingFirstFrame) and self.rng.rand() >= 0.7        # lib/dataloader.py:146 (30 % of the sequences)
            H, W = self._frame_size(sd)
            oy, ox = self.rng.randint(H - tar + 1), self.rng.randint(W - tar + 1)
            flip = bool(F.flip) and self.rng.rand() < 0.5
            lt = fy = fx = None
            if moving:
                # camera-motion augmentation (lib/dataloader.py:112-125,138-146): every frame is the FIRST frame, cropped at
                # a random walk of integer offsets in [-4, 4] per step; the walk is shifted so that all crops stay inside
                off = np.floor(self.rng.uniform(-3.5, 4.5, size=(F.RNN_N, 2))).astype(np.int64)
                pos = np.concatenate((np.zeros((1, 2), np.int64), np.cumsum(off, 0)[:-1]))       # exclusive cumsum, (x, y)
                mn = pos.min(0)
                rng_xy = pos.max(0) - mn
                lt = pos - mn
                fy = int(np.clip(oy, 0, H - tar - rng_xy[1]))
                fx = int(np.clip(ox, 0, W - tar - rng_xy[0]))
            n = 1 if moving else F.RNN_N
            plans.append((len(files), n, moving, flip, oy, ox, lt, fy, fx))
            files += [os.path.join(sd, 'col_high_%04d.png' % (t0 + i)) for i in range(n)]
        imgs = self._decode_many(files)
        seqs = []
        for i0, n, moving, flip, oy, ox, lt, fy, fx in plans:
            frames = imgs[i0:i0 + n]
            if flip:
                frames = [f[:, ::-1] for f in frames]
            if moving:
                src = frames[0]
                clip = np.stack([src[fy + lt[i, 1]:fy + lt[i, 1] + tar, fx + lt[i, 0]:fx + lt[i, 0] + tar]
                                 for i in range(F.RNN_N)])
            else:
                clip = np.stack([f[oy:oy + tar, ox:ox + tar] for f in frames])
            seqs.append(np.ascontiguousarray(clip).astype(np.float32) / np.float32(255.0))
        return (np.stack(seqs),)

    def _validate_geometry(self):
        """Every crop (incl. the moving-first-frame walk of up to 4 px per step) must fit the frames: checked once, up front,
        instead of failing inside the prefetch thread."""
        F = self.F
        tar = F.crop_size * 4 + 2 * int(1.5 * 3.0)
        first = _read_png(os.path.join(self.scenes[0], 'col_high_%04d.png' % 0))
        H, W = first.shape[:2]
        need = tar + (4 * (F.RNN_N - 1) if F.movingFirstFrame else 0)
        if H < need or W < need:
            raise ValueError('frames of %dx%d are too small for crop_size %d (need %d pixels%s)' % (
                W, H, F.crop_size, need, ' incl. the movingFirstFrame walk' if F.movingFirstFrame else ''))

    def next_batch(self):
        F = self.F
        if self._prefetch > 0:
            if self._thread is None:
                self._validate_geometry()
                tar = F.crop_size * 4 + 2 * int(1.5 * 3.0)
                self._start_worker([(F.batch_size, F.RNN_N, tar, tar, 3)])
            hr, = self._to_device(*self._next_host())                           # [B,T,tar,tar,3] in [0,1]
        else:
            hr = torch.from_numpy(self._host_batch()[0]).to(self.dev, non_blocking=True)
        B, T, tar = hr.shape[0], hr.shape[1], hr.shape[2]
        lr, tgt = _ops.gauss_down_crop_preprocess(hr.reshape(B * T, tar, tar, 3), 1.5)
        cs = F.crop_size
        return _ops.preprocessLR(lr).reshape(B, T, cs, cs, 3), tgt.reshape(B, T, 4 * cs, 4 * cs, 3)


def frvsr_gpu_data_loader(FLAGS, useValData_ph=None, device="cuda", synthetic=False, rank=0):
    """reference lib/dataloader.py:276-348.  Returns Data(paths_HR, s_inputs, s_targets, image_count,
    steps_per_epoch) where s_inputs / s_targets hold the first batch and `Data.loader.next_batch()` /
    `Data.val_loader.next_batch()` stream the following ones."""
    # data-parallel training: every rank draws its OWN stream (seed offset by the rank) -- with identical batches the
    # averaged gradient would equal a single-GPU step and the extra GPUs would be redundant compute
    if synthetic or FLAGS.input_video_dir == '':
        train = val = SyntheticSequences(FLAGS, device, seed=1234 + rank)
    else:
        # the validation loader is read once every summary_freq steps: 1/16 of the node's cache budget, the rest for training
        train = SceneSequences(FLAGS, device, FLAGS.str_dir, FLAGS.end_dir, FLAGS.rand_seed + 1000 * rank, cache_share=15.0 / 16)
        try:
            val = SceneSequences(FLAGS, device, FLAGS.end_dir + 1, FLAGS.end_dir_val, FLAGS.rand_seed + 1 + 1000 * rank,
                                 cache_share=1.0 / 16)
        except ValueError:
            val = train
    x, y = train.next_batch()
    Data = collections.namedtuple('Data', 'paths_HR, s_inputs, s_targets, image_count, steps_per_epoch, loader, val_loader')
    return Data(None, x, y, train.image_count, train.steps_per_epoch, train, val)
