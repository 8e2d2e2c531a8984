"""Output step after the path: the reference's `save_img` (lib/ops.py:521-523, called once per frame from
the inference loop main.py:262-267) as an asynchronous pipeline.

    frame in [0,1] on the device --tg_frame_to_u8--> uint8 on the device --async D2H on a copy stream--> pinned host
    buffer --worker thread--> PNG / JPEG file (PIL)

The compute stream only pays the uint8 conversion (a 25 MB read + 6 MB write at 1080p); the PCIe copy (6 MB instead of
the reference's 25 MB fp32 fetch) and the image encoding overlap the next frames.  `slots` frames may be in flight.
"""
import os
import queue
import threading

import torch

from . import kernels as K


class FrameWriter:
    def __init__(self, shape, device="cuda", slots=4):
        """shape: (H, W, 3) of the frames that will be submitted."""
        self.dev = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.dev_u8 = [torch.empty(shape, dtype=torch.uint8, device=self.dev) for _ in range(slots)]
        self.host_u8 = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.free = queue.Queue()
        for i in range(slots):
            self.free.put(i)
        self.jobs = queue.Queue()
        self.error = None
        self.thread = threading.Thread(target=self._worker, daemon=True)
        self.thread.start()

    def submit(self, path, frame01):
        """Enqueue `frame01` ([H,W,3] fp32 in [0,1], device; e.g. the inference engine's state) for writing to `path`.
        Returns immediately; the frame buffer may be overwritten by the caller as soon as its stream moves on."""
        if self.error is not None:
            raise self.error
        slot = self.free.get()                              # blocks only when `slots` frames are still in flight
        main = torch.cuda.current_stream()
        K.frame_to_u8(frame01.contiguous(), self.dev_u8[slot])
        ready = torch.cuda.Event()
        ready.record(main)
        self.copy_stream.wait_event(ready)
        with torch.cuda.stream(self.copy_stream):
            self.host_u8[slot].copy_(self.dev_u8[slot], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        self.jobs.put((slot, done, path))

    def _worker(self):
        from PIL import Image
        while True:
            job = self.jobs.get()
            if job is None:
                return
            slot, done, path = job
            try:
                done.synchronize()
                os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
                Image.fromarray(self.host_u8[slot].numpy()).save(path)
            except Exception as e:                          # surfaced on the next submit() / close()
                self.error = e
            finally:
                self.free.put(slot)
                self.jobs.task_done()

    def close(self):
        """Wait for every submitted frame to be on disk."""
        self.jobs.join()
        self.jobs.put(None)
        self.thread.join()
        if self.error is not None:
            raise self.error
