"""Process-wide HIP streams shared by every engine (see shared_stream), and the capture guard."""
import contextlib
import gc

import torch

_STREAMS = {}


def shared_stream(dev, key):
    """One side ("S") / communication ("C") / warm-up ("W") stream per device and PROCESS, shared by every engine created in it.
    HIP maps streams onto a few hardware queues in creation order; an engine created late in a process (its side stream = the
    n-th stream of torch's pool) was observed to run its side segments SERIALISED with the main stream -- 9.8 instead of 8.2 ms per
    step for the third engine of one process (round 5, session L) -- while the first one always overlaps.  Sharing the first
    engine's streams keeps every later engine on the same queues (stream ORDER is all the engines rely on: sharing is safe)."""
    k = (str(dev), key)
    if k not in _STREAMS:
        _STREAMS[k] = torch.cuda.Stream(device=dev)
    return _STREAMS[k]


@contextlib.contextmanager
def capture_guard():
    """No cyclic garbage collection while a stream captures.  A dead reference cycle that holds a torch.cuda.CUDAGraph (an engine of
    an earlier run whose closures or a caught exception's traceback kept it in a cycle) is freed whenever the collector happens to
    run; inside a capture its destructor's hipGraphDestroy / hipGraphExecDestroy fails with "operation not permitted when stream is
    capturing", the exception leaves a C++ destructor and the process aborts (met in the GPU suite: profiles/r06zz_pytest_gpu.log;
    torch.cuda.graph no longer collects on entry by default).  Collect BEFORE, keep the collector off DURING the capture."""
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()
