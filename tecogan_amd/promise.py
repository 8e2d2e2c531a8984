"""Lookahead promises, checked instead of trusted.

`TrainEngine.step(x, y, next_targets=y_next)` and `InferenceEngine.step(frame, next_frame=f_next)` start work on the NEXT
call's input one call early (VGG-19 target features / FNet flow on a side stream).  The stored result may only be used if the
next call really passes that input.  Comparing values would cost a device sync per step; instead the announcement records the
tensor's MEMORY IDENTITY -- data pointer, shape, strides, dtype -- and torch's in-place version counter (shared by all views of
a storage), and keeps a reference to the announced tensor so that its storage cannot be freed and re-used in between.  The same
memory, unmodified, holds the same values: the promise is kept by construction; anything else (another tensor, an in-place
write since the announcement, a fresh copy of equal values) counts as broken and the engine falls back to computing in-step.
"""


def announce(t):
    """Record tensor `t` as the announced next input."""
    return (t, t._version, t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype, t.device)


def kept(ann, t):
    """True iff `t` is the announced memory, not written since the announcement."""
    return (ann is not None and t is not None and t.data_ptr() == ann[2] and tuple(t.shape) == ann[3] and tuple(t.stride()) == ann[4]
            and t.dtype == ann[5] and t.device == ann[6] and t._version == ann[1] and ann[0]._version == ann[1])
