"""A minimal eager stand-in for the TensorFlow-1.x surface that the reference's `lib/ops.py`, `lib/frvsr.py`
and `lib/Teco.py` touch, backed by torch-CPU and by the [TF1] op semantics of `oracle/ops.py`.

TEST INFRASTRUCTURE ONLY.  Purpose: import the reference's OWN Python files, unmodified, from /root/reference and run
`fnet`, `generator_F`, `discriminator_F`, `VGG19_slim`, `TecoGAN`, `FRVSR` here, so that the graph WIRING of the
reference (layer order, variable names, frame indices, D-input packing, loss formulas, optimiser plumbing, in-tree numerics
such as upscale_four / bicubic_four) pins the oracle through golden vectors (`oracle/make_golden.py` -> `tests/golden/`).
It does NOT pin the numerics that live inside TensorFlow itself (conv padding, conv_transpose alignment, legacy resize,
dense_image_warp, batch_norm, Adam, EMA): those are provided here by `oracle/ops.py` and stay "parity unpinned".

Execution model: eager, except that side effects (variable assigns, optimiser applies, moving-average updates) are
deferred thunks run by `flush()` -- which realises the semantics fixed in DESIGN.md: every gradient is taken from the
pre-update weights and the D-gate reads the OLD balance average.
"""
import contextlib
import sys
import types

import numpy as np
import torch

from . import ops as O

_STATE = types.SimpleNamespace()


def reset(preset=None, trainable_prefixes=("generator", "fnet", "tdiscriminator")):
    _STATE.vars = {}                 # name -> Variable (creation order preserved)
    _STATE.preset = dict(preset or {})
    _STATE.scope = []                # [(name, reuse)]
    _STATE.collections = {}
    _STATE.pending = []              # deferred side effects
    _STATE.arg_scope = []            # [(funcs, kwargs)]
    _STATE.trainable_prefixes = trainable_prefixes
    _STATE.unique = {}
    _STATE.captured_grads = {}


reset()


def flush():
    for fn in _STATE.pending:
        fn()
    _STATE.pending = []


# --------------------------------------------------------------------------------------------------
class T:
    """tf.Tensor stand-in: wraps a torch tensor, supports negative-step slicing and TF-style shape calls."""
    __array_priority__ = 1000

    def __init__(self, v):
        self.v = v if isinstance(v, torch.Tensor) else torch.as_tensor(v, dtype=torch.float32)

    # shape protocol
    def get_shape(self):
        return _Shape(self.v.shape)

    def set_shape(self, shape):
        want = [int(s) for s in shape]
        assert list(self.v.shape) == want, "set_shape mismatch %s vs %s" % (list(self.v.shape), want)

    @property
    def shape(self):
        return _Shape(self.v.shape)

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        out, flips, norm = self.v, [], []
        dim = 0
        for it in idx:
            if isinstance(it, slice) and it.step is not None and it.step < 0:
                n = out.shape[dim]
                sel = list(range(n))[it]
                norm.append(torch.tensor(sel, dtype=torch.long))
            else:
                norm.append(it)
            if it is not None:
                dim += 1
        for d, it in enumerate(norm):
            if isinstance(it, torch.Tensor):
                out = out.index_select(d, it)
        norm2 = tuple(slice(None) if isinstance(it, torch.Tensor) else it for it in norm)
        return T(out[norm2])

    def _b(self, o, f):
        return T(f(self.v, o.v if isinstance(o, T) else o))

    def __add__(self, o): return self._b(o, lambda a, b: a + b)
    def __radd__(self, o): return self._b(o, lambda a, b: b + a)
    def __sub__(self, o): return self._b(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._b(o, lambda a, b: b - a)
    def __mul__(self, o): return self._b(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._b(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._b(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._b(o, lambda a, b: b / a)
    def __neg__(self): return T(-self.v)


class _Shape(tuple):
    def as_list(self):
        return [int(s) for s in self]


class Variable(T):
    def __init__(self, name, value, trainable):
        super().__init__(value.clone().requires_grad_(trainable))
        self.name, self.trainable = name + ":0", trainable
        self.op = types.SimpleNamespace(type="VariableV2", name=name)


def _u(x):
    return x.v if isinstance(x, T) else x


def _w(x):
    return x if isinstance(x, T) else T(x)


def _scope_path():
    return "/".join(s for s, _ in _STATE.scope)


def _reuse():
    return any(r for _, r in _STATE.scope)


def _unique(name):
    """slim/tf.layers default-name uniquification inside the current scope: Conv, Conv_1, ..."""
    key = (_scope_path(), name)
    n = _STATE.unique.get(key, 0)
    _STATE.unique[key] = n + 1
    return name if n == 0 else "%s_%d" % (name, n)


def _get_var(name, shape, trainable=True):
    full = (_scope_path() + "/" + name) if _scope_path() else name
    if full in _STATE.vars:
        if not _reuse():
            raise ValueError("Variable %s already exists, disallowed." % full)
        return _STATE.vars[full]
    if _reuse():
        raise ValueError("Variable %s does not exist (reuse=True)" % full)
    if full not in _STATE.preset:
        raise KeyError("reference created variable %r that the oracle's spec does not know" % full)
    val = _STATE.preset[full]
    assert tuple(val.shape) == tuple(shape), "shape of %s: reference %s vs oracle %s" % (full, tuple(shape), tuple(val.shape))
    is_tr = trainable and full.split("/")[0] in _STATE.trainable_prefixes
    v = Variable(full, val, is_tr)
    _STATE.vars[full] = v
    return v


# --------------------------------------------------------------------------------------------------
class _VarScope:
    def __init__(self, name, default_name=None, values=None, reuse=None):
        self.name_arg, self.reuse = name if name is not None else default_name, reuse

    def __enter__(self):
        _STATE.scope.append((self.name_arg, bool(self.reuse)))
        self.name = _scope_path()
        return self

    def __exit__(self, *a):
        # TF's close_variable_subscopes: default-name counters of all sub-scopes restart when a scope is left, which
        # is what lets a re-entered scope (reuse=True) regenerate the same 'Conv', 'BatchNorm', 'dense' names
        path = _scope_path()
        for key in list(_STATE.unique):
            if key[0] == path or key[0].startswith(path + "/"):
                del _STATE.unique[key]
        _STATE.scope.pop()


def _collection(name):
    return _STATE.collections.setdefault(name, [])


class _GraphKeys:
    TRAINABLE_VARIABLES, GLOBAL_VARIABLES, MODEL_VARIABLES = "trainable_variables", "variables", "model_variables"
    UPDATE_OPS, SUMMARIES = "update_ops", "summaries"


def _get_collection(key, scope=None):
    if key in (_GraphKeys.TRAINABLE_VARIABLES, _GraphKeys.GLOBAL_VARIABLES, _GraphKeys.MODEL_VARIABLES):
        out = []
        for name, v in _STATE.vars.items():
            if scope is not None and not name.startswith(scope):
                continue
            if key == _GraphKeys.TRAINABLE_VARIABLES and not v.trainable:
                continue
            out.append(v)
        return out
    return list(_collection(key))


# --------------------------------------------------------------------------------------------------
# slim
# --------------------------------------------------------------------------------------------------
def _arg(fn_name, kw):
    merged = {}
    for funcs, akw in _STATE.arg_scope:
        if fn_name in funcs:
            merged.update(akw)
    merged.update(kw)
    return merged


@contextlib.contextmanager
def _arg_scope(funcs, **kw):
    _STATE.arg_scope.append(([f.__name__ for f in funcs], kw))
    try:
        yield None
    finally:
        _STATE.arg_scope.pop()


def _record_output(kw, alias, out):
    coll = kw.get("outputs_collections")
    if coll:
        _collection(coll).append((alias, out))


def conv2d(inputs, num_outputs, kernel_size, stride=1, padding='SAME', **kw):
    kw = _arg("conv2d", kw)
    assert padding == 'SAME' and kw.get("data_format", "NHWC") == "NHWC"
    k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
    act = kw.get("activation_fn", "relu_default")
    scope = kw.get("scope") or _unique("Conv")
    with _VarScope(scope, reuse=kw.get("reuse")):
        x = _u(inputs)
        w = _get_var("weights", (k, k, x.shape[-1], num_outputs))
        b = None if ("biases_initializer" in kw and kw["biases_initializer"] is None) else _get_var("biases", (num_outputs,))
        y = O.conv2(x, w.v, b.v if b is not None else None, stride)
        if act == "relu_default" or act is relu:
            y = O.relu(y)
        elif act is not None:
            y = _u(act(T(y)))
        out = T(y)
        _record_output(kw, _scope_path(), out)
    return out


def conv2d_transpose(inputs, num_outputs, kernel_size, stride=1, padding='SAME', **kw):
    kw = _arg("conv2d_transpose", kw)
    assert padding == 'SAME' and kw.get("activation_fn", None) is None
    k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
    with _VarScope(kw.get("scope") or _unique("Conv2d_transpose"), reuse=kw.get("reuse")):
        x = _u(inputs)
        w = _get_var("weights", (k, k, num_outputs, x.shape[-1]))
        b = None if ("biases_initializer" in kw and kw["biases_initializer"] is None) else _get_var("biases", (num_outputs,))
        return T(O.conv2_tran(x, w.v, b.v if b is not None else None, stride))


def max_pool2d(inputs, kernel_size, stride=2, padding='VALID', **kw):
    kw = _arg("max_pool2d", kw)
    assert list(kernel_size) == [2, 2] and stride == 2 and padding == 'VALID'
    out = T(O.maxpool(_u(inputs)))
    _record_output(kw, (_scope_path() + "/" if _scope_path() else "") + (kw.get("scope") or "MaxPool2D"), out)
    return out


def batch_norm(inputs, decay=0.999, epsilon=0.001, updates_collections="update_ops", scale=False, fused=None,
               is_training=True, **kw):
    assert is_training and not scale
    with _VarScope(kw.get("scope") or _unique("BatchNorm")):
        x = _u(inputs)
        c = x.shape[-1]
        beta = _get_var("beta", (c,))
        mm, mv = _get_var("moving_mean", (c,), False), _get_var("moving_variance", (c,), False)
        y, mean, var = O.batchnorm(x, beta.v, epsilon)
        n = x.numel() // c

        def upd(mean=mean.detach(), var=var.detach()):
            mm.v.data.mul_(decay).add_(mean * (1 - decay))
            mv.v.data.mul_(decay).add_(var * (n / max(n - 1, 1)) * (1 - decay))
        _collection(updates_collections).append(upd)
        _STATE.pending.append(upd)
        return T(y)


def repeat(inputs, repetitions, layer, *args, **kw):
    scope = kw.pop("scope")
    net = inputs
    with _VarScope(scope):
        for i in range(repetitions):
            net = layer(net, *args, scope="%s_%d" % (scope, i + 1), **kw)
    return net


def fully_connected(*a, **k):
    raise NotImplementedError


def relu(x):
    return T(O.relu(_u(x)))


# --------------------------------------------------------------------------------------------------
# optimiser / EMA
# --------------------------------------------------------------------------------------------------
class AdamOptimizer:
    def __init__(self, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t, self.slots = learning_rate, beta1, beta2, epsilon, 0, {}

    def compute_gradients(self, loss, var_list):
        gs = torch.autograd.grad(_u(loss), [v.v for v in var_list], retain_graph=True, allow_unused=True)
        out = []
        for g, v in zip(gs, var_list):
            g = torch.zeros_like(v.v) if g is None else g
            _STATE.captured_grads[v.name[:-2]] = g.detach().clone()
            out.append((T(g), v))
        return out

    def apply_gradients(self, grads_and_vars):
        def run():
            self.t += 1
            lr = float(_u(self.lr)) if isinstance(self.lr, T) else float(self.lr)
            for g, v in grads_and_vars:
                m, vv = self.slots.setdefault(v.name, (torch.zeros_like(v.v), torch.zeros_like(v.v)))
                O.adam_tf_step(v.v.data, _u(g).detach(), m, vv, self.t, lr, self.b1, self.b2, self.eps)
        _STATE.pending.append(run)
        return run


class ExponentialMovingAverage:
    def __init__(self, decay):
        self.decay, self.shadow = decay, {}

    def apply(self, values):
        for x in values:
            self.shadow.setdefault(id(x), T(torch.zeros(())))

        def run():
            for x in values:
                s = self.shadow[id(x)]
                s.v = O.ema_tf(s.v, _u(x).detach() if isinstance(_u(x), torch.Tensor) else float(x), self.decay)
        _STATE.pending.append(run)
        return run

    def average(self, x):
        return self.shadow[id(x)]


# --------------------------------------------------------------------------------------------------
# module assembly
# --------------------------------------------------------------------------------------------------
def _reduce(fn):
    def f(x, axis=None, keepdims=False, **kw):
        x = _u(x)
        if axis is None:
            return T(fn(x))
        return T(fn(x, dim=tuple(axis) if isinstance(axis, (list, tuple)) else axis, keepdim=keepdims))
    return f


def _pad(x, paddings, mode="CONSTANT"):
    x, p = _u(x), np.asarray(_u(paddings)).astype(int).tolist()
    if mode == "CONSTANT":
        out = torch.zeros([s + a + b for s, (a, b) in zip(x.shape, p)], dtype=x.dtype)
        out[tuple(slice(a, a + s) for s, (a, b) in zip(x.shape, p))] = x
        return T(out)
    assert mode == "SYMMETRIC"
    for d, (a, b) in enumerate(p):
        assert a == 0
        if b:
            x = torch.cat((x, x.narrow(d, x.shape[d] - b, b).flip(d)), d)
    return T(x)


def _cond(pred, fn1, fn2):
    return fn1() if bool(_u(pred)) else fn2()


def _assign(var, value):
    def run():
        var.v.data.copy_(_u(value)) if var.v.dim() else var.v.data.fill_(float(_u(value)))
    _STATE.pending.append(run)
    return run


def _resize_images(x, size, *a, **k):
    size = [int(s) for s in (_u(size).tolist() if isinstance(_u(size), (torch.Tensor, np.ndarray)) else size)]
    return T(O.resize_bilinear_legacy(_u(x), size[0], size[1]))


def build_modules():
    """Create fake `tensorflow`, `tensorflow.contrib.slim`, `keras`, `cv2` ... modules (dict name -> module)."""
    tf = types.ModuleType("tensorflow")
    null = contextlib.nullcontext
    tf.float32, tf.int32, tf.int64, tf.uint8, tf.string = torch.float32, torch.int32, torch.int64, torch.uint8, str
    tf.variable_scope, tf.name_scope, tf.device = _VarScope, (lambda *a, **k: null()), (lambda *a, **k: null())
    tf.control_dependencies = lambda deps: null()
    tf.GraphKeys = _GraphKeys
    tf.get_collection, tf.add_to_collection = _get_collection, (lambda name, value: _collection(name).append(value))
    tf.reshape = lambda x, shape: T(_u(x).reshape([int(s) for s in shape]))
    tf.transpose = lambda x, perm: T(_u(x).permute(*perm))
    tf.concat = lambda xs, axis: T(torch.cat([_u(x) for x in xs], dim=axis))
    tf.stack = lambda xs, axis=0: T(torch.stack([_u(x) for x in xs], dim=axis))
    tf.split = lambda x, n, axis: [T(c) for c in torch.chunk(_u(x), n, dim=axis)]
    tf.identity = lambda x, **k: _w(x)
    tf.stop_gradient = lambda x: T(_u(x).detach())
    tf.shape = lambda x: np.array(list(_u(x).shape), dtype=np.int64)
    tf.zeros = lambda shape, dtype=torch.float32: T(torch.zeros([int(s) for s in shape], dtype=dtype))
    tf.zeros_like = lambda x: T(torch.zeros_like(_u(x)))
    tf.constant = lambda v, dtype=None, shape=None, name=None: T(torch.as_tensor(np.asarray(v), dtype=dtype or (
        torch.float32 if np.asarray(v).dtype.kind == "f" else torch.int64)))
    tf.cast = lambda x, dtype: T(torch.as_tensor(_u(x)).to(dtype)) if dtype in (torch.float32, torch.int32, torch.int64) else _w(x)
    tf.reduce_mean, tf.reduce_sum = _reduce(torch.mean), _reduce(torch.sum)
    tf.square, tf.abs, tf.sqrt = (lambda x: T(_u(x) ** 2)), (lambda x: T(_u(x).abs())), (lambda x: T(_u(x).sqrt()))
    tf.log, tf.tanh = (lambda x: T(torch.log(_u(x)))), (lambda x: T(torch.tanh(_u(x))))
    tf.multiply = lambda a, b: _w(a) * b
    tf.minimum = lambda a, b: T(torch.minimum(torch.as_tensor(_u(a), dtype=torch.float32), torch.as_tensor(_u(b), dtype=torch.float32)))
    tf.maximum = lambda a, b: T(torch.maximum(torch.as_tensor(_u(a), dtype=torch.float32), torch.as_tensor(_u(b), dtype=torch.float32)))
    tf.less = lambda a, b: T(torch.as_tensor(_u(a)) < torch.as_tensor(_u(b)))
    tf.equal = lambda a, b: T(torch.as_tensor(_u(a)) == torch.as_tensor(_u(b)))
    tf.floormod = lambda a, b: T(torch.as_tensor(_u(a)) % torch.as_tensor(_u(b)))
    tf.where = lambda c, a, b: T(torch.where(_u(c), _u(a), _u(b)))
    tf.cond, tf.pad, tf.assign = _cond, _pad, _assign
    tf.assign_add = lambda var, d: _assign(var, T(_u(var).detach() + d))
    tf.group = lambda *ops: list(ops)
    tf.zeros_initializer = lambda: "zeros"
    tf.get_variable = lambda name=None, shape=(), initializer=None, dtype=None, **k: _get_scalar(name, dtype)
    tf.nn = types.SimpleNamespace(relu=relu, sigmoid=lambda x: T(torch.sigmoid(_u(x))))
    tf.image = types.SimpleNamespace(
        resize_images=_resize_images,
        crop_to_bounding_box=lambda x, oy, ox, h, w: T(_u(x)[:, oy:oy + h, ox:ox + w]),
        flip_left_right=lambda x: T(_u(x).flip(-2)))
    tf.train = types.SimpleNamespace(
        AdamOptimizer=AdamOptimizer, ExponentialMovingAverage=ExponentialMovingAverage,
        get_or_create_global_step=lambda: _global_step(),
        exponential_decay=lambda lr, gs, ds, dr, staircase=False: T(torch.tensor(
            O.exponential_decay(lr, float(_u(gs)), ds, dr, staircase), dtype=torch.float32)))
    tf.layers = types.SimpleNamespace(Dense=_Dense)
    tf.logging = types.SimpleNamespace(warning=lambda *a, **k: None)
    tf.Summary, tf.py_func = object, None

    contrib = types.ModuleType("tensorflow.contrib")
    slim = types.ModuleType("tensorflow.contrib.slim")
    for f in (conv2d, conv2d_transpose, max_pool2d, batch_norm, repeat, fully_connected):
        setattr(slim, f.__name__, f)
    slim.arg_scope = _arg_scope
    slim.l2_regularizer = lambda *a, **k: None
    slim.utils = types.SimpleNamespace(convert_collection_to_dict=lambda name: dict(_collection(name)))
    contrib.slim = slim
    contrib.layers = types.SimpleNamespace(xavier_initializer=lambda *a, **k: "xavier")
    contrib.image = types.SimpleNamespace(dense_image_warp=lambda img, flow: T(O.dense_image_warp(_u(img), _u(flow))))
    tf.contrib = contrib
    py = types.ModuleType("tensorflow.python")
    pyops = types.ModuleType("tensorflow.python.ops")
    sou = types.ModuleType("tensorflow.python.ops.summary_op_util")
    pyops.summary_op_util = sou
    py.ops = pyops
    tf.python = py

    keras = types.ModuleType("keras")

    class LeakyReLU:
        def __init__(self, alpha):
            self.alpha = alpha

        def call(self, x):
            return T(O.lrelu(_u(x), self.alpha))
    keras.layers = types.SimpleNamespace(LeakyReLU=LeakyReLU)
    cv2 = types.ModuleType("cv2")
    return {"tensorflow": tf, "tensorflow.contrib": contrib, "tensorflow.contrib.slim": slim, "tensorflow.python": py,
            "tensorflow.python.ops": pyops, "tensorflow.python.ops.summary_op_util": sou, "keras": keras, "cv2": cv2}


class _Dense:
    def __init__(self, units, activation=None, kernel_initializer=None):
        self.units = units

    def apply(self, inputs):
        with _VarScope(_unique("dense")):
            x = _u(inputs)
            self.kernel = _get_var("kernel", (x.shape[-1], self.units))
            self.bias = _get_var("bias", (self.units,))
            return T(O.denselayer(x, self.kernel.v, self.bias.v))


def _global_step():
    if "global_step" not in _STATE.vars:
        _STATE.vars["global_step"] = Variable("global_step", torch.zeros((), dtype=torch.float32), False)
    return _STATE.vars["global_step"]


def _get_scalar(name, dtype):
    full = (_scope_path() + "/" + name) if _scope_path() else name
    if full not in _STATE.vars:
        _STATE.vars[full] = Variable(full, torch.zeros(()), False)
    return _STATE.vars[full]


@contextlib.contextmanager
def reference_modules(reference_root="/root/reference"):
    """Temporarily install the fake modules and make `lib.*` resolve to the reference's own files."""
    fakes = build_modules()
    saved = {k: sys.modules.get(k) for k in list(fakes) + ["lib", "lib.ops", "lib.frvsr", "lib.Teco", "lib.dataloader"]}
    for k in ("lib", "lib.ops", "lib.frvsr", "lib.Teco", "lib.dataloader"):
        sys.modules.pop(k, None)
    sys.modules.update(fakes)
    import scipy.signal
    if not hasattr(scipy.signal, "gaussian"):          # moved to scipy.signal.windows in SciPy >= 1.13
        scipy.signal.gaussian = scipy.signal.windows.gaussian
    # the reference's lib/ has no __init__.py (namespace package): a regular `lib` package anywhere on sys.path
    # (this repository's mirror) would win, so hide every other path entry that has a lib/ directory
    import os
    hidden = [p for p in sys.path if p != reference_root and os.path.isdir(os.path.join(p or ".", "lib"))]
    old_path = list(sys.path)
    sys.path[:] = [reference_root] + [p for p in sys.path if p not in hidden]
    try:
        import lib.Teco as RT          # the reference's file (imports lib.frvsr -> lib.dataloader -> lib.ops)
        RT.gif_summary = lambda *a, **k: None     # TensorBoard gifs: out of scope
        yield RT
    finally:
        sys.path[:] = old_path
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
