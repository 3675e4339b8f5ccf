"""Stand-in for the `tensorflow.compat.v1` symbols of cornac/models/wmf/{wmf.py,recom_wmf.py} — TEST INFRASTRUCTURE.

A lazily evaluated expression graph over torch float32 tensors.  Forward values and gradients are torch's (autograd of
the loss the REFERENCE's code builds).  Restated from TensorFlow (2.12, the version the model pins), not computed by it:

  * the gradient of `tf.gather(var, ids)` w.r.t. `var` is an IndexedSlices (values = the rows' gradients, indices = ids)
    — tensorflow/python/ops/array_grad.py `_GatherV2Grad`;
  * `tf.clip_by_value` of an IndexedSlices clips its values and keeps it sparse — ops/clip_ops.py `clip_by_value`;
  * `tf.train.AdamOptimizer` (python/training/adam.py), beta1 0.9, beta2 0.999, epsilon 1e-8,
    lr_t = lr * sqrt(1 - beta2_power) / (1 - beta1_power) with the powers kept as float32 variables multiplied after
    every apply (`_finish`);  dense (`ApplyAdam`): m += (g - m)(1 - beta1); v += (g^2 - v)(1 - beta2);
    var -= lr_t m / (sqrt(v) + epsilon);  sparse (`_apply_sparse_shared`): m <- m beta1 for EVERY row, then
    scatter-add of g (1 - beta1) on the slice rows, same for v, then var -= lr_t m / (sqrt(v) + epsilon) for EVERY row;
  * `tf.nn.l2_loss(t)` = sum(t^2) / 2.
"""
import contextlib
import types

import numpy as np
import torch

float32, int32 = torch.float32, torch.int32
_state = types.SimpleNamespace(graph=None)


class Graph:
    def __init__(self):
        self.variables = []

    @contextlib.contextmanager
    def as_default(self):
        prev, _state.graph = _state.graph, self
        try:
            yield self
        finally:
            _state.graph = prev


def _graph():
    if _state.graph is None:
        _state.graph = Graph()
    return _state.graph


def reset_default_graph():
    _state.graph = None


def set_random_seed(seed):
    torch.manual_seed(int(seed))


class Node:
    def __init__(self, fn, deps=()):
        self.fn, self.deps = fn, deps

    def eval(self, run):
        if self not in run.cache:
            run.cache[self] = self.fn(run, *[d.eval(run) if isinstance(d, Node) else d for d in self.deps])
        return run.cache[self]

    def __sub__(self, o): return Node(lambda r, a, b: a - b, (self, o))
    def __rsub__(self, o): return Node(lambda r, a, b: b - a, (self, o))
    def __add__(self, o): return Node(lambda r, a, b: a + b, (self, o))
    __radd__ = __add__
    def __mul__(self, o): return Node(lambda r, a, b: a * b, (self, o))
    __rmul__ = __mul__
    __hash__ = object.__hash__


class Variable(Node):
    def __init__(self, name, init):
        super().__init__(None)
        self.name, self.init, self.value, self.gathered_by = name, init, None, None

    def eval(self, run):
        return self.value


class IndexedSlices:
    def __init__(self, values, indices):
        self.values, self.indices = values, indices


def constant(value, dtype=None):
    t = torch.tensor(np.asarray(value))
    return Node(lambda r: t)


def placeholder(dtype=None, shape=None, name=None):
    node = Node(None)
    node.fn = lambda r: torch.as_tensor(np.asarray(r.feed[node]), dtype=dtype)
    return node


@contextlib.contextmanager
def variable_scope(name):
    yield


def get_variable(name, dtype=None, initializer=None):
    v = Variable(name, initializer)
    _graph().variables.append(v)
    return v


def trainable_variables():
    return list(_graph().variables)


def global_variables_initializer():
    def init(run):
        for v in run.graph.variables:
            v.value = v.init.eval(run).clone().to(torch.float32).requires_grad_(True)
    return Node(init)


def gather(params, indices):
    if isinstance(params, Variable):
        params.gathered_by = indices
    return Node(lambda r, p, i: p[i.long()], (params, indices))


def reshape(t, shape):
    return Node(lambda r, a: a.reshape(*shape), (t,))


def matmul(a, b, transpose_b=False):
    return Node(lambda r, x, y: x @ (y.T if transpose_b else y), (a, b))


def square(t):
    return Node(lambda r, a: a * a, (t,))


def multiply(a, b):
    return Node(lambda r, x, y: x * y, (a, b))


def reduce_sum(t):
    return Node(lambda r, a: a.sum(), (t,))


def clip_by_value(t, lo, hi):
    def clip(r, g):
        if isinstance(g, IndexedSlices):
            return IndexedSlices(g.values.clamp(lo, hi), g.indices)
        return g.clamp(lo, hi)
    return Node(clip, (t,))


nn = types.SimpleNamespace(l2_loss=lambda t: Node(lambda r, a: (a * a).sum() / 2, (t,)))


class AdamOptimizer:
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon
        self.b1p, self.b2p = np.float32(beta1), np.float32(beta2)
        self.slots = {}

    def compute_gradients(self, loss, var_list=None):
        var_list = list(var_list)

        def all_grads(run, loss_value):
            return dict(zip(var_list, torch.autograd.grad(loss_value, [v.value for v in var_list])))
        grads = Node(all_grads, (loss,))

        def one(v):
            def pick(run, g):
                dense = g[v]
                if v.gathered_by is not None:
                    ids = v.gathered_by.eval(run).long().reshape(-1)
                    return IndexedSlices(dense[ids], ids)
                return dense
            return Node(pick, (grads,))
        return [(one(v), v) for v in var_list]

    def apply_gradients(self, grads_and_vars):
        f = np.float32

        def apply(run, *grads):
            lr_t = f(f(self.lr) * np.sqrt(f(1) - self.b2p) / (f(1) - self.b1p))
            with torch.no_grad():
                for g, (_, v) in zip(grads, grads_and_vars):
                    m, s = self.slots.setdefault(v, (torch.zeros_like(v.value), torch.zeros_like(v.value)))
                    if isinstance(g, IndexedSlices):
                        m.mul_(float(f(self.b1)))
                        m.index_add_(0, g.indices, g.values * float(f(1) - f(self.b1)))
                        s.mul_(float(f(self.b2)))
                        s.index_add_(0, g.indices, g.values * g.values * float(f(1) - f(self.b2)))
                    else:
                        m.add_((g - m) * float(f(1) - f(self.b1)))
                        s.add_((g * g - s) * float(f(1) - f(self.b2)))
                    v.value.sub_(float(lr_t) * m / (s.sqrt() + float(f(self.eps))))
            self.b1p, self.b2p = f(self.b1p * f(self.b1)), f(self.b2p * f(self.b2))
        return Node(apply, tuple(g for g, _ in grads_and_vars))


train = types.SimpleNamespace(AdamOptimizer=AdamOptimizer)


class ConfigProto:
    def __init__(self):
        self.gpu_options = types.SimpleNamespace(allow_growth=False)


class _Run:
    def __init__(self, graph, feed):
        self.graph, self.feed, self.cache = graph, feed or {}, {}


class Session:
    def __init__(self, config=None, graph=None):
        self.graph = graph or _graph()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def run(self, fetches, feed_dict=None):
        run = _Run(self.graph, feed_dict)
        single = not isinstance(fetches, (list, tuple))
        items = [fetches] if single else list(fetches)
        # values first (the forward pass reads the variables before any update op of the same run touches them)
        order = sorted(range(len(items)), key=lambda t: isinstance(items[t].fn, type(None)) or items[t].fn.__name__ == "apply")
        out = [None] * len(items)
        for t in order:
            val = items[t].eval(run)
            if isinstance(val, torch.Tensor):
                val = val.detach().cpu().numpy()
                val = val.copy() if val.ndim else val[()]
            out[t] = val
        return out[0] if single else out


logging = types.SimpleNamespace(set_verbosity=lambda level: None, ERROR=40)
compat = types.SimpleNamespace(v1=types.SimpleNamespace(logging=logging))
