"""TEST INFRASTRUCTURE ONLY — Python face of the CPU oracle (oracle/cornac_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
It restates, on the CPU, what the reference's seeded (num_threads == 1) path
computes, end to end:

  * BPROracle / WBPROracle  -> cornac/models/bpr/recom_bpr.pyx:145-206, recom_wbpr.pyx:103-144
  * MFOracle               -> cornac/models/mf/recom_mf.py:138-209 + backend_cpu.pyx:35-97
  * fast_dot / score / rank -> cornac/utils/fast_dot.pyx:40-43, recom_bpr.pyx:272-297,
                               recom_mf.py:254-286, cornac/models/recommender.py:476-530

The initialisers use NumPy's RandomState exactly like the reference
(cornac/utils/init_utils.py:33-57 `uniform`, :60-82 `normal`) so seeded runs
start from bit-identical factors.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcornac_oracle.so")
_lib = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "cornac_oracle.c")
    fast = _SO.replace("libcornac_oracle.so", "libcornac_oracle_fast.so")
    if force or not os.path.exists(_SO) or not os.path.exists(fast) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_fast = None


def fast_lib():
    """same source compiled with the reference's flags (-O3 -ffast-math): CPU-baseline timing only"""
    global _fast
    if _fast is None:
        build()
        _fast = _bind(C.CDLL(_SO.replace("libcornac_oracle.so", "libcornac_oracle_fast.so")))
    return _fast


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _bind(C.CDLL(_SO))
    return _lib


def _bind(L):
    if True:
        L.oracle_mt_seed.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_mt_next.argtypes = [C.c_void_p]
        L.oracle_mt_next.restype = C.c_uint32
        L.oracle_boost_uniform.argtypes = [C.c_void_p, C.c_uint64]
        L.oracle_boost_uniform.restype = C.c_int64
        L.oracle_boost_uniform_fill.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, i64p]
        L.oracle_mt_fill_raw.argtypes = [C.c_void_p, C.c_int64, u32p]
        L.oracle_bpr_epoch_seq.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, i32p, i32p,
                                           i32p, i32p, f32p, f32p, f32p, C.c_int, C.c_float, C.c_float, C.c_int,
                                           C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p,
                                           C.c_void_p]
        f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        L.oracle_bpr_epoch_seq_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, i32p, i32p,
                                               i32p, i32p, f64p, f64p, f64p, C.c_int, C.c_double, C.c_double, C.c_int,
                                               C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.oracle_bpr_epoch_omp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int64, i32p,
                                           i32p, i32p, i32p, f32p, f32p, f32p, C.c_int, C.c_float, C.c_float,
                                           C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.oracle_vebpr_epoch_seq.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, i32p, i32p, i32p,
                                             i32p, i32p, f32p, f32p, C.c_int, C.c_float, C.c_float, C.c_float,
                                             C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.oracle_vebpr_epoch_seq_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, i32p, i32p, i32p,
                                                 i32p, i32p, f64p, f64p, C.c_int, C.c_double, C.c_double, C.c_double,
                                                 C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.oracle_mf_fit.argtypes = [i64p, i64p, f32p, C.c_int64, f32p, f32p, f32p, f32p, C.c_int, C.c_float,
                                    C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_fast_dot.argtypes = [f32p, f32p, f32p, C.c_int64, C.c_int, C.c_int]
        L.oracle_score_block.argtypes = [f32p, f32p, C.c_void_p, C.c_void_p, i32p, C.c_int64, C.c_int64, C.c_int,
                                         f32p]
        L.oracle_philox4x32.argtypes = [C.c_uint32] * 6 + [u32p]
        L.oracle_hogwild_sample.argtypes = [C.c_uint64, C.c_uint32, C.c_int64, C.c_int64, C.c_uint32, C.c_uint32,
                                            i64p, i64p]
        L.oracle_hogwild_sample_owned.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                  C.c_int64, C.c_int64, i64p, i64p]
        L.oracle_strata_rot.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_strata_rot.restype = C.c_uint32
        L.oracle_strata_key.argtypes = [C.c_uint64, C.c_uint32]
        L.oracle_strata_key.restype = C.c_uint32
        L.oracle_strata_partitions.argtypes = [C.c_uint32, C.c_int64, np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")]
        L.oracle_strata_sample.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.c_uint32, i64p, i64p]
        L.oracle_strata_sample.restype = C.c_int64
        L.oracle_ldsbin_key.argtypes = [C.c_uint64, C.c_uint32]
        L.oracle_ldsbin_key.restype = C.c_uint32
        L.oracle_ldsbin_hot_shuffle_key.argtypes = [C.c_uint32]
        L.oracle_ldsbin_hot_shuffle_key.restype = C.c_uint32
        L.oracle_ldsbin_deal.argtypes = [C.c_uint32] * 6 + [i32p, i32p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_ldsbin_deal.restype = None
        L.oracle_ldsbin_layout.argtypes = [C.c_uint32] * 4 + [i32p, C.c_void_p, C.c_void_p]
        L.oracle_ldsbin_layout.restype = None
        L.oracle_ldsbin_epoch_skips.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                C.c_uint32, C.c_uint32, i32p,
                                                i32p, i32p, i32p, i32p, C.c_uint32, i32p, i32p, C.POINTER(C.c_int64),
                                                C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_ldsbin_epoch_skips.restype = C.c_int64
        L.oracle_ldsbin_epoch_skips_range.argtypes = list(L.oracle_ldsbin_epoch_skips.argtypes) + [C.c_uint32, C.c_uint32]
        L.oracle_ldsbin_epoch_skips_range.restype = C.c_int64
        L.oracle_num_threads.restype = C.c_int
        L.oracle_sizeof_mt.restype = C.c_int
    return L


class MT19937:
    """boost::random::mt19937 (recom_bpr.pxd:26-32)."""

    def __init__(self, seed):
        self.buf = C.create_string_buffer(lib().oracle_sizeof_mt())
        lib().oracle_mt_seed(self.buf, int(seed) & 0xFFFFFFFF)

    @property
    def ptr(self):
        return C.cast(self.buf, C.c_void_p)

    def raw(self, n):
        out = np.empty(n, np.uint32)
        lib().oracle_mt_fill_raw(self.ptr, n, out)
        return out

    def uniform_int(self, hi, n):
        out = np.empty(n, np.int64)
        if lib().oracle_boost_uniform_fill(self.ptr, int(hi), n, out) != 0:
            raise ValueError("range >= 2**32 is not reachable in the reference")
        return out


def rngvector_seed(seed):
    """RNGVector(1, rows, seed): thread 0's mt19937 seed = RandomState(seed).randint(2**31)
    (recom_bpr.pyx:55-59)."""
    return int(np.random.RandomState(seed).randint(2 ** 31))


def rngvector_seeds(seed, num_threads):
    rng = np.random.RandomState(seed)
    return [int(rng.randint(2 ** 31)) for _ in range(num_threads)]


def _uniform(shape, rng, dtype=np.float32):
    # cornac/utils/init_utils.py:33-57  uniform(shape, low=0, high=1).astype(dtype)
    return rng.uniform(0.0, 1.0, shape).astype(dtype)


def _normal(shape, rng, std, dtype=np.float32):
    # cornac/utils/init_utils.py:60-82
    return rng.normal(0.0, std, shape).astype(dtype)


def csr_arrays(train_set):
    X = train_set.matrix
    indptr = np.ascontiguousarray(X.indptr, dtype=np.int32)
    indices = np.ascontiguousarray(X.indices, dtype=np.int32)
    user_ids = np.repeat(np.arange(train_set.num_users), np.ediff1d(indptr)).astype(np.int32)
    return indptr, indices, user_ids


class BPROracle:
    """Seeded BPR exactly as the reference runs it with `seed is not None`."""

    weighted = False  # WBPR: one shared stream, negatives drawn from X.indices

    def __init__(self, k=10, max_iter=100, learning_rate=0.001, lambda_reg=0.01, use_bias=True, seed=None,
                 init_params=None):
        self.k, self.max_iter, self.lr, self.reg, self.use_bias = int(k), max_iter, learning_rate, lambda_reg, use_bias
        self.seed = seed
        self.rng = np.random.RandomState(seed)  # created in __init__ (recom_bpr.pyx:130)
        ip = init_params or {}
        self.u_factors, self.i_factors, self.i_biases = ip.get("U"), ip.get("V"), ip.get("Bi")
        self.record = None

    def _init(self, train_set):
        # Recommender.total_users/total_items = len(uid_map)/len(iid_map) (recommender.py:150-158)
        nu, ni = len(train_set.uid_map), len(train_set.iid_map)
        if self.u_factors is None:
            self.u_factors = (_uniform((nu, self.k), self.rng) - 0.5) / self.k
        if self.i_factors is None:
            self.i_factors = (_uniform((ni, self.k), self.rng) - 0.5) / self.k
        if self.i_biases is None or not self.use_bias:
            self.i_biases = np.zeros(ni, np.float32)
        # `_fit_sgd` is a fused-type function (recom_bpr.pyx:211-214): three float64 tables train in double, three
        # float32 tables in float; a mix fails the buffer check (the reference raises ValueError)
        kinds = {np.asarray(getattr(self, n)).dtype for n in ("u_factors", "i_factors", "i_biases")}
        if kinds == {np.dtype(np.float64)}:
            self.dtype = np.float64
        elif kinds == {np.dtype(np.float32)}:
            self.dtype = np.float32
        else:
            raise ValueError("Buffer dtype mismatch: U, V and Bi must share one dtype")
        for n in ("u_factors", "i_factors", "i_biases"):
            setattr(self, n, np.ascontiguousarray(getattr(self, n), dtype=self.dtype))

    def fit(self, train_set, record=False):
        L = lib()
        self._init(train_set)
        indptr, indices, user_ids = csr_arrays(train_set)
        nnz = len(user_ids)
        if self.weighted:
            g = MT19937(rngvector_seed(self.rng.randint(2 ** 31)))
            gp = gn = g
            neg_ids, neg_hi = indices, nnz - 1
        else:
            neg_ids = np.arange(train_set.num_items, dtype=np.int32)
            gp = MT19937(rngvector_seed(self.rng.randint(2 ** 31)))
            gn = MT19937(rngvector_seed(self.rng.randint(2 ** 31)))
            neg_hi = train_set.num_items - 1
        self.correct, self.skipped = [], []
        if record:
            self.record = []
        for _ in range(self.max_iter):
            c, s = C.c_int64(), C.c_int64()
            rec = [None, None, None]
            if record:
                rec = [np.empty(nnz, np.int64), np.empty(nnz, np.int64), np.empty(nnz, np.uint8)]
            if self.dtype == np.float64:
                assert not record, "the float64 restatement keeps no draw record"
                rc = L.oracle_bpr_epoch_seq_f64(gp.ptr, gn.ptr, nnz - 1, neg_hi, nnz, user_ids, indices, neg_ids, indptr,
                                                self.u_factors, self.i_factors, self.i_biases, self.k, self.lr, self.reg,
                                                int(self.use_bias), C.byref(c), C.byref(s))
            else:
                rc = L.oracle_bpr_epoch_seq(gp.ptr, gn.ptr, nnz - 1, neg_hi, nnz, user_ids, indices, neg_ids, indptr,
                                            self.u_factors, self.i_factors, self.i_biases, self.k, self.lr, self.reg,
                                            int(self.use_bias), C.byref(c), C.byref(s),
                                            *[r.ctypes.data if r is not None else None for r in rec])
            if rc != 0:
                raise ValueError("oracle_bpr_epoch_seq failed")
            self.correct.append(c.value)
            self.skipped.append(s.value)
            if record:
                self.record.append(rec)
        return self

    def score(self, user_idx, mode=1):
        out = np.copy(self.i_biases)
        if out.dtype == np.float64:  # fast_dot's ddot variant (fast_dot.pyx:25-43); the BLAS summation order is not pinned
            return out + self.i_factors @ self.u_factors[user_idx]
        lib().oracle_fast_dot(self.u_factors[user_idx], self.i_factors, out, len(out), self.k, mode)
        return out


class WBPROracle(BPROracle):
    weighted = True


class VEBPROracle:
    """Seeded VEBPR (cornac/models/bpr/recom_vebpr.pyx:100-209): train_set must expose `matrix`
    (purchases) and `view_matrix` (views minus purchases, sorted CSR)."""

    def __init__(self, k=10, max_iter=100, learning_rate=0.01, lambda_reg=0.1, alpha=0.5, seed=None,
                 init_params=None):
        self.k, self.max_iter, self.lr, self.reg, self.alpha = int(k), max_iter, learning_rate, lambda_reg, float(alpha)
        self.rng = np.random.RandomState(seed)
        ip = init_params or {}
        self.u_factor, self.i_factor = ip.get("U"), ip.get("V")

    def fit(self, train_set):
        L = lib()
        nu, ni = len(train_set.uid_map), len(train_set.iid_map)
        if self.u_factor is None:
            self.u_factor = (_uniform((nu, self.k), self.rng) - 0.5) / self.k
        if self.i_factor is None:
            self.i_factor = (_uniform((ni, self.k), self.rng) - 0.5) / self.k
        # float64 tables (both given through init_params) train in double: recom_vebpr.pyx:219 is a fused-type function
        f64 = np.asarray(self.u_factor).dtype == np.float64 and np.asarray(self.i_factor).dtype == np.float64
        dt = np.float64 if f64 else np.float32
        self.u_factor = np.ascontiguousarray(self.u_factor, dt)
        self.i_factor = np.ascontiguousarray(self.i_factor, dt)
        indptr, indices, user_ids = csr_arrays(train_set)
        Vw = train_set.view_matrix
        v_indptr = np.ascontiguousarray(Vw.indptr, np.int32)
        v_indices = np.ascontiguousarray(Vw.indices, np.int32)
        if len(v_indices) == 0:
            v_indices = np.zeros(1, np.int32)
        gp = MT19937(rngvector_seed(self.rng.randint(2 ** 31)))
        gv = MT19937(rngvector_seed(self.rng.randint(2 ** 31)))
        gn = MT19937(rngvector_seed(self.rng.randint(2 ** 31)))
        self.correct, self.skipped = [], []
        for _ in range(self.max_iter):
            c, s = C.c_int64(), C.c_int64()
            (L.oracle_vebpr_epoch_seq_f64 if f64 else L.oracle_vebpr_epoch_seq)(
                gp.ptr, gv.ptr, gn.ptr, len(user_ids), train_set.num_items, user_ids, indices, indptr, v_indices, v_indptr,
                self.u_factor, self.i_factor, self.k, self.lr, self.reg, self.alpha, C.byref(c), C.byref(s))
            self.correct.append(c.value)
            self.skipped.append(s.value)
        return self


def bpr_hogwild_epochs(indptr, indices, user_ids, n_items, U, V, B, k, lr, reg, use_bias, seed, num_threads,
                       epochs, num_samples=None, fast=True):
    """The reference's unseeded multi-thread path (racy), for the CPU throughput baseline.
    num_samples: draws per epoch (default nnz, like the reference)."""
    L = fast_lib() if fast else lib()
    nnz = len(user_ids)
    n_draw = nnz if num_samples is None else int(num_samples)
    T = num_threads
    sz = L.oracle_sizeof_mt()
    rng = np.random.RandomState(seed)
    bufs, keep = [], []
    for _ in range(2):
        raw = C.create_string_buffer(sz * T + 64)
        base = (C.addressof(raw) + 63) & ~63  # cache-line aligned array of T engines
        for t, s in enumerate(rngvector_seeds(rng.randint(2 ** 31), T)):
            L.oracle_mt_seed(C.c_void_p(base + t * sz), s)
        keep.append(raw)
        bufs.append(C.c_void_p(base))
    neg_ids = np.arange(n_items, dtype=np.int32)
    tot_c = tot_s = 0
    for _ in range(epochs):
        c, s = C.c_int64(), C.c_int64()
        L.oracle_bpr_epoch_omp(bufs[0], bufs[1], T, nnz - 1, n_items - 1, n_draw, user_ids, indices, neg_ids, indptr,
                               U, V, B, k, lr, reg, int(use_bias), C.byref(c), C.byref(s))
        tot_c += c.value
        tot_s += s.value
    return tot_c, tot_s


class MFOracle:
    """MF(backend='cpu') seeded (recom_mf.py:138-209)."""

    def __init__(self, k=10, max_iter=20, learning_rate=0.01, lambda_reg=0.02, use_bias=True, early_stop=False,
                 seed=None, init_params=None, num_threads=1):
        self.k, self.max_iter, self.lr, self.reg = int(k), max_iter, learning_rate, lambda_reg
        self.use_bias, self.early_stop, self.seed, self.num_threads = use_bias, early_stop, seed, num_threads
        ip = init_params or {}
        self.u_factors, self.i_factors = ip.get("U"), ip.get("V")
        self.u_biases, self.i_biases = ip.get("Bu"), ip.get("Bi")

    def fit(self, train_set):
        rng = np.random.RandomState(self.seed)
        nu, ni = train_set.num_users, train_set.num_items
        if self.u_factors is None:
            self.u_factors = _normal((nu, self.k), rng, 0.01)
        if self.i_factors is None:
            self.i_factors = _normal((ni, self.k), rng, 0.01)
        if self.u_biases is None:
            self.u_biases = np.zeros(nu, np.float32)
        if self.i_biases is None:
            self.i_biases = np.zeros(ni, np.float32)
        for n in ("u_factors", "i_factors", "u_biases", "i_biases"):
            setattr(self, n, np.ascontiguousarray(getattr(self, n), dtype=np.float32))
        self.global_mean = np.float32(train_set.global_mean if self.use_bias else 0.0)
        rid, cid, val = train_set.uir_tuple
        rid = np.ascontiguousarray(rid, np.int64)
        cid = np.ascontiguousarray(cid, np.int64)
        val = np.ascontiguousarray(val, np.float32)
        self.loss = np.zeros(max(self.max_iter, 1), np.float32)
        self.epochs_run = lib().oracle_mf_fit(rid, cid, val, len(val), self.u_factors, self.i_factors, self.u_biases,
                                              self.i_biases, self.k, self.lr, self.reg, float(self.global_mean),
                                              self.max_iter, self.num_threads, int(self.use_bias),
                                              int(self.early_stop), self.loss.ctypes.data)
        return self

    def score(self, user_idx, mode=1):
        out = (self.global_mean + self.i_biases).astype(np.float32)
        out += self.u_biases[user_idx]
        lib().oracle_fast_dot(self.u_factors[user_idx], self.i_factors, out, len(out), self.k, mode)
        return out


def fast_dot(vec, mat, output, mode=0):
    """output += mat @ vec in place (cornac/utils/fast_dot.pyx:40-43)."""
    vec = np.ascontiguousarray(vec, np.float32)
    mat = np.ascontiguousarray(mat, np.float32)
    assert output.dtype == np.float32 and output.flags.c_contiguous
    lib().oracle_fast_dot(vec, mat, output, mat.shape[0], mat.shape[1], mode)


def score_block(U, V, item_base, user_base, users):
    users = np.ascontiguousarray(users, np.int32)
    out = np.empty((len(users), V.shape[0]), np.float32)
    ib = None if item_base is None else np.ascontiguousarray(item_base, np.float32).ctypes.data
    ub = None if user_base is None else np.ascontiguousarray(user_base, np.float32).ctypes.data
    lib().oracle_score_block(np.ascontiguousarray(U, np.float32), np.ascontiguousarray(V, np.float32), ib, ub, users,
                             len(users), V.shape[0], V.shape[1], out)
    return out


def rank(all_item_scores, num_items, total_items, item_indices=None, k=-1):
    """Recommender.rank after score() (cornac/models/recommender.py:503-530).

    The reference's tie order is unspecified (np.argsort default / argpartition;
    acknowledged in tests/cornac/models/test_recommender.py:89-93).  The oracle pins
    it: descending score, ties by descending position in `item_indices` (= a
    stable ascending argsort read backwards).  With k != -1 only the first k
    entries are defined (the reference leaves the tail in argpartition order).
    """
    known = np.asarray(all_item_scores)
    if len(known) != total_items:
        full = np.ones(total_items) * np.min(known)
        full[:num_items] = known
        known = full
    item_indices = np.arange(num_items) if item_indices is None else np.asarray(item_indices)
    item_scores = known[item_indices]
    order = np.argsort(item_scores, kind="stable")[::-1]
    ranked = item_indices[order]
    if k != -1:
        ranked = ranked[:k]
    return ranked, item_scores


def hogwild_sample(seed, epoch, s0, n, n_pos, n_neg):
    ii = np.empty(n, np.int64)
    jj = np.empty(n, np.int64)
    lib().oracle_hogwild_sample(int(seed), int(epoch), int(s0), int(n), int(n_pos), int(n_neg), ii, jj)
    return ii, jj


def strata_key(seed, epoch):
    return int(lib().oracle_strata_key(int(seed), int(epoch)))


def strata_partitions(key, n_items):
    """partition (0..7) of every popularity rank under the epoch key (csrc/bpr_strata.inc)"""
    out = np.empty(n_items, np.uint8)
    lib().oracle_strata_partitions(int(key), int(n_items), out)
    return out


def strata_sample(seed, epoch, key, wave_id, p, length, n_items):
    """(index into the wave's bucket p, popularity rank of the negative) for the bucket's `length` draws"""
    r = np.empty(length, np.int64)
    code = np.empty(length, np.int64)
    n = lib().oracle_strata_sample(int(seed), int(epoch), int(key), int(wave_id), int(p), int(length), int(n_items), r, code)
    return r[:n], code[:n]


def strata_buckets(wave_ptr, own_u, own_i, deg, key, n_hot):
    """CPU restatement of strata_bucket_kernel: every wave slice stably re-ordered by the partition of its items.
    Returns (sptr [8 W + 1], rec_u, rec_i with bit 31 = hot, rank_item)."""
    n_items = len(deg)
    rank_item = np.argsort(-deg.astype(np.int64), kind="stable").astype(np.int32)
    item_rank = np.empty(n_items, np.int64)
    item_rank[rank_item] = np.arange(n_items)
    part_rank = strata_partitions(key, n_items)
    W = len(wave_ptr) - 1
    code = item_rank[own_i]
    part = part_rank[code].astype(np.int64)
    wave_of = np.repeat(np.arange(W, dtype=np.int64), np.diff(wave_ptr))
    order = np.argsort(wave_of * 8 + part, kind="stable")
    counts = np.bincount(wave_of * 8 + part, minlength=8 * W)
    sptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    rec_u = own_u[order]
    rec_i = own_i[order].astype(np.int64)
    rec_i = np.where(code[order] < n_hot, rec_i | 0x80000000, rec_i).astype(np.uint32).view(np.int32)
    return sptr, rec_u, rec_i, rank_item


def _ldsbin_mix(h):
    """murmur3 finaliser variant of csrc/bpr_ldsbin.inc ldsbin_mix on uint32 arrays."""
    h = np.asarray(h, np.uint64) & 0xFFFFFFFF
    h ^= h >> np.uint64(15); h = (h * np.uint64(0x85EBCA77)) & 0xFFFFFFFF
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE3D)) & 0xFFFFFFFF
    h ^= h >> np.uint64(16)
    return h.astype(np.uint32)


def ldsbin_tables(indptr, indices, n_items, n_bins, hot_x1000, share=None):
    """The static tables of the LDS-bin form (csrc/bpr.hip ldsbin_build): CSC, popularity ranks, hot items and their
    interaction list in the shuffled order (stable order of a hash of the item-major list index).  share: the interaction
    count the hot threshold is a fraction of — nnz / n_bins (resident bins, the default) or nnz / (4 x CUs) (passing bins)."""
    import scipy.sparse as sp

    indptr = np.ascontiguousarray(indptr, np.int32)
    indices = np.ascontiguousarray(indices, np.int32)
    nnz, n_users = len(indices), len(indptr) - 1
    X = sp.csr_matrix((np.ones(nnz, np.int8), indices, indptr), shape=(n_users, n_items)).tocsc()
    X.sort_indices()
    cptr, cusers = X.indptr.astype(np.int32), X.indices.astype(np.int32)
    deg = np.diff(cptr)
    rank_item = np.argsort(-deg.astype(np.int64), kind="stable").astype(np.int32)
    share = nnz / n_bins if share is None else float(share)
    n_hot = 0
    while n_hot < n_items and deg[rank_item[n_hot]] * 1000.0 > share * hot_x1000:
        n_hot += 1
    hot_u = np.concatenate([cusers[cptr[i]:cptr[i + 1]] for i in rank_item[:n_hot]] + [np.zeros(0, np.int32)]).astype(np.int32)
    hot_i = np.repeat(rank_item[:n_hot], deg[rank_item[:n_hot]]).astype(np.int32)
    n_hot_inter = len(hot_u)
    if n_hot_inter:
        t = np.arange(n_hot_inter, dtype=np.uint64)
        order = np.argsort(_ldsbin_mix((t * np.uint64(0x9E3779B1) + np.uint64(0x5BD1E995)) & 0xFFFFFFFF), kind="stable")
        hot_u, hot_i = np.ascontiguousarray(hot_u[order]), np.ascontiguousarray(hot_i[order])
    else:
        hot_u = hot_i = np.zeros(1, np.int32)
    return dict(indptr=indptr, indices=indices, cptr=cptr, cusers=cusers, rank_item=rank_item, n_hot=n_hot, hot_u=hot_u,
                hot_i=hot_i, n_hot_inter=n_hot_inter, deg=deg)


def ldsbin_n_strata(n_items, n_bins, strata_groups):
    return max(1, ((n_items + n_bins - 1) // n_bins) // max(1, strata_groups))


def ldsbin_deal_key(key, n_bins, n_items, n_hot, n_strata, hot_cost_x16, rank_item, cptr, n_hot_inter):
    """One deal by its 32-bit key: (bin_of_item, cold_mass, hot_off)."""
    bin_of = np.empty(n_items, np.int32)
    cold, off = np.empty(n_bins, np.uint32), np.empty(n_bins + 1, np.uint32)
    lib().oracle_ldsbin_deal(int(key), int(n_bins), int(n_items), int(n_hot), int(n_strata), int(hot_cost_x16),
                             np.ascontiguousarray(rank_item, np.int32), np.ascontiguousarray(cptr, np.int32),
                             int(n_hot_inter), bin_of.ctypes.data, cold.ctypes.data, off.ctypes.data)
    return bin_of, cold, off


def ldsbin_deal(seed, epoch, n_bins, hot_x1000, indptr, indices, n_items, strata_groups=16, hot_cost_x16=32):
    """CPU restatement of the deal of (seed, epoch): (bin_of_item, cold_mass, hot_off, hot_u, hot_i, n_hot)."""
    t = ldsbin_tables(indptr, indices, n_items, n_bins, hot_x1000)
    key = int(lib().oracle_ldsbin_key(int(seed), int(epoch)))
    bin_of, cold, off = ldsbin_deal_key(key, n_bins, n_items, t["n_hot"], ldsbin_n_strata(n_items, n_bins, strata_groups),
                                        hot_cost_x16, t["rank_item"], t["cptr"], t["n_hot_inter"])
    return bin_of, cold, off, t["hot_u"][:t["n_hot_inter"]], t["hot_i"][:t["n_hot_inter"]], t["n_hot"]


def ldsbin_layout(deal_seed, layout_epoch, n_bins, n_items, rank_item, strata_groups=16):
    """(slot_item [n_bins cap], item_slot [n_items]) of the deal keyed by (deal_seed, layout_epoch): the order in which the
    conveyor's block buffers hold their rows (csrc/bpr_ldsbin.inc ldsbin_layout_kernel)."""
    cap = (n_items + n_bins - 1) // n_bins
    slot_item, item_slot = np.empty(n_bins * cap, np.int32), np.empty(n_items, np.int32)
    lib().oracle_ldsbin_layout(int(lib().oracle_ldsbin_key(int(deal_seed), int(layout_epoch))), int(n_bins), int(n_items),
                               int(ldsbin_n_strata(n_items, n_bins, strata_groups)), np.ascontiguousarray(rank_item, np.int32),
                               slot_item.ctypes.data, item_slot.ctypes.data)
    return slot_item, item_slot


def ldsbin_epoch(seed, epoch, n_bins, hot_x1000, indptr, indices, n_items, count_touches=False, neg_pop=False,
                 strata_groups=16, hot_cost_x16=32, tables=None, share=None, deal=None, bins=None):
    """CPU restatement of one epoch of the LDS-bin sampler (csrc/bpr_ldsbin.inc): returns (skipped, draws, n_hot[,
    positive touches per item, negative touches per item]).  deal = (deal_seed, layout_epoch): the deal's key comes from
    there instead of (seed, epoch) — the conveyor layout, whose ranks share the deal but not the draws' seed.  bins = (lo, hi): the
    draws of the bins [lo, hi) only."""
    t = tables if tables is not None else ldsbin_tables(indptr, indices, n_items, n_bins, hot_x1000, share)
    key = int(lib().oracle_ldsbin_key(int(seed), int(epoch)) if deal is None else lib().oracle_ldsbin_key(int(deal[0]), int(deal[1])))
    draws = C.c_int64()
    pos = np.zeros(n_items, np.int64) if count_touches else None
    neg = np.zeros(n_items, np.int64) if count_touches else None
    lo, hi = (0, int(n_bins)) if bins is None else (int(bins[0]), int(bins[1]))
    s = lib().oracle_ldsbin_epoch_skips_range(int(seed), int(epoch), key, int(n_bins), int(n_items), int(t["n_hot"]),
                                              int(ldsbin_n_strata(n_items, n_bins, strata_groups)), int(hot_cost_x16),
                                              t["rank_item"], t["cptr"], t["cusers"], t["hot_u"], t["hot_i"],
                                              int(t["n_hot_inter"]), t["indptr"], t["indices"], C.byref(draws),
                                              pos.ctypes.data if count_touches else None, neg.ctypes.data if count_touches else None,
                                              int(bool(neg_pop)), lo, hi)
    return (int(s), int(draws.value), t["n_hot"]) + ((pos, neg) if count_touches else ())


def hogwild_sample_owned(seed, epoch, wave_id, length, n_neg, lo, hi):
    r = np.empty(hi - lo, np.int64)
    jj = np.empty(hi - lo, np.int64)
    lib().oracle_hogwild_sample_owned(int(seed), int(epoch), int(wave_id), int(length), int(n_neg), int(lo), int(hi),
                                      r, jj)
    return r, jj
