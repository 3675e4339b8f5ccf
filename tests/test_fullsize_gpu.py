"""Parity and properties at BASELINE.json's full size (configs[1]: ML-20M shape, k = 64)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from cornac_amd import _lib

pytestmark = pytest.mark.gpu

sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def ml20m():
    from bench import init_factors, load_dataset

    n_users, n_items, indptr, indices = load_dataset("ml20m", 0, os.environ.get("TMPDIR", "/tmp"))
    return n_users, n_items, indptr, indices, init_factors


def test_deterministic_full_epoch_matches_sequential_oracle(oracle, ml20m):
    """SURVEY §8d parity case: seed 7, one full epoch (20 000 263 sequential updates).  The oracle
    runs the single-thread reference loop (~12 s); the device result must agree to 1e-4 (north_star
    tolerance; measured agreement is ~1e-7) and the (correct, skipped) counters exactly."""
    import ctypes as C

    n_users, n_items, indptr, indices, init_factors = ml20m
    k, lr, reg = 64, 0.05, 0.01
    rng = np.random.RandomState(7)
    U = ((rng.uniform(0, 1, (n_users, k)).astype(np.float32) - 0.5) / k)
    V = ((rng.uniform(0, 1, (n_items, k)).astype(np.float32) - 0.5) / k)
    B = np.zeros(n_items, np.float32)
    seed_pos = oracle.rngvector_seed(rng.randint(2 ** 31))
    seed_neg = oracle.rngvector_seed(rng.randint(2 ** 31))
    tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
    tr.set_factors(U, V, B)
    tr.seed_mt19937(seed_pos, seed_neg)
    c, s = tr.fit_epochs(1, lr, reg, True, _lib.NEG_UNIFORM, _lib.MODE_DETERMINISTIC)
    timing = tr.last_timing()
    Ud, Vd, Bd = tr.get_factors()
    tr.close()
    # oracle
    L = oracle.lib()
    user_ids = np.repeat(np.arange(n_users), np.diff(indptr)).astype(np.int32)
    gp, gn = oracle.MT19937(seed_pos), oracle.MT19937(seed_neg)
    oc, os_ = C.c_int64(), C.c_int64()
    nnz = len(indices)
    rc = L.oracle_bpr_epoch_seq(gp.ptr, gn.ptr, nnz - 1, n_items - 1, nnz, user_ids, indices,
                                np.arange(n_items, dtype=np.int32), indptr, U, V, B, k, lr, reg, 1, C.byref(oc),
                                C.byref(os_), None, None, None)
    assert rc == 0
    assert (c, s) == (oc.value, os_.value)
    err = max(np.abs(Ud - U).max(), np.abs(Vd - V).max(), np.abs(Bd - B).max())
    print("full-size deterministic epoch: max |err| = %.3g, timing %s" % (err, timing))
    assert err <= 1e-4
    assert np.mean(Ud == U) > 0.99, "expected (almost) bit-identical user factors"


@pytest.mark.parametrize("form", ["ldsbin", "fused"])
def test_hogwild_full_size_invariants(ml20m, form):
    """Size-independent properties of the throughput kernels at full size: with reg = 0 every triplet's item-row deltas
    cancel (dV_i = -dV_j, dB_i = -dB_j), so the column sums of V and the sum of B are conserved by exact updates —
    the LDS-bin form (the default at this shape: LDS read-modify-writes under row locks) and the fused kernel (fp32
    atomics) both apply every update exactly once; the counters cover exactly nnz draws per epoch."""
    n_users, n_items, indptr, indices, init_factors = ml20m
    k = 64
    U, V, B = init_factors(n_users, n_items, k, 3)
    tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
    if form == "ldsbin":
        assert tr.ldsbin_stats()["bins"] == 256, "the default form at the ML-20M shape must be the LDS-bin kernel"
    tr.set_factors(U, V, B)
    tr.seed_hogwild(2024)
    c, s = tr.fit_epochs(2, 0.05, 0.0, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD, flags=0 if form == "ldsbin" else _lib.FORM_FUSED)
    U2, V2, B2 = tr.get_factors()
    tr.close()
    nnz = len(indices)
    assert 0 < s < 0.03 * 2 * nnz and 0 < c <= 2 * nnz - s
    assert np.isfinite(U2).all() and np.isfinite(V2).all()
    assert np.abs(V2 - V).max() > 1e-3, "the model must have trained"
    col0, col1 = V.astype(np.float64).sum(0), V2.astype(np.float64).sum(0)
    moved = np.abs(V2.astype(np.float64) - V).sum(0)  # total |update| mass per column
    assert np.abs(col1 - col0).max() <= 1e-5 * moved.max() + 1e-3, (np.abs(col1 - col0).max(), moved.max())
    assert abs(float(B2.astype(np.float64).sum())) <= 1e-5 * np.abs(B2).sum() + 1e-3


def test_rank_full_size_fused_equals_materialised(oracle, ml20m):
    """fused MFMA top-10 over all 26 744 items == the materialised-score path (score_block +
    oracle ranking) for a sample of users; results sorted; exclusions honoured."""
    n_users, n_items, indptr, indices, init_factors = ml20m
    rs = np.random.RandomState(5)
    U = rs.normal(0, 0.2, (n_users, 64)).astype(np.float32)
    V = rs.normal(0, 0.2, (n_items, 64)).astype(np.float32)
    Bi = rs.normal(0, 0.3, n_items).astype(np.float32)
    sc = _lib.Scorer(U, V, Bi, None)
    users = rs.choice(n_users, 300, replace=False).astype(np.int32)
    excl_ptr = np.concatenate([[0], np.cumsum(indptr[users + 1] - indptr[users])]).astype(np.int64)
    excl_idx = np.concatenate([indices[indptr[u]:indptr[u + 1]] for u in users]).astype(np.int32)
    items, scores = sc.rank_topk(users, 10)
    items_x, scores_x = sc.rank_topk(users, 10, exclude=(excl_ptr, excl_idx))
    full = sc.score_block(users)
    assert np.array_equal(full[:7], oracle.score_block(U, V, Bi, None, users[:7]))
    for b, u in enumerate(users):
        want, _ = oracle.rank(full[b], n_items, n_items, k=10)
        assert np.array_equal(items[b], want)
        assert np.array_equal(scores[b], full[b][want])
        seen = indices[indptr[u]:indptr[u + 1]]
        cand = np.setdiff1d(np.arange(n_items), seen)
        want_x, _ = oracle.rank(full[b], n_items, n_items, item_indices=cand, k=10)
        assert np.array_equal(items_x[b], want_x)
        assert not np.intersect1d(items_x[b], seen).size
    assert (np.diff(scores, axis=1) <= 0).all()
    sc.close()


def test_hogwild_full_size_statistical_parity_with_cpu_threads(oracle, ml20m):
    """Throughput mode at BASELINE size vs the reference's OpenMP Hogwild path (CPU port, all host
    threads): same data, init, hyper-parameters and epochs -> same pairwise loss / accuracy on a fixed
    sample of (u, i, j) and the same per-epoch 'correct' fraction, within noise."""
    n_users, n_items, indptr, indices, init_factors = ml20m
    k, lr, reg, epochs = 64, 0.05, 0.01, 4
    nnz = len(indices)
    user_ids = np.repeat(np.arange(n_users), np.diff(indptr)).astype(np.int32)

    def sample_loss(U, V, B, n=200000):
        rs = np.random.RandomState(1)
        pick = rs.randint(nnz, size=n)
        u, i = user_ids[pick], indices[pick]
        j = rs.randint(n_items, size=n)
        x = B[i] - B[j] + np.einsum("nk,nk->n", U[u], V[i] - V[j])
        return float(np.mean(np.log1p(np.exp(-x)))), float(np.mean(x > 0))

    U, V, B = init_factors(n_users, n_items, k, 3)
    l0, _ = sample_loss(U, V, B)
    tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
    tr.set_factors(U, V, B)
    tr.seed_hogwild(11)
    tr.fit_epochs(epochs - 1, lr, reg, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
    c_gpu, s_gpu = tr.fit_epochs(1, lr, reg, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
    Ug, Vg, Bg = tr.get_factors()
    tr.close()
    Uc, Vc, Bc = init_factors(n_users, n_items, k, 3)
    threads = min(32, oracle.lib().oracle_num_threads())
    oracle.bpr_hogwild_epochs(indptr, indices, user_ids, n_items, Uc, Vc, Bc, k, lr, reg, True, 5, threads, epochs - 1)
    c_cpu, s_cpu = oracle.bpr_hogwild_epochs(indptr, indices, user_ids, n_items, Uc, Vc, Bc, k, lr, reg, True, 6,
                                             threads, 1)
    lg, ag = sample_loss(Ug, Vg, Bg)
    lc, ac = sample_loss(Uc, Vc, Bc)
    print("full-size hogwild parity: loss0 %.4f gpu %.4f cpu %.4f | acc gpu %.4f cpu %.4f | correct gpu %.4f cpu %.4f"
          % (l0, lg, lc, ag, ac, c_gpu / (nnz - s_gpu), c_cpu / (nnz - s_cpu)))
    assert lc < 0.97 * l0, "the task must be learnable for the gate to mean anything"
    assert abs(lg - lc) < 0.03 * l0 and abs(ag - ac) < 0.015
    assert abs(c_gpu / (nnz - s_gpu) - c_cpu / (nnz - s_cpu)) < 0.015
    assert abs(s_gpu - s_cpu) < 0.02 * s_cpu


def test_hogwild_full_size_gate_against_the_reference_threads(ml20m):
    """The default throughput form (LDS-bin) at BASELINE size against the REAL reference's threads: the compiled
    `BPR._fit_sgd` of oracle/_ref (cornac/models/bpr/recom_bpr.pyx:231-267, OpenMP, 32 threads) driven with raw arrays as
    `BPR.fit` drives it (recom_bpr.pyx:186-201).  Same data, init, hyper-parameters and epochs.  Two probes on fixed
    samples: j uniform over all items (the reference's own negative population), and j among the 2 x 128
    popularity-rank neighbours of i — the (positive, negative) pairs round 3's static rank groups could never draw; the
    binned sampler has to have learned them as well as the reference's global draw did."""
    from oracle import ref_loader

    if not (ref_loader.available() or ref_loader.kernel_available()):
        pytest.skip("oracle/_ref (the compiled reference kernel) is not built")
    RNGVector, RefBPR = ref_loader.load_kernel_only()
    n_users, n_items, indptr, indices, init_factors = ml20m
    k, lr, reg, epochs = 64, 0.05, 0.01, 12  # (the rank-neighbour probe starts to move after ~8 epochs)
    nnz = len(indices)
    user_ids = np.repeat(np.arange(n_users), np.diff(indptr)).astype(np.int32)
    deg = np.bincount(indices, minlength=n_items)
    rank_item = np.argsort(-deg, kind="stable")
    item_rank = np.empty(n_items, np.int64)
    item_rank[rank_item] = np.arange(n_items)
    rs = np.random.RandomState(1)
    n = 300000
    pick = rs.randint(nnz, size=n)
    pu, pi = user_ids[pick], indices[pick]
    pj_all = rs.randint(n_items, size=n)
    delta = rs.randint(1, 129, size=n) * rs.choice([-1, 1], size=n)
    pj_near = rank_item[np.clip(item_rank[pi] + delta, 0, n_items - 1)]
    allk = np.sort(user_ids.astype(np.int64) * n_items + indices)
    near_ok = (pj_near != pi) & ~np.isin(pu.astype(np.int64) * n_items + pj_near, allk)

    def probes(U, V, B):
        out = []
        for j, ok in ((pj_all, None), (pj_near, near_ok)):
            x = B[pi] - B[j] + np.einsum("nk,nk->n", U[pu], V[pi] - V[j])
            x = x if ok is None else x[ok]
            out += [float(np.mean(np.log1p(np.exp(-x)))), float(np.mean(x > 0))]
        return out

    U, V, B = init_factors(n_users, n_items, k, 3)
    l0 = probes(U, V, B)
    tr = _lib.BprTrainer(indptr, indices, n_users, n_items, n_users, n_items, k)
    assert tr.ldsbin_stats()["bins"] == 256
    tr.set_factors(U, V, B)
    tr.seed_hogwild(11)
    tr.fit_epochs(epochs - 1, lr, reg, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
    c_gpu, s_gpu = tr.fit_epochs(1, lr, reg, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
    g = probes(*tr.get_factors())
    tr.close()
    (Ur, Vr, Br), (c_ref, s_ref), threads = _reference_threads_fit(ml20m, k, lr, reg, epochs)
    r = probes(Ur, Vr, Br)
    print("full-size gate vs the reference's %d threads after %d epochs: all-items probe loss %.4f -> gpu %.4f ref %.4f, acc "
          "gpu %.4f ref %.4f | rank-neighbour probe loss %.4f -> gpu %.4f ref %.4f, acc gpu %.4f ref %.4f | correct gpu %.4f "
          "ref %.4f | skipped gpu %d ref %d" % (threads, epochs, l0[0], g[0], r[0], g[1], r[1], l0[2], g[2], r[2], g[3], r[3],
                                                 c_gpu / (nnz - s_gpu), c_ref / (nnz - s_ref), s_gpu, s_ref))
    assert r[0] < 0.97 * l0[0] and r[2] < 0.98 * l0[2], "the task must be learnable on both probes for the gate to mean anything"
    assert abs(g[0] - r[0]) < 0.03 * l0[0] and abs(g[1] - r[1]) < 0.015
    assert abs(g[2] - r[2]) < 0.03 * l0[2] and abs(g[3] - r[3]) < 0.015
    assert abs(c_gpu / (nnz - s_gpu) - c_ref / (nnz - s_ref)) < 0.015
    assert abs(s_gpu - s_ref) < 0.02 * s_ref


_REF_FITS = {}


def _reference_threads_fit(ml20m, k, lr, reg, epochs):
    """the REAL reference's `BPR._fit_sgd` (oracle/_ref, OpenMP threads) for `epochs` epochs from init_factors(seed 3), driven with
    raw arrays as `BPR.fit` drives it (recom_bpr.pyx:186-201); cached per module (two gates compare against the same fit)"""
    from oracle import ref_loader

    key = (k, lr, reg, epochs)
    if key not in _REF_FITS:
        RNGVector, RefBPR = ref_loader.load_kernel_only()
        n_users, n_items, indptr, indices, init_factors = ml20m
        nnz = len(indices)
        user_ids = np.repeat(np.arange(n_users), np.diff(indptr)).astype(np.int32)
        Ur, Vr, Br = init_factors(n_users, n_items, k, 3)
        model = RefBPR(k=k, learning_rate=lr, lambda_reg=reg)
        threads = min(32, os.cpu_count() or 1)
        neg_item_ids = np.arange(n_items, dtype=np.int32)
        ip, ix = np.ascontiguousarray(indptr, np.int32), np.ascontiguousarray(indices, np.int32)
        for e in range(epochs):
            c_ref, s_ref = model._fit_sgd(RNGVector(threads, nnz - 1, 100 + e), RNGVector(threads, n_items - 1, 200 + e), threads,
                                          user_ids, ix, neg_item_ids, ip, Ur, Vr, Br)
        _REF_FITS[key] = ((Ur, Vr, Br), (c_ref, s_ref), threads)
    return _REF_FITS[key]


def _gate_probes(ml20m, n=300000):
    """(probes(U, V, B) -> [loss_all, acc_all, loss_near, acc_near]): j uniform over all items (the reference's own negative
    population), and j among the 2 x 128 popularity-rank neighbours of i"""
    n_users, n_items, indptr, indices, _ = ml20m
    nnz = len(indices)
    user_ids = np.repeat(np.arange(n_users), np.diff(indptr)).astype(np.int32)
    deg = np.bincount(indices, minlength=n_items)
    rank_item = np.argsort(-deg, kind="stable")
    item_rank = np.empty(n_items, np.int64)
    item_rank[rank_item] = np.arange(n_items)
    rs = np.random.RandomState(1)
    pick = rs.randint(nnz, size=n)
    pu, pi = user_ids[pick], indices[pick]
    pj_all = rs.randint(n_items, size=n)
    delta = rs.randint(1, 129, size=n) * rs.choice([-1, 1], size=n)
    pj_near = rank_item[np.clip(item_rank[pi] + delta, 0, n_items - 1)]
    allk = np.sort(user_ids.astype(np.int64) * n_items + indices)
    near_ok = (pj_near != pi) & ~np.isin(pu.astype(np.int64) * n_items + pj_near, allk)

    def probes(U, V, B):
        out = []
        for j, ok in ((pj_all, None), (pj_near, near_ok)):
            x = B[pi] - B[j] + np.einsum("nk,nk->n", U[pu], V[pi] - V[j])
            x = x if ok is None else x[ok]
            out += [float(np.mean(np.log1p(np.exp(-x)))), float(np.mean(x > 0))]
        return out

    return probes


@pytest.mark.parametrize("rings", [1, 4])
def test_conveyor_of_eight_virtual_ranks_at_the_ml20m_shape_against_the_reference_threads(ml20m, rings):
    """Multi-GPU regime 2 (cornac_amd/dist.py BinConveyorBprTrainer) at the ML-20M shape and skew, EIGHT virtual ranks on one
    device — users cut into eight ranges of equal interaction counts, one handle per rank in conveyor layout, 16 blocks (one
    ring) or 64 (four rings: a step of a rank = one launch over four bin ranges), the blocks' rows re-dealt to the next epoch's
    slots between the epochs; the schedule executed step by step, which the gloo tests show the ranks' parallel run to equal
    bit for bit — against the REAL reference's threads on the same data, init, hyper-parameters and epochs.  Same probes and
    tolerances as the single-GPU gate above: j over ALL items (recom_bpr.pyx:235-238 — round 5's conveyor fixed item i to
    block i % 2NK and could never draw 15 / 16 resp. 63 / 64 of these pairs) and j among i's popularity-rank neighbours."""
    import torch

    from cornac_amd.dist import partition_users_by_nnz, ring_strides, slice_csr
    from oracle import ref_loader

    if not (ref_loader.available() or ref_loader.kernel_available()):
        pytest.skip("oracle/_ref (the compiled reference kernel) is not built")
    n_users, n_items, indptr, indices, init_factors = ml20m
    k, lr, reg, epochs, R = 64, 0.05, 0.01, 12, 8
    nnz = len(indices)
    probes = _gate_probes(ml20m)
    U, V, B = init_factors(n_users, n_items, k, 3)
    l0 = probes(U, V, B)
    dev = torch.device("cuda", 0)
    bounds = partition_users_by_nnz(indptr, R)
    strides = ring_strides(R, rings)
    K, nb = len(strides), 2 * R
    order = np.argsort(-np.bincount(indices, minlength=n_items), kind="stable").astype(np.int32)
    Ut = torch.as_tensor(U).to(dev)
    handles, dims = [], None
    for r in range(R):
        u0, u1 = int(bounds[r]), int(bounds[r + 1])
        ip, ix = slice_csr(indptr, indices, u0, u1)
        t = _lib.BprTrainer(ip, ix, u1 - u0, n_items, u1 - u0, n_items, k)
        t.bind_device(Ut[u0:u1].data_ptr(), None, None)
        t.seed_hogwild(9000 + r)
        d = t.conveyor_setup(nb * K, order, 777)
        assert dims is None or d == dims
        dims = d
        handles.append(t)
    n_bins, bpb, cap = dims
    W = bpb * cap
    table = torch.zeros((nb * K, W * (k + 1)), dtype=torch.float32, device=dev)

    def layout(e):
        si = torch.empty(n_bins * cap, dtype=torch.int32, device=dev)
        handles[0].conveyor_layout(e, si.data_ptr(), None)
        handles[0].sync()
        return si.long()

    def scatter(Vd, Bd, si):
        ok = si >= 0
        rows, bias = torch.zeros((nb * K * W, k), device=dev), torch.zeros(nb * K * W, device=dev)
        rows[ok], bias[ok] = Vd[si[ok]], Bd[si[ok]]
        table[:, : W * k] = rows.view(nb * K, W * k)
        table[:, W * k:] = bias.view(nb * K, W)

    def collect(si):
        ok = si >= 0
        Vd, Bd = torch.zeros((n_items, k), device=dev), torch.zeros(n_items, device=dev)
        Vd[si[ok]] = table[:, : W * k].reshape(nb * K * W, k)[ok]
        Bd[si[ok]] = table[:, W * k:].reshape(nb * K * W)[ok]
        return Vd, Bd

    si = layout(0)
    scatter(torch.as_tensor(V).to(dev), torch.as_tensor(B).to(dev), si)
    torch.cuda.synchronize()
    c_gpu = s_gpu = 0
    for e in range(epochs):
        if e:
            Vd, Bd = collect(si)
            si = layout(e)
            scatter(Vd, Bd, si)
            torch.cuda.synchronize()
        for t_ in handles:
            t_.sync()
        for step in range(nb):
            for r in range(R):
                blocks = [((2 * ((r * pow(s_, -1, R)) % R) + step) % nb) * K + g for g, s_ in enumerate(strides)]
                handles[r].conveyor_enqueue(e, e, blocks, [table[b].data_ptr() for b in blocks], lr, reg, True, _lib.NEG_UNIFORM, 0)
            for r in range(R):      # the ranks of a step touch disjoint user rows and disjoint blocks: they may overlap
                c, s_k = handles[r].sync()
                if e == epochs - 1:
                    c_gpu, s_gpu = c_gpu + c, s_gpu + s_k
    Vd, Bd = collect(si)
    assert sum(t.ldsbin_stats()["lock_timeouts"] for t in handles) == 0
    for t in handles:
        t.close()
    g = probes(Ut.cpu().numpy(), Vd.cpu().numpy(), Bd.cpu().numpy())
    (Ur, Vr, Br), (c_ref, s_ref), threads = _reference_threads_fit(ml20m, k, lr, reg, epochs)
    r = probes(Ur, Vr, Br)
    print("conveyor of 8 virtual ranks, %d blocks of %d bins (%d rows per bin), vs the reference's %d threads after %d epochs: "
          "all-items probe loss %.4f -> conveyor %.4f ref %.4f, acc %.4f ref %.4f | rank-neighbour probe loss %.4f -> %.4f ref "
          "%.4f, acc %.4f ref %.4f | correct %.4f ref %.4f | skipped %d ref %d"
          % (nb * K, bpb, cap, threads, epochs, l0[0], g[0], r[0], g[1], r[1], l0[2], g[2], r[2], g[3], r[3],
             c_gpu / (nnz - s_gpu), c_ref / (nnz - s_ref), s_gpu, s_ref))
    assert r[0] < 0.97 * l0[0] and r[2] < 0.98 * l0[2], "the task must be learnable on both probes for the gate to mean anything"
    assert abs(g[0] - r[0]) < 0.03 * l0[0] and abs(g[1] - r[1]) < 0.015
    assert abs(g[2] - r[2]) < 0.03 * l0[2] and abs(g[3] - r[3]) < 0.015
    assert abs(c_gpu / (nnz - s_gpu) - c_ref / (nnz - s_ref)) < 0.015


def test_mf_full_size_deterministic_and_hogwild(oracle, ml20m):
    """BASELINE configs[2] family at ML-20M size (20 M ratings, k = 128): one deterministic epoch against the
    sequential oracle (the reference's seeded loop), then the hogwild kernel's loss against the same oracle."""
    n_users, n_items, indptr, indices, _ = ml20m
    k, lr, reg = 128, 0.01, 0.02
    rs = np.random.RandomState(11)
    rid = np.repeat(np.arange(n_users, dtype=np.int64), np.diff(indptr))
    cid = indices.astype(np.int64)
    order = rs.permutation(len(cid))  # insertion order != CSR order, like a real uir_tuple
    rid, cid = rid[order], cid[order]
    bu, bi = rs.normal(0, 0.5, n_users), rs.normal(0, 0.5, n_items)
    val = np.clip(np.rint(3.5 + bu[rid] + bi[cid] + rs.normal(0, 0.7, len(rid))), 1, 5).astype(np.float32)
    mu = np.float32(val.mean())
    U0 = rs.normal(0, 0.01, (n_users, k)).astype(np.float32)
    V0 = rs.normal(0, 0.01, (n_items, k)).astype(np.float32)
    zu, zi = np.zeros(n_users, np.float32), np.zeros(n_items, np.float32)
    Uo, Vo, Buo, Bio = U0.copy(), V0.copy(), zu.copy(), zi.copy()
    loss_o = np.zeros(1, np.float32)
    n_run = oracle.lib().oracle_mf_fit(rid, cid, val, len(val), Uo, Vo, Buo, Bio, k, lr, reg, float(mu), 1, 1, 1, 0,
                                       loss_o.ctypes.data)
    assert n_run == 1
    tr = _lib.MfTrainer(rid, cid, val, n_users, n_items, k)
    tr.set_factors(U0, V0, zu, zi)
    loss_d, n = tr.fit(1, lr, reg, float(mu), True, False, _lib.MODE_DETERMINISTIC)
    Ud, Vd, Bud, Bid = tr.get_factors()
    err = max(np.abs(Ud - Uo).max(), np.abs(Vd - Vo).max(), np.abs(Bud - Buo).max(), np.abs(Bid - Bio).max())
    print("full-size MF deterministic epoch: max |err| = %.3g" % err)
    assert err <= 1e-4 and np.mean(Ud == Uo) > 0.99
    # the reference accumulates 20 M squared errors sequentially in float32 (backend_cpu.pyx:61,85): at this size
    # its own loss value carries several percent of rounding error; the device sums in fp64
    assert abs(loss_d[0] - loss_o[0]) <= 0.1 * loss_o[0]
    tr.set_factors(U0, V0, zu, zi)
    loss_h, _ = tr.fit(1, lr, reg, float(mu), True, False, _lib.MODE_HOGWILD)
    tr.close()
    assert abs(loss_h[0] - loss_d[0]) <= 0.02 * loss_d[0], (loss_h, loss_d)


@pytest.mark.timeout(900)
def test_mf_netflix_shape_deterministic_and_hogwild(oracle):
    """BASELINE configs[2] at its TRUE shape: 480 189 users x 17 770 items, 100 480 507 ratings (int64 COO, the
    reference's uir_tuple dtypes), k = 128.  One deterministic epoch over ALL ratings against the sequential oracle
    (the reference's seeded loop, backend_cpu.pyx:35-97) with both clocks in the log, a chain-like stored order on a
    subsample, then the hogwild kernel's epoch loss against a float64 evaluation."""
    from bench import synth_ratings
    from cornac_amd import synth

    n_users, n_items, nnz, zipf_a, seed = synth.CONFIGS["netflix"]
    rid, cid, val = synth_ratings(n_users, n_items, nnz, zipf_a, seed)
    assert len(val) == 100_480_507 and rid.dtype == np.int64 and cid.dtype == np.int64
    k, lr, reg = 128, 0.01, 0.02
    mu = np.float32(val.mean(dtype=np.float64))
    rs = np.random.RandomState(11)
    U0 = rs.normal(0, 0.01, (n_users, k)).astype(np.float32)
    V0 = rs.normal(0, 0.01, (n_items, k)).astype(np.float32)
    zu, zi = np.zeros(n_users, np.float32), np.zeros(n_items, np.float32)
    import time

    def det_epoch(r_, c_, v_):
        """(device result, device seconds, oracle result, oracle seconds) of one sequential epoch"""
        Uo, Vo, Buo, Bio = U0.copy(), V0.copy(), zu.copy(), zi.copy()
        loss_o = np.zeros(1, np.float32)
        t0 = time.perf_counter()
        assert oracle.lib().oracle_mf_fit(r_, c_, v_, len(v_), Uo, Vo, Buo, Bio, k, lr, reg, float(mu), 1, 1, 1, 0,
                                          loss_o.ctypes.data) == 1
        t_cpu = time.perf_counter() - t0
        tr = _lib.MfTrainer(r_, c_, v_, n_users, n_items, k)
        tr.set_factors(U0, V0, zu, zi)
        tr.fit(1, lr, reg, float(mu), True, False, _lib.MODE_DETERMINISTIC)
        timing = tr.last_timing()
        got = tr.get_factors()
        tr.close()
        return got, timing, (Uo, Vo, Buo, Bio), t_cpu

    # (1) ALL 100 480 507 ratings in the order of the Netflix Prize files: one block of ratings per movie (item-major;
    # the users inside a block in generation order).  The deterministic mode runs as ONE persistent dataflow launch
    # (mf_det_chain_kernel: the waves own the items, user rows are handed over through per-row update counters); the
    # reference's own sequential loop (the oracle, one host thread) is timed beside it.
    order = np.argsort(cid, kind="stable")
    rid_i, cid_i, val_i = (np.ascontiguousarray(x[order]) for x in (rid, cid, val))
    got, timing, want, t_cpu = det_epoch(rid_i, cid_i, val_i)
    err = max(np.abs(g - w).max() for g, w in zip(got, want))
    print("Netflix-shape MF deterministic epoch, item-major order, %d ratings: max |err| = %.3g; device %s; the sequential "
          "oracle on one host thread: %.1f s" % (len(val_i), err, timing, t_cpu))
    assert err <= 1e-4 and np.mean(got[0] == want[0]) > 0.99
    assert timing["sgd_ms"] / 1e3 < t_cpu, "the deterministic epoch must not be slower than the reference's sequential loop"
    Vd = got[1]
    del rid_i, cid_i, val_i, order
    # (2) a stored order whose dependency DAG is almost a single chain — user-major with the items of a user in random
    # order, blocks of the list shuffled (61 M links over 100 M ratings: an early rating of one user waits for a late
    # rating of the previous one; no schedule can run it in parallel) — on every 10th rating: correctness of the waiting
    # protocol where nearly every rating waits
    blocks = rs.permutation(1024)
    cuts = np.linspace(0, nnz, 1025).astype(np.int64)
    order = np.concatenate([np.arange(cuts[b], cuts[b + 1]) for b in blocks])
    rid, cid, val = rid[order], cid[order], val[order]
    got, timing, want, t_cpu = det_epoch(*(np.ascontiguousarray(x[::10]) for x in (rid, cid, val)))
    err = max(np.abs(g - w).max() for g, w in zip(got, want))
    print("Netflix-shape MF deterministic epoch, chain-like order, %d ratings: max |err| = %.3g; device %s; oracle %.1f s"
          % (len(val) // 10 + 1, err, timing, t_cpu))
    assert err <= 1e-4
    # the hogwild kernel over ALL 100 480 507 ratings; its epoch loss against a float64 evaluation of the same
    # predictions from the start tables (the reference's own float32 running sum loses 30 % of its value at 1e8 terms)
    tr = _lib.MfTrainer(rid, cid, val, n_users, n_items, k)
    tr.set_factors(U0, V0, zu, zi)
    loss_h, _ = tr.fit(1, lr, reg, float(mu), True, False, _lib.MODE_HOGWILD)
    Uh, Vh, Buh, Bih = tr.get_factors()
    loss_h2, _ = tr.fit(1, lr, reg, float(mu), True, False, _lib.MODE_HOGWILD)
    tr.close()
    assert np.isfinite(Uh).all() and np.isfinite(Vh).all()
    sse0 = 0.0
    for a in range(0, nnz, 1 << 22):   # loss of the untrained model: an upper bound the first epoch must beat
        b = min(a + (1 << 22), nnz)
        p = mu + np.einsum("nk,nk->n", U0[rid[a:b]], V0[cid[a:b]])
        sse0 += float(np.sum((val[a:b].astype(np.float64) - p) ** 2))
    # (the reported loss is 0.5 x the sum of squared errors seen during the epoch, backend_cpu.pyx:88)
    assert 0.05 * sse0 < loss_h[0] < 0.5 * sse0 and loss_h2[0] < loss_h[0], (sse0, loss_h, loss_h2)
    # same optimisation problem as the sequential pass on the subsample: item rows moved in the same direction
    dv_h, dv_d = (Vh - V0).ravel(), (Vd - V0).ravel()  # (Vd: the sequential pass over all ratings in item-major order)
    assert float(dv_h @ dv_d) / (np.linalg.norm(dv_h) * np.linalg.norm(dv_d)) > 0.5


@pytest.mark.timeout(900)
@pytest.mark.parametrize("zipf", [None, 0.8])
def test_mf_netflix_shape_hogwild_gate_against_the_reference_threads(zipf):
    """The throughput form of MF (block rotation: exact row updates, its own visiting order) against the REAL reference's
    racy threads at the Netflix shape: 3 epochs of `backend_cpu.fit_sgd(num_threads=32)` (backend_cpu.pyx:62-88, the
    compiled extension in oracle/_ref; the oracle's C port with the same thread count where that is absent) and 3 hogwild
    epochs on the device, the SAME 100 480 507-rating COO, start tables and hyper-parameters.  Both are order-dependent
    stochastic runs of one optimisation, so the gate is on what they optimise: the mean squared error of the trained
    model over a fixed 10 M-rating sample of the training set, evaluated in float64 — within 0.25 % of each other (measured:
    0.52901 against 0.52899, 0.004 %) — and the learned item side pointing the same way (cosines 0.983 / 0.988).
    zipf = 0.8: SURVEY 8d's own generator — one title holds 3.2 % of the ratings; its row is split into virtual rows merged
    after every phase (csrc/mf_blocks.inc) and the same gate holds."""
    import time

    from bench import synth_ratings
    from cornac_amd import synth

    n_users, n_items, nnz, zipf_a, seed = synth.CONFIGS["netflix"]
    zipf_a = zipf_a if zipf is None else zipf
    rid, cid, val = synth_ratings(n_users, n_items, nnz, zipf_a, seed)
    k, lr, reg, epochs, threads = 128, 0.01, 0.02, 3, 32
    mu = float(val.mean(dtype=np.float64))
    rs = np.random.RandomState(11)
    U0 = rs.normal(0, 0.01, (n_users, k)).astype(np.float32)
    V0 = rs.normal(0, 0.01, (n_items, k)).astype(np.float32)
    sample = np.sort(rs.choice(nnz, 10_000_000, replace=False))
    rs_, cs_, vs_ = rid[sample], cid[sample], val[sample].astype(np.float64)

    def mse(U, V, Bu, Bi):
        tot = 0.0
        for a in range(0, len(vs_), 1 << 21):
            b = min(a + (1 << 21), len(vs_))
            p = mu + Bu[rs_[a:b]].astype(np.float64) + Bi[cs_[a:b]] + np.einsum("nk,nk->n", U[rs_[a:b]], V[cs_[a:b]], dtype=np.float64)
            tot += float(np.sum((vs_[a:b] - p) ** 2))
        return tot / len(vs_)

    Ur, Vr = U0.copy(), V0.copy()
    Bur, Bir = np.zeros(n_users, np.float32), np.zeros(n_items, np.float32)
    t0 = time.perf_counter()
    try:
        from oracle import ref_loader

        ref_loader.load_mf_kernel()(rid, cid, val, Ur, Vr, Bur, Bir, lr, reg, mu, epochs, threads, True, False, False)
        who = "the reference's compiled fit_sgd"
    except Exception as e:  # oracle/_ref not built on this box: the C port's threads
        from oracle import oracle as orc

        loss = np.zeros(epochs, np.float32)
        orc.lib().oracle_mf_fit(rid, cid, val, nnz, Ur, Vr, Bur, Bir, k, lr, reg, mu, epochs, threads, 1, 0, loss.ctypes.data)
        who = "the oracle's C port (%r)" % (e,)
    t_cpu = time.perf_counter() - t0
    tr = _lib.MfTrainer(rid, cid, val, n_users, n_items, k)
    tr.set_factors(U0, V0, np.zeros(n_users, np.float32), np.zeros(n_items, np.float32))
    t0 = time.perf_counter()
    tr.fit(epochs, lr, reg, mu, True, False, _lib.MODE_HOGWILD)
    t_dev = time.perf_counter() - t0
    Uh, Vh, Buh, Bih = tr.get_factors()
    tr.close()
    m0 = mse(U0, V0, np.zeros(n_users, np.float32), np.zeros(n_items, np.float32))
    m_ref, m_dev = mse(Ur, Vr, Bur, Bir), mse(Uh, Vh, Buh, Bih)
    dv_h, dv_r = (Vh - V0).ravel().astype(np.float64), (Vr - V0).ravel().astype(np.float64)
    cos = float(dv_h @ dv_r) / (np.linalg.norm(dv_h) * np.linalg.norm(dv_r))
    cos_b = float((Bih.astype(np.float64) @ Bir) / (np.linalg.norm(Bih) * np.linalg.norm(Bir)))
    print("Netflix-shape (Zipf %.2f) hogwild MF, %d epochs: training MSE on a 10 M sample %.5f untrained -> %s, %d threads: %.5f (%.1f s); "
          "device: %.5f (%.2f s); relative difference %.3f %%; cosine of the item-factor moves %.3f, of the item biases %.3f"
          % (zipf_a, epochs, m0, who, threads, m_ref, t_cpu, m_dev, t_dev, 100 * abs(m_dev - m_ref) / m_ref, cos, cos_b))
    assert np.isfinite(Uh).all() and np.isfinite(Vh).all()
    assert m_ref < 0.8 * m0, "the reference run itself must have learned something"
    assert abs(m_dev - m_ref) <= 0.0025 * m_ref, (m_dev, m_ref)
    assert cos_b > 0.97 and cos > 0.9, (cos, cos_b)


@pytest.mark.timeout(900)
def test_vbpr_tradesy_shape_matches_the_torch_oracle():
    """configs[3] at its real size (19 243 users x 165 906 items, 4096-d features, k = k2 = 64, batch 100): 40 minibatch
    steps on the device — stamped batch rows, gradient-free dense sweep on the second stream, E / beta' step — against
    torch autograd + torch.optim.Adam over the same pre-sampled batches (oracle/vbpr_oracle.py, bit-identical to the
    live reference at toy size): every table within 1e-4, the tolerance north_star states for learned factors."""
    from oracle import vbpr_oracle

    nu, ni, nf, k, k2, B, steps = 19243, 165906, 4096, 64, 64, 100, 40
    rs = np.random.RandomState(44)
    F = rs.random_sample((ni, nf)).astype(np.float32)
    n = steps * B
    u = rs.randint(0, nu, n).astype(np.int32)
    i = rs.randint(0, ni, n).astype(np.int32)
    j = rs.randint(0, ni, n).astype(np.int32)
    u[B:B + 40] = u[B]                      # a batch with a heavily repeated user ...
    i[2 * B:2 * B + 30] = i[2 * B]          # ... a repeated positive item ...
    j[3 * B + 5] = i[3 * B + 6]             # ... and an item that is positive in one triplet and negative in another
    lim = np.sqrt(3.0) * np.sqrt(2.0 / (nu + k))
    params = dict(Bi=np.zeros(ni, np.float32), Gu=rs.uniform(-lim, lim, (nu, k)).astype(np.float32),
                  Gi=rs.uniform(-lim, lim, (ni, k)).astype(np.float32), Tu=rs.uniform(-lim, lim, (nu, k2)).astype(np.float32),
                  E=rs.uniform(-0.03, 0.03, (nf, k2)).astype(np.float32), Bp=rs.uniform(-0.03, 0.03, nf).astype(np.float32))
    lr, lw, lb, le = 0.005, 0.01, 0.01, 0.002
    tr = _lib.VbprTrainer(F, nu, ni, k, k2)
    tr.set_params(**params)
    nll = tr.fit_batches(u, i, j, B, lr, lw, lb, le)
    got = tr.get_params()
    tr.close()
    done, _, want = vbpr_oracle.timed_steps(F, params, u, i, j, B, lr=lr, lambda_w=lw, lambda_b=lb, lambda_e=le,
                                             budget_s=1e9, max_steps=steps, return_params=True)
    assert done == steps and np.isfinite(nll)
    for name in ("Bi", "Gu", "Gi", "Tu", "E", "Bp"):
        a, b = np.asarray(got[name], np.float64).ravel(), np.asarray(want[name], np.float64).ravel()
        assert a.shape == b.shape
        err = np.abs(a - b).max()
        moved = np.abs(b - np.asarray(params[name], np.float64).ravel()).max()
        assert err <= 1e-4, (name, err)
        assert moved > 1e-3, (name, moved)   # the comparison is not between two untouched tables


@pytest.mark.timeout(900)
def test_wmf_netflix_user_count_matches_the_oracle():
    """configs[2]'s user count (480 189 users, k = 128, 128-item batches): every workgroup of the fused user-step kernel
    walks 7-8 consecutive user tiles (column cursors carried from tile to tile, the LDS-resident G tile and the dV
    accumulators re-used across them, a ragged last tile) — none of which the small parity cases reach — against the
    numpy restatement of the reference's graph (oracle/wmf_oracle.py; pinned to the reference's own WMF code over oracle/tf1_shim, TensorFlow itself absent)."""
    import scipy.sparse as sp

    from oracle.wmf_oracle import WmfOracle

    nu, ni, k, nnz = 480189, 600, 128, 2_000_000
    rs = np.random.RandomState(12)
    keys = np.unique(rs.randint(0, nu * ni, size=int(nnz * 1.02), dtype=np.int64))[:nnz]
    u, i = keys // ni, keys % ni
    vals = rs.randint(1, 6, len(keys)).astype(np.float32)
    vals[::97] = 0.0                         # explicit zeros stay "unobserved"
    R = sp.csc_matrix((vals, (u, i)), shape=(nu, ni))
    U = rs.normal(0, 0.1, (nu, k)).astype(np.float32)
    V = rs.normal(0, 0.1, (ni, k)).astype(np.float32)
    perm = rs.permutation(ni)
    batches = [perm[0:128], perm[128:256], perm[256:293]]    # two full batches and a ragged one
    lu, lv, a, b, lr = 0.02, 0.03, 1.0, 0.01, 0.001
    o = WmfOracle(U, V, R, lu, lv, a, b, lr)
    lo = np.array(o.fit_batches(batches))
    tr = _lib.WmfTrainer(R, k)
    tr.set_factors(U, V)
    lg = np.array(tr.fit_batches(batches, lu, lv, a, b, lr))
    Ug, Vg = tr.get_factors()
    tr.close()
    assert np.abs(Ug - o.U).max() <= 1e-4, np.abs(Ug - o.U).max()
    assert np.abs(Vg - o.V).max() <= 1e-4, np.abs(Vg - o.V).max()
    assert np.allclose(lg, lo, rtol=1e-4), np.abs(lg / lo - 1).max()
    assert np.abs(o.U - U).max() > 1e-3 and np.abs(o.V - V).max() > 1e-3


# ---- BASELINE configs[4], one GPU's slice: 12.5 M users x 10 M items, k = 128 (bench.py legs.bpr_k128_scale) -----------------
@pytest.fixture(scope="module")
def scale_slice0():
    from bench import SCALE, scale_slice

    nu, ni, indptr, indices = scale_slice(0)
    return nu, ni, indptr, indices, SCALE["k"]


def _device_tables(torch, nu, ni, k, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    U = (torch.rand((nu, k), device="cuda", generator=g) - 0.5) / k
    V = (torch.rand((ni, k), device="cuda", generator=g) - 0.5) / k
    B = torch.randn(ni, device="cuda", generator=g) * 0.01
    return U, V, B


def test_configs4_slice_passing_bins_invariants(oracle, scale_slice0):
    """The passing-bin regime of the LDS-bin form at the size it exists for (90 112 bins of <= 111 rows, CSR membership
    without a bitmap, U 6.4 GB + V 5.1 GB) — until round 6 only bench.py ran this shape, asserting nothing.
      * the deal is a partition: every bin holds <= cap items, the bins' draw counts sum to nnz (recom_bpr.pyx:218,234: nnz
        draws per epoch);
      * lr = 0 leaves U / V / B bit-identical (every row passes through the LDS and is written back);
      * the sampler equals the oracle's restatement draw for draw on a sample of the bins: a 1 / 128 block of the conveyor
        layout of the same handle data (cornac_hip_bpr_conveyor_enqueue launches exactly those bins), skip counter == oracle;
      * reg = 0 conserves the column sums of V and the sum of B (a step adds +d to the positive's row and -d to the
        negative's, exactly, in LDS), while U, V, B all move; no row lock timed out."""
    import torch

    nu, ni, indptr, indices, k = scale_slice0
    nnz = len(indices)
    tr = _lib.BprTrainer(indptr, indices, nu, ni, nu, ni, k)
    st = tr.ldsbin_stats()
    assert st["bins"] >= 80_000 and st["rows_per_bin"] <= 128 and st["bitmap_words"] == 0 and st["block_threads"] == 512, st
    bin_of, cold, off, _, _ = tr.debug_ldsbin_deal(7, 0)
    per_bin = np.bincount(bin_of, minlength=st["bins"])
    assert per_bin.max() <= st["rows_per_bin"] and per_bin.min() >= st["rows_per_bin"] - 1
    assert int(cold.astype(np.int64).sum()) + int(off[-1]) == nnz
    U, V, B = _device_tables(torch, nu, ni, k, 0)
    U0, V0, B0 = U.clone(), V.clone(), B.clone()
    torch.cuda.synchronize()
    tr.bind_device(U.data_ptr(), V.data_ptr(), B.data_ptr())
    tr.seed_hogwild(7)
    c, s = tr.fit_epochs(1, 0.0, 0.0, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
    assert 0 <= s < 1e-4 * nnz and 0 < c <= nnz - s          # 5 of 10 M items per user: a negative is almost never a positive
    assert torch.equal(U, U0) and torch.equal(V, V0) and torch.equal(B, B0)
    c, s = tr.fit_epochs(1, 0.05, 0.0, True, _lib.NEG_UNIFORM, _lib.MODE_HOGWILD)
    assert tr.ldsbin_stats()["lock_timeouts"] == 0
    dv = (V.double().sum(0) - V0.double().sum(0)).abs().max().item()
    db = abs((B.double().sum() - B0.double().sum()).item())
    moved = [(x - y).abs().max().item() for x, y in ((U, U0), (V, V0), (B, B0))]
    print("configs[4] slice, one epoch at reg = 0: column sums of V drift by %.3g, sum of B by %.3g; max moves %s" % (dv, db, moved))
    assert dv < 0.02 and db < 0.02 and min(moved) > 1e-4
    assert torch.isfinite(U).all() and torch.isfinite(V).all()
    tr.close()
    # the sampler on a sample of the bins, through the conveyor layout
    U.copy_(U0)
    tr = _lib.BprTrainer(indptr, indices, nu, ni, nu, ni, k)
    tr.bind_device(U.data_ptr(), None, None)
    tr.seed_hogwild(99)
    n_bins, bpb, cap = tr.conveyor_setup(128, None, 4242)
    W = bpb * cap
    buf = torch.zeros(W * (k + 1), device="cuda")
    blk, epoch = 37, 3
    tr.conveyor_enqueue(epoch, epoch, [blk], [buf.data_ptr()], 0.0, 0.0, True, _lib.NEG_UNIFORM, 0)
    c, s = tr.sync()
    tables = oracle.ldsbin_tables(indptr, indices, ni, n_bins, 10 ** 9)
    want, draws, _ = oracle.ldsbin_epoch(99, epoch, n_bins, 10 ** 9, indptr, indices, ni, tables=tables, deal=(4242, epoch),
                                         bins=(blk * bpb, (blk + 1) * bpb))
    assert 0.5 * nnz / 128 < draws < 2 * nnz / 128 and s == want, (s, want, draws)
    assert torch.equal(U, U0) and float(buf.abs().max()) == 0.0
    tr.close()


@pytest.mark.parametrize("rings", [1, 4])
def test_configs4_slice_conveyor_as_a_node_rank(scale_slice0, rings):
    """regime 2 at the configs[4] size on one rank laid out like a node of eight (16 or 64 blocks, 16 steps per epoch, a launch =
    one or four bin ranges): lr = 0 through three epochs and two re-deals returns the item table bit for bit; one trained epoch
    at reg = 0 conserves its column sums, moves every table, no lock timeout"""
    import torch

    from cornac_amd.dist import BinConveyorBprTrainer

    nu, ni, indptr, indices, k = scale_slice0
    nnz = len(indices)
    rs = np.random.RandomState(3)
    V = ((rs.random_sample((ni, k)).astype(np.float32) - 0.5) / k)
    B = (rs.standard_normal(ni) * 0.01).astype(np.float32)
    ring = BinConveyorBprTrainer(indptr, indices, nu, ni, k, torch.device("cuda", 0), seed=5, emulate_traffic=True, rings=rings,
                                 virtual_world=8)
    assert ring.nb == 16 and ring.nb_total == 16 * rings and ring.n_bins >= 80_000 and ring.cap <= 128
    g = torch.Generator(device="cuda").manual_seed(1)
    ring.U.copy_((torch.rand((nu, k), device="cuda", generator=g) - 0.5) / k)
    U0 = ring.U.clone()
    ring.load_items(V, B)
    for _ in range(3):
        ring.run_epoch(0.0, 0.0)
    c, s = ring.finish()
    V1, B1 = ring.gather()
    assert ring.redeals == 2 and 0 <= s < 1e-4 * 3 * nnz and 0 < c <= 3 * nnz
    assert np.array_equal(V1, V) and np.array_equal(B1, B) and torch.equal(ring.U, U0)
    del V1, B1
    ring.run_epoch(0.05, 0.0)
    ring.finish()
    V2, B2 = ring.gather()
    assert ring.trainer.tr.ldsbin_stats()["lock_timeouts"] == 0
    dv = np.abs(V2.sum(0, dtype=np.float64) - V.sum(0, dtype=np.float64)).max()
    assert dv < 0.02 and abs(B2.sum(dtype=np.float64) - B.sum(dtype=np.float64)) < 0.02, dv
    assert np.abs(V2 - V).max() > 1e-4 and (ring.U - U0).abs().max().item() > 1e-4 and np.isfinite(V2).all()
    ring.close()
