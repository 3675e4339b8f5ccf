"""GPU parity tests of the MF path (cornac_hip_mf_*), through the C ABI."""
import numpy as np
import pytest

from conftest import golden_dataset, load_golden, synth_dataset
from cornac_amd import MF, _lib

pytestmark = pytest.mark.gpu
CASES = ["tiny", "small", "odd_k", "ml100k_shape"]


def _kw(fx):
    return dict(k=int(fx["k"]), max_iter=int(fx["epochs"]), learning_rate=float(fx["lr"]),
                lambda_reg=2 * float(fx["reg"]), seed=int(fx["seed"]))


@pytest.mark.parametrize("name", CASES)
def test_deterministic_matches_oracle_and_reference_golden(oracle, name):
    fx = load_golden(name)
    ds = golden_dataset(fx)
    m = MF(**_kw(fx)).fit(ds)
    o = oracle.MFOracle(**_kw(fx)).fit(ds)
    assert m.effective_mode == "deterministic"
    for a, b, g in ((m.u_factors, o.u_factors, "mf_U"), (m.i_factors, o.i_factors, "mf_V"),
                    (m.u_biases, o.u_biases, "mf_Bu"), (m.i_biases, o.i_biases, "mf_Bi")):
        assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(b).max()), "HIP deterministic vs oracle"
        assert np.abs(a - fx[g]).max() <= 1e-4 * max(1.0, np.abs(fx[g]).max()), "vs reference golden"
    assert m.global_mean == o.global_mean
    # loss: device sums err^2 in fp64 tree order, the reference sequentially in fp32
    assert np.allclose(m.loss_history, o.loss[: len(m.loss_history)], rtol=2e-4)
    assert np.mean(m.u_factors == o.u_factors) > 0.999


def test_no_bias_and_one_shot_reference_signature(oracle):
    fx = load_golden("small")
    ds = golden_dataset(fx)
    kw = _kw(fx)
    m = MF(use_bias=False, **kw).fit(ds)
    assert np.abs(m.u_factors - fx["mf_nobias_U"]).max() <= 1e-4 * max(1.0, np.abs(fx["mf_nobias_U"]).max())
    assert not m.u_biases.any() and not m.i_biases.any() and m.global_mean == 0
    # backend_cpu.fit_sgd's exact argument list, in place
    o = oracle.MFOracle(use_bias=True, **kw).fit(ds)
    rng = np.random.RandomState(kw["seed"])
    U = rng.normal(0, 0.01, (ds.num_users, kw["k"])).astype(np.float32)
    V = rng.normal(0, 0.01, (ds.num_items, kw["k"])).astype(np.float32)
    Bu, Bi = np.zeros(ds.num_users, np.float32), np.zeros(ds.num_items, np.float32)
    rid, cid, val = ds.uir_tuple
    loss = _lib.mf_fit_sgd(rid, cid, val.astype(np.float32), U, V, Bu, Bi, kw["learning_rate"], kw["lambda_reg"],
                           float(np.float32(ds.global_mean)), kw["max_iter"], True, False, _lib.MODE_DETERMINISTIC)
    assert np.abs(U - o.u_factors).max() <= 1e-6 and np.abs(Bi - o.i_biases).max() <= 1e-6
    assert len(loss) == kw["max_iter"]


@pytest.mark.parametrize("k", [1, 7, 32, 128, 200])
def test_various_k(oracle, k):
    ds = synth_dataset(150, 100, 3000, seed=k)
    kw = dict(k=k, max_iter=4, learning_rate=0.01, lambda_reg=0.02, seed=3)
    m, o = MF(**kw).fit(ds), oracle.MFOracle(**kw).fit(ds)
    assert np.abs(m.u_factors - o.u_factors).max() <= 1e-6
    assert np.abs(m.i_factors - o.i_factors).max() <= 1e-6


def test_early_stop_semantics(oracle):
    """|loss - last_loss| < 1e-5 stops (backend_cpu.pyx:89-93): lr = 0 makes the loss constant, so
    the second epoch triggers the stop in both implementations."""
    ds = synth_dataset(40, 30, 300, seed=1)
    kw = dict(k=4, max_iter=10, learning_rate=0.0, lambda_reg=0.0, seed=1, early_stop=True)
    m, o = MF(**kw).fit(ds), oracle.MFOracle(**kw).fit(ds)
    assert m.epochs_run == o.epochs_run == 2


@pytest.mark.parametrize("k", [10, 64])
def test_hogwild_statistical_parity(oracle, k):
    """throughput mode vs the sequential oracle and the reference's static-chunk multi-thread
    path: training loss per epoch within a few percent."""
    ds = synth_dataset(3000, 2000, 200000, zipf=0.9, seed=5)
    kw = dict(k=k, max_iter=12, learning_rate=0.01, lambda_reg=0.02)
    seq = oracle.MFOracle(seed=2, **kw).fit(ds)
    omp = oracle.MFOracle(seed=2, num_threads=4, **kw).fit(ds)
    m = MF(seed=2, mode="hogwild", **kw).fit(ds)
    assert m.effective_mode == "hogwild"
    assert seq.loss[-1] < 0.9 * seq.loss[0]
    # epoch 1 pays for update staleness (the GPU keeps ~160 k ratings in flight, most of this
    # 200 k-rating set; the CPU 1..4); from epoch 2 on the trajectories coincide
    assert np.allclose(m.loss_history[:1], seq.loss[:1], rtol=0.10), (m.loss_history, seq.loss)
    assert np.allclose(m.loss_history[1:], seq.loss[1:], rtol=0.01), (m.loss_history, seq.loss)
    assert np.allclose(m.loss_history[1:], omp.loss[1:], rtol=0.01)
    assert np.abs(m.i_biases - seq.i_biases).mean() < 0.02


def test_hogwild_owned_path_statistical_parity(oracle):
    """k = 64 with >= 524k ratings takes the user-ownership kernel (plain U / Bu updates by the
    owner wave, round-robin interleaved ratings): same loss trajectory as the sequential oracle."""
    ds = synth_dataset(6000, 3000, 700000, zipf=0.8, seed=9)
    kw = dict(k=64, max_iter=6, learning_rate=0.01, lambda_reg=0.02)
    seq = oracle.MFOracle(seed=2, **kw).fit(ds)
    m = MF(seed=2, mode="hogwild", **kw).fit(ds)
    assert seq.loss[-1] < 0.9 * seq.loss[0]
    assert np.allclose(m.loss_history[:1], seq.loss[:1], rtol=0.06), (m.loss_history, seq.loss)
    assert np.allclose(m.loss_history[1:], seq.loss[1:], rtol=0.015), (m.loss_history, seq.loss)
    assert np.abs(m.u_biases - seq.u_biases).mean() < 0.03
    assert np.isfinite(m.u_factors).all() and np.isfinite(m.i_factors).all()


def test_errors():
    ds = synth_dataset(20, 15, 100, seed=1)
    with pytest.raises(ValueError, match="not supported"):
        MF(backend="cpu").fit(ds)
    rid, cid, val = ds.uir_tuple
    with pytest.raises(_lib.HipError, match="out-of-range"):
        _lib.MfTrainer(rid + 1000, cid, val, ds.num_users, ds.num_items, 4)


def test_hogwild_divergence_is_reported_not_returned():
    """a learning rate far too large drives the racy run to non-finite values: the call fails loudly (advisor r3) instead
    of returning a NaN model; the same tables in sequential mode reproduce the reference, NaNs and all, without an error"""
    rid, cid, val = _coo(400, 60, 40_000, 3)
    rs = np.random.RandomState(0)
    U, V = rs.normal(0, 0.5, (400, 8)).astype(np.float32), rs.normal(0, 0.5, (60, 8)).astype(np.float32)
    tr = _lib.MfTrainer(rid, cid, val, 400, 60, 8)
    tr.set_factors(U, V, np.zeros(400, np.float32), np.zeros(60, np.float32))
    with pytest.raises(_lib.HipError, match="diverged"):
        tr.fit(4, 5.0, 0.0, 3.5, True, False, _lib.MODE_HOGWILD)
    tr.set_factors(U, V, np.zeros(400, np.float32), np.zeros(60, np.float32))
    tr.fit(4, 5.0, 0.0, 3.5, True, False, _lib.MODE_DETERMINISTIC)  # (no error: parity with the reference's seeded loop)
    tr.close()


@pytest.mark.parametrize("k,form", [(16, 1), (64, 1), (64, 2), (300, 1)])
def test_hogwild_trains_through_a_very_popular_item_instead_of_diverging(k, form):
    """One item holding a fifth of 2 M ratings: thousands of atomic updates of its row, all computed from one stale copy, are
    in flight at once — round 4's fused kernel diverged on such data (and raised), the block rotation serialised the row in
    one workgroup.  Hot rows (> 0.1 % of the ratings) now train through copies merged after every launch / phase
    (csrc/mf_blocks.inc "virtual rows"), in BOTH hogwild forms: the run stays finite, lr = 0 leaves the tables untouched,
    and the training error follows the sequential engine's (backend_cpu.pyx:62-88 on one thread) within a few per cent,
    the hot item's row and bias included.  k = 300: the generic kernel (k > 256) resolves the copies' ids too (round 5's
    did not: it read and wrote past the end of V)."""
    n_users, n_items, nnz = 60_000, 3_000, 1 << 21
    rs = np.random.RandomState(5)
    act = rs.lognormal(0, 1.0, n_users)
    rid = np.sort(rs.choice(n_users, nnz, p=act / act.sum())).astype(np.int64)
    pop = 1.0 / np.arange(1, n_items + 1) ** 1.0
    pop[0] = 0.25 * pop.sum()                              # the hot item: ~20 % of all ratings
    perm = rs.permutation(n_items)
    cid = perm[rs.choice(n_items, nnz, p=pop / pop.sum())].astype(np.int64)
    bu, bi = rs.normal(0, 0.5, n_users), rs.normal(0, 0.5, n_items)
    val = np.clip(np.rint(3.5 + bu[rid] + bi[cid] + rs.normal(0, 0.7, nnz)), 1, 5).astype(np.float32)
    hot = int(perm[0])
    assert (cid == hot).mean() > 0.15
    mu, lr, reg = float(val.mean()), 0.01, 0.02
    U0 = rs.normal(0, 0.01, (n_users, k)).astype(np.float32)
    V0 = rs.normal(0, 0.01, (n_items, k)).astype(np.float32)
    z_u, z_i = np.zeros(n_users, np.float32), np.zeros(n_items, np.float32)
    tr = _lib.MfTrainer(rid, cid, val, n_users, n_items, k)
    tr.hogwild_form(form)
    tr.set_factors(U0, V0, z_u, z_i)
    tr.fit(1, 0.0, 0.0, mu, True, False, _lib.MODE_HOGWILD)
    assert tr.hogwild_stats()["form_used"] == form
    U1, V1, Bu1, Bi1 = tr.get_factors()
    assert np.array_equal(V1, V0) and np.array_equal(U1, U0) and not Bi1.any(), "lr = 0: the copies must fold back to the untouched rows"
    loss_h, _ = tr.fit(4, lr, reg, mu, True, False, _lib.MODE_HOGWILD)
    Uh, Vh, Buh, Bih = tr.get_factors()
    tr.set_factors(U0, V0, z_u, z_i)
    loss_d, _ = tr.fit(4, lr, reg, mu, True, False, _lib.MODE_DETERMINISTIC)
    Ud, Vd, Bud, Bid = tr.get_factors()
    tr.close()
    assert np.isfinite(Vh).all() and np.isfinite(Uh).all() and np.isfinite(loss_h).all()
    assert loss_h[-1] < 0.85 * loss_h[0] and np.allclose(loss_h[1:], loss_d[1:], rtol=0.04), (loss_h, loss_d)
    # (hogwild spread of the hot item's bias over boxes / runs: |diff| mostly < 0.05, once 0.177 in ~20 runs of round 6)
    assert abs(Bih[hot] - Bid[hot]) < 0.25 + 0.2 * abs(Bid[hot]), (Bih[hot], Bid[hot])
    # (the factor rows themselves are only defined up to the rotation the trajectory picks: compare what they predict)
    m = cid == hot
    ph = mu + Buh[rid[m]] + Bih[hot] + Uh[rid[m]] @ Vh[hot]
    pd = mu + Bud[rid[m]] + Bid[hot] + Ud[rid[m]] @ Vd[hot]
    eh, ed = float(np.mean((val[m] - ph) ** 2)), float(np.mean((val[m] - pd) ** 2))
    assert eh < 1.1 * ed + 0.02, (eh, ed)


def _coo(n_users, n_items, nnz, seed):
    rs = np.random.RandomState(seed)
    act = rs.lognormal(0, 1.0, n_users)
    rid = rs.choice(n_users, nnz, p=act / act.sum()).astype(np.int64)
    pop = 1.0 / np.arange(1, n_items + 1) ** 0.6
    cid = rs.permutation(n_items)[rs.choice(n_items, nnz, p=pop / pop.sum())].astype(np.int64)
    bu, bi = rs.normal(0, 0.5, n_users), rs.normal(0, 0.5, n_items)
    val = np.clip(np.rint(3.5 + bu[rid] + bi[cid] + rs.normal(0, 0.7, nnz)), 1, 5).astype(np.float32)
    return rid, cid, val


@pytest.mark.parametrize("k", [64, 100, 128, 200])
def test_block_rotation_applies_every_rating_exactly_once(k):
    """The hogwild block rotation (csrc/mf_blocks.inc; 8 launches x 32 sub-rounds, item bins in LDS, user blocks rotating
    inside an XCD): with lr = 0 the epoch loss is 0.5 x the squared error of the START tables over ALL ratings — every
    rating visited exactly once, none twice — and the tables come back bit-identical (rows went through the LDS)."""
    n_users, n_items, nnz = 30_000, 2_000, 1_200_000
    rid, cid, val = _coo(n_users, n_items, nnz, 4)
    rs = np.random.RandomState(k)
    U = rs.normal(0, 0.1, (n_users, k)).astype(np.float32)
    V = rs.normal(0, 0.1, (n_items, k)).astype(np.float32)
    Bu, Bi = rs.normal(0, 0.1, n_users).astype(np.float32), rs.normal(0, 0.1, n_items).astype(np.float32)
    mu = float(val.mean())
    tr = _lib.MfTrainer(rid, cid, val, n_users, n_items, k)
    tr.hogwild_form(2)
    tr.set_factors(U, V, Bu, Bi)
    loss, _ = tr.fit(2, 0.0, 0.0, mu, True, False, _lib.MODE_HOGWILD)
    st = tr.hogwild_stats()
    got = tr.get_factors()
    tr.close()
    # an XCD that receives more than its 32 workgroups (placement is the box's) makes the handle give the rotation up for
    # the fused kernel: the invariants below hold for either form
    assert st["form_used"] == (1 if st["gave_up"] else 2), st
    if st["gave_up"]:
        import warnings

        warnings.warn("MF block rotation gave up on this box (workgroup placement): %r" % (st,))
    pred = mu + Bu[rid] + Bi[cid] + np.einsum("nk,nk->n", U[rid].astype(np.float64), V[cid].astype(np.float64))
    want = 0.5 * float(np.sum((val.astype(np.float64) - pred) ** 2))
    assert abs(loss[0] - want) <= 2e-5 * want and abs(loss[1] - want) <= 2e-5 * want, (loss, want)
    for g, w in zip(got, (U, V, Bu, Bi)):
        assert np.array_equal(g, w)


def test_block_rotation_learns_like_the_fused_kernel():
    """same optimisation problem, same data: the exact block rotation and the fused atomic kernel reach the same
    training error within noise; biases off leaves them untouched"""
    n_users, n_items, nnz, k = 30_000, 2_000, 1_500_000, 64
    rid, cid, val = _coo(n_users, n_items, nnz, 5)
    rs = np.random.RandomState(0)
    U = rs.normal(0, 0.01, (n_users, k)).astype(np.float32)
    V = rs.normal(0, 0.01, (n_items, k)).astype(np.float32)
    zu, zi = np.zeros(n_users, np.float32), np.zeros(n_items, np.float32)
    mu = float(val.mean())
    out = {}
    for form in (2, 1):
        tr = _lib.MfTrainer(rid, cid, val, n_users, n_items, k)
        tr.hogwild_form(form)
        tr.set_factors(U, V, zu, zi)
        loss, _ = tr.fit(8, 0.01, 0.02, mu, True, False, _lib.MODE_HOGWILD)
        assert tr.hogwild_stats()["form_used"] in (form, 1)   # (the rotation may give up on a box with uneven placement)
        out[form] = (loss, tr.get_factors())
        tr.close()
    l2, l1 = out[2][0], out[1][0]
    assert l2[-1] < 0.85 * l2[0] and abs(l2[-1] - l1[-1]) < 0.03 * l1[-1], (l2, l1)
    assert all(np.isfinite(x).all() for x in out[2][1])
    tr = _lib.MfTrainer(rid, cid, val, n_users, n_items, k)
    tr.hogwild_form(2)
    tr.set_factors(U, V, zu, zi)
    tr.fit(2, 0.01, 0.02, mu, False, False, _lib.MODE_HOGWILD)
    _, _, Bu2, Bi2 = tr.get_factors()
    tr.close()
    assert not Bu2.any() and not Bi2.any()


@pytest.mark.parametrize("shape", [(6_000, 300, 67_000, 64), (20_000, 1_100, 300_000, 128), (2_000, 64, 9_000, 48), (3_000, 200, 30_000, 16)])
def test_step_form_trains_few_rows_with_every_rating_in_flight(shape):
    """hogwild form 3 (cornac_hip.h; the per-block step handles of dist.MfBlockRotationTrainer): a handle of few item rows
    whose ratings are nearly all in flight in ONE launch.  The plain fused kernel (form 1) sums dozens of updates of a row
    computed from one stale copy and diverges on the first shape (measured, round 6); form 3 trains such rows through copies
    merged after the launch at ANY size: finite, lr = 0 leaves the tables untouched and counts every rating once, and the
    training error follows the sequential engine's (backend_cpu.pyx:62-88 on one thread)."""
    n_users, n_items, nnz, k = shape
    rid, cid, val = _coo(n_users, n_items, nnz, 11)
    rs = np.random.RandomState(3)
    U0 = rs.normal(0, 0.01, (n_users, k)).astype(np.float32)
    V0 = rs.normal(0, 0.01, (n_items, k)).astype(np.float32)
    zu, zi = np.zeros(n_users, np.float32), np.zeros(n_items, np.float32)
    mu, lr, reg = float(val.mean()), 0.01, 0.02
    tr = _lib.MfTrainer(rid, cid, val, n_users, n_items, k)
    tr.hogwild_form(3)
    tr.set_factors(U0, V0, zu, zi)
    loss0, _ = tr.fit(1, 0.0, 0.0, mu, True, False, _lib.MODE_HOGWILD)
    assert tr.hogwild_stats()["form_used"] == 1
    U1, V1, Bu1, Bi1 = tr.get_factors()
    assert np.array_equal(V1, V0) and np.array_equal(U1, U0) and not Bi1.any() and not Bu1.any()
    pred = mu + np.einsum("nk,nk->n", U0[rid].astype(np.float64), V0[cid].astype(np.float64))
    want = 0.5 * float(np.sum((val.astype(np.float64) - pred) ** 2))
    assert abs(loss0[0] - want) <= 2e-5 * want, (loss0, want)
    loss_h, _ = tr.fit(6, lr, reg, mu, True, False, _lib.MODE_HOGWILD)
    got = tr.get_factors()
    with pytest.raises(_lib.HipError, match="before the handle's first epoch"):
        tr.hogwild_form(1)
    tr.set_factors(U0, V0, zu, zi)
    loss_d, _ = tr.fit(6, lr, reg, mu, True, False, _lib.MODE_DETERMINISTIC)
    tr.close()
    assert all(np.isfinite(x).all() for x in got) and np.isfinite(loss_h).all()
    # (the copies' merge is the "align" rule: between the sum and the mean of their steps — never ahead of the serial run
    # by more than noise, behind it by at most the damping of correlated copies)
    assert loss_h[-1] < loss_h[0] and np.all(loss_h[1:] <= 1.25 * loss_d[1:]) and np.all(loss_h[1:] >= 0.9 * loss_d[1:]), (loss_h, loss_d)
