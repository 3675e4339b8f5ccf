"""GPU parity tests of scoring / ranking (cornac_hip_score_*, cornac_hip_rank_topk)."""
import numpy as np
import pytest

from conftest import golden_dataset, load_golden, synth_dataset
from cornac_amd import BPR, MF, ScoreException, _lib

pytestmark = pytest.mark.gpu


def _tables(n_users, n_items, k, seed, ties=False):
    rs = np.random.RandomState(seed)
    U = rs.normal(0, 0.3, (n_users, k)).astype(np.float32)
    V = rs.normal(0, 0.3, (n_items, k)).astype(np.float32)
    if ties:
        V[n_items // 2:] = V[: n_items - n_items // 2]  # exact score ties between item pairs
    ib = rs.normal(0, 0.1, n_items).astype(np.float32)
    if ties:
        ib[n_items // 2:] = ib[: n_items - n_items // 2]
    ub = rs.normal(0, 0.1, n_users).astype(np.float32)
    return U, V, ib, ub


@pytest.mark.parametrize("k", [1, 2, 7, 10, 64, 100, 128, 200])
@pytest.mark.parametrize("with_user_base", [False, True])
def test_scores_bit_exact_vs_oracle_fma_chain(oracle, k, with_user_base):
    """score_user (VALU) and score_block (fp32 MFMA for k <= 128) == index-ordered fmaf chain."""
    U, V, ib, ub = _tables(77, 333, k, k)
    sc = _lib.Scorer(U, V, ib, ub if with_user_base else None)
    users = np.array([0, 5, 76, 33, 5], np.int32)
    want = oracle.score_block(U, V, ib, ub if with_user_base else None, users)
    got = sc.score_block(users)
    assert np.array_equal(got, want)
    assert np.array_equal(sc.score_user(33), want[3])
    many = np.arange(77, dtype=np.int32)
    assert np.array_equal(sc.score_block(many), oracle.score_block(U, V, ib, ub if with_user_base else None, many))
    sc.close()


@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("topk", [1, 5, 10, 100, 333])
def test_rank_topk_exact_order(oracle, topk, ties):
    U, V, ib, ub = _tables(50, 333, 16, 3, ties=ties)
    sc = _lib.Scorer(U, V, ib, None)
    users = np.arange(50, dtype=np.int32)
    items, scores = sc.rank_topk(users, topk)
    full = oracle.score_block(U, V, ib, None, users)
    for b in range(50):
        want, _ = oracle.rank(full[b], 333, 333, k=topk)
        assert np.array_equal(items[b], want[:topk])
        assert np.array_equal(scores[b], full[b][want[:topk]])
    sc.close()


def test_rank_topk_with_exclusions_and_short_rows(oracle):
    U, V, ib, ub = _tables(20, 64, 8, 4)
    sc = _lib.Scorer(U, V, ib, None)
    users = np.array([3, 7, 11], np.int32)
    excl_lists = [np.arange(0, 64, 2), np.array([], np.int64), np.arange(60)]  # last user keeps 4 candidates
    indptr = np.cumsum([0] + [len(x) for x in excl_lists]).astype(np.int64)
    indices = np.concatenate(excl_lists).astype(np.int32)
    items, scores = sc.rank_topk(users, 10, exclude=(indptr, indices))
    full = oracle.score_block(U, V, ib, None, users)
    for b in range(3):
        cand = np.setdiff1d(np.arange(64), excl_lists[b])
        want, _ = oracle.rank(full[b], 64, 64, item_indices=cand, k=10)
        n = min(10, len(cand))
        assert np.array_equal(items[b][:n], want[:n])
        assert (items[b][n:] == -1).all() and np.isneginf(scores[b][n:]).all()
    sc.close()


def test_full_ranking_large_item_count(oracle):
    """topk == n_items > 2048 goes through the full-sort kernel."""
    U, V, ib, ub = _tables(6, 5000, 12, 9)
    sc = _lib.Scorer(U, V, ib, None)
    users = np.arange(6, dtype=np.int32)
    items, scores = sc.rank_topk(users, 5000)
    full = oracle.score_block(U, V, ib, None, users)
    for b in range(6):
        want, _ = oracle.rank(full[b], 5000, 5000, k=-1)
        assert np.array_equal(items[b], want)
    sc.close()


@pytest.mark.parametrize("name", ["small", "ml100k_shape"])
def test_model_score_and_rank_vs_reference_golden(oracle, name):
    """BPR.score / rank after a seeded fit vs what the real reference returned (golden): scores
    within BLAS-order tolerance, ranked order identical wherever the reference's adjacent score
    gaps exceed that tolerance, top-k identical when its boundary is not a near-tie."""
    fx = load_golden(name)
    ds = golden_dataset(fx)
    kw = dict(k=int(fx["k"]), max_iter=int(fx["epochs"]), learning_rate=float(fx["lr"]),
              lambda_reg=float(fx["reg"]), seed=int(fx["seed"]))
    for tag, m in (("bpr", BPR(**kw).fit(ds)), ("mf", MF(**dict(kw, lambda_reg=2 * kw["lambda_reg"])).fit(ds))):
        kk = int(fx[tag + "_rank_k"])
        for t, u in enumerate(fx[tag + "_rank_users"]):
            ref_s, ref_rank = fx[tag + "_score_%d" % t], fx[tag + "_rank_full_%d" % t]
            s = m.score(int(u))
            assert s.dtype == np.float32 and np.abs(s - ref_s).max() < 2e-5
            ranked, scores = m.rank(int(u))
            assert np.abs(scores - fx[tag + "_rank_scores_%d" % t]).max() < 2e-5
            assert sorted(ranked.tolist()) == sorted(ref_rank.tolist())
            # SURVEY.md section 7 (hard part 4): the order must be the reference's wherever adjacent scores differ by
            # more than the summation-order noise of a k-term fp32 dot product, 8 ulp x sqrt(k) of the score magnitude
            tol = 8 * np.finfo(np.float32).eps * np.sqrt(kw["k"]) * max(float(np.abs(ref_s).max()), 1e-30)
            gap_ok = np.abs(np.diff(ref_s[ref_rank])) > tol
            safe = np.concatenate([[True], gap_ok]) & np.concatenate([gap_ok, [True]])
            assert safe.mean() > 0.2, "the gate must not be vacuous (near-ties below the noise floor are exempt)"
            assert np.array_equal(ranked[safe], ref_rank[safe])
            top, _ = m.rank(int(u), k=kk)
            # recommender.py:521-528: every candidate comes back, the first k in order
            assert len(top) == len(ranked) and sorted(top.tolist()) == sorted(ranked.tolist())
            if safe[: kk + 1].all():
                assert np.array_equal(top[:kk], fx[tag + "_rank_top_%d" % t][:kk])


def test_recommender_surface(tmp_path, oracle):
    """rank with candidates, recommend(remove_seen), save/load, clone, exceptions — the reference's
    tests/cornac/models/test_recommender.py scenarios."""
    ds = synth_dataset(30, 25, 300, seed=6)
    m = MF(k=4, max_iter=3, seed=123).fit(ds)
    assert m.knows_user(0) and not m.knows_user(ds.num_users) and m.knows_item(3)
    cand = np.array([3, 1, 20, 7, 11])
    ranked, scores = m.rank(2, item_indices=cand)
    s = m.score(2)
    assert np.array_equal(scores, s[cand])
    assert np.array_equal(ranked, cand[np.argsort(s[cand], kind="stable")[::-1]])
    uid = ds.user_ids[2]
    rec_all = m.recommend(uid, k=5)
    rec_unseen = m.recommend(uid, k=5, remove_seen=True, train_set=ds)
    seen = {ds.item_ids[i] for i in ds.matrix.getrow(2).indices}
    assert len(rec_all) == 5 and not (set(rec_unseen) & seen)
    with pytest.raises(ValueError):
        m.recommend("nobody")
    with pytest.raises(ScoreException):
        m.score(0, ds.num_items + 5)
    assert m.rate(0, ds.num_items + 5) == pytest.approx(np.clip(m.global_mean, ds.min_rating, ds.max_rating))
    path = m.save(str(tmp_path))
    m2 = MF.load(path)
    assert np.array_equal(m2.u_factors, m.u_factors) and np.array_equal(m2.score(2), s)
    c = m.clone({"k": 7})
    assert c.k == 7 and c.max_iter == 3 and c.u_factors is None and not c.is_fitted
    items, sc = m.rank_batch(np.arange(10), k=3, exclude=(ds.matrix.indptr[:11].astype(np.int64), ds.matrix.indices[: ds.matrix.indptr[10]]))
    for b in range(10):
        assert not (set(items[b].tolist()) & set(ds.matrix.getrow(b).indices.tolist()))


def test_rate_batch_matches_per_pair_rate(oracle):
    """batched rating prediction (rating_eval's hot loop) == rate() pair by pair, incl. clipping,
    unknown items (default score) and users unknown to MF (bias-only score is NOT table driven)."""
    ds = synth_dataset(40, 30, 500, seed=11)
    m = MF(k=6, max_iter=5, seed=3).fit(ds)
    rs = np.random.RandomState(0)
    u = rs.randint(0, ds.num_users, 200)
    i = rs.randint(0, ds.num_items + 3, 200)  # a few unknown items
    got = m.rate_batch(u, i)
    want = np.array([float(m.rate(int(a), int(b))) for a, b in zip(u, i)])
    assert np.allclose(got, want, atol=2e-6)
    raw = m.rate_batch(u, i, clipping=False)
    want_raw = np.array([float(m.rate(int(a), int(b), clipping=False)) for a, b in zip(u, i)])
    assert np.allclose(raw, want_raw, atol=2e-6)
    # bit-exact vs the oracle's fma chain for known pairs
    sc = m._get_scorer()
    known = i < ds.num_items
    full = oracle.score_block(m.u_factors, m.i_factors, (m.global_mean + m.i_biases).astype(np.float32), m.u_biases,
                              u[known].astype(np.int32))
    assert np.array_equal(sc.score_pairs(u[known], i[known]), full[np.arange(known.sum()), i[known]])


@pytest.mark.parametrize("k,with_excl,ni", [(16, True, 3000), (64, False, 3000), (100, True, 3000), (64, True, 3001),
                                             (32, False, 2998), (64, True, 2999), (16, True, 3)])
def test_rank_positions_equal_counts_over_the_score_block(k, with_excl, ni):
    """cornac_hip_rank_positions vs NumPy counts over the same (bit-identical) device scores: tied scores, targets
    at the extremes of a row, rows without targets, more targets per row than one register pass, excluded and
    out-of-range targets; item counts that leave the rows of the score tile at every 16-byte misalignment (scalar head
    and tail around the 16-byte body, byte-wise exclusion flags) and a catalogue shorter than one vector"""
    rs = np.random.RandomState(k + ni)
    nu = 70
    U = rs.normal(0, 0.3, (nu, k)).astype(np.float32)
    V = rs.normal(0, 0.3, (ni, k)).astype(np.float32)
    if ni > 300:
        V[100:160] = V[200:260]                   # runs of exactly tied scores
    ib = np.zeros(ni, np.float32)
    sc = _lib.Scorer(U, V, ib, None)
    users = rs.permutation(nu)[:50].astype(np.int32)
    S = sc.score_block(users)
    excl = [np.sort(rs.choice(ni, rs.randint(0, min(40, ni)), replace=False)).astype(np.int32) if with_excl else
            np.empty(0, np.int32) for _ in users]
    tgts = []
    for r in range(len(users)):
        n_t = min([0, 1, 3, 4, 5, 9, 33][r % 7], ni)
        t = rs.choice(ni, n_t, replace=False).astype(np.int32)
        if n_t >= 3 and ni > 300:
            t[0], t[1] = 120, 220                 # two members of one tie run
            t[2] = int(np.argmax(S[r]))           # the row's best item
        if n_t >= 9:
            t[3] = ni + 5                         # out of range
            if with_excl and len(excl[r]):
                t[4] = excl[r][0]                 # an excluded item
        tgts.append(t)
    csr = lambda rows: (np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.int64),
                        np.concatenate(rows).astype(np.int32))
    greater, pos, ge, scores = sc.rank_positions(users, csr(tgts), exclude=csr(excl) if with_excl else None)
    p = 0
    for r in range(len(users)):
        cand = np.setdiff1d(np.arange(ni), excl[r])
        s_c = S[r][cand]
        for t in tgts[r]:
            if t >= ni or t in excl[r]:
                want = (-1, -1, -1)
                assert scores[p] == -np.inf
            else:
                s = S[r][t]
                want = (int((s_c > s).sum()), int((s_c > s).sum() + ((s_c == s) & (cand > t)).sum()),
                        int((s_c >= s).sum()))
                assert scores[p] == s
            assert (greater[p], pos[p], ge[p]) == want, (r, t, want)
            p += 1
    assert p == len(greater)
    # consistent with the ranking itself
    items, _ = sc.rank_topk(users[:8], ni, exclude=csr(excl[:8]) if with_excl else None)
    p = 0
    for r in range(8):
        for t in tgts[r]:
            if pos[p] >= 0:
                assert items[r, pos[p]] == t
            p += 1
    sc.close()


def test_batched_evaluation_equals_per_user_flow():
    """cornac_amd.eval.ranking_eval (one fused GEMM + top-k launch with exclusion lists) gives the
    same per-user metric values as the reference-style loop over model.rank(); rating_eval (one
    gather-dot-clip kernel) the same RMSE/MAE as rate() pair by pair."""
    from cornac_amd import Dataset, eval as ev, metrics as mm

    rs = np.random.RandomState(3)
    keys = rs.permutation(np.unique(rs.randint(400, size=30000).astype(np.int64) * 300 + rs.randint(300, size=30000)))
    data = [(int(k // 300), int(k % 300), float(1 + k % 5)) for k in keys]
    train = Dataset.build(data[:18000])
    test = Dataset.build(data[18000:], global_uid_map=train.uid_map, global_iid_map=train.iid_map,
                         exclude_unknowns=True)
    m = BPR(k=16, max_iter=10, learning_rate=0.05, seed=1).fit(train)
    metrics = [mm.Recall(k=10), mm.NDCG(k=10), mm.Precision(k=5), mm.HitRatio(k=1)]
    avg, user = ev.ranking_eval(m, metrics, train, test)

    class PerUser:  # same model without rank_batch -> forces the per-user flow
        def __init__(self, m):
            self.m = m

        def rank(self, **kw):
            return self.m.rank(**kw)

    avg_ref, user_ref = ev.ranking_eval(PerUser(m), metrics, train, test)
    assert user[0].keys() == user_ref[0].keys() and len(user[0]) > 100
    for a, b in zip(user, user_ref):
        assert all(a[u] == pytest.approx(b[u]) for u in a)
    assert 0 < avg[0] < 1
    # metrics over the full candidate list: device full rankings of user blocks + tie-aware batched forms vs the
    # per-user flow (a pair of candidates whose scores differ in the last ulp between the two score kernels may
    # swap: one pair moves a user's AUC by 1 / (positives * negatives))
    class RankBatchOnly(PerUser):  # without rank_positions_batch -> full rankings of user blocks are copied back
        def rank_batch(self, *a, **kw):
            return self.m.rank_batch(*a, **kw)

        @property
        def total_items(self):
            return self.m.total_items

    for full in ([mm.AUC(), mm.MAP(), mm.MRR()], [mm.AUC(), mm.MAP(), mm.NCRR(k=10), mm.FMeasure(k=5)]):
        avg_fr, user_fr = ev.ranking_eval(PerUser(m), full, train, test)
        for mdl in (m, RankBatchOnly(m)):     # device position counts / device full rankings
            avg_f, user_f = ev.ranking_eval(mdl, full, train, test, batch_users_full=96, batch_users=150)
            assert np.allclose(avg_f, avg_fr, rtol=0, atol=2e-5), (avg_f, avg_fr)
            for a, b in zip(user_f, user_fr):
                assert a.keys() == b.keys() and all(abs(a[u] - b[u]) < 5e-3 for u in a)
            assert 0.5 < avg_f[0] < 1
    mf = MF(k=8, max_iter=10, seed=2).fit(train)
    (rmse, mae), _ = ev.rating_eval(mf, [mm.RMSE(), mm.MAE()], test)
    u, i, r = test.uir_tuple
    pred = np.array([float(mf.rate(int(a), int(b))) for a, b in zip(u, i)])
    assert rmse == pytest.approx(np.sqrt(np.mean((r - pred) ** 2)), rel=1e-6)
    assert mae == pytest.approx(np.mean(np.abs(r - pred)), rel=1e-6)


@pytest.mark.parametrize("k,topk", [(100, 10), (64, 7), (128, 24), (128, 32), (64, 32), (16, 25), (10, 1)])
def test_fused_rank_long_item_ranges_and_many_segments(oracle, k, topk):
    """large catalogue, few users: one row block is cut into many segments (threshold hand-off between them),
    k = 100/128 takes the two-workgroups-per-CU variant with topk + 32 candidate slots; exact order and scores
    against the oracle's fma-chain scores, with user and item bases and tied scores in the data"""
    rs = np.random.RandomState(k + topk)
    nu, ni = 300, 150_000
    U = rs.normal(0, 0.3, (nu, k)).astype(np.float32)
    V = rs.normal(0, 0.3, (ni, k)).astype(np.float32)
    V[1000:1040] = V[2000:2040]  # exact ties across distant tiles
    ib = rs.normal(0, 0.2, ni).astype(np.float32)
    ib[1000:1040] = ib[2000:2040]
    ub = rs.normal(0, 0.2, nu).astype(np.float32)
    sc = _lib.Scorer(U, V, ib, ub)
    users = np.arange(nu, dtype=np.int32)
    items, scores = sc.rank_topk(users, topk)
    sc.close()
    full = oracle.score_block(U, V, ib, ub, users)
    for b in range(nu):
        want, _ = oracle.rank(full[b], ni, ni, k=topk)
        assert np.array_equal(items[b], want[:topk]), b
        assert np.array_equal(scores[b], full[b][want[:topk]])


@pytest.mark.parametrize("k,n_items", [(64, 5000), (100, 40000)])
def test_resident_exclusion_lists_rank_like_per_call_lists(oracle, k, n_items):
    """cornac_hip_scorer_set_exclusions + cornac_hip_rank_topk_resident (lists keyed by USER id, kept on the device,
    turned into exclusion bitmaps in rank order) against the per-call form and the oracle: a user range, an arbitrary
    user list, results in pageable and in the scorer's page-locked host memory, scores optional; heavy users whose
    lists cover most of the catalogue, empty lists, item 0 and the last item excluded"""
    rs = np.random.RandomState(k)
    nu, topk = 700, 10
    U = rs.normal(0, 0.3, (nu, k)).astype(np.float32)
    V = rs.normal(0, 0.3, (n_items, k)).astype(np.float32)
    ib = rs.normal(0, 0.2, n_items).astype(np.float32)
    lens = rs.randint(0, 60, nu)
    lens[3], lens[11], lens[12] = n_items - topk, 0, n_items - topk - 5
    rows = [np.sort(rs.choice(n_items, n, replace=False)).astype(np.int32) for n in lens]
    rows[5] = np.array([0, n_items - 1], np.int32)
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    indices = np.concatenate(rows).astype(np.int32)
    sc = _lib.Scorer(U, V, ib, None)
    sc.set_exclusions(indptr, indices)
    full = oracle.score_block(U, V, ib, None, np.arange(nu, dtype=np.int32))

    def check(users, items, scores):
        for b, u in enumerate(users):
            s = full[u].copy()
            keep = np.ones(n_items, bool)
            keep[rows[u]] = False
            cand = np.flatnonzero(keep)
            want, _ = oracle.rank(s, n_items, n_items, k=-1, item_indices=cand)
            assert np.array_equal(items[b], want[:topk]), (b, u)
            if scores is not None:
                assert np.array_equal(scores[b], s[want[:topk]])

    items, scores = sc.rank_topk_resident((0, nu), topk)
    check(range(nu), items, scores)
    users = rs.permutation(nu)[:333].astype(np.int32)
    items2, none, ms = sc.rank_topk_resident(users, topk, fetch="items", timed=True, pinned=True)
    assert none is None and ms > 0
    check(users, items2.copy(), None)
    items3, scores3 = sc.rank_topk_resident(users, topk, pinned=True)   # re-uses (and outgrows) the pinned buffer
    check(users, items3, scores3)
    # the per-call form with the same lists
    sel_ptr = np.concatenate([[0], np.cumsum([len(rows[u]) for u in users])]).astype(np.int64)
    sel_idx = np.concatenate([rows[u] for u in users]).astype(np.int32)
    items4, scores4 = sc.rank_topk(users, topk, exclude=(sel_ptr, sel_idx))
    assert np.array_equal(items4, items3) and np.array_equal(scores4, scores3)
    assert sc.rank_topk_resident((5, 1), topk, fetch=False) == (None, None)
    sc.close()


def test_exclusion_bitmap_is_chunked_by_rows_beyond_its_budget():
    """the exclusion bitmap of the fused top-k costs rows x ceil(n_items / 32) x 4 bytes; beyond 512 MB
    the rows are ranked in chunks.  36 000 users x 200 000 items (900 MB unchunked): the chunked call must equal the same
    ranking asked for in two halves and honour every exclusion."""
    rs = np.random.RandomState(3)
    n_users, n_items, k = 36_000, 200_000, 16
    U = rs.normal(0, 0.3, (n_users, k)).astype(np.float32)
    V = rs.normal(0, 0.3, (n_items, k)).astype(np.float32)
    Bi = rs.normal(0, 0.1, n_items).astype(np.float32)
    sc = _lib.Scorer(U, V, Bi, None)
    users = np.arange(n_users, dtype=np.int32)
    per = 6
    excl_idx = np.sort(rs.randint(0, n_items, (n_users, per)), axis=1)
    excl_idx[:, 1:] += (np.diff(excl_idx, axis=1) == 0)  # (distinct within a row is not required, sorted is)
    excl_idx = np.sort(np.minimum(excl_idx, n_items - 1), axis=1).astype(np.int32).ravel()
    excl_ptr = (np.arange(n_users + 1, dtype=np.int64) * per)
    items, scores = sc.rank_topk(users, 10, exclude=(excl_ptr, excl_idx))
    h = n_users // 2
    a, _ = sc.rank_topk(users[:h], 10, exclude=(excl_ptr[:h + 1], excl_idx[:h * per]))
    b, _ = sc.rank_topk(users[h:], 10, exclude=(excl_ptr[h:] - excl_ptr[h], excl_idx[h * per:]))
    sc.close()
    assert np.array_equal(items[:h], a) and np.array_equal(items[h:], b)
    ex = excl_idx.reshape(n_users, per)
    assert not (items[:, :, None] == ex[:, None, :]).any()
    assert (np.diff(scores, axis=1) <= 0).all()


@pytest.mark.parametrize("n_items", [32, 64, 33, 31, 12, 1024, 1025])
@pytest.mark.parametrize("with_excl", [False, True])
def test_fused_rank_catalogues_around_the_tile_size(oracle, n_items, with_excl):
    """The fused top-k kernel stages whole 32-item tiles without clamps: its rank-order tables are padded to whole tiles on
    the host (zero rows, NaN item bases, item id 0).  Catalogues of exactly one / two / many tiles, one item over and under
    a tile, fewer items than topk + exclusions leave: no padded item may ever be returned, negative scores included (a
    padded row scores NaN, not 0), the exclusion bitmap's last word is partial."""
    rs = np.random.RandomState(n_items)
    nu, k, topk = 200, 64, 10
    U = rs.normal(0, 0.3, (nu, k)).astype(np.float32)
    V = rs.normal(0, 0.3, (n_items, k)).astype(np.float32)
    ib = (rs.normal(0, 0.2, n_items) - 5.0).astype(np.float32)      # every real score is far below a padded row's 0 + 0
    sc = _lib.Scorer(U, V, ib, None)
    users = np.arange(nu, dtype=np.int32)
    tk = min(topk, n_items)
    if with_excl:
        rows = [np.sort(rs.choice(n_items, rs.randint(0, max(1, n_items - tk)), replace=False)).astype(np.int32) for _ in users]
        rows[0] = np.array([n_items - 1], np.int32)                 # the last real item: the bit next to the padding
        indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
        items, scores = sc.rank_topk(users, tk, exclude=(indptr, np.concatenate(rows).astype(np.int32)))
    else:
        rows = [np.empty(0, np.int32) for _ in users]
        items, scores = sc.rank_topk(users, tk)
    sc.close()
    full = oracle.score_block(U, V, ib, None, users)
    for b in range(nu):
        keep = np.ones(n_items, bool)
        keep[rows[b]] = False
        cand = np.flatnonzero(keep)
        want, _ = oracle.rank(full[b], n_items, n_items, k=-1, item_indices=cand)
        n_want = min(tk, len(cand))
        assert np.array_equal(items[b][:n_want], want[:n_want]), (b, items[b], want[:tk])
        assert np.array_equal(scores[b][:n_want], full[b][want[:n_want]])
