/*
 * TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the cornac hot path.
 *
 * This file is the checker for the HIP kernels in cornac_amd/csrc/.  Only
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load
 * it.  Nothing under cornac_amd/ links, imports or calls it; the product path
 * fails loudly when libcornac_hip.so is missing.
 *
 * Parity status: PINNED.  Every function below is validated against the real
 * compiled reference (oracle/_ref, built from /root/reference by
 * oracle/build_ref.py) in tests/test_oracle_vs_reference.py, and against the
 * golden vectors that the real reference produced (tests/golden/ npz files, made by
 * tests/golden/make_golden.py).  The reference's own test-suite holds exactly
 * one known-answer test on this path (tests/cornac/utils/test_fastdot.py:26-37)
 * which is replayed in tests/test_oracle_golden.py.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference).  Written from the algorithm's description; no reference
 * source text is reproduced.
 *
 * Build: oracle/Makefile  (gcc -O2 -fopenmp, NO -ffast-math: the oracle's
 * floating point is strict IEEE in index order, so it is a well-defined
 * function of its inputs; the reference's -ffast-math build differs from it by
 * a few ulp per dot product, far inside the 1e-4 parity tolerance).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- *
 * MT19937 — boost::random::mt19937 (cornac/utils/external/boost/random/
 * mersenne_twister.hpp:624 typedef; seeding = the standard init_genrand,
 * which is also NumPy's RandomState legacy seeding).  Used by RNGVector
 * (cornac/models/bpr/recom_bpr.pyx:54-62).
 * ------------------------------------------------------------------------- */
typedef struct {
    uint32_t mt[624];
    int32_t idx; /* next unread word; 624 => regenerate first */
    int32_t pad[15]; /* sizeof == 2560 = 40 cache lines: per-thread engines never share a line */
} oracle_mt19937;

void oracle_mt_seed(oracle_mt19937 *g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static void mt_twist(oracle_mt19937 *g) {
    uint32_t *mt = g->mt;
    for (int i = 0; i < 624; ++i) {
        uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        uint32_t v = mt[(i + 397) % 624] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908b0dfu;
        mt[i] = v;
    }
    g->idx = 0;
}

uint32_t oracle_mt_next(oracle_mt19937 *g) {
    if (g->idx >= 624) mt_twist(g);
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* ------------------------------------------------------------------------- *
 * boost 1.72 uniform_int_distribution<long>(0, hi)(mt19937)
 * (cornac/utils/external/boost/random/uniform_int_distribution.hpp:49-228,
 * the `brange > range` branch :188-227 and the trivial branches :64-70).
 * hi >= 2^32 would take the multi-draw branch (:71-187); the reference BPR
 * cannot reach it (int32 indices, recom_bpr.pyx:186) so it is an error here.
 * Returns -1 for that unsupported case.
 * ------------------------------------------------------------------------- */
int64_t oracle_boost_uniform(oracle_mt19937 *g, uint64_t hi) {
    if (hi == 0) return 0; /* no engine call */
    if (hi == 0xFFFFFFFFull) return (int64_t)oracle_mt_next(g);
    if (hi > 0xFFFFFFFFull) return -1;
    uint32_t range = (uint32_t)hi;
    uint32_t bucket = 0xFFFFFFFFu / (range + 1u);
    if (0xFFFFFFFFu % (range + 1u) == range) ++bucket;
    for (;;) {
        uint32_t r = oracle_mt_next(g) / bucket;
        if (r <= range) return (int64_t)r;
    }
}

/* Bulk draws (used by tests of the device sampler). */
int oracle_boost_uniform_fill(oracle_mt19937 *g, uint64_t hi, int64_t n, int64_t *out) {
    for (int64_t s = 0; s < n; ++s) {
        int64_t v = oracle_boost_uniform(g, hi);
        if (v < 0) return -1;
        out[s] = v;
    }
    return 0;
}

/* raw tempered outputs (tests of the device MT19937 block generator) */
void oracle_mt_fill_raw(oracle_mt19937 *g, int64_t n, uint32_t *out) {
    for (int64_t s = 0; s < n; ++s) out[s] = oracle_mt_next(g);
}

/* ------------------------------------------------------------------------- *
 * has_non_zero (cornac/models/bpr/recom_bpr.pyx:46-51): std::binary_search of
 * colid in the sorted CSR row.
 * ------------------------------------------------------------------------- */
static int has_non_zero(const int32_t *indptr, const int32_t *indices, int32_t row, int32_t col) {
    int32_t lo = indptr[row], hi = indptr[row + 1];
    while (lo < hi) {
        int32_t mid = lo + ((hi - lo) >> 1);
        if (indices[mid] < col) lo = mid + 1; else hi = mid;
    }
    return lo < indptr[row + 1] && indices[lo] == col;
}

/* one BPR SGD step (cornac/models/bpr/recom_bpr.pyx:246-267).
 * exp: Cython's libc.math exp on a C++ float resolves to the float overload
 * (expf); 1.0/(1.0+...) is evaluated in double and rounded to float on
 * assignment.  The oracle defines e = (float)exp((double)score), i.e. the
 * correctly rounded float exponential: glibc's expf equals it except in the
 * ~0.4% of arguments where its 0.502-ulp bound matters, and the device's
 * double-precision exp rounds to the same float, which is what lets the HIP
 * deterministic kernels be compared bit-for-bit against this file. */
static inline int bpr_step(float *user, float *item_i, float *item_j, float *B, int32_t i_id, int32_t j_id,
                           int k, float lr, float reg, int use_bias) {
    float score = B[i_id] - B[j_id];
    for (int f = 0; f < k; ++f) score = score + user[f] * (item_i[f] - item_j[f]);
    float e = (float)exp((double)score);
    float z = (float)(1.0 / (1.0 + (double)e));
    for (int f = 0; f < k; ++f) {
        float temp = user[f];
        user[f] += lr * (z * (item_i[f] - item_j[f]) - reg * user[f]);
        item_i[f] += lr * (z * temp - reg * item_i[f]);
        item_j[f] += lr * (-z * temp - reg * item_j[f]);
    }
    if (use_bias) {
        B[i_id] += lr * (z - reg * B[i_id]);
        B[j_id] += lr * (-z - reg * B[j_id]);
    }
    return z < .5f;
}

/* ------------------------------------------------------------------------- *
 * BPR._fit_sgd, seeded (num_threads == 1) — one epoch = num_samples draws
 * (cornac/models/bpr/recom_bpr.pyx:208-269).  rng_pos / rng_neg persist across
 * epochs (RNGVector objects are created once per fit, :188-191).  WBPR passes
 * the SAME generator for both and neg_item_ids = X.indices
 * (cornac/models/bpr/recom_wbpr.pyx:131-139): pass rng_neg == rng_pos.
 * Optionally records the sampled (i_index, j_index) pairs and skip flags.
 * ------------------------------------------------------------------------- */
int oracle_bpr_epoch_seq(oracle_mt19937 *rng_pos, oracle_mt19937 *rng_neg, uint64_t pos_hi, uint64_t neg_hi,
                         int64_t num_samples, const int32_t *user_ids, const int32_t *item_ids,
                         const int32_t *neg_item_ids, const int32_t *indptr, float *U, float *V, float *B, int k,
                         float lr, float reg, int use_bias, int64_t *correct_out, int64_t *skipped_out,
                         int64_t *rec_ii, int64_t *rec_jj, uint8_t *rec_skip) {
    int64_t correct = 0, skipped = 0;
    for (int64_t s = 0; s < num_samples; ++s) {
        int64_t i_index = oracle_boost_uniform(rng_pos, pos_hi);
        int64_t j_index = oracle_boost_uniform(rng_neg, neg_hi);
        if (i_index < 0 || j_index < 0) return -1;
        int32_t i_id = item_ids[i_index];
        int32_t j_id = neg_item_ids[j_index];
        int32_t u_id = user_ids[i_index];
        if (rec_ii) { rec_ii[s] = i_index; rec_jj[s] = j_index; }
        if (has_non_zero(indptr, item_ids, u_id, j_id)) {
            ++skipped;
            if (rec_skip) rec_skip[s] = 1;
            continue;
        }
        if (rec_skip) rec_skip[s] = 0;
        correct += bpr_step(U + (int64_t)u_id * k, V + (int64_t)i_id * k, V + (int64_t)j_id * k, B, i_id, j_id, k,
                            lr, reg, use_bias);
    }
    *correct_out = correct;
    *skipped_out = skipped;
    return 0;
}

/* ------------------------------------------------------------------------- *
 * The same epoch for float64 factors: `_fit_sgd` is a fused-type (`floating`) function (recom_bpr.pyx:211-214), so
 * with double U / V / B every local (`z`, `score`, `temp`, `lr`, `reg`, :219-224) is a double and the whole step is
 * double arithmetic in the order written at :245-266.
 * ------------------------------------------------------------------------- */
static inline int bpr_step_f64(double *user, double *item_i, double *item_j, double *B, int32_t i_id, int32_t j_id,
                               int k, double lr, double reg, int use_bias) {
    double score = B[i_id] - B[j_id];
    for (int f = 0; f < k; ++f) score = score + user[f] * (item_i[f] - item_j[f]);
    double z = 1.0 / (1.0 + exp(score));
    for (int f = 0; f < k; ++f) {
        double temp = user[f];
        user[f] += lr * (z * (item_i[f] - item_j[f]) - reg * user[f]);
        item_i[f] += lr * (z * temp - reg * item_i[f]);
        item_j[f] += lr * (-z * temp - reg * item_j[f]);
    }
    if (use_bias) {
        B[i_id] += lr * (z - reg * B[i_id]);
        B[j_id] += lr * (-z - reg * B[j_id]);
    }
    return z < .5;
}

int oracle_bpr_epoch_seq_f64(oracle_mt19937 *rng_pos, oracle_mt19937 *rng_neg, uint64_t pos_hi, uint64_t neg_hi,
                             int64_t num_samples, const int32_t *user_ids, const int32_t *item_ids,
                             const int32_t *neg_item_ids, const int32_t *indptr, double *U, double *V, double *B, int k,
                             double lr, double reg, int use_bias, int64_t *correct_out, int64_t *skipped_out) {
    int64_t correct = 0, skipped = 0;
    for (int64_t s = 0; s < num_samples; ++s) {
        int64_t i_index = oracle_boost_uniform(rng_pos, pos_hi);
        int64_t j_index = oracle_boost_uniform(rng_neg, neg_hi);
        if (i_index < 0 || j_index < 0) return -1;
        int32_t i_id = item_ids[i_index];
        int32_t j_id = neg_item_ids[j_index];
        int32_t u_id = user_ids[i_index];
        if (has_non_zero(indptr, item_ids, u_id, j_id)) {
            ++skipped;
            continue;
        }
        correct += bpr_step_f64(U + (int64_t)u_id * k, V + (int64_t)i_id * k, V + (int64_t)j_id * k, B, i_id, j_id, k,
                                lr, reg, use_bias);
    }
    *correct_out = correct;
    *skipped_out = skipped;
    return 0;
}

/* ------------------------------------------------------------------------- *
 * BPR._fit_sgd, unseeded (num_threads = T > 1): racy Hogwild, one generator
 * pair per thread (recom_bpr.pyx:228-267, `prange(..., schedule='guided')`).
 * This is the CPU throughput baseline ("port" of the OpenMP path); results
 * are not reproducible, exactly like the reference's.  rngs_pos/rngs_neg are
 * arrays of T generators.
 * ------------------------------------------------------------------------- */
int oracle_bpr_epoch_omp(oracle_mt19937 *rngs_pos, oracle_mt19937 *rngs_neg, int num_threads, uint64_t pos_hi,
                         uint64_t neg_hi, int64_t num_samples, const int32_t *user_ids, const int32_t *item_ids,
                         const int32_t *neg_item_ids, const int32_t *indptr, float *U, float *V, float *B, int k,
                         float lr, float reg, int use_bias, int64_t *correct_out, int64_t *skipped_out) {
    int64_t correct = 0, skipped = 0;
#pragma omp parallel num_threads(num_threads) reduction(+ : correct, skipped)
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        oracle_mt19937 *gp = &rngs_pos[t], *gn = &rngs_neg[t];
#pragma omp for schedule(guided)
        for (int64_t s = 0; s < num_samples; ++s) {
            int64_t i_index = oracle_boost_uniform(gp, pos_hi);
            int64_t j_index = oracle_boost_uniform(gn, neg_hi);
            int32_t i_id = item_ids[i_index];
            int32_t j_id = neg_item_ids[j_index];
            int32_t u_id = user_ids[i_index];
            if (has_non_zero(indptr, item_ids, u_id, j_id)) { ++skipped; continue; }
            correct += bpr_step(U + (int64_t)u_id * k, V + (int64_t)i_id * k, V + (int64_t)j_id * k, B, i_id, j_id,
                                k, lr, reg, use_bias);
        }
    }
    *correct_out = correct;
    *skipped_out = skipped;
    return 0;
}

/* ------------------------------------------------------------------------- *
 * VEBPR._fit_sgd_viewloss, seeded (num_threads == 1) — one epoch
 * (cornac/models/bpr/recom_vebpr.pyx:211-337).  Three engines: rng_pos over the
 * purchases, rng_view and rng_neg over [0, n_items-1].  Per sample: positive
 * (u, i); if the user has no views -> plain BPR step without biases, clamped
 * score (:243-268, draws rng_neg only); else view item v = views[u][draw %
 * num_view] (:270-271), negative j, skipped when j is purchased OR viewed
 * (:274-279), three clamped pairwise terms (:286-312) and a 4-row update
 * (:317-335).  The literal 1.0 in `(1.0 - alpha)` is a C double, so those
 * update terms are evaluated in double exactly as the generated C does; the
 * expressions below are written with the same operand types and order.
 * ------------------------------------------------------------------------- */
static inline float clamp50(float x) { return x > 50.0f ? 50.0f : (x < -50.0f ? -50.0f : x); }
static inline float sig_neg(float x) {
    float e = (float)exp((double)x);
    return (float)(1.0 / (1.0 + (double)e));
}

int oracle_vebpr_epoch_seq(oracle_mt19937 *rng_pos, oracle_mt19937 *rng_view, oracle_mt19937 *rng_neg,
                           int64_t num_samples, int32_t n_items, const int32_t *user_ids, const int32_t *p_indices,
                           const int32_t *p_indptr, const int32_t *v_indices, const int32_t *v_indptr, float *U,
                           float *V, int k, float lr, float reg, float alpha, int64_t *correct_out,
                           int64_t *skipped_out) {
    int64_t correct = 0, skipped = 0;
    const uint64_t pos_hi = (uint64_t)num_samples - 1, item_hi = (uint64_t)n_items - 1;
    for (int64_t s = 0; s < num_samples; ++s) {
        int64_t i_index = oracle_boost_uniform(rng_pos, pos_hi) % num_samples;
        int32_t u_id = user_ids[i_index], i_id = p_indices[i_index];
        int32_t num_view = v_indptr[u_id + 1] - v_indptr[u_id];
        float *user = U + (int64_t)u_id * k, *item_i = V + (int64_t)i_id * k;
        if (num_view == 0) {
            int32_t j_id = (int32_t)oracle_boost_uniform(rng_neg, item_hi);
            if (has_non_zero(p_indptr, p_indices, u_id, j_id)) { ++skipped; continue; }
            float *item_j = V + (int64_t)j_id * k;
            float x_uij = 0.0f;
            for (int f = 0; f < k; ++f) x_uij = x_uij + user[f] * (item_i[f] - item_j[f]);
            x_uij = clamp50(x_uij);
            float delta_ij = sig_neg(x_uij);
            if (delta_ij < 0.5f) ++correct;
            for (int f = 0; f < k; ++f) {
                float u_old = user[f], i_old = item_i[f], j_old = item_j[f];
                user[f] -= lr * (-delta_ij * (i_old - j_old) + reg * u_old);
                item_i[f] -= lr * (-delta_ij * u_old + reg * i_old);
                item_j[f] -= lr * (delta_ij * u_old + reg * j_old);
            }
            continue;
        }
        int64_t v_index = v_indptr[u_id] + (oracle_boost_uniform(rng_view, item_hi) % num_view);
        int32_t v_id = v_indices[v_index];
        int32_t j_id = (int32_t)oracle_boost_uniform(rng_neg, item_hi);
        if (has_non_zero(p_indptr, p_indices, u_id, j_id) || has_non_zero(v_indptr, v_indices, u_id, j_id)) {
            ++skipped;
            continue;
        }
        float *item_v = V + (int64_t)v_id * k, *item_j = V + (int64_t)j_id * k;
        float x_uij = 0.0f, x_uiv = 0.0f, x_uvj = 0.0f;
        for (int f = 0; f < k; ++f) {
            x_uij = x_uij + user[f] * (item_i[f] - item_j[f]);
            x_uiv = x_uiv + user[f] * (item_i[f] - item_v[f]);
            x_uvj = x_uvj + user[f] * (item_v[f] - item_j[f]);
        }
        x_uij = clamp50(x_uij); x_uiv = clamp50(x_uiv); x_uvj = clamp50(x_uvj);
        float delta_ij = sig_neg(x_uij), delta_iv = sig_neg(x_uiv), delta_vj = sig_neg(x_uvj);
        if (delta_ij < 0.5f && delta_iv < 0.5f && delta_vj < 0.5f) ++correct;
        for (int f = 0; f < k; ++f) {
            float u_old = user[f], i_old = item_i[f], v_old = item_v[f], j_old = item_j[f];
            user[f] -= lr * (-delta_ij * (i_old - j_old) - alpha * delta_iv * (i_old - v_old)
                             - (1.0 - alpha) * delta_vj * (v_old - j_old) + reg * u_old);
            item_i[f] -= lr * (-delta_ij * u_old - alpha * delta_iv * u_old + reg * i_old);
            item_v[f] -= lr * (alpha * delta_iv * u_old - (1.0 - alpha) * delta_vj * u_old + reg * v_old);
            item_j[f] -= lr * (delta_ij * u_old + (1.0 - alpha) * delta_vj * u_old + reg * j_old);
        }
    }
    *correct_out = correct;
    *skipped_out = skipped;
    return 0;
}

/* The float64 instantiation of the same fused-type function (recom_vebpr.pyx:219: `floating[:, :] U, V`; reached by
 * float64 init_params): every local of the step is a double, so are the clamps and 1.0 / (1.0 + exp(x)). */
static inline double clamp50d(double x) { return x > 50.0 ? 50.0 : (x < -50.0 ? -50.0 : x); }

int oracle_vebpr_epoch_seq_f64(oracle_mt19937 *rng_pos, oracle_mt19937 *rng_view, oracle_mt19937 *rng_neg,
                               int64_t num_samples, int32_t n_items, const int32_t *user_ids, const int32_t *p_indices,
                               const int32_t *p_indptr, const int32_t *v_indices, const int32_t *v_indptr, double *U,
                               double *V, int k, double lr, double reg, double alpha, int64_t *correct_out,
                               int64_t *skipped_out) {
    int64_t correct = 0, skipped = 0;
    const uint64_t pos_hi = (uint64_t)num_samples - 1, item_hi = (uint64_t)n_items - 1;
    for (int64_t s = 0; s < num_samples; ++s) {
        int64_t i_index = oracle_boost_uniform(rng_pos, pos_hi) % num_samples;
        int32_t u_id = user_ids[i_index], i_id = p_indices[i_index];
        int32_t num_view = v_indptr[u_id + 1] - v_indptr[u_id];
        double *user = U + (int64_t)u_id * k, *item_i = V + (int64_t)i_id * k;
        if (num_view == 0) {
            int32_t j_id = (int32_t)oracle_boost_uniform(rng_neg, item_hi);
            if (has_non_zero(p_indptr, p_indices, u_id, j_id)) { ++skipped; continue; }
            double *item_j = V + (int64_t)j_id * k;
            double x_uij = 0.0;
            for (int f = 0; f < k; ++f) x_uij = x_uij + user[f] * (item_i[f] - item_j[f]);
            x_uij = clamp50d(x_uij);
            double delta_ij = 1.0 / (1.0 + exp(x_uij));
            if (delta_ij < 0.5) ++correct;
            for (int f = 0; f < k; ++f) {
                double u_old = user[f], i_old = item_i[f], j_old = item_j[f];
                user[f] -= lr * (-delta_ij * (i_old - j_old) + reg * u_old);
                item_i[f] -= lr * (-delta_ij * u_old + reg * i_old);
                item_j[f] -= lr * (delta_ij * u_old + reg * j_old);
            }
            continue;
        }
        int64_t v_index = v_indptr[u_id] + (oracle_boost_uniform(rng_view, item_hi) % num_view);
        int32_t v_id = v_indices[v_index];
        int32_t j_id = (int32_t)oracle_boost_uniform(rng_neg, item_hi);
        if (has_non_zero(p_indptr, p_indices, u_id, j_id) || has_non_zero(v_indptr, v_indices, u_id, j_id)) {
            ++skipped;
            continue;
        }
        double *item_v = V + (int64_t)v_id * k, *item_j = V + (int64_t)j_id * k;
        double x_uij = 0.0, x_uiv = 0.0, x_uvj = 0.0;
        for (int f = 0; f < k; ++f) {
            x_uij = x_uij + user[f] * (item_i[f] - item_j[f]);
            x_uiv = x_uiv + user[f] * (item_i[f] - item_v[f]);
            x_uvj = x_uvj + user[f] * (item_v[f] - item_j[f]);
        }
        x_uij = clamp50d(x_uij); x_uiv = clamp50d(x_uiv); x_uvj = clamp50d(x_uvj);
        double delta_ij = 1.0 / (1.0 + exp(x_uij)), delta_iv = 1.0 / (1.0 + exp(x_uiv)), delta_vj = 1.0 / (1.0 + exp(x_uvj));
        if (delta_ij < 0.5 && delta_iv < 0.5 && delta_vj < 0.5) ++correct;
        for (int f = 0; f < k; ++f) {
            double u_old = user[f], i_old = item_i[f], v_old = item_v[f], j_old = item_j[f];
            user[f] -= lr * (-delta_ij * (i_old - j_old) - alpha * delta_iv * (i_old - v_old)
                             - (1.0 - alpha) * delta_vj * (v_old - j_old) + reg * u_old);
            item_i[f] -= lr * (-delta_ij * u_old - alpha * delta_iv * u_old + reg * i_old);
            item_v[f] -= lr * (alpha * delta_iv * u_old - (1.0 - alpha) * delta_vj * u_old + reg * v_old);
            item_j[f] -= lr * (delta_ij * u_old + (1.0 - alpha) * delta_vj * u_old + reg * j_old);
        }
    }
    *correct_out = correct;
    *skipped_out = skipped;
    return 0;
}

/* ------------------------------------------------------------------------- *
 * backend_cpu.fit_sgd (cornac/models/mf/backend_cpu.pyx:35-97): all epochs in
 * one call, COO order, in-place SGD; loss_per_epoch[e] = 0.5*sum(err^2) with
 * the float accumulator the reference uses; early stop on |dloss| < 1e-5.
 * Returns the number of epochs run.  num_threads > 1 = racy static-chunk
 * Hogwild (`prange(..., schedule='static')`, :62).
 * ------------------------------------------------------------------------- */
int oracle_mf_fit(const int64_t *rid, const int64_t *cid, const float *val, int64_t nnz, float *U, float *V,
                  float *Bu, float *Bi, int k, float lr, float reg, float mu, int max_iter, int num_threads,
                  int use_bias, int early_stop, float *loss_per_epoch) {
    float loss = 0.f, last_loss = 0.f;
    int epoch = 0;
    for (; epoch < max_iter; ++epoch) {
        last_loss = loss;
        loss = 0.f;
        if (num_threads <= 1) {
            for (int64_t j = 0; j < nnz; ++j) {
                int64_t u = rid[j], i = cid[j];
                float r = val[j];
                float *user = U + u * k, *item = V + i * k;
                float r_pred = mu + Bu[u] + Bi[i];
                for (int f = 0; f < k; ++f) r_pred = r_pred + user[f] * item[f];
                float error = r - r_pred;
                loss += error * error;
                for (int f = 0; f < k; ++f) {
                    float u_f = user[f], i_f = item[f];
                    user[f] += lr * (error * i_f - reg * u_f);
                    item[f] += lr * (error * u_f - reg * i_f);
                }
                if (use_bias) {
                    Bu[u] += lr * (error - reg * Bu[u]);
                    Bi[i] += lr * (error - reg * Bi[i]);
                }
            }
        } else {
            float lsum = 0.f;
#pragma omp parallel for schedule(static) num_threads(num_threads) reduction(+ : lsum)
            for (int64_t j = 0; j < nnz; ++j) {
                int64_t u = rid[j], i = cid[j];
                float r = val[j];
                float *user = U + u * k, *item = V + i * k;
                float r_pred = mu + Bu[u] + Bi[i];
                for (int f = 0; f < k; ++f) r_pred = r_pred + user[f] * item[f];
                float error = r - r_pred;
                lsum += error * error;
                for (int f = 0; f < k; ++f) {
                    float u_f = user[f], i_f = item[f];
                    user[f] += lr * (error * i_f - reg * u_f);
                    item[f] += lr * (error * u_f - reg * i_f);
                }
                if (use_bias) {
                    Bu[u] += lr * (error - reg * Bu[u]);
                    Bi[i] += lr * (error - reg * Bi[i]);
                }
            }
            loss = lsum;
        }
        loss = 0.5f * loss;
        if (loss_per_epoch) loss_per_epoch[epoch] = loss;
        float delta = loss - last_loss;
        if (early_stop && fabsf(delta) < 1e-5f) { ++epoch; break; }
    }
    return epoch;
}

/* ------------------------------------------------------------------------- *
 * fast_dot (cornac/utils/fast_dot.pyx:25-43): output[i] += dot(vec, mat[i]).
 * The reference calls BLAS sdot, whose summation order is implementation
 * defined; the oracle fixes it: mode 0 = index-order mul+add, mode 1 =
 * index-order fmaf chain (bit-identical to the gfx950 fp32 MFMA / v_fmac
 * accumulation used by the HIP scoring kernels), mode 2 = fp64 accumulate.
 * ------------------------------------------------------------------------- */
void oracle_fast_dot(const float *vec, const float *mat, float *output, int64_t n_rows, int k, int mode) {
    for (int64_t i = 0; i < n_rows; ++i) {
        const float *row = mat + i * k;
        if (mode == 0) {
            float acc = 0.f;
            for (int f = 0; f < k; ++f) acc = acc + vec[f] * row[f];
            output[i] += acc;
        } else if (mode == 1) {
            float acc = 0.f;
            for (int f = 0; f < k; ++f) acc = fmaf(vec[f], row[f], acc);
            output[i] += acc;
        } else {
            double acc = 0.0;
            for (int f = 0; f < k; ++f) acc += (double)vec[f] * (double)row[f];
            output[i] = (float)((double)output[i] + acc);
        }
    }
}

/* scores for a block of users: out[b, i] = base[i] + ubase[b] + dot_fma(U[users[b]], V[i])
 * = BPR.score (recom_bpr.pyx:288-291: copy(B) then fast_dot) and MF.score
 * (recom_mf.py:273-278: mu + Bi (+ Bu[u] + fast_dot)), accumulation mode 1
 * with the bias added AFTER the dot product like fast_dot's `output[i] +=`. */
void oracle_score_block(const float *U, const float *V, const float *item_base, const float *user_base,
                        const int32_t *users, int64_t n_block, int64_t n_items, int k, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < n_block; ++b) {
        const float *u = U + (int64_t)users[b] * k;
        float ub = user_base ? user_base[users[b]] : 0.f;
        for (int64_t i = 0; i < n_items; ++i) {
            const float *row = V + i * k;
            float acc = 0.f;
            for (int f = 0; f < k; ++f) acc = fmaf(u[f], row[f], acc);
            float base = item_base ? item_base[i] : 0.f;
            out[b * n_items + i] = (base + ub) + acc;
        }
    }
}

/* Hogwild-mode pair sampler of the HIP throughput kernel, restated on the CPU
 * so tests can check the device sampler bit-exactly (this one has no
 * reference counterpart: the reference's multi-thread streams are not
 * reproducible; see DESIGN.md "hogwild sampler").  Philox4x32-10, key =
 * (seed_lo, seed_hi), counter = (sample_lo, sample_hi, epoch, stream).
 * Bounded draw = Lemire multiply-shift, words (0,1) for the positive index and
 * (2,3) for the negative index. */
static inline void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

void oracle_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                       uint32_t out[4]) {
    uint32_t c[4] = {c0, c1, c2, c3};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    memcpy(out, c, sizeof(uint32_t) * 4);
}

static inline uint32_t lemire_bounded2(uint32_t wa, uint32_t wb, uint32_t n) {
    /* uniform in [0, n): multiply-shift; if the low word of the first product falls
     * in the biased zone (< 2^32 mod n) the second word is used unconditionally
     * (residual bias < (n/2^32)^2). */
    uint32_t thresh = (uint32_t)(-n) % n;
    uint64_t m = (uint64_t)wa * n;
    if ((uint32_t)m < thresh) m = (uint64_t)wb * n;
    return (uint32_t)(m >> 32);
}

void oracle_hogwild_sample(uint64_t seed, uint32_t epoch, int64_t s0, int64_t n, uint32_t n_pos, uint32_t n_neg,
                           int64_t *ii_out, int64_t *jj_out) {
    for (int64_t t = 0; t < n; ++t) {
        uint64_t s = (uint64_t)(s0 + t);
        uint32_t w[4];
        oracle_philox4x32((uint32_t)s, (uint32_t)(s >> 32), epoch, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
        ii_out[t] = lemire_bounded2(w[0], w[1], n_pos);
        jj_out[t] = lemire_bounded2(w[2], w[3], n_neg);
    }
}

/* ownership variant of the device sampler (bpr.hip: hog_sample_owned): wave `wave_id` draws its
 * local samples [lo, hi) of an epoch from its own slice of length `len`; counter =
 * (local, wave_id, epoch, 1). */
void oracle_hogwild_sample_owned(uint64_t seed, uint32_t epoch, uint32_t wave_id, uint32_t len, uint32_t n_neg,
                                 int64_t lo, int64_t hi, int64_t *r_out, int64_t *jj_out) {
    for (int64_t t = lo; t < hi; ++t) {
        uint32_t w[4];
        oracle_philox4x32((uint32_t)t, wave_id, epoch, 1u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
        r_out[t - lo] = lemire_bounded2(w[0], w[1], len);
        jj_out[t - lo] = lemire_bounded2(w[2], w[3], n_neg);
    }
}

/* XCD-strata variant of the device sampler (csrc/bpr_strata.inc; like the two above it restates the HIP
 * path's own integer arithmetic — the reference's multi-thread streams are not reproducible).
 * strata_rot: rotation of rank group g's 8 items over the 8 item partitions under an epoch key;
 * strata_key: the key of epoch e; strata_sample: the draws of wave `wave_id` in partition p, whose bucket
 * holds `len` interactions: r_out = index into the bucket, code_out = popularity rank of the negative item
 * (item = rank_item[code]); counter = (local, wave_id, epoch, 0x10 | p). */
uint32_t oracle_strata_rot(uint32_t g, uint32_t key) {
    uint32_t h = g * 0x9E3779B1u + key;
    h ^= h >> 15; h *= 0x85EBCA77u;
    h ^= h >> 13; h *= 0xC2B2AE3Du;
    h ^= h >> 16;
    return h & 7u;
}

uint32_t oracle_strata_key(uint64_t seed, uint32_t epoch) {
    uint32_t w[4];
    oracle_philox4x32(epoch, 0x57A7Au, 0u, 3u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    return w[0];
}

/* partition of every popularity rank under `key` */
void oracle_strata_partitions(uint32_t key, int64_t n_items, uint8_t *part_of_rank) {
    for (int64_t c = 0; c < n_items; ++c)
        part_of_rank[c] = (uint8_t)((((uint32_t)c & 7u) + oracle_strata_rot((uint32_t)(c >> 3), key)) & 7u);
}

int64_t oracle_strata_sample(uint64_t seed, uint32_t epoch, uint32_t key, uint32_t wave_id, uint32_t p, uint32_t len,
                             uint32_t n_items, int64_t *r_out, int64_t *code_out) {
    const uint32_t n_full = n_items >> 3, rem = n_items & 7u;
    uint32_t n_p = n_full;
    if (rem && ((p - oracle_strata_rot(n_full, key)) & 7u) < rem) n_p += 1;
    if (n_p == 0) return 0;
    for (uint32_t t = 0; t < len; ++t) {
        uint32_t w[4];
        oracle_philox4x32(t, wave_id, epoch, 0x10u | p, (uint32_t)seed, (uint32_t)(seed >> 32), w);
        r_out[t] = lemire_bounded2(w[0], w[1], len);
        const uint32_t g = lemire_bounded2(w[2], w[3], n_p);
        code_out[t] = (int64_t)g * 8 + ((p - oracle_strata_rot(g, key)) & 7u);
    }
    return (int64_t)len;
}

/* LDS-bin variant of the device sampler (csrc/bpr_ldsbin.inc; restates the HIP path's own integer arithmetic).
 * The deal of an epoch key: n_groups = ceil(n_items / n_bins) groups of n_bins positions; the groups are cut into
 * n_strata strata [s n_groups / n_strata, (s + 1) n_groups / n_strata); position p of stratum s holds the item of
 * popularity rank base_s + perm_s(p - base_s) (perm = a keyed Feistel bijection of the stratum's ranks, cycle-walked) and
 * belongs to bin (p mod n_bins + rot(p / n_bins)) mod n_bins.  Ranks < n_hot are hot: no local positives; their
 * interactions (a list shuffled once by the stable order of a hash of its index) are dealt to the bins in contiguous
 * runs [hot_off[b], hot_off[b + 1]) whose lengths level the bins (oracle_ldsbin_deal).
 * Draw `local` of bin b: counter (local, b, epoch, 0x20); words (0,1) pick the positive among cold_mass + hot_share
 * interactions, words (2,3) the negative among the bin's OTHER slots (all slots for a hot positive).  Returns the number of draws whose negative is a
 * positive of the user (the device's skip counter for the epoch) and the total number of draws. */
static uint32_t ldsbin_mix(uint32_t h) {
    h ^= h >> 15; h *= 0x85EBCA77u;
    h ^= h >> 13; h *= 0xC2B2AE3Du;
    h ^= h >> 16;
    return h;
}

static uint32_t ldsbin_rot(uint32_t g, uint32_t key, uint32_t n_bins) { return ldsbin_mix(g * 0x9E3779B1u + key) % n_bins; }

uint32_t oracle_ldsbin_hot_shuffle_key(uint32_t t) { return ldsbin_mix(t * 0x9E3779B1u + 0x5BD1E995u); }

static uint32_t ldsbin_perm(uint32_t x, uint32_t n, uint32_t key) {
    if (n <= 1u) return 0u;
    int bits = 0;
    while (bits < 32 && (n - 1u) >> bits) ++bits;
    if (bits < 2) bits = 2;
    const int lb = bits / 2, rb = bits - lb;
    const uint32_t lm = (1u << lb) - 1u, rm = (1u << rb) - 1u;
    do {
        uint32_t l = x >> rb, r = x & rm;
        for (uint32_t round = 0; round < 3u; ++round) {
            l ^= (ldsbin_mix(r * 0x9E3779B1u + key + round * 0x7F4A7C15u) >> 7) & lm;
            r ^= (ldsbin_mix(l * 0x85EBCA6Bu + (key ^ 0x5BD1E995u) + round * 0x165667B1u) >> 7) & rm;
        }
        x = (l << rb) | r;
    } while (x >= n);
    return x;
}

static uint32_t ldsbin_deal_rank(uint32_t p, uint32_t key, uint32_t n_bins, uint32_t n_items, uint32_t n_groups,
                                 uint32_t n_strata) {
    const uint32_t g = p / n_bins;
    uint32_t s = (uint32_t)(((uint64_t)g * n_strata) / n_groups);
    if ((uint32_t)(((uint64_t)(s + 1u) * n_groups) / n_strata) <= g) ++s;
    const uint32_t g_lo = (uint32_t)(((uint64_t)s * n_groups) / n_strata);
    const uint32_t g_hi = (uint32_t)(((uint64_t)(s + 1u) * n_groups) / n_strata);
    const uint32_t base = g_lo * n_bins;
    const uint64_t end = (uint64_t)g_hi * n_bins;
    const uint32_t size = (uint32_t)(end < n_items ? end : n_items) - base;
    return base + ldsbin_perm(p - base, size, key ^ ldsbin_mix(s * 0x27D4EB2Fu + 0x165667B1u));
}

static uint64_t level_share(uint64_t level, uint32_t cold, uint32_t c16) {
    const uint64_t t = 16ull * cold;
    return level > t ? (level - t) / c16 : 0ull;
}

/* The bookkeeping of one deal: bin_of_item [n_items] (may be NULL), cold_mass [n_bins], hot_off [n_bins + 1].  A cold
 * draw costs 16, a hot one hot_cost_x16: level = the largest L with sum_b floor(max(0, L - 16 cold_b) / cost) <= H; the
 * rest is dealt round robin from bin 0 (csrc/bpr_ldsbin.inc ldsbin_level_kernel). */
void oracle_ldsbin_deal(uint32_t key, uint32_t n_bins, uint32_t n_items, uint32_t n_hot, uint32_t n_strata,
                        uint32_t hot_cost_x16, const int32_t *rank_item, const int32_t *cptr, uint32_t n_hot_inter,
                        int32_t *bin_of_item, uint32_t *cold_mass, uint32_t *hot_off) {
    const uint32_t n_groups = (n_items + n_bins - 1) / n_bins;
    for (uint32_t b = 0; b < n_bins; ++b) cold_mass[b] = 0;
    for (uint32_t p = 0; p < n_items; ++p) {
        const uint32_t g = p / n_bins, o = p % n_bins;
        const uint32_t code = ldsbin_deal_rank(p, key, n_bins, n_items, n_groups, n_strata);
        const uint32_t b = (o + ldsbin_rot(g, key, n_bins)) % n_bins;
        const int32_t it = rank_item[code];
        if (bin_of_item) bin_of_item[it] = (int32_t)b;
        if (code >= n_hot) cold_mass[b] += (uint32_t)(cptr[it + 1] - cptr[it]);
    }
    uint64_t level = 0, given = 0;
    if (hot_cost_x16 && n_hot_inter) {
        uint64_t tot = 0;
        for (uint32_t b = 0; b < n_bins; ++b) tot += cold_mass[b];
        uint64_t lo = 0, hi = 16ull * tot + (uint64_t)hot_cost_x16 * ((uint64_t)n_hot_inter + 1ull);
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo + 1) / 2;
            uint64_t sum = 0;
            for (uint32_t b = 0; b < n_bins; ++b) sum += level_share(mid, cold_mass[b], hot_cost_x16);
            if (sum <= n_hot_inter) lo = mid; else hi = mid - 1;
        }
        level = lo;
        for (uint32_t b = 0; b < n_bins; ++b) given += level_share(level, cold_mass[b], hot_cost_x16);
    }
    const uint64_t rest = (uint64_t)n_hot_inter - given;
    hot_off[0] = 0;
    for (uint32_t b = 0; b < n_bins; ++b) {
        const uint32_t share = (uint32_t)((hot_cost_x16 && n_hot_inter) ? level_share(level, cold_mass[b], hot_cost_x16) : 0ull) +
                               (uint32_t)(rest / n_bins) + (b < (uint32_t)(rest % n_bins) ? 1u : 0u);
        hot_off[b + 1] = hot_off[b] + share;
    }
}

/* The (bin, slot) layout of a deal (csrc/bpr_ldsbin.inc ldsbin_layout_kernel; the conveyor's block buffers hold their rows in
 * this order): slot_item[b cap + g] = the item dealt to slot g of bin b, -1 where the last group has none; item_slot = its
 * inverse.  cap = ceil(n_items / n_bins). */
void oracle_ldsbin_layout(uint32_t key, uint32_t n_bins, uint32_t n_items, uint32_t n_strata, const int32_t *rank_item,
                          int32_t *slot_item, int32_t *item_slot) {
    const uint32_t n_groups = (n_items + n_bins - 1) / n_bins;
    for (uint64_t p = 0; p < (uint64_t)n_groups * n_bins; ++p) {
        const uint32_t g = (uint32_t)(p / n_bins), o = (uint32_t)(p % n_bins);
        const uint32_t b = (o + ldsbin_rot(g, key, n_bins)) % n_bins;
        const uint64_t slot = (uint64_t)b * n_groups + g;
        if (p < n_items) {
            const int32_t it = rank_item[ldsbin_deal_rank((uint32_t)p, key, n_bins, n_items, n_groups, n_strata)];
            slot_item[slot] = it;
            item_slot[it] = (int32_t)slot;
        } else {
            slot_item[slot] = -1;
        }
    }
}

uint32_t oracle_ldsbin_key(uint64_t seed, uint32_t epoch) {
    uint32_t w[4];
    oracle_philox4x32(epoch, 0x1D5B1Au, 0u, 4u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    return w[0];
}

static int csr_has(const int32_t *indices, int32_t lo, int32_t hi, int32_t col) {
    const int32_t end = hi;
    while (lo < hi) {
        const int32_t mid = lo + ((hi - lo) >> 1);
        if (indices[mid] < col) lo = mid + 1; else hi = mid;
    }
    return lo < end && indices[lo] == col;
}

int64_t oracle_ldsbin_epoch_skips_range(uint64_t seed, uint32_t epoch, uint32_t key, uint32_t n_bins, uint32_t n_items,
                                        uint32_t n_hot, uint32_t n_strata, uint32_t hot_cost_x16, const int32_t *rank_item,
                                        const int32_t *cptr, const int32_t *cusers,
                                        const int32_t *hot_u, const int32_t *hot_i, uint32_t n_hot_inter, const int32_t *indptr,
                                        const int32_t *indices, int64_t *n_draws_out, int64_t *pos_count, int64_t *neg_count,
                                        int neg_pop, uint32_t b_lo, uint32_t b_hi);

int64_t oracle_ldsbin_epoch_skips(uint64_t seed, uint32_t epoch, uint32_t key, uint32_t n_bins, uint32_t n_items,
                                  uint32_t n_hot, uint32_t n_strata, uint32_t hot_cost_x16, const int32_t *rank_item,
                                  const int32_t *cptr, const int32_t *cusers,
                                  const int32_t *hot_u, const int32_t *hot_i, uint32_t n_hot_inter, const int32_t *indptr,
                                  const int32_t *indices, int64_t *n_draws_out, int64_t *pos_count, int64_t *neg_count,
                                  int neg_pop) {
    return oracle_ldsbin_epoch_skips_range(seed, epoch, key, n_bins, n_items, n_hot, n_strata, hot_cost_x16, rank_item, cptr, cusers,
                                           hot_u, hot_i, n_hot_inter, indptr, indices, n_draws_out, pos_count, neg_count, neg_pop,
                                           0u, n_bins);
}

/* the same for the bins [b_lo, b_hi) only (a conveyor block's launch; a sample of the bins at sizes where the whole epoch
 * would take minutes on one core) */
int64_t oracle_ldsbin_epoch_skips_range(uint64_t seed, uint32_t epoch, uint32_t key, uint32_t n_bins, uint32_t n_items,
                                        uint32_t n_hot, uint32_t n_strata, uint32_t hot_cost_x16, const int32_t *rank_item,
                                        const int32_t *cptr, const int32_t *cusers,
                                        const int32_t *hot_u, const int32_t *hot_i, uint32_t n_hot_inter, const int32_t *indptr,
                                        const int32_t *indices, int64_t *n_draws_out, int64_t *pos_count, int64_t *neg_count,
                                        int neg_pop, uint32_t b_lo, uint32_t b_hi) {
    const uint32_t n_groups = (n_items + n_bins - 1) / n_bins;
    uint32_t *cold_all = (uint32_t *)malloc(sizeof(uint32_t) * n_bins);
    uint32_t *hot_off = (uint32_t *)malloc(sizeof(uint32_t) * (n_bins + 1));
    oracle_ldsbin_deal(key, n_bins, n_items, n_hot, n_strata, hot_cost_x16, rank_item, cptr, n_hot_inter, NULL, cold_all,
                       hot_off);
    int32_t *item = (int32_t *)malloc(sizeof(int32_t) * n_groups);
    int32_t *cp = (int32_t *)malloc(sizeof(int32_t) * n_groups);
    uint32_t *cum = (uint32_t *)malloc(sizeof(uint32_t) * (n_groups + 1));
    uint8_t *hot = (uint8_t *)malloc(n_groups);
    int64_t skipped = 0, total = 0;
    for (uint32_t b = b_lo; b < b_hi && b < n_bins; ++b) {
        uint32_t n_slots = n_groups;
        cum[0] = 0;
        for (uint32_t g = 0; g < n_groups; ++g) {
            const uint32_t pos = g * n_bins + (b + n_bins - ldsbin_rot(g, key, n_bins)) % n_bins;
            uint32_t d = 0;
            item[g] = -1; cp[g] = 0; hot[g] = 0;
            if (pos < n_items) {
                const uint32_t code = ldsbin_deal_rank(pos, key, n_bins, n_items, n_groups, n_strata);
                item[g] = rank_item[code];
                cp[g] = cptr[item[g]];
                hot[g] = code < n_hot;
                if (!hot[g]) d = (uint32_t)(cptr[item[g] + 1] - cp[g]);
            }
            cum[g + 1] = cum[g] + d;
        }
        if (n_groups && item[n_groups - 1] == -1) n_slots = n_groups - 1;
        const uint32_t cold_mass = cum[n_groups];
        const uint32_t hot_lo = hot_off[b], hot_share = hot_off[b + 1] - hot_lo;
        const uint32_t n_draws = n_slots ? cold_mass + hot_share : 0u;
        if (!n_slots) skipped += cold_mass + hot_share;
        total += cold_mass + hot_share;
        for (uint32_t local = 0; local < n_draws; ++local) {
            uint32_t w[4];
            oracle_philox4x32(local, b, epoch, 0x20u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
            const uint32_t r_pos = lemire_bounded2(w[0], w[1], n_draws);
            uint32_t s_j, excl = 0, excl_lo = 0xffffffffu;
            int32_t u, i;
            if (r_pos < cold_mass) {
                uint32_t lo = 0, hi = n_slots;
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (cum[mid] <= r_pos) lo = mid; else hi = mid;
                }
                i = item[lo];
                u = cusers[cp[lo] + (int32_t)(r_pos - cum[lo])];
                s_j = n_slots > 1 ? lemire_bounded2(w[2], w[3], n_slots - 1) : 0u;  /* the other slots of the bin */
                if (n_slots > 1 && s_j >= lo) ++s_j;
                excl_lo = cum[lo];
                excl = cum[lo + 1] - excl_lo;
            } else {
                const uint32_t h = hot_lo + (r_pos - cold_mass);
                u = hot_u[h];
                i = hot_i[h];
                s_j = lemire_bounded2(w[2], w[3], n_slots);
            }
            int32_t j = item[s_j];
            if (neg_pop) { /* WBPR: the item of a second interaction of the bin's draw space, the positive slot's left out */
                const uint32_t m = n_draws - excl;
                uint32_t r_neg = (uint32_t)(((uint64_t)w[2] * m) >> 32);
                if (r_neg >= excl_lo) r_neg += excl;
                if (m == 0) {
                    j = i;
                } else if (r_neg < cold_mass) {
                    uint32_t lo = 0, hi = n_slots;
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (cum[mid] <= r_neg) lo = mid; else hi = mid;
                    }
                    j = item[lo];
                } else {
                    j = hot_i[hot_lo + (r_neg - cold_mass)];
                }
            }
            if (csr_has(indices, indptr[u], indptr[u + 1], j)) { ++skipped; continue; }
            if (pos_count) ++pos_count[i];
            if (neg_count) ++neg_count[j];
        }
    }
    free(item); free(cp); free(cum); free(hot); free(cold_all); free(hot_off);
    if (n_draws_out) *n_draws_out = total;
    return skipped;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int oracle_sizeof_mt(void) { return (int)sizeof(oracle_mt19937); }
