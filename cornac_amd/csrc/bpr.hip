// BPR / WBPR training on MI355X (gfx950): triplet sampler + pairwise-loss SGD.
//
// Replaces the reference's BPR._fit_sgd hot loop (cornac/models/bpr/recom_bpr.pyx:208-269), its
// sampler RNGVector (:54-62) and the CSR membership test has_non_zero (:46-51).
//
// Two execution modes:
//   deterministic — reproduces the seeded (single-thread) reference: bit-faithful mt19937/boost
//                   draw streams on the device, then the nnz sequential updates are executed as a
//                   level schedule of the row-conflict DAG (levels = kernels, samples of a level
//                   touch disjoint rows), with the reference's float expression order.
//   hogwild       — throughput mode (the reference's num_threads > 1 path): counter-based
//                   sampling, one 64-sample tile per wave compacted through LDS, then row-wise lanes
//                   (the lanes of a group own consecutive floats of a row: fully coalesced gathers, one
//                   atomic instruction per row), wave-shuffle dot products, user rows owned by waves
//                   (plain stores) and fp32 device-scope atomics on the item side.  Bound by the atomic
//                   line-touch rate of the L2s; no MFMA (nothing is GEMM-shaped).
// sharded.inc adds the sample / gather / apply / scatter-add pieces of the row-sharded multi-GPU regime.
#include <algorithm>
#include <numeric>
#include <queue>

#include "common.h"
#include "rng.h"
#include "sgd_device.h"

namespace chip {

// ------------------------------------------------------------------------------------------------
// deterministic mode: sampler -> (u, i, j) per sample
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void bpr_det_sample_kernel(
    const uint32_t *__restrict__ pos_draw, int pos_stride, const uint32_t *__restrict__ neg_draw, int neg_stride,
    int64_t n, const int32_t *__restrict__ user_ids, const int32_t *__restrict__ indices,
    const int32_t *__restrict__ indptr, int neg_population, int32_t *__restrict__ su, int32_t *__restrict__ si,
    int32_t *__restrict__ sj, unsigned long long *__restrict__ counters) {
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    bool skip = false;
    if (s < n) {
        const uint32_t ii = pos_draw[s * pos_stride];
        const uint32_t jj = neg_draw[s * neg_stride];
        const int32_t u = user_ids[ii];
        const int32_t i = indices[ii];
        const int32_t j = neg_population == CORNAC_HIP_NEG_POPULARITY ? indices[jj] : (int32_t)jj;
        skip = csr_row_contains(indices, indptr[u], indptr[u + 1], j);
        su[s] = skip ? -1 : u;
        si[s] = i;
        sj[s] = j;
    }
    const unsigned long long m = __ballot(skip);
    if (lane_id() == 0 && m) atomicAdd(&counters[1], (unsigned long long)__popcll(m));
}

// One level of the conflict-free schedule: samples [off, off+cnt) touch pairwise-disjoint rows.
// G lanes per triplet.  The dot product is accumulated in index order (score = B_i - B_j, then
// += u_f * (vi_f - vj_f) for f = 0..k-1) so the result is bit-identical to the sequential oracle.
// T = float (the reference's default tables) or double: `_fit_sgd` is a fused-type function (recom_bpr.pyx:211-214) and
// runs in double when the model was given float64 factors through init_params — every local is then a double (:219-224)
template <int G, class T>
__global__ __launch_bounds__(kBlock) void bpr_det_level_kernel(const int32_t *__restrict__ su,
                                                               const int32_t *__restrict__ si,
                                                               const int32_t *__restrict__ sj, int64_t off, int cnt,
                                                               T *U, T *V, T *B, int k, T lr,
                                                               T reg, int use_bias,
                                                               unsigned long long *__restrict__ counters) {
    const int gid = (blockIdx.x * kBlock + threadIdx.x) / G;
    const int lg = threadIdx.x & (G - 1);
    const bool active = gid < cnt;
    const int64_t t = off + (active ? gid : cnt - 1);
    const int32_t u = su[t], i = si[t], j = sj[t];
    T *pu = U + (size_t)u * k, *pi = V + (size_t)i * k, *pj = V + (size_t)j * k;
    const T bi = B[i], bj = B[j];
    T score = bi - bj;
    // rows are read ONCE (kept in registers for the update when k <= 4 G): a level's latency is its chain of
    // dependent memory round trips, ids -> rows -> stores
    constexpr int RMAX = 4;
    T ru[RMAX], ri[RMAX], rj[RMAX];
    const bool in_regs = k <= RMAX * G;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int f = r * G + lg;
        ru[r] = ri[r] = rj[r] = T(0);
        if (in_regs && f < k) {
            ru[r] = pu[f];
            ri[r] = pi[f];
            rj[r] = pj[f];
        }
    }
    auto ordered_add = [&](T p, int lim) { score = ordered_lane_sum_t<G>(score, p, lim); };
    if (in_regs) {
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int base = r * G;
            if (base < k) ordered_add(base + lg < k ? ru[r] * (ri[r] - rj[r]) : T(0), min(G, k - base));
        }
    } else {
        for (int base = 0; base < k; base += G) {
            const int f = base + lg;
            ordered_add(f < k ? pu[f] * (pi[f] - pj[f]) : T(0), min(G, k - base));
        }
    }
    const T z = sigmoid_neg_exact_t(score);
    if (active) {
        if (in_regs) {
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                const int f = r * G + lg;
                if (f < k) {
                    const T uf = ru[r], vi = ri[r], vj = rj[r];
                    pu[f] = uf + lr * (z * (vi - vj) - reg * uf);
                    pi[f] = vi + lr * (z * uf - reg * vi);
                    pj[f] = vj + lr * (-z * uf - reg * vj);
                }
            }
        } else {
            for (int f = lg; f < k; f += G) {
                const T uf = pu[f], vi = pi[f], vj = pj[f];
                pu[f] = uf + lr * (z * (vi - vj) - reg * uf);
                pi[f] = vi + lr * (z * uf - reg * vi);
                pj[f] = vj + lr * (-z * uf - reg * vj);
            }
        }
        if (lg == 0 && use_bias) {
            B[i] = bi + lr * (z - reg * bi);
            B[j] = bj + lr * (-z - reg * bj);
        }
    }
    const unsigned long long m = __ballot(active && lg == 0 && z < T(.5));
    if (lane_id() == 0 && m) atomicAdd(&counters[0], (unsigned long long)__popcll(m));
}

// bucket the sampled triplets by level on the device (they are already here): cursor[l] starts at level_ptr[l];
// the order inside a level is irrelevant (its samples touch disjoint rows)
__global__ __launch_bounds__(kBlock) void bpr_det_bucket_kernel(const int32_t *__restrict__ level,
                                                                const int32_t *__restrict__ su,
                                                                const int32_t *__restrict__ si,
                                                                const int32_t *__restrict__ sj, int64_t n,
                                                                unsigned long long *__restrict__ cursor,
                                                                int32_t *__restrict__ ou, int32_t *__restrict__ oi,
                                                                int32_t *__restrict__ oj) {
    for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s < n; s += (int64_t)gridDim.x * kBlock) {
        const int32_t l = level[s];
        if (l <= 0) continue;
        const unsigned long long pos = atomicAdd(&cursor[l], 1ull);
        ou[pos] = su[s];
        oi[pos] = si[s];
        oj[pos] = sj[s];
    }
}

// ------------------------------------------------------------------------------------------------
// hogwild mode
// ------------------------------------------------------------------------------------------------
struct HogArgs {
    const int32_t *user_ids, *indices, *indptr;
    const int32_t *neg_items;  // population of the popularity-weighted negative draw: `indices` (recom_wbpr.pyx:135), or the
                               // caller's (cornac_hip_bpr_set_negative_population: the GLOBAL popularity over a rank's slice)
    float *U, *V, *B;
    unsigned long long *counters;
    int64_t n;        // samples in this launch
    uint64_t s_begin; // sample counter of the first one
    uint64_t seed;
    uint32_t epoch;
    uint32_t n_pos, n_neg, th_pos, th_neg;
    int k, neg_population, use_bias;
    float lr, reg;
    // user-row ownership (see build_ownership): wave w samples positives only from
    // own_u/own_i[wave_ptr[w] .. wave_ptr[w+1]); own_u < 0 encodes a shared (heavy) user as ~u
    const int32_t *own_u, *own_i;
    const int64_t *wave_ptr;
    int64_t own_tmax;  // tiles of 64 samples of the longest wave slice
    int64_t nnz;
    int bstride;  // element stride of the (padded) bias table handed to the hogwild kernels
    int vpitch;   // element pitch of the item table (k, or the record pitch of the strata form's packed table)
    int ablate;  // profiling-only switches (hogwild_flags bits 8..): see DESIGN.md "ablations"
    // row-sharded item table (sharded.inc): the owned kernel in its EMIT / STAGED forms exchanges the triplets of
    // tile t of wave w through trip_*[((trip_tile0 + t) * total_waves + w) * 64 + lane]
    int32_t *trip_u, *trip_i, *trip_j;
    int64_t trip_tile0, trip_tiles;
    // XCD strata (bpr_strata.inc): per-wave partition buckets of the ownership slices, rank -> item, epoch key
    const int32_t *rec_u, *rec_i, *rank_item;
    const int64_t *sptr;
    uint32_t strata_key, n_hot;
    uint32_t *xcd_claim;  // [9] per-XCD slot counters + the surplus ticket, zero at launch (bpr_strata.inc)
    int phase;
};
constexpr int32_t kTripShared = 0x40000000;  // bit 30 of an emitted user id: a shared (heavy) user, atomics on its row

// per-lane: draw one (u, i, j) and test membership; returns validity
__device__ __forceinline__ bool hog_sample(const HogArgs &a, int64_t local, int32_t &u, int32_t &i, int32_t &j,
                                           bool &in_range) {
    in_range = local < a.n;
    const uint64_t s = a.s_begin + (uint64_t)(in_range ? local : 0);
    uint32_t w[4];
    philox4x32_10((uint32_t)s, (uint32_t)(s >> 32), a.epoch, 0u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), w);
    const uint32_t ii = lemire_bounded2(w[0], w[1], a.n_pos, a.th_pos);
    const uint32_t jj = lemire_bounded2(w[2], w[3], a.n_neg, a.th_neg);
    u = a.user_ids[ii];
    i = a.indices[ii];
    j = a.neg_population == CORNAC_HIP_NEG_POPULARITY ? a.neg_items[jj] : (int32_t)jj;
    const bool skip = HOG_ABLATE(a, 1) ? false : csr_row_contains(a.indices, a.indptr[u], a.indptr[u + 1], j);
    return in_range && !skip;
}

// ownership variant: the positive is drawn from the calling wave's own slice [base, base+len);
// `local` indexes the wave's samples of this epoch.  u is returned encoded (negative = shared user).
__device__ __forceinline__ bool hog_sample_owned(const HogArgs &a, uint32_t wave_id, int64_t base, uint32_t len,
                                                 uint32_t th_len, int64_t local, bool in_range, int32_t &u_enc,
                                                 int32_t &i, int32_t &j, bool share_neg = false) {
    const uint64_t s = (uint64_t)(in_range ? local : 0);
    uint32_t w[4];
    philox4x32_10((uint32_t)s, wave_id, a.epoch, 1u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), w);
    const uint32_t r = lemire_bounded2(w[0], w[1], len, th_len);
    const uint32_t jj = lemire_bounded2(w[2], w[3], a.n_neg, a.th_neg);
    u_enc = a.own_u[base + r];
    i = a.own_i[base + r];
    const int32_t u = u_enc < 0 ? ~u_enc : u_enc;
    j = a.neg_population == CORNAC_HIP_NEG_POPULARITY ? a.neg_items[jj] : (int32_t)jj;
    if (share_neg) j = __shfl(j, lane_id() & ~3, kWave);  // every lane draws; groups of 4 keep their leader's
    const bool skip = HOG_ABLATE(a, 1) ? false : csr_row_contains(a.indices, a.indptr[u], a.indptr[u + 1], j);
    return in_range && !skip;
}

// k % 4 == 0 and k <= 4*G: each lane owns one 16-byte slice of the three rows (registers only).
template <int G, bool ATOMIC>
__global__ __launch_bounds__(kBlock) void bpr_hogwild_vec4_kernel(const HogArgs a) {
    __shared__ int32_t stage[kWavesPerBlock][3][kWave];
    constexpr int TPW = kWave / G;  // triplets processed concurrently by one wave
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int grp = lane / G, lg = lane & (G - 1);
    const int64_t total_waves = (int64_t)gridDim.x * kWavesPerBlock;
    const int64_t n_tiles = (a.n + kWave - 1) / kWave;
    const int f0 = 4 * lg;
    const bool inb = f0 < a.k;
    unsigned int n_correct = 0, n_skipped = 0;
    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave; tile < n_tiles; tile += total_waves) {
        int32_t u, i, j;
        bool in_range;
        const bool valid = hog_sample(a, tile * kWave + lane, u, i, j, in_range);
        const unsigned long long mask = __ballot(valid);
        n_skipped += (in_range && !valid) ? 1u : 0u;
        if (valid) {
            const int pos = __popcll(mask & ((1ull << lane) - 1ull));
            stage[wave][0][pos] = u;
            stage[wave][1][pos] = i;
            stage[wave][2][pos] = j;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nvalid = __popcll(mask);
        for (int b = 0; b < nvalid; b += TPW) {
            const int slot = b + grp;
            const bool act = slot < nvalid;
            const int sl = act ? slot : b;
            const int32_t tu = stage[wave][0][sl], ti = stage[wave][1][sl], tj = stage[wave][2][sl];
            float *pu = a.U + (size_t)tu * a.k + f0;
            float *pi = a.V + (size_t)ti * a.k + f0;
            float *pj = a.V + (size_t)tj * a.k + f0;
            v4f u4 = {0.f, 0.f, 0.f, 0.f}, vi4 = u4, vj4 = u4;
            if (inb) {
                u4 = load_row4_fresh(pu);
                vi4 = load_row4_fresh(pi);
                vj4 = load_row4_fresh(pj);
            }
            const float bi = a.B[(size_t)ti * a.bstride], bj = a.B[(size_t)tj * a.bstride];
            const v4f d = vi4 - vj4;
            const float part = u4.x * d.x + u4.y * d.y + u4.z * d.z + u4.w * d.w;
            const float score = (bi - bj) + group_sum<G>(part);
            const float z = sigmoid_neg_fast(score);
            if (act && inb) {
                const v4f du = a.lr * (z * d - a.reg * u4);
                const v4f dvi = a.lr * (z * u4 - a.reg * vi4);
                const v4f dvj = a.lr * (-z * u4 - a.reg * vj4);
                if (ATOMIC) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        atomic_add_f32(pu + c, du[c]);
                        atomic_add_f32(pi + c, dvi[c]);
                        atomic_add_f32(pj + c, dvj[c]);
                    }
                } else {
                    *reinterpret_cast<v4f *>(pu) = u4 + du;
                    *reinterpret_cast<v4f *>(pi) = vi4 + dvi;
                    *reinterpret_cast<v4f *>(pj) = vj4 + dvj;
                }
            }
            if (act && lg == 0) {
                if (a.use_bias) {
                    const float dbi = a.lr * (z - a.reg * bi), dbj = a.lr * (-z - a.reg * bj);
                    if (ATOMIC) {
                        atomic_add_f32(a.B + (size_t)ti * a.bstride, dbi);
                        atomic_add_f32(a.B + (size_t)tj * a.bstride, dbj);
                    } else {
                        a.B[(size_t)ti * a.bstride] = bi + dbi;
                        a.B[(size_t)tj * a.bstride] = bj + dbj;
                    }
                }
                n_correct += z < .5f ? 1u : 0u;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // wave-level reduction of the counters, one atomic per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        n_correct += __shfl_xor(n_correct, o, kWave);
        n_skipped += __shfl_xor(n_skipped, o, kWave);
    }
    if (lane == 0) {
        if (n_correct) atomicAdd(&a.counters[0], (unsigned long long)n_correct);
        if (n_skipped) atomicAdd(&a.counters[1], (unsigned long long)n_skipped);
    }
}

// Row-wise layout (the production shape): the G lanes of a group own consecutive floats of a row,
// so every gather and every fp32-atomic scatter instruction covers whole 128-byte lines
// (tools/atomic_probe.hip: device-scope atomics cost one slot per touched line and instruction,
// ~10 G line-requests/s chip-wide, regardless of how many dwords of the line are active — the
// float4-per-lane layout above needs 4x the requests).  k <= G * R; UNR batches of TPW triplets
// are kept in flight per wave to cover the HBM/fabric latency.
//
// OWNED (G == 64 only): every wave owns a fixed set of users for the whole launch and draws its
// positives only from their interactions, so a user row is read and written by exactly one wave:
// plain load/store instead of 2 atomic line-requests per triplet, and no lost or stale U update.
// Heavy users (more interactions than half a wave's share) are split over all waves and keep atomics.
//
// MODE (OWNED only; the row-sharded item table of sharded.inc): 0 = the fused kernel; 1 = EMIT: sample only and write
// the wave's triplets to trip_* (-1 = skipped draw), no table access; 2 = STAGED: take the triplets from trip_* — item
// entries are SLOTS of the staging table a.V / a.B — and update.  The same grid and ownership tables in both launches
// keep every user row with one wave, so U stays on plain loads/stores exactly as in the fused kernel.
template <int G, int R, int UNR, bool ATOMIC, bool OWNED, int MODE = 0, bool SHARE = false>
__global__ __launch_bounds__(kBlock) void bpr_hogwild_rowwise_kernel(const HogArgs a) {
    static_assert(!SHARE || (OWNED && MODE == 0), "the shared-negative experiment exists for the fused owned kernel only");
    static_assert(!OWNED || G == kWave, "ownership needs one triplet per wave step");
    static_assert(MODE == 0 || OWNED, "the emit / staged forms exist for the owned kernel only");
    float *const Vt = a.V;
    float *const Bt = a.B;
    __shared__ int32_t stage[kWavesPerBlock][3][kWave];
    constexpr int TPW = kWave / G;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int grp = lane / G, lg = lane & (G - 1);
    const int64_t total_waves = (int64_t)gridDim.x * kWavesPerBlock;
    const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    int64_t n_tiles = (a.n + kWave - 1) / kWave, tile0 = wave_id, tile_step = total_waves;
    int64_t own_base = 0, own_lo = 0, own_hi = 0;
    uint32_t own_len = 1, own_th = 0;
    if (OWNED && MODE != 2) {
        own_base = a.wave_ptr[wave_id];
        own_len = (uint32_t)(a.wave_ptr[wave_id + 1] - own_base);
        // this launch covers the epoch fraction [s_begin, s_begin + n) / nnz of every wave's samples
        // chunked epochs (multi-GPU exchange points): a launch covers the tile range [lo_tile, hi_tile) of EVERY
        // wave's samples (tiles of 64 samples; own_tmax = the longest wave's tile count), so all waves run the same
        // number of whole tiles per launch and consecutive launches cover every sample exactly once
        const int64_t lo_tile = (int64_t)(((unsigned __int128)a.own_tmax * a.s_begin) / (uint64_t)a.nnz);
        const int64_t hi_tile = a.s_begin + (uint64_t)a.n >= (uint64_t)a.nnz
                                    ? a.own_tmax
                                    : (int64_t)(((unsigned __int128)a.own_tmax * (a.s_begin + (uint64_t)a.n)) / (uint64_t)a.nnz);
        own_lo = min((int64_t)own_len, lo_tile * kWave);
        own_hi = min((int64_t)own_len, hi_tile * kWave);
        n_tiles = own_len ? (own_hi - own_lo + kWave - 1) / kWave : 0;
        tile0 = 0;
        tile_step = 1;
        own_th = own_len ? lemire_thresh(own_len) : 0;
    }
    unsigned int n_correct = 0, n_skipped = 0;
    if (MODE == 1) {
        for (int64_t tile = 0; tile < a.trip_tiles; ++tile) {
            int32_t su, si, sj;
            const int64_t local = own_lo + tile * kWave + lane;
            const bool in_range = local < own_hi;
            const bool valid = hog_sample_owned(a, (uint32_t)wave_id, own_base, own_len, own_th, local, in_range, su, si, sj);
            const int64_t pos = ((a.trip_tile0 + tile) * total_waves + wave_id) * kWave + lane;
            a.trip_u[pos] = valid ? (su < 0 ? (~su | kTripShared) : su) : -1;
            a.trip_i[pos] = valid ? si : -1;
            a.trip_j[pos] = valid ? sj : -1;
            n_skipped += (in_range && !valid) ? 1u : 0u;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n_skipped += __shfl_xor(n_skipped, o, kWave);
        if (lane == 0 && n_skipped) atomicAdd(&a.counters[1], (unsigned long long)n_skipped);
        return;
    }
    if (MODE == 2) {
        n_tiles = a.trip_tiles;
        tile0 = 0;
        tile_step = 1;
    }
    bool inb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) inb[r] = lg + G * r < a.k;
    // (lanes beyond k hold zeros: they enter the dot product; so do all lanes under the "no row loads" ablation)
    for (int64_t tile = tile0; tile < n_tiles; tile += tile_step) {
        int32_t su, si, sj;
        bool in_range;
        bool valid;
        if (MODE == 2) {
            const int64_t pos = ((a.trip_tile0 + tile) * total_waves + wave_id) * kWave + lane;
            su = __builtin_nontemporal_load(a.trip_u + pos);
            si = __builtin_nontemporal_load(a.trip_i + pos);
            sj = __builtin_nontemporal_load(a.trip_j + pos);
            valid = su >= 0;
            in_range = valid;  // skipped draws were counted by the emit launch
            if (valid && (su & kTripShared)) su = ~(su & ~kTripShared);
        } else if (OWNED) {
            const int64_t local = own_lo + tile * kWave + lane;
            in_range = local < own_hi;
            valid = hog_sample_owned(a, (uint32_t)wave_id, own_base, own_len, own_th, local, in_range, su, si, sj, SHARE);
        } else {
            valid = hog_sample(a, tile * kWave + lane, su, si, sj, in_range);
        }
        const unsigned long long mask = __ballot(valid);
        n_skipped += (in_range && !valid) ? 1u : 0u;
        if (valid) {
            const int pos = __popcll(mask & ((1ull << lane) - 1ull));
            stage[wave][0][pos] = su;
            stage[wave][1][pos] = si;
            stage[wave][2][pos] = sj;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nvalid = __popcll(mask);
        for (int b = 0; b < nvalid; b += TPW * UNR) {
            float u[UNR][R], vi[UNR][R], vj[UNR][R], bi[UNR], bj[UNR];
            float *pu[UNR], *pi[UNR], *pj[UNR];
            int32_t ti[UNR], tj[UNR], tue[UNR];
            bool act[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                const int slot = b + q * TPW + grp;
                act[q] = slot < nvalid;
                const int sl = act[q] ? slot : b;
                int32_t tu = stage[wave][0][sl];
                if (OWNED) tu = __builtin_amdgcn_readfirstlane(tu);  // one triplet per wave step: wave-uniform
                tue[q] = tu;
                if (OWNED && tu < 0) tu = ~tu;
                ti[q] = stage[wave][1][sl];
                tj[q] = stage[wave][2][sl];
                pu[q] = a.U + (size_t)tu * a.k + lg;
                pi[q] = Vt + (size_t)ti[q] * a.k + lg;
                pj[q] = Vt + (size_t)tj[q] * a.k + lg;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool ld = inb[r] && !HOG_ABLATE(a, 4);
                    u[q][r] = ld ? __builtin_nontemporal_load(pu[q] + G * r) : 0.f;
                    vi[q][r] = ld ? __builtin_nontemporal_load(pi[q] + G * r) : 0.f;
                    vj[q][r] = ld ? __builtin_nontemporal_load(pj[q] + G * r) : 0.f;
                }
                bi[q] = HOG_ABLATE(a, 8) ? 0.f : __builtin_nontemporal_load(Bt + (size_t)ti[q] * a.bstride);
                bj[q] = HOG_ABLATE(a, 8) ? 0.f : __builtin_nontemporal_load(Bt + (size_t)tj[q] * a.bstride);
            }
            float du_all[UNR][R];
            float dvj_all[SHARE ? UNR : 1][R], dbj_all[SHARE ? UNR : 1];  // (SHARE: the negative rows' deltas, combined below)
            if (SHARE) {
#pragma unroll
                for (int q = 0; q < UNR; ++q) dbj_all[SHARE ? q : 0] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                float part = 0.f;
#pragma unroll
                for (int r = 0; r < R; ++r) part += u[q][r] * (vi[q][r] - vj[q][r]);
                const float score = (bi[q] - bj[q]) + group_sum<G>(part);
                const float z = sigmoid_neg_fast(score);
#pragma unroll
                for (int r = 0; r < R; ++r) du_all[q][r] = a.lr * (z * (vi[q][r] - vj[q][r]) - a.reg * u[q][r]);
                if (act[q]) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if (inb[r] && !HOG_ABLATE(a, 2)) {
                            const float du = du_all[q][r];
                            const float dvi = a.lr * (z * u[q][r] - a.reg * vi[q][r]);
                            const float dvj = a.lr * (-z * u[q][r] - a.reg * vj[q][r]);
                            if (OWNED) {
                                if (tue[q] < 0) atomic_add_f32(pu[q] + G * r, du);  // shared heavy user
                                atomic_add_f32(pi[q] + G * r, dvi);
                                if (SHARE) dvj_all[SHARE ? q : 0][r] = dvj;
                                if (!SHARE) atomic_add_f32(pj[q] + G * r, dvj);
                            } else if (ATOMIC) {
                                atomic_add_f32(pu[q] + G * r, du);
                                atomic_add_f32(pi[q] + G * r, dvi);
                                atomic_add_f32(pj[q] + G * r, dvj);
                            } else {
                                pu[q][G * r] = u[q][r] + du;
                                pi[q][G * r] = vi[q][r] + dvi;
                                pj[q][G * r] = vj[q][r] + dvj;
                            }
                        }
                    }
                    if (lg == 0) {
                        if (a.use_bias && !HOG_ABLATE(a, 8)) {
                            const float dbi = a.lr * (z - a.reg * bi[q]), dbj = a.lr * (-z - a.reg * bj[q]);
                            if (SHARE) dbj_all[SHARE ? q : 0] = dbj;
                            if (ATOMIC || OWNED) {
                                atomic_add_f32(a.B + (size_t)ti[q] * a.bstride, dbi);
                                if (!SHARE) atomic_add_f32(a.B + (size_t)tj[q] * a.bstride, dbj);
                            } else {
                                a.B[(size_t)ti[q] * a.bstride] = bi[q] + dbi;
                                a.B[(size_t)tj[q] * a.bstride] = bj[q] + dbj;
                            }
                        }
                        n_correct += z < .5f ? 1u : 0u;
                    }
                }
            }
            if (SHARE) {
                // one atomic row update per DISTINCT negative item of the batch: the first triplet naming it carries
                // the sum of the deltas of all of them (each computed against the same loaded row)
#pragma unroll
                for (int q = 0; q < UNR; ++q) {
                    if (!act[q] || HOG_ABLATE(a, 2)) continue;
                    bool lead = true;
#pragma unroll
                    for (int q2 = 0; q2 < q; ++q2) lead = lead && !(act[q2] && tj[q2] == tj[q]);
                    if (!lead) continue;
                    float tot[R], totb = dbj_all[SHARE ? q : 0];
#pragma unroll
                    for (int r = 0; r < R; ++r) tot[r] = dvj_all[SHARE ? q : 0][r];
#pragma unroll
                    for (int q2 = q + 1; q2 < UNR; ++q2) {
                        if (act[q2] && tj[q2] == tj[q]) {
#pragma unroll
                            for (int r = 0; r < R; ++r) tot[r] += dvj_all[SHARE ? q2 : 0][r];
                            totb += dbj_all[SHARE ? q2 : 0];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (inb[r]) atomic_add_f32(pj[q] + G * r, tot[r]);
                    if (lg == 0 && a.use_bias && !HOG_ABLATE(a, 8)) atomic_add_f32(a.B + (size_t)tj[q] * a.bstride, totb);
                }
            }
            if (OWNED) {
                // exclusive users: plain store of u_old + (sum of the deltas of every triplet of this
                // batch that has the same user) — all four loads saw the same u_old
#pragma unroll
                for (int q = 0; q < UNR; ++q) {
                    if (act[q] && tue[q] >= 0 && !HOG_ABLATE(a, 2)) {
                        float tot[R];
#pragma unroll
                        for (int r = 0; r < R; ++r) tot[r] = du_all[q][r];
#pragma unroll
                        for (int q2 = 0; q2 < UNR; ++q2) {
                            if (q2 != q && act[q2] && tue[q2] == tue[q]) {
#pragma unroll
                                for (int r = 0; r < R; ++r) tot[r] += du_all[q2][r];
                            }
                        }
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (inb[r]) pu[q][G * r] = u[q][r] + tot[r];
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        n_correct += __shfl_xor(n_correct, o, kWave);
        n_skipped += __shfl_xor(n_skipped, o, kWave);
    }
    if (lane == 0) {
        if (n_correct) atomicAdd(&a.counters[0], (unsigned long long)n_correct);
        if (n_skipped) atomicAdd(&a.counters[1], (unsigned long long)n_skipped);
    }
}

// any k: G lanes per triplet stride over the factors (two passes over the rows).
template <int G, bool ATOMIC>
__global__ __launch_bounds__(kBlock) void bpr_hogwild_generic_kernel(const HogArgs a) {
    __shared__ int32_t stage[kWavesPerBlock][3][kWave];
    constexpr int TPW = kWave / G;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int grp = lane / G, lg = lane & (G - 1);
    const int64_t total_waves = (int64_t)gridDim.x * kWavesPerBlock;
    const int64_t n_tiles = (a.n + kWave - 1) / kWave;
    unsigned int n_correct = 0, n_skipped = 0;
    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave; tile < n_tiles; tile += total_waves) {
        int32_t u, i, j;
        bool in_range;
        const bool valid = hog_sample(a, tile * kWave + lane, u, i, j, in_range);
        const unsigned long long mask = __ballot(valid);
        n_skipped += (in_range && !valid) ? 1u : 0u;
        if (valid) {
            const int pos = __popcll(mask & ((1ull << lane) - 1ull));
            stage[wave][0][pos] = u;
            stage[wave][1][pos] = i;
            stage[wave][2][pos] = j;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nvalid = __popcll(mask);
        for (int b = 0; b < nvalid; b += TPW) {
            const int slot = b + grp;
            const bool act = slot < nvalid;
            const int sl = act ? slot : b;
            const int32_t tu = stage[wave][0][sl], ti = stage[wave][1][sl], tj = stage[wave][2][sl];
            float *pu = a.U + (size_t)tu * a.k, *pi = a.V + (size_t)ti * a.k, *pj = a.V + (size_t)tj * a.k;
            float part = 0.f;
            for (int f = lg; f < a.k; f += G)
                part += load_f32_fresh(pu + f) * (load_f32_fresh(pi + f) - load_f32_fresh(pj + f));
            const float bi = a.B[(size_t)ti * a.bstride], bj = a.B[(size_t)tj * a.bstride];
            const float score = (bi - bj) + group_sum<G>(part);
            const float z = sigmoid_neg_fast(score);
            if (act) {
                for (int f = lg; f < a.k; f += G) {
                    const float uf = load_f32_fresh(pu + f), vi = load_f32_fresh(pi + f),
                                vj = load_f32_fresh(pj + f);
                    const float du = a.lr * (z * (vi - vj) - a.reg * uf);
                    const float dvi = a.lr * (z * uf - a.reg * vi);
                    const float dvj = a.lr * (-z * uf - a.reg * vj);
                    if (ATOMIC) {
                        atomic_add_f32(pu + f, du);
                        atomic_add_f32(pi + f, dvi);
                        atomic_add_f32(pj + f, dvj);
                    } else {
                        pu[f] = uf + du;
                        pi[f] = vi + dvi;
                        pj[f] = vj + dvj;
                    }
                }
                if (lg == 0) {
                    if (a.use_bias) {
                        const float dbi = a.lr * (z - a.reg * bi), dbj = a.lr * (-z - a.reg * bj);
                        if (ATOMIC) {
                            atomic_add_f32(a.B + (size_t)ti * a.bstride, dbi);
                            atomic_add_f32(a.B + (size_t)tj * a.bstride, dbj);
                        } else {
                            a.B[(size_t)ti * a.bstride] = bi + dbi;
                            a.B[(size_t)tj * a.bstride] = bj + dbj;
                        }
                    }
                    n_correct += z < .5f ? 1u : 0u;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        n_correct += __shfl_xor(n_correct, o, kWave);
        n_skipped += __shfl_xor(n_skipped, o, kWave);
    }
    if (lane == 0) {
        if (n_correct) atomicAdd(&a.counters[0], (unsigned long long)n_correct);
        if (n_skipped) atomicAdd(&a.counters[1], (unsigned long long)n_skipped);
    }
}

}  // namespace chip
#include "bpr_binned.inc"
#include "bpr_strata.inc"
#include "bpr_ldsbin.inc"

namespace chip {

static int pow2_group(int k) {  // lanes per triplet for scalar-per-lane kernels
    int g = 4;
    while (g < k && g < 64) g <<= 1;
    return g;
}

}  // namespace chip

using namespace chip;

// ================================================================================================
// handle
// ================================================================================================
struct cornac_hip_bpr {
    int device = 0;
    int64_t n_users = 0, n_items = 0, total_users = 0, total_items = 0, nnz = 0;
    int k = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    DevBuf<int32_t> indptr, indices, user_ids;
    DevBuf<int32_t> neg_pop;   // caller's negative population (cornac_hip_bpr_set_negative_population), empty: `indices`
    int64_t neg_pop_n = 0;
    DevBuf<float> U, V, B;
    bool vebpr_owned = false;  // the last VEBPR hogwild epoch ran with user-row ownership
    bool f64 = false;  // float64 tables (set_factors_f64): deterministic mode only, like the reference's fused-type loop
    DevBuf<double> U64, V64, B64;
    DevBuf<float> Bpad;  // hogwild-mode view of B, one bias per 128-byte line
    // XCD-strata form inside fit_epochs: ONE record per item — the row (padded to whole 128-byte lines) with its bias line
    // behind it — so that a triplet names 3 random locations instead of 5.  While strata_packed is set the records are
    // authoritative and the dense V / B are stale; every other entry point unpacks first (bpr_check).
    DevBuf<float> VB;
    bool strata_packed = false, strata_allow_pack = false;
    bool chunk_records = false;  // cornac_hip_bpr_chunk_records: the chunk API may keep the records too (multi-GPU driver)
    int vb_pitch = 0, vb_kp = 0;
    DevBuf<unsigned long long> counters;  // [0] correct, [1] skipped, [2] strata workgroups placed off their logical XCD
    // deterministic sampler state
    DevBuf<uint32_t> mt_state;  // 2 x 624
    DevBuf<int32_t> mt_idx;     // 2
    DevBuf<MtStreamParams> mt_params;
    bool mt_seeded = false, shared_stream = false;
    DevBuf<uint32_t> draws;  // 2 x chunk
    DevBuf<int32_t> trip;    // 6 x chunk : su si sj | ou oi oj
    PinnedBuf<int32_t> h_trip;
    std::vector<int32_t> lvl_u, lvl_i, level;
    LevelSchedule sched;
    DevBuf<int32_t> level_dev;
    DevBuf<int64_t> cursor_dev;
    // hogwild sampler state
    bool hog_seeded = false;
    uint64_t hog_seed = 0;
    uint32_t hog_epoch = 0;
    int64_t hog_offset = 0;  // samples already consumed in the current epoch
    double timing[4] = {0, 0, 0, 0};
    EventTimer ktimer;  // hogwild SGD kernel launches
    void (*hog_kernel)(const chip::HogArgs) = nullptr;
    int hog_blocks_per_cu = 8;
    int staged_blocks_per_cu = 0;  // sharded.inc: occupancy of the staged (MODE 2) kernel, 0 = not queried yet
    // user-row ownership tables of the hogwild kernel (built lazily for the persistent grid width)
    std::vector<int32_t> h_indptr, h_indices;
    int64_t own_waves = 0;
    DevBuf<int32_t> own_u, own_i;
    DevBuf<int64_t> wave_ptr;
    std::vector<int32_t> h_own_u, h_own_i;
    std::vector<int64_t> h_wave_ptr;
    int64_t own_tmax = 0;
    // XCD strata (bpr_strata.inc): popularity ranks, per-epoch partition buckets of the ownership slices
    bool strata_ranked = false;
    DevBuf<int32_t> item_rank, rank_item, rec_u, rec_i;
    DevBuf<int64_t> sptr;
    std::vector<int32_t> h_rank_item;
    uint32_t strata_n_hot = 0, strata_key_built = 0;
    bool strata_built = false;
    int strata_hot_permille = 120, strata_hot_min_mult_x100 = 200, strata_rehash_period = 1;
    int64_t strata_misplaced = 0, strata_builds = 0;
    DevBuf<int32_t> strata_own_code;  // item_rank[own_i[t]] (static; rebuilt with the ownership tables)
    int64_t strata_code_waves = -1;
    DevBuf<uint32_t> strata_claim;  // [8 phases][16]: per-XCD slot counters of a phase launch (bpr_strata.inc), zeroed per epoch
    void (*strata_kernel)(const chip::HogArgs) = nullptr;
    int strata_blocks_per_cu = 0;
    // LDS-resident item bins (bpr_ldsbin.inc): CSC, hot interaction list, membership bitmap
    bool lb_built = false;
    DevBuf<int32_t> lb_cptr, lb_cusers, lb_hot_u, lb_hot_i;
    DevBuf<uint32_t> lb_bitmap;
    int lb_bins = 0, lb_cap = 0, lb_n_hot = 0, lb_n_hot_inter = 0, lb_bm_words = 0;
    int lb_hot_x1000 = 75, lb_min_candidates = 48, lb_max_rounds = 4;
    int lb_pass_enable = 1, lb_pass_waves = 8, lb_pass_kb = 64, lb_pass_min_draws_x100 = 200;  // passing bins (ldsbin_plan)
    int lb_block = kLbBlock;
    bool lb_passing = false;
    int lb_strata_groups = 16, lb_hot_cost_x16 = 32;  // the deal: stratum width in groups, price of a hot draw
    int lb_n_strata = 1;
    DevBuf<uint32_t> lb_mass, lb_cold, lb_hot_off;
    DevBuf<int32_t> lb_pre_rec;               // pre-sampled tiles (ldsbin_presample_kernel): [pool][3][64]
    DevBuf<uint32_t> lb_pre_cnt, lb_pre_base; // [pool] valid triplets per tile; [bins + 1] first record of a bin
    DevBuf<unsigned long long> lb_wg_clock;  // profile build, CORNAC_HIP_LDSBIN_CLOCKS=<file>
    bool lb_deal_valid = false;   // lb_hot_off / lb_cold hold the deal of (lb_deal_seed, lb_deal_epoch)
    uint64_t lb_deal_seed = 0;
    uint32_t lb_deal_epoch = 0;
    size_t lb_lds_bytes = 0;
    bool lb_attr_set = false;
    int64_t lb_lock_timeouts = 0;
    // conveyor layout of the LDS-bin form (multi-GPU regime 2; cornac_hip_bpr_conveyor_setup): cv_blocks != 0 — the bins are
    // planned as cv_blocks equal ranges (the conveyor's item blocks), no hot items, the deal keyed by cv_deal_seed (shared
    // by the ranks of a fit, unlike hog_seed) over cv_rank_item (the popularity order all ranks agree on)
    int cv_blocks = 0;
    uint64_t cv_deal_seed = 0;
    bool cv_own_order = false;
    DevBuf<int32_t> cv_rank_item;
    // binned item updates (bpr_binned.inc): item -> (bucket, local row), bucket -> items, message segments
    int bin_buckets = 0, bin_neg_population = -1, bin_max_rows = 0, bin_wg_per_cu = 0, bin_n_hot = 0;
    int bin_hot_threshold = 0;
    int64_t bin_chunk = 0;
    bool bin_attr_set = false;
    DevBuf<int32_t> bin_item_slot, bin_bucket_ptr, bin_bucket_items, bin_seg_count, bin_hot_items;
    DevBuf<uint4> bin_seg;
    DevBuf<float> bin_snap, bin_hot_bias;
    // VEBPR: view CSR + third sampler stream
    bool has_views = false, view_seeded = false;
    DevBuf<int32_t> v_indptr, v_indices, view_rank;
    DevBuf<uint8_t> has_view;
};

static constexpr int64_t kDetChunk = int64_t(1) << 24;

static void strata_unpack(cornac_hip_bpr_t h);
// every entry point that may read or write the dense tables goes through here: packed strata records are written back
static void bpr_check(cornac_hip_bpr_t h) {
    REQUIRE(h != nullptr, "BPR handle is NULL");
    HIP_CHECK(hipSetDevice(h->device));
    if (h->strata_packed) strata_unpack(h);
}
// ... except the epoch loop of fit_epochs, which keeps them across calls
static void bpr_check_keep_packed(cornac_hip_bpr_t h) {
    REQUIRE(h != nullptr, "BPR handle is NULL");
    HIP_CHECK(hipSetDevice(h->device));
}

extern "C" {

int cornac_hip_bpr_create(cornac_hip_bpr_t *out, int device, int64_t n_users, int64_t n_items, int64_t total_users,
                          int64_t total_items, int k, const int32_t *indptr, const int32_t *indices, int64_t nnz) {
    return guarded([&] {
        REQUIRE(out != nullptr, "out handle pointer is NULL");
        *out = nullptr;
        REQUIRE(n_users > 0 && n_items > 0, "n_users and n_items must be positive");
        REQUIRE(total_users >= n_users && total_items >= n_items, "total_* must cover the train counts");
        REQUIRE(k > 0, "k must be positive");
        REQUIRE(indptr && indices, "CSR pointers are NULL");
        REQUIRE(nnz > 0, "empty interaction matrix (nnz == 0)");
        REQUIRE(nnz < (int64_t(1) << 31), "nnz >= 2^31 is beyond the reference (int32 CSR, recom_bpr.pyx:186)");
        REQUIRE(n_items < (int64_t(1) << 31) && total_users < (int64_t(1) << 31), "index range exceeds int32");
        REQUIRE(indptr[0] == 0 && (int64_t)indptr[n_users] == nnz, "indptr does not match nnz");
        use_device(device);
        std::unique_ptr<cornac_hip_bpr> h(new cornac_hip_bpr());
        h->device = device;
        h->n_users = n_users; h->n_items = n_items; h->total_users = total_users; h->total_items = total_items;
        h->nnz = nnz; h->k = k;
        HIP_CHECK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
        h->stream = h->own_stream;
        std::vector<int32_t> uid((size_t)nnz);
        for (int64_t u = 0; u < n_users; ++u) {
            REQUIRE(indptr[u + 1] >= indptr[u], "indptr is not monotone at row %lld", (long long)u);
            for (int32_t p = indptr[u]; p < indptr[u + 1]; ++p) {
                uid[(size_t)p] = (int32_t)u;
                REQUIRE(indices[p] >= 0 && indices[p] < n_items, "column index out of range at %d", p);
                REQUIRE(p == indptr[u] || indices[p] > indices[p - 1], "CSR row %lld is not strictly sorted",
                        (long long)u);
            }
        }
        h->h_indptr.assign(indptr, indptr + n_users + 1);
        h->h_indices.assign(indices, indices + nnz);
        h->indptr.alloc((size_t)n_users + 1);
        h->indices.alloc((size_t)nnz);
        h->user_ids.alloc((size_t)nnz);
        h->indptr.upload(indptr, (size_t)n_users + 1, h->stream);
        h->indices.upload(indices, (size_t)nnz, h->stream);
        h->user_ids.upload(uid.data(), (size_t)nnz, h->stream);
        h->U.alloc((size_t)total_users * k);
        h->V.alloc((size_t)total_items * k);
        h->B.alloc((size_t)total_items);
        HIP_CHECK(hipMemsetAsync(h->U.p, 0, h->U.n * sizeof(float), h->stream));
        HIP_CHECK(hipMemsetAsync(h->V.p, 0, h->V.n * sizeof(float), h->stream));
        HIP_CHECK(hipMemsetAsync(h->B.p, 0, h->B.n * sizeof(float), h->stream));
        h->counters.alloc(4);
        HIP_CHECK(hipMemsetAsync(h->counters.p, 0, 4 * sizeof(unsigned long long), h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        *out = h.release();
    });
}

int cornac_hip_bpr_destroy(cornac_hip_bpr_t h) {
    return guarded([&] {
        if (!h) return;
        (void)hipSetDevice(h->device);
        if (h->own_stream) {
            (void)hipStreamSynchronize(h->own_stream);
            (void)hipStreamDestroy(h->own_stream);
        }
        delete h;
    });
}

int cornac_hip_bpr_set_factors(cornac_hip_bpr_t h, const float *U, const float *V, const float *B) {
    return guarded([&] {
        bpr_check(h);
        h->f64 = false;
        if (U) h->U.upload(U, (size_t)h->total_users * h->k, h->stream);
        if (V) h->V.upload(V, (size_t)h->total_items * h->k, h->stream);
        if (B) h->B.upload(B, (size_t)h->total_items, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}

int cornac_hip_bpr_get_factors(cornac_hip_bpr_t h, float *U, float *V, float *B) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(!h->f64, "the handle holds float64 tables (use cornac_hip_bpr_get_factors_f64)");
        if (U) h->U.download(U, (size_t)h->total_users * h->k, h->stream);
        if (V) h->V.download(V, (size_t)h->total_items * h->k, h->stream);
        if (B) h->B.download(B, (size_t)h->total_items, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}

int cornac_hip_bpr_set_factors_f64(cornac_hip_bpr_t h, const double *U, const double *V, const double *B) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(U && V && B, "float64 tables are set together (the reference's fused type is one type for U, V and B)");
        h->U64.ensure((size_t)h->total_users * h->k);
        h->V64.ensure((size_t)h->total_items * h->k);
        h->B64.ensure((size_t)h->total_items);
        h->U64.upload(U, (size_t)h->total_users * h->k, h->stream);
        h->V64.upload(V, (size_t)h->total_items * h->k, h->stream);
        h->B64.upload(B, (size_t)h->total_items, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->f64 = true;
    });
}
int cornac_hip_bpr_get_factors_f64(cornac_hip_bpr_t h, double *U, double *V, double *B) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(h->f64, "the handle holds float32 tables (use cornac_hip_bpr_get_factors)");
        if (U) h->U64.download(U, (size_t)h->total_users * h->k, h->stream);
        if (V) h->V64.download(V, (size_t)h->total_items * h->k, h->stream);
        if (B) h->B64.download(B, (size_t)h->total_items, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}
int cornac_hip_bpr_bind_device(cornac_hip_bpr_t h, float *dU, float *dV, float *dB) {
    return guarded([&] {
        bpr_check(h);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        if (dU) h->U.bind(dU, (size_t)h->total_users * h->k);
        if (dV) h->V.bind(dV, (size_t)h->total_items * h->k);
        if (dB) h->B.bind(dB, (size_t)h->total_items);
    });
}

int cornac_hip_bpr_set_negative_population(cornac_hip_bpr_t h, const int32_t *items, int64_t n) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(n >= 0 && n < (int64_t(1) << 32), "population size out of range");
        REQUIRE(n == 0 || items != nullptr, "items is NULL");
        for (int64_t t = 0; t < n; ++t)
            REQUIRE(items[t] >= 0 && items[t] < h->n_items, "population entry %lld = %d is not a train item", (long long)t, items[t]);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->neg_pop_n = n;
        if (n) {
            h->neg_pop.ensure((size_t)n);
            h->neg_pop.upload(items, (size_t)n, h->stream);
            HIP_CHECK(hipStreamSynchronize(h->stream));
        }
    });
}

int cornac_hip_bpr_rebind_items(cornac_hip_bpr_t h, float *dV, float *dB) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(dV && dB, "NULL device pointer");
        REQUIRE(!h->V.owned && !h->B.owned, "the item tables are the handle's own: bind caller-owned ones first "
                "(cornac_hip_bpr_bind_device); rebind_items only swaps them");
        // no stream synchronisation: launches already enqueued keep the pointers they were given, later ones on the
        // handle's stream see the new ones (the ring conveyor of cornac_amd/dist.py rebinds once per step)
        h->V.bind(dV, (size_t)h->total_items * h->k);
        h->B.bind(dB, (size_t)h->total_items);
    });
}

int cornac_hip_bpr_device_ptrs(cornac_hip_bpr_t h, float **dU, float **dV, float **dB) {
    return guarded([&] {
        bpr_check(h);
        if (dU) *dU = h->U.p;
        if (dV) *dV = h->V.p;
        if (dB) *dB = h->B.p;
    });
}

int cornac_hip_bpr_set_stream(cornac_hip_bpr_t h, void *hip_stream) {
    return guarded([&] {
        bpr_check(h);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    });
}

int cornac_hip_bpr_switch_stream(cornac_hip_bpr_t h, void *hip_stream) {
    return guarded([&] {
        bpr_check(h);
        h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    });
}

static void mt_init_genrand(uint32_t seed, uint32_t *mt) {
    mt[0] = seed;
    for (int i = 1; i < MT_N; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}

int cornac_hip_bpr_seed_mt19937(cornac_hip_bpr_t h, uint32_t mt_seed_pos, uint32_t mt_seed_neg, int shared_stream) {
    return guarded([&] {
        bpr_check(h);
        std::vector<uint32_t> st(2 * MT_N);
        mt_init_genrand(mt_seed_pos, st.data());
        mt_init_genrand(mt_seed_neg, st.data() + MT_N);
        const int32_t idx[2] = {MT_N, MT_N};
        h->mt_state.ensure(3 * MT_N);  // [0] positive, [1] negative, [2] view (VEBPR)
        h->mt_idx.ensure(3);
        h->mt_params.ensure(2);
        h->mt_state.upload(st.data(), 2 * MT_N, h->stream);
        h->mt_idx.upload(idx, 2, h->stream);
        h->view_seeded = false;
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->mt_seeded = true;
        h->shared_stream = shared_stream != 0;
    });
}

int cornac_hip_bpr_seed_hogwild(cornac_hip_bpr_t h, uint64_t seed) {
    return guarded([&] {
        bpr_check_keep_packed(h);  // (touches no table: packed strata records stay)
        h->hog_seed = seed;
        h->hog_epoch = 0;
        h->hog_offset = 0;
        h->hog_seeded = true;
    });
}
}  // extern "C"

// ---- deterministic draws -------------------------------------------------------------------------
static MtStreamParams make_mt_params(cornac_hip_bpr_t h, int stream, uint64_t hi, int64_t need, uint32_t *out) {
    REQUIRE(hi <= 0xFFFFFFFFull, "sampling range >= 2^32 is not reachable in the reference (boost multi-draw branch)");
    MtStreamParams p;
    p.state = h->mt_state.p + (size_t)stream * MT_N;
    p.idx = h->mt_idx.p + stream;
    p.out = out;
    p.need = need;
    p.range = (uint32_t)hi;
    if (hi == 0xFFFFFFFFull) {
        p.bucket = 1;
    } else {
        const uint32_t r1 = (uint32_t)hi + 1u;
        p.bucket = 0xFFFFFFFFu / r1;
        if (0xFFFFFFFFu % r1 == (uint32_t)hi) ++p.bucket;
    }
    p.out_stride = 1;
    p.out_offset = 0;
    return p;
}

// launches the generator for up to two streams; hi == 0 streams draw nothing (boost returns min
// without touching the engine, uniform_int_distribution.hpp:64-65)
static void mt_draw(cornac_hip_bpr_t h, int n_streams, const int *streams, const uint64_t *his, const int64_t *needs,
                    uint32_t *const *outs) {
    MtStreamParams ps[2];
    int n = 0;
    for (int s = 0; s < n_streams; ++s) {
        if (his[s] == 0) {
            HIP_CHECK(hipMemsetAsync(outs[s], 0, (size_t)needs[s] * sizeof(uint32_t), h->stream));
        } else if (needs[s] > 0) {
            ps[n++] = make_mt_params(h, streams[s], his[s], needs[s], outs[s]);
        }
    }
    if (n == 0) return;
    HIP_CHECK(hipMemcpyAsync(h->mt_params.p, ps, sizeof(MtStreamParams) * n, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(mt19937_draw_kernel, dim3(n), dim3(MT_THREADS), 0, h->stream, h->mt_params.p);
    HIP_CHECK(hipGetLastError());
    // params live in pageable host memory: make sure the copy has been consumed before ps dies
    HIP_CHECK(hipStreamSynchronize(h->stream));
}

template <int G>
static void launch_det_level(cornac_hip_bpr_t h, const int32_t *ou, const int32_t *oi, const int32_t *oj, int64_t off,
                             int cnt, double lr, double reg, int use_bias) {
    const int groups_per_block = kBlock / G;
    const int grid = (cnt + groups_per_block - 1) / groups_per_block;
    if (h->f64)
        hipLaunchKernelGGL((bpr_det_level_kernel<G, double>), dim3(grid), dim3(kBlock), 0, h->stream, ou, oi, oj, off, cnt,
                           h->U64.p, h->V64.p, h->B64.p, h->k, lr, reg, use_bias, h->counters.p);
    else
        hipLaunchKernelGGL((bpr_det_level_kernel<G, float>), dim3(grid), dim3(kBlock), 0, h->stream, ou, oi, oj, off, cnt,
                           h->U.p, h->V.p, h->B.p, h->k, (float)lr, (float)reg, use_bias, h->counters.p);
}

static void bpr_epoch_deterministic(cornac_hip_bpr_t h, double lr, double reg, int use_bias, int neg_population) {
    REQUIRE(h->mt_seeded, "deterministic mode needs cornac_hip_bpr_seed_mt19937 first");
    // (the sequential engine restates recom_wbpr.pyx:131-139, whose negatives are the items of THIS matrix's interactions; a
    // caller-supplied population — the multi-GPU driver's global popularity — belongs to the hogwild forms)
    REQUIRE(!(neg_population == CORNAC_HIP_NEG_POPULARITY && h->neg_pop_n),
            "a negative population set with cornac_hip_bpr_set_negative_population is not honoured in deterministic mode: "
            "clear it (n = 0) or fit in hogwild mode");
    const int64_t nnz = h->nnz;
    const uint64_t pos_hi = (uint64_t)nnz - 1;
    const uint64_t neg_hi = neg_population == CORNAC_HIP_NEG_POPULARITY ? (uint64_t)nnz - 1 : (uint64_t)h->n_items - 1;
    const int64_t chunk = std::min(nnz, kDetChunk);
    h->draws.ensure((size_t)2 * chunk);
    h->trip.ensure((size_t)6 * chunk);
    h->h_trip.ensure((size_t)6 * chunk);
    const int G = pow2_group(h->k);
    for (int64_t c0 = 0; c0 < nnz; c0 += chunk) {
        const int64_t n = std::min(chunk, nnz - c0);
        Timer t_s;
        uint32_t *d_pos = h->draws.p, *d_neg = h->draws.p + chunk;
        int pos_stride = 1, neg_stride = 1;
        if (h->shared_stream) {
            // WBPR: one engine, draws alternate pos, neg, pos, ... with the same range (recom_wbpr.pyx:131-139)
            REQUIRE(pos_hi == neg_hi, "shared-stream sampling needs equal ranges (WBPR uses X.indices for both)");
            const int st = 0;
            const int64_t need = 2 * n;
            uint32_t *o = h->draws.p;
            mt_draw(h, 1, &st, &pos_hi, &need, &o);
            d_pos = h->draws.p;
            d_neg = h->draws.p + 1;
            pos_stride = neg_stride = 2;
        } else {
            const int st[2] = {0, 1};
            const uint64_t his[2] = {pos_hi, neg_hi};
            const int64_t needs[2] = {n, n};
            uint32_t *outs[2] = {d_pos, d_neg};
            mt_draw(h, 2, st, his, needs, outs);
        }
        int32_t *su = h->trip.p, *si = su + chunk, *sj = si + chunk;
        int32_t *ou = sj + chunk, *oi = ou + chunk, *oj = oi + chunk;
        hipLaunchKernelGGL(bpr_det_sample_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           h->stream, d_pos, pos_stride, d_neg, neg_stride, n, h->user_ids.p, h->indices.p,
                           h->indptr.p, neg_population, su, si, sj, h->counters.p);
        HIP_CHECK(hipGetLastError());
        int32_t *hsu = h->h_trip.p, *hsi = hsu + chunk, *hsj = hsi + chunk;
        int32_t *hou = hsj + chunk;
        HIP_CHECK(hipMemcpyAsync(hsu, su, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipMemcpyAsync(hsi, si, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipMemcpyAsync(hsj, sj, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->timing[0] += t_s.ms();
        Timer t_l;
        // host: longest-path level of every sample (inherently sequential) into the pinned buffer; device: bucketing
        int32_t *hlevel = hou;  // the pinned "sorted" region is free now: only the levels travel back
        build_levels(hsu, hsi, hsj, n, h->total_users, h->total_items, h->sched, h->lvl_u, h->lvl_i, hlevel);
        h->timing[1] += t_l.ms();
        Timer t_k;
        const int64_t na = h->sched.n_active;
        if (na > 0) {
            const std::vector<int64_t> &lp0 = h->sched.level_ptr;
            h->level_dev.ensure((size_t)n);
            h->cursor_dev.ensure(lp0.size());
            HIP_CHECK(hipMemcpyAsync(h->level_dev.p, hlevel, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
            HIP_CHECK(hipMemcpyAsync(h->cursor_dev.p, lp0.data(), lp0.size() * sizeof(int64_t), hipMemcpyHostToDevice,
                                     h->stream));
            const unsigned bg = (unsigned)std::min<int64_t>((n + kBlock - 1) / kBlock, 8192);
            hipLaunchKernelGGL(bpr_det_bucket_kernel, dim3(bg), dim3(kBlock), 0, h->stream, h->level_dev.p, su, si, sj, n,
                               reinterpret_cast<unsigned long long *>(h->cursor_dev.p), ou, oi, oj);
        }
        const std::vector<int64_t> &lp = h->sched.level_ptr;
        for (size_t l = 1; l + 1 < lp.size(); ++l) {
            const int64_t off = lp[l];
            const int cnt = (int)(lp[l + 1] - lp[l]);
            if (cnt <= 0) continue;
            switch (G) {
                case 4: launch_det_level<4>(h, ou, oi, oj, off, cnt, lr, reg, use_bias); break;
                case 8: launch_det_level<8>(h, ou, oi, oj, off, cnt, lr, reg, use_bias); break;
                case 16: launch_det_level<16>(h, ou, oi, oj, off, cnt, lr, reg, use_bias); break;
                case 32: launch_det_level<32>(h, ou, oi, oj, off, cnt, lr, reg, use_bias); break;
                default: launch_det_level<64>(h, ou, oi, oj, off, cnt, lr, reg, use_bias); break;
            }
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->timing[2] += t_k.ms();
    }
}

// ---- hogwild launch -------------------------------------------------------------------------------
typedef void (*HogKernel)(const HogArgs);

template <bool ATOMIC>
static HogKernel pick_hogwild_kernel(int k, int flags) {
    const bool vec4_layout = (flags & 2) != 0;  // experiment switch: the float4-per-lane layout
    if (!vec4_layout && k <= 256) {
        const bool owned = ATOMIC && (flags & 4) == 0 && k > 32;  // bit2 disables user-row ownership
        if (k <= 4) return bpr_hogwild_rowwise_kernel<4, 1, 2, ATOMIC, false>;
        if (k <= 8) return bpr_hogwild_rowwise_kernel<8, 1, 2, ATOMIC, false>;
        if (k <= 16) return bpr_hogwild_rowwise_kernel<16, 1, 2, ATOMIC, false>;
        if (k <= 32) return bpr_hogwild_rowwise_kernel<32, 1, 4, ATOMIC, false>;
        if (owned) {
            if (k <= 64 && (flags & 16)) return bpr_hogwild_rowwise_kernel<64, 1, 4, true, true, 0, true>;  // experiment
            if (k <= 64) return bpr_hogwild_rowwise_kernel<64, 1, 4, true, true>;
            if (k <= 128) return bpr_hogwild_rowwise_kernel<64, 2, 2, true, true>;
            if (k <= 192) return bpr_hogwild_rowwise_kernel<64, 3, 2, true, true>;
            return bpr_hogwild_rowwise_kernel<64, 4, 1, true, true>;
        }
        if (k <= 64) return bpr_hogwild_rowwise_kernel<64, 1, 4, ATOMIC, false>;
        if (k <= 128) return bpr_hogwild_rowwise_kernel<64, 2, 2, ATOMIC, false>;
        if (k <= 192) return bpr_hogwild_rowwise_kernel<64, 3, 2, ATOMIC, false>;
        return bpr_hogwild_rowwise_kernel<64, 4, 1, ATOMIC, false>;
    }
    if (k % 4 == 0 && k <= 256) {
        const int q = k / 4;
        if (q <= 4) return bpr_hogwild_vec4_kernel<4, ATOMIC>;
        if (q <= 8) return bpr_hogwild_vec4_kernel<8, ATOMIC>;
        if (q <= 16) return bpr_hogwild_vec4_kernel<16, ATOMIC>;
        if (q <= 32) return bpr_hogwild_vec4_kernel<32, ATOMIC>;
        return bpr_hogwild_vec4_kernel<64, ATOMIC>;
    }
    switch (pow2_group(k)) {
        case 4: return bpr_hogwild_generic_kernel<4, ATOMIC>;
        case 8: return bpr_hogwild_generic_kernel<8, ATOMIC>;
        case 16: return bpr_hogwild_generic_kernel<16, ATOMIC>;
        case 32: return bpr_hogwild_generic_kernel<32, ATOMIC>;
        default: return bpr_hogwild_generic_kernel<64, ATOMIC>;
    }
}

// Assign every user to one wave of the persistent grid, balanced by interaction count (LPT greedy).
// Users heavier than half a wave's share are "shared": their interactions are dealt out evenly
// to all waves and their rows keep atomic updates (encoded as ~u in own_u).
static void build_ownership(cornac_hip_bpr_t h, int64_t W) {
    if (h->own_waves == W) return;
    const int64_t nnz = h->nnz, nu = h->n_users;
    const int32_t *indptr = h->h_indptr.data(), *indices = h->h_indices.data();
    const int64_t cap = std::max<int64_t>(1, nnz / W / 2);
    std::vector<int64_t> load((size_t)W, 0);
    std::vector<int32_t> owner((size_t)nu, -1);
    // shared users first: their positions are dealt in equal contiguous blocks
    int64_t n_shared_pos = 0;
    for (int64_t u = 0; u < nu; ++u)
        if (indptr[u + 1] - indptr[u] > cap) n_shared_pos += indptr[u + 1] - indptr[u];
    const int64_t blk = (n_shared_pos + W - 1) / W;
    for (int64_t w = 0; w < W && blk > 0; ++w) load[(size_t)w] = std::max<int64_t>(0, std::min(blk, n_shared_pos - w * blk));
    // exclusive users: heaviest first onto the least loaded wave
    std::vector<int32_t> order;
    order.reserve((size_t)nu);
    for (int64_t u = 0; u < nu; ++u) {
        const int64_t d = indptr[u + 1] - indptr[u];
        if (d > 0 && d <= cap) order.push_back((int32_t)u);
    }
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
        return indptr[x + 1] - indptr[x] > indptr[y + 1] - indptr[y];
    });
    typedef std::pair<int64_t, int64_t> LW;  // (load, wave)
    std::priority_queue<LW, std::vector<LW>, std::greater<LW>> heap;
    for (int64_t w = 0; w < W; ++w) heap.push(LW(load[(size_t)w], w));
    for (int32_t u : order) {
        LW t = heap.top();
        heap.pop();
        owner[(size_t)u] = (int32_t)t.second;
        t.first += indptr[u + 1] - indptr[u];
        load[(size_t)t.second] = t.first;
        heap.push(t);
    }
    h->h_wave_ptr.assign((size_t)W + 1, 0);
    for (int64_t w = 0; w < W; ++w) h->h_wave_ptr[(size_t)w + 1] = h->h_wave_ptr[(size_t)w] + load[(size_t)w];
    REQUIRE(h->h_wave_ptr[(size_t)W] == nnz, "ownership tables do not cover the interaction matrix");
    h->h_own_u.resize((size_t)nnz);
    h->h_own_i.resize((size_t)nnz);
    std::vector<int64_t> cursor(h->h_wave_ptr.begin(), h->h_wave_ptr.end() - 1);
    int64_t sp = 0;  // running index over shared positions
    for (int64_t u = 0; u < nu; ++u) {
        const int64_t d = indptr[u + 1] - indptr[u];
        if (d == 0) continue;
        if (d > cap) {
            for (int32_t p = indptr[u]; p < indptr[u + 1]; ++p, ++sp) {
                const int64_t w = sp / blk;
                const int64_t pos = cursor[(size_t)w]++;
                h->h_own_u[(size_t)pos] = ~(int32_t)u;
                h->h_own_i[(size_t)pos] = indices[p];
            }
        } else {
            const int64_t w = owner[(size_t)u];
            for (int32_t p = indptr[u]; p < indptr[u + 1]; ++p) {
                const int64_t pos = cursor[(size_t)w]++;
                h->h_own_u[(size_t)pos] = (int32_t)u;
                h->h_own_i[(size_t)pos] = indices[p];
            }
        }
    }
    h->own_u.ensure((size_t)nnz);
    h->own_i.ensure((size_t)nnz);
    h->wave_ptr.ensure((size_t)W + 1);
    h->own_u.upload(h->h_own_u.data(), (size_t)nnz, h->stream);
    h->own_i.upload(h->h_own_i.data(), (size_t)nnz, h->stream);
    h->own_tmax = 0;
    for (int64_t w = 0; w < W; ++w)
        h->own_tmax = std::max(h->own_tmax, (h->h_wave_ptr[(size_t)w + 1] - h->h_wave_ptr[(size_t)w] + kWave - 1) / kWave);
    h->wave_ptr.upload(h->h_wave_ptr.data(), (size_t)W + 1, h->stream);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->own_waves = W;
}

static bool hogwild_uses_ownership(cornac_hip_bpr_t h, int flags) {
    // bit0 plain stores, bit1 float4 layout, bit2 explicit opt-out; needs one triplet per wave step
    // (k > 32) and at least one 64-sample tile per wave and epoch
    flags &= ~8;  // bit3 (dense bias) is independent of ownership
    if ((flags & 7) != 0 || h->k <= 32 || h->k > 256) return false;  // bits >= 8 are profiling ablations
    const int64_t W = (int64_t)device_info(h->device).cus * 8 * kWavesPerBlock;
    return h->nnz >= W * kWave;
}

static void launch_hogwild(cornac_hip_bpr_t h, HogArgs a, int flags) {
    const DeviceInfo &di = device_info(h->device);
    const bool owned = hogwild_uses_ownership(h, flags);
    if (!owned) flags |= 4;
    HogKernel kern = (flags & 1) ? pick_hogwild_kernel<false>(h->k, flags) : pick_hogwild_kernel<true>(h->k, flags);
    // persistent grid: exactly the number of workgroups that are co-resident, so the tile loop of
    // every wave starts at once (no second dispatch round with a ragged tail)
    if (h->hog_kernel != kern) {
        int per_cu = 0;
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kBlock, 0));
        h->hog_kernel = kern;
        h->hog_blocks_per_cu = std::max(1, std::min(per_cu, 8));
    }
    int grid;
    if (owned) {
        grid = di.cus * h->hog_blocks_per_cu;
        build_ownership(h, (int64_t)grid * kWavesPerBlock);
        a.own_u = h->own_u.p;
        a.own_i = h->own_i.p;
        a.wave_ptr = h->wave_ptr.p;
        a.own_tmax = h->own_tmax;
    } else {
        const int64_t n_tiles = (a.n + kWave - 1) / kWave;
        const int64_t want_blocks = (n_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
        grid = (int)std::max<int64_t>(1, std::min<int64_t>(want_blocks, (int64_t)di.cus * h->hog_blocks_per_cu));
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, h->stream, a);
    HIP_CHECK(hipGetLastError());
}

// ---- binned (segmented) item updates: bucket tables and launch -----------------------------------------------
// Expected messages per epoch of item i: one per positive draw (its degree) plus its share of the negative draws.
// Items whose expectation per CHUNK exceeds hot_threshold are "hot": they keep device-scope atomics (their rows
// must not see hundreds of updates computed from one stale copy).  The cold items are dealt to n_buckets buckets by
// LPT on that weight, so the apply kernel's workgroups carry equal loads whatever the popularity skew.
static void build_item_buckets(cornac_hip_bpr_t h, int n_buckets, int neg_population, int64_t chunk, int hot_threshold) {
    if (h->bin_buckets == n_buckets && h->bin_neg_population == neg_population && h->bin_chunk == chunk &&
        h->bin_hot_threshold == hot_threshold)
        return;
    const int64_t ni = h->n_items;
    std::vector<double> w((size_t)ni, 0.0);
    for (int64_t p = 0; p < h->nnz; ++p) w[(size_t)h->h_indices[(size_t)p]] += 1.0;
    const double uni = (double)h->nnz / (double)ni;
    for (int64_t i = 0; i < ni; ++i) w[(size_t)i] += neg_population == CORNAC_HIP_NEG_POPULARITY ? w[(size_t)i] : uni;
    const double per_chunk = (double)chunk / (double)h->nnz;
    std::vector<int32_t> order, hot;
    order.reserve((size_t)ni);
    for (int64_t i = 0; i < ni; ++i) {
        if (w[(size_t)i] * per_chunk > (double)hot_threshold) hot.push_back((int32_t)i);
        else order.push_back((int32_t)i);
    }
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return w[(size_t)x] > w[(size_t)y]; });
    typedef std::pair<double, int> LB;  // (load, bucket)
    std::priority_queue<LB, std::vector<LB>, std::greater<LB>> heap;
    for (int b = 0; b < n_buckets; ++b) heap.push(LB(0.0, b));
    std::vector<int32_t> bucket_of((size_t)ni, -1), rows((size_t)n_buckets, 0);
    for (int32_t i : order) {
        LB t = heap.top();
        heap.pop();
        bucket_of[(size_t)i] = t.second;
        ++rows[(size_t)t.second];
        t.first += w[(size_t)i];
        heap.push(t);
    }
    std::vector<int32_t> ptr((size_t)n_buckets + 1, 0), items(order.size()), slot((size_t)ni);
    for (int b = 0; b < n_buckets; ++b) ptr[(size_t)b + 1] = ptr[(size_t)b] + rows[(size_t)b];
    std::vector<int32_t> cur(ptr.begin(), ptr.end() - 1);
    h->bin_max_rows = 1;
    for (int32_t i : order) {
        const int b = bucket_of[(size_t)i];
        const int32_t local = cur[(size_t)b] - ptr[(size_t)b];
        REQUIRE(local < 65536, "item bucket too large for the 16-bit local row index");
        items[(size_t)cur[(size_t)b]++] = i;
        slot[(size_t)i] = (int32_t)((uint32_t)b << 16 | (uint32_t)local);
    }
    for (size_t t = 0; t < hot.size(); ++t) slot[(size_t)hot[t]] = -(int32_t)t - 1;
    for (int b = 0; b < n_buckets; ++b) h->bin_max_rows = std::max(h->bin_max_rows, rows[(size_t)b]);
    h->bin_n_hot = (int)hot.size();
    h->bin_item_slot.ensure((size_t)ni);
    h->bin_bucket_ptr.ensure((size_t)n_buckets + 1);
    h->bin_bucket_items.ensure(std::max<size_t>(1, items.size()));
    h->bin_hot_items.ensure(std::max<size_t>(1, hot.size()));
    h->bin_hot_bias.ensure(std::max<size_t>(1, hot.size()) * kBiasStride);
    h->bin_item_slot.upload(slot.data(), (size_t)ni, h->stream);
    h->bin_bucket_ptr.upload(ptr.data(), (size_t)n_buckets + 1, h->stream);
    if (!items.empty()) h->bin_bucket_items.upload(items.data(), items.size(), h->stream);
    if (!hot.empty()) h->bin_hot_items.upload(hot.data(), hot.size(), h->stream);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->bin_buckets = n_buckets;
    h->bin_neg_population = neg_population;
    h->bin_chunk = chunk;
    h->bin_hot_threshold = hot_threshold;
}

static int env_int(const char *name, int dflt) { return prof_env_int(name, dflt); }  // profile builds only
typedef void (*BinTripletKernel)(const HogArgs, const BinArgs);
typedef void (*BinApplyKernel)(const BinApplyArgs);
static void pick_binned_kernels(int k, int occ, BinTripletKernel *ka, BinApplyKernel *kb) {
    if (k <= 64) { *ka = occ >= 2 ? bpr_binned_triplet_kernel<1, 4, 2> : env_int("CORNAC_HIP_BIN_UNRA", 4) >= 8 ? bpr_binned_triplet_kernel<1, 8, 1> : bpr_binned_triplet_kernel<1, 4, 1>; *kb = env_int("CORNAC_HIP_BIN_UNRB", 4) >= 4 ? bpr_binned_apply_kernel<1, 4> : bpr_binned_apply_kernel<1, 2>; }
    else if (k <= 128) { *ka = occ >= 2 ? bpr_binned_triplet_kernel<2, 2, 2> : bpr_binned_triplet_kernel<2, 2, 1>; *kb = bpr_binned_apply_kernel<2, 2>; }
    else if (k <= 192) { *ka = bpr_binned_triplet_kernel<3, 2, 1>; *kb = bpr_binned_apply_kernel<3, 1>; }
    else { *ka = bpr_binned_triplet_kernel<4, 1, 1>; *kb = bpr_binned_apply_kernel<4, 1>; }
}


// Plan of the binned path for this handle (grid of the triplet kernel, LDS sizes, chunk length); ok == false when
// the shape does not qualify: k outside (32, 256], too few interactions for one tile per wave, an item bucket that
// does not fit a workgroup's LDS, or an experiment switch of the fused kernel is set.
static constexpr size_t kBinMaxLds = 156 * 1024;  // of the 160 KiB per CU
struct BinPlan {
    bool ok = false;
    int grid_a = 0, n_buckets = 0;
    size_t lds_a = 0;
    int64_t chunk = 0;
    int cap = 0;  // messages per segment (a multiple of 64)
    int hot_threshold = 0;
};
static BinPlan plan_binned(cornac_hip_bpr_t h, int flags, float lr) {
    BinPlan pl;
    // opt-in (hogwild_flags bit 6): measured slower than the fused atomic kernel at the ML-20M shape
    if ((flags & 0xff) != 64 || h->k <= 32 || h->k > 256) return pl;
    const DeviceInfo &di = device_info(h->device);
    pl.n_buckets = di.cus;
    const int R = (h->k + kWave - 1) / kWave;
    const int64_t max_rows = (h->n_items + pl.n_buckets - 1) / pl.n_buckets + 1;  // LPT keeps row counts close to even
    pl.lds_a = ((size_t)pl.n_buckets + (size_t)kBinWaves * 5 * kWave) * sizeof(int32_t);
    if ((size_t)max_rows * (size_t)(2 * (kWave * R + 4) + 1) * sizeof(float) > kBinMaxLds || max_rows >= 65536) return pl;
    BinTripletKernel ka;
    BinApplyKernel kb;
    if (h->bin_wg_per_cu == 0) {
        const int want = std::max(1, std::min(2, env_int("CORNAC_HIP_BIN_WG_PER_CU", 1)));
        pick_binned_kernels(h->k, want, &ka, &kb);
        int per_cu = 0;
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ka, kBinBlock, pl.lds_a));
        h->bin_wg_per_cu = std::max(1, std::min(per_cu, want));
    }
    pl.grid_a = di.cus * h->bin_wg_per_cu;
    if (pl.grid_a > 1024) return pl;  // the apply kernel keeps one segment count per lane and wave
    if (h->nnz < 8 * (int64_t)pl.grid_a * kBinWaves * kWave) return pl;  // too few tiles per wave and chunk to fill the pipeline
    // Chunk length: every launch pays a pipeline fill/drain of about one tile per wave (measured ~0.16 ms), so chunks
    // should be few; every cold row receives up to hot_threshold updates computed from the chunk's start, so they
    // must not be long.  2 M triplets (a tenth of an ML-20M epoch) by default, never more than a quarter of an epoch.
    const int64_t want = env_int("CORNAC_HIP_BIN_CHUNK", 0) > 0 ? env_int("CORNAC_HIP_BIN_CHUNK", 0) : (int64_t(1) << 21);
    pl.chunk = std::max<int64_t>((int64_t)pl.grid_a * kBinWaves * kWave, std::min<int64_t>(want, h->nnz / 4));
    // a segment (bucket x producing workgroup) receives 2 chunk / (buckets x workgroups) messages on average: size
    // it for twice the mean (the surplus of an overflowing segment falls back to atomics)
    const int64_t mean = 2 * pl.chunk / ((int64_t)pl.n_buckets * pl.grid_a) + 1;
    pl.cap = (int)std::min<int64_t>(1024, (2 * mean + kWave - 1) / kWave * kWave);
    // rows with more than hot_threshold messages per chunk keep device-scope atomics: the apply kernel serialises the
    // messages of one row (~250 cycles each under its lock), which must stay well below the kernel's duration
    pl.hot_threshold = env_int("CORNAC_HIP_BIN_HOT", 0) > 0 ? env_int("CORNAC_HIP_BIN_HOT", 0) : 4096;
    pl.ok = true;
    return pl;
}

static void fill_hog_args(cornac_hip_bpr_t h, HogArgs &a, int64_t n, float lr, float reg, int use_bias,
                          int neg_population, int flags) {
    a.user_ids = h->user_ids.p; a.indices = h->indices.p; a.indptr = h->indptr.p;
    a.U = h->U.p; a.V = h->V.p; a.B = h->B.p;
    a.bstride = 1;
    a.vpitch = h->k;
    a.counters = h->counters.p;
    a.n = n;
    a.s_begin = (uint64_t)h->hog_offset;
    a.seed = h->hog_seed;
    a.epoch = h->hog_epoch;
    a.n_pos = (uint32_t)h->nnz;
    a.neg_items = h->neg_pop_n ? h->neg_pop.p : h->indices.p;
    a.n_neg = neg_population == CORNAC_HIP_NEG_POPULARITY ? (uint32_t)(h->neg_pop_n ? h->neg_pop_n : h->nnz) : (uint32_t)h->n_items;
    a.th_pos = lemire_thresh(a.n_pos);
    a.th_neg = lemire_thresh(a.n_neg);
    a.k = h->k; a.neg_population = neg_population; a.use_bias = use_bias;
    a.lr = lr; a.reg = reg;
    a.own_u = nullptr; a.own_i = nullptr; a.wave_ptr = nullptr; a.own_tmax = 0;
    a.xcd_claim = nullptr;
    a.nnz = h->nnz;
    a.ablate = (flags >> 8) & 0xff;
}

static void advance_hog_offset(cornac_hip_bpr_t h, int64_t n) {
    h->hog_offset += n;
    if (h->hog_offset >= h->nnz) {
        h->hog_offset = 0;
        ++h->hog_epoch;
    }
}

static void binned_enqueue(cornac_hip_bpr_t h, const BinPlan &pl, int64_t n_samples, float lr, float reg, int use_bias,
                           int neg_population, int flags) {
    BinTripletKernel ka;
    BinApplyKernel kb;
    pick_binned_kernels(h->k, h->bin_wg_per_cu, &ka, &kb);
    if (!h->bin_attr_set) {
        HIP_CHECK(hipFuncSetAttribute((const void *)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinMaxLds));
        h->bin_attr_set = true;
    }
    const int64_t W = (int64_t)pl.grid_a * kBinWaves;
    build_ownership(h, W);
    build_item_buckets(h, pl.n_buckets, neg_population, pl.chunk, pl.hot_threshold);
    const int R = (h->k + kWave - 1) / kWave;
    const size_t lds_b = (size_t)h->bin_max_rows * (size_t)(2 * (kWave * R + 4) + 1) * sizeof(float);
    REQUIRE(lds_b <= kBinMaxLds, "item bucket does not fit the LDS");
    // tiles of 64 samples per wave in one launch: ceil(own_tmax * chunk / nnz) + 1 covers every [lo_tile, hi_tile)
    const int64_t snap_tiles = std::min<int64_t>(h->own_tmax, (h->own_tmax * pl.chunk + h->nnz - 1) / h->nnz + 1);
    REQUIRE(W * snap_tiles * kWave < (int64_t(1) << 32), "snapshot index exceeds 32 bits");
    h->bin_seg.ensure((size_t)pl.n_buckets * pl.grid_a * pl.cap);
    h->bin_seg_count.ensure((size_t)pl.n_buckets * pl.grid_a);
    h->bin_snap.ensure((size_t)W * snap_tiles * kWave * h->k);
    const unsigned hgrid = (unsigned)((h->bin_n_hot + kBlock - 1) / kBlock);
    int64_t left = n_samples;
    while (left > 0) {
        const int64_t n = std::min(std::min(left, h->nnz - h->hog_offset), pl.chunk);
        HogArgs a;
        fill_hog_args(h, a, n, lr, reg, use_bias, neg_population, flags);
        a.own_u = h->own_u.p; a.own_i = h->own_i.p; a.wave_ptr = h->wave_ptr.p; a.own_tmax = h->own_tmax;
        BinArgs g;
        g.item_slot = h->bin_item_slot.p;
        g.seg = reinterpret_cast<v4u *>(h->bin_seg.p);
        g.seg_count = h->bin_seg_count.p;
        g.snap = h->bin_snap.p;
        g.hot_bias = h->bin_hot_bias.p;
        g.n_buckets = pl.n_buckets;
        g.cap = pl.cap;
        g.snap_tiles = (int)snap_tiles;
        BinApplyArgs p;
        p.bucket_ptr = h->bin_bucket_ptr.p; p.bucket_items = h->bin_bucket_items.p;
        p.seg = reinterpret_cast<const v4u *>(h->bin_seg.p); p.seg_count = h->bin_seg_count.p;
        p.snap = h->bin_snap.p; p.V = h->V.p; p.B = h->B.p;
        p.n_seg = pl.grid_a; p.cap = pl.cap; p.k = h->k; p.use_bias = use_bias; p.max_rows = h->bin_max_rows; p.lr = lr; p.reg = reg;
        p.ablate = a.ablate;
        if (h->bin_n_hot)
            hipLaunchKernelGGL(hot_bias_pad_kernel, dim3(hgrid), dim3(kBlock), 0, h->stream, h->B.p, h->bin_hot_items.p,
                               h->bin_hot_bias.p, h->bin_n_hot);
        h->ktimer.before(h->stream);
        hipLaunchKernelGGL(ka, dim3(pl.grid_a), dim3(kBinBlock), pl.lds_a, h->stream, a, g);
        if (!HOG_ABLATE(a, 16))
            hipLaunchKernelGGL(kb, dim3(pl.n_buckets), dim3(kBinBlock), lds_b, h->stream, p);
        h->ktimer.after(h->stream);
        if (h->bin_n_hot)
            hipLaunchKernelGGL(hot_bias_unpad_kernel, dim3(hgrid), dim3(kBlock), 0, h->stream, h->bin_hot_bias.p,
                               h->bin_hot_items.p, h->B.p, h->bin_n_hot);
        HIP_CHECK(hipGetLastError());
        advance_hog_offset(h, n);
        left -= n;
    }
}

// ---- XCD strata (bpr_strata.inc) ------------------------------------------------------------------------------
typedef void (*StrataKernel)(const HogArgs);
static StrataKernel pick_strata_kernel(int k) {
#ifdef CORNAC_PROFILE
    switch (k <= 64 ? prof_env_int("CORNAC_HIP_STRATA_VARIANT", 0) : 0) {  // A/B forms
        case 1: return bpr_strata_kernel<1, 4, 1>;
        case 2: return bpr_strata_kernel<1, 4, 2>;
        case 3: return bpr_strata_kernel<1, 2, 0>;
        case 4: return bpr_strata_kernel<1, 2, 1>;
        case 5: return bpr_strata_kernel<1, 8, 0>;
        default: break;
    }
#endif
    if (k <= 64) return bpr_strata_kernel<1, 4>;
    if (k <= 128) return bpr_strata_kernel<2, 2>;
    if (k <= 192) return bpr_strata_kernel<3, 2>;
    return bpr_strata_kernel<4, 1>;
}

// popularity ranks of the items (expected row touches per epoch = degree + the uniform share of the negative draws;
// the second term is the same for every item, so the order is by degree, ties by id) and the hot set: the leading
// ranks that (a) are touched at least hot_min_mult times as often as the average row and (b) together receive at most
// hot_permille / 1000 of all item-row touches.  Those rows keep device-scope atomics.
static void build_item_ranks(cornac_hip_bpr_t h) {
    if (h->strata_ranked) return;
    const int64_t ni = h->n_items, nnz = h->nnz;
    std::vector<int64_t> deg((size_t)ni, 0);
    for (int64_t t = 0; t < nnz; ++t) ++deg[(size_t)h->h_indices[(size_t)t]];
    h->h_rank_item.resize((size_t)ni);
    for (int64_t i = 0; i < ni; ++i) h->h_rank_item[(size_t)i] = (int32_t)i;
    std::stable_sort(h->h_rank_item.begin(), h->h_rank_item.end(),
                     [&](int32_t x, int32_t y) { return deg[(size_t)x] > deg[(size_t)y]; });
    std::vector<int32_t> item_rank((size_t)ni);
    for (int64_t r = 0; r < ni; ++r) item_rank[(size_t)h->h_rank_item[(size_t)r]] = (int32_t)r;
    const double neg_share = (double)nnz / (double)ni, mean_touch = 2.0 * neg_share;
    const double cap = 2.0 * (double)nnz * h->strata_hot_permille / 1000.0;
    double cum = 0;
    int64_t n_hot = 0;
    for (int64_t r = 0; r < ni; ++r) {
        const double touch = (double)deg[(size_t)h->h_rank_item[(size_t)r]] + neg_share;
        if (touch * 100.0 < mean_touch * h->strata_hot_min_mult_x100 || cum + touch > cap) break;
        cum += touch;
        n_hot = r + 1;
    }
    h->strata_n_hot = (uint32_t)n_hot;
    h->item_rank.ensure((size_t)ni);
    h->rank_item.ensure((size_t)ni);
    std::vector<int32_t> up_rank_item(h->h_rank_item);
#ifdef CORNAC_PROFILE
    if (prof_env_int("CORNAC_HIP_STRATA_CONTIG", 0)) {
        // experiment (how much do the translation misses cost?): a static deal whose partitions are contiguous id ranges
        // — "ranks" re-assigned so that the items [p n_full, (p + 1) n_full) form partition p under the fixed key below,
        // in popularity order inside a partition.  Used with the strata kernel only (the LDS-bin deal reads the same table).
        const uint32_t key = 0x12345u;
        const int64_t n_full = ni >> 3;
        std::vector<std::vector<int32_t>> part(8);
        for (int64_t r = 0; r < ni; ++r) {
            const int32_t it = h->h_rank_item[(size_t)r];
            if (it < 8 * n_full) part[(size_t)(it / n_full)].push_back(it);
        }
        for (int pp = 0; pp < 8; ++pp)
            for (int64_t g = 0; g < n_full; ++g) {
                const uint32_t code = (uint32_t)g * 8u + (((uint32_t)pp - strata_rot((uint32_t)g, key)) & 7u);
                up_rank_item[code] = part[(size_t)pp][(size_t)g];
            }
        for (int64_t it = 8 * n_full; it < ni; ++it) up_rank_item[(size_t)it] = (int32_t)it;
        for (int64_t r = 0; r < ni; ++r) item_rank[(size_t)up_rank_item[(size_t)r]] = (int32_t)r;
    }
#endif
    h->item_rank.upload(item_rank.data(), (size_t)ni, h->stream);
    h->rank_item.upload(up_rank_item.data(), (size_t)ni, h->stream);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->strata_ranked = true;
    h->strata_built = false;
    h->strata_code_waves = -1;
}

static bool hogwild_uses_strata(cornac_hip_bpr_t h, int64_t n_samples, int neg_population, int flags) {
    const int form = (flags >> 16) & 15;  // 2 = asked for; 0 = automatic: item tables of >= 2^20 rows
#ifdef CORNAC_PROFILE
    flags &= ~0xff00;  // (profile builds: the ablation bits 8..15 are honoured by the strata kernel too)
#endif
    if ((flags & 0xffff) != 0 || !(form == 2 || (form == 0 && h->n_items >= (int64_t(1) << 20)))) return false;
    (void)n_samples;  // any chunk of an epoch: a launch runs the partition phases that begin inside it
    return neg_population == CORNAC_HIP_NEG_UNIFORM && hogwild_uses_ownership(h, 0) && h->n_items >= 64 &&
           device_info(h->device).xcds == 8;
}

// grid / ownership / rank tables of the strata kernel; returns the grid width
static int strata_prepare(cornac_hip_bpr_t h) {
    const DeviceInfo &di = device_info(h->device);
    StrataKernel kern = pick_strata_kernel(h->k);
    if (h->strata_kernel != kern) {
        int per_cu = 0;
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kBlock, 0));
        h->strata_kernel = kern;
        h->strata_blocks_per_cu = std::max(1, std::min(per_cu, 8));
    }
    const int grid = di.cus * h->strata_blocks_per_cu;
    const int64_t W = (int64_t)grid * kWavesPerBlock;
    if (h->own_waves != W) h->strata_built = false;
    build_ownership(h, W);
    build_item_ranks(h);
    h->rec_u.ensure((size_t)h->nnz);
    h->rec_i.ensure((size_t)h->nnz);
    h->sptr.ensure((size_t)W * 8 + 1);
    REQUIRE(grid % 8 == 0, "the strata grid must be a multiple of the 8 XCDs");
    h->strata_claim.ensure(8 * 16);
    return grid;
}

static void strata_build_buckets(cornac_hip_bpr_t h, int grid, uint32_t key) {
    if (h->strata_built && h->strata_key_built == key) return;
    StrataArgs s;
    s.own_u = h->own_u.p; s.own_i = h->own_i.p; s.wave_ptr = h->wave_ptr.p; s.item_rank = h->item_rank.p;
    s.rec_u = h->rec_u.p; s.rec_i = h->rec_i.p; s.sptr = h->sptr.p;
    s.key = key; s.n_hot = h->strata_n_hot; s.n_waves = (int64_t)grid * kWavesPerBlock;
    h->strata_own_code.ensure((size_t)h->nnz);
    s.own_code = h->strata_own_code.p;
    if (h->strata_code_waves != h->own_waves) {  // once per ownership layout
        hipLaunchKernelGGL(strata_code_kernel, dim3((unsigned)std::min<int64_t>(4096, (h->nnz + kBlock - 1) / kBlock)),
                           dim3(kBlock), 0, h->stream, s, h->nnz);
        h->strata_code_waves = h->own_waves;
    }
    hipLaunchKernelGGL(strata_bucket_kernel, dim3(grid), dim3(kBlock), 0, h->stream, s);
    HIP_CHECK(hipGetLastError());
    h->strata_built = true;
    h->strata_key_built = key;
    ++h->strata_builds;
}

// 8 phase launches per epoch, buckets re-dealt when the epoch key changes.  A chunk of an epoch (the multi-GPU driver's
// exchange points) runs the phases whose nominal start p * nnz / 8 lies inside it, so consecutive chunks run every
// phase of the epoch exactly once.
static void strata_pack(cornac_hip_bpr_t h) {
    if (h->strata_packed) return;
    h->vb_kp = (h->k + kBiasStride - 1) / kBiasStride * kBiasStride;
    h->vb_pitch = h->vb_kp + kBiasStride;
    h->VB.ensure((size_t)h->total_items * h->vb_pitch);
    const int64_t n = (int64_t)h->total_items * h->vb_pitch;
    hipLaunchKernelGGL(strata_pack_kernel, dim3((unsigned)std::min<int64_t>((n + kBlock - 1) / kBlock, 65536)), dim3(kBlock), 0,
                       h->stream, h->V.p, h->B.p, h->total_items, h->k, h->vb_kp, h->vb_pitch, h->VB.p);
    HIP_CHECK(hipGetLastError());
    h->strata_packed = true;
}

static void strata_unpack(cornac_hip_bpr_t h) {
    if (!h->strata_packed) return;
    const int64_t n = (int64_t)h->total_items * h->k;
    hipLaunchKernelGGL(strata_unpack_kernel, dim3((unsigned)std::min<int64_t>((n + kBlock - 1) / kBlock, 65536)), dim3(kBlock), 0,
                       h->stream, h->VB.p, h->total_items, h->k, h->vb_kp, h->vb_pitch, h->V.p, h->B.p);
    HIP_CHECK(hipGetLastError());
    h->strata_packed = false;
}

static void strata_enqueue(cornac_hip_bpr_t h, int64_t n_samples, float lr, float reg, int use_bias, int flags) {
    const int grid = strata_prepare(h);
    // packed records only inside fit_epochs (strata_allow_pack) and only for tables the handle owns — or, in the chunk
    // API, when the caller said so (cornac_hip_bpr_chunk_records: the multi-GPU driver, which touches its dense replica
    // only through cornac_hip_bpr_table_delta_* between chunks and reads it after cornac_hip_bpr_sync)
    const bool packed = h->strata_allow_pack && ((h->V.owned && h->B.owned) || h->chunk_records) &&
                        !prof_env_set("CORNAC_HIP_STRATA_NO_PACK");
    if (packed) strata_pack(h); else strata_unpack(h);
    if (!packed) h->Bpad.ensure((size_t)h->total_items * kBiasStride);
    const unsigned bgrid = (unsigned)((h->total_items + kBlock - 1) / kBlock);
    int64_t left = n_samples;
    while (left > 0) {
        const int64_t n = std::min(left, h->nnz - h->hog_offset);
        const auto first_phase = [&](int64_t s) { return (int)((8 * (__int128)s + h->nnz - 1) / h->nnz); };
        const int p_lo = first_phase(h->hog_offset), p_hi = h->hog_offset + n >= h->nnz ? 8 : first_phase(h->hog_offset + n);
        if (p_hi > p_lo) {
            uint32_t key = strata_key(h->hog_seed, h->hog_epoch / (uint32_t)std::max(1, h->strata_rehash_period));
            if (prof_env_int("CORNAC_HIP_STRATA_CONTIG", 0)) key = 0x12345u;  // (profile builds: the static contiguous deal)
            strata_build_buckets(h, grid, key);
            HogArgs a;
            fill_hog_args(h, a, h->nnz, lr, reg, use_bias, CORNAC_HIP_NEG_UNIFORM, flags);
            const bool dense_bias = !packed && prof_env_int("CORNAC_HIP_STRATA_DENSE_BIAS", 0) != 0;  // (profile builds: experiment)
            a.B = dense_bias ? h->B.p : h->Bpad.p;
            a.bstride = dense_bias ? 1 : kBiasStride;
            if (packed) {
                a.V = h->VB.p;
                a.vpitch = h->vb_pitch;
                a.B = h->VB.p + h->vb_kp;
                a.bstride = h->vb_pitch;
            }
            a.rec_u = h->rec_u.p; a.rec_i = h->rec_i.p; a.rank_item = h->rank_item.p; a.sptr = h->sptr.p;
            a.strata_key = key; a.n_hot = h->strata_n_hot;
            HIP_CHECK(hipMemsetAsync(h->strata_claim.p, 0, 8 * 16 * sizeof(uint32_t), h->stream));
            if (!dense_bias && !packed)
                hipLaunchKernelGGL(bias_pad_kernel, dim3(bgrid), dim3(kBlock), 0, h->stream, h->B.p, h->Bpad.p, h->total_items);
            for (int ph = p_lo; ph < p_hi; ++ph) {
                a.phase = ph;
                a.xcd_claim = h->strata_claim.p + 16 * ph;
                h->ktimer.before(h->stream);
                hipLaunchKernelGGL(h->strata_kernel, dim3(grid), dim3(kBlock), 0, h->stream, a);
                h->ktimer.after(h->stream);
            }
            if (!dense_bias && !packed)
                hipLaunchKernelGGL(bias_unpad_kernel, dim3(bgrid), dim3(kBlock), 0, h->stream, h->Bpad.p, h->B.p, h->total_items);
            HIP_CHECK(hipGetLastError());
        }
        advance_hog_offset(h, n);
        left -= n;
    }
}

// ---- LDS-resident item bins (bpr_ldsbin.inc) ---------------------------------------------------------------------
typedef void (*LdsBinKernel)(const LdsBinArgs);
constexpr int kLdsbinPresampleDefault = 0;
static LdsBinKernel pick_ldsbin_kernel(int k, bool pop, bool exch = false, bool passing = false, bool conv = false, bool pre = false) {
    if (pre) {  // resident bins trained from pre-sampled tiles (ldsbin_presample_kernel; opt-in, uniform negatives, k <= 128)
        if (k <= 64) return bpr_ldsbin_kernel<1, 4, false, false, 1, false, true>;
        return bpr_ldsbin_kernel<2, 2, false, false, 1, false, true>;
    }
    if (conv) {  // conveyor launches (passing bins whose rows live in block buffers): the passing regime's shapes
        if (pop) {
            if (k <= 64) return bpr_ldsbin_kernel<1, 4, true, false, 1, true>;
            if (k <= 128) return bpr_ldsbin_kernel<2, 4, true, false, 1, true>;
            if (k <= 192) return bpr_ldsbin_kernel<3, 2, true, false, 1, true>;
            return bpr_ldsbin_kernel<4, 1, true, false, 1, true>;
        }
        if (k <= 64) return bpr_ldsbin_kernel<1, 4, false, false, 1, true>;
        if (k <= 128) return bpr_ldsbin_kernel<2, 4, false, false, 1, true>;
        if (k <= 192) return bpr_ldsbin_kernel<3, 2, false, false, 1, true>;
        return bpr_ldsbin_kernel<4, 1, false, false, 1, true>;
    }
    // passing bins (8-wave workgroups, two per CU): 4 triplets in flight per wave at k in 65..128 — measured on the
    // configs[4] slice (profiles/r05_exp_scale_passing.log): 2 in flight 38.1 ms, 3: 35.1, 4: 33.8 per epoch; with the
    // rows of a bin loaded 4 at a time 30.5; requesting the user rows two steps ahead instead of one: no gain (31.0)
    if (passing && !exch && k > 64 && k <= 128) {
#ifdef CORNAC_PROFILE
        if (prof_env_int("CORNAC_HIP_LDSBIN_UNR", 0) == 0)
#endif
            return pop ? bpr_ldsbin_kernel<2, 4, true> : bpr_ldsbin_kernel<2, 4>;
    }
    if (exch) {  // resident exchange (multi-GPU regime 1 inside one launch per epoch)
        if (pop) {
            if (k <= 64) return bpr_ldsbin_kernel<1, 4, true, true>;
            if (k <= 128) return bpr_ldsbin_kernel<2, 2, true, true>;
            if (k <= 192) return bpr_ldsbin_kernel<3, 2, true, true>;
            return bpr_ldsbin_kernel<4, 1, true, true>;
        }
        if (k <= 64) return bpr_ldsbin_kernel<1, 4, false, true>;
        if (k <= 128) return bpr_ldsbin_kernel<2, 2, false, true>;
        if (k <= 192) return bpr_ldsbin_kernel<3, 2, false, true>;
        return bpr_ldsbin_kernel<4, 1, false, true>;
    }
    if (pop) {
        if (k <= 64) return bpr_ldsbin_kernel<1, 4, true>;
        if (k <= 128) return bpr_ldsbin_kernel<2, 2, true>;
        if (k <= 192) return bpr_ldsbin_kernel<3, 2, true>;
        return bpr_ldsbin_kernel<4, 1, true>;
    }
#ifdef CORNAC_PROFILE
    if (k <= 64) switch (prof_env_int("CORNAC_HIP_LDSBIN_UNR", 0)) {
        case 1: return bpr_ldsbin_kernel<1, 1>;
        case 2: return bpr_ldsbin_kernel<1, 2>;
        case 8: return bpr_ldsbin_kernel<1, 8>;
        default: break;
    }
    else if (k <= 128) switch (prof_env_int("CORNAC_HIP_LDSBIN_UNR", 0)) {
        case 3: return bpr_ldsbin_kernel<2, 3>;
        case 4: return bpr_ldsbin_kernel<2, 4>;
        case 22: return bpr_ldsbin_kernel<2, 2, false, false, 2>;
        case 32: return bpr_ldsbin_kernel<2, 3, false, false, 2>;
        case 42: return bpr_ldsbin_kernel<2, 4, false, false, 2>;
        default: break;
    }
#endif
    if (k <= 64) return bpr_ldsbin_kernel<1, 4>;
    if (k <= 128) return bpr_ldsbin_kernel<2, 2>;
    if (k <= 192) return bpr_ldsbin_kernel<3, 2>;
    return bpr_ldsbin_kernel<4, 1>;
}

constexpr size_t kLbLdsBudget = 96 * 1024;   // of the 160 KiB per CU
constexpr size_t kLbLdsExclusive = 82 * 1024;  // requested at least: two workgroups never share a CU's LDS
static size_t ldsbin_lds_bytes(int cap, int k, int waves = kLbWaves) {
    const int kp = ((k + kWave - 1) / kWave) * kWave;
    return ((size_t)cap * (kp + 5) + 1) * sizeof(float) + (size_t)waves * 3 * kWave * sizeof(int32_t);
}

// Two regimes of the form (bpr_ldsbin.inc):
//   * resident bins — the whole item table fits the chip's LDS in <= lb_max_rounds (4) rounds of one 16-wave workgroup
//     per CU: a bin owns its CU for the epoch (>= kLbLdsExclusive requested so that no second workgroup joins it);
//   * passing bins — a large item table (configs[4]: 10 M x 128) passes through the LDS once per epoch in hundreds of
//     rounds: 8-wave workgroups with <= lb_pass_kb of LDS each, so that two of them share a CU and one loads / stores
//     its rows while the other trains.  An item row then costs ONE read and ONE write of HBM per epoch however often it
//     is drawn, and its updates are exact LDS read-modify-writes; worth it when a row is drawn a few times per epoch
//     (nnz >= lb_pass_min_draws_x100 / 100 x n_items), else the row traffic would exceed what the triplets need.
struct LbPlan {
    int bins = 0, block = kLbBlock, cap = 0;
    size_t lds = 0;
    bool passing = false;
};
static LbPlan ldsbin_plan(cornac_hip_bpr_t h) {
    LbPlan pl;
    const int cus = device_info(h->device).cus;
    if (h->cv_blocks) {
        // conveyor layout: passing bins (8-wave workgroups, <= lb_pass_kb of LDS), their number a multiple of the block count.
        // As few rows per bin as lb_min_candidates allows while a block still has few bins (a step launches one block per
        // ring: small tables get more, smaller bins so that the step fills more CUs), never more rows than the LDS holds.
        if (h->k > 256) return pl;
        const int waves = h->lb_pass_waves;
        const size_t budget = (size_t)h->lb_pass_kb << 10;
        const int kp = ((h->k + kWave - 1) / kWave) * kWave;
        const size_t fixed = sizeof(float) + (size_t)waves * 3 * kWave * sizeof(int32_t);
        if (budget <= fixed) return pl;
        const int64_t cap_lds = (int64_t)((budget - fixed) / ((size_t)(kp + 5) * sizeof(float)));
        const int64_t cap_min = std::max<int64_t>(h->lb_min_candidates, 16);
        if (cap_lds < cap_min) return pl;
        int64_t bins = (h->n_items + cap_lds - 1) / cap_lds;
        const int64_t more = std::min<int64_t>(h->n_items / (cap_min + cap_min / 3), (int64_t)2 * cus * h->cv_blocks);
        bins = std::max(bins, more);
        bins = std::max<int64_t>(1, (bins + h->cv_blocks - 1) / h->cv_blocks) * h->cv_blocks;
        const int64_t cap = (h->n_items + bins - 1) / bins;
        if (cap > cap_lds || bins > (int64_t(1) << 22) || bins * cap >= (int64_t(1) << 31)) return pl;
        pl.bins = (int)bins;
        pl.cap = (int)cap;
        pl.block = waves * kWave;
        pl.lds = ldsbin_lds_bytes((int)cap, h->k, waves);
        pl.passing = true;
        return pl;
    }
    if (h->k > 256 || h->nnz < (int64_t)cus * kLbWaves * kWave) return pl;
    // (profile builds: CORNAC_HIP_LDSBIN_MIN_ROUNDS / _RES_WAVES / _EXCL_KB vary the resident regime's bin count, the waves of a
    // bin's workgroup and the LDS floor that keeps a CU to one workgroup — the experiments of DESIGN.md 7)
    const int res_waves = prof_env_int("CORNAC_HIP_LDSBIN_RES_WAVES", kLbWaves);
    for (int rounds = prof_env_int("CORNAC_HIP_LDSBIN_MIN_ROUNDS", 1); rounds <= h->lb_max_rounds; ++rounds) {
        const int64_t bins = (int64_t)cus * rounds;
        const int64_t cap = (h->n_items + bins - 1) / bins;
        if (cap < h->lb_min_candidates) return pl;  // negatives would be drawn from too few items
        if (ldsbin_lds_bytes((int)cap, h->k, res_waves) <= (size_t)prof_env_int("CORNAC_HIP_LDSBIN_BUDGET_KB", (int)(kLbLdsBudget >> 10)) << 10) {
            pl.bins = (int)bins;
            pl.cap = (int)cap;
            pl.block = res_waves * kWave;
            pl.lds = std::max(ldsbin_lds_bytes((int)cap, h->k, res_waves), (size_t)prof_env_int("CORNAC_HIP_LDSBIN_EXCL_KB", (int)(kLbLdsExclusive >> 10)) << 10);
            return pl;
        }
    }
    if (!h->lb_pass_enable || (double)h->nnz * 100.0 < (double)h->n_items * h->lb_pass_min_draws_x100) return pl;
    const int waves = prof_env_int("CORNAC_HIP_LDSBIN_PASS_WAVES", h->lb_pass_waves);
    const size_t budget = (size_t)prof_env_int("CORNAC_HIP_LDSBIN_PASS_KB", h->lb_pass_kb) << 10;
    const int kp = ((h->k + kWave - 1) / kWave) * kWave;
    const size_t fixed = sizeof(float) + (size_t)waves * 3 * kWave * sizeof(int32_t);
    if (budget <= fixed) return pl;
    int64_t cap = (int64_t)((budget - fixed) / ((size_t)(kp + 5) * sizeof(float)));
    if (cap < h->lb_min_candidates) return pl;
    int64_t bins = (h->n_items + cap - 1) / cap;
    bins = ((bins + cus - 1) / cus) * cus;
    cap = (h->n_items + bins - 1) / bins;
    if (cap < h->lb_min_candidates || bins > (int64_t(1) << 22)) return pl;
    pl.bins = (int)bins;
    pl.cap = (int)cap;
    pl.block = waves * kWave;
    pl.lds = ldsbin_lds_bytes((int)cap, h->k, waves);
    pl.passing = true;
    return pl;
}
// bins = a multiple of the CU count; 0 = this shape does not use the form
static int ldsbin_plan_bins(cornac_hip_bpr_t h) { return ldsbin_plan(h).bins; }

static bool hogwild_uses_ldsbin(cornac_hip_bpr_t h, int64_t n_samples, int neg_population, int flags) {
    const int form = (flags >> 16) & 15;
#ifdef CORNAC_PROFILE
    flags &= ~0xff00;
#endif
    if (!(form == 0 || form == 3) || (flags & 0xffff) != 0) return false;
    // uniform (BPR) and popularity-weighted (WBPR) negatives both have a binned form — the latter weights by the handle's
    // OWN interactions; a caller-supplied population (the global popularity of a multi-GPU fit) takes the fused kernel
    if (neg_population == CORNAC_HIP_NEG_POPULARITY && h->neg_pop_n) return false;
    // any chunk of an epoch: a launch takes its share of every bin's draws — but a launch of passing bins moves the whole
    // item table through the LDS, so only launches of at least a quarter of an epoch take that regime
    const LbPlan pl = ldsbin_plan(h);
    return pl.bins > 0 && (!pl.passing || n_samples * 4 >= h->nnz);
}

static void ldsbin_build(cornac_hip_bpr_t h) {
    const LbPlan plan = ldsbin_plan(h);
    const int bins = plan.bins;
    if (h->lb_built && h->lb_bins == bins && h->lb_block == plan.block && h->lb_lds_bytes == plan.lds) return;
    build_item_ranks(h);
    const int64_t ni = h->n_items, nnz = h->nnz, nu = h->n_users;
    // CSC: users of every item, in user order
    std::vector<int32_t> cptr((size_t)ni + 1, 0), cusers((size_t)nnz);
    for (int64_t t = 0; t < nnz; ++t) ++cptr[(size_t)h->h_indices[(size_t)t] + 1];
    for (int64_t i = 0; i < ni; ++i) cptr[(size_t)i + 1] += cptr[(size_t)i];
    {
        std::vector<int32_t> cur(cptr.begin(), cptr.end() - 1);
        for (int64_t u = 0; u < nu; ++u)
            for (int32_t p = h->h_indptr[(size_t)u]; p < h->h_indptr[(size_t)u + 1]; ++p)
                cusers[(size_t)cur[(size_t)h->h_indices[(size_t)p]]++] = (int32_t)u;
    }
    // hot items: degree above hot_x1000 / 1000 of a bin's share of the interactions (they would serialise their bin).
    // Passing bins are scheduled dynamically over 2 x CUs slots, so what a heavy item costs is the tail its bin adds to the
    // epoch: the share that counts is a slot's, priced at a quarter (hot_x1000 = 75: degree > nnz / 13 653 at 256 CUs)
    const double share = plan.passing ? (double)nnz / (4.0 * device_info(h->device).cus) : (double)nnz / bins;
    int n_hot = 0;
    while (!h->cv_blocks && n_hot < ni &&
           (double)(cptr[(size_t)h->h_rank_item[(size_t)n_hot] + 1] - cptr[(size_t)h->h_rank_item[(size_t)n_hot]]) * 1000.0 >
               share * h->lb_hot_x1000)
        ++n_hot;
    std::vector<int32_t> hot_u, hot_i;
    {
        // item-major list, then shuffled once (stable order of a hash of the list index): a bin draws a contiguous run
        // of the list (ldsbin_level_kernel), and a run should spread over all hot items
        std::vector<int32_t> lu, li;
        for (int r = 0; r < n_hot; ++r) {
            const int32_t it = h->h_rank_item[(size_t)r];
            for (int32_t p = cptr[(size_t)it]; p < cptr[(size_t)it + 1]; ++p) {
                lu.push_back(cusers[(size_t)p]);
                li.push_back(it);
            }
        }
        std::vector<uint32_t> order(lu.size());
        for (size_t t = 0; t < order.size(); ++t) order[t] = (uint32_t)t;
        std::stable_sort(order.begin(), order.end(), [](uint32_t x, uint32_t y) {
            return ldsbin_mix(x * 0x9E3779B1u + 0x5BD1E995u) < ldsbin_mix(y * 0x9E3779B1u + 0x5BD1E995u);
        });
        hot_u.resize(lu.size());
        hot_i.resize(lu.size());
        for (size_t t = 0; t < order.size(); ++t) {
            hot_u[t] = lu[order[t]];
            hot_i[t] = li[order[t]];
        }
    }
    h->lb_cptr.ensure((size_t)ni + 1);
    h->lb_cusers.ensure((size_t)nnz);
    h->lb_cptr.upload(cptr.data(), (size_t)ni + 1, h->stream);
    h->lb_cusers.upload(cusers.data(), (size_t)nnz, h->stream);
    h->lb_hot_u.ensure(std::max<size_t>(1, hot_u.size()));
    h->lb_hot_i.ensure(std::max<size_t>(1, hot_i.size()));
    if (!hot_u.empty()) {
        h->lb_hot_u.upload(hot_u.data(), hot_u.size(), h->stream);
        h->lb_hot_i.upload(hot_i.data(), hot_i.size(), h->stream);
    }
    // membership bitmap when it fits 2 GiB, else the kernel searches the CSR row
    const int64_t words = (ni + 31) / 32;
    h->lb_bm_words = 0;
    if (nu * words * 4 <= (int64_t(2) << 30)) {
        h->lb_bm_words = (int)words;
        h->lb_bitmap.ensure((size_t)(nu * words));
        HIP_CHECK(hipMemsetAsync(h->lb_bitmap.p, 0, (size_t)(nu * words) * sizeof(uint32_t), h->stream));
        hipLaunchKernelGGL(ldsbin_bitmap_kernel, dim3((unsigned)((nnz + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream,
                           h->user_ids.p, h->indices.p, nnz, h->lb_bm_words, h->lb_bitmap.p);
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->lb_bins = bins;
    h->lb_cap = plan.cap;
    h->lb_block = plan.block;
    h->lb_passing = plan.passing;
    h->lb_n_hot = n_hot;
    h->lb_n_hot_inter = (int)hot_u.size();
    h->lb_lds_bytes = plan.lds;
    const int n_groups = (int)((ni + bins - 1) / bins);
    h->lb_n_strata = std::max(1, n_groups / std::max(1, h->lb_strata_groups));
    h->lb_mass.ensure((size_t)bins);
    h->lb_cold.ensure((size_t)bins);
    h->lb_hot_off.ensure((size_t)bins + 1);
    HIP_CHECK(hipMemsetAsync(h->lb_mass.p, 0, (size_t)bins * sizeof(uint32_t), h->stream));
    h->lb_deal_valid = false;
    h->lb_built = true;
}

static const int32_t *ldsbin_rank_item(cornac_hip_bpr_t h) { return (h->cv_blocks && h->cv_own_order) ? h->cv_rank_item.p : h->rank_item.p; }

// the deal bookkeeping of (seed, epoch): cold masses per bin and the hot runs that level them (two small launches,
// once per epoch: the chunks of an epoch share them)
static void ldsbin_deal(cornac_hip_bpr_t h, uint64_t seed, uint32_t epoch, uint32_t key) {
    if (h->lb_deal_valid && h->lb_deal_seed == seed && h->lb_deal_epoch == epoch) return;
    LdsDealArgs d;
    d.cptr = h->lb_cptr.p; d.rank_item = ldsbin_rank_item(h);
    d.mass = h->lb_mass.p; d.cold_out = h->lb_cold.p; d.hot_off = h->lb_hot_off.p;
    d.key = key;
    d.n_items = (int32_t)h->n_items; d.n_bins = h->lb_bins; d.n_hot = h->lb_n_hot; d.n_hot_inter = h->lb_n_hot_inter;
    d.n_strata = h->lb_n_strata; d.hot_cost_x16 = h->lb_hot_cost_x16;
    // (64 workgroups cover the resident regime's tables in one pass; the 10 M items of a passing-bin table took 0.5 ms of the
    // 0.7 ms of per-epoch bookkeeping that way: up to 1024 workgroups)
    const unsigned grid = (unsigned)std::min<int64_t>(h->lb_passing ? 1024 : 64, (h->n_items + kLbBlock - 1) / kLbBlock);
    hipLaunchKernelGGL(ldsbin_mass_kernel, dim3(grid), dim3(kLbBlock), 0, h->stream, d);
    hipLaunchKernelGGL(ldsbin_level_kernel, dim3(1), dim3(kLbBlock), 0, h->stream, d);
    HIP_CHECK(hipGetLastError());
    h->lb_deal_valid = true;
    h->lb_deal_seed = seed;
    h->lb_deal_epoch = epoch;
}

static uint32_t ldsbin_key(uint64_t seed, uint32_t epoch) {
    uint32_t w[4];
    philox4x32_10(epoch, 0x1D5B1Au, 0u, 4u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    return w[0];
}

static void ldsbin_fill_args(cornac_hip_bpr_t h, LdsBinArgs &a, float lr, float reg, int use_bias, int neg_population,
                             int flags) {
    a.neg_pop = neg_population == CORNAC_HIP_NEG_POPULARITY ? 1 : 0;
    a.cptr = h->lb_cptr.p; a.cusers = h->lb_cusers.p; a.rank_item = ldsbin_rank_item(h);
    a.hot_u = h->lb_hot_u.p; a.hot_i = h->lb_hot_i.p; a.hot_off = h->lb_hot_off.p;
    a.indptr = h->indptr.p; a.indices = h->indices.p;
    a.bitmap = h->lb_bm_words ? h->lb_bitmap.p : nullptr;
    a.U = h->U.p; a.V = h->V.p; a.B = h->B.p;
    a.counters = h->counters.p;
    a.seed = h->hog_seed; a.epoch = h->hog_epoch; a.key = ldsbin_key(h->hog_seed, h->hog_epoch);
    a.n_items = (int32_t)h->n_items; a.n_bins = h->lb_bins; a.n_hot = h->lb_n_hot; a.n_hot_inter = h->lb_n_hot_inter;
    a.bm_words = h->lb_bm_words; a.cap = h->lb_cap; a.n_strata = h->lb_n_strata;
    a.k = h->k; a.use_bias = use_bias; a.lr = lr; a.reg = reg;
    a.ablate = ((flags >> 8) & 0xff) | (prof_env_int("CORNAC_HIP_LDSBIN_X", 0) << 8);
    a.wg_clock = nullptr;
    a.ex = LdsBinExchange{};
    a.pre_rec = nullptr; a.pre_cnt = nullptr; a.pre_base = nullptr; a.pre_waves = 0;
    a.conv_bpb = 0;
    for (int r = 0; r < kLbMaxRanges; ++r) {
        a.conv_first[r] = 0;
        a.conv_rows[r] = nullptr;
    }
}

// CORNAC_HIP_LDSBIN_PRESAMPLE=0 / 1 overrides the default (read once)
static bool ldsbin_presample_enabled() {
    static const int v = [] {
        const char *e = getenv("CORNAC_HIP_LDSBIN_PRESAMPLE");
        return e ? atoi(e) : kLdsbinPresampleDefault;
    }();
    return v != 0;
}

// one launch per epoch (or per chunk of an epoch: the multi-GPU driver's exchange points), one workgroup per bin
static void ldsbin_enqueue(cornac_hip_bpr_t h, int64_t n_samples, float lr, float reg, int use_bias, int neg_population,
                           int flags) {
    ldsbin_build(h);
    // resident bins: the draws in a launch of their own (ldsbin_presample_kernel), the bins' workgroups only train
    const bool pop = neg_population == CORNAC_HIP_NEG_POPULARITY;
    // (opt-in, CORNAC_HIP_LDSBIN_PRESAMPLE=1: measured at the ML-20M shape, the training launch alone takes 5.77 ms against the
    // fused launch's 5.86 — the draws were already hidden behind the updates — and the sampling launch adds 1.0 ms;
    // profiles/r06_headline_ablation.log)
    const bool pre = !h->lb_passing && !pop && h->k <= 128 && ldsbin_presample_enabled();
    LdsBinKernel kern = pick_ldsbin_kernel(h->k, pop, false, h->lb_passing, false, pre);
    HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lb_lds_bytes));
    int64_t left = n_samples;
    while (left > 0) {
        const int64_t n = std::min(left, h->nnz - h->hog_offset);
        LdsBinArgs a;
        ldsbin_fill_args(h, a, lr, reg, use_bias, neg_population, flags);
        a.s_begin = (uint64_t)h->hog_offset;
        a.n = (uint64_t)n;
        a.nnz = (uint64_t)h->nnz;
        ldsbin_deal(h, a.seed, a.epoch, a.key);
        if (pre) {
            const size_t pool = (size_t)n / 32 + 2 * (size_t)h->lb_bins + 8;   // >= the tiles of all bins at either tile width
            h->lb_pre_rec.ensure(pool * 3 * kWave);
            h->lb_pre_cnt.ensure(pool);
            h->lb_pre_base.ensure((size_t)h->lb_bins + 1);
            a.pre_rec = h->lb_pre_rec.p; a.pre_cnt = h->lb_pre_cnt.p; a.pre_base = h->lb_pre_base.p;
            a.pre_waves = (uint32_t)(h->lb_block / kWave);
            LdsTileBaseArgs tb;
            tb.cold = h->lb_cold.p; tb.hot_off = h->lb_hot_off.p; tb.base = h->lb_pre_base.p;
            tb.s_begin = a.s_begin; tb.n = a.n; tb.nnz = a.nnz; tb.n_bins = h->lb_bins; tb.n_waves = a.pre_waves;
            hipLaunchKernelGGL(ldsbin_tilebase_kernel, dim3(1), dim3(kLbBlock), 0, h->stream, tb);
            // enough sampling workgroups to fill the chip several times over: their memory latencies hide behind each other
            const int64_t tiles_per_bin = std::max<int64_t>(1, n / 64 / h->lb_bins);
            const unsigned splits = (unsigned)std::max<int64_t>(1, std::min<int64_t>(tiles_per_bin / 16, 64));
            const size_t pre_lds = ((size_t)3 * h->lb_cap + 1) * sizeof(uint32_t);
            hipLaunchKernelGGL((ldsbin_presample_kernel<false>), dim3(h->lb_bins, splits), dim3(kLbPreBlock), pre_lds, h->stream, a);
        }
#ifdef CORNAC_PROFILE
        const char *clock_path = getenv("CORNAC_HIP_LDSBIN_CLOCKS");
        if (clock_path) {
            h->lb_wg_clock.ensure((size_t)h->lb_bins * 2);
            a.wg_clock = h->lb_wg_clock.p;
        }
#endif
        h->ktimer.before(h->stream);
        hipLaunchKernelGGL(kern, dim3(h->lb_bins), dim3(h->lb_block), h->lb_lds_bytes, h->stream, a);
        h->ktimer.after(h->stream);
        HIP_CHECK(hipGetLastError());
#ifdef CORNAC_PROFILE
        if (clock_path) {  // per-bin durations next to the bin's cold / hot draw counts (100 MHz wall clock)
            std::vector<unsigned long long> clk((size_t)h->lb_bins * 2);
            std::vector<uint32_t> cold((size_t)h->lb_bins), off((size_t)h->lb_bins + 1);
            h->lb_wg_clock.download(clk.data(), clk.size(), h->stream);
            h->lb_cold.download(cold.data(), cold.size(), h->stream);
            h->lb_hot_off.download(off.data(), off.size(), h->stream);
            HIP_CHECK(hipStreamSynchronize(h->stream));
            if (FILE *f = fopen(clock_path, "a")) {
                unsigned long long t0 = ~0ull;
                for (int b = 0; b < h->lb_bins; ++b) t0 = std::min(t0, clk[2 * (size_t)b]);
                for (int b = 0; b < h->lb_bins; ++b)
                    fprintf(f, "%u %d %llu %llu %u %u\n", a.epoch, b, clk[2 * (size_t)b] - t0, clk[2 * (size_t)b + 1] - t0,
                            cold[(size_t)b], off[(size_t)b + 1] - off[(size_t)b]);
                fclose(f);
            }
        }
#endif
        advance_hog_offset(h, n);
        left -= n;
    }
}

// One whole epoch in ONE launch with the item-table exchanges of multi-GPU regime 1 inside it (bpr_ldsbin.inc, "resident
// exchange"): the launch publishes its rows' deltas at n_ex points and applies the landed sums itself.
static void ldsbin_resident_enqueue(cornac_hip_bpr_t h, float lr, float reg, int use_bias, int neg_population, int flags,
                                    const LdsBinExchange &ex) {
    ldsbin_build(h);
    LdsBinKernel kern = pick_ldsbin_kernel(h->k, neg_population == CORNAC_HIP_NEG_POPULARITY, true);
    HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lb_lds_bytes));
    LdsBinArgs a;
    ldsbin_fill_args(h, a, lr, reg, use_bias, neg_population, flags);
    a.s_begin = 0;
    a.n = (uint64_t)h->nnz;
    a.nnz = (uint64_t)h->nnz;
    a.ex = ex;
    ldsbin_deal(h, a.seed, a.epoch, a.key);
    h->ktimer.before(h->stream);
    hipLaunchKernelGGL(kern, dim3(h->lb_bins), dim3(h->lb_block), h->lb_lds_bytes, h->stream, a);
    h->ktimer.after(h->stream);
    HIP_CHECK(hipGetLastError());
    advance_hog_offset(h, h->nnz);
}

static void hogwild_enqueue(cornac_hip_bpr_t h, int64_t n_samples, float lr, float reg, int use_bias,
                            int neg_population, int flags) {
    REQUIRE(h->hog_seeded, "hogwild mode needs cornac_hip_bpr_seed_hogwild first");
    if (hogwild_uses_ldsbin(h, n_samples, neg_population, flags)) {
        ldsbin_enqueue(h, n_samples, lr, reg, use_bias, neg_population, flags);
        return;
    }
    if (hogwild_uses_strata(h, n_samples, neg_population, flags)) {
        strata_enqueue(h, n_samples, lr, reg, use_bias, flags);
        return;
    }
    flags &= 0xffff & ~128;  // bits 16..19 select the form, bit7 only opts out of the LDS-bin / strata forms
    const BinPlan pl = plan_binned(h, flags, lr);
    if (pl.ok) {
        binned_enqueue(h, pl, n_samples, lr, reg, use_bias, neg_population, flags);
        return;
    }
    flags &= ~(32 | 64);
    if ((flags & 8) == 0) h->Bpad.ensure((size_t)h->total_items * kBiasStride);
    int64_t left = n_samples;
    while (left > 0) {
        const int64_t n = std::min(left, h->nnz - h->hog_offset);
        HogArgs a;
        fill_hog_args(h, a, n, lr, reg, use_bias, neg_population, flags);
        const bool pad_bias = (flags & 8) == 0;  // bit3: experiment switch, dense bias table
        a.B = pad_bias ? h->Bpad.p : h->B.p;
        a.bstride = pad_bias ? kBiasStride : 1;
        const unsigned bgrid = (unsigned)((h->total_items + kBlock - 1) / kBlock);
        if (pad_bias)
            hipLaunchKernelGGL(bias_pad_kernel, dim3(bgrid), dim3(kBlock), 0, h->stream, h->B.p, h->Bpad.p,
                               h->total_items);
        h->ktimer.before(h->stream);
        launch_hogwild(h, a, flags);
        h->ktimer.after(h->stream);
        if (pad_bias)
            hipLaunchKernelGGL(bias_unpad_kernel, dim3(bgrid), dim3(kBlock), 0, h->stream, h->Bpad.p, h->B.p,
                               h->total_items);
        advance_hog_offset(h, n);
        left -= n;
    }
}

static void fetch_counters(cornac_hip_bpr_t h, int64_t *correct, int64_t *skipped) {
    unsigned long long c[4];
    HIP_CHECK(hipMemcpyAsync(c, h->counters.p, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipMemsetAsync(h->counters.p, 0, sizeof c, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    if (correct) *correct += (int64_t)c[0];
    if (skipped) *skipped += (int64_t)c[1];
    h->strata_misplaced += (int64_t)c[2];
    h->lb_lock_timeouts += (int64_t)c[3];
}

extern "C" {

int cornac_hip_bpr_fit_epochs(cornac_hip_bpr_t h, int n_epochs, float lr, float reg, int use_bias, int neg_population,
                              int mode, int hogwild_flags, int64_t *correct, int64_t *skipped) {
    return guarded([&] {
        bpr_check_keep_packed(h);
        REQUIRE(n_epochs >= 0, "n_epochs must be >= 0");
        REQUIRE(neg_population == CORNAC_HIP_NEG_UNIFORM || neg_population == CORNAC_HIP_NEG_POPULARITY,
                "unknown neg_population %d", neg_population);
        REQUIRE(mode == CORNAC_HIP_MODE_DETERMINISTIC || mode == CORNAC_HIP_MODE_HOGWILD, "unknown mode %d", mode);
        REQUIRE(!h->f64, "the handle holds float64 tables: use cornac_hip_bpr_fit_epochs_f64 (sequential semantics only)");
        // the strata form keeps its packed item records from one fit_epochs call to the next; any other form (and any
        // other entry point: bpr_check) gets the dense tables back first
        struct AllowPack {
            cornac_hip_bpr_t h;
            ~AllowPack() { h->strata_allow_pack = false; }
        } allow_guard{h};
        h->strata_allow_pack = mode == CORNAC_HIP_MODE_HOGWILD && h->hog_seeded &&
                               !hogwild_uses_ldsbin(h, h->nnz, neg_population, hogwild_flags) &&
                               hogwild_uses_strata(h, h->nnz, neg_population, hogwild_flags);
        if (!h->strata_allow_pack) strata_unpack(h);
        if (correct) *correct = 0;
        if (skipped) *skipped = 0;
        for (double &t : h->timing) t = 0;
        Timer total;
        HIP_CHECK(hipMemsetAsync(h->counters.p, 0, 4 * sizeof(unsigned long long), h->stream));
        for (int e = 0; e < n_epochs; ++e) {
            if (mode == CORNAC_HIP_MODE_DETERMINISTIC) {
                bpr_epoch_deterministic(h, lr, reg, use_bias, neg_population);
            } else {
                Timer t_k;
                hogwild_enqueue(h, h->nnz, lr, reg, use_bias, neg_population, hogwild_flags);
                h->timing[2] += t_k.ms();
            }
        }
        Timer t_sync;
        fetch_counters(h, correct, skipped);
        if (mode == CORNAC_HIP_MODE_HOGWILD) h->timing[2] += t_sync.ms();
        h->timing[3] = total.ms();
    });
}

int cornac_hip_bpr_fit_epochs_f64(cornac_hip_bpr_t h, int n_epochs, double lr, double reg, int use_bias, int neg_population,
                                  int64_t *correct, int64_t *skipped) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(h->f64, "set float64 tables first (cornac_hip_bpr_set_factors_f64)");
        REQUIRE(n_epochs >= 0, "n_epochs must be >= 0");
        REQUIRE(neg_population == CORNAC_HIP_NEG_UNIFORM || neg_population == CORNAC_HIP_NEG_POPULARITY,
                "unknown neg_population %d", neg_population);
        if (correct) *correct = 0;
        if (skipped) *skipped = 0;
        for (double &t : h->timing) t = 0;
        Timer total;
        HIP_CHECK(hipMemsetAsync(h->counters.p, 0, 4 * sizeof(unsigned long long), h->stream));
        for (int e = 0; e < n_epochs; ++e) bpr_epoch_deterministic(h, lr, reg, use_bias, neg_population);
        fetch_counters(h, correct, skipped);
        h->timing[3] = total.ms();
    });
}

int cornac_hip_bpr_hogwild_enqueue(cornac_hip_bpr_t h, int64_t n_samples, float lr, float reg, int use_bias,
                                   int neg_population, int hogwild_flags) {
    return guarded([&] {
        if (h && h->chunk_records) bpr_check_keep_packed(h); else bpr_check(h);
        REQUIRE(n_samples >= 0, "n_samples must be >= 0");
        REQUIRE(!h->f64, "the handle holds float64 tables: use cornac_hip_bpr_fit_epochs_f64 (sequential semantics only)");
        struct AllowPack {
            cornac_hip_bpr_t h;
            ~AllowPack() { h->strata_allow_pack = false; }
        } allow_guard{h};
        h->strata_allow_pack = h->chunk_records && h->hog_seeded && !hogwild_uses_ldsbin(h, n_samples, neg_population, hogwild_flags) &&
                               hogwild_uses_strata(h, n_samples, neg_population, hogwild_flags);
        if (!h->strata_allow_pack) strata_unpack(h);
        hogwild_enqueue(h, n_samples, lr, reg, use_bias, neg_population, hogwild_flags);
    });
}

static LdsBinExchange resident_exchange_args(cornac_hip_bpr_t h, int n_exchanges, int rule, float *d_base, float *d_buckets,
                                             int64_t bucket_stride, float *d_keeps, int64_t keep_stride,
                                             uint32_t *d_arrive, const uint32_t *d_landed, uint32_t *d_applied) {
    REQUIRE(n_exchanges >= 1 && n_exchanges <= kLbMaxExchanges, "n_exchanges must be in [1, %d]", kLbMaxExchanges);
    REQUIRE(rule == 0 || rule == 1, "rule must be 0 (sqrt) or 1 (align)");
    REQUIRE(d_base && d_buckets && d_keeps && d_applied, "NULL device pointer");  // (arrive / landed: the launch checks its own)
    const int64_t nt = h->total_items, width = nt * h->k + nt;
    REQUIRE(bucket_stride >= width + 2 * nt && keep_stride >= width,
            "a bucket holds [dV | dB | wV | wB] = %lld floats, a keep buffer [dV | dB] = %lld", (long long)(width + 2 * nt),
            (long long)width);
    LdsBinExchange ex;
    ex.base = d_base; ex.bucket = d_buckets; ex.keep = d_keeps;
    ex.arrive = d_arrive; ex.landed = d_landed; ex.applied = d_applied;
    ex.bucket_stride = bucket_stride; ex.keep_stride = keep_stride; ex.nt = nt;
    ex.n_ex = n_exchanges; ex.rule = rule;
    return ex;
}

int cornac_hip_bpr_resident_exchange_bins(cornac_hip_bpr_t h, int neg_population, int hogwild_flags, int *n_bins) {
    return guarded([&] {
        bpr_check_keep_packed(h);
        REQUIRE(n_bins != nullptr, "n_bins is NULL");
        *n_bins = (h->hog_seeded && !h->f64 && hogwild_uses_ldsbin(h, h->nnz, neg_population, hogwild_flags) && !ldsbin_plan(h).passing)
                      ? ldsbin_plan_bins(h) : 0;
    });
}

int cornac_hip_bpr_epoch_resident_enqueue(cornac_hip_bpr_t h, float lr, float reg, int use_bias, int neg_population,
                                          int hogwild_flags, int n_exchanges, int rule, float *d_base, float *d_buckets,
                                          int64_t bucket_stride, float *d_keeps, int64_t keep_stride, uint32_t *d_arrive,
                                          const uint32_t *d_landed, uint32_t *d_applied, int *n_arrivals) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(h->hog_seeded, "hogwild mode needs cornac_hip_bpr_seed_hogwild first");
        REQUIRE(!h->f64, "the handle holds float64 tables");
        REQUIRE(neg_population == CORNAC_HIP_NEG_UNIFORM || neg_population == CORNAC_HIP_NEG_POPULARITY,
                "unknown neg_population %d", neg_population);
        REQUIRE(hogwild_uses_ldsbin(h, h->nnz, neg_population, hogwild_flags),
                "the resident exchange lives in the LDS-bin form, which this shape / these flags do not take "
                "(cornac_hip_bpr_resident_exchange_bins tells): use cornac_hip_bpr_hogwild_enqueue chunks");
        REQUIRE(!ldsbin_plan(h).passing, "the resident exchange needs resident bins (the item table in <= %d rounds); this table "
                "passes through the LDS in many rounds: exchange between whole-epoch launches", h->lb_max_rounds);
        REQUIRE(h->hog_offset == 0, "an epoch is under way (%lld of its samples drawn): the resident launch covers whole epochs",
                (long long)h->hog_offset);
        REQUIRE(d_arrive && d_landed, "NULL device pointer");
        const LdsBinExchange ex = resident_exchange_args(h, n_exchanges, rule, d_base, d_buckets, bucket_stride, d_keeps,
                                                         keep_stride, d_arrive, d_landed, d_applied);
        ldsbin_resident_enqueue(h, lr, reg, use_bias, neg_population, hogwild_flags, ex);
        if (n_arrivals) *n_arrivals = h->lb_bins;
    });
}

// ---- conveyor layout (multi-GPU regime 2: the item table sharded by row, its blocks rotating over the ranks; cornac_amd/dist.py
// BinConveyorBprTrainer).  The reference has no counterpart (one process); what a step computes is
// recom_bpr.pyx:231-267 for the draws of the block's bins.
int cornac_hip_bpr_conveyor_setup(cornac_hip_bpr_t h, int n_blocks, const int32_t *rank_item, uint64_t deal_seed,
                                  int release_item_tables, int *n_bins, int *bins_per_block, int *cap) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(n_blocks >= 1 && n_blocks <= (1 << 16), "n_blocks out of range");
        REQUIRE(!h->f64, "the handle holds float64 tables");
        HIP_CHECK(hipStreamSynchronize(h->stream));
        if (rank_item) {
            std::vector<uint8_t> seen((size_t)h->n_items, 0);
            for (int64_t r = 0; r < h->n_items; ++r) {
                REQUIRE(rank_item[r] >= 0 && rank_item[r] < h->n_items && !seen[(size_t)rank_item[r]],
                        "rank_item is not a permutation of the train items (entry %lld)", (long long)r);
                seen[(size_t)rank_item[r]] = 1;
            }
            h->cv_rank_item.ensure((size_t)h->n_items);
            h->cv_rank_item.upload(rank_item, (size_t)h->n_items, h->stream);
            HIP_CHECK(hipStreamSynchronize(h->stream));
        }
        h->cv_own_order = rank_item != nullptr;
        h->cv_blocks = n_blocks;
        h->cv_deal_seed = deal_seed;
        const LbPlan pl = ldsbin_plan(h);
        if (pl.bins == 0) {
            h->cv_blocks = 0;
            fail(CORNAC_HIP_ERR_INVALID, "no conveyor layout for %lld items in %d blocks at k = %d (a bin needs >= %d rows and its "
                 "rows must fit %d KB of LDS)", (long long)h->n_items, n_blocks, h->k, h->lb_min_candidates, h->lb_pass_kb);
        }
        h->lb_built = false;
        ldsbin_build(h);
        HIP_CHECK(hipMemsetAsync(h->lb_hot_off.p, 0, ((size_t)pl.bins + 1) * sizeof(uint32_t), h->stream));
        if (release_item_tables) {  // the rows live in the caller's block buffers: the handle's own item tables are not needed
            h->V.bind(nullptr, 0);
            h->B.bind(nullptr, 0);
        }
        HIP_CHECK(hipStreamSynchronize(h->stream));
        if (n_bins) *n_bins = pl.bins;
        if (bins_per_block) *bins_per_block = pl.bins / n_blocks;
        if (cap) *cap = pl.cap;
    });
}

int cornac_hip_bpr_conveyor_layout(cornac_hip_bpr_t h, uint32_t layout_epoch, int32_t *d_slot_item, int32_t *d_item_slot) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(h->cv_blocks && h->lb_built, "cornac_hip_bpr_conveyor_setup first");
        REQUIRE(d_slot_item || d_item_slot, "NULL device pointers");
        const int64_t total = (int64_t)h->lb_bins * h->lb_cap;
        const unsigned grid = (unsigned)std::min<int64_t>(2048, (total + kLbBlock - 1) / kLbBlock);
        hipLaunchKernelGGL(ldsbin_layout_kernel, dim3(grid), dim3(kLbBlock), 0, h->stream, ldsbin_rank_item(h),
                           ldsbin_key(h->cv_deal_seed, layout_epoch), (int32_t)h->n_items, h->lb_bins, h->lb_cap, h->lb_n_strata,
                           d_slot_item, d_item_slot);
        HIP_CHECK(hipGetLastError());
    });
}

int cornac_hip_bpr_conveyor_enqueue(cornac_hip_bpr_t h, uint32_t epoch, uint32_t layout_epoch, int n_ranges,
                                    const int32_t *first_block, float *const *d_rows, float lr, float reg, int use_bias,
                                    int neg_population, int hogwild_flags) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(h->cv_blocks && h->lb_built, "cornac_hip_bpr_conveyor_setup first");
        REQUIRE(h->hog_seeded, "hogwild mode needs cornac_hip_bpr_seed_hogwild first");
        REQUIRE(n_ranges >= 1 && n_ranges <= kLbMaxRanges, "n_ranges must be in [1, %d]", kLbMaxRanges);
        REQUIRE(first_block && d_rows, "NULL argument");
        REQUIRE(neg_population == CORNAC_HIP_NEG_UNIFORM || neg_population == CORNAC_HIP_NEG_POPULARITY,
                "unknown neg_population %d", neg_population);
        const int bpb = h->lb_bins / h->cv_blocks;
        LdsBinKernel kern = pick_ldsbin_kernel(h->k, neg_population == CORNAC_HIP_NEG_POPULARITY, false, true, true);
        HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lb_lds_bytes));
        LdsBinArgs a;
        ldsbin_fill_args(h, a, lr, reg, use_bias, neg_population, hogwild_flags);
        a.V = nullptr; a.B = nullptr;   // every row of these bins is in a block buffer
        a.epoch = epoch;
        a.key = ldsbin_key(h->cv_deal_seed, layout_epoch);
        a.s_begin = 0;
        a.n = (uint64_t)h->nnz;
        a.nnz = (uint64_t)h->nnz;
        a.conv_bpb = bpb;
        for (int r = 0; r < n_ranges; ++r) {
            REQUIRE(first_block[r] >= 0 && first_block[r] < h->cv_blocks && d_rows[r], "range %d: block %d / NULL buffer", r, first_block[r]);
            for (int q = 0; q < r; ++q) REQUIRE(first_block[q] != first_block[r], "block %d twice in one launch", first_block[r]);
            a.conv_first[r] = first_block[r] * bpb;
            a.conv_rows[r] = d_rows[r];
        }
        h->ktimer.before(h->stream);
        hipLaunchKernelGGL(kern, dim3((unsigned)(bpb * n_ranges)), dim3(h->lb_block), h->lb_lds_bytes, h->stream, a);
        h->ktimer.after(h->stream);
        HIP_CHECK(hipGetLastError());
    });
}

int cornac_hip_bpr_resident_flush(cornac_hip_bpr_t h, int n_exchanges, int rule, float *d_base, const float *d_buckets,
                                  int64_t bucket_stride, const float *d_keeps, int64_t keep_stride,
                                  const uint32_t *d_applied, const uint32_t *d_landed) {
    return guarded([&] {
        bpr_check(h);
        // d_landed (may be NULL = all landed): an exchange whose flag is still 0 — the communication stream gave up waiting
        // for its arrivals — is left unapplied
        const LdsBinExchange ex = resident_exchange_args(h, n_exchanges, rule, d_base, const_cast<float *>(d_buckets), bucket_stride,
                                                         const_cast<float *>(d_keeps), keep_stride, nullptr, d_landed,
                                                         const_cast<uint32_t *>(d_applied));
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((h->n_items + kWavesPerBlock - 1) / kWavesPerBlock, 4096));
        hipLaunchKernelGGL(ldsbin_exchange_flush_kernel, dim3(grid), dim3(kBlock), 0, h->stream, ex, h->V.p, h->B.p,
                           (int32_t)h->n_items, h->k);
        HIP_CHECK(hipGetLastError());
    });
}

// The communication stream's two steps of the hand-shake (no handle: they run on the caller's stream): wait until
// *d_counter >= target (gives up after timeout_ms and sets *d_error = 1), and set a flag.
int cornac_hip_stream_wait_counter(int device, void *hip_stream, const uint32_t *d_counter, uint32_t target,
                                   uint32_t *d_error, int timeout_ms) {
    return guarded([&] {
        REQUIRE(d_counter && d_error, "NULL device pointer");
        REQUIRE(timeout_ms > 0, "timeout_ms must be positive");
        use_device(device);
        hipLaunchKernelGGL(stream_wait_counter_kernel, dim3(1), dim3(kWave), 0, (hipStream_t)hip_stream, d_counter, target, d_error,
                           (long long)timeout_ms * 100000ll);
        HIP_CHECK(hipGetLastError());
    });
}

int cornac_hip_stream_set_flag(int device, void *hip_stream, uint32_t *d_flag, uint32_t value, const uint32_t *d_unless) {
    return guarded([&] {
        REQUIRE(d_flag != nullptr, "NULL device pointer");
        use_device(device);
        hipLaunchKernelGGL(stream_set_flag_kernel, dim3(1), dim3(kWave), 0, (hipStream_t)hip_stream, d_flag, value, d_unless);
        HIP_CHECK(hipGetLastError());
    });
}

int cornac_hip_stream_ring_standin(int device, void *hip_stream, const float *d_src, int64_t src_floats, float *d_dst,
                                   int64_t dst_floats, int64_t n_floats, int n_workgroups) {
    return guarded([&] {
        REQUIRE(d_src && d_dst && src_floats > 0 && dst_floats > 0, "NULL / empty buffer");
        REQUIRE(n_floats >= 0 && n_workgroups >= 1 && n_workgroups <= 1024, "n_workgroups must be in [1, 1024]");
        REQUIRE(src_floats >= 4 && dst_floats >= 4 && ((uintptr_t)d_src & 15) == 0 && ((uintptr_t)d_dst & 15) == 0,
                "buffers must be 16-byte aligned and hold at least 4 floats");
        use_device(device);
        if (n_floats == 0) return;
        hipLaunchKernelGGL(stream_ring_standin_kernel, dim3((unsigned)n_workgroups), dim3(512), 0, (hipStream_t)hip_stream, d_src,
                           (long long)src_floats, d_dst, (long long)dst_floats, (long long)n_floats);
        HIP_CHECK(hipGetLastError());
    });
}

int cornac_hip_bpr_chunk_records(cornac_hip_bpr_t h, int enable) {
    return guarded([&] {
        bpr_check(h);  // (dense tables current: the mode starts and ends with them)
        h->chunk_records = enable != 0;
    });
}

int cornac_hip_bpr_sync(cornac_hip_bpr_t h, int64_t *correct, int64_t *skipped) {
    return guarded([&] {
        bpr_check_keep_packed(h);  // (touches no table: packed strata records stay ...
        if (!h->V.owned || !h->B.owned) strata_unpack(h);  // ... unless the dense table is the caller's: it reads it after this call)
        if (correct) *correct = 0;
        if (skipped) *skipped = 0;
        fetch_counters(h, correct, skipped);
    });
}

int cornac_hip_bpr_debug_draw(cornac_hip_bpr_t h, int stream, uint64_t hi, int64_t n, int64_t *out) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(h->mt_seeded, "seed the mt19937 streams first");
        REQUIRE(stream == 0 || stream == 1, "stream must be 0 or 1");
        REQUIRE(n >= 0 && out != nullptr, "bad output");
        if (n == 0) return;
        DevBuf<uint32_t> d;
        d.alloc((size_t)n);
        uint32_t *o = d.p;
        mt_draw(h, 1, &stream, &hi, &n, &o);
        std::vector<uint32_t> tmp((size_t)n);
        d.download(tmp.data(), (size_t)n, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        for (int64_t t = 0; t < n; ++t) out[t] = (int64_t)tmp[(size_t)t];
    });
}

int cornac_hip_bpr_debug_ownership(cornac_hip_bpr_t h, int64_t *n_waves, int64_t *wave_ptr, int32_t *own_u,
                                   int32_t *own_i) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(n_waves != nullptr, "n_waves is NULL");
        *n_waves = 0;
        if (!hogwild_uses_ownership(h, 0)) return;
        // the tables of the default throughput path (flags == 0): the strata kernel's grid where that form applies,
        // else the fused kernel's; tables already built for a launch are returned as they are
        if (h->own_waves == 0) {
            if (!hogwild_uses_ldsbin(h, h->nnz, CORNAC_HIP_NEG_UNIFORM, 0) &&
                hogwild_uses_strata(h, h->nnz, CORNAC_HIP_NEG_UNIFORM, 0)) {
                strata_prepare(h);
            } else {
                HogKernel kern = pick_hogwild_kernel<true>(h->k, 0);
                int per_cu = 0;
                HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kBlock, 0));
                build_ownership(h, (int64_t)device_info(h->device).cus * std::max(1, std::min(per_cu, 8)) * kWavesPerBlock);
            }
        }
        const int64_t W = h->own_waves;
        *n_waves = W;
        if (wave_ptr) std::copy(h->h_wave_ptr.begin(), h->h_wave_ptr.end(), wave_ptr);
        if (own_u) std::copy(h->h_own_u.begin(), h->h_own_u.end(), own_u);
        if (own_i) std::copy(h->h_own_i.begin(), h->h_own_i.end(), own_i);
    });
}

int cornac_hip_bpr_kernel_timing(cornac_hip_bpr_t h, int enable, double *total_ms, int64_t *launches) {
    return guarded([&] {
        bpr_check_keep_packed(h);  // (touches no table: packed strata records stay)
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->ktimer.collect(total_ms, launches);
        h->ktimer.enabled = enable != 0;
    });
}

int cornac_hip_bpr_strata_config(cornac_hip_bpr_t h, int hot_permille, int hot_min_mult_x100, int rehash_period) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(hot_permille >= 0 && hot_permille <= 1000, "hot_permille must be in [0, 1000]");
        REQUIRE(hot_min_mult_x100 >= 0, "hot_min_mult_x100 must be >= 0");
        REQUIRE(rehash_period >= 1, "rehash_period must be >= 1");
        h->strata_hot_permille = hot_permille;
        h->strata_hot_min_mult_x100 = hot_min_mult_x100;
        h->strata_rehash_period = rehash_period;
        h->strata_ranked = false;  // the hot set is re-derived (and the buckets re-dealt) at the next launch
        h->strata_built = false;
    });
}

int cornac_hip_bpr_strata_stats(cornac_hip_bpr_t h, int64_t *out4) {
    return guarded([&] {
        bpr_check_keep_packed(h);  // (touches no table: packed strata records stay)
        REQUIRE(out4 != nullptr, "out4 is NULL");
        HIP_CHECK(hipStreamSynchronize(h->stream));
        out4[0] = h->strata_ranked ? (int64_t)h->strata_n_hot : -1;
        out4[1] = h->strata_misplaced;
        out4[2] = h->strata_builds;
        out4[3] = h->strata_kernel ? h->own_waves : 0;
    });
}

int cornac_hip_bpr_debug_strata(cornac_hip_bpr_t h, uint32_t epoch, int64_t *sptr, int32_t *rec_u, int32_t *rec_i,
                                int32_t *rank_item, uint32_t *key) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(h->hog_seeded, "seed the hogwild sampler first");
        REQUIRE(hogwild_uses_strata(h, h->nnz, CORNAC_HIP_NEG_UNIFORM, 2 << 16), "this shape cannot run the strata form");
        const int grid = strata_prepare(h);
        const uint32_t kk = strata_key(h->hog_seed, epoch / (uint32_t)std::max(1, h->strata_rehash_period));
        strata_build_buckets(h, grid, kk);
        const size_t W = (size_t)grid * kWavesPerBlock;
        if (sptr) h->sptr.download(sptr, W * 8 + 1, h->stream);
        if (rec_u) h->rec_u.download(rec_u, (size_t)h->nnz, h->stream);
        if (rec_i) h->rec_i.download(rec_i, (size_t)h->nnz, h->stream);
        if (rank_item) std::copy(h->h_rank_item.begin(), h->h_rank_item.end(), rank_item);
        if (key) *key = kk;
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}

int cornac_hip_bpr_ldsbin_config(cornac_hip_bpr_t h, int hot_x1000, int min_candidates, int max_rounds) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(hot_x1000 >= 1, "hot_x1000 must be >= 1");
        REQUIRE(min_candidates >= 1, "min_candidates must be >= 1");
        REQUIRE(max_rounds >= 1 && max_rounds <= 1024, "max_rounds must be in [1, 1024]");
        h->lb_hot_x1000 = hot_x1000;
        h->lb_min_candidates = min_candidates;
        h->lb_max_rounds = max_rounds;
        h->lb_built = false;
    });
}

int cornac_hip_bpr_ldsbin_pass_config(cornac_hip_bpr_t h, int enable, int waves, int lds_kb, int min_draws_x100) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(waves == 4 || waves == 8 || waves == 16, "waves per workgroup must be 4, 8 or 16");
        REQUIRE(lds_kb >= 16 && lds_kb <= 156, "lds_kb must be in [16, 156]");
        REQUIRE(min_draws_x100 >= 0, "min_draws_x100 must be >= 0");
        h->lb_pass_enable = enable != 0;
        h->lb_pass_waves = waves;
        h->lb_pass_kb = lds_kb;
        h->lb_pass_min_draws_x100 = min_draws_x100;
        h->lb_built = false;
    });
}

int cornac_hip_bpr_ldsbin_deal_config(cornac_hip_bpr_t h, int strata_groups, int hot_cost_x16) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(strata_groups >= 1, "strata_groups must be >= 1");
        REQUIRE(hot_cost_x16 >= 0 && hot_cost_x16 <= 4096, "hot_cost_x16 must be in [0, 4096]");
        h->lb_strata_groups = strata_groups;
        h->lb_hot_cost_x16 = hot_cost_x16;
        h->lb_built = false;
    });
}

int cornac_hip_bpr_debug_ldsbin_deal(cornac_hip_bpr_t h, uint64_t seed, uint32_t epoch, int32_t *bin_of_item,
                                     uint32_t *cold_mass, uint32_t *hot_off, int32_t *hot_u, int32_t *hot_i) {
    return guarded([&] {
        bpr_check(h);
        const int bins = ldsbin_plan_bins(h);
        REQUIRE(bins > 0, "this shape does not use the LDS-bin form");
        ldsbin_build(h);
        const uint32_t key = ldsbin_key(seed, epoch);
        h->lb_deal_valid = false;
        ldsbin_deal(h, seed, epoch, key);
        h->lb_deal_valid = false;  // a test hook: the next epoch deals again
        if (cold_mass) h->lb_cold.download(cold_mass, (size_t)bins, h->stream);
        if (hot_off) h->lb_hot_off.download(hot_off, (size_t)bins + 1, h->stream);
        if (hot_u && h->lb_n_hot_inter) h->lb_hot_u.download(hot_u, (size_t)h->lb_n_hot_inter, h->stream);
        if (hot_i && h->lb_n_hot_inter) h->lb_hot_i.download(hot_i, (size_t)h->lb_n_hot_inter, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        if (bin_of_item) {  // the host's evaluation of the same deal functions the kernels use
            const uint32_t nb = (uint32_t)bins, ni = (uint32_t)h->n_items, n_groups = (ni + nb - 1) / nb;
            for (uint32_t p = 0; p < ni; ++p) {
                const uint32_t g = p / nb, o = p - g * nb;
                const uint32_t code = ldsbin_deal_rank(p, key, nb, ni, n_groups, (uint32_t)h->lb_n_strata);
                bin_of_item[(size_t)h->h_rank_item[code]] = (int32_t)((o + ldsbin_rot(g, key, nb)) % nb);
            }
        }
    });
}

int cornac_hip_bpr_ldsbin_stats(cornac_hip_bpr_t h, int64_t *out6) {
    return guarded([&] {
        bpr_check(h);
        REQUIRE(out6 != nullptr, "out8 is NULL");
        out6[6] = h->lb_lock_timeouts;
        const int bins = ldsbin_plan_bins(h);
        if (bins > 0) ldsbin_build(h);
        out6[0] = bins;
        out6[1] = bins ? h->lb_cap : 0;
        out6[2] = bins ? h->lb_n_hot : 0;
        out6[3] = bins ? h->lb_n_hot_inter : 0;
        out6[4] = bins ? h->lb_bm_words : 0;
        out6[5] = bins ? (int64_t)h->lb_lds_bytes : 0;
        out6[7] = bins ? h->lb_block : 0;
    });
}

int cornac_hip_bpr_last_timing(cornac_hip_bpr_t h, double *ms4) {
    return guarded([&] {
        REQUIRE(h && ms4, "NULL argument");
        for (int t = 0; t < 4; ++t) ms4[t] = h->timing[t];
    });
}
}

#include "vebpr.inc"
#include "sharded.inc"
