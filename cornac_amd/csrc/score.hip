// Scoring and batched ranking on MI355X (gfx950).
//
// Replaces fast_dot (cornac/utils/fast_dot.pyx:40-43) as called by BPR.score / MF.score and the
// per-user argsort / argpartition of Recommender.rank (cornac/models/recommender.py:503-530):
//
//   score(u, i) = (item_base[i] + user_base[u]) + fma-chain_{f=0..k-1}(U[u,f] * V[i,f])
//
//  * score_user / score_block : one lane per item, explicit fmaf chain in index order.
//  * score_gemm_mfma          : the batched users x items scoring GEMM on the fp32 matrix cores
//                               (v_mfma_f32_32x32x2_f32: exact fp32, and — because an MFMA is a
//                               k-ordered fma chain — bit-identical to the fmaf chain above).
//  * topk_select / full_sort  : per-user ranking.  Every (score, item) pair is mapped to a unique
//                               64-bit key  [order-preserving score bits | item index], so
//                               "descending score, ties by higher item index" (the oracle's pinned
//                               tie rule) is a plain descending sort of integers.
#include <algorithm>
#include <cmath>

#include "common.h"

namespace chip {

constexpr int kBlk = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t order_key(float s) {
    const uint32_t b = __float_as_uint(s);
    const uint32_t k = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return k ? k : 1u;
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

// ---- VALU scoring: out[b * n_items + i] for users[b] (users == nullptr: user = u0 + b) ------------
__global__ __launch_bounds__(kBlk) void score_valu_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                          const float *__restrict__ item_base,
                                                          const float *__restrict__ user_base,
                                                          const int32_t *__restrict__ users, int64_t u0,
                                                          int64_t n_items, int k, float *__restrict__ out) {
    extern __shared__ float urow[];
    const int64_t b = blockIdx.y;
    const int64_t u = users ? (int64_t)users[b] : u0 + b;
    for (int f = threadIdx.x; f < k; f += kBlk) urow[f] = U[u * k + f];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x;
    if (i >= n_items) return;
    const float *row = V + i * k;
    float acc = 0.f;
    for (int f = 0; f < k; ++f) acc = fmaf(urow[f], row[f], acc);
    const float ub = user_base ? user_base[u] : 0.f;
    const float ib = item_base ? item_base[i] : 0.f;
    out[b * n_items + i] = (ib + ub) + acc;
}

// ---- MFMA scoring GEMM ---------------------------------------------------------------------------
// One wave owns MT stacked 32-user tiles and walks a strip of 32-item tiles.  Operand layout of
// v_mfma_f32_32x32x2_f32: lane l supplies A[row = l & 31][kk = l >> 5] and B[kk = l >> 5][col = l & 31];
// step t of the chain covers factors (2t, 2t+1), so the accumulation is the index-ordered fma chain.
// Accumulator register r of lane l is C[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31].
template <int KT, int MT>
__global__ __launch_bounds__(kBlk) void score_gemm_mfma_kernel(const float *__restrict__ U,
                                                               const float *__restrict__ V,
                                                               const float *__restrict__ item_base,
                                                               const float *__restrict__ user_base,
                                                               const int32_t *__restrict__ users, int64_t u0,
                                                               int64_t n_rows, int64_t n_items, int k,
                                                               int tiles_per_strip, float *__restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 31, half = lane >> 5;
    const int64_t row_tile0 = ((int64_t)blockIdx.y * (kBlk / 64) + wave) * MT;  // first 32-row tile of this wave
    if (row_tile0 * 32 >= n_rows) return;
    float a[MT][KT];
    float ubias[MT];
    int64_t urow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int64_t r = (row_tile0 + m) * 32 + col;  // A row handled by this lane
        const bool ok = r < n_rows;
        const int64_t u = ok ? (users ? (int64_t)users[r] : u0 + r) : 0;
        urow[m] = u;
        const float *p = U + u * k;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const int f = 2 * t + half;
            a[m][t] = (ok && f < k) ? p[f] : 0.f;
        }
        ubias[m] = 0.f;
    }
    const int64_t n_item_tiles = (n_items + 31) / 32;
    const int64_t t_begin = (int64_t)blockIdx.x * tiles_per_strip;
    const int64_t t_end = min(n_item_tiles, t_begin + tiles_per_strip);
    for (int64_t it = t_begin; it < t_end; ++it) {
        const int64_t item = it * 32 + col;
        const bool iok = item < n_items;
        const float *q = V + (iok ? item : 0) * k;
        float bfrag[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const int f = 2 * t + half;
            bfrag[t] = (iok && f < k) ? q[f] : 0.f;
        }
        const float ib = (iok && item_base) ? item_base[item] : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < KT; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][t], bfrag[t], acc, 0, 0, 0);
            if (iok) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = (row_tile0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < n_rows) {
                        float ub = 0.f;
                        if (user_base) ub = user_base[users ? (int64_t)users[row] : u0 + row];
                        out[row * n_items + item] = (ib + ub) + acc[r];
                    }
                }
            }
        }
    }
    (void)ubias;
    (void)urow;
}

// ---- exclusion: out[row, item] = NaN-tagged "excluded" marker (handled as key 0) ----------------------
__global__ __launch_bounds__(kBlk) void mark_excluded_kernel(const int64_t *__restrict__ excl_indptr,
                                                             const int32_t *__restrict__ excl_indices, int64_t row0,
                                                             int64_t n_items, uint8_t *__restrict__ excl_mask) {
    const int64_t row = blockIdx.x;
    const int64_t lo = excl_indptr[row0 + row], hi = excl_indptr[row0 + row + 1];
    for (int64_t p = lo + threadIdx.x; p < hi; p += kBlk) {
        const int32_t it = excl_indices[p];
        if (it >= 0 && it < n_items) excl_mask[row * n_items + it] = 1;
    }
}

__device__ __forceinline__ unsigned long long composite_key(const float *__restrict__ scores,
                                                            const uint8_t *__restrict__ excl, int64_t i) {
    if (excl && excl[i]) return 0ull;
    return ((unsigned long long)order_key(scores[i]) << 32) | (unsigned long long)(uint32_t)i;
}

// ---- top-k by 8-pass MSB radix select on the 64-bit key + bitonic sort of the survivors -----------------
// one workgroup per row; topk <= TOPK_MAX
constexpr int TOPK_MAX = 2048;

__global__ __launch_bounds__(kBlk) void topk_select_kernel(const float *__restrict__ scores,
                                                           const uint8_t *__restrict__ excl_mask, int64_t n_items,
                                                           int topk, int topk_pad, int32_t *__restrict__ items_out,
                                                           float *__restrict__ scores_out) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long sh_prefix;
    __shared__ unsigned int sh_remaining, sh_count;
    extern __shared__ unsigned long long list[];  // topk_pad entries
    const int64_t row = blockIdx.x;
    const float *srow = scores + row * n_items;
    const uint8_t *erow = excl_mask ? excl_mask + row * n_items : nullptr;
    const int tid = threadIdx.x;
    if (tid == 0) {
        sh_prefix = 0ull;
        sh_remaining = (unsigned)topk;
        sh_count = 0;
    }
    __syncthreads();
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        const unsigned long long prefix = sh_prefix;
        for (int64_t i = tid; i < n_items; i += kBlk) {
            const unsigned long long key = composite_key(srow, erow, i);
            const bool match = pass == 0 ? true : ((key >> (shift + 8)) == prefix);
            if (match) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned int rem = sh_remaining, acc = 0;
            int b = 255;
            for (; b > 0; --b) {
                if (acc + hist[b] >= rem) break;
                acc += hist[b];
            }
            sh_remaining = rem - acc;  // rank of the wanted key inside bucket b
            sh_prefix = (prefix << 8) | (unsigned long long)b;
        }
        __syncthreads();
    }
    const unsigned long long thresh = sh_prefix;  // the topk-th largest key (0 if fewer candidates)
    for (int i = tid; i < topk_pad; i += kBlk) list[i] = 0ull;
    __syncthreads();
    for (int64_t i = tid; i < n_items; i += kBlk) {
        const unsigned long long key = composite_key(srow, erow, i);
        if (key >= thresh && key != 0ull) {
            const unsigned int pos = atomicAdd(&sh_count, 1u);
            if (pos < (unsigned)topk_pad) list[pos] = key;
        }
    }
    __syncthreads();
    // bitonic sort, descending
    for (int kk = 2; kk <= topk_pad; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < topk_pad; i += kBlk) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = list[i], y = list[ixj];
                    const bool desc = (i & kk) == 0;
                    if (desc ? (x < y) : (x > y)) {
                        list[i] = y;
                        list[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < topk; i += kBlk) {
        const unsigned long long key = list[i];
        items_out[row * topk + i] = key ? (int32_t)(uint32_t)key : -1;
        scores_out[row * topk + i] = key ? key_to_float((uint32_t)(key >> 32)) : -INFINITY;
    }
}

// ---- full ranking: bitonic sort of all keys of a row in a global scratch (one workgroup per row) ---------
constexpr int kSortBlk = 1024;
__global__ __launch_bounds__(kSortBlk) void full_sort_kernel(const float *__restrict__ scores,
                                                             const uint8_t *__restrict__ excl_mask, int64_t n_items,
                                                             int64_t n_pad, unsigned long long *__restrict__ scratch,
                                                             int topk, int32_t *__restrict__ items_out,
                                                             float *__restrict__ scores_out) {
    const int64_t row = blockIdx.x;
    const float *srow = scores + row * n_items;
    const uint8_t *erow = excl_mask ? excl_mask + row * n_items : nullptr;
    unsigned long long *keys = scratch + row * n_pad;
    const int tid = threadIdx.x;
    for (int64_t i = tid; i < n_pad; i += kSortBlk) keys[i] = i < n_items ? composite_key(srow, erow, i) : 0ull;
    __syncthreads();
    for (int64_t kk = 2; kk <= n_pad; kk <<= 1) {
        for (int64_t j = kk >> 1; j > 0; j >>= 1) {
            for (int64_t i = tid; i < n_pad; i += kSortBlk) {
                const int64_t ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = keys[i], y = keys[ixj];
                    const bool desc = (i & kk) == 0;
                    if (desc ? (x < y) : (x > y)) {
                        keys[i] = y;
                        keys[ixj] = x;
                    }
                }
            }
            __syncthreads();  // workgroup-scope: one workgroup owns the row, L1 is shared by its waves
        }
    }
    for (int64_t i = tid; i < topk; i += kSortBlk) {
        const unsigned long long key = keys[i];
        items_out[row * topk + i] = key ? (int32_t)(uint32_t)key : -1;
        scores_out[row * topk + i] = key ? key_to_float((uint32_t)(key >> 32)) : -INFINITY;
    }
}

}  // namespace chip

using namespace chip;

struct cornac_hip_scorer {
    int device = 0;
    int64_t n_users = 0, n_items = 0;
    int k = 0;
    hipStream_t stream = nullptr;
    DevBuf<float> U, V, item_base, user_base;
    bool has_user_base = false, is_set = false;
    DevBuf<float> scores;  // workspace [rows_cap, n_items]
    DevBuf<uint8_t> excl;
    DevBuf<int32_t> d_users, d_items_out, d_excl_indices;
    DevBuf<int64_t> d_excl_indptr;
    DevBuf<float> d_scores_out;
    DevBuf<unsigned long long> sort_scratch;
};

static void sc_check(cornac_hip_scorer_t h, bool need_set = true) {
    REQUIRE(h != nullptr, "scorer handle is NULL");
    HIP_CHECK(hipSetDevice(h->device));
    if (need_set) REQUIRE(h->is_set, "cornac_hip_scorer_set has not been called");
}

static int64_t rows_per_batch(cornac_hip_scorer_t h) {
    // keep the score workspace <= 2 GiB
    const int64_t cap = (int64_t(2) << 30) / (h->n_items * (int64_t)sizeof(float));
    return std::max<int64_t>(32, std::min<int64_t>(cap / 32 * 32, 16384));
}

// scores for rows [0, n) of the current batch -> h->scores
static void launch_scores(cornac_hip_scorer_t h, const int32_t *d_users, int64_t u0, int64_t n, bool use_mfma) {
    const float *ub = h->has_user_base ? h->user_base.p : nullptr;
    const int k = h->k;
    if (use_mfma && k <= 128) {
        const int KT = (k + 1) / 2;
        const int64_t n_item_tiles = (h->n_items + 31) / 32;
        const DeviceInfo &di = device_info(h->device);
        auto go = [&](auto kernel, int MT) {
            const int64_t row_tiles = (n + 31) / 32;
            const int64_t wg_rows = (row_tiles + (int64_t)MT * 4 - 1) / ((int64_t)MT * 4);
            int64_t strips = std::max<int64_t>(1, ((int64_t)di.cus * 8 + wg_rows - 1) / wg_rows);
            strips = std::min(strips, n_item_tiles);
            const int tiles_per_strip = (int)((n_item_tiles + strips - 1) / strips);
            const int64_t gx = (n_item_tiles + tiles_per_strip - 1) / tiles_per_strip;
            hipLaunchKernelGGL(kernel, dim3((unsigned)gx, (unsigned)wg_rows), dim3(kBlk), 0, h->stream, h->U.p, h->V.p,
                               h->item_base.p, ub, d_users, u0, n, h->n_items, k, tiles_per_strip, h->scores.p);
        };
        if (KT <= 8) go(score_gemm_mfma_kernel<8, 2>, 2);
        else if (KT <= 16) go(score_gemm_mfma_kernel<16, 2>, 2);
        else if (KT <= 32) go(score_gemm_mfma_kernel<32, 2>, 2);
        else go(score_gemm_mfma_kernel<64, 1>, 1);
    } else {
        dim3 grid((unsigned)((h->n_items + kBlk - 1) / kBlk), (unsigned)n);
        hipLaunchKernelGGL(score_valu_kernel, grid, dim3(kBlk), (size_t)k * sizeof(float), h->stream, h->U.p, h->V.p,
                           h->item_base.p, ub, d_users, u0, h->n_items, k, h->scores.p);
    }
    HIP_CHECK(hipGetLastError());
}

static void launch_rank(cornac_hip_scorer_t h, int64_t n, int topk, bool have_excl, int32_t *items_out,
                        float *scores_out) {
    const uint8_t *excl = have_excl ? h->excl.p : nullptr;
    if (topk <= TOPK_MAX) {
        int pad = 1;
        while (pad < topk) pad <<= 1;
        hipLaunchKernelGGL(topk_select_kernel, dim3((unsigned)n), dim3(kBlk), (size_t)pad * 8, h->stream, h->scores.p,
                           excl, h->n_items, topk, pad, items_out, scores_out);
    } else {
        int64_t pad = 1;
        while (pad < h->n_items) pad <<= 1;
        h->sort_scratch.ensure((size_t)(n * pad));
        hipLaunchKernelGGL(full_sort_kernel, dim3((unsigned)n), dim3(kSortBlk), 0, h->stream, h->scores.p, excl,
                           h->n_items, pad, h->sort_scratch.p, topk, items_out, scores_out);
    }
    HIP_CHECK(hipGetLastError());
}

extern "C" {

int cornac_hip_scorer_create(cornac_hip_scorer_t *out, int device, int64_t n_users, int64_t n_items, int k) {
    return guarded([&] {
        REQUIRE(out != nullptr, "out handle pointer is NULL");
        *out = nullptr;
        REQUIRE(n_users > 0 && n_items > 0 && k > 0, "n_users, n_items and k must be positive");
        REQUIRE(n_items < (int64_t(1) << 31), "n_items exceeds int32");
        use_device(device);
        std::unique_ptr<cornac_hip_scorer> h(new cornac_hip_scorer());
        h->device = device; h->n_users = n_users; h->n_items = n_items; h->k = k;
        HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->U.alloc((size_t)n_users * k);
        h->V.alloc((size_t)n_items * k);
        h->item_base.alloc((size_t)n_items);
        h->user_base.alloc((size_t)n_users);
        *out = h.release();
    });
}

int cornac_hip_scorer_destroy(cornac_hip_scorer_t h) {
    return guarded([&] {
        if (!h) return;
        (void)hipSetDevice(h->device);
        if (h->stream) {
            (void)hipStreamSynchronize(h->stream);
            (void)hipStreamDestroy(h->stream);
        }
        delete h;
    });
}

int cornac_hip_scorer_set(cornac_hip_scorer_t h, const float *U, const float *V, const float *item_base,
                          const float *user_base) {
    return guarded([&] {
        sc_check(h, false);
        REQUIRE(U && V, "U and V are required");
        h->U.upload(U, (size_t)h->n_users * h->k, h->stream);
        h->V.upload(V, (size_t)h->n_items * h->k, h->stream);
        if (item_base) h->item_base.upload(item_base, (size_t)h->n_items, h->stream);
        else HIP_CHECK(hipMemsetAsync(h->item_base.p, 0, (size_t)h->n_items * 4, h->stream));
        h->has_user_base = user_base != nullptr;
        if (user_base) h->user_base.upload(user_base, (size_t)h->n_users, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->is_set = true;
    });
}

int cornac_hip_score_user(cornac_hip_scorer_t h, int64_t user, float *out) {
    return guarded([&] {
        sc_check(h);
        REQUIRE(user >= 0 && user < h->n_users, "user %lld out of range", (long long)user);
        REQUIRE(out != nullptr, "out is NULL");
        h->scores.ensure((size_t)h->n_items);
        launch_scores(h, nullptr, user, 1, false);
        h->scores.download(out, (size_t)h->n_items, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}

int cornac_hip_score_block(cornac_hip_scorer_t h, const int32_t *users, int64_t n, float *out) {
    return guarded([&] {
        sc_check(h);
        REQUIRE(users && out && n > 0, "bad arguments");
        for (int64_t b = 0; b < n; ++b)
            REQUIRE(users[b] >= 0 && users[b] < h->n_users, "user %d out of range", users[b]);
        const int64_t cap = rows_per_batch(h);
        h->scores.ensure((size_t)(std::min(cap, n) * h->n_items));
        h->d_users.ensure((size_t)std::min(cap, n));
        for (int64_t b0 = 0; b0 < n; b0 += cap) {
            const int64_t nb = std::min(cap, n - b0);
            h->d_users.upload(users + b0, (size_t)nb, h->stream);
            launch_scores(h, h->d_users.p, 0, nb, true);
            h->scores.download(out + b0 * h->n_items, (size_t)(nb * h->n_items), h->stream);
            HIP_CHECK(hipStreamSynchronize(h->stream));
        }
    });
}

int cornac_hip_rank_topk(cornac_hip_scorer_t h, const int32_t *users, int64_t n, int topk, const int64_t *excl_indptr,
                         const int32_t *excl_indices, int32_t *items_out, float *scores_out) {
    return guarded([&] {
        sc_check(h);
        REQUIRE(users && items_out && scores_out && n > 0, "bad arguments");
        REQUIRE(topk >= 1 && topk <= h->n_items, "topk must be in [1, n_items]");
        for (int64_t b = 0; b < n; ++b)
            REQUIRE(users[b] >= 0 && users[b] < h->n_users, "user %d out of range", users[b]);
        const bool have_excl = excl_indptr != nullptr && excl_indices != nullptr;
        int64_t cap = rows_per_batch(h);
        if (topk > TOPK_MAX) cap = std::min<int64_t>(cap, 1024);
        const int64_t nb_max = std::min(cap, n);
        h->scores.ensure((size_t)(nb_max * h->n_items));
        h->d_users.ensure((size_t)nb_max);
        h->d_items_out.ensure((size_t)(nb_max * topk));
        h->d_scores_out.ensure((size_t)(nb_max * topk));
        if (have_excl) {
            h->excl.ensure((size_t)(nb_max * h->n_items));
            h->d_excl_indptr.ensure((size_t)n + 1);
            h->d_excl_indptr.upload(excl_indptr, (size_t)n + 1, h->stream);
            const int64_t ne = excl_indptr[n];
            h->d_excl_indices.ensure((size_t)std::max<int64_t>(ne, 1));
            if (ne > 0) h->d_excl_indices.upload(excl_indices, (size_t)ne, h->stream);
        }
        for (int64_t b0 = 0; b0 < n; b0 += cap) {
            const int64_t nb = std::min(cap, n - b0);
            h->d_users.upload(users + b0, (size_t)nb, h->stream);
            launch_scores(h, h->d_users.p, 0, nb, true);
            if (have_excl) {
                HIP_CHECK(hipMemsetAsync(h->excl.p, 0, (size_t)(nb * h->n_items), h->stream));
                hipLaunchKernelGGL(mark_excluded_kernel, dim3((unsigned)nb), dim3(kBlk), 0, h->stream,
                                   h->d_excl_indptr.p, h->d_excl_indices.p, b0, h->n_items, h->excl.p);
            }
            launch_rank(h, nb, topk, have_excl, h->d_items_out.p, h->d_scores_out.p);
            h->d_items_out.download(items_out + b0 * topk, (size_t)(nb * topk), h->stream);
            h->d_scores_out.download(scores_out + b0 * topk, (size_t)(nb * topk), h->stream);
            HIP_CHECK(hipStreamSynchronize(h->stream));
        }
    });
}

int cornac_hip_rank_topk_device(cornac_hip_scorer_t h, int64_t u0, int64_t n, int topk, int repeats, double *ms) {
    return guarded([&] {
        sc_check(h);
        REQUIRE(u0 >= 0 && n > 0 && u0 + n <= h->n_users, "user range out of bounds");
        REQUIRE(topk >= 1 && topk <= h->n_items && topk <= TOPK_MAX, "topk out of range for the device probe");
        REQUIRE(repeats >= 1 && ms, "bad arguments");
        const int64_t cap = rows_per_batch(h);
        const int64_t nb_max = std::min(cap, n);
        h->scores.ensure((size_t)(nb_max * h->n_items));
        h->d_items_out.ensure((size_t)(nb_max * topk));
        h->d_scores_out.ensure((size_t)(nb_max * topk));
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, h->stream));
        for (int r = 0; r < repeats; ++r) {
            for (int64_t b0 = 0; b0 < n; b0 += cap) {
                const int64_t nb = std::min(cap, n - b0);
                launch_scores(h, nullptr, u0 + b0, nb, true);
                launch_rank(h, nb, topk, false, h->d_items_out.p, h->d_scores_out.p);
            }
        }
        HIP_CHECK(hipEventRecord(e1, h->stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float t = 0.f;
        HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *ms = (double)t;
    });
}
}
