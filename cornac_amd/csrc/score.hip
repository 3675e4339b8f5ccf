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
//  * rank_fused               : scoring GEMM + top-k (topk <= 32) without materialising the scores.
//  * topk_select / full_sort  : per-user ranking of a materialised score tile (larger topk, full rankings,
//                               candidate lists).  Every (score, item) pair is mapped to a unique
//                               64-bit key  [order-preserving score bits | item index], so
//                               "descending score, ties by higher item index" (the oracle's pinned
//                               tie rule) is a plain descending sort of integers.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"

namespace chip {

constexpr int kBlk = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

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
                                                          int64_t n_items, int k, int ld, float *__restrict__ out) {
    extern __shared__ float urow[];
    const int64_t b = blockIdx.y;
    const int64_t u = users ? (int64_t)users[b] : u0 + b;
    for (int f = threadIdx.x; f < k; f += kBlk) urow[f] = U[u * ld + f];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x;
    if (i >= n_items) return;
    const float *row = V + i * ld;
    float acc = 0.f;
    for (int f = 0; f < k; ++f) acc = fmaf(urow[f], row[f], acc);
    const float ub = user_base ? user_base[u] : 0.f;
    const float ib = item_base ? item_base[i] : 0.f;
    out[b * n_items + i] = (ib + ub) + acc;
}

// ---- (user, item) pair scoring: the batched form of Recommender.rate()'s score(u, i) calls ----------------
// (cornac/eval_methods/base_method.py:35-105 calls rate() once per test rating).  One lane per pair,
// fmaf chain in index order; optional clipping to [lo, hi] (cornac/utils/common.py clip).
__global__ __launch_bounds__(kBlk) void score_pairs_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                           const float *__restrict__ item_base,
                                                           const float *__restrict__ user_base,
                                                           const int32_t *__restrict__ users,
                                                           const int32_t *__restrict__ items, int64_t n, int k, int ld,
                                                           int do_clip, float lo, float hi, float *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * kBlk + threadIdx.x;
    if (p >= n) return;
    const int64_t u = users[p], i = items[p];
    const float *pu = U + u * ld, *pv = V + i * ld;
    float acc = 0.f;
    for (int f = 0; f < k; ++f) acc = fmaf(pu[f], pv[f], acc);
    float s = ((item_base ? item_base[i] : 0.f) + (user_base ? user_base[u] : 0.f)) + acc;
    if (do_clip) s = fminf(fmaxf(s, lo), hi);
    out[p] = s;
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
        const float *p = U + u * (2 * KT) + half;  // tables are zero-padded to 2*KT columns
#pragma unroll
        for (int t = 0; t < KT; ++t) a[m][t] = p[2 * t];
        ubias[m] = 0.f;
    }
    const int64_t n_item_tiles = (n_items + 31) / 32;
    const int64_t t_begin = (int64_t)blockIdx.x * tiles_per_strip;
    const int64_t t_end = min(n_item_tiles, t_begin + tiles_per_strip);
    for (int64_t it = t_begin; it < t_end; ++it) {
        const int64_t item = it * 32 + col;
        const bool iok = item < n_items;
        const float *q = V + (iok ? item : 0) * (2 * KT) + half;
        float bfrag[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) bfrag[t] = q[2 * t];
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

// byte offset of a __shared__ object inside the workgroup's LDS allocation
template <typename T>
__device__ __forceinline__ uint32_t lds_offset(T *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) T *)p;
}

// ---- fused scoring GEMM + top-k: the users x items score tile never leaves the registers ------------------
// One wave owns a 32-user tile (A fragments in registers) and walks a range of 32-item tiles with
// v_mfma_f32_32x32x2_f32; the 4 waves of a workgroup share each B tile through LDS.
// Every accumulator value is compared with its row's running threshold (the score of the row's
// current topk-th candidate); the survivors are appended to a per-row candidate buffer in LDS (CAP
// slots; slot = register-resident count + ballot prefix, no LDS atomics).  When a buffer would overflow the
// wave compacts the row by rank counting (lane i counts the candidates that beat candidate i from LDS
// broadcast reads), keeps the topk best and raises the threshold.  Optional exclusion lists (sorted CSR per
// row) are consulted only at compaction.  Work is cut into equal contiguous ranges of (row block, item tile),
// one per resident workgroup; every piece ("segment") of a row block emits topk keys per row and
// rank_merge_kernel merges them.  See DESIGN.md section 4 for the measurements behind each choice.
template <int KT, int CAP, bool UB>
__global__ __launch_bounds__(kBlk, ((KT <= 32 || CAP <= 42) ? 2 : 1)) void rank_fused_kernel(const float *__restrict__ U, const float *__restrict__ V,
                                                          const float *__restrict__ item_base,
                                                          const float *__restrict__ user_base,
                                                          const int32_t *__restrict__ users, int64_t u0,
                                                          int64_t n_rows, int64_t n_items, int64_t work_per_wg,
                                                          int topk, const int64_t *__restrict__ excl_indptr,
                                                          const int32_t *__restrict__ excl_indices, int64_t excl_row0,
                                                          const uint32_t *__restrict__ excl_bits,
                                                          const int32_t *__restrict__ perm, float *tau_pub,
                                                          unsigned long long *__restrict__ part, int ablate) {
    // V / item_base are the scorer's RANK-ORDER copies: row p is item perm[p] (items sorted by a cheap upper
    // estimate of their scores, see build_rank_order); the candidates carry the original item ids.
    constexpr int KP = 2 * KT;  // padded row length of the device tables
    __shared__ unsigned long long keys[kBlk / 64][32][CAP];
    __shared__ int32_t ipos[3][32];   // (triple-buffered with wmask: one buffer index for all of a tile's meta words)
    __shared__ int cnt[kBlk / 64][32];
    __shared__ float tau[kBlk / 64][32];
    // the 4 waves of a workgroup walk the same item tiles: the B tile is staged once per workgroup
    // (coalesced 16-byte global loads, double-buffered) instead of gathered 4x through the TA
    __shared__ float btile[2][32][KP + 1];  // row stride == 1 (mod 32): the 32 lanes of a lane group of a fragment read hit 32 banks
    __shared__ uint32_t meta_dump[1];
    // exclusion bitmap words of the tile being compared: wmask[t % 3][wave][row] has bit c set when the c-th item of
    // tile t is excluded for that row (see excl_bitmap_kernel; 128 four-byte loads per workgroup and tile, L2-served).  Three buffers: tile t's words are read in the
    // survivor path of step t while faster waves already stage tile t+2.
    __shared__ uint32_t wmask[3][kBlk / 64][32];
    __shared__ float ibase[3][32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 31, half = lane >> 5;
    const int64_t n_item_tiles = (n_items + 31) / 32;
    // Balanced persistent decomposition: the (row block, item tile) space is linearised and cut into equal
    // contiguous ranges, one per resident workgroup, so the grid is exactly one wave of workgroups with no
    // ragged last round.  A range covers the tail of one row block, whole row blocks and the head of another;
    // every piece ("segment") emits topk candidates per row into part[segment ordinal within its row block].
    const int64_t n_row_blocks = (n_rows + 127) / 128;
    const int64_t work_total = n_row_blocks * n_item_tiles;
    int64_t pos = (int64_t)blockIdx.x * work_per_wg;
    const int64_t work_end = min(work_total, pos + work_per_wg);
    int64_t cur_rb = 0;  // row block of the segment being walked (stage_load reads its exclusion words)
    float bcur[KT];
    // staging: 32 items x KP floats = 8*KP float4; thread i moves float4 #i, #i+256, ...
    // Staging instructions are issue time taken from the matrix pipes (4.62 -> 4.05 ms with staging ablated at the ML-20M
    // shape), so the loads carry no per-tile address arithmetic: the tile's base is wave-uniform (scalar registers), the
    // thread's byte offset inside a tile is loop invariant, the tables are padded to whole tiles on the host (zero rows, NaN
    // item bases: no clamp, no validity test), and the tile's item bases / exclusion words come through ONE per-thread
    // pointer that advances by the thread's own stride (32 floats for the item-base lanes, one word for the exclusion-word
    // lanes, 0 for the rest: an unconditional load instead of two exec-mask branches).
    constexpr int STG = (32 * KP / 4 + kBlk - 1) / kBlk;
    v4f32 stg[STG];
    uint32_t voff[STG];
#pragma unroll
    for (int q = 0; q < STG; ++q) {
        const int idx = threadIdx.x + q * kBlk;
        voff[q] = idx < 32 * KP / 4 ? (uint32_t)(((idx / (KP / 4)) * KP + 4 * (idx % (KP / 4))) * 4) : 0u;
    }
    uint32_t stg_a = 0;     // item base (lanes 0..31 of wave 0) or exclusion word (threads 32..159) of the staged tile
    int32_t stg_id = 0;
    const uint32_t *pa = reinterpret_cast<const uint32_t *>(item_base);
    int64_t pa_stride = 0;
    const bool wave0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0;
    const int32_t *pb = perm;
    auto stage_seek = [&](int64_t it) {   // a segment starts: the meta pointers of tile `it`
        if (threadIdx.x < 32) {
            pa = reinterpret_cast<const uint32_t *>(item_base) + it * 32 + threadIdx.x;
            pa_stride = 32;
        } else if (excl_bits && threadIdx.x < 32 + 32 * (kBlk / 64) && !(ablate & 32)) {
            const int w = (threadIdx.x - 32) >> 5, rl = (threadIdx.x - 32) & 31;
            pa = excl_bits + ((cur_rb * (kBlk / 64) + w) * 32 + rl) * n_item_tiles + it;
            pa_stride = 1;
            // (a layout with the workgroup's 128 words of a tile contiguous was emulated — wrong words, right addresses —
            // and is not faster: 4.73 -> 4.63-4.71 ms with the survivor path off; the lines are L2-resident for 16 tiles.
            // Ablation bit 32 drops these loads altogether: 4.73 -> 4.60 ms — 128 lanes with 128 different lines per tile cost
            // 0.13 ms of address processing whatever the layout)
        } else {
            pa = reinterpret_cast<const uint32_t *>(item_base);
            pa_stride = 0;
        }
        pb = perm + it * 32 + (threadIdx.x & 31);
    };
    auto stage_load = [&](int64_t it) {   // tiles are staged in sequence: it = the previous call's + 1
        const char *vb = reinterpret_cast<const char *>(V) + it * (int64_t)(32 * KP * 4);
#pragma unroll
        for (int q = 0; q < STG; ++q) stg[q] = *reinterpret_cast<const v4f32 *>(vb + voff[q]);
        stg_a = *pa;
        pa += pa_stride;
        if (wave0) {   // (a scalar branch)
            stg_id = *pb;
            pb += 32;
        }
    };
    // the meta word's LDS slot: the thread's own (item-base lanes: ibase[.][lane], exclusion-word lanes: wmask[.][w][rl], the
    // rest: a dump word) + wbuf x the thread's own buffer stride — an unconditional store, like the load
    uint32_t meta_addr0 = lds_offset(&meta_dump[0]), meta_stride = 0;
    if (threadIdx.x < 32) {
        meta_addr0 = lds_offset(&ibase[0][threadIdx.x]);
        meta_stride = 32 * sizeof(float);
    } else if (excl_bits && threadIdx.x < 32 + 32 * (kBlk / 64)) {
        meta_addr0 = lds_offset(&wmask[0][(threadIdx.x - 32) >> 5][(threadIdx.x - 32) & 31]);
        meta_stride = (kBlk / 64) * 32 * sizeof(uint32_t);
    }
    auto stage_store = [&](int buf, int wbuf) {
#pragma unroll
        for (int q = 0; q < STG; ++q) {
            const int idx = threadIdx.x + q * kBlk;
            if ((q + 1) * kBlk <= 32 * KP / 4 || idx < 32 * KP / 4) {   // (compile-time true except in a ragged last group)
                float *dst = &btile[buf][idx / (KP / 4)][4 * (idx % (KP / 4))];
                dst[0] = stg[q].x; dst[1] = stg[q].y; dst[2] = stg[q].z; dst[3] = stg[q].w;
            }
        }
        *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(meta_addr0 + (uint32_t)wbuf * meta_stride) = stg_a;
        if (wave0) ipos[wbuf][threadIdx.x & 31] = stg_id;   // (a scalar branch; both halves of the wave hold the same 32 ids)
    };
    // B fragments of one tile: issued in groups of FG so that the loads of group g+1 are in flight while the
    // MFMAs of group g run (the scheduling barriers keep the compiler from sinking every load to its use)
    constexpr int FG = KT < 8 ? KT : 8;
    auto load_frags = [&](int buf, int g) {
#pragma unroll
        for (int t = g * FG; t < (g + 1) * FG; ++t) bcur[t] = btile[buf][col][2 * t + half];
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // The range is walked from its END: the head of a row block (tiles from 0: the likely winners, visited first)
    // is then the first thing its workgroup does, while the tail of the same row block is the LAST thing the
    // next workgroup does — by then the head's final thresholds are published in tau_pub and the tail starts
    // from them instead of -inf.  A value read too early is -inf: any lower bound keeps the result exact.
    int64_t pos_hi = work_end;
    while (pos_hi > pos) {
        const int64_t rb = (pos_hi - 1) / n_item_tiles;
        cur_rb = rb;
        const int64_t seg_lo = max(pos, rb * n_item_tiles);
        const int64_t t_begin = seg_lo - rb * n_item_tiles;
        const int64_t t_end = pos_hi - rb * n_item_tiles;
        const int64_t seg = (int64_t)blockIdx.x - (rb * n_item_tiles) / work_per_wg;
        pos_hi = seg_lo;
        const int64_t row_tile = rb * (kBlk / 64) + wave;
        if (lane < 32) {
            cnt[wave][lane] = 0;
            // rows beyond n_rows: tau = +inf (nothing ever passes); items beyond n_items get a NaN item base
            // (NaN >= thr is false), so the hot compare needs no validity masks
            const int64_t row = row_tile * 32 + lane;
            float t0 = -INFINITY;
            if (row >= n_rows) t0 = INFINITY;
            else if (t_begin > 0) t0 = __hip_atomic_load(tau_pub + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tau[wave][lane] = t0;
        }
        // A fragments + this lane's user bias
        float a[KT];
        {
            const int64_t r = row_tile * 32 + col;
            const bool ok = r < n_rows;
            const int64_t u = ok ? (users ? (int64_t)users[r] : u0 + r) : 0;
            const float *p = U + u * (2 * KT) + half;  // tables are zero-padded to 2*KT columns
#pragma unroll
            for (int t = 0; t < KT; ++t) a[t] = p[2 * t];
        }
        // per accumulator register: the row it belongs to, that row's user bias, threshold and the number of
        // candidates buffered for it (cnt_r is uniform inside each half-wave: kept in registers so that the
        // append path needs no LDS atomic and no LDS read to decide about compaction)
        float ubias[16], thr[16];
        int cnt_r[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int64_t row = row_tile * 32 + rl;
            const bool row_ok = row < n_rows;
            ubias[r] = 0.f;
            if (user_base && row_ok) ubias[r] = user_base[users ? (int64_t)users[row] : u0 + row];
            thr[r] = row_ok ? -INFINITY : INFINITY;
            cnt_r[r] = 0;
        }
        if (t_begin > 0) {  // thresholds published by the row block's head segment (wave-uniform branch)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) thr[r] = tau[wave][(r & 3) + 8 * (r >> 2) + 4 * half];
        }
        // wave-wide compaction of the rows whose buffer would overflow with this tile's survivors (or all rows at
        // the end of the segment): keep the topk best candidates of the row (in descending order) and raise its threshold to the topk-th.
        // Selection by RANK COUNTING — lane i holds candidate i and counts the candidates that beat it, each read
        // as an LDS broadcast: c independent loads and 3c VALU instructions with a single LDS round trip of
        // latency, instead of a 64-lane bitonic network's 21 DEPENDENT cross-lane exchanges (which held the whole
        // workgroup at the tile barrier for thousands of cycles per sort).
        // need_l: this half-wave's rows to compact (bit = row within the wave's 32-row tile)
        auto compact = [&](unsigned need_l) __attribute__((always_inline)) {
            unsigned need = __builtin_amdgcn_readlane(need_l, 0) | __builtin_amdgcn_readlane(need_l, 32);
            if (col == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) cnt[wave][(r & 3) + 8 * (r >> 2) + 4 * half] = cnt_r[r];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            while (need) {
                const int rl = __builtin_ctz(need);
                need &= need - 1u;
                const int c = cnt[wave][rl];
                unsigned long long *krow = keys[wave][rl];
                unsigned long long mine = lane < c ? krow[lane] : 0ull;
                if (excl_indptr && !excl_bits) {  // no bitmap (huge catalogue): drop excluded items before they can raise the threshold
                    bool dropped = false;
                    if (mine != 0ull) {
                        const int32_t item = (int32_t)(uint32_t)mine;
                        const int64_t grow = excl_row0 + row_tile * 32 + rl;
                        int64_t lo = excl_indptr[grow], hi = excl_indptr[grow + 1];
                        const int64_t end = hi;
                        while (lo < hi) {
                            const int64_t mid = lo + ((hi - lo) >> 1);
                            if (excl_indices[mid] < item) lo = mid + 1; else hi = mid;
                        }
                        dropped = lo < end && excl_indices[lo] == item;
                    }
                    if (dropped) {
                        mine = 0ull;
                        krow[lane] = 0ull;  // the broadcast reads below must see it as absent
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                int rank = 0;
                for (int j0 = 0; j0 < c; j0 += 8) {  // 8 independent broadcast loads in flight, then 8 compares
                    unsigned long long kj[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) kj[q] = krow[min(j0 + q, CAP - 1)];
#pragma unroll
                    for (int q = 0; q < 8; ++q) rank += (j0 + q < c && kj[q] > mine) ? 1 : 0;
                }
                const bool live = mine != 0ull;
                const int n_keep = min(__popcll(__ballot(live)), topk);
                const unsigned long long kth_mask = __ballot(live && rank == topk - 1);
                if (lane < CAP) krow[lane] = 0ull;       // LDS operations of one wave execute in order
                if (live && rank < topk) krow[rank] = mine;
                if (kth_mask != 0ull) {  // wave-uniform
                    const int src = __builtin_ctzll(kth_mask);
                    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(mine >> 32), src);
                    if (lane == 0 && row_tile * 32 + rl < n_rows) tau[wave][rl] = key_to_float(hi);
                }
                if (lane == 0) cnt[wave][rl] = n_keep;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
                thr[r] = tau[wave][rl];
                cnt_r[r] = cnt[wave][rl];
            }
        };
        // Software pipeline: step `it` issues the MFMA chain of tile it+1 and, interleaved between those MFMAs
        // in program order, evaluates the finished accumulators of tile it.  On this hardware VALU issue cycles
        // ADD to the matrix pipe's time (tools/mfma_probe: 2048 + ~4 cycles per VALU instruction per tile), so the
        // compare is kept to one add and one v_cmp (lane mask straight into SGPRs) per accumulator register, and
        // the loop is unrolled by two so that the accumulators ping-pong instead of being copied.
        f32x16 acc_a = zero16, acc_b = zero16;
        int wb = 0;   // meta buffer (tile index mod 3) of the tile a step compares
        float ib_a = 0.f, ib_b = 0.f;
        int32_t id_a = 0, id_b = 0;
        stage_seek(t_begin);
        stage_load(t_begin);
        stage_store(0, 0);
        __syncthreads();
        if (t_begin + 1 < t_end) stage_load(t_begin + 1);
        {
#pragma unroll
            for (int t = 0; t < KT; ++t) bcur[t] = btile[0][col][2 * t + half];
            ib_a = ibase[0][col];
            id_a = ipos[0][col];
#pragma unroll
            for (int t = 0; t < KT; ++t) acc_a = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bcur[t], acc_a, 0, 0, 0);
        }
        if (t_begin + 1 < t_end) stage_store(1, 1);
        __syncthreads();
        auto step = [&](int64_t it, f32x16 &acc_cur, f32x16 &acc_nxt, float &ib_cur, float &ib_nxt, int32_t &id_cur,
                        int32_t &id_nxt) __attribute__((always_inline)) {
            const int buf_next = (int)((it + 1 - t_begin) & 1);
            // ---- B fragments + MFMA chain of tile it+1, compare of tile it -----------------------------------
            // (the first fragment group is requested before anything else: its LDS round trip is the one nothing hides —
            // the tile's buffer only became visible at the barrier)
            load_frags(buf_next, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (it + 2 < t_end && !(ablate & 2)) stage_load(it + 2);
            const int wb1 = wb == 2 ? 0 : wb + 1, wb2 = wb1 == 2 ? 0 : wb1 + 1;   // meta buffers of tiles it + 1, it + 2
            ib_nxt = ibase[wb1][col];
            id_nxt = ipos[wb1][col];
            const int32_t item = id_cur;
            unsigned long long hm[16];
            float sc[16];
            // (moving the first four compares in front of the chain, into the wait for the first fragment group: no gain)
#pragma unroll
            for (int g = 0; g < KT / FG; ++g) {
                if (g + 1 < KT / FG) load_frags(buf_next, g + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = g * FG; t < (g + 1) * FG; ++t) {
                    if (t == 0) acc_nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bcur[t], zero16, 0, 0, 0);
                    else acc_nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bcur[t], acc_nxt, 0, 0, 0);
                    if (t < 16) {
                        const int r = t;
                        // (v_pk_add_f32 on register pairs — 8 packed adds instead of 16 — was measured: 5.900 against 5.903 ms,
                        // no gain: the packed add takes the issue time of two)
                        sc[r] = UB ? (ib_cur + ubias[r]) + acc_cur[r] : ib_cur + acc_cur[r];
                        hm[r] = __ballot(sc[r] >= thr[r]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (KT < 16) {
#pragma unroll
                for (int r = KT; r < 16; ++r) {
                    sc[r] = UB ? (ib_cur + ubias[r]) + acc_cur[r] : ib_cur + acc_cur[r];
                    hm[r] = __ballot(sc[r] >= thr[r]);
                }
            }
            unsigned long long any = 0ull;
#pragma unroll
            for (int r = 0; r < 16; ++r) any |= hm[r];
            if (ablate & 1) any = 0ull;
            // excluded items are removed from the survivor masks with scalar operations: one LDS read brings the 32
            // bitmap words of this wave's rows for tile `it`, two v_readlane per register that has survivors at all
            auto drop_excluded = [&]() __attribute__((always_inline)) {
                // (requesting this word at the top of the step, behind the MFMA chain, was measured in round 4: 5.50-5.55 ms
                // against 5.45 ms — no gain, one more live register across the chain)
                const uint32_t wv = wmask[wb][wave][col];
                unsigned long long left = 0ull;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (hm[r] != 0ull) {  // wave-uniform
                        const int rl = (r & 3) + 8 * (r >> 2);
                        // (v_readlane returns int: widen through uint32_t, a set bit 31 must not sign-extend)
                        const unsigned long long m = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(wv, rl) |
                                                     ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(wv, rl + 4) << 32);
                        hm[r] &= ~m;
                        left |= hm[r];
                    }
                }
                return left;
            };
            if (any != 0ull && excl_bits) any = drop_excluded();
            // ---- rare path: append the survivors of tile it ---------------------------------------------------
            if (any != 0ull) {
                // rows whose buffer cannot take this tile's survivors are compacted first (their thresholds rise,
                // so the survivors are re-evaluated); CAP >= topk + 32 guarantees room afterwards
                // (VALU work only under wave-uniform `hm[r] != 0` branches: scalar instructions are free here, vector
                // instructions are not)
                unsigned over_l = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (hm[r] != 0ull) {
                        const unsigned mine = half ? (unsigned)(hm[r] >> 32) : (unsigned)hm[r];
                        if (cnt_r[r] + __popc(mine) > CAP) over_l |= 1u << ((r & 3) + 8 * (r >> 2) + 4 * half);
                    }
                }
                if ((__builtin_amdgcn_readlane(over_l, 0) | __builtin_amdgcn_readlane(over_l, 32)) != 0u) {
                    compact(over_l);
#pragma unroll
                    for (int r = 0; r < 16; ++r) hm[r] = __ballot(sc[r] >= thr[r]);
                    if (excl_bits) (void)drop_excluded();
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned long long m = hm[r];
                    if (m != 0ull) {  // wave-uniform
                        const unsigned mine = half ? (unsigned)(m >> 32) : (unsigned)m;
                        if ((mine >> col) & 1u) {
                            const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
                            const int slot = cnt_r[r] + __popc(mine & ((1u << col) - 1u));
                            keys[wave][rl][slot] =
                                ((unsigned long long)order_key(sc[r]) << 32) | (unsigned long long)(uint32_t)item;
                        }
                        cnt_r[r] += __popc(mine);
                    }
                }
            }
            if (it + 2 < t_end && !(ablate & 8)) stage_store((int)((it + 2 - t_begin) & 1), wb2);
            if (!(ablate & 4)) __syncthreads();  // tile it+2 visible; the buffer of tile it+1 is fully read by everybody
            wb = wb1;
        };
        for (int64_t it = t_begin; it < t_end; it += 2) {
            step(it, acc_a, acc_b, ib_a, ib_b, id_a, id_b);
            if (it + 1 < t_end) step(it + 1, acc_b, acc_a, ib_b, ib_a, id_b, id_a);
        }
        compact(0xffffffffu);
        // emit this segment's candidates: part[segment][row][topk]
        for (int rl = 0; rl < 32; ++rl) {
            const int64_t row = row_tile * 32 + rl;
            if (row >= n_rows) break;  // also covers waves whose whole row tile is out of range
            if (lane < topk) part[(seg * n_rows + row) * topk + lane] = keys[wave][rl][lane];
        }
        if (t_begin == 0 && t_end < n_item_tiles && lane < 32 && row_tile * 32 + lane < n_rows)
            __hip_atomic_store(tau_pub + row_tile * 32 + lane, tau[wave][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// Exclusion bitmap of the fused top-k kernel: one word per (row, tile of 32 items in the scorer's RANK ORDER),
//     bits[row * n_item_tiles + item_tile]  bit c  <=>  the item at position item_tile*32 + c is excluded for that row
// (row = position in this call's user list, position p = inv_perm[item]).  One WAVE per TWO rows gathers each row's exclusion
// list into a private LDS bitmap (LDS integer atomics: ~2.4 cycles per set bit) in chunks of `chunk_tiles` words and
// writes it out contiguously.  The lists come either per ROW of this call (indptr[r], by_user = 0) or per USER id from
// the resident CSR registered with cornac_hip_scorer_set_exclusions (by_user = 1: row r is user users[r] or u0 + r).
// Rows beyond n_rows (the padding of the last workgroup of the top-k kernel) get all-zero words.
constexpr int kBitmapChunk = 1024;  // words per row and pass (4 KB of LDS per row)
constexpr int kBitmapRows = 1;      // rows per wave.  A row is a chain of dependent round trips (indptr -> indices -> inv_perm ->
                                    // LDS bit); two rows per wave (two chains in flight) were measured: 263 us against 200-212 us
                                    // for all 138 493 users — the second 4 KB of LDS per wave costs more occupancy than it hides
__global__ __launch_bounds__(kBlk) void excl_bitmap_kernel(const int64_t *__restrict__ indptr,
                                                          const int32_t *__restrict__ indices,
                                                          const int32_t *__restrict__ users, int64_t u0, int by_user,
                                                          const int32_t *__restrict__ inv_perm, int64_t n_rows,
                                                          int64_t n_rows_padded, int64_t n_item_tiles,
                                                          uint32_t *__restrict__ bits) {
    constexpr int NR = kBitmapRows;
    __shared__ uint32_t bm_all[kBlk / 64][NR][kBitmapChunk];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * (kBlk / 64) + wave) * NR;
    if (row0 >= n_rows_padded) return;
    int64_t lo[NR], cnt[NR];
    int64_t cnt_max = 0;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        lo[q] = 0;
        cnt[q] = 0;
        const int64_t row = row0 + q;
        if (row < n_rows) {
            const int64_t key = by_user ? (users ? (int64_t)users[row] : u0 + row) : row;
            lo[q] = indptr[key];
            cnt[q] = indptr[key + 1] - lo[q];
        }
        cnt_max = max(cnt_max, cnt[q]);
    }
    for (int64_t c0 = 0; c0 < n_item_tiles; c0 += kBitmapChunk) {
        const int nt = (int)min((int64_t)kBitmapChunk, n_item_tiles - c0);
#pragma unroll
        for (int q = 0; q < NR; ++q)
            for (int i = lane; i < nt; i += 64) bm_all[wave][q][i] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int64_t p = lane; p < cnt_max; p += 64) {
            int32_t e[NR], pos[NR];
#pragma unroll
            for (int q = 0; q < NR; ++q) e[q] = p < cnt[q] ? indices[lo[q] + p] : 0;   // (item 0: a valid index of inv_perm)
#pragma unroll
            for (int q = 0; q < NR; ++q) pos[q] = inv_perm[e[q]];
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int t = (pos[q] >> 5) - (int)c0;
                if (p < cnt[q] && t >= 0 && t < nt) atomicOr(&bm_all[wave][q][t], 1u << (pos[q] & 31));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (written once, read once by the top-k kernel much later: streamed past the caches)
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            if (row0 + q < n_rows_padded)
                for (int i = lane; i < nt; i += 64)
                    __builtin_nontemporal_store(bm_all[wave][q][i], &bits[(row0 + q) * n_item_tiles + c0 + i]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// merge the segments' candidates of one row: bitonic sort of n_strips*topk keys in LDS, emit the topk best
__global__ __launch_bounds__(64) void rank_merge_kernel(const unsigned long long *__restrict__ part, int n_strips,
                                                        int64_t n_rows, int topk, int pad,
                                                        int32_t *__restrict__ items_out,
                                                        float *__restrict__ scores_out) {
    extern __shared__ unsigned long long mlist[];
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = n_strips * topk;
    for (int i = tid; i < pad; i += 64) {
        unsigned long long key = 0ull;
        if (i < n) key = part[((int64_t)(i / topk) * n_rows + row) * topk + (i % topk)];
        mlist[i] = key;
    }
    __syncthreads();
    for (int kk = 2; kk <= pad; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < pad; i += 64) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = mlist[i], y = mlist[ixj];
                    const bool desc = (i & kk) == 0;
                    if (desc ? (x < y) : (x > y)) {
                        mlist[i] = y;
                        mlist[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < topk; i += 64) {
        const unsigned long long key = mlist[i];
        items_out[row * topk + i] = key ? (int32_t)(uint32_t)key : -1;
        scores_out[row * topk + i] = key ? key_to_float((uint32_t)(key >> 32)) : -INFINITY;
    }
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

// ---- full ranking: bitonic sort of all keys of a row (one workgroup per row) ---------------------------------
// The network's stages with a partner distance below kSortChunk run inside LDS on 64 KB chunks; only the few
// stages with a larger distance (3 of 120 for 32 768 keys) touch the global scratch.  Directions follow the
// GLOBAL key index, so the chunks assemble into the standard bitonic sort.
constexpr int kSortBlk = 1024;
constexpr int kSortChunk = 8192;  // keys per LDS chunk
__global__ __launch_bounds__(kSortBlk) void full_sort_kernel(const float *__restrict__ scores,
                                                             const uint8_t *__restrict__ excl_mask, int64_t n_items,
                                                             int64_t n_pad, unsigned long long *__restrict__ scratch,
                                                             int topk, int32_t *__restrict__ items_out,
                                                             float *__restrict__ scores_out) {
    __shared__ unsigned long long lk[kSortChunk];
    const int64_t row = blockIdx.x;
    const float *srow = scores + row * n_items;
    const uint8_t *erow = excl_mask ? excl_mask + row * n_items : nullptr;
    unsigned long long *keys = scratch + row * n_pad;
    const int tid = threadIdx.x;
    const int64_t ch = n_pad < kSortChunk ? n_pad : kSortChunk;  // both powers of two
    // stages j = j_hi .. 1 of merge step kk on the chunk starting at global index c0, in LDS
    auto lds_stages = [&](int64_t c0, int64_t kk, int64_t j_hi) {
        for (int64_t j = j_hi; j > 0; j >>= 1) {
            for (int64_t p = tid; p < (ch >> 1); p += kSortBlk) {  // one thread per compare-exchange pair
                const int64_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), ixj = i | j;
                const unsigned long long x = lk[i], y = lk[ixj];
                const bool desc = ((c0 + i) & kk) == 0;
                if (desc ? (x < y) : (x > y)) {
                    lk[i] = y;
                    lk[ixj] = x;
                }
            }
            __syncthreads();
        }
    };
    // phase A: build the keys chunk by chunk and sort every chunk completely (merge steps kk <= ch)
    for (int64_t c0 = 0; c0 < n_pad; c0 += ch) {
        for (int64_t i = tid; i < ch; i += kSortBlk) {
            const int64_t g = c0 + i;
            lk[i] = g < n_items ? composite_key(srow, erow, g) : 0ull;
        }
        __syncthreads();
        for (int64_t kk = 2; kk <= ch; kk <<= 1) lds_stages(c0, kk, kk >> 1);
        for (int64_t i = tid; i < ch; i += kSortBlk) keys[c0 + i] = lk[i];
        __syncthreads();
    }
    // phase B: merge steps across chunks — far stages in the global scratch, the rest per chunk in LDS
    for (int64_t kk = ch << 1; kk <= n_pad; kk <<= 1) {
        for (int64_t j = kk >> 1; j >= ch; j >>= 1) {
            for (int64_t p = tid; p < (n_pad >> 1); p += kSortBlk) {
                const int64_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), ixj = i | j;
                const unsigned long long x = keys[i], y = keys[ixj];
                const bool desc = (i & kk) == 0;
                if (desc ? (x < y) : (x > y)) {
                    keys[i] = y;
                    keys[ixj] = x;
                }
            }
            __syncthreads();  // workgroup-scope: one workgroup owns the row, L1 is shared by its waves
        }
        for (int64_t c0 = 0; c0 < n_pad; c0 += ch) {
            for (int64_t i = tid; i < ch; i += kSortBlk) lk[i] = keys[c0 + i];
            __syncthreads();
            lds_stages(c0, kk, ch >> 1);
            for (int64_t i = tid; i < ch; i += kSortBlk) keys[c0 + i] = lk[i];
            __syncthreads();
        }
    }
    for (int64_t i = tid; i < topk; i += kSortBlk) {
        const unsigned long long key = keys[i];
        items_out[row * topk + i] = key ? (int32_t)(uint32_t)key : -1;
        scores_out[row * topk + i] = key ? key_to_float((uint32_t)(key >> 32)) : -INFINITY;
    }
}


// ---- where listed items stand in a row's ranking, without ranking the row ------------------------------------
// For every listed (row, item) "target" (the evaluation loop's test positives of that user) count the row's
// candidates that precede it:  greater = candidates with a strictly higher score,  pos = its 0-based position in
// the ranked order (descending score, ties by higher item index — the order rank_topk produces),  ge = candidates
// with score >= its own (itself included).  That is everything the full-list metrics (AUC, MAP, MRR) read off a
// full ranking, so neither the sort nor the [users, items] ranking ever has to exist.  One workgroup per row; the
// row's targets go through registers kPosT at a time while the row's scores stream from the score tile (one row
// is ~100 KB: L2-resident after the first pass).  Integer compares on the order-preserving score bits: exact.
constexpr int kPosT = 8;
// The row is read with 16-byte loads (4 candidates per lane) wherever its alignment allows, eight targets per pass
// over the row; counts are per-lane registers reduced across the workgroup once per pass.
__global__ __launch_bounds__(kBlk) void rank_positions_kernel(const float *__restrict__ scores,
                                                              const uint8_t *__restrict__ excl_mask, int64_t n_items,
                                                              const int64_t *__restrict__ tgt_indptr,
                                                              const int32_t *__restrict__ tgt_indices, int64_t row0,
                                                              int32_t *__restrict__ greater_out,
                                                              int32_t *__restrict__ pos_out,
                                                              int32_t *__restrict__ ge_out,
                                                              float *__restrict__ score_out) {
    const int64_t row = blockIdx.x;
    const float *srow = scores + row * n_items;
    const uint8_t *erow = excl_mask ? excl_mask + row * n_items : nullptr;
    const int64_t lo = tgt_indptr[row0 + row], hi = tgt_indptr[row0 + row + 1];
    __shared__ int cnt[kPosT * 3];
    // items [head, body_end) are read four at a time from 16-byte aligned addresses; the rest one by one
    const int64_t mis = (int64_t)((reinterpret_cast<uintptr_t>(srow) >> 2) & 3);
    const int64_t head = min(n_items, (4 - mis) & 3);
    const int64_t body_end = head + ((n_items - head) & ~int64_t(3));
    const bool vec_mask = !erow || ((reinterpret_cast<uintptr_t>(erow) + head) & 3) == 0;
    for (int64_t t0 = lo; t0 < hi; t0 += kPosT) {
        uint32_t ts[kPosT];
        bool ok[kPosT];
        int32_t titem[kPosT];
#pragma unroll
        for (int q = 0; q < kPosT; ++q) {
            const int64_t t = t0 + q;
            const int32_t item = t < hi ? tgt_indices[t] : -1;
            ok[q] = item >= 0 && item < n_items && !(erow && erow[item]);
            ts[q] = ok[q] ? order_key(srow[item]) : 0xFFFFFFFFu;
            titem[q] = item;
        }
        // per-lane counts: g = candidates strictly above the target, eq = candidates with the target's score,
        // tie = of those, the ones rank() orders first (higher item index).  The common case costs two VALU per
        // (candidate, target) — v_cmp_gt + add-with-carry — plus one v_cmp_eq whose lane mask is OR-ed on the scalar
        // unit; only a group of candidates in which some score EQUALS some target's (the target itself, real ties)
        // takes the exact path below.  A dead candidate gets key 0, which is above no target.
        int g[kPosT], eq[kPosT], tie[kPosT];
#pragma unroll
        for (int q = 0; q < kPosT; ++q) g[q] = eq[q] = tie[q] = 0;
        auto visit = [&](const bool *live, const float *sc, int64_t i, int n) __attribute__((always_inline)) {
            uint32_t oc[4];
            unsigned long long any_eq = 0;
#pragma unroll
            for (int c = 0; c < n; ++c) {
                oc[c] = live[c] ? order_key(sc[c]) : 0u;
#pragma unroll
                for (int q = 0; q < kPosT; ++q) {
                    g[q] += oc[c] > ts[q];
                    any_eq |= __ballot(oc[c] == ts[q]);
                }
            }
            if (any_eq) {
#pragma unroll
                for (int c = 0; c < n; ++c) {
                    asm volatile("" : "+v"(oc[c]));  // recompute the compares here: keeping 32 lane masks alive spills
#pragma unroll
                    for (int q = 0; q < kPosT; ++q) {
                        const bool same = live[c] && oc[c] == ts[q];
                        eq[q] += same;
                        tie[q] += same && (uint32_t)(i + c) > (uint32_t)titem[q];
                    }
                }
            }
        };
        for (int64_t i = threadIdx.x; i < head; i += kBlk) {
            const bool lv = !(erow && erow[i]);
            const float sc = srow[i];
            visit(&lv, &sc, i, 1);
        }
        // (the loop bound keeps whole waves together: lanes beyond the body contribute nothing; the next iteration's
        // 16-byte loads are issued before the current one's compares: two loads in flight per lane)
        auto fetch = [&](int64_t i0, v4f32 &sv, uint32_t &em) __attribute__((always_inline)) {
            const bool in = i0 < body_end;
            sv = in ? *reinterpret_cast<const v4f32 *>(srow + i0) : v4f32{0.f, 0.f, 0.f, 0.f};
            em = 0u;
            if (erow && in) {
                if (vec_mask) em = *reinterpret_cast<const uint32_t *>(erow + i0);
                else em = (uint32_t)erow[i0] | ((uint32_t)erow[i0 + 1] << 8) | ((uint32_t)erow[i0 + 2] << 16) | ((uint32_t)erow[i0 + 3] << 24);
            }
        };
        const int64_t i_first = head + 4 * (int64_t)threadIdx.x, i_stop = body_end + 4 * (int64_t)(kBlk - 1);
        v4f32 sv_n = {0.f, 0.f, 0.f, 0.f};
        uint32_t em_n = 0u;
        if (i_first < i_stop) fetch(i_first, sv_n, em_n);
        for (int64_t i0 = i_first; i0 < i_stop; i0 += 4 * kBlk) {
            const v4f32 sv = sv_n;
            const uint32_t em = em_n;
            if (i0 + 4 * kBlk < i_stop) fetch(i0 + 4 * kBlk, sv_n, em_n);
            const bool in = i0 < body_end;
            const bool lv[4] = {in && !(em & 0xffu), in && !(em & 0xff00u), in && !(em & 0xff0000u), in && !(em & 0xff000000u)};
            const float sc[4] = {sv.x, sv.y, sv.z, sv.w};
            visit(lv, sc, i0, 4);
        }
        for (int64_t i = body_end + threadIdx.x; i < n_items; i += kBlk) {
            const bool lv = !(erow && erow[i]);
            const float sc = srow[i];
            visit(&lv, &sc, i, 1);
        }
#pragma unroll
        for (int q = 0; q < kPosT; ++q)
            for (int o = 32; o > 0; o >>= 1) {
                g[q] += __shfl_xor(g[q], o, 64);
                eq[q] += __shfl_xor(eq[q], o, 64);
                tie[q] += __shfl_xor(tie[q], o, 64);
            }
        if (threadIdx.x < kPosT * 3) cnt[threadIdx.x] = 0;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int q = 0; q < kPosT; ++q) {
                atomicAdd(&cnt[q * 3 + 0], g[q]);
                atomicAdd(&cnt[q * 3 + 1], g[q] + tie[q]);
                atomicAdd(&cnt[q * 3 + 2], g[q] + eq[q]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kPosT; ++q) {
            const int64_t t = t0 + q;
            if (threadIdx.x == q && t < hi) {
                greater_out[t] = ok[q] ? cnt[q * 3 + 0] : -1;
                pos_out[t] = ok[q] ? cnt[q * 3 + 1] : -1;
                ge_out[t] = ok[q] ? cnt[q * 3 + 2] : -1;
                score_out[t] = ok[q] ? srow[titem[q]] : -INFINITY;
            }
        }
        __syncthreads();
    }
}

}  // namespace chip

using namespace chip;

struct cornac_hip_scorer {
    int device = 0;
    int64_t n_users = 0, n_items = 0;
    int k = 0;
    int ld = 0;  // row stride of the device tables: k zero-padded to 16/32/64/128 for the MFMA kernels
    hipStream_t stream = nullptr;
    DevBuf<float> U, V, item_base, user_base;
    // rank-order copies for the fused top-k kernel: row p = item perm[p] (build_rank_order)
    DevBuf<float> Vr, ibr;
    DevBuf<int32_t> perm, inv_perm;
    DevBuf<uint32_t> excl_bits;  // exclusion bitmap of the current fused launch (excl_bitmap_kernel)
    void *pinned_out = nullptr;  // page-locked host memory lent to the caller for result copies (cornac_hip_scorer_host_buffer)
    size_t pinned_bytes = 0;
    std::vector<void *> pinned_retired;
    // exclusion lists per USER id, resident on the device (cornac_hip_scorer_set_exclusions)
    DevBuf<int64_t> res_excl_indptr;
    DevBuf<int32_t> res_excl_indices;
    bool has_res_excl = false;
    DevBuf<float> tau_pub;  // per row: threshold published by the head segment of its row block
    bool has_user_base = false, is_set = false;
    DevBuf<float> scores;  // workspace [rows_cap, n_items]
    DevBuf<uint8_t> excl;
    DevBuf<int32_t> d_users, d_items_out, d_excl_indices;
    DevBuf<int64_t> d_excl_indptr;
    DevBuf<float> d_scores_out;
    DevBuf<int64_t> d_tgt_indptr;
    DevBuf<int32_t> d_tgt_indices, d_tgt_counts;
    DevBuf<float> d_tgt_scores;
    DevBuf<unsigned long long> sort_scratch, part;
    // float64 tables of a model trained in double (cornac_hip_scorer_set_f64): score_user only
    DevBuf<double> U64, V64, ib64, ub64, scores64;
    bool f64_set = false, f64_user_base = false;
};

// The fused top-k kernel appends every score that beats its row's running topk-th score, so its cost depends on
// the order in which items are visited: ~topk ln(N/topk) survivors per row in random order, far fewer when
// likely-high items come first.  Items are therefore visited in descending order of a cheap upper estimate of
// their scores over the user population,  item_base[i] + <mean u, v_i> + 2 sqrt(v_i^T Cov(u) v_i)  (moments of the
// user vectors from a sample of users; the Cauchy-Schwarz bound is far too loose in k dimensions).  Any order gives
// the same result (candidates carry original item ids; ties are decided on those).
static __global__ __launch_bounds__(256) void permute_rows_kernel(const float *__restrict__ V, const float *__restrict__ ib,
                                                                  const int32_t *__restrict__ perm, int64_t n, int ld,
                                                                  float *__restrict__ Vr, float *__restrict__ ibr) {
    const int gl = threadIdx.x & 15;
    for (int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4; p < n; p += ((int64_t)gridDim.x * 256) >> 4) {
        const int64_t src = perm[p];
        for (int f = gl * 4; f < ld; f += 64)
            *reinterpret_cast<v4f32 *>(Vr + p * ld + f) = *reinterpret_cast<const v4f32 *>(V + src * ld + f);
        if (gl == 0) ibr[p] = ib[src];
    }
}

static void build_rank_order(cornac_hip_scorer_t h, const float *U, const float *V, const float *item_base) {
    const int64_t ni = h->n_items, nu = h->n_users;
    const int k = h->k;
    // first and second moments of the user vectors from a sample of users: the score of item i over users has
    // mean  b_i + <ubar, v_i>  and variance  v_i^T Cov(u) v_i ; the priority is mean + 2 standard deviations
    // (when the k x k quadratic form per item is too expensive, the isotropic estimate |v_i| rms|u| / sqrt(k))
    const int64_t sample = std::min<int64_t>(nu, 16384), stride = std::max<int64_t>(1, nu / sample);
    const bool full_cov = (double)ni * k * k <= 4e9;
    std::vector<double> mean((size_t)k, 0.0), cov(full_cov ? (size_t)k * k : 0, 0.0);
    double ss = 0;
    int64_t cnt = 0;
    for (int64_t u = 0; u < nu; u += stride, ++cnt) {
        const float *ur = U + u * k;
        for (int f = 0; f < k; ++f) {
            mean[(size_t)f] += ur[f];
            ss += (double)ur[f] * ur[f];
        }
        if (full_cov)
            for (int f = 0; f < k; ++f)
                for (int g = f; g < k; ++g) cov[(size_t)f * k + g] += (double)ur[f] * ur[g];
    }
    const double inv = 1.0 / (double)std::max<int64_t>(cnt, 1);
    for (int f = 0; f < k; ++f) mean[(size_t)f] *= inv;
    if (full_cov)
        for (int f = 0; f < k; ++f)
            for (int g = f; g < k; ++g) {
                const double c = cov[(size_t)f * k + g] * inv - mean[(size_t)f] * mean[(size_t)g];
                cov[(size_t)f * k + g] = cov[(size_t)g * k + f] = c;
            }
    const double iso = std::max(0.0, ss * inv - [&] { double m2 = 0; for (int f = 0; f < k; ++f) m2 += mean[(size_t)f] * mean[(size_t)f]; return m2; }()) / (double)k;
    std::vector<float> pri((size_t)ni);
    std::vector<double> tmp((size_t)k);
    for (int64_t i = 0; i < ni; ++i) {
        const float *vr = V + i * k;
        double m = 0, var = 0;
        for (int f = 0; f < k; ++f) m += mean[(size_t)f] * vr[f];
        if (full_cov) {
            for (int f = 0; f < k; ++f) {
                double t = 0;
                for (int g = 0; g < k; ++g) t += cov[(size_t)f * k + g] * vr[g];
                var += t * vr[f];
            }
        } else {
            for (int f = 0; f < k; ++f) var += (double)vr[f] * vr[f];
            var *= iso;
        }
        pri[(size_t)i] = (float)((item_base ? (double)item_base[i] : 0.0) + m + 2.0 * std::sqrt(std::max(var, 0.0)));
    }
    std::vector<int32_t> order((size_t)ni);
    for (int64_t i = 0; i < ni; ++i) order[(size_t)i] = (int32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        const float pa = pri[(size_t)a], pb = pri[(size_t)b];
        return (pa > pb) || (pa != pa && pb == pb);  // NaN priorities first: any order is valid, this one is total
    });
    // padded to whole tiles of 32 items for the fused kernel's staging (zero rows, NaN bases — a NaN never beats a threshold —
    // item id 0): its loads need no clamp
    const int64_t ni_pad = (ni + 31) / 32 * 32;
    h->perm.ensure((size_t)ni_pad);
    h->Vr.ensure((size_t)ni_pad * h->ld);
    h->ibr.ensure((size_t)ni_pad);
    HIP_CHECK(hipMemsetAsync(h->perm.p, 0, (size_t)ni_pad * sizeof(int32_t), h->stream));
    HIP_CHECK(hipMemsetAsync(h->Vr.p, 0, (size_t)ni_pad * h->ld * sizeof(float), h->stream));
    HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)h->ibr.p, 0x7fc00000u /* NaN */, (size_t)ni_pad, h->stream));
    h->perm.upload(order.data(), (size_t)ni, h->stream);
    std::vector<int32_t> inv_order((size_t)ni);
    for (int64_t p = 0; p < ni; ++p) inv_order[(size_t)order[(size_t)p]] = (int32_t)p;
    h->inv_perm.ensure((size_t)ni);
    h->inv_perm.upload(inv_order.data(), (size_t)ni, h->stream);
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((ni * 16 + 255) / 256, 8192));
    hipLaunchKernelGGL(permute_rows_kernel, dim3(grid), dim3(256), 0, h->stream, h->V.p, h->item_base.p, h->perm.p, ni, h->ld,
                       h->Vr.p, h->ibr.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(h->stream));  // `order` is pageable host memory
}

// a CSR handed over the C ABI: indptr non-decreasing from 0, every index a valid item (the kernels index tables
// with them); the length of `indices` is taken to be indptr[n], as in scipy
static void validate_csr(const int64_t *indptr, const int32_t *indices, int64_t n, int64_t n_items, const char *what) {
    REQUIRE(indptr[0] == 0, "%s_indptr must start at 0", what);
    for (int64_t r = 0; r < n; ++r)
        REQUIRE(indptr[r + 1] >= indptr[r], "%s_indptr decreases at row %lld", what, (long long)r);
    const int64_t ne = indptr[n];
    int32_t lo = 0, hi = 0;
    for (int64_t p = 0; p < ne; ++p) {
        lo = std::min(lo, indices[p]);
        hi = std::max(hi, indices[p]);
    }
    REQUIRE(lo >= 0 && hi < n_items, "%s_indices out of range [0, %lld)", what, (long long)n_items);
}

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
        if (h->ld == 16) go(score_gemm_mfma_kernel<8, 2>, 2);
        else if (h->ld == 32) go(score_gemm_mfma_kernel<16, 2>, 2);
        else if (h->ld == 64) go(score_gemm_mfma_kernel<32, 2>, 2);
        else go(score_gemm_mfma_kernel<64, 1>, 1);
    } else {
        dim3 grid((unsigned)((h->n_items + kBlk - 1) / kBlk), (unsigned)n);
        hipLaunchKernelGGL(score_valu_kernel, grid, dim3(kBlk), (size_t)k * sizeof(float), h->stream, h->U.p, h->V.p,
                           h->item_base.p, ub, d_users, u0, h->n_items, k, h->ld, h->scores.p);
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

constexpr int kFusedMaxTopk = 32;

static bool can_fuse(cornac_hip_scorer_t h, int topk) { return topk <= kFusedMaxTopk && h->k <= 128; }

// fused GEMM + top-k for rows [0, n): users from d_users (or u0 + row); optional exclusion CSR on the device
constexpr int64_t kMaxBitmapTiles = 8192;  // item tiles (x32 items) up to which the exclusion bitmap is used

// excl_by_user: d_excl_indptr is indexed by user id (resident lists) instead of by row of this call
static void launch_rank_fused_rows(cornac_hip_scorer_t h, const int32_t *d_users, int64_t u0, int64_t n, int topk,
                                   const int64_t *d_excl_indptr, const int32_t *d_excl_indices, int64_t excl_row0,
                                   int32_t *items_out, float *scores_out, bool excl_by_user) {
    const float *ub = h->has_user_base ? h->user_base.p : nullptr;
    const DeviceInfo &di = device_info(h->device);
    const int ablate = prof_env_int("CORNAC_HIP_RANK_ABLATE", 0);  // profile builds only (csrc/common.h)
    const int64_t n_item_tiles = (h->n_items + 31) / 32;
    const int64_t wg_rows = (n + 127) / 128;  // 4 waves x 32 rows per workgroup
    // balanced persistent decomposition (see rank_fused_kernel): one range of (row block, item tile) work per
    // resident workgroup, at least 16 tiles long
    const int wgs_per_cu = (h->ld <= 64 || topk <= 10) ? 2 : 1;
    const int64_t work_total = wg_rows * n_item_tiles;
    int64_t work_per_wg = std::max<int64_t>((work_total + (int64_t)di.cus * wgs_per_cu - 1) / ((int64_t)di.cus * wgs_per_cu),
                                            std::min<int64_t>(16, n_item_tiles));
    const int64_t n_wgs = (work_total + work_per_wg - 1) / work_per_wg;
    const int64_t max_segs = (n_item_tiles + work_per_wg - 1) / work_per_wg + 1;
    h->part.ensure((size_t)(max_segs * n * topk));
    HIP_CHECK(hipMemsetAsync(h->part.p, 0, (size_t)(max_segs * n * topk) * sizeof(unsigned long long), h->stream));
    h->tau_pub.ensure((size_t)n);
    HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)h->tau_pub.p, 0xff800000u /* -inf */, (size_t)n, h->stream));
    // exclusion lists -> bitmap in rank order (the kernel then masks excluded items out of its survivor masks with
    // scalar operations; without it they flood the candidate buffers — a trained model scores a user's training
    // positives highest — and are only dropped at compaction)
    const uint32_t *d_bits = nullptr;
    const bool no_bitmap = prof_env_set("CORNAC_HIP_RANK_NO_BITMAP");  // A/B switch, profile builds only
    if (d_excl_indptr && n_item_tiles <= kMaxBitmapTiles && !no_bitmap) {
        const int64_t rows_padded = wg_rows * 128;  // whole workgroups of the top-k kernel: it reads every wave's words
        h->excl_bits.ensure((size_t)(rows_padded * n_item_tiles));
        hipLaunchKernelGGL(excl_bitmap_kernel, dim3((unsigned)((rows_padded + kBitmapRows * (kBlk / 64) - 1) / (kBitmapRows * (kBlk / 64)))), dim3(kBlk), 0,
                           h->stream, d_excl_indptr + (excl_by_user ? 0 : excl_row0), d_excl_indices, d_users, u0,
                           excl_by_user ? 1 : 0, h->inv_perm.p, n, rows_padded, n_item_tiles, h->excl_bits.p);
        d_bits = h->excl_bits.p;
    } else if (d_excl_indptr) {
        REQUIRE(!excl_by_user, "resident exclusion lists need the bitmap path (catalogue too large)");
    }
    dim3 grid((unsigned)n_wgs), block(kBlk);
#define FUSED(KT_, CAP_) do {                                                                                    \
    if (prof_env_set("CORNAC_HIP_RANK_ABLATE")) {                                                                 \
        int occ = 0;                                                                                              \
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, rank_fused_kernel<KT_, CAP_, true>, kBlk, 0);     \
        fprintf(stderr, "[rank_fused<%d,%d>] %u workgroups x %lld block-tiles, <= %lld segments/row, %d workgroups/CU\n", \
                KT_, CAP_, grid.x, (long long)work_per_wg, (long long)max_segs, occ);                              \
    }                                                                                                             \
    if (ub)                                                                                                       \
        hipLaunchKernelGGL((rank_fused_kernel<KT_, CAP_, true>), grid, block, 0, h->stream, h->U.p, h->Vr.p,       \
                           h->ibr.p, ub, d_users, u0, n, h->n_items, work_per_wg, topk, d_bits ? nullptr : d_excl_indptr, \
                           d_excl_indices, excl_row0, d_bits, h->perm.p, h->tau_pub.p, h->part.p, ablate);       \
    else                                                                                                          \
        hipLaunchKernelGGL((rank_fused_kernel<KT_, CAP_, false>), grid, block, 0, h->stream, h->U.p, h->Vr.p,      \
                           h->ibr.p, ub, d_users, u0, n, h->n_items, work_per_wg, topk, d_bits ? nullptr : d_excl_indptr, \
                           d_excl_indices, excl_row0, d_bits, h->perm.p, h->tau_pub.p, h->part.p, ablate); } while (0)
    if (topk <= 10 && h->ld == 128) {
        FUSED(64, 42);  // k = 128: the smallest buffer (topk + 32) lets two workgroups share a CU's LDS
    } else if (topk <= 24) {  // CAP = 56: 24 slots of slack above the 32 a tile can add
        if (h->ld == 16) FUSED(8, 56);
        else if (h->ld == 32) FUSED(16, 56);
        else if (h->ld == 64) FUSED(32, 56);
        else FUSED(64, 56);
    } else {
        if (h->ld == 16) FUSED(8, 64);
        else if (h->ld == 32) FUSED(16, 64);
        else if (h->ld == 64) FUSED(32, 64);
        else FUSED(64, 64);
    }
#undef FUSED
    int pad = 1;
    while (pad < (int)max_segs * topk) pad <<= 1;
    hipLaunchKernelGGL(rank_merge_kernel, dim3((unsigned)n), dim3(64), (size_t)pad * 8, h->stream, h->part.p,
                       (int)max_segs, n, topk, pad, items_out, scores_out);
    HIP_CHECK(hipGetLastError());
}

// The exclusion bitmap costs rows x ceil(n_items / 32) x 4 bytes.  Rows are ranked in chunks so that it never exceeds
// kExclBitmapBudget (one chunk at the ML-20M shape: 138 493 users x 836 words = 463 MB; 65 536 users x 262 144 items
// would otherwise ask for 2 GiB, 1 M users x 200 k items for 25 GB).
constexpr int64_t kExclBitmapBudget = int64_t(512) << 20;
static void launch_rank_fused(cornac_hip_scorer_t h, const int32_t *d_users, int64_t u0, int64_t n, int topk,
                              const int64_t *d_excl_indptr, const int32_t *d_excl_indices, int64_t excl_row0,
                              int32_t *items_out, float *scores_out, bool excl_by_user = false) {
    const int64_t n_item_tiles = (h->n_items + 31) / 32;
    int64_t rows = n;
    if (d_excl_indptr && n_item_tiles <= kMaxBitmapTiles)
        rows = std::max<int64_t>(128, (kExclBitmapBudget / (n_item_tiles * 4)) / 128 * 128);
    for (int64_t r0 = 0; r0 < n; r0 += rows) {
        const int64_t m = std::min(rows, n - r0);
        launch_rank_fused_rows(h, d_users ? d_users + r0 : nullptr, u0 + r0, m, topk, d_excl_indptr, d_excl_indices,
                               excl_row0 + r0, items_out + r0 * topk, scores_out ? scores_out + r0 * topk : nullptr,
                               excl_by_user);
    }
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
        h->ld = k <= 16 ? 16 : k <= 32 ? 32 : k <= 64 ? 64 : k <= 128 ? 128 : k;
        HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->U.alloc((size_t)n_users * h->ld);
        h->V.alloc((size_t)n_items * h->ld);
        HIP_CHECK(hipMemsetAsync(h->U.p, 0, h->U.n * sizeof(float), h->stream));
        HIP_CHECK(hipMemsetAsync(h->V.p, 0, h->V.n * sizeof(float), h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
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
        if (h->pinned_out) (void)hipHostFree(h->pinned_out);
        for (void *q : h->pinned_retired) (void)hipHostFree(q);
        delete h;
    });
}

int cornac_hip_scorer_set(cornac_hip_scorer_t h, const float *U, const float *V, const float *item_base,
                          const float *user_base) {
    return guarded([&] {
        sc_check(h, false);
        REQUIRE(U && V, "U and V are required");
        HIP_CHECK(hipMemcpy2DAsync(h->U.p, (size_t)h->ld * 4, U, (size_t)h->k * 4, (size_t)h->k * 4, (size_t)h->n_users,
                                   hipMemcpyHostToDevice, h->stream));
        HIP_CHECK(hipMemcpy2DAsync(h->V.p, (size_t)h->ld * 4, V, (size_t)h->k * 4, (size_t)h->k * 4, (size_t)h->n_items,
                                   hipMemcpyHostToDevice, h->stream));
        if (item_base) h->item_base.upload(item_base, (size_t)h->n_items, h->stream);
        else HIP_CHECK(hipMemsetAsync(h->item_base.p, 0, (size_t)h->n_items * 4, h->stream));
        h->has_user_base = user_base != nullptr;
        if (user_base) h->user_base.upload(user_base, (size_t)h->n_users, h->stream);
        build_rank_order(h, U, V, item_base);
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

namespace chip {
// fast_dot's float64 variant (cornac/utils/fast_dot.pyx:25-43: ddot per item row): one thread per item, the products
// summed in index order in double
__global__ __launch_bounds__(256) void score_user_f64_kernel(const double *__restrict__ u, const double *__restrict__ V,
                                                             const double *__restrict__ item_base, double user_base,
                                                             int64_t n_items, int k, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_items) return;
    double acc = 0.0;
    const double *v = V + i * k;
    for (int f = 0; f < k; ++f) acc = acc + u[f] * v[f];
    out[i] = (item_base ? item_base[i] : 0.0) + user_base + acc;
}
}  // namespace chip

int cornac_hip_scorer_set_f64(cornac_hip_scorer_t h, const double *U, const double *V, const double *item_base,
                              const double *user_base) {
    return guarded([&] {
        sc_check(h, false);
        REQUIRE(U && V, "U and V are required");
        h->U64.ensure((size_t)h->n_users * h->k);
        h->V64.ensure((size_t)h->n_items * h->k);
        h->U64.upload(U, (size_t)h->n_users * h->k, h->stream);
        h->V64.upload(V, (size_t)h->n_items * h->k, h->stream);
        h->ib64.ensure((size_t)h->n_items);
        if (item_base) h->ib64.upload(item_base, (size_t)h->n_items, h->stream);
        else HIP_CHECK(hipMemsetAsync(h->ib64.p, 0, (size_t)h->n_items * sizeof(double), h->stream));
        h->f64_user_base = user_base != nullptr;
        if (user_base) {
            h->ub64.ensure((size_t)h->n_users);
            h->ub64.upload(user_base, (size_t)h->n_users, h->stream);
        }
        h->scores64.ensure((size_t)h->n_items);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->f64_set = true;
    });
}

int cornac_hip_score_user_f64(cornac_hip_scorer_t h, int64_t user, double *out) {
    return guarded([&] {
        sc_check(h, false);
        REQUIRE(h->f64_set, "no float64 tables (cornac_hip_scorer_set_f64)");
        REQUIRE(user >= 0 && user < h->n_users, "user %lld out of range", (long long)user);
        REQUIRE(out != nullptr, "out is NULL");
        double ub = 0.0;
        if (h->f64_user_base) HIP_CHECK(hipMemcpyAsync(&ub, h->ub64.p + user, sizeof ub, hipMemcpyDeviceToHost, h->stream));
        if (h->f64_user_base) HIP_CHECK(hipStreamSynchronize(h->stream));
        hipLaunchKernelGGL(score_user_f64_kernel, dim3((unsigned)((h->n_items + 255) / 256)), dim3(256), 0, h->stream,
                           h->U64.p + (size_t)user * h->k, h->V64.p, h->ib64.p, ub, h->n_items, h->k, h->scores64.p);
        HIP_CHECK(hipGetLastError());
        h->scores64.download(out, (size_t)h->n_items, h->stream);
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
        if (have_excl) {
            validate_csr(excl_indptr, excl_indices, n, h->n_items, "excl");
            h->d_excl_indptr.ensure((size_t)n + 1);
            h->d_excl_indptr.upload(excl_indptr, (size_t)n + 1, h->stream);
            const int64_t ne = excl_indptr[n];
            h->d_excl_indices.ensure((size_t)std::max<int64_t>(ne, 1));
            if (ne > 0) h->d_excl_indices.upload(excl_indices, (size_t)ne, h->stream);
        }
        if (can_fuse(h, topk)) {
            // fused path: no score workspace, all users in one launch.  Catalogues too large for the exclusion bitmap
            // binary-search the lists at compaction: the rows must then be sorted
            if (have_excl && (h->n_items + 31) / 32 > kMaxBitmapTiles)
                for (int64_t b = 0; b < n; ++b)
                    for (int64_t p = excl_indptr[b] + 1; p < excl_indptr[b + 1]; ++p)
                        REQUIRE(excl_indices[p] > excl_indices[p - 1], "exclusion row %lld is not strictly sorted",
                                (long long)b);
            h->d_users.ensure((size_t)n);
            h->d_items_out.ensure((size_t)(n * topk));
            h->d_scores_out.ensure((size_t)(n * topk));
            h->d_users.upload(users, (size_t)n, h->stream);
            launch_rank_fused(h, h->d_users.p, 0, n, topk, have_excl ? h->d_excl_indptr.p : nullptr,
                              have_excl ? h->d_excl_indices.p : nullptr, 0, h->d_items_out.p, h->d_scores_out.p);
            h->d_items_out.download(items_out, (size_t)(n * topk), h->stream);
            h->d_scores_out.download(scores_out, (size_t)(n * topk), h->stream);
            HIP_CHECK(hipStreamSynchronize(h->stream));
            return;
        }
        int64_t cap = rows_per_batch(h);
        if (topk > TOPK_MAX) cap = std::min<int64_t>(cap, 1024);
        const int64_t nb_max = std::min(cap, n);
        h->scores.ensure((size_t)(nb_max * h->n_items));
        h->d_users.ensure((size_t)nb_max);
        h->d_items_out.ensure((size_t)(nb_max * topk));
        h->d_scores_out.ensure((size_t)(nb_max * topk));
        if (have_excl) h->excl.ensure((size_t)(nb_max * h->n_items));
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

int cornac_hip_rank_positions(cornac_hip_scorer_t h, const int32_t *users, int64_t n, const int64_t *excl_indptr,
                              const int32_t *excl_indices, const int64_t *tgt_indptr, const int32_t *tgt_indices,
                              int32_t *greater_out, int32_t *pos_out, int32_t *ge_out, float *tgt_scores_out) {
    return guarded([&] {
        sc_check(h);
        REQUIRE(users && tgt_indptr && n > 0, "bad arguments");
        REQUIRE(tgt_indptr[0] == 0, "tgt_indptr must start at 0");
        for (int64_t b = 0; b < n; ++b) {
            REQUIRE(users[b] >= 0 && users[b] < h->n_users, "user %d out of range", users[b]);
            REQUIRE(tgt_indptr[b + 1] >= tgt_indptr[b], "tgt_indptr must not decrease");
        }
        const int64_t nt = tgt_indptr[n];
        if (nt == 0) return;
        REQUIRE(tgt_indices && greater_out && pos_out && ge_out && tgt_scores_out, "bad arguments");
        const bool have_excl = excl_indptr != nullptr && excl_indices != nullptr;
        if (have_excl) {
            validate_csr(excl_indptr, excl_indices, n, h->n_items, "excl");
            h->d_excl_indptr.ensure((size_t)n + 1);
            h->d_excl_indptr.upload(excl_indptr, (size_t)n + 1, h->stream);
            const int64_t ne = excl_indptr[n];
            h->d_excl_indices.ensure((size_t)std::max<int64_t>(ne, 1));
            if (ne > 0) h->d_excl_indices.upload(excl_indices, (size_t)ne, h->stream);
        }
        h->d_tgt_indptr.ensure((size_t)n + 1);
        h->d_tgt_indptr.upload(tgt_indptr, (size_t)n + 1, h->stream);
        h->d_tgt_indices.ensure((size_t)nt);
        h->d_tgt_indices.upload(tgt_indices, (size_t)nt, h->stream);
        h->d_tgt_counts.ensure((size_t)(3 * nt));
        h->d_tgt_scores.ensure((size_t)nt);
        const int64_t cap = rows_per_batch(h);
        const int64_t nb_max = std::min(cap, n);
        h->scores.ensure((size_t)(nb_max * h->n_items));
        h->d_users.ensure((size_t)nb_max);
        if (have_excl) h->excl.ensure((size_t)(nb_max * h->n_items));
        for (int64_t b0 = 0; b0 < n; b0 += cap) {
            const int64_t nb = std::min(cap, n - b0);
            h->d_users.upload(users + b0, (size_t)nb, h->stream);
            launch_scores(h, h->d_users.p, 0, nb, true);
            if (have_excl) {
                HIP_CHECK(hipMemsetAsync(h->excl.p, 0, (size_t)(nb * h->n_items), h->stream));
                hipLaunchKernelGGL(mark_excluded_kernel, dim3((unsigned)nb), dim3(kBlk), 0, h->stream,
                                   h->d_excl_indptr.p, h->d_excl_indices.p, b0, h->n_items, h->excl.p);
            }
            hipLaunchKernelGGL(rank_positions_kernel, dim3((unsigned)nb), dim3(kBlk), 0, h->stream, h->scores.p,
                               have_excl ? h->excl.p : nullptr, h->n_items, h->d_tgt_indptr.p, h->d_tgt_indices.p, b0,
                               h->d_tgt_counts.p, h->d_tgt_counts.p + nt, h->d_tgt_counts.p + 2 * nt,
                               h->d_tgt_scores.p);
            HIP_CHECK(hipGetLastError());
            HIP_CHECK(hipStreamSynchronize(h->stream));  // d_users is re-used by the next batch's upload
        }
        HIP_CHECK(hipMemcpyAsync(greater_out, h->d_tgt_counts.p, (size_t)nt * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipMemcpyAsync(pos_out, h->d_tgt_counts.p + nt, (size_t)nt * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipMemcpyAsync(ge_out, h->d_tgt_counts.p + 2 * nt, (size_t)nt * 4, hipMemcpyDeviceToHost, h->stream));
        h->d_tgt_scores.download(tgt_scores_out, (size_t)nt, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}

int cornac_hip_score_pairs(cornac_hip_scorer_t h, const int32_t *users, const int32_t *items, int64_t n, int clip,
                           float lo, float hi, float *out) {
    return guarded([&] {
        sc_check(h);
        REQUIRE(users && items && out && n > 0, "bad arguments");
        for (int64_t p = 0; p < n; ++p)
            REQUIRE(users[p] >= 0 && users[p] < h->n_users && items[p] >= 0 && items[p] < h->n_items,
                    "pair %lld is out of range", (long long)p);
        DevBuf<int32_t> du, di;
        DevBuf<float> dout;
        du.alloc((size_t)n);
        di.alloc((size_t)n);
        dout.alloc((size_t)n);
        du.upload(users, (size_t)n, h->stream);
        di.upload(items, (size_t)n, h->stream);
        hipLaunchKernelGGL(score_pairs_kernel, dim3((unsigned)((n + kBlk - 1) / kBlk)), dim3(kBlk), 0, h->stream,
                           h->U.p, h->V.p, h->item_base.p, h->has_user_base ? h->user_base.p : nullptr, du.p, di.p, n,
                           h->k, h->ld, clip, lo, hi, dout.p);
        HIP_CHECK(hipGetLastError());
        dout.download(out, (size_t)n, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}

int cornac_hip_scorer_set_exclusions(cornac_hip_scorer_t h, const int64_t *indptr, const int32_t *indices) {
    return guarded([&] {
        sc_check(h, false);
        if (!indptr) {
            h->has_res_excl = false;
            return;
        }
        REQUIRE(indices != nullptr || indptr[h->n_users] == 0, "indices is NULL");
        validate_csr(indptr, indices, h->n_users, h->n_items, "excl");
        const int64_t ne = indptr[h->n_users];
        h->res_excl_indptr.ensure((size_t)h->n_users + 1);
        h->res_excl_indices.ensure((size_t)std::max<int64_t>(ne, 1));
        h->res_excl_indptr.upload(indptr, (size_t)h->n_users + 1, h->stream);
        if (ne > 0) h->res_excl_indices.upload(indices, (size_t)ne, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->has_res_excl = true;
    });
}

int cornac_hip_rank_topk_resident(cornac_hip_scorer_t h, const int32_t *users, int64_t u0, int64_t n, int topk,
                                  int32_t *items_out, float *scores_out, double *device_ms) {
    return guarded([&] {
        sc_check(h);
        REQUIRE(h->has_res_excl, "cornac_hip_scorer_set_exclusions has not been called");
        REQUIRE(n > 0 && topk >= 1 && topk <= h->n_items, "bad arguments");
        REQUIRE(can_fuse(h, topk) && (h->n_items + 31) / 32 <= kMaxBitmapTiles,
                "resident exclusion lists serve the fused top-k path (topk <= 32, k <= 128, <= 262144 items)");
        if (users) {
            for (int64_t b = 0; b < n; ++b) REQUIRE(users[b] >= 0 && users[b] < h->n_users, "user %d out of range", users[b]);
            h->d_users.ensure((size_t)n);
            h->d_users.upload(users, (size_t)n, h->stream);
        } else {
            REQUIRE(u0 >= 0 && u0 + n <= h->n_users, "user range out of bounds");
        }
        h->d_items_out.ensure((size_t)(n * topk));
        h->d_scores_out.ensure((size_t)(n * topk));
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (device_ms) {
            HIP_CHECK(hipEventCreate(&e0));
            HIP_CHECK(hipEventCreate(&e1));
            HIP_CHECK(hipEventRecord(e0, h->stream));
        }
        launch_rank_fused(h, users ? h->d_users.p : nullptr, u0, n, topk, h->res_excl_indptr.p, h->res_excl_indices.p, 0,
                          h->d_items_out.p, h->d_scores_out.p, true);
        if (device_ms) HIP_CHECK(hipEventRecord(e1, h->stream));
        if (items_out) h->d_items_out.download(items_out, (size_t)(n * topk), h->stream);
        if (scores_out) h->d_scores_out.download(scores_out, (size_t)(n * topk), h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        if (device_ms) {
            float t = 0.f;
            HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            *device_ms = (double)t;
        }
    });
}

int cornac_hip_scorer_host_buffer(cornac_hip_scorer_t h, size_t bytes, void **out) {
    return guarded([&] {
        sc_check(h);
        REQUIRE(out != nullptr, "out is NULL");
        if (bytes > h->pinned_bytes) {
            // an outgrown buffer may still be referenced by the caller (arrays viewing it): it is kept until the scorer
            // is destroyed; growth at least doubles, so the retired ones add up to less than the live one
            if (h->pinned_out) h->pinned_retired.push_back(h->pinned_out);
            h->pinned_out = nullptr;
            const size_t want = std::max(bytes, 2 * h->pinned_bytes);
            h->pinned_bytes = 0;
            HIP_CHECK(hipHostMalloc(&h->pinned_out, want, hipHostMallocDefault));
            h->pinned_bytes = want;
        }
        *out = h->pinned_out;
    });
}

int cornac_hip_rank_topk_device(cornac_hip_scorer_t h, int64_t u0, int64_t n, int topk, int repeats, double *ms) {
    return guarded([&] {
        sc_check(h);
        REQUIRE(u0 >= 0 && n > 0 && u0 + n <= h->n_users, "user range out of bounds");
        REQUIRE(topk >= 1 && topk <= h->n_items, "topk out of range");
        REQUIRE(repeats >= 1 && ms, "bad arguments");
        const bool fused = can_fuse(h, topk);
        const int64_t cap = fused ? n : rows_per_batch(h);
        const int64_t nb_max = std::min(cap, n);
        if (!fused) h->scores.ensure((size_t)(nb_max * h->n_items));
        if (!fused && topk > TOPK_MAX) {  // the sort scratch is (re)allocated here, not between the timing events
            int64_t pad = 1;
            while (pad < h->n_items) pad <<= 1;
            h->sort_scratch.ensure((size_t)(nb_max * pad));
        }
        h->d_items_out.ensure((size_t)(nb_max * topk));
        h->d_scores_out.ensure((size_t)(nb_max * topk));
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, h->stream));
        for (int r = 0; r < repeats; ++r) {
            for (int64_t b0 = 0; b0 < n; b0 += cap) {
                const int64_t nb = std::min(cap, n - b0);
                if (fused) {
                    launch_rank_fused(h, nullptr, u0 + b0, nb, topk, nullptr, nullptr, 0, h->d_items_out.p,
                                      h->d_scores_out.p);
                } else {
                    launch_scores(h, nullptr, u0 + b0, nb, true);
                    launch_rank(h, nb, topk, false, h->d_items_out.p, h->d_scores_out.p);
                }
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
