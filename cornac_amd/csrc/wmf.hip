// WMF (weighted matrix factorisation) training step on gfx950.
//
// Reference: cornac/models/wmf/wmf.py:34-55 (graph) and cornac/models/wmf/recom_wmf.py:160-207 (loop):
// per batch of <= 128 items,   P = U V_b^T  over ALL users,  loss = sum C (R_b - P)^2 + l2 terms,
// gradients clipped to [-5, 5], TF1 Adam on U (dense) and V (IndexedSlices: every row's moments decay,
// every row moves).  The three GEMM-shaped pieces run on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32) through one LDS-tiled block routine with fused epilogues:
//
//   wmf_pred_kernel    G  = 2 b (U V_b^T)                   [n_users x B]   (+ b sum P^2 into the loss)
//   wmf_fixup_kernel   G[u,c] = 2 a (U_u . V_c - r)  at the batch's non-zeros (CSC columns), loss fix-up
//   wmf_grad_v_kernel  dV = G^T U   split over user chunks  [B x k]         (fp32 atomics into 64 KB)
//   wmf_update_u_kernel  dU = G V_b, g = clip(dU + lambda_u U), Adam on U, m_U, v_U   (fused epilogue)
//   wmf_update_v_*     g = clip(dV + lambda_v V_b) scattered into a dense gradient, dense Adam over V
//
// The dense R_b / C of the reference (n_users x B each per step) are never materialised.
#include <algorithm>
#include <cmath>

#include "common.h"
#include "mfma_gemm.h"

namespace chip {

constexpr int kMaxBatch = 128;  // items per step (the reference's default batch_size); one N tile

__device__ __forceinline__ double block_sum_f64(double v, double *scratch) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    double s = 0;
    for (int w = 0; w < kWb / 64; ++w) s += scratch[w];
    return s;
}

// V_b = V[ids]  (contiguous [B x ld]);  also lambda_v/2 * |V_b|^2 into the loss
__global__ __launch_bounds__(kWb) void wmf_gather_kernel(const float *__restrict__ V, const int32_t *__restrict__ ids,
                                                         int B, int ld, float *__restrict__ Vb,
                                                         float *__restrict__ VbT, float half_lambda_v, double *loss,
                                                         uint32_t *__restrict__ slot_tag, uint32_t step) {
    __shared__ double scratch[kWb / 64];
    double part = 0;
    const int64_t n = (int64_t)B * ld;
    for (int64_t e = (int64_t)blockIdx.x * kWb + threadIdx.x; e < n; e += (int64_t)gridDim.x * kWb) {
        const int c = (int)(e / ld), f = (int)(e % ld);
        if (f == 0) slot_tag[ids[c]] = (step << 7) | (uint32_t)c;   // this step's batch column of the row (wmf_adam_v_kernel)
        const float v = V[(int64_t)ids[c] * ld + f];
        Vb[e] = v;
        if (VbT) VbT[(int64_t)f * kMaxBatch + c] = v;  // [ld][128]: the n-contiguous B operand of P = U V_b^T
        part += (double)v * v;
    }
    const double s = block_sum_f64(part, scratch);
    if (threadIdx.x == 0 && s != 0) atomicAdd(loss, (double)half_lambda_v * s);
}

// G[u, c] = 2 b P[u, c];   loss += b sum P^2     (the C = b, R = 0 case for every cell)
__global__ __launch_bounds__(kWb) void wmf_pred_kernel(const float *__restrict__ U, const float *__restrict__ Vb,
                                                       int64_t n_users, int B, int k, int ld,
                                                       float *__restrict__ G, float b, double *loss) {
    __shared__ GemmSmem sm;
    __shared__ double scratch[kWb / 64];
    f32x16 acc[2][2];
    const int64_t m0 = (int64_t)blockIdx.x * kBM;
    gemm_block<true, false>(U, ld, 1, Vb, 1, ld, n_users, B, m0, 0, 0, k, sm, acc);
    double part = 0;
    const float two_b = 2.f * b;
    for_each_acc(acc, m0, 0, [&](int64_t row, int64_t col, float p) {
        if (row < n_users && col < kMaxBatch) {
            const bool live = col < B;
            G[row * kMaxBatch + col] = live ? two_b * p : 0.f;
            if (live) part += (double)p * p;
        }
    });
    const double s = block_sum_f64(part, scratch);
    if (threadIdx.x == 0) atomicAdd(loss, (double)b * s);
}

// non-zeros of the batch's columns: G[u, c] = 2 a (p - r), loss += a (r - p)^2 - b p^2.
// One 16-lane group per non-zero; grid.y = column of the batch.
__global__ __launch_bounds__(kWb) void wmf_fixup_kernel(const float *__restrict__ U, const float *__restrict__ Vb,
                                                        const int64_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ rows,
                                                        const float *__restrict__ vals,
                                                        const int32_t *__restrict__ ids, int k, int ld,
                                                        float *__restrict__ G, float a, float b, double *loss) {
    __shared__ double scratch[kWb / 64];
    const int c = blockIdx.y;
    const int64_t beg = indptr[ids[c]], end = indptr[ids[c] + 1];
    const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15;
    const float *vrow = Vb + (int64_t)c * ld;
    double part = 0;
    for (int64_t e = beg + (int64_t)blockIdx.x * (kWb / 16) + grp; e < end; e += (int64_t)gridDim.x * (kWb / 16)) {
        const float r = vals[e];
        const int64_t u = rows[e];
        const float *urow = U + u * ld;
        float p = 0.f;
        for (int f = gl * 4; f < k; f += 64) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(urow + f), y = *reinterpret_cast<const f32x4 *>(vrow + f);
            p += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) p += __shfl_xor(p, o, 16);
        if (gl == 0 && r != 0.f) {  // explicit zeros stay "unobserved" (batch_R.nonzero(), recom_wmf.py:186)
            G[u * kMaxBatch + c] = 2.f * a * (p - r);
            part += (double)a * (double)(r - p) * (double)(r - p) - (double)b * (double)p * (double)p;
        }
    }
    const double s = block_sum_f64(part, scratch);
    if (threadIdx.x == 0 && s != 0) atomicAdd(loss, s);
}

// dV[c, f] += sum_{u in chunk} G[u, c] U[u, f]
__global__ __launch_bounds__(kWb) void wmf_grad_v_kernel(const float *__restrict__ G, const float *__restrict__ U,
                                                         int64_t n_users, int B, int ld, int64_t chunk,
                                                         float *__restrict__ dV) {
    __shared__ GemmSmem sm;
    f32x16 acc[2][2];
    const int64_t n0 = (int64_t)blockIdx.y * kBN;
    const int64_t k_begin = (int64_t)blockIdx.x * chunk;
    const int64_t k_end = k_begin + chunk < n_users ? k_begin + chunk : n_users;
    gemm_block<false, true>(G, 1, kMaxBatch, U, ld, 1, B, ld, 0, n0, k_begin, k_end, sm, acc);
    for_each_acc(acc, 0, n0, [&](int64_t row, int64_t col, float v) {
        if (row < B && col < ld && v != 0.f) atomicAdd(dV + row * ld + col, v);
    });
}

struct TfAdam {
    float beta1, beta2, one_minus_beta1, one_minus_beta2, lr_t, eps;
};

// dU = G V_b; g = clip(dU + lambda_u U, -5, 5); dense TF1 Adam on the tile; loss += lambda_u/2 |U|^2 (pre-update)
__global__ __launch_bounds__(kWb) void wmf_update_u_kernel(const float *__restrict__ G, const float *__restrict__ Vb,
                                                           int64_t n_users, int B, int k, int ld,
                                                           float *__restrict__ U, float *__restrict__ mU,
                                                           float *__restrict__ vU, float lambda_u, const TfAdam ad,
                                                           double *loss) {
    __shared__ GemmSmem sm;
    __shared__ double scratch[kWb / 64];
    f32x16 acc[2][2];
    const int64_t m0 = (int64_t)blockIdx.x * kBM, n0 = (int64_t)blockIdx.y * kBN;
    gemm_block<true, true>(G, kMaxBatch, 1, Vb, ld, 1, n_users, ld, m0, n0, 0, B, sm, acc);
    double part = 0;
    for_each_acc(acc, m0, n0, [&](int64_t row, int64_t col, float du) {
        if (row < n_users && col < k) {
            const int64_t o = row * ld + col;
            const float u = U[o];
            part += (double)u * u;
            float g = du + lambda_u * u;
            g = fminf(fmaxf(g, -5.f), 5.f);
            float m = mU[o], v = vU[o];
            m = m + ad.one_minus_beta1 * (g - m);
            v = v + ad.one_minus_beta2 * (g * g - v);
            mU[o] = m;
            vU[o] = v;
            U[o] = u - ad.lr_t * m / (sqrtf(v) + ad.eps);
        }
    });
    const double s = block_sum_f64(part, scratch);
    if (threadIdx.x == 0) atomicAdd(loss, 0.5 * (double)lambda_u * s);
}

// ---- the user side of one step in ONE kernel (k <= 128) ---------------------------------------------------------
// Per 128-user tile, in a persistent workgroup: P = U_t V_b^T (MFMA) -> G = 2 b P written to a 64 KB scratch tile of
// the workgroup (L2-resident: written and re-read by the same workgroup, never by another) -> the batch's non-zeros
// of this tile patched in place (CSC columns are sorted by row and a workgroup walks consecutive tiles: a cursor
// per column) -> dV += G_t^T U_t accumulated in registers ACROSS the workgroup's tiles -> dU = G_t V_b with the clipped
// TF1-Adam update of U, m_U, v_U as the epilogue.  The n_users x 128 matrix G of the unfused path (written once,
// patched, read twice: 4 x 246 MB per step at the Netflix user count) never exists, U is read once for the three
// products.  Each workgroup leaves its dV partial in `dv_part`; wmf_reduce_dv_kernel sums them.
__global__ __launch_bounds__(kWb, 2) void wmf_user_step_kernel(const float *__restrict__ Vb, const float *__restrict__ VbT,
                                                            int64_t n_users, int B, int k,
                                                            int ld, float *U, float *mU, float *vU, float *g_scratch,
                                                            const int64_t *__restrict__ indptr,
                                                            const int32_t *__restrict__ rows,
                                                            const float *__restrict__ vals,
                                                            const int32_t *__restrict__ ids, float a, float b,
                                                            float lambda_u, const TfAdam ad,
                                                            float *__restrict__ dv_part, double *loss, int ablate) {
    __shared__ GemmSmem sm;
    __shared__ double scratch[kWb / 64];
    float *Gs = g_scratch + (size_t)blockIdx.x * kBM * kMaxBatch;  // [128 users][128 batch columns]
    f32x16 acc[2][2], accv[2][2];  // accv: this workgroup's dV partial, accumulated across its tiles
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accv[i][j][r] = 0.f;
    double part = 0;  // loss terms of this thread
    const float two_b = 2.f * b;
    const int64_t n_tiles = (n_users + kBM - 1) / kBM;
    // a workgroup walks a CONTIGUOUS range of user tiles, so that a thread's cursor into its (row-sorted) CSC column
    // only ever moves forward by the few non-zeros of the current tile: one binary search per column and kernel
    const int64_t per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
    const int64_t tile_lo = (int64_t)blockIdx.x * per_wg, tile_hi = min(n_tiles, tile_lo + per_wg);
    int64_t cur = 0, col_end = 0;
    if ((int)threadIdx.x < B && tile_lo < tile_hi) {
        const int32_t item = ids[threadIdx.x];
        int64_t lo = indptr[item], hi = indptr[item + 1];
        col_end = hi;
        const int64_t key = tile_lo * kBM;
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if ((int64_t)rows[mid] < key) lo = mid + 1; else hi = mid;
        }
        cur = lo;
    }
    for (int64_t tile = tile_lo; tile < tile_hi; ++tile) {
        const int64_t m0 = tile * kBM;
        const int64_t rows_here = min((int64_t)kBM, n_users - m0);
        // (a) P = U_t V_b^T
        if (!(ablate & 1)) gemm_block<true, true>(U, ld, 1, VbT, kMaxBatch, 1, n_users, kMaxBatch, m0, 0, 0, k, sm, acc);  // (columns >= B of V_b^T are zero)
        // (b) G = 2 b P into the scratch tile (zero outside the live rows / columns); loss += b sum P^2
        float sq = 0.f;  // (64 squares of |P| <= a few units per thread and tile: fp32 partial, fp64 across tiles)
        const int nrow = (int)rows_here;
        if (!(ablate & 32)) for_each_acc_local(acc, kMaxBatch, [&](int off, int rl, int cl, float pv) {
            const bool live = rl < nrow && cl < B;
            Gs[off] = live ? two_b * pv : 0.f;
            if (live) sq += pv * pv;
        });
        part += (double)b * (double)sq;
        __syncthreads();
        // (c) the batch's non-zeros whose user is in this tile: G = 2 a (p - r), loss += a (r - p)^2 - b p^2
        if ((int)threadIdx.x < B && !(ablate & 4)) {
            const int c = threadIdx.x;
            const float *vrow = Vb + (int64_t)c * ld;
            const int64_t m1 = m0 + rows_here;
            while (cur < col_end) {
                const int64_t u = rows[cur];
                if (u >= m1) break;
                const float r = vals[cur];
                ++cur;
                if (r == 0.f) continue;  // explicit zeros stay "unobserved" (batch_R.nonzero(), recom_wmf.py:186)
                float *gp = Gs + (u - m0) * kMaxBatch + c;
                float pv;
                if (two_b != 0.f) {
                    pv = __builtin_nontemporal_load(gp) / two_b;  // the prediction the MFMA pass just produced
                } else {                                           // b == 0: G carries nothing to recover p from
                    const float *urow = U + u * ld;
                    pv = 0.f;
                    for (int f = 0; f < k; f += 4) {
                        const f32x4 x = *reinterpret_cast<const f32x4 *>(urow + f), y = *reinterpret_cast<const f32x4 *>(vrow + f);
                        pv += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
                    }
                }
                *gp = 2.f * a * (pv - r);
                part += (double)a * (double)(r - pv) * (double)(r - pv) - (double)b * (double)pv * (double)pv;
            }
        }
        __syncthreads();
        // (e) dV += G_t^T U_t  (U before its update): M = batch columns, N = ld, K = the tile's users
        if (!(ablate & 1)) gemm_block<false, true, false, true>(Gs, 1, kMaxBatch, U + m0 * ld, ld, 1, B, ld, 0, 0, 0, rows_here, sm, accv);
        // (d) dU = G_t V_b, then g = clip(dU + lambda_u U), TF1 Adam on U, m_U, v_U; loss += lambda_u/2 |U|^2 (pre-update)
        if (!(ablate & 8)) gemm_block<true, true, true, true>(Gs, kMaxBatch, 1, Vb, ld, 1, rows_here, ld, 0, 0, 0, B, sm, acc);
        // The accumulators hold 32 consecutive columns per half-wave and row: staged through LDS (64 rows at a time,
        // the GEMM tiles' 33 KB are free here) so that every thread updates float4s of U, m_U, v_U — 16-byte
        // accesses, whole 512-byte rows per half-wave — instead of 3 x 64 dword loads and stores
        if (!(ablate & 16)) {
            float *stg = &sm.a[0][0][0];  // [64][128]
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const int wm = wave >> 1;
            (void)lane;
            float usq = 0.f;
            for (int h = 0; h < 2; ++h) {
                if (wm == h) for_each_acc_local(acc, kBN, [&](int off, int, int, float v) { stg[off - h * 64 * kBN] = v; });
                __syncthreads();
                const int c4 = (threadIdx.x & 31) * 4;
                if (c4 < ld) {
#pragma unroll 1
                    for (int q = 0; q < 8; ++q) {
                        const int row = (threadIdx.x >> 5) + 8 * q;
                        const int64_t gr = m0 + h * 64 + row;
                        if (gr < n_users) {
                            const f32x4 du = *reinterpret_cast<const f32x4 *>(stg + row * kBN + c4);
                            f32x4 *pu = reinterpret_cast<f32x4 *>(U + gr * ld + c4);
                            f32x4 *pm = reinterpret_cast<f32x4 *>(mU + gr * ld + c4);
                            f32x4 *pv = reinterpret_cast<f32x4 *>(vU + gr * ld + c4);
                            f32x4 u = *pu, m = *pm, v = *pv;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                usq += u[e] * u[e];
                                float g = du[e] + lambda_u * u[e];
                                g = fminf(fmaxf(g, -5.f), 5.f);
                                m[e] = m[e] + ad.one_minus_beta1 * (g - m[e]);
                                v[e] = v[e] + ad.one_minus_beta2 * (g * g - v[e]);
                                u[e] = u[e] - ad.lr_t * m[e] / (sqrtf(v[e]) + ad.eps);
                            }
                            *pm = m;
                            *pv = v;
                            *pu = u;
                        }
                    }
                }
                __syncthreads();
            }
            part += 0.5 * (double)lambda_u * (double)usq;
        }
        __syncthreads();  // the scratch tile is rewritten by the next tile
    }
    float *mine = dv_part + (size_t)blockIdx.x * kMaxBatch * kBN;
    for_each_acc_local(accv, kBN, [&](int off, int, int, float v) { mine[off] = v; });
    const double s = block_sum_f64(part, scratch);
    if (threadIdx.x == 0 && s != 0) atomicAdd(loss, s);
}

// ---- the same step with the G tile resident in the LDS ------------------------------------------------------------
// Counters of the kernel above (profiles/r02_legs_pmc.csv): 3.05 GB of fabric traffic per step against the 1.48 GB of
// U, m_U, v_U read + written — the 64 KB scratch tiles do not stay in the L2 beside the streamed tables (0.25 GB written,
// 0.49 GB re-read per step) and U is fetched again for every product.  Here G lives in the LDS for the whole tile:
//   LDS (80 KB per workgroup, two workgroups per CU):  Gl[128][128] floats, column index XOR-swizzled with the row so
//   that BOTH products read it without bank conflicts — dV = G^T U walks it by rows (k = user), dU = G V_b by columns
//   (k = batch column) — and a 16 KB double buffer for the other operand's k-tiles.  The P = U_t V_b^T product runs
//   first with its two staging tiles laid over the (not yet live) G region; the Adam epilogue stages dU through it
//   once G is dead.
constexpr int kGl = kBM * kMaxBatch;                        // floats of the G tile
constexpr size_t kWmfLdsBytes = (size_t)(kGl + 2 * kBK * kBN) * sizeof(float);   // 81 920 B = half of a CU's LDS

__device__ __forceinline__ int gl_index(int row, int col) { return row * kMaxBatch + (col ^ (row & 31)); }

// acc (+)= A B over k in [0, k_end): A comes from the LDS-resident G tile — K_IS_ROW: A(m, kk) = G[kk][m] (dV: k = user,
// m = batch column), else A(m, kk) = G[m][kk] (dU: m = user, k = batch column); B(kk, n) = Bp[kk * b_sk + n], n < N
// contiguous, staged through the double buffer sb[2][kBK][kBN].
template <bool K_IS_ROW, bool ZERO>
__device__ __forceinline__ void gemm_lds_a(const float *Gl, const float *__restrict__ Bp, int64_t b_sk, int N, int k_end,
                                           float *sb, f32x16 (&acc)[2][2], bool skip_loads = false) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // (opaque: see gemm_block)
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    if (ZERO) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    // (a second k-tile in flight in registers — 2-ahead prefetch — costs 8 spilled registers here and measured slower:
    // 0.79 vs 0.77 ms per step)
    f32x4 rb[2];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kk = k0 + (tid >> 5) + 8 * i, n = (tid & 31) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kk < k_end) {
                const float *q = Bp + (int64_t)kk * b_sk + n;
                if (n + 3 < N && ((b_sk | n) & 3) == 0) {
                    v = *reinterpret_cast<const f32x4 *>(q);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < N) v[e] = q[e];
                }
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<f32x4 *>(sb + (buf * kBK + (tid >> 5) + 8 * i) * kBN + (tid & 31) * 4) = rb[i];
    };
    const int n_steps = (k_end + kBK - 1) / kBK;
    if (n_steps <= 0) return;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int m_a = wm * 64 + l31;
    for (int s = 0; s < n_steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < n_steps && !skip_loads) load_tile((s + 1) * kBK);
        // (fragments of k pair t + 1 requested before the MFMAs of pair t issue, as in gemm_block)
        float fa[2][2], fb[2][2];
        auto frag = [&](int t, int slot) {
            const int kk = s * kBK + 2 * t + half;
            if (K_IS_ROW) {
                fa[slot][0] = Gl[kk * kMaxBatch + (m_a ^ (kk & 31))];
                fa[slot][1] = Gl[kk * kMaxBatch + ((m_a + 32) ^ (kk & 31))];
            } else {
                fa[slot][0] = Gl[m_a * kMaxBatch + (kk ^ (m_a & 31))];
                fa[slot][1] = Gl[(m_a + 32) * kMaxBatch + (kk ^ (m_a & 31))];
            }
            fb[slot][0] = sb[(buf * kBK + 2 * t + half) * kBN + wn * 64 + l31];
            fb[slot][1] = sb[(buf * kBK + 2 * t + half) * kBN + wn * 64 + 32 + l31];
        };
        frag(0, 0);
#pragma unroll
        for (int t = 0; t < kBK / 2; ++t) {
            const int c = t & 1;
            if (t + 1 < kBK / 2) frag(t + 1, c ^ 1);
            __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the reads below the MFMAs to save registers)
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][1], acc[1][1], 0, 0, 0);
        }
        if (s + 1 < n_steps) store_tile(buf ^ 1);
        __syncthreads();
    }
}

// ---- full tiles (k = ld = 128, a batch of 128 items, 128 users): the three products with every address a per-thread constant ----
// The k-step loops above pay 100-200 VALU instructions per 32 MFMAs for 64-bit address arithmetic and range checks, and on this
// part VALU issue time adds to matrix-pipe time (tools/mfma_probe.hip).  With the extents known the 8 k-steps unroll: operand
// rows come off a scalar base that advances per step plus one 32-bit per-thread offset, LDS stores and fragment reads take
// immediate offsets; the swizzled reads of G need 16 per-thread column offsets (`xv4`: ((lane & 31) ^ half ^ 2 j) * 4 bytes).
//   MODE 0:  P  = U_t VbT    A(m, kk) = U_t[m][kk] staged k-major through `sa` (over the G region), B = VbT rows
//   MODE 1:  dV += G^T U_t   A(m, kk) = G[kk][m],  B = U_t rows
//   MODE 2:  dU = G Vb       A(m, kk) = G[m][kk],  B = Vb rows
template <int MODE, bool ZERO>
__device__ __forceinline__ void product_full(const float *Gl, float *sa, float *sb, const float *__restrict__ Arows,
                                             const float *__restrict__ Brows, const int (&xv4)[16], f32x16 (&acc)[2][2],
                                             bool skip_loads = false) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // (opaque: see gemm_block)
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    if (ZERO) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    const uint32_t off_b = (uint32_t)(((tid >> 5) * 128 + (tid & 31) * 4) * 4);   // bytes: row tid / 32 (+ 8 i), 4 floats
    const uint32_t off_a = (uint32_t)(((tid >> 2) * 128 + (tid & 3) * 4) * 4);    // row tid / 4 (+ 64 i), k (tid & 3) * 4
    const char *bb = reinterpret_cast<const char *>(Brows), *ab = reinterpret_cast<const char *>(Arows);
    f32x4 rb[2], ra[2];
    auto load_tile = [&](int s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rb[i] = *reinterpret_cast<const f32x4 *>(bb + (size_t)s * (kBK * 512) + (size_t)i * (8 * 512) + off_b);
            if (MODE == 0) ra[i] = *reinterpret_cast<const f32x4 *>(ab + (size_t)i * (64 * 512) + (size_t)s * (kBK * 4) + off_a);
        }
    };
    float *sb_w = sb + (tid >> 5) * kBN + (tid & 31) * 4;
    float *sa_w = sa + ((tid & 3) * 4) * kLdT + (tid >> 2);
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4 *>(sb_w + (buf * kBK + 8 * i) * kBN) = rb[i];
            if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) sa_w[(buf * kBK + q) * kLdT + 64 * i] = ra[i][q];
            }
        }
    };
    const float *sb_r = sb + half * kBN + wn * 64 + l31;
    const float *sa_r = sa + half * kLdT + wm * 64 + l31;
    const char *g_r = reinterpret_cast<const char *>(Gl) +
                      (MODE == 1 ? (half * kMaxBatch + wm * 64) * 4 : (wm * 64 + l31) * kMaxBatch * 4);
    load_tile(0);
    store_tile(0);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kBM / kBK; ++s) {
        const int buf = s & 1;
        if (s + 1 < kBM / kBK && !skip_loads) load_tile(s + 1);   // (skip_loads: profile builds' latency ablation)
        float fa[2][2], fb[2][2];
        auto frag = [&](int t, int slot) {
            if (MODE == 0) {
                fa[slot][0] = sa_r[(buf * kBK + 2 * t) * kLdT];
                fa[slot][1] = sa_r[(buf * kBK + 2 * t) * kLdT + 32];
            } else if (MODE == 1) {   // G[kk][m ^ (kk & 31)], kk = 16 s + 2 t + half
                const char *q = g_r + (kBK * s + 2 * t) * kMaxBatch * 4 + xv4[8 * (s & 1) + t];
                fa[slot][0] = *reinterpret_cast<const float *>(q);
                fa[slot][1] = *reinterpret_cast<const float *>(q + 32 * 4);
            } else {                  // G[m][kk ^ (m & 31)]
                const char *q = g_r + (s >> 1) * 32 * 4 + xv4[8 * (s & 1) + t];
                fa[slot][0] = *reinterpret_cast<const float *>(q);
                fa[slot][1] = *reinterpret_cast<const float *>(q + 32 * kMaxBatch * 4);
            }
            fb[slot][0] = sb_r[(buf * kBK + 2 * t) * kBN];
            fb[slot][1] = sb_r[(buf * kBK + 2 * t) * kBN + 32];
        };
        frag(0, 0);
#pragma unroll
        for (int t = 0; t < kBK / 2; ++t) {
            const int c = t & 1;
            if (t + 1 < kBK / 2) frag(t + 1, c ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][1], acc[1][1], 0, 0, 0);
        }
        if (s + 1 < kBM / kBK) store_tile(buf ^ 1);
        __syncthreads();
    }
}

// VAR (bits): 1 / 2 — the Adam epilogue keeps 2 / 4 row groups of U, m_U, v_U in flight per thread; 4 — m_U, v_U and the
// stores of U bypass the caches' retention (touched once per step: the L2 keeps the U tile the three products re-read);
// 8 — the second workgroup of a CU starts half a tile late, so that one streams the Adam state while the other multiplies;
// 16 — without the full-tile product routine (product_full)
template <int VAR>
__global__ __launch_bounds__(kWb, 2) void wmf_user_step_lds_kernel(const float *__restrict__ Vb, const float *__restrict__ VbT,
                                                                   int64_t n_users, int B, int k, int ld, float *U, float *mU,
                                                                   float *vU, const int64_t *__restrict__ indptr,
                                                                   const int32_t *__restrict__ rows,
                                                                   const float *__restrict__ vals,
                                                                   const int32_t *__restrict__ ids, float a, float b,
                                                                   float lambda_u, const TfAdam ad,
                                                                   float *__restrict__ dv_part, double *loss, int ablate,
                                                                   int stagger_ticks, int *cu_arrivals) {
#ifndef CORNAC_PROFILE
    ablate = 0;   // (profile builds: CORNAC_HIP_WMF_ABLATE bit 0 skips the P product, 1 the dV product, 2 the dU product, 3 the Adam
                  // epilogue, 4 the non-zeros' fix-up — the time each piece costs; results are garbage)
#endif
#ifdef CORNAC_PROFILE
    const long long prof_c0 = (long long)clock64(), prof_w0 = (long long)wall_clock64();
    // phase stamps (100 MHz ticks since the kernel's start) of workgroups 0 and gridDim.x - 1: 6 per tile, 16 tiles at most
    long long *prof_st = (cu_arrivals && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))
                             ? reinterpret_cast<long long *>(cu_arrivals + 1040) + (blockIdx.x == 0 ? 0 : 100) : nullptr;
    int prof_n = 0;
#define WMF_STAMP() do { if (prof_st && prof_n < 96) prof_st[prof_n++] = (long long)wall_clock64() - prof_w0; } while (0)
#else
#define WMF_STAMP() do { } while (0)
#endif
    extern __shared__ float lds[];
    float *Gl = lds;                  // [128][128] swizzled; also the P product's staging and the epilogue's stage
    float *sb = lds + kGl;            // [2][kBK][kBN]
    GemmSmem &sm = *reinterpret_cast<GemmSmem *>(lds);
    static_assert(sizeof(GemmSmem) <= kGl * sizeof(float), "the P product's staging tiles must fit the G region");
    f32x16 acc[2][2], accv[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accv[i][j][r] = 0.f;
    double part = 0;
    const float two_b = 2.f * b;
    const int64_t n_tiles = (n_users + kBM - 1) / kBM;
    const int64_t per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
    const int64_t tile_lo = (int64_t)blockIdx.x * per_wg, tile_hi = min(n_tiles, tile_lo + per_wg);
    int64_t cur = 0, col_end = 0;
    if ((int)threadIdx.x < B && tile_lo < tile_hi) {
        const int32_t item = ids[threadIdx.x];
        int64_t lo = indptr[item], hi = indptr[item + 1];
        col_end = hi;
        const int64_t key = tile_lo * kBM;
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            if ((int64_t)rows[mid] < key) lo = mid + 1; else hi = mid;
        }
        cur = lo;
    }
    if (VAR & 8) {
        // the second workgroup to arrive on a CU (a counter per physical CU, zeroed before the launch) waits
        // `stagger_ticks` of the 100 MHz clock
        int *late = reinterpret_cast<int *>(sb);
        if (threadIdx.x == 0) *late = atomicAdd(cu_arrivals + __smid(), 1) & 1;
        __syncthreads();
        const bool wait = *late != 0;
        __syncthreads();
        if (wait) {
            if (threadIdx.x == 0) atomicAdd(cu_arrivals + 1024, 1);
            const uint64_t t0 = wall_clock64();
            while (wall_clock64() - t0 < (uint64_t)stagger_ticks) __builtin_amdgcn_s_sleep(32);
        }
    }
    int xv4[16];   // product_full: byte offsets of the swizzled G columns this lane reads
    {
        const int x = (threadIdx.x & 31) ^ ((threadIdx.x >> 5) & 1);
#pragma unroll
        for (int j = 0; j < 16; ++j) xv4[j] = (x ^ (2 * j)) * 4;
    }
    const bool full_shape = !(VAR & 16) && ld == kBN && k == kBN && B == kMaxBatch;
    for (int64_t tile = tile_lo; tile < tile_hi; ++tile) {
        const int64_t m0 = tile * kBM;
        const int nrow = (int)min((int64_t)kBM, n_users - m0);
        const bool full = full_shape && nrow == kBM;
        WMF_STAMP();
        // (a) P = U_t V_b^T  (staging tiles over the G region; the trailing barrier of the product separates their last
        // read from the G stores below)
        if (!(ablate & 1)) {
            if (full) product_full<0, true>(Gl, &sm.a[0][0][0], sb, U + m0 * kBN, VbT, xv4, acc, (ablate & 64) != 0);
            else gemm_block<true, true>(U, ld, 1, VbT, kMaxBatch, 1, n_users, kMaxBatch, m0, 0, 0, k, sm, acc, (ablate & 64) != 0);
        }
        WMF_STAMP();
        // (b) G = 2 b P (zero outside the live rows / columns); loss += b sum P^2
        float sq = 0.f;
        for_each_acc_local(acc, kMaxBatch, [&](int, int rl, int cl, float pv) {
            const bool live = rl < nrow && cl < B;
            Gl[gl_index(rl, cl)] = live ? two_b * pv : 0.f;
            if (live) sq += pv * pv;
        });
        part += (double)b * (double)sq;
        __syncthreads();
        // (c) the batch's non-zeros whose user is in this tile: G = 2 a (p - r), loss += a (r - p)^2 - b p^2
        if ((int)threadIdx.x < B && !(ablate & 16)) {
            const int c = threadIdx.x;
            const float *vrow = Vb + (int64_t)c * ld;
            const int64_t m1 = m0 + nrow;
            while (cur < col_end) {
                const int64_t u = rows[cur];
                if (u >= m1) break;
                const float r = vals[cur];
                ++cur;
                if (r == 0.f) continue;  // explicit zeros stay "unobserved" (batch_R.nonzero(), recom_wmf.py:186)
                float *gp = Gl + gl_index((int)(u - m0), c);
                float pv;
                if (two_b != 0.f) {
                    pv = *gp / two_b;  // the prediction the MFMA pass just produced
                } else {              // b == 0: G carries nothing to recover p from
                    const float *urow = U + u * ld;
                    pv = 0.f;
                    for (int f = 0; f < k; f += 4) {
                        const f32x4 x = *reinterpret_cast<const f32x4 *>(urow + f), y = *reinterpret_cast<const f32x4 *>(vrow + f);
                        pv += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
                    }
                }
                *gp = 2.f * a * (pv - r);
                part += (double)a * (double)(r - pv) * (double)(r - pv) - (double)b * (double)pv * (double)pv;
            }
        }
        __syncthreads();
        WMF_STAMP();
        // (e) dV += G_t^T U_t  (U before its update): M = batch columns, N = ld, K = the tile's users
        if (!(ablate & 2)) {
            if (full) product_full<1, false>(Gl, nullptr, sb, nullptr, U + m0 * kBN, xv4, accv, (ablate & 64) != 0);
            else gemm_lds_a<true, false>(Gl, U + m0 * ld, ld, ld, nrow, sb, accv, (ablate & 64) != 0);
        }
        WMF_STAMP();
        // (d) dU = G_t V_b
        if (!(ablate & 4)) {
            if (full) product_full<2, true>(Gl, nullptr, sb, nullptr, Vb, xv4, acc, (ablate & 64) != 0);
            else gemm_lds_a<false, true>(Gl, Vb, ld, ld, B, sb, acc, (ablate & 64) != 0);
        }
        // (the trailing barrier of the product: G is dead, its region now stages dU, 64 rows at a time) clipped TF1 Adam
        // on float4s of U, m_U, v_U; loss += lambda_u/2 |U|^2 (pre-update).  (Parking the whole 128-row tile at once and
        // keeping the loads of 2 or 4 row groups in flight was measured slower: 0.82 vs 0.77 ms per step.)
        WMF_STAMP();
        if (!(ablate & 8)) {
            float *stg = Gl;  // [128][128]: the whole dU tile parked, the accumulators are dead during the sweep
            constexpr int UQ = (VAR & 2) ? 4 : (VAR & 1) ? 2 : 1;
            constexpr bool NT = (VAR & 4) != 0;
            float usq = 0.f;
            for_each_acc_local(acc, kBN, [&](int off, int, int, float v) { stg[off] = v; });
            __syncthreads();
            int tx = threadIdx.x;
            asm volatile("" : "+v"(tx));   // (opaque: see gemm_block)
            const int c4 = (tx & 31) * 4;
            if (c4 < ld) {
#pragma unroll 1
                for (int q0 = 0; q0 < 16; q0 += UQ) {
                    f32x4 u[UQ], m[UQ], v[UQ];
                    int64_t o[UQ];
#pragma unroll
                    for (int i = 0; i < UQ; ++i) {
                        const int row = (tx >> 5) + 8 * (q0 + i);
                        const int64_t gr = m0 + row;
                        o[i] = gr < n_users ? gr * ld + c4 : -1;
                        if (o[i] >= 0) {
                            u[i] = *reinterpret_cast<const f32x4 *>(U + o[i]);
                            if (NT) {
                                m[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(mU + o[i]));
                                v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(vU + o[i]));
                            } else {
                                m[i] = *reinterpret_cast<const f32x4 *>(mU + o[i]);
                                v[i] = *reinterpret_cast<const f32x4 *>(vU + o[i]);
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < UQ; ++i) {
                        if (o[i] < 0) continue;
                        const int row = (tx >> 5) + 8 * (q0 + i);
                        const f32x4 du = *reinterpret_cast<const f32x4 *>(stg + row * kBN + c4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            usq += u[i][e] * u[i][e];
                            float g = du[e] + lambda_u * u[i][e];
                            g = fminf(fmaxf(g, -5.f), 5.f);
                            m[i][e] = m[i][e] + ad.one_minus_beta1 * (g - m[i][e]);
                            v[i][e] = v[i][e] + ad.one_minus_beta2 * (g * g - v[i][e]);
                            u[i][e] = u[i][e] - ad.lr_t * m[i][e] / (sqrtf(v[i][e]) + ad.eps);
                        }
                        if (NT) {
                            __builtin_nontemporal_store(m[i], reinterpret_cast<f32x4 *>(mU + o[i]));
                            __builtin_nontemporal_store(v[i], reinterpret_cast<f32x4 *>(vU + o[i]));
                            __builtin_nontemporal_store(u[i], reinterpret_cast<f32x4 *>(U + o[i]));
                        } else {
                            *reinterpret_cast<f32x4 *>(mU + o[i]) = m[i];
                            *reinterpret_cast<f32x4 *>(vU + o[i]) = v[i];
                            *reinterpret_cast<f32x4 *>(U + o[i]) = u[i];
                        }
                    }
                }
            }
            __syncthreads();
            part += 0.5 * (double)lambda_u * (double)usq;
        }
        WMF_STAMP();
    }
#ifdef CORNAC_PROFILE
    if (prof_st) prof_st[99] = prof_n;
    if (blockIdx.x == 0 && threadIdx.x == 0 && cu_arrivals) {   // effective shader clock of this launch: cycles / (ticks / 100 MHz)
        reinterpret_cast<long long *>(cu_arrivals + 1026)[0] = (long long)clock64() - prof_c0;
        reinterpret_cast<long long *>(cu_arrivals + 1026)[1] = (long long)wall_clock64() - prof_w0;
    }
#endif
    float *mine = dv_part + (size_t)blockIdx.x * kMaxBatch * kBN;
    for_each_acc_local(accv, kBN, [&](int off, int, int, float v) { mine[off] = v; });
    const double s = block_sum_f64(part, reinterpret_cast<double *>(sb));
    if (threadIdx.x == 0 && s != 0) atomicAdd(loss, s);
}

#include "wmf_ws.inc"

// dV[c, f] += sum over a slice of the workgroups' partials (fused path; grid.y slices, dV is zero on entry: the scatter
// kernel re-zeroes it after every step)
__global__ __launch_bounds__(kWb) void wmf_reduce_dv_kernel(const float *__restrict__ dv_part, int n_parts, int B, int ld,
                                                            float *__restrict__ dV) {
    const int64_t n = (int64_t)B * ld;
    const int per = (n_parts + gridDim.y - 1) / gridDim.y;
    const int w0 = blockIdx.y * per, w1 = min(n_parts, w0 + per);
    for (int64_t e = (int64_t)blockIdx.x * kWb + threadIdx.x; e < n; e += (int64_t)gridDim.x * kWb) {
        const int c = (int)(e / ld), f = (int)(e % ld);
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
        int w = w0;
        for (; w + 3 < w1; w += 4) {
            acc0 += dv_part[((size_t)w * kMaxBatch + c) * kBN + f];
            acc1 += dv_part[((size_t)(w + 1) * kMaxBatch + c) * kBN + f];
            acc2 += dv_part[((size_t)(w + 2) * kMaxBatch + c) * kBN + f];
            acc3 += dv_part[((size_t)(w + 3) * kMaxBatch + c) * kBN + f];
        }
        for (; w < w1; ++w) acc0 += dv_part[((size_t)w * kMaxBatch + c) * kBN + f];
        atomicAdd(dV + e, (acc0 + acc1) + (acc2 + acc3));
    }
}

// TF1 Adam for an IndexedSlices gradient: m = beta1 m (+ (1-beta1) g on the slice rows), same for v, every row moves.
// The rows of the batch carry this step's tag (set by wmf_gather_kernel): their gradient g = clip(dV + lambda_v V_b) is built
// here from the reduced dV, which is re-zeroed for the next step's atomics — no dense gradient array, no scatter kernel.
__global__ __launch_bounds__(kWb) void wmf_adam_v_kernel(float *__restrict__ V, float *__restrict__ mV,
                                                         float *__restrict__ vV, int64_t n4, int k, int ld,
                                                         const uint32_t *__restrict__ slot_tag, uint32_t step,
                                                         float *__restrict__ dV, const float *__restrict__ Vb,
                                                         float lambda_v, const TfAdam ad) {
    const int ld4 = ld >> 2;
    for (int64_t e4 = (int64_t)blockIdx.x * kWb + threadIdx.x; e4 < n4; e4 += (int64_t)gridDim.x * kWb) {
        const int64_t row = e4 / ld4;
        const int f = (int)(e4 - row * ld4) * 4;
        const uint32_t tag = slot_tag[row];
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if ((tag >> 7) == step) {
            const int64_t o = (int64_t)(tag & 127u) * ld + f;
            const f32x4 d = *reinterpret_cast<const f32x4 *>(dV + o), vb = *reinterpret_cast<const f32x4 *>(Vb + o);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (f + q < k) g[q] = fminf(fmaxf(d[q] + lambda_v * vb[q], -5.f), 5.f);
            *reinterpret_cast<f32x4 *>(dV + o) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 m = *reinterpret_cast<const f32x4 *>(mV + e4 * 4), v = *reinterpret_cast<const f32x4 *>(vV + e4 * 4);
        f32x4 x = *reinterpret_cast<const f32x4 *>(V + e4 * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            m[q] *= ad.beta1;
            v[q] *= ad.beta2;
            if (g[q] != 0.f) {
                m[q] = m[q] + ad.one_minus_beta1 * g[q];
                v[q] = v[q] + ad.one_minus_beta2 * (g[q] * g[q]);
            }
            x[q] = x[q] - ad.lr_t * m[q] / (sqrtf(v[q]) + ad.eps);
        }
        *reinterpret_cast<f32x4 *>(mV + e4 * 4) = m;
        *reinterpret_cast<f32x4 *>(vV + e4 * 4) = v;
        *reinterpret_cast<f32x4 *>(V + e4 * 4) = x;
    }
}

__global__ __launch_bounds__(kWb) void wmf_pad_kernel(const float *__restrict__ src, int64_t rows, int k, int ld,
                                                      float *__restrict__ dst) {
    const int64_t n = rows * ld;
    for (int64_t e = (int64_t)blockIdx.x * kWb + threadIdx.x; e < n; e += (int64_t)gridDim.x * kWb) {
        const int64_t r = e / ld;
        const int f = (int)(e % ld);
        dst[e] = f < k ? src[r * k + f] : 0.f;
    }
}

__global__ __launch_bounds__(kWb) void wmf_unpad_kernel(const float *__restrict__ src, int64_t rows, int k, int ld,
                                                        float *__restrict__ dst) {
    const int64_t n = rows * k;
    for (int64_t e = (int64_t)blockIdx.x * kWb + threadIdx.x; e < n; e += (int64_t)gridDim.x * kWb) {
        const int64_t r = e / k;
        const int f = (int)(e % k);
        dst[e] = src[r * ld + f];
    }
}

}  // namespace chip

using namespace chip;

struct cornac_hip_wmf {
    int device = 0;
    int64_t n_users = 0, n_items = 0, nnz = 0;
    int k = 0, ld = 0;
    hipStream_t stream = nullptr;
    DevBuf<float> U, V, mU, vU, mV, vV;       // [rows x ld], zero padded columns
    DevBuf<uint32_t> slot_tag;                // [n_items]: (step << 7 | batch column) of the last batch the row was in
    DevBuf<float> G, Vb, dV, stage;           // [n_users x 128], [128 x ld], [128 x ld], host<->device staging
    DevBuf<float> g_scratch, dv_part, VbT;    // fused path: per-workgroup G tile [wgs x 128 x 128], dV partials, V_b^T [ld x 128]
    int fused_wgs = 0;
    DevBuf<int64_t> indptr;                   // CSC
    DevBuf<int32_t> rows, ids;          // ids: all batches of the current call, back to back
    DevBuf<float> vals;
    DevBuf<double> loss;
    DevBuf<float> ws_dump;                    // wave-specialised kernel: where the updates of rows that do not exist go
    int ws_wgs = 0;
    DevBuf<int> cu_arrivals;                  // [1024 physical CU ids + 1]: arrival order of the fused kernel's workgroups
    std::vector<int64_t> h_indptr;
    int64_t step = 0;
    EventTimer timer;
    double last_kernel_ms = 0;
};

static void wmf_check(cornac_hip_wmf_t h) {
    REQUIRE(h != nullptr, "WMF handle is NULL");
    HIP_CHECK(hipSetDevice(h->device));
}

typedef void (*WmfLdsKernel)(const float *, const float *, int64_t, int, int, int, float *, float *, float *, const int64_t *,
                             const int32_t *, const float *, const int32_t *, float, float, float, const TfAdam, float *, double *,
                             int, int, int *);
constexpr int kWmfDefaultVariant = 0;
// the variant of the fused kernel: kWmfDefaultVariant, or CORNAC_HIP_WMF_VARIANT in profile builds (tools/wmf_ablate.py)
static WmfLdsKernel pick_wmf_lds_kernel(int var) {
#ifdef CORNAC_PROFILE
    switch (var & 31) {
#define V(i) case i: return wmf_user_step_lds_kernel<i>;
        V(0) V(1) V(2) V(4) V(5) V(6) V(8) V(9) V(13) V(16) V(21)
#undef V
        default: break;
    }
#endif
    (void)var;
    return wmf_user_step_lds_kernel<kWmfDefaultVariant>;
}

static int grid_for(int64_t n, int cap = 4096) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + kWb - 1) / kWb, cap)); }

extern "C" {

int cornac_hip_wmf_create(cornac_hip_wmf_t *out, int device, int64_t n_users, int64_t n_items, int k,
                          const int64_t *csc_indptr, const int32_t *csc_rows, const float *csc_vals, int64_t nnz) {
    return guarded([&] {
        REQUIRE(out != nullptr, "out handle pointer is NULL");
        *out = nullptr;
        REQUIRE(n_users > 0 && n_items > 0 && k > 0, "sizes must be positive");
        REQUIRE(k <= 1024, "k <= 1024 supported");
        REQUIRE(csc_indptr && (nnz == 0 || (csc_rows && csc_vals)), "CSC arrays are NULL");
        REQUIRE(csc_indptr[0] == 0 && csc_indptr[n_items] == nnz, "CSC indptr does not match nnz");
        for (int64_t i = 0; i < n_items; ++i) REQUIRE(csc_indptr[i] <= csc_indptr[i + 1], "CSC indptr not monotone");
        for (int64_t e = 0; e < nnz; ++e) REQUIRE(csc_rows[e] >= 0 && csc_rows[e] < n_users, "CSC row out of range");
        use_device(device);
        std::unique_ptr<cornac_hip_wmf> h(new cornac_hip_wmf());
        h->device = device; h->n_users = n_users; h->n_items = n_items; h->k = k; h->nnz = nnz;
        h->ld = (k + 31) / 32 * 32;
        HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        const size_t nu = (size_t)n_users * h->ld, ni = (size_t)n_items * h->ld;
        // (U, m_U, v_U are padded to whole 128-row tiles + one tile, zero filled: the wave-specialised user step reads operand rows
        // of the ragged last tile and of "the next tile" without clamping, and its Adam sweep runs over whole tiles — rows that
        // do not exist hold zeros and stay zero)
        const size_t nu_pad = ((size_t)(n_users + kBM - 1) / kBM + 1) * kBM * h->ld;
        for (DevBuf<float> *b : {&h->U, &h->mU, &h->vU}) { b->alloc(nu_pad); HIP_CHECK(hipMemsetAsync(b->p, 0, nu_pad * 4, h->stream)); }
        for (DevBuf<float> *b : {&h->V, &h->mV, &h->vV}) { b->alloc(ni); HIP_CHECK(hipMemsetAsync(b->p, 0, ni * 4, h->stream)); }
        h->slot_tag.alloc((size_t)n_items);
        HIP_CHECK(hipMemsetAsync(h->slot_tag.p, 0, (size_t)n_items * 4, h->stream));
        h->Vb.alloc((size_t)kMaxBatch * h->ld);
        HIP_CHECK(hipMemsetAsync(h->Vb.p, 0, h->Vb.n * 4, h->stream));   // (rows >= the batch size are multiplied by zeros of G)
        h->dV.alloc((size_t)kMaxBatch * h->ld);
        HIP_CHECK(hipMemsetAsync(h->dV.p, 0, h->dV.n * 4, h->stream));
        h->stage.alloc(std::max(nu, ni));
        h->indptr.alloc((size_t)n_items + 1);
        h->indptr.upload(csc_indptr, (size_t)n_items + 1, h->stream);
        h->h_indptr.assign(csc_indptr, csc_indptr + n_items + 1);
        h->rows.alloc((size_t)std::max<int64_t>(nnz, 1));
        h->vals.alloc((size_t)std::max<int64_t>(nnz, 1));
        // the fused step kernel binary-searches a column for the rows of a user tile: columns sorted by row
        std::vector<int32_t> srows;
        std::vector<float> svals;
        bool sorted = true;
        for (int64_t i = 0; i < n_items && sorted; ++i)
            for (int64_t e = csc_indptr[i] + 1; e < csc_indptr[i + 1]; ++e)
                if (csc_rows[e] < csc_rows[e - 1]) { sorted = false; break; }
        if (!sorted) {
            srows.assign(csc_rows, csc_rows + nnz);
            svals.assign(csc_vals, csc_vals + nnz);
            std::vector<int64_t> perm;
            for (int64_t i = 0; i < n_items; ++i) {
                const int64_t lo = csc_indptr[i], hi = csc_indptr[i + 1];
                perm.resize((size_t)(hi - lo));
                for (int64_t e = lo; e < hi; ++e) perm[(size_t)(e - lo)] = e;
                std::stable_sort(perm.begin(), perm.end(), [&](int64_t x, int64_t y) { return csc_rows[x] < csc_rows[y]; });
                for (int64_t e = lo; e < hi; ++e) {
                    srows[(size_t)e] = csc_rows[perm[(size_t)(e - lo)]];
                    svals[(size_t)e] = csc_vals[perm[(size_t)(e - lo)]];
                }
            }
            csc_rows = srows.data();
            csc_vals = svals.data();
        }
        if (nnz) { h->rows.upload(csc_rows, (size_t)nnz, h->stream); h->vals.upload(csc_vals, (size_t)nnz, h->stream); }
        HIP_CHECK(hipStreamSynchronize(h->stream));
        *out = h.release();
    });
}

void cornac_hip_wmf_destroy(cornac_hip_wmf_t h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    delete h;
}

// uploads U [n_users x k], V [n_items x k] and resets the Adam state (a fresh TF graph per fit, recom_wmf.py:166-178)
int cornac_hip_wmf_set_factors(cornac_hip_wmf_t h, const float *U, const float *V) {
    return guarded([&] {
        wmf_check(h);
        REQUIRE(U && V, "factor pointers are NULL");
        h->stage.upload(U, (size_t)h->n_users * h->k, h->stream);
        wmf_pad_kernel<<<grid_for(h->n_users * h->ld), kWb, 0, h->stream>>>(h->stage.p, h->n_users, h->k, h->ld, h->U.p);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->stage.upload(V, (size_t)h->n_items * h->k, h->stream);
        wmf_pad_kernel<<<grid_for(h->n_items * h->ld), kWb, 0, h->stream>>>(h->stage.p, h->n_items, h->k, h->ld, h->V.p);
        for (DevBuf<float> *b : {&h->mU, &h->vU, &h->mV, &h->vV}) HIP_CHECK(hipMemsetAsync(b->p, 0, b->n * 4, h->stream));
        HIP_CHECK(hipMemsetAsync(h->slot_tag.p, 0, h->slot_tag.n * 4, h->stream));   // (steps restart at 1: no stale tag may match)
        h->step = 0;
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}

int cornac_hip_wmf_get_factors(cornac_hip_wmf_t h, float *U, float *V) {
    return guarded([&] {
        wmf_check(h);
        REQUIRE(U && V, "factor pointers are NULL");
        wmf_unpad_kernel<<<grid_for(h->n_users * h->k), kWb, 0, h->stream>>>(h->U.p, h->n_users, h->k, h->ld, h->stage.p);
        h->stage.download(U, (size_t)h->n_users * h->k, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        wmf_unpad_kernel<<<grid_for(h->n_items * h->k), kWb, 0, h->stream>>>(h->V.p, h->n_items, h->k, h->ld, h->stage.p);
        h->stage.download(V, (size_t)h->n_items * h->k, h->stream);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}

// One Adam step per batch of item ids (batch_ptr[b] .. batch_ptr[b+1]), in order; loss_out[b] = the step's loss
// (the `_loss` the reference sums for its progress bar, recom_wmf.py:195-199).
int cornac_hip_wmf_fit_batches(cornac_hip_wmf_t h, const int32_t *item_ids, const int64_t *batch_ptr,
                               int64_t n_batches, float lambda_u, float lambda_v, float a, float b,
                               float learning_rate, double *loss_out) {
    return guarded([&] {
        wmf_check(h);
        REQUIRE(n_batches >= 0, "n_batches < 0");
        if (n_batches == 0) return;
        REQUIRE(item_ids && batch_ptr, "batch arrays are NULL");
        for (int64_t bi = 0; bi < n_batches; ++bi) {
            const int64_t B = batch_ptr[bi + 1] - batch_ptr[bi];
            REQUIRE(B > 0 && B <= kMaxBatch, "batch %lld has %lld items; 1..%d supported", (long long)bi, (long long)B, kMaxBatch);
            for (int64_t c = 0; c < B; ++c) {
                const int32_t it = item_ids[batch_ptr[bi] + c];
                REQUIRE(it >= 0 && it < h->n_items, "item id %d out of range", it);
            }
        }
        const double beta1 = 0.9, beta2 = 0.999;
        hipStream_t s = h->stream;
        const int ld = h->ld, k = h->k;
        const int64_t nu = h->n_users;
        const int m_tiles = (int)((nu + kBM - 1) / kBM), n_tiles = (ld + kBN - 1) / kBN;
        // user chunks of the split-K gradient: enough workgroups to fill the chip, chunk a multiple of the k tile
        int64_t chunk = std::max<int64_t>(kBK, (nu + 511) / 512);
        chunk = (chunk + kBK - 1) / kBK * kBK;
        const int n_chunks = (int)((nu + chunk - 1) / chunk);
        // k <= 128: the whole user side of a step is one persistent kernel (wmf_user_step_kernel)
        const bool no_fuse = prof_env_set("CORNAC_HIP_WMF_UNFUSED");  // A/B switch, profile builds only (csrc/common.h)
        const int wmf_ablate = prof_env_int("CORNAC_HIP_WMF_ABLATE", 0);
        const bool fused = ld <= kBN && !no_fuse;
        const bool g_scratch = prof_env_set("CORNAC_HIP_WMF_GSCRATCH");  // A/B switch: G in a global scratch tile
        const int wmf_variant = prof_env_int("CORNAC_HIP_WMF_VARIANT", kWmfDefaultVariant);
        const int stagger_ticks = prof_env_int("CORNAC_HIP_WMF_STAGGER", 2000);   // 20 us: half of a tile's time
        const WmfLdsKernel lds_kernel = pick_wmf_lds_kernel(wmf_variant);
        if (fused && !g_scratch)
            HIP_CHECK(hipFuncSetAttribute((const void *)lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWmfLdsBytes));
        if (fused && h->fused_wgs == 0) {
            int per_cu = 0;
            if (g_scratch) {
                HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wmf_user_step_kernel, kWb, 0));
            } else {
                HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wmf_user_step_lds_kernel<kWmfDefaultVariant>, kWb, kWmfLdsBytes));
            }
            if (prof_env_set("CORNAC_HIP_WMF_DEBUG")) fprintf(stderr, "[wmf] fused kernel: %d workgroups per CU\n", per_cu);
            per_cu = std::min(per_cu, prof_env_int("CORNAC_HIP_WMF_WGS_PER_CU", 2));   // (profile builds: single-wave-per-SIMD timing)
            h->fused_wgs = (int)std::min<int64_t>(m_tiles, (int64_t)device_info(h->device).cus * std::max(1, std::min(per_cu, 2)));
            h->g_scratch.alloc((size_t)h->fused_wgs * kBM * kMaxBatch);
            h->dv_part.alloc((size_t)h->fused_wgs * kMaxBatch * kBN);
            h->VbT.alloc((size_t)ld * kMaxBatch);
            h->cu_arrivals.alloc(1040 + 400);
            HIP_CHECK(hipMemsetAsync(h->VbT.p, 0, h->VbT.n * 4, h->stream));
        }
        // ld == 128: one 8-wave workgroup per CU, MFMA waves beside streaming waves (wmf_ws.inc)
        const bool ws = fused && !g_scratch && ld == kBN && !prof_env_set("CORNAC_HIP_WMF_NO_WS");
        if (ws && h->ws_wgs == 0) {
            HIP_CHECK(hipFuncSetAttribute((const void *)wmf_user_step_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)kWmfWsLdsBytes));
            h->ws_wgs = (int)std::min<int64_t>(m_tiles, (int64_t)device_info(h->device).cus);
            h->ws_dump.alloc((size_t)h->ws_wgs * kWsDumpFloats);
        }
        if (!fused && !h->G.p) h->G.alloc((size_t)nu * kMaxBatch);
        const int64_t n_ids = batch_ptr[n_batches] - batch_ptr[0];
        h->ids.ensure((size_t)n_ids);
        h->loss.ensure((size_t)n_batches);
        HIP_CHECK(hipMemcpyAsync(h->ids.p, item_ids + batch_ptr[0], (size_t)n_ids * sizeof(int32_t), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemsetAsync(h->loss.p, 0, (size_t)n_batches * sizeof(double), s));
        h->timer.before(s);
        for (int64_t bi = 0; bi < n_batches; ++bi) {
            const int32_t *ids = item_ids + batch_ptr[bi];
            const int32_t *d_ids = h->ids.p + (batch_ptr[bi] - batch_ptr[0]);
            double *d_loss = h->loss.p + bi;
            const int B = (int)(batch_ptr[bi + 1] - batch_ptr[bi]);
            int64_t max_col = 0;
            for (int c = 0; c < B; ++c) max_col = std::max(max_col, h->h_indptr[ids[c] + 1] - h->h_indptr[ids[c]]);
            ++h->step;
            TfAdam ad;
            ad.beta1 = (float)beta1; ad.beta2 = (float)beta2;
            ad.one_minus_beta1 = 1.f - ad.beta1; ad.one_minus_beta2 = 1.f - ad.beta2;
            ad.lr_t = (float)((double)learning_rate * std::sqrt(1.0 - std::pow(beta2, (double)h->step)) /
                              (1.0 - std::pow(beta1, (double)h->step)));
            ad.eps = 1e-8f;
            wmf_gather_kernel<<<grid_for((int64_t)B * ld, 64), kWb, 0, s>>>(h->V.p, d_ids, B, ld, h->Vb.p, fused ? h->VbT.p : nullptr, 0.5f * lambda_v, d_loss,
                                                                           h->slot_tag.p, (uint32_t)h->step);
            if (fused) {
                if (ws && prof_env_set("CORNAC_HIP_WMF_CLOCK")) {
                    const unsigned long long init[4] = {~0ull, 0ull, ~0ull, 0ull};
                    HIP_CHECK(hipMemcpyAsync(h->cu_arrivals.p + 1030, init, sizeof(init), hipMemcpyHostToDevice, s));
                }
#ifdef CORNAC_PROFILE
                if (ws) {   // timing experiments of the stream role (wrong results): wmf_ws.inc WS_ABL
                    static int abl;
                    abl = prof_env_int("CORNAC_HIP_WMF_WS_ABLATE", 0);
                    HIP_CHECK(hipMemcpyAsync(h->cu_arrivals.p + 1038, &abl, sizeof(int), hipMemcpyHostToDevice, s));
                }
#endif
                if (ws)
                    wmf_user_step_ws_kernel<<<h->ws_wgs, kWsThreads, kWmfWsLdsBytes, s>>>(h->Vb.p, h->VbT.p, nu, B, h->U.p, h->mU.p, h->vU.p,
                                                                                          h->indptr.p, h->rows.p, h->vals.p, h->nnz, d_ids, a, b,
                                                                                          lambda_u, ad, h->dv_part.p, h->ws_dump.p, d_loss,
                                                                                          h->cu_arrivals.p);
                if (ws && prof_env_set("CORNAC_HIP_WMF_CLOCK") && bi == n_batches - 1) {
                    static long long st[200];
                    HIP_CHECK(hipMemcpyAsync(st, h->cu_arrivals.p + 1040, sizeof(st), hipMemcpyDeviceToHost, s));
                    HIP_CHECK(hipStreamSynchronize(s));
                    // per tile: start, P done, G stored, fix-up done, dV done, dU done, parked (us); math wave 0 of workgroup 0
                    unsigned long long mm[4];
                    HIP_CHECK(hipMemcpyAsync(mm, h->cu_arrivals.p + 1030, sizeof(mm), hipMemcpyDeviceToHost, s));
                    HIP_CHECK(hipStreamSynchronize(s));
                    fprintf(stderr, "[wmf-ws] first start to last end over all workgroups: math waves %.1f us, stream waves %.1f us\n",
                            (mm[1] - mm[0]) / 100.0, (mm[3] - mm[2]) / 100.0);
                    fprintf(stderr, "[wmf-ws] %.3f GHz; stamps of workgroup 0 (us):", st[98] / 10000.0);
                    for (int i = 0; i < (int)st[99] && i < 98; ++i) fprintf(stderr, "%s%.1f", i % 7 ? " " : " | ", st[i] / 100.0);
                    fprintf(stderr, "\n");
                }
                if (ws) {
                } else if (g_scratch)
                    wmf_user_step_kernel<<<h->fused_wgs, kWb, 0, s>>>(h->Vb.p, h->VbT.p, nu, B, k, ld, h->U.p, h->mU.p, h->vU.p, h->g_scratch.p,
                                                                      h->indptr.p, h->rows.p, h->vals.p, d_ids, a, b, lambda_u, ad,
                                                                      h->dv_part.p, d_loss, wmf_ablate);
                else {
                    if (wmf_variant & 8) HIP_CHECK(hipMemsetAsync(h->cu_arrivals.p, 0, 1025 * sizeof(int), s));
                    lds_kernel<<<h->fused_wgs, kWb, kWmfLdsBytes, s>>>(h->Vb.p, h->VbT.p, nu, B, k, ld, h->U.p, h->mU.p, h->vU.p, h->indptr.p,
                                                                       h->rows.p, h->vals.p, d_ids, a, b, lambda_u, ad, h->dv_part.p, d_loss,
                                                                       wmf_ablate, stagger_ticks, h->cu_arrivals.p);
                    if (prof_env_set("CORNAC_HIP_WMF_CLOCK") && bi == n_batches - 1 && !ws) {
                        long long c[2] = {0, 0};
                        HIP_CHECK(hipMemcpyAsync(c, h->cu_arrivals.p + 1026, sizeof(c), hipMemcpyDeviceToHost, s));
                        HIP_CHECK(hipStreamSynchronize(s));
                        fprintf(stderr, "[wmf] workgroup 0: %lld shader cycles in %.1f us = %.3f GHz\n", c[0], c[1] / 100.0,
                                c[1] ? c[0] / (c[1] * 10.0) : 0.0);
                        static long long st[200];
                        HIP_CHECK(hipMemcpyAsync(st, h->cu_arrivals.p + 1040, sizeof(st), hipMemcpyDeviceToHost, s));
                        HIP_CHECK(hipStreamSynchronize(s));
                        for (int w = 0; w < 2; ++w) {   // per tile: start, P done, G + fix-up done, dV done, dU done, Adam done (us)
                            fprintf(stderr, "[wmf] stamps of workgroup %s (us):", w ? "last" : "0");
                            for (int i = 0; i < (int)st[w * 100 + 99] && i < 96; ++i)
                                fprintf(stderr, "%s%.1f", i % 6 ? " " : " | ", st[w * 100 + i] / 100.0);
                            fprintf(stderr, "\n");
                        }
                    }
                    if ((wmf_variant & 8) && prof_env_set("CORNAC_HIP_WMF_DEBUG") && bi == 0) {
                        int late = 0;
                        HIP_CHECK(hipMemcpyAsync(&late, h->cu_arrivals.p + 1024, sizeof(int), hipMemcpyDeviceToHost, s));
                        HIP_CHECK(hipStreamSynchronize(s));
                        fprintf(stderr, "[wmf] %d of %d workgroups started late\n", late, h->fused_wgs);
                    }
                }
                wmf_reduce_dv_kernel<<<dim3(grid_for((int64_t)B * ld, 64), 32), kWb, 0, s>>>(h->dv_part.p, ws ? h->ws_wgs : h->fused_wgs, B, ld, h->dV.p);
                wmf_adam_v_kernel<<<grid_for(h->n_items * ld / 4), kWb, 0, s>>>(h->V.p, h->mV.p, h->vV.p, h->n_items * ld / 4, k, ld, h->slot_tag.p,
                                                                                (uint32_t)h->step, h->dV.p, h->Vb.p, lambda_v, ad);
                HIP_CHECK(hipGetLastError());
                continue;
            }
            wmf_pred_kernel<<<m_tiles, kWb, 0, s>>>(h->U.p, h->Vb.p, nu, B, k, ld, h->G.p, b, d_loss);
            if (max_col > 0) {
                const int gx = (int)std::max<int64_t>(1, std::min<int64_t>((max_col + kWb / 16 - 1) / (kWb / 16), 64));
                wmf_fixup_kernel<<<dim3(gx, B), kWb, 0, s>>>(h->U.p, h->Vb.p, h->indptr.p, h->rows.p, h->vals.p, d_ids, k, ld, h->G.p, a, b, d_loss);
            }
            wmf_grad_v_kernel<<<dim3(n_chunks, n_tiles), kWb, 0, s>>>(h->G.p, h->U.p, nu, B, ld, chunk, h->dV.p);
            wmf_update_u_kernel<<<dim3(m_tiles, n_tiles), kWb, 0, s>>>(h->G.p, h->Vb.p, nu, B, k, ld, h->U.p, h->mU.p, h->vU.p, lambda_u, ad, d_loss);
            wmf_adam_v_kernel<<<grid_for(h->n_items * ld / 4), kWb, 0, s>>>(h->V.p, h->mV.p, h->vV.p, h->n_items * ld / 4, k, ld, h->slot_tag.p,
                                                                            (uint32_t)h->step, h->dV.p, h->Vb.p, lambda_v, ad);
            HIP_CHECK(hipGetLastError());
        }
        h->timer.after(s);
        HIP_CHECK(hipStreamSynchronize(s));
        if (h->timer.enabled) {
            int64_t launches = 0;
            h->timer.collect(&h->last_kernel_ms, &launches);
        }
        if (loss_out) {
            h->loss.download(loss_out, (size_t)n_batches, s);
            HIP_CHECK(hipStreamSynchronize(s));
        }
    });
}

int cornac_hip_wmf_kernel_timing(cornac_hip_wmf_t h, int enabled) {
    return guarded([&] {
        wmf_check(h);
        h->timer.enabled = enabled != 0;
    });
}

int cornac_hip_wmf_last_timing(cornac_hip_wmf_t h, double *device_ms) {
    return guarded([&] {
        wmf_check(h);
        if (device_ms) *device_ms = h->last_kernel_ms;
    });
}

}  // extern "C"
