// VBPR (visual BPR) minibatch training on MI355X (gfx950).
//
// Replaces the per-batch body of VBPR._fit_torch (cornac/models/vbpr/recom_vbpr.py:228-262): gather of
// the user/item rows and of the 4096-d item features, the pairwise score
//   X_uij = b_i - b_j + <g_u, g_i - g_j> + <t_u, (f_i - f_j) E> + (f_i - f_j) b'
// its analytic gradient (what torch autograd produces for -sum(logsigmoid(X)) + L2 terms, :242-253) and
// `torch.optim.Adam` over ALL tables (:209, dense: rows outside the batch have zero gradient but their
// moments still decay and move the parameter).  The (u, i, j) batches come from the host sampler
// (Dataset.uij_iter, a Python-level iterator in the reference too) and are uploaded per fit call.
//
// Per Adam step t (batch of B triplets), on two streams of the handle:
//   main stream
//   vbpr_featdiff_kernel  one workgroup per triplet: DF[b] = F[i_b] - F[j_b] (the "auxiliary-feature gather"),
//                         v_b = DF[b] . b'; also stamps the batch's user / item rows with the step number and clears
//                         proj[b]
//   vbpr_proj_kernel      proj = DF E on the fp32 matrix cores (mfma_gemm.h), split over feature chunks
//   vbpr_score_kernel     s_b from the gathered rows and proj_b (one wave per triplet)  [waits for sweep t-1]
//   vbpr_pair_grad_kernel the reference's B x B broadcast objective -> gs_b, gv_b (one workgroup per b)
//   vbpr_scatter_kernel   sparse row gradients scattered with fp32 atomics
//   vbpr_touched_adam_kernel  Adam step of the rows the batch touched (each distinct row once; consumes and clears
//                         its scattered gradient)
//   vbpr_feat_adam_kernel gradient of E / beta' (DF^T x [gs t_u | gv] on the matrix cores) fused with their Adam step
//   sweep stream (starts as soon as the rows are stamped, runs beside everything above)
//   adam_sweep_kernel     dense Adam over Bi, Gu, Gi, Tu for every row NOT stamped with t: their gradient is exactly
//                         zero (the L2 terms only cover the batch rows, recom_vbpr.py:251-253), so the sweep reads and
//                         writes p, m, v only — 24 bytes per parameter and step
// HBM-bound by the dense Adam sweep (all parameters + two moments per step); the latency-bound chain of small
// kernels hides behind it.
#include <algorithm>
#include <cmath>

#include <hip/hip_ext.h>

#include "common.h"
#include "mfma_gemm.h"

namespace chip {

constexpr int kVb = 256;
constexpr int kMaxK2 = 256;

struct VbprTables {
    const float *F;        // [n_items, n_feat]
    float *Bi, *Gu, *Gi, *Tu, *E, *Bp;
    float *gBi, *gGu, *gGi, *gTu;  // scattered (dense) gradients of the row tables
    int32_t *stamp_u, *stamp_i;     // step number of the last batch that touched the row (negated once its Adam step ran)
    int64_t n_users, n_items;
    int k, k2, n_feat;
};

// NOTE on the reference's score shape: `feat_diff.mm(Bp)` is [B, 1] while the other terms are [B], so
// `Xuij` broadcasts to a B x B matrix X[a, b] = s_b + v_a  (s_b = b_i - b_j + <g_u, g_i - g_j> + <t_u, df_b E>,
// v_a = df_a . b') and `logsigmoid(Xuij).sum()` runs over all B^2 entries (recom_vbpr.py:242-249).  The
// reference's results ARE that objective, so it is reproduced: with G[a, b] = -sigmoid(-(s_b + v_a)),
//   d loss / d s_b = gs_b = sum_a G[a, b]   (drives b_i, b_j, g_u, g_i, g_j, t_u and E)
//   d loss / d v_a = gv_a = sum_b G[a, b]   (drives b')

// stage 1a — kFeatSlices workgroups per triplet: feature difference DF[b, :] = F[i_b] - F[j_b] (coalesced row reads,
// written once for the two feature GEMMs of the step) and the slice's part of v_b = DF[b] . b' (summed in a fixed
// order by the consumer).  Slice 0 also stamps the batch's rows with the step number and clears proj[b].
constexpr int kFeatSlices = 4;
__global__ __launch_bounds__(kVb) void vbpr_featdiff_kernel(const VbprTables t, const int32_t *__restrict__ bu,
                                                            const int32_t *__restrict__ bi,
                                                            const int32_t *__restrict__ bj, float *__restrict__ DF,
                                                            float *__restrict__ v_part, float *__restrict__ proj,
                                                            int32_t step) {
    __shared__ float red[kVb / 64];
    const int b = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
    if (sl == 0) {
        if (tid == 0) {  // (several triplets may name the same row: they all write the same value)
            t.stamp_u[bu[b]] = step;
            t.stamp_i[bi[b]] = step;
            t.stamp_i[bj[b]] = step;
        }
        for (int c = tid; c < t.k2; c += kVb) proj[(size_t)b * t.k2 + c] = 0.f;
    }
    const float *fi = t.F + (int64_t)bi[b] * t.n_feat, *fj = t.F + (int64_t)bj[b] * t.n_feat;
    float *df = DF + (int64_t)b * t.n_feat;
    const int per = (t.n_feat + kFeatSlices - 1) / kFeatSlices;
    const int f_end = min(t.n_feat, (sl + 1) * per);
    float vb = 0.f;
    for (int f = sl * per + tid; f < f_end; f += kVb) {
        const float d = fi[f] - fj[f];
        df[f] = d;
        vb = fmaf(d, t.Bp[f], vb);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vb += __shfl_xor(vb, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = vb;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int w = 0; w < kVb / 64; ++w) tot += red[w];
        v_part[(size_t)b * kFeatSlices + sl] = tot;
    }
}

// stage 1b — proj = DF E  ([B, n_feat] x [n_feat, k2]) on the fp32 matrix cores, split over feature chunks
// (blockIdx.x) with fp32 atomics into the zeroed proj; blockIdx.y / .z tile B and k2 when they exceed 128
__global__ __launch_bounds__(kWb) void vbpr_proj_kernel(const float *__restrict__ DF, const float *__restrict__ E, int n,
                                                        int n_feat, int k2, int chunk, float *__restrict__ proj) {
    __shared__ GemmSmem sm;
    f32x16 acc[2][2];
    const int64_t m0 = (int64_t)blockIdx.y * kBM, n0 = (int64_t)blockIdx.z * kBN;
    const int64_t k_begin = (int64_t)blockIdx.x * chunk;
    const int64_t k_end = k_begin + chunk < n_feat ? k_begin + chunk : n_feat;
    gemm_block<true, true>(DF, n_feat, 1, E, k2, 1, n, k2, m0, n0, k_begin, k_end, sm, acc);
    for_each_acc(acc, m0, n0, [&](int64_t row, int64_t col, float v) {
        if (row < n && col < k2 && v != 0.f) atomicAdd(proj + row * k2 + col, v);
    });
}

// stage 1c — s_b = b_i - b_j + <g_u, g_i - g_j> + <t_u, proj_b>: one wave per triplet
__global__ __launch_bounds__(kVb) void vbpr_score_kernel(const VbprTables t, const int32_t *__restrict__ bu,
                                                         const int32_t *__restrict__ bi,
                                                         const int32_t *__restrict__ bj, int n,
                                                         const float *__restrict__ proj, float *__restrict__ s_out) {
    const int b = (blockIdx.x * kVb + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (b >= n) return;
    const int64_t u = bu[b], i = bi[b], j = bj[b];
    const float *gu = t.Gu + u * t.k, *gi = t.Gi + i * t.k, *gj = t.Gi + j * t.k, *tu = t.Tu + u * t.k2;
    float part = 0.f;
    for (int q = lane; q < t.k; q += 64) part = fmaf(gu[q], gi[q] - gj[q], part);
    for (int q = lane; q < t.k2; q += 64) part = fmaf(tu[q], proj[(size_t)b * t.k2 + q], part);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) s_out[b] = (t.Bi[i] - t.Bi[j]) + part;
}

// stage 2 — the B x B broadcast objective: gs_b = sum_a G[a,b], gv_a = sum_b G[a,b], NLL = sum softplus(-X).
// One workgroup per b, threads over a (strided): column b and row b of G in one pass.
__global__ __launch_bounds__(kVb) void vbpr_pair_grad_kernel(const float *__restrict__ s, const float *__restrict__ v,
                                                             int n, float *__restrict__ gs, float *__restrict__ gv,
                                                             double *__restrict__ loss_acc) {
    __shared__ float red_c[kVb / 64], red_r[kVb / 64];
    __shared__ double red_n[kVb / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    auto v_of = [&](int x) {  // the feature slices' parts of v_x, always summed in the same order
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < kFeatSlices; ++q) tot += v[(size_t)x * kFeatSlices + q];
        return tot;
    };
    const float sb = s[b], vb = v_of(b);
    float col = 0.f, row = 0.f;
    double nll = 0.0;
    for (int a = tid; a < n; a += kVb) {
        const float X = sb + v_of(a);              // X[a, b]
        col += -1.0f / (1.0f + expf(X));
        nll += (X > 0.f) ? log1p(exp(-(double)X)) : (-(double)X + log1p(exp((double)X)));
        const float Y = s[a] + vb;                 // X[b, a]
        row += -1.0f / (1.0f + expf(Y));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        col += __shfl_xor(col, o, 64);
        row += __shfl_xor(row, o, 64);
        nll += __shfl_xor(nll, o, 64);
    }
    if ((tid & 63) == 0) {
        red_c[tid >> 6] = col;
        red_r[tid >> 6] = row;
        red_n[tid >> 6] = nll;
    }
    __syncthreads();
    if (tid == 0) {
        float c = 0.f, r = 0.f;
        double l = 0.0;
        for (int w = 0; w < kVb / 64; ++w) {
            c += red_c[w];
            r += red_r[w];
            l += red_n[w];
        }
        gs[b] = c;
        gv[b] = r;
        atomicAdd(loss_acc, l);
    }
}

// stage 3 — sparse row gradients (duplicates inside a batch accumulate, like autograd's index_put accumulate)
__global__ __launch_bounds__(kVb) void vbpr_scatter_kernel(const VbprTables t, const int32_t *__restrict__ bu,
                                                           const int32_t *__restrict__ bi,
                                                           const int32_t *__restrict__ bj, int n,
                                                           const float *__restrict__ gs,
                                                           const float *__restrict__ gv,
                                                           const float *__restrict__ proj, float lambda_w,
                                                           float lambda_b, float *__restrict__ W, int ldw) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t u = bu[b], i = bi[b], j = bj[b];
    const float g = gs[b];
    const int k2 = t.k2;
    // right-hand side of the E / beta' gradient GEMM: W[b, :] = [gs_b * t_u | gv_b | 0 padding]
    for (int q = tid; q < ldw; q += kVb)
        W[(size_t)b * ldw + q] = q < k2 ? g * t.Tu[u * k2 + q] : (q == k2 ? gv[b] : 0.f);
    const float *gu = t.Gu + u * t.k, *gi = t.Gi + i * t.k, *gj = t.Gi + j * t.k, *tu = t.Tu + u * k2;
    if (tid == 0) {
        atomicAdd(t.gBi + i, g + lambda_b * t.Bi[i]);
        atomicAdd(t.gBi + j, -g + (lambda_b / 10.f) * t.Bi[j]);
    }
    for (int q = tid; q < t.k; q += kVb) {
        const float a = gu[q], vi = gi[q], vj = gj[q];
        atomicAdd(t.gGu + u * t.k + q, g * (vi - vj) + lambda_w * a);
        atomicAdd(t.gGi + i * t.k + q, g * a + lambda_w * vi);
        atomicAdd(t.gGi + j * t.k + q, -g * a + lambda_w * vj);
    }
    for (int q = tid; q < k2; q += kVb) atomicAdd(t.gTu + u * k2 + q, g * proj[(size_t)b * k2 + q] + lambda_w * tu[q]);
}

struct AdamScalars {
    float beta1, beta2, one_minus_beta1, one_minus_beta2, step_size, bc2_sqrt, eps;
};

// torch.optim.Adam (single-tensor path): exp_avg.lerp_(g, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2);
// denom = sqrt(exp_avg_sq)/sqrt(bc2) + eps; p.addcdiv_(exp_avg, denom, value=-lr/bc1)
__device__ __forceinline__ void adam_update(float &p, float &m, float &v, float g, const AdamScalars &a) {
    m = m + a.one_minus_beta1 * (g - m);
    v = v * a.beta2;
    v = v + a.one_minus_beta2 * (g * g);
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p + (-a.step_size) * (m / denom);
}

// Gradient of E / beta' over the batch fused with their Adam step:
//   [dE | dbeta'] = DF^T [gs_b t_u | gv_b]   ([n_feat, B] x [B, k2 + 1]; W is written by the scatter kernel)
// fpb (<= 8) features per workgroup, fpb * (k2 + 1) <= 4 outputs per thread.  This kernel runs BESIDE the dense Adam sweep, where every memory round trip
// costs several microseconds: all its loads (its parameters and moments, its DF columns, W) are independent and issued
// up front, and there are n_feat / 8 workgroups of them in flight; the 26 MFLOP of products come out of the LDS.
// (A 128-feature MFMA tile per workgroup — 32 workgroups, 7 dependent k-steps — took 74 us there instead of 12.)
constexpr int kFeatPerBlock = 8;
__global__ __launch_bounds__(kVb) void vbpr_feat_adam_kernel(const float *__restrict__ DF, const float *__restrict__ W,
                                                             int n, int n_feat, int k2, int ldw, int fpb, float *E, float *mE,
                                                             float *vE, float *Bp, float *mBp, float *vBp, float lambda_e,
                                                             const AdamScalars a) {
    extern __shared__ float shm[];  // Wl[n][ldw] | dfs[fpb][n]
    float *Wl = shm, *dfs = shm + (size_t)n * ldw;
    const int tid = threadIdx.x;
    const int f0 = blockIdx.x * fpb;
    constexpr int kOutMax = 4;  // outputs per thread: fpb * (k2 + 1) <= kOutMax * kVb
    const int n_out = fpb * (k2 + 1);
    float *P[kOutMax], *M[kOutMax], *V[kOutMax];
    float p[kOutMax], m[kOutMax], v[kOutMax];
#pragma unroll
    for (int o = 0; o < kOutMax; ++o) {
        const int idx = tid + o * kVb;
        const int q = idx / (k2 + 1), c = idx % (k2 + 1), f = f0 + q;
        const bool live = idx < n_out && f < n_feat;
        P[o] = nullptr;
        if (live) {
            P[o] = c < k2 ? E + (size_t)f * k2 + c : Bp + f;
            M[o] = c < k2 ? mE + (size_t)f * k2 + c : mBp + f;
            V[o] = c < k2 ? vE + (size_t)f * k2 + c : vBp + f;
            p[o] = *P[o]; m[o] = *M[o]; v[o] = *V[o];
        }
    }
    for (int idx = tid; idx < n * ldw / 4; idx += kVb)
        reinterpret_cast<f32x4 *>(Wl)[idx] = reinterpret_cast<const f32x4 *>(W)[idx];
    for (int idx = tid; idx < fpb * n; idx += kVb) {
        const int b = idx / fpb, q = idx % fpb;
        dfs[q * n + b] = f0 + q < n_feat ? DF[(size_t)b * n_feat + f0 + q] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < kOutMax; ++o) {
        if (!P[o]) continue;
        const int idx = tid + o * kVb;
        const int q = idx / (k2 + 1), c = idx % (k2 + 1);
        float g = 0.f;
        for (int b = 0; b < n; ++b) g = fmaf(dfs[q * n + b], Wl[b * ldw + c], g);
        adam_update(p[o], m[o], v[o], g + lambda_e * p[o], a);
        *P[o] = p[o]; *M[o] = m[o]; *V[o] = v[o];
    }
}

// Dense Adam over the four row tables for the rows the current batch did NOT touch (zero gradient: the moments decay
// and the parameter moves, torch.optim.Adam's dense semantics).  One launch over all tables; 16-byte accesses.
struct SweepTable {
    float *p, *m, *v;
    const int32_t *stamp;
    int64_t begin;       // first unit of this table in the launch's global index space
    int units_per_row;   // float4 units (width / 4) when vec, floats (width) otherwise
    int vec;
};
struct SweepArgs {
    SweepTable tab[4];
    int64_t total;
    int32_t step;
};

__global__ __launch_bounds__(kVb) void adam_sweep_kernel(const SweepArgs s, const AdamScalars a) {
    for (int64_t i = (int64_t)blockIdx.x * kVb + threadIdx.x; i < s.total; i += (int64_t)gridDim.x * kVb) {
        int q = 0;
#pragma unroll
        for (int c = 1; c < 4; ++c) q += i >= s.tab[c].begin ? 1 : 0;
        const SweepTable &tb = s.tab[q];
        const int64_t local = i - tb.begin;
        const int32_t st = tb.stamp[local / tb.units_per_row];
        if (st == s.step || st == -s.step) continue;  // the batch's rows: vbpr_touched_adam_kernel
        if (tb.vec) {
            f32x4 p = reinterpret_cast<const f32x4 *>(tb.p)[local], m = reinterpret_cast<const f32x4 *>(tb.m)[local],
                  v = reinterpret_cast<const f32x4 *>(tb.v)[local];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], me = m[e], ve = v[e];
                adam_update(pe, me, ve, 0.f, a);
                p[e] = pe; m[e] = me; v[e] = ve;
            }
            reinterpret_cast<f32x4 *>(tb.p)[local] = p;
            reinterpret_cast<f32x4 *>(tb.m)[local] = m;
            reinterpret_cast<f32x4 *>(tb.v)[local] = v;
        } else {
            float pe = tb.p[local], me = tb.m[local], ve = tb.v[local];
            adam_update(pe, me, ve, 0.f, a);
            tb.p[local] = pe; tb.m[local] = me; tb.v[local] = ve;
        }
    }
}

// Adam step of the rows the batch touched, each distinct row exactly once: the first of the (up to 3 B) references
// that flips the row's stamp from +step to -step owns it.  Reads and clears the scattered gradient.
__global__ __launch_bounds__(kVb) void vbpr_touched_adam_kernel(const VbprTables t, const int32_t *__restrict__ bu,
                                                                const int32_t *__restrict__ bi,
                                                                const int32_t *__restrict__ bj, int32_t step,
                                                                float *mBi, float *vBi, float *mGu, float *vGu, float *mGi,
                                                                float *vGi, float *mTu, float *vTu, const AdamScalars a) {
    __shared__ int own[3];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t u = bu[b], i = bi[b], j = bj[b];
    if (tid == 0) own[0] = atomicCAS(t.stamp_u + u, step, -step) == step;
    if (tid == 1) own[1] = atomicCAS(t.stamp_i + i, step, -step) == step;
    __syncthreads();  // (i == j cannot happen for a valid triplet, but stay exact if it does: j claims after i)
    if (tid == 2) own[2] = atomicCAS(t.stamp_i + j, step, -step) == step;
    __syncthreads();
    auto row = [&](float *P, float *M, float *V, float *G, int64_t r, int width) {
        for (int q = tid; q < width; q += kVb) {
            const int64_t o = r * width + q;
            const float g = G[o];
            float p = P[o], m = M[o], v = V[o];
            adam_update(p, m, v, g, a);
            P[o] = p; M[o] = m; V[o] = v;
            G[o] = 0.f;
        }
    };
    if (own[0]) {
        row(t.Gu, mGu, vGu, t.gGu, u, t.k);
        row(t.Tu, mTu, vTu, t.gTu, u, t.k2);
    }
    if (own[1]) {
        row(t.Gi, mGi, vGi, t.gGi, i, t.k);
        row(t.Bi, mBi, vBi, t.gBi, i, 1);
    }
    if (own[2]) {
        row(t.Gi, mGi, vGi, t.gGi, j, t.k);
        row(t.Bi, mBi, vBi, t.gBi, j, 1);
    }
}

// theta_item = F E, visual_bias = F beta'  (recom_vbpr.py:132-133, :273-274)
__global__ __launch_bounds__(kVb) void vbpr_item_tables_kernel(const VbprTables t, float *__restrict__ theta_item,
                                                               float *__restrict__ visual_bias) {
    extern __shared__ float shm[];  // frow[n_feat] | red[kVb]
    float *frow = shm, *red = shm + t.n_feat;
    const int64_t item = blockIdx.x;
    const int tid = threadIdx.x;
    for (int f = tid; f < t.n_feat; f += kVb) frow[f] = t.F[item * t.n_feat + f];
    __syncthreads();
    const int k2 = t.k2, n_slices = max(1, kVb / k2), c = tid % k2, sl = tid / k2;
    float acc = 0.f;
    if (sl < n_slices)
        for (int f = sl; f < t.n_feat; f += n_slices) acc = fmaf(frow[f], t.E[(size_t)f * k2 + c], acc);
    red[tid] = sl < n_slices ? acc : 0.f;
    __syncthreads();
    if (tid < k2) {
        float s = 0.f;
        for (int q = 0; q < n_slices; ++q) s += red[q * k2 + tid];
        theta_item[item * k2 + tid] = s;
    }
    __syncthreads();
    float vb = 0.f;
    for (int f = tid; f < t.n_feat; f += kVb) vb = fmaf(frow[f], t.Bp[f], vb);
    red[tid] = vb;
    __syncthreads();
    for (int o = kVb / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) visual_bias[item] = red[0];
}

}  // namespace chip

using namespace chip;

struct cornac_hip_vbpr {
    int device = 0;
    int64_t n_users = 0, n_items = 0;
    int k = 0, k2 = 0, n_feat = 0;
    hipStream_t stream = nullptr;
    hipStream_t sweep_stream = nullptr;           // the dense Adam sweep runs beside the step's small kernels
    hipEvent_t ev_stamped = nullptr, ev_swept = nullptr;
    DevBuf<float> F, Bi, Gu, Gi, Tu, E, Bp;
    DevBuf<float> gBi, gGu, gGi, gTu;
    DevBuf<int32_t> stamp_u, stamp_i;
    bool sweep_pending = false;
    DevBuf<float> mBi, vBi, mGu, vGu, mGi, vGi, mTu, vTu, mE, vE, mBp, vBp;
    DevBuf<int32_t> bu, bi, bj;
    DevBuf<float> sX, vX, gS, gV, proj, DF, W;
    DevBuf<double> loss;
    int64_t step = 0;
};

static void vb_check(cornac_hip_vbpr_t h) {
    REQUIRE(h != nullptr, "VBPR handle is NULL");
    HIP_CHECK(hipSetDevice(h->device));
}

static VbprTables vb_tables(cornac_hip_vbpr_t h) {
    VbprTables t;
    t.F = h->F.p; t.Bi = h->Bi.p; t.Gu = h->Gu.p; t.Gi = h->Gi.p; t.Tu = h->Tu.p; t.E = h->E.p; t.Bp = h->Bp.p;
    t.gBi = h->gBi.p; t.gGu = h->gGu.p; t.gGi = h->gGi.p; t.gTu = h->gTu.p;
    t.stamp_u = h->stamp_u.p; t.stamp_i = h->stamp_i.p;
    t.n_users = h->n_users; t.n_items = h->n_items; t.k = h->k; t.k2 = h->k2; t.n_feat = h->n_feat;
    return t;
}

extern "C" {

int cornac_hip_vbpr_create(cornac_hip_vbpr_t *out, int device, int64_t n_users, int64_t n_items, int k, int k2,
                           int n_feat, const float *features) {
    return guarded([&] {
        REQUIRE(out != nullptr, "out handle pointer is NULL");
        *out = nullptr;
        REQUIRE(n_users > 0 && n_items > 0 && k > 0 && k2 > 0 && n_feat > 0, "sizes must be positive");
        REQUIRE(k2 <= kMaxK2, "k2 <= %d supported", kMaxK2);
        REQUIRE((size_t)(n_feat + kVb + k2) * sizeof(float) <= 160 * 1024, "n_feat too large for the LDS staging");
        REQUIRE(features != nullptr, "features is NULL");
        use_device(device);
        std::unique_ptr<cornac_hip_vbpr> h(new cornac_hip_vbpr());
        h->device = device; h->n_users = n_users; h->n_items = n_items; h->k = k; h->k2 = k2; h->n_feat = n_feat;
        HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&h->sweep_stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&h->ev_stamped, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&h->ev_swept, hipEventDisableTiming));
        h->stamp_u.alloc((size_t)2 * n_users);   // two sets, used by alternate steps: the sweep of step t still reads
        h->stamp_i.alloc((size_t)2 * n_items);   // its set while step t+1's rows are being stamped
        HIP_CHECK(hipMemsetAsync(h->stamp_u.p, 0, (size_t)2 * n_users * sizeof(int32_t), h->stream));
        HIP_CHECK(hipMemsetAsync(h->stamp_i.p, 0, (size_t)2 * n_items * sizeof(int32_t), h->stream));
        h->F.alloc((size_t)n_items * n_feat);
        h->F.upload(features, (size_t)n_items * n_feat, h->stream);
        struct Tab { DevBuf<float> *p, *m, *v, *g; size_t n; };
        const Tab tabs[] = {{&h->Bi, &h->mBi, &h->vBi, &h->gBi, (size_t)n_items},
                            {&h->Gu, &h->mGu, &h->vGu, &h->gGu, (size_t)n_users * k},
                            {&h->Gi, &h->mGi, &h->vGi, &h->gGi, (size_t)n_items * k},
                            {&h->Tu, &h->mTu, &h->vTu, &h->gTu, (size_t)n_users * k2},
                            {&h->E, &h->mE, &h->vE, nullptr, (size_t)n_feat * k2},
                            {&h->Bp, &h->mBp, &h->vBp, nullptr, (size_t)n_feat}};
        for (const Tab &t : tabs) {
            for (DevBuf<float> *b : {t.p, t.m, t.v, t.g}) {
                if (!b) continue;
                b->alloc(t.n);
                HIP_CHECK(hipMemsetAsync(b->p, 0, t.n * sizeof(float), h->stream));
            }
        }
        h->loss.alloc(1);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        *out = h.release();
    });
}

int cornac_hip_vbpr_destroy(cornac_hip_vbpr_t h) {
    return guarded([&] {
        if (!h) return;
        (void)hipSetDevice(h->device);
        if (h->sweep_stream) {
            (void)hipStreamSynchronize(h->sweep_stream);
            (void)hipStreamDestroy(h->sweep_stream);
        }
        if (h->stream) {
            (void)hipStreamSynchronize(h->stream);
            (void)hipStreamDestroy(h->stream);
        }
        if (h->ev_stamped) (void)hipEventDestroy(h->ev_stamped);
        if (h->ev_swept) (void)hipEventDestroy(h->ev_swept);
        delete h;
    });
}

int cornac_hip_vbpr_set_params(cornac_hip_vbpr_t h, const float *Bi, const float *Gu, const float *Gi, const float *Tu,
                               const float *E, const float *Bp) {
    return guarded([&] {
        vb_check(h);
        if (Bi) h->Bi.upload(Bi, h->Bi.n, h->stream);
        if (Gu) h->Gu.upload(Gu, h->Gu.n, h->stream);
        if (Gi) h->Gi.upload(Gi, h->Gi.n, h->stream);
        if (Tu) h->Tu.upload(Tu, h->Tu.n, h->stream);
        if (E) h->E.upload(E, h->E.n, h->stream);
        if (Bp) h->Bp.upload(Bp, h->Bp.n, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}

int cornac_hip_vbpr_get_params(cornac_hip_vbpr_t h, float *Bi, float *Gu, float *Gi, float *Tu, float *E, float *Bp) {
    return guarded([&] {
        vb_check(h);
        if (Bi) h->Bi.download(Bi, h->Bi.n, h->stream);
        if (Gu) h->Gu.download(Gu, h->Gu.n, h->stream);
        if (Gi) h->Gi.download(Gi, h->Gi.n, h->stream);
        if (Tu) h->Tu.download(Tu, h->Tu.n, h->stream);
        if (E) h->E.download(E, h->E.n, h->stream);
        if (Bp) h->Bp.download(Bp, h->Bp.n, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}

int cornac_hip_vbpr_fit_batches(cornac_hip_vbpr_t h, const int32_t *u, const int32_t *i, const int32_t *j,
                                int64_t n_total, int batch_size, float lr, float lambda_w, float lambda_b,
                                float lambda_e, double *sum_nll) {
    return guarded([&] {
        vb_check(h);
        REQUIRE(u && i && j && n_total >= 0 && batch_size > 0, "bad batch arguments");
        REQUIRE(((size_t)batch_size * (((size_t)h->k2 + 4) & ~(size_t)3) + (size_t)kFeatPerBlock * batch_size) * sizeof(float) <= 64 * 1024,
                "batch_size x k2 too large for the LDS staging of the feature-gradient kernel");
        for (int64_t s = 0; s < n_total; ++s)
            REQUIRE(u[s] >= 0 && u[s] < h->n_users && i[s] >= 0 && i[s] < h->n_items && j[s] >= 0 &&
                        j[s] < h->n_items, "triplet %lld is out of range", (long long)s);
        if (n_total == 0) return;
        h->bu.ensure((size_t)n_total); h->bi.ensure((size_t)n_total); h->bj.ensure((size_t)n_total);
        h->bu.upload(u, (size_t)n_total, h->stream);
        h->bi.upload(i, (size_t)n_total, h->stream);
        h->bj.upload(j, (size_t)n_total, h->stream);
        h->sX.ensure((size_t)batch_size); h->vX.ensure((size_t)batch_size * kFeatSlices);
        h->gS.ensure((size_t)batch_size); h->gV.ensure((size_t)batch_size);
        h->proj.ensure((size_t)batch_size * h->k2);
        const int ldw = (h->k2 + 1 + 3) & ~3;
        const int fpb = std::max(1, std::min(kFeatPerBlock, 4 * kVb / (h->k2 + 1)));
        h->W.ensure((size_t)batch_size * ldw);
        HIP_CHECK(hipMemsetAsync(h->loss.p, 0, sizeof(double), h->stream));
        const VbprTables t0 = vb_tables(h);
        const DeviceInfo &di = device_info(h->device);
        h->DF.ensure((size_t)batch_size * h->n_feat);
        // feature chunks of the split-K projection GEMM: a multiple of the k tile, ~2 workgroups per CU
        // (two k-tiles per workgroup measured best: 16 / 32 / 64 / 128 features per chunk -> 63.1 / 61.0 / 64.5 / 72.1 us per
        // step of the small-kernel chain at n_feat = 4096)
        int feat_chunk = std::max(2 * kBK, (h->n_feat + 2 * di.cus - 1) / (2 * di.cus));
        feat_chunk = prof_env_int("CORNAC_HIP_VBPR_PROJ_CHUNK", feat_chunk);  // (profile builds only, csrc/common.h)
        feat_chunk = (feat_chunk + kBK - 1) / kBK * kBK;
        const int n_chunks = (h->n_feat + feat_chunk - 1) / feat_chunk;
        // the dense sweep: one launch over the four row tables, 16-byte accesses where the row width allows
        SweepArgs sw;
        bool sw_user[4];
        {
            struct T { DevBuf<float> *p, *m, *v; DevBuf<int32_t> *stamp; int width; };
            const T tabs[4] = {{&h->Gi, &h->mGi, &h->vGi, &h->stamp_i, h->k}, {&h->Gu, &h->mGu, &h->vGu, &h->stamp_u, h->k},
                               {&h->Tu, &h->mTu, &h->vTu, &h->stamp_u, h->k2}, {&h->Bi, &h->mBi, &h->vBi, &h->stamp_i, 1}};
            for (int q = 0; q < 4; ++q) sw_user[q] = tabs[q].stamp == &h->stamp_u;
            int64_t at = 0;
            for (int q = 0; q < 4; ++q) {
                const bool vec = tabs[q].width % 4 == 0;
                sw.tab[q].p = tabs[q].p->p; sw.tab[q].m = tabs[q].m->p; sw.tab[q].v = tabs[q].v->p;
                sw.tab[q].stamp = tabs[q].stamp->p;
                sw.tab[q].begin = at;
                sw.tab[q].units_per_row = vec ? tabs[q].width / 4 : tabs[q].width;
                sw.tab[q].vec = vec ? 1 : 0;
                at += vec ? (int64_t)tabs[q].p->n / 4 : (int64_t)tabs[q].p->n;
            }
            sw.total = at;
        }
        // 7 of the 8 workgroup slots of a CU: the sweep is persistent (grid-stride) and would otherwise hold every wave
        // slot of the chip until it ends, and the kernels of the main stream could not even start beside it
        // (measured per step: 8 slots on ONE stream 99.8 us; two streams 7 slots 91.9, 6 slots 94.2, 5 slots 95.8)
        const int sweep_wg_per_cu = prof_env_int("CORNAC_HIP_VBPR_SWEEP_WGS", 7);
        const int sweep_grid = (int)std::min<int64_t>((sw.total + kVb - 1) / kVb, (int64_t)di.cus * sweep_wg_per_cu);
        const bool one_stream = prof_env_set("CORNAC_HIP_VBPR_ONE_STREAM");  // A/B switch (profile builds): everything in stream order
        const bool ext_events = !prof_env_set("CORNAC_HIP_VBPR_PLAIN_EVENTS");  // A/B switch (profile builds): hipEventRecord hand-overs
        for (int64_t b0 = 0; b0 < n_total; b0 += batch_size) {
            const int n = (int)std::min<int64_t>(batch_size, n_total - b0);
            ++h->step;
            REQUIRE(h->step < (int64_t(1) << 31), "step counter exceeds 31 bits");
            const int32_t step = (int32_t)h->step;
            const int par = (int)(h->step & 1);
            VbprTables t = t0;
            t.stamp_u = h->stamp_u.p + (size_t)par * h->n_users;
            t.stamp_i = h->stamp_i.p + (size_t)par * h->n_items;
            // torch computes these in Python doubles and hands float scalars to the kernels
            const double b1 = 0.9, b2 = 0.999;
            const double bc1 = 1.0 - std::pow(b1, (double)h->step), bc2 = 1.0 - std::pow(b2, (double)h->step);
            AdamScalars a;
            a.beta1 = (float)b1; a.beta2 = (float)b2;
            a.one_minus_beta1 = (float)(1.0 - b1); a.one_minus_beta2 = (float)(1.0 - b2);
            a.step_size = (float)((double)lr / bc1);
            a.bc2_sqrt = (float)std::sqrt(bc2);
            a.eps = 1e-8f;
            // (these two only read F, E, beta' and the batch's indices: they run beside the PREVIOUS step's sweep)
            hipLaunchKernelGGL(vbpr_featdiff_kernel, dim3(n, kFeatSlices), dim3(kVb), 0, h->stream, t, h->bu.p + b0, h->bi.p + b0,
                               h->bj.p + b0, h->DF.p, h->vX.p, h->proj.p, step);
            hipLaunchKernelGGL(vbpr_proj_kernel, dim3(n_chunks, (n + kBM - 1) / kBM, (h->k2 + kBN - 1) / kBN), dim3(kWb), 0,
                               h->stream, h->DF.p, h->E.p, n, h->n_feat, h->k2, feat_chunk, h->proj.p);
            // the score reads rows the previous step's sweep may still be updating
            if (h->sweep_pending) HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_swept, 0));
            hipLaunchKernelGGL(vbpr_score_kernel, dim3((n * 64 + kVb - 1) / kVb), dim3(kVb), 0, h->stream, t, h->bu.p + b0,
                               h->bi.p + b0, h->bj.p + b0, n, h->proj.p, h->sX.p);
            hipLaunchKernelGGL(vbpr_pair_grad_kernel, dim3(n), dim3(kVb), 0, h->stream, h->sX.p, h->vX.p, n, h->gS.p,
                               h->gV.p, h->loss.p);
            hipLaunchKernelGGL(vbpr_scatter_kernel, dim3(n), dim3(kVb), 0, h->stream, t, h->bu.p + b0, h->bi.p + b0,
                               h->bj.p + b0, n, h->gS.p, h->gV.p, h->proj.p, lambda_w, lambda_b, h->W.p, ldw);
            // the batch rows' own Adam step: a latency chain (index -> claim -> row) that took 27 us beside the sweep and
            // 7 us alone, so it runs before the sweep starts (W keeps the Tu rows the E / beta' step still needs)
            // (hipExtLaunchKernelGGL attaches the hand-over event to the kernel's own completion signal: a separate
            // hipEventRecord costs a barrier packet and ~7 us of idle stream per hand-over)
            if (one_stream || !ext_events)
                hipLaunchKernelGGL(vbpr_touched_adam_kernel, dim3(n), dim3(kVb), 0, h->stream, t, h->bu.p + b0, h->bi.p + b0,
                                   h->bj.p + b0, step, h->mBi.p, h->vBi.p, h->mGu.p, h->vGu.p, h->mGi.p, h->vGi.p,
                                   h->mTu.p, h->vTu.p, a);
            else
                hipExtLaunchKernelGGL(vbpr_touched_adam_kernel, dim3(n), dim3(kVb), 0, h->stream, nullptr, h->ev_stamped, 0,
                                      t, (const int32_t *)(h->bu.p + b0), (const int32_t *)(h->bi.p + b0),
                                      (const int32_t *)(h->bj.p + b0), step, h->mBi.p, h->vBi.p, h->mGu.p, h->vGu.p, h->mGi.p,
                                      h->vGi.p, h->mTu.p, h->vTu.p, a);
            // The sweep of this step starts here — behind the latency-bound score / gradient kernels, which a
            // bandwidth-saturating neighbour slows several-fold — and runs beside the E / beta' step and the next step's
            // feature gather and projection.
            sw.step = step;
            for (int q = 0; q < 4; ++q) sw.tab[q].stamp = (sw_user[q] ? t.stamp_u : t.stamp_i);
            if (one_stream) {
                hipLaunchKernelGGL(adam_sweep_kernel, dim3(sweep_grid), dim3(kVb), 0, h->stream, sw, a);
            } else {
                if (!ext_events) HIP_CHECK(hipEventRecord(h->ev_stamped, h->stream));
                HIP_CHECK(hipStreamWaitEvent(h->sweep_stream, h->ev_stamped, 0));
                if (ext_events) {
                    hipExtLaunchKernelGGL(adam_sweep_kernel, dim3(sweep_grid), dim3(kVb), 0, h->sweep_stream, nullptr,
                                          h->ev_swept, 0, sw, a);
                } else {
                    hipLaunchKernelGGL(adam_sweep_kernel, dim3(sweep_grid), dim3(kVb), 0, h->sweep_stream, sw, a);
                    HIP_CHECK(hipEventRecord(h->ev_swept, h->sweep_stream));
                }
                h->sweep_pending = true;
            }
            hipLaunchKernelGGL(vbpr_feat_adam_kernel, dim3((h->n_feat + fpb - 1) / fpb), dim3(kVb),
                               ((size_t)n * ldw + (size_t)fpb * n) * sizeof(float), h->stream, h->DF.p, h->W.p, n,
                               h->n_feat, h->k2, ldw, fpb, h->E.p, h->mE.p, h->vE.p, h->Bp.p, h->mBp.p, h->vBp.p, lambda_e, a);
        }
        // everything the caller does next runs on the main stream: it must see the last sweep
        if (h->sweep_pending) HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_swept, 0));
        h->sweep_pending = false;
        HIP_CHECK(hipGetLastError());
        double l = 0;
        HIP_CHECK(hipMemcpyAsync(&l, h->loss.p, sizeof l, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        if (sum_nll) *sum_nll = l;
    });
}

int cornac_hip_vbpr_item_tables(cornac_hip_vbpr_t h, float *theta_item, float *visual_bias) {
    return guarded([&] {
        vb_check(h);
        REQUIRE(theta_item && visual_bias, "NULL output");
        DevBuf<float> th, vb;
        th.alloc((size_t)h->n_items * h->k2);
        vb.alloc((size_t)h->n_items);
        const VbprTables t = vb_tables(h);
        hipLaunchKernelGGL(vbpr_item_tables_kernel, dim3((unsigned)h->n_items), dim3(kVb),
                           (size_t)(h->n_feat + kVb) * sizeof(float), h->stream, t, th.p, vb.p);
        HIP_CHECK(hipGetLastError());
        th.download(theta_item, th.n, h->stream);
        vb.download(visual_bias, vb.n, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}
}
