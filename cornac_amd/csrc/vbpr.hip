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
// Per Adam step t (batch of B triplets), on three streams of the handle.  The batches of a call are known up front, so
// every step looks ONE batch ahead: the dense sweep of step t leaves out the rows of batch t AND of batch t + 1, and the
// rows of batch t + 1 get their (gradient-free) step-t update from the small kernel that also updates batch t's rows.
// The score of step t + 1 therefore never waits for sweep t — the sweeps run back to back on their own stream and the
// small-kernel chain runs beside them (round 3: sweep t -> score t+1 -> ... -> touched t+1 -> sweep t+1 was ONE serial
// chain, 54 + 24 us of a 94 us step).
//   gather stream (one step ahead of the main stream; reads F and the index arrays only)
//   vbpr_featdiff_kernel  DF[t & 1][b] = F[i_b] - F[j_b] (the "auxiliary-feature gather"); stamps batch t's rows
//                         with t in stamp set t & 3 and clears proj[t & 1]
//   main stream
//   vbpr_proj_kernel      proj = DF E on the fp32 matrix cores (mfma_gemm.h), split over feature chunks
//   vbpr_score_kernel     s_b from the gathered rows and proj_b, v_b = DF[b] . beta' (one wave per triplet)
//   vbpr_pair_scatter_kernel  the reference's B x B broadcast objective -> gs_b, gv_b (one workgroup per b), which then
//                         scatters b's sparse row gradients with fp32 atomics and writes b's row of W = [gs t_u | gv]
//   vbpr_touched_adam_kernel  [waits for sweep t-1]  Adam step of the rows batch t touched (each distinct row once;
//                         consumes and clears its scattered gradient) + the gradient-free step-t update of the rows of
//                         batch t + 1 that batch t did not touch (each once)
//   vbpr_feat_adam_kernel gradient of E / beta' (DF^T x [gs t_u | gv] on the matrix cores) fused with their Adam step
//   sweep stream  [sweep t waits for touched t-1 and for the stamps of batch t + 1]
//   adam_sweep_kernel     dense Adam over Bi, Gu, Gi, Tu for every row NOT stamped with t or t + 1: their gradient is
//                         exactly zero (the L2 terms only cover the batch rows, recom_vbpr.py:251-253), so the sweep
//                         reads and writes p, m, v only — 24 bytes per parameter and step
// Why this is the reference's arithmetic: every row receives exactly one Adam update per step, computed from its state
// after the previous step, with the step's own scalars — by the sweep, by the batch's own update, or by the look-ahead
// update (the same adam_update(g = 0) as the sweep's) — and a row is only ever read for a score after its last update.
// HBM-bound by the dense Adam sweep (all parameters + two moments per step).
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

// Stamp sets.  Step s stamps its batch rows in set s & 3 with one of three codes, all of which mean "row of batch s":
//   s            stamped by the gather (vbpr_featdiff_kernel)
//   s | kStPre   the look-ahead update of step s - 1 ran (vbpr_touched_adam_kernel of step s - 1 claimed the row)
//   -s           the batch's own update of step s ran (vbpr_touched_adam_kernel of step s claimed the row)
// Four sets: the sweep of step s reads sets s & 3 and (s + 1) & 3 while batch s + 2 is being stamped.
constexpr int32_t kStPre = 1 << 30;
__device__ __forceinline__ bool stamp_is(int32_t v, int32_t s) { return v == s || v == (s | kStPre) || v == -s; }

// stage 0 (gather stream, one step ahead) — kFeatSlices workgroups per triplet: feature difference
// DF[b, :] = F[i_b] - F[j_b] (coalesced row reads, written once for the two feature GEMMs and the beta' product of the
// step).  Slice 0 also stamps the batch's rows with the step number and clears proj[b].  Reads nothing that training
// writes.
constexpr int kFeatSlices = 4;
__global__ __launch_bounds__(kVb) void vbpr_featdiff_kernel(const VbprTables t, const int32_t *__restrict__ bu,
                                                            const int32_t *__restrict__ bi,
                                                            const int32_t *__restrict__ bj, float *__restrict__ DF,
                                                            float *__restrict__ proj, int32_t step) {
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
    const int per = (((t.n_feat + kFeatSlices - 1) / kFeatSlices) + 3) & ~3;
    const int f_begin = sl * per, f_end = min(t.n_feat, (sl + 1) * per);
    if ((t.n_feat & 3) == 0) {
        for (int f = f_begin + 4 * tid; f < f_end; f += 4 * kVb) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(fi + f), y = *reinterpret_cast<const f32x4 *>(fj + f);
            *reinterpret_cast<f32x4 *>(df + f) = x - y;
        }
    } else {
        for (int f = f_begin + tid; f < f_end; f += kVb) df[f] = fi[f] - fj[f];
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

// stage 1c — s_b = b_i - b_j + <g_u, g_i - g_j> + <t_u, proj_b> and v_b = DF[b] . beta': one wave per triplet
__global__ __launch_bounds__(kVb) void vbpr_score_kernel(const VbprTables t, const int32_t *__restrict__ bu,
                                                         const int32_t *__restrict__ bi,
                                                         const int32_t *__restrict__ bj, int n,
                                                         const float *__restrict__ proj, const float *__restrict__ DF,
                                                         float *__restrict__ s_out, float *__restrict__ v_out) {
    const int b = (blockIdx.x * kVb + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (b >= n) return;
    const int64_t u = bu[b], i = bi[b], j = bj[b];
    const float *gu = t.Gu + u * t.k, *gi = t.Gi + i * t.k, *gj = t.Gi + j * t.k, *tu = t.Tu + u * t.k2;
    float part = 0.f;
    for (int q = lane; q < t.k; q += 64) part = fmaf(gu[q], gi[q] - gj[q], part);
    for (int q = lane; q < t.k2; q += 64) part = fmaf(tu[q], proj[(size_t)b * t.k2 + q], part);
    const float *df = DF + (int64_t)b * t.n_feat;
    float vb = 0.f;
    if ((t.n_feat & 3) == 0) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int f = 4 * lane; f < t.n_feat; f += 256) {
            const f32x4 d = *reinterpret_cast<const f32x4 *>(df + f), w = *reinterpret_cast<const f32x4 *>(t.Bp + f);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(d[e], w[e], acc[e]);
        }
        vb = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    } else {
        for (int f = lane; f < t.n_feat; f += 64) vb = fmaf(df[f], t.Bp[f], vb);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        part += __shfl_xor(part, o, 64);
        vb += __shfl_xor(vb, o, 64);
    }
    if (lane == 0) {
        s_out[b] = (t.Bi[i] - t.Bi[j]) + part;
        v_out[b] = vb;
    }
}

// stage 2 — the B x B broadcast objective and the sparse row gradients in one launch.  One workgroup per b, threads over
// a (strided): column b and row b of G in one pass give gs_b = sum_a G[a,b] and gv_b = sum_b' G[b,b'] (NLL = sum
// softplus(-X)) — both belong to triplet b, so the same workgroup goes on to scatter b's row gradients with fp32 atomics
// (duplicates inside a batch accumulate, like autograd's index_put accumulate) and to write b's row of the right-hand
// side of the E / beta' gradient GEMM, W[b, :] = [gs_b * t_u | gv_b | 0 padding].
__global__ __launch_bounds__(kVb) void vbpr_pair_scatter_kernel(const VbprTables t, const int32_t *__restrict__ bu,
                                                                const int32_t *__restrict__ bi,
                                                                const int32_t *__restrict__ bj, int n,
                                                                const float *__restrict__ s, const float *__restrict__ v,
                                                                const float *__restrict__ proj, float lambda_w,
                                                                float lambda_b, float *__restrict__ W, int ldw,
                                                                double *__restrict__ loss_acc) {
    __shared__ float red_c[kVb / 64], red_r[kVb / 64], sh_g[2];
    __shared__ double red_n[kVb / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t u = bu[b], i = bi[b], j = bj[b];
    const int k2 = t.k2;
    // the batch rows this workgroup needs after the reduction: requested before it
    const float *gu = t.Gu + u * t.k, *gi = t.Gi + i * t.k, *gj = t.Gi + j * t.k, *tu = t.Tu + u * k2;
    const float sb = s[b], vb = v[b];
    float col = 0.f, row = 0.f;
    double nll = 0.0;
    for (int a = tid; a < n; a += kVb) {
        const float X = sb + v[a];                 // X[a, b]
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
        sh_g[0] = c;
        sh_g[1] = r;
        atomicAdd(loss_acc, l);
    }
    __syncthreads();
    const float g = sh_g[0], gvb = sh_g[1];
    for (int q = tid; q < ldw; q += kVb)
        W[(size_t)b * ldw + q] = q < k2 ? g * tu[q] : (q == k2 ? gvb : 0.f);
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
//   [dE | dbeta'] = DF^T [gs_b t_u | gv_b]   ([n_feat, B] x [B, k2 + 1]; W is written by vbpr_pair_scatter_kernel)
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
    const int32_t *stamp, *stamp_next;  // the row's stamps in the sets of step and step + 1 (the same set when there is no next batch)
    int64_t begin;       // first unit of this table in the launch's global index space
    int units_per_row;   // float4 units (width / 4) when vec, floats (width) otherwise
    int row_shift;       // log2(units_per_row) when that is a power of two, else -1 (a 64-bit division per unit otherwise)
    int vec;
};
struct SweepArgs {
    SweepTable tab[4];
    int64_t total;
    int32_t step, step_next;  // step_next = step when the call has no further batch
};

// One 16-byte unit per thread and iteration.  (Two units per thread with half the workgroups — the same bytes in flight
// from half the waves — was measured: the sweep alone slows from 52 to 70 us at 3 workgroups per CU and the step is no
// faster at any workgroup count, profiles/r04_vbpr_ab_sweep2unit.log: the sweep and the latency-bound kernels of the
// other streams share one memory system, and what one gains the other loses.)
__global__ __launch_bounds__(kVb) void adam_sweep_kernel(const SweepArgs s, const AdamScalars a) {
    for (int64_t i = (int64_t)blockIdx.x * kVb + threadIdx.x; i < s.total; i += (int64_t)gridDim.x * kVb) {
        int q = 0;
#pragma unroll
        for (int c = 1; c < 4; ++c) q += i >= s.tab[c].begin ? 1 : 0;
        const SweepTable &tb = s.tab[q];
        const int64_t local = i - tb.begin;
        const int64_t row = tb.row_shift >= 0 ? local >> tb.row_shift : local / tb.units_per_row;
        // rows of this step's batch and of the next one: vbpr_touched_adam_kernel
        if (stamp_is(tb.stamp[row], s.step) || stamp_is(tb.stamp_next[row], s.step_next)) continue;
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

// Workgroups [0, n): Adam step of the rows batch `step` touched, each distinct row exactly once — the first of the
// (up to 3 B) references that flips the row's stamp from step (or step | kStPre) to -step owns it; reads and clears the
// scattered gradient.  Workgroups [n, n + n_next): the rows of batch step + 1 that batch `step` did not touch get this
// step's gradient-free update here (the sweep of this step leaves them out), each once: claimed by flipping the stamp in
// the next set from step + 1 to (step + 1) | kStPre.
struct VbprMoments {
    float *mBi, *vBi, *mGu, *vGu, *mGi, *vGi, *mTu, *vTu;
};
__global__ __launch_bounds__(kVb) void vbpr_touched_adam_kernel(const VbprTables t, const int32_t *__restrict__ bu,
                                                                const int32_t *__restrict__ bi,
                                                                const int32_t *__restrict__ bj, int n, int32_t step,
                                                                int32_t *stamp_u_next, int32_t *stamp_i_next,
                                                                const VbprMoments mo, const AdamScalars a) {
    __shared__ int own[3];
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool ahead = b >= n;  // (the index arrays of batch step + 1 follow batch step's: b indexes both)
    const int64_t u = bu[b], i = bi[b], j = bj[b];
    auto claim = [&](int32_t *cur, int32_t *next, int64_t r) -> int {
        if (!ahead) {
            if (atomicCAS(cur + r, step, -step) == step) return 1;
            return atomicCAS(cur + r, step | kStPre, -step) == (step | kStPre);
        }
        if (stamp_is(cur[r], step)) return 0;  // in this step's batch too: updated with its gradient above
        return atomicCAS(next + r, step + 1, (step + 1) | kStPre) == step + 1;
    };
    // the three claims are independent atomics (one round trip, not three: this kernel is a latency chain beside the
    // bandwidth-saturating sweep); i == j cannot happen for a valid triplet, and if it does the row is claimed once
    if (tid == 0) own[0] = claim(t.stamp_u, stamp_u_next, u);
    if (tid == 64) own[1] = claim(t.stamp_i, stamp_i_next, i);
    if (tid == 128) own[2] = j != i ? claim(t.stamp_i, stamp_i_next, j) : 0;
    __syncthreads();
    auto row = [&](float *P, float *M, float *V, float *G, int64_t r, int width) {
        for (int q = tid; q < width; q += kVb) {
            const int64_t o = r * width + q;
            float g = 0.f;
            if (!ahead) {
                g = G[o];
                G[o] = 0.f;
            }
            float p = P[o], m = M[o], v = V[o];
            adam_update(p, m, v, g, a);
            P[o] = p; M[o] = m; V[o] = v;
        }
    };
    if (own[0]) {
        row(t.Gu, mo.mGu, mo.vGu, t.gGu, u, t.k);
        row(t.Tu, mo.mTu, mo.vTu, t.gTu, u, t.k2);
    }
    if (own[1]) {
        row(t.Gi, mo.mGi, mo.vGi, t.gGi, i, t.k);
        row(t.Bi, mo.mBi, mo.vBi, t.gBi, i, 1);
    }
    if (own[2]) {
        row(t.Gi, mo.mGi, mo.vGi, t.gGi, j, t.k);
        row(t.Bi, mo.mBi, mo.vBi, t.gBi, j, 1);
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
    hipStream_t sweep_stream = nullptr;           // the dense Adam sweeps, back to back beside the steps' small kernels
    hipStream_t gather_stream = nullptr;          // feature gather + row stamps, one step ahead
    hipStream_t row_stream = nullptr;             // the batch rows' own update, beside the E / beta' step of the main stream
    hipEvent_t ev_pair = nullptr;                 // pair-gradient + scatter of the step done
    // hand-overs (attached to the producing kernel's completion signal, hipExtLaunchKernelGGL): [step & 1]
    hipEvent_t ev_df[2] = {nullptr, nullptr};     // featdiff of the step done: DF / proj / the stamps of its batch are ready
    hipEvent_t ev_fa[2] = {nullptr, nullptr};     // feat_adam of the step done: its DF / proj buffers are free again
    hipEvent_t ev_swept[2] = {nullptr, nullptr};  // sweep of the step done
    hipEvent_t ev_touched = nullptr;              // the batch rows' own update of the step done
    DevBuf<float> F, Bi, Gu, Gi, Tu, E, Bp;
    DevBuf<float> gBi, gGu, gGi, gTu;
    DevBuf<int32_t> stamp_u, stamp_i;
    DevBuf<float> mBi, vBi, mGu, vGu, mGi, vGi, mTu, vTu, mE, vE, mBp, vBp;
    DevBuf<int32_t> bu, bi, bj;
    DevBuf<float> sX, vX, proj, DF, W;
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
        HIP_CHECK(hipStreamCreateWithFlags(&h->gather_stream, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&h->row_stream, hipStreamNonBlocking));
        for (hipEvent_t *e : {&h->ev_df[0], &h->ev_df[1], &h->ev_fa[0], &h->ev_fa[1], &h->ev_swept[0], &h->ev_swept[1],
                              &h->ev_touched, &h->ev_pair})
            HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        h->stamp_u.alloc((size_t)4 * n_users);   // four sets, set s & 3 for step s (see kStPre)
        h->stamp_i.alloc((size_t)4 * n_items);
        HIP_CHECK(hipMemsetAsync(h->stamp_u.p, 0, (size_t)4 * n_users * sizeof(int32_t), h->stream));
        HIP_CHECK(hipMemsetAsync(h->stamp_i.p, 0, (size_t)4 * n_items * sizeof(int32_t), h->stream));
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
        for (hipStream_t st : {h->gather_stream, h->sweep_stream, h->row_stream, h->stream}) {
            if (!st) continue;
            (void)hipStreamSynchronize(st);
            (void)hipStreamDestroy(st);
        }
        for (hipEvent_t e : {h->ev_df[0], h->ev_df[1], h->ev_fa[0], h->ev_fa[1], h->ev_swept[0], h->ev_swept[1], h->ev_touched,
                             h->ev_pair})
            if (e) (void)hipEventDestroy(e);
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
        h->sX.ensure((size_t)batch_size); h->vX.ensure((size_t)batch_size);
        h->proj.ensure((size_t)2 * batch_size * h->k2);          // [step & 1]
        const int ldw = (h->k2 + 1 + 3) & ~3;
        const int fpb = std::max(1, std::min(kFeatPerBlock, 4 * kVb / (h->k2 + 1)));
        h->W.ensure((size_t)batch_size * ldw);
        HIP_CHECK(hipMemsetAsync(h->loss.p, 0, sizeof(double), h->stream));
        const VbprTables t0 = vb_tables(h);
        const DeviceInfo &di = device_info(h->device);
        h->DF.ensure((size_t)2 * batch_size * h->n_feat);        // [step & 1]
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
                sw.tab[q].stamp = sw.tab[q].stamp_next = tabs[q].stamp->p;
                sw.tab[q].begin = at;
                sw.tab[q].units_per_row = vec ? tabs[q].width / 4 : tabs[q].width;
                {
                    const int upr = sw.tab[q].units_per_row;
                    sw.tab[q].row_shift = (upr & (upr - 1)) == 0 ? __builtin_ctz((unsigned)upr) : -1;
                }
                sw.tab[q].vec = vec ? 1 : 0;
                at += vec ? (int64_t)tabs[q].p->n / 4 : (int64_t)tabs[q].p->n;
            }
            sw.total = at;
        }
        // 5 of the 8 workgroup slots of a CU: the sweep is persistent (grid-stride) and would otherwise hold every wave
        // slot of the chip until it ends, and the kernels of the other streams could not even start beside it; every slot
        // it gives up speeds the latency-bound chain up more than it slows the sweep, down to 5 (per step: 8 slots 90.8 us,
        // 7: 80.8, 6: 76.7, 5: 72.4; profiles/r04_vbpr_ab_rowstream.log)
        const int sweep_wg_per_cu = prof_env_int("CORNAC_HIP_VBPR_SWEEP_WGS", 5);
        const int sweep_grid = (int)std::min<int64_t>((sw.total + kVb - 1) / kVb, (int64_t)di.cus * sweep_wg_per_cu);
        // A/B switch (profile builds): no look-ahead — the sweep leaves out its own batch only and the next score waits for it
        const bool look_ahead = !prof_env_set("CORNAC_HIP_VBPR_NO_LOOKAHEAD");
        const bool split_rows = !prof_env_set("CORNAC_HIP_VBPR_ROWS_ON_MAIN");  // A/B switch: the batch rows' update in stream order
        const VbprMoments mo = {h->mBi.p, h->vBi.p, h->mGu.p, h->vGu.p, h->mGi.p, h->vGi.p, h->mTu.p, h->vTu.p};
        const int64_t n_steps = (n_total + batch_size - 1) / batch_size;
        REQUIRE(h->step + n_steps < (int64_t(1) << 30), "step counter exceeds 30 bits");
        auto set_of = [&](VbprTables &t, int64_t step) {
            t.stamp_u = h->stamp_u.p + (size_t)(step & 3) * h->n_users;
            t.stamp_i = h->stamp_i.p + (size_t)(step & 3) * h->n_items;
        };
        // the gather of the batch at b0 (step number `step`): waits until the step that last used its buffers has let go
        auto enqueue_gather = [&](int64_t b0, int64_t step, bool wait_free) {
            const int n = (int)std::min<int64_t>(batch_size, n_total - b0), par = (int)(step & 1);
            VbprTables t = t0;
            set_of(t, step);
            if (wait_free) HIP_CHECK(hipStreamWaitEvent(h->gather_stream, h->ev_fa[par], 0));
            hipExtLaunchKernelGGL(vbpr_featdiff_kernel, dim3(n, kFeatSlices), dim3(kVb), 0, h->gather_stream, nullptr,
                                  h->ev_df[par], 0, t, (const int32_t *)(h->bu.p + b0), (const int32_t *)(h->bi.p + b0),
                                  (const int32_t *)(h->bj.p + b0), h->DF.p + (size_t)par * batch_size * h->n_feat,
                                  h->proj.p + (size_t)par * batch_size * h->k2, (int32_t)step);
        };
        // the index arrays are read by all three streams: uploaded before anything is enqueued
        HIP_CHECK(hipStreamSynchronize(h->stream));
        // (the previous call ended with every stream drained and every row swept: nothing to wait for at the first step)
        enqueue_gather(0, h->step + 1, false);
        HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_df[(h->step + 1) & 1], 0));
        int64_t first = h->step + 1;
        for (int64_t b0 = 0; b0 < n_total; b0 += batch_size) {
            const int n = (int)std::min<int64_t>(batch_size, n_total - b0);
            ++h->step;
            const int32_t step = (int32_t)h->step;
            const int par = (int)(h->step & 1);
            const bool has_next = b0 + batch_size < n_total;
            const bool ahead = has_next && look_ahead;
            const int n_next = ahead ? (int)std::min<int64_t>(batch_size, n_total - b0 - batch_size) : 0;
            VbprTables t = t0, t_next = t0;
            set_of(t, h->step);
            set_of(t_next, ahead ? h->step + 1 : h->step);
            float *DF = h->DF.p + (size_t)par * batch_size * h->n_feat, *proj = h->proj.p + (size_t)par * batch_size * h->k2;
            // torch computes these in Python doubles and hands float scalars to the kernels
            const double b1 = 0.9, b2 = 0.999;
            const double bc1 = 1.0 - std::pow(b1, (double)h->step), bc2 = 1.0 - std::pow(b2, (double)h->step);
            AdamScalars a;
            a.beta1 = (float)b1; a.beta2 = (float)b2;
            a.one_minus_beta1 = (float)(1.0 - b1); a.one_minus_beta2 = (float)(1.0 - b2);
            a.step_size = (float)((double)lr / bc1);
            a.bc2_sqrt = (float)std::sqrt(bc2);
            a.eps = 1e-8f;
            // gather stream: the next batch's feature rows and stamps (its buffers were last used by step - 1)
            if (has_next) enqueue_gather(b0 + batch_size, h->step + 1, h->step > first);
            // sweep stream: this step's sweep — behind the previous sweep, the previous batch rows' own update, and the
            // stamps it tests (this batch's: gathered before the next batch's on the same stream)
            sw.step = step;
            sw.step_next = ahead ? step + 1 : step;
            for (int q = 0; q < 4; ++q) {
                sw.tab[q].stamp = sw_user[q] ? t.stamp_u : t.stamp_i;
                sw.tab[q].stamp_next = sw_user[q] ? t_next.stamp_u : t_next.stamp_i;
            }
            if (look_ahead) {
                HIP_CHECK(hipStreamWaitEvent(h->sweep_stream, h->ev_df[has_next ? par ^ 1 : par], 0));
                if (h->step > first) HIP_CHECK(hipStreamWaitEvent(h->sweep_stream, h->ev_touched, 0));
                hipExtLaunchKernelGGL(adam_sweep_kernel, dim3(sweep_grid), dim3(kVb), 0, h->sweep_stream, nullptr,
                                      h->ev_swept[par], 0, sw, a);
            }
            // main stream: the E / beta' chain  proj -> score -> pair gradient + scatter -> E / beta' step -> next proj
            if (h->step > first) HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_df[par], 0));  // (the first step waited above)
            hipLaunchKernelGGL(vbpr_proj_kernel, dim3(n_chunks, (n + kBM - 1) / kBM, (h->k2 + kBN - 1) / kBN), dim3(kWb), 0,
                               h->stream, DF, h->E.p, n, h->n_feat, h->k2, feat_chunk, proj);
            // the score reads the batch's rows: after the previous step's update of them (row stream); without the look-ahead
            // also after the previous step's sweep, which may still be updating them
            if (h->step > first) {
                if (split_rows) HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_touched, 0));
                if (!look_ahead) HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_swept[par ^ 1], 0));
            }
            hipLaunchKernelGGL(vbpr_score_kernel, dim3((n * 64 + kVb - 1) / kVb), dim3(kVb), 0, h->stream, t, h->bu.p + b0,
                               h->bi.p + b0, h->bj.p + b0, n, proj, DF, h->sX.p, h->vX.p);
            hipExtLaunchKernelGGL(vbpr_pair_scatter_kernel, dim3(n), dim3(kVb), 0, h->stream, nullptr, h->ev_pair, 0, t,
                                  (const int32_t *)(h->bu.p + b0), (const int32_t *)(h->bi.p + b0),
                                  (const int32_t *)(h->bj.p + b0), n, (const float *)h->sX.p, (const float *)h->vX.p,
                                  (const float *)proj, lambda_w, lambda_b, h->W.p, ldw, h->loss.p);
            // row stream (or the main stream): the batch rows' own update + the look-ahead update of the next batch's rows —
            // the latter were last written by the PREVIOUS step's sweep, and their stamps come from the next batch's gather.
            // It is a latency chain (index -> claim -> row) that takes 22 us beside the sweep: on its own stream it runs
            // beside the E / beta' step instead of in front of it.
            hipStream_t rs = split_rows ? h->row_stream : h->stream;
            if (split_rows) HIP_CHECK(hipStreamWaitEvent(rs, h->ev_pair, 0));
            if (look_ahead) {
                if (h->step > first) HIP_CHECK(hipStreamWaitEvent(rs, h->ev_swept[par ^ 1], 0));
                if (has_next) HIP_CHECK(hipStreamWaitEvent(rs, h->ev_df[par ^ 1], 0));
            }
            hipExtLaunchKernelGGL(vbpr_touched_adam_kernel, dim3(n + n_next), dim3(kVb), 0, rs, nullptr, h->ev_touched,
                                  0, t, (const int32_t *)(h->bu.p + b0), (const int32_t *)(h->bi.p + b0),
                                  (const int32_t *)(h->bj.p + b0), n, step, t_next.stamp_u, t_next.stamp_i, mo, a);
            if (!look_ahead) {  // round 3's order: the sweep starts behind the batch rows' update, on its own stream
                HIP_CHECK(hipStreamWaitEvent(h->sweep_stream, h->ev_touched, 0));
                hipExtLaunchKernelGGL(adam_sweep_kernel, dim3(sweep_grid), dim3(kVb), 0, h->sweep_stream, nullptr,
                                      h->ev_swept[par], 0, sw, a);
            }
            hipExtLaunchKernelGGL(vbpr_feat_adam_kernel, dim3((h->n_feat + fpb - 1) / fpb), dim3(kVb),
                                  ((size_t)n * ldw + (size_t)fpb * n) * sizeof(float), h->stream, nullptr, h->ev_fa[par], 0,
                                  (const float *)DF, (const float *)h->W.p, n, h->n_feat, h->k2, ldw, fpb, h->E.p, h->mE.p,
                                  h->vE.p, h->Bp.p, h->mBp.p, h->vBp.p, lambda_e, a);
        }
        // everything the caller does next runs on the main stream: it must see the last row update and the last sweep
        HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_touched, 0));
        HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_swept[h->step & 1], 0));
        HIP_CHECK(hipGetLastError());
        double l = 0;
        HIP_CHECK(hipMemcpyAsync(&l, h->loss.p, sizeof l, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        HIP_CHECK(hipStreamSynchronize(h->gather_stream));
        HIP_CHECK(hipStreamSynchronize(h->sweep_stream));
        HIP_CHECK(hipStreamSynchronize(h->row_stream));
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
