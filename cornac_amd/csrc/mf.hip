// Biased matrix factorisation SGD on MI355X (gfx950).
//
// Replaces backend_cpu.fit_sgd (cornac/models/mf/backend_cpu.pyx:35-97): per epoch, for every
// rating in stored COO order: pred = mu + Bu[u] + Bi[i] + <U[u], V[i]>, err = r - pred, in-place
// SGD on the two rows and two biases, loss = 0.5 * sum(err^2).
//
//   deterministic — the seeded (1-thread) reference order, executed as a level schedule of the
//                   row-conflict DAG (built once: the COO order never changes between epochs);
//                   float expression order identical to the reference => bit-identical factors.
//   hogwild       — the reference's `prange(..., schedule='static')` racy path: a wave takes 64
//                   consecutive ratings (coalesced int64/float reads), G lanes per rating gather the
//                   two rows with 16-byte loads and scatter fp32 atomic updates.
#include <algorithm>
#include <cmath>
#include <queue>

#include "common.h"
#include "sgd_device.h"

namespace chip {

template <int G>
__global__ __launch_bounds__(kBlock) void mf_det_level_kernel(const int32_t *__restrict__ ou,
                                                              const int32_t *__restrict__ oi,
                                                              const float *__restrict__ orat, int64_t off, int cnt,
                                                              float *U, float *V, float *Bu, float *Bi, int k, float lr,
                                                              float reg, float mu, int use_bias,
                                                              double *__restrict__ loss_acc) {
    const int gid = (blockIdx.x * kBlock + threadIdx.x) / G;
    const int lg = threadIdx.x & (G - 1);
    const bool active = gid < cnt;
    const int64_t t = off + (active ? gid : cnt - 1);
    const int32_t u = ou[t], i = oi[t];
    const float r = orat[t];
    float *pu = U + (size_t)u * k, *pi = V + (size_t)i * k;
    float pred = mu + Bu[u] + Bi[i];
    for (int base = 0; base < k; base += G) {
        const int f = base + lg;
        float p = 0.f;
        if (f < k) p = pu[f] * pi[f];
        pred = ordered_lane_sum<G>(pred, p, min(G, k - base));
    }
    const float err = r - pred;
    if (active) {
        for (int f = lg; f < k; f += G) {
            const float uf = pu[f], vf = pi[f];
            pu[f] = uf + lr * (err * vf - reg * uf);
            pi[f] = vf + lr * (err * uf - reg * vf);
        }
        if (lg == 0 && use_bias) {
            const float bu = Bu[u], bi = Bi[i];
            Bu[u] = bu + lr * (err - reg * bu);
            Bi[i] = bi + lr * (err - reg * bi);
        }
    }
    // loss: fp64 partial sums (the reference sums err^2 into a float sequentially)
    double e2 = (active && lg == 0) ? (double)err * (double)err : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o, kWave);
    if (lane_id() == 0 && e2 != 0.0) atomicAdd(loss_acc, e2);
}

// Deterministic mode WITHOUT one launch per level: a dataflow ("chain") kernel.
//
// The level schedule above costs a kernel launch per link of the longest row-dependency chain (255 k links for the
// Netflix-shape epoch: 304 s).  Here the whole epoch is ONE persistent launch in which every rating waits for exactly
// what the sequential loop (backend_cpu.pyx:58-83) would have finished before it:
//   * every row of the OWNED side (users or items — the host picks the side whose ratings are adjacent in the stored
//     order, else the one with the longer hottest chain) belongs to one wave, and that wave applies the row's ratings
//     in their stored order: the row is read and written by one wave only — plain loads / stores;
//   * a row of the other (SHARED) side can be needed by many waves: `ver[row]` counts its finished updates, `cseq[t]` is
//     the position of rating t among that row's ratings; a rating runs when ver[row] == cseq[t], and afterwards the wave
//     drains its stores and publishes ver[row] = cseq[t] + 1.  Shared rows, their biases and the counters travel with
//     system-scope (sc0 sc1) loads and stores — they bypass the non-coherent per-XCD L2s, the valid cross-XCD form
//     without fences (MI355X_MICROARCH.md, inter-workgroup visibility);
//   * a wave never blocks on one rating: its lanes hold the cursors of 64 of its rows at a time, all 64 next ratings are
//     polled with one vector load of the counters, the ready ones are executed one after the other (each wave-wide),
//     the chunk is polled again while anything in it moves, then the next 64 rows — cyclically over all its rows.
//     A stored order in which a row's early rating waits for another row's late one (items in random order inside
//     a user's run) therefore stalls that row only, not the wave: measured 259 s -> see DESIGN.md 1.1 for the in-order
//     form of this kernel.
// Progress: the globally first unfinished rating is the next rating of its owned row, every earlier rating of its
// shared row is finished, and its wave visits all its rows cyclically without ever blocking — it runs.  No wave waits
// for a wave that is not resident: nothing blocks at all.  The float expression tree is the level kernel's (the
// reference's), every row sees its updates in the stored order, so the result is bit-identical to the sequential loop.
// A wave that makes no progress for `wait_bound_ticks` (a bug, never contention) raises `abort` and every wave leaves.
struct MfChainArgs {
    const int64_t *wrow_ptr;        // [W + 1] rows of wave w = [wrow_ptr[w], wrow_ptr[w + 1])
    const int32_t *row_id;          // owned-side id of a row
    const int64_t *row_end;         // end of the row's ratings in csid / cseq / cr
    int64_t *row_cur;               // next rating of the row (reset to the row's begin every epoch)
    const int32_t *csid, *cseq;     // shared-side id of a rating, its position among that row's ratings
    const float *cr;
    unsigned int *ver;              // [rows of the shared side] finished updates of the row in this epoch
    unsigned int *abort;            // [8]: [0] a wave gave up, [1..5] who / what (diagnostics)
    long long wait_bound_ticks;     // of the 100 MHz real-time counter
    float *U, *V, *Bu, *Bi;
    double *loss_acc;
    int k, use_bias;
    float lr, reg, mu;
};

__device__ __forceinline__ float load_f32_sys(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void store_f32_sys(float *p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int R, bool OWN_USER>  // k <= 64 R
__global__ __launch_bounds__(kBlock) void mf_det_chain_kernel(const MfChainArgs a) {
    const int lane = lane_id();
    const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t q0 = a.wrow_ptr[w], q1 = a.wrow_ptr[w + 1];
    double loss = 0.0;
    unsigned long long t_idle = 0;  // real-time stamp of the first fruitless sweep in a row (0: progressing)
    unsigned int idle_sweeps = 0;
    bool give_up = false;
    while (!give_up) {
        bool unfinished = false, progressed = false;
        for (int64_t base = q0; base < q1; base += kWave) {
            const int64_t q = base + lane;
            const bool mine = q < q1;
            int64_t cur = mine ? a.row_cur[q] : 0;
            const int64_t rend = mine ? a.row_end[q] : 0;
            const int32_t oid = mine ? a.row_id[q] : 0;
            for (;;) {
                // ---- poll the next rating of up to 64 rows at once ----
                const bool has = mine && cur < rend;
                const int32_t sid = has ? a.csid[cur] : 0;
                const unsigned int seq = has ? (unsigned int)a.cseq[cur] : 0u;
                const unsigned int v = has ? __hip_atomic_load(a.ver + sid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : ~0u;
                unsigned long long ready = __ballot(has && v == seq);
                if (!ready) break;
                asm volatile("" ::: "memory");  // (compiler: the row loads below stay behind the poll; the hardware issues in order)
                progressed = true;
                const float rr = has ? a.cr[cur] : 0.f;
                while (ready) {
                    const int l = __builtin_ctzll(ready);
                    ready &= ready - 1;
                    const int32_t o = __builtin_amdgcn_readlane(oid, l), s = __builtin_amdgcn_readlane(sid, l);
                    const unsigned int sq = (unsigned int)__builtin_amdgcn_readlane((int)seq, l);
                    const float r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rr), l));
                    const int32_t u = OWN_USER ? o : s, i = OWN_USER ? s : o;
                    float *pu = a.U + (size_t)u * a.k, *pi = a.V + (size_t)i * a.k;
                    float uf[R], vf[R];
#pragma unroll
                    for (int c = 0; c < R; ++c) {
                        const int f = lane + kWave * c;
                        uf[c] = f < a.k ? (OWN_USER ? pu[f] : load_f32_sys(pu + f)) : 0.f;
                        vf[c] = f < a.k ? (OWN_USER ? load_f32_sys(pi + f) : pi[f]) : 0.f;
                    }
                    const float bu = OWN_USER ? a.Bu[u] : load_f32_sys(a.Bu + u), bi = OWN_USER ? load_f32_sys(a.Bi + i) : a.Bi[i];
                    // ---- the reference's expression tree (mf_det_level_kernel) ----
                    float pred = a.mu + bu + bi;
#pragma unroll
                    for (int c = 0; c < R; ++c) {
                        if (kWave * c < a.k) pred = ordered_lane_sum<kWave>(pred, uf[c] * vf[c], min(kWave, a.k - kWave * c));
                    }
                    const float err = r - pred;
#pragma unroll
                    for (int c = 0; c < R; ++c) {
                        const int f = lane + kWave * c;
                        if (f < a.k) {
                            const float un = uf[c] + a.lr * (err * vf[c] - a.reg * uf[c]), vn = vf[c] + a.lr * (err * uf[c] - a.reg * vf[c]);
                            if (OWN_USER) {
                                pu[f] = un;
                                store_f32_sys(pi + f, vn);
                            } else {
                                store_f32_sys(pu + f, un);
                                pi[f] = vn;
                            }
                        }
                    }
                    if (lane == 0 && a.use_bias) {
                        const float bun = bu + a.lr * (err - a.reg * bu), bin = bi + a.lr * (err - a.reg * bi);
                        if (OWN_USER) {
                            a.Bu[u] = bun;
                            store_f32_sys(a.Bi + i, bin);
                        } else {
                            store_f32_sys(a.Bu + u, bun);
                            a.Bi[i] = bin;
                        }
                    }
                    if (lane == 0) loss += (double)err * (double)err;
                    // ---- publish: every store of this wave has left before the counter moves ----
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(a.ver + s, sq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (lane == l) ++cur;
                }
            }
            if (mine) a.row_cur[q] = cur;
            unfinished = unfinished || __ballot(mine && cur < rend) != 0ull;
        }
        if (!unfinished) break;
        if (progressed) {
            t_idle = 0;
            idle_sweeps = 0;
            continue;
        }
        // nothing of this wave can run yet: back off (longer the longer it lasts) and watch the bound
        ++idle_sweeps;
        if (idle_sweeps < 8) __builtin_amdgcn_s_sleep(8);
        else if (idle_sweeps < 64) __builtin_amdgcn_s_sleep(64);
        else __builtin_amdgcn_s_sleep(127);
        if ((idle_sweeps & 255u) == 0) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t_idle == 0) t_idle = now;
            int bad = 0;
            if (lane == 0 && (now - t_idle > (unsigned long long)a.wait_bound_ticks ||
                              __hip_atomic_load(a.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))) {
                if (atomicCAS(a.abort, 0u, 1u) == 0u) a.abort[1] = (unsigned int)w;
                bad = 1;
            }
            give_up = __builtin_amdgcn_readfirstlane(bad) != 0;
        }
    }
    if (lane == 0 && loss != 0.0) atomicAdd(a.loss_acc, loss);
}

struct MfHogArgs {
    const int64_t *rid, *cid;  // COO order (no ownership)
    const float *val;
    // ownership tables: wave w processes own_*[wave_ptr[w] .. wave_ptr[w+1]) every epoch;
    // own_u < 0 encodes a shared (heavy) user as ~u
    const int32_t *own_u, *own_i;
    const float *own_r;
    const int64_t *wave_ptr;
    float *U, *V, *Bu, *Bi;
    // split items (mf_blocks.inc, "virtual rows"): an item id >= n_items names copy id - n_items of a hot row, which lives in
    // the side table; Vx_shifted = that table - n_items * k, so the row of id i is (i < n_items ? V : Vx_shifted) + i * k.
    // The padded bias table simply has the copies' lines after the items'.  No split: n_items = INT32_MAX.
    float *Vx_shifted;
    int32_t n_items;
    double *loss_acc;
    int64_t n;
    int k, use_bias, bstride;
    float lr, reg, mu;
};

// Row-wise layout as in bpr.hip: the G lanes of a group own consecutive floats of a row, so every
// gather / fp32-atomic scatter instruction covers whole 128-byte lines.  OWNED (G == 64): every
// wave owns a fixed set of users and all of their ratings, so U rows and user biases are updated
// with plain stores by exactly one wave (no atomics, nothing lost); item rows and item biases use
// device-scope atomics.  Ratings of a wave are interleaved round-robin over its users so that the
// UNR ratings in flight rarely share a user; when they do, their deltas are summed.
template <int G, int R, int UNR, bool OWNED>
__global__ __launch_bounds__(kBlock) void mf_hogwild_rowwise_kernel(const MfHogArgs a) {
    static_assert(!OWNED || G == kWave, "ownership needs one rating per wave step");
    constexpr int TPW = kWave / G;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int grp = lane / G, lg = lane & (G - 1);
    const int64_t total_waves = (int64_t)gridDim.x * kWavesPerBlock;
    const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    int64_t begin = 0, end = a.n, tile0 = wave_id, tile_step = total_waves;
    if (OWNED) {
        begin = a.wave_ptr[wave_id];
        end = a.wave_ptr[wave_id + 1];
        tile0 = 0;
        tile_step = 1;
    }
    const int64_t n_tiles = (end - begin + kWave - 1) / kWave;
    bool inb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) inb[r] = lg + G * r < a.k;
    double loss = 0.0;
    for (int64_t tile = tile0; tile < n_tiles; tile += tile_step) {
        const int64_t s = begin + tile * kWave + lane;
        int32_t mu_ = 0, mi_ = 0;
        float mr = 0.f;
        if (s < end) {
            if (OWNED) {
                mu_ = a.own_u[s];
                mi_ = a.own_i[s];
                mr = a.own_r[s];
            } else {
                mu_ = (int32_t)a.rid[s];
                mi_ = (int32_t)a.cid[s];
                mr = a.val[s];
            }
        }
        const int nvalid = (int)min((int64_t)kWave, end - (begin + tile * kWave));
        for (int b = 0; b < nvalid; b += TPW * UNR) {
            float u[UNR][R], v[UNR][R], bu[UNR], bi[UNR], rt[UNR], du_all[UNR][R], dbu_all[UNR];
            float *pu[UNR], *pi[UNR];
            int32_t tue[UNR], tu[UNR], ti[UNR];
            bool act[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                const int slot = b + q * TPW + grp;
                act[q] = slot < nvalid;
                const int sl = act[q] ? slot : b;
                int32_t x = __shfl(mu_, sl, kWave);
                if (OWNED) x = __builtin_amdgcn_readfirstlane(x);
                tue[q] = x;
                tu[q] = (OWNED && x < 0) ? ~x : x;
                ti[q] = __shfl(mi_, sl, kWave);
                rt[q] = __shfl(mr, sl, kWave);
                pu[q] = a.U + (size_t)tu[q] * a.k + lg;
                pi[q] = (ti[q] < a.n_items ? a.V : a.Vx_shifted) + (size_t)ti[q] * a.k + lg;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    u[q][r] = inb[r] ? __builtin_nontemporal_load(pu[q] + G * r) : 0.f;
                    v[q][r] = inb[r] ? __builtin_nontemporal_load(pi[q] + G * r) : 0.f;
                }
                bu[q] = __builtin_nontemporal_load(a.Bu + tu[q]);
                bi[q] = __builtin_nontemporal_load(a.Bi + (size_t)ti[q] * a.bstride);
            }
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                float part = 0.f;
#pragma unroll
                for (int r = 0; r < R; ++r) part += u[q][r] * v[q][r];
                const float err = rt[q] - ((a.mu + bu[q] + bi[q]) + group_sum<G>(part));
#pragma unroll
                for (int r = 0; r < R; ++r) du_all[q][r] = a.lr * (err * v[q][r] - a.reg * u[q][r]);
                dbu_all[q] = a.lr * (err - a.reg * bu[q]);
                if (act[q]) {
                    const bool excl = OWNED && tue[q] >= 0;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if (inb[r]) {
                            atomic_add_f32(pi[q] + G * r, a.lr * (err * u[q][r] - a.reg * v[q][r]));
                            if (!excl) atomic_add_f32(pu[q] + G * r, du_all[q][r]);
                        }
                    }
                    if (lg == 0) {
                        if (a.use_bias) {
                            atomic_add_f32(a.Bi + (size_t)ti[q] * a.bstride, a.lr * (err - a.reg * bi[q]));
                            if (!excl) atomic_add_f32(a.Bu + tu[q], dbu_all[q]);
                        }
                        loss += (double)err * (double)err;
                    }
                }
            }
            if (OWNED) {
                // exclusive users: u_old + the summed deltas of every rating of this batch with the same user
#pragma unroll
                for (int q = 0; q < UNR; ++q) {
                    if (act[q] && tue[q] >= 0) {
                        float tot[R], totb = dbu_all[q];
#pragma unroll
                        for (int r = 0; r < R; ++r) tot[r] = du_all[q][r];
#pragma unroll
                        for (int q2 = 0; q2 < UNR; ++q2) {
                            if (q2 != q && act[q2] && tue[q2] == tue[q]) {
#pragma unroll
                                for (int r = 0; r < R; ++r) tot[r] += du_all[q2][r];
                                totb += dbu_all[q2];
                            }
                        }
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (inb[r]) pu[q][G * r] = u[q][r] + tot[r];
                        if (lg == 0 && a.use_bias) a.Bu[tu[q]] = bu[q] + totb;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o, kWave);
    if (lane == 0 && loss != 0.0) atomicAdd(a.loss_acc, loss);
}

// any k (> 256): G lanes per rating stride over the factors, two passes
template <int G>
__global__ __launch_bounds__(kBlock) void mf_hogwild_generic_kernel(const MfHogArgs a) {
    constexpr int TPW = kWave / G;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int grp = lane / G, lg = lane & (G - 1);
    const int64_t total_waves = (int64_t)gridDim.x * kWavesPerBlock;
    const int64_t n_tiles = (a.n + kWave - 1) / kWave;
    double loss = 0.0;
    for (int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave; tile < n_tiles; tile += total_waves) {
        const int64_t s = tile * kWave + lane;
        int32_t mu_ = 0, mi_ = 0;
        float mr = 0.f;
        if (s < a.n) {
            mu_ = (int32_t)a.rid[s];
            mi_ = (int32_t)a.cid[s];
            mr = a.val[s];
        }
        const int nvalid = (int)min((int64_t)kWave, a.n - tile * kWave);
        for (int b = 0; b < nvalid; b += TPW) {
            const int slot = b + grp;
            const bool act = slot < nvalid;
            const int sl = act ? slot : b;
            const int32_t tu = __shfl(mu_, sl, kWave), ti = __shfl(mi_, sl, kWave);
            const float tr = __shfl(mr, sl, kWave);
            // (an id >= n_items names a copy of a split hot item's row: MfHogArgs::Vx_shifted)
            float *pu = a.U + (size_t)tu * a.k, *pi = (ti < a.n_items ? a.V : a.Vx_shifted) + (size_t)ti * a.k;
            const float bu = load_f32_fresh(a.Bu + tu), bi = load_f32_fresh(a.Bi + (size_t)ti * a.bstride);
            float part = 0.f;
            for (int f = lg; f < a.k; f += G) part += load_f32_fresh(pu + f) * load_f32_fresh(pi + f);
            const float err = tr - ((a.mu + bu + bi) + group_sum<G>(part));
            if (act) {
                for (int f = lg; f < a.k; f += G) {
                    const float uf = load_f32_fresh(pu + f), vf = load_f32_fresh(pi + f);
                    atomic_add_f32(pu + f, a.lr * (err * vf - a.reg * uf));
                    atomic_add_f32(pi + f, a.lr * (err * uf - a.reg * vf));
                }
                if (lg == 0) {
                    if (a.use_bias) {
                        atomic_add_f32(a.Bu + tu, a.lr * (err - a.reg * bu));
                        atomic_add_f32(a.Bi + (size_t)ti * a.bstride, a.lr * (err - a.reg * bi));
                    }
                    loss += (double)err * (double)err;
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o, kWave);
    if (lane == 0 && loss != 0.0) atomicAdd(a.loss_acc, loss);
}

static int mf_pow2_group(int k) {
    int g = 4;
    while (g < k && g < 64) g <<= 1;
    return g;
}

}  // namespace chip

using namespace chip;

struct cornac_hip_mf {
    int device = 0;
    int64_t n_users = 0, n_items = 0, nnz = 0;
    int k = 0;
    hipStream_t stream = nullptr, own_stream = nullptr;   // stream == own_stream unless cornac_hip_mf_set_stream
    DevBuf<int64_t> rid, cid;
    DevBuf<float> val;
    DevBuf<float> U, V, Bu, Bi;
    DevBuf<double> loss;  // one slot per epoch
    // deterministic schedule (built lazily, reused by every epoch)
    bool sched_built = false;
    DevBuf<int32_t> ou, oi;
    DevBuf<float> orat;
    LevelSchedule sched;
    std::vector<int64_t> host_rid, host_cid;
    std::vector<float> host_val;
    // deterministic dataflow ("chain") kernel: ratings grouped by the wave that owns their item, per-user sequence numbers
    bool chain_built = false, chain_own_user = false;
    int chain_grid = 0;
    bool chain_refused = false;  // a cooperative launch of the dataflow kernel was refused: level schedule from then on
    DevBuf<int32_t> c_row_id, c_sid, c_seq;
    DevBuf<float> c_r;
    DevBuf<int64_t> c_wrow_ptr, c_row_beg, c_row_end, c_row_cur;
    DevBuf<unsigned int> uver, chain_abort;
    // hogwild block rotation (mf_blocks.inc): ratings grouped by (phase, xcd, slot, sub-round), tiles of whole user runs
    bool blocks_built = false, blocks_failed = false, blocks_probed = false;
    int hog_form = 0, hog_form_used = 0;  // 0 automatic, 1 fused atomic kernel, 2 block rotation
    int mb_cap = 0;
    size_t mb_lds = 0;
    DevBuf<int64_t> mb_blk_tile_ptr, mb_tile_ptr;
    DevBuf<int32_t> mb_u, mb_slot, mb_bin_ptr, mb_bin_items;
    DevBuf<float> mb_r;
    DevBuf<unsigned int> mb_sync;  // [8 slot counters | 8 barrier counters | abort]
    // split items (mf_blocks.inc): copies of their rows, the items, the copies of each
    DevBuf<float> mb_vx, mb_bix;
    DevBuf<int32_t> mb_split_item, mb_split_ptr;
    int mb_n_split = 0, mb_n_virtual = 0;
    bool split_built = false;
    std::vector<int32_t> split_of, split_ptr_h;   // item -> split index (-1: not split); copies of split j: [ptr[j], ptr[j + 1])
    DevBuf<int64_t> cid_split;                    // the COO item ids with the split items' ratings renamed to their copies (fused kernel)
    double timing[4] = {0, 0, 0, 0};
    EventTimer ktimer;  // hogwild SGD kernel launches
    DevBuf<float> Bipad;  // hogwild-mode view of Bi, one bias per 128-byte line
    void (*hog_kernel)(const chip::MfHogArgs) = nullptr;
    int hog_blocks_per_cu = 8;
    int step_grid = 0;           // form 3: workgroups of the throttled launch (0: not chosen yet)
    int64_t step_inflight = 0;   // ... and the ratings it keeps in flight together
    int64_t own_waves = 0;
    DevBuf<int32_t> own_u, own_i;
    DevBuf<float> own_r;
    DevBuf<int64_t> wave_ptr;
    // minibatch path (mf_minibatch.inc): dense gradients and optimiser state of [U, V, Bu, Bi]
    DevBuf<float> opt_g[4], opt_s1[4], opt_s2[4];
    DevBuf<int64_t> opt_order;
    DevBuf<uint8_t> opt_keep;  // dropout keep masks of a fit_minibatch_dropout call: [2][n_total][k]
    int64_t opt_step = 0;
    int opt_kind = -1;
    bool enqueue_open = false;  // epoch_enqueue has accumulated into loss[0] since the last cornac_hip_mf_sync
};

#include "mf_blocks.inc"

static void mf_check(cornac_hip_mf_t h) {
    REQUIRE(h != nullptr, "MF handle is NULL");
    HIP_CHECK(hipSetDevice(h->device));
}

static void mf_build_schedule(cornac_hip_mf_t h) {
    if (h->sched_built) return;
    Timer t;
    const int64_t n = h->nnz;
    std::vector<int32_t> su((size_t)n), si((size_t)n), outu((size_t)n), outi((size_t)n);
    for (int64_t s = 0; s < n; ++s) {
        su[(size_t)s] = (int32_t)h->host_rid[(size_t)s];
        si[(size_t)s] = (int32_t)h->host_cid[(size_t)s];
    }
    // level[s] = 1 + max(level of the previous rating of the same user / same item); the rating
    // values ride along through the counting sort.
    std::vector<int32_t> lvl_u((size_t)h->n_users, 0), lvl_i((size_t)h->n_items, 0), level((size_t)n);
    int32_t max_level = 0;
    for (int64_t s = 0; s < n; ++s) {
        int32_t l = std::max(lvl_u[(size_t)su[(size_t)s]], lvl_i[(size_t)si[(size_t)s]]) + 1;
        lvl_u[(size_t)su[(size_t)s]] = l;
        lvl_i[(size_t)si[(size_t)s]] = l;
        level[(size_t)s] = l;
        max_level = std::max(max_level, l);
    }
    h->sched.level_ptr.assign((size_t)max_level + 2, 0);
    for (int64_t s = 0; s < n; ++s) ++h->sched.level_ptr[(size_t)level[(size_t)s] + 1];
    for (size_t l = 1; l < h->sched.level_ptr.size(); ++l) h->sched.level_ptr[l] += h->sched.level_ptr[l - 1];
    std::vector<int64_t> cursor(h->sched.level_ptr.begin(), h->sched.level_ptr.end());
    std::vector<float> outr((size_t)n);
    for (int64_t s = 0; s < n; ++s) {
        const int64_t pos = cursor[(size_t)level[(size_t)s]]++;
        outu[(size_t)pos] = su[(size_t)s];
        outi[(size_t)pos] = si[(size_t)s];
        outr[(size_t)pos] = h->host_val[(size_t)s];
    }
    h->sched.n_active = n;
    h->ou.alloc((size_t)n);
    h->oi.alloc((size_t)n);
    h->orat.alloc((size_t)n);
    h->ou.upload(outu.data(), (size_t)n, h->stream);
    h->oi.upload(outi.data(), (size_t)n, h->stream);
    h->orat.upload(outr.data(), (size_t)n, h->stream);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->sched_built = true;
    h->timing[1] += t.ms();
}

template <int G>
static void launch_mf_level(cornac_hip_mf_t h, int64_t off, int cnt, float lr, float reg, float mu, int use_bias,
                            double *loss_slot) {
    const int groups_per_block = kBlock / G;
    const int grid = (cnt + groups_per_block - 1) / groups_per_block;
    hipLaunchKernelGGL(mf_det_level_kernel<G>, dim3(grid), dim3(kBlock), 0, h->stream, h->ou.p, h->oi.p, h->orat.p,
                       off, cnt, h->U.p, h->V.p, h->Bu.p, h->Bi.p, h->k, lr, reg, mu, use_bias, loss_slot);
}

static void mf_epoch_deterministic(cornac_hip_mf_t h, float lr, float reg, float mu, int use_bias, double *loss_slot) {
    mf_build_schedule(h);
    const int G = mf_pow2_group(h->k);
    const std::vector<int64_t> &lp = h->sched.level_ptr;
    for (size_t l = 1; l + 1 < lp.size(); ++l) {
        const int64_t off = lp[l];
        const int cnt = (int)(lp[l + 1] - lp[l]);
        if (cnt <= 0) continue;
        switch (G) {
            case 4: launch_mf_level<4>(h, off, cnt, lr, reg, mu, use_bias, loss_slot); break;
            case 8: launch_mf_level<8>(h, off, cnt, lr, reg, mu, use_bias, loss_slot); break;
            case 16: launch_mf_level<16>(h, off, cnt, lr, reg, mu, use_bias, loss_slot); break;
            case 32: launch_mf_level<32>(h, off, cnt, lr, reg, mu, use_bias, loss_slot); break;
            default: launch_mf_level<64>(h, off, cnt, lr, reg, mu, use_bias, loss_slot); break;
        }
    }
    HIP_CHECK(hipGetLastError());
}

// ---- deterministic dataflow kernel: schedule and launch ---------------------------------------------------------------
typedef void (*MfChainKernel)(const MfChainArgs);
static MfChainKernel pick_chain_kernel(int k, bool own_user) {
    if (own_user) {
        if (k <= 64) return mf_det_chain_kernel<1, true>;
        if (k <= 128) return mf_det_chain_kernel<2, true>;
        if (k <= 192) return mf_det_chain_kernel<3, true>;
        return mf_det_chain_kernel<4, true>;
    }
    if (k <= 64) return mf_det_chain_kernel<1, false>;
    if (k <= 128) return mf_det_chain_kernel<2, false>;
    if (k <= 192) return mf_det_chain_kernel<3, false>;
    return mf_det_chain_kernel<4, false>;
}

static bool mf_uses_chain(cornac_hip_mf_t h) { return h->k <= 256 && h->nnz >= 4096; }

static void mf_build_chain(cornac_hip_mf_t h) {
    if (h->chain_built) return;
    Timer t;
    const int64_t n = h->nnz, ni = h->n_items, nu = h->n_users;
    // Which side do the waves own?  The side whose ratings are ADJACENT in the stored order (a uir_tuple sorted by user
    // or by item): a run then stays inside one wave.  Without adjacency: the side with the longer hottest chain (its
    // links are then wave-local instead of cross-wave hand-offs).
    int64_t adj_u = 0, adj_i = 0;
    std::vector<int64_t> cnt_u((size_t)nu, 0), cnt_i((size_t)ni, 0);
    for (int64_t s = 0; s < n; ++s) {
        ++cnt_u[(size_t)h->host_rid[(size_t)s]];
        ++cnt_i[(size_t)h->host_cid[(size_t)s]];
        if (s) {
            adj_u += h->host_rid[(size_t)s] == h->host_rid[(size_t)s - 1];
            adj_i += h->host_cid[(size_t)s] == h->host_cid[(size_t)s - 1];
        }
    }
    const int64_t max_u = *std::max_element(cnt_u.begin(), cnt_u.end()), max_i = *std::max_element(cnt_i.begin(), cnt_i.end());
    h->chain_own_user = std::max(adj_u, adj_i) * 10 > n * 3 ? adj_u > adj_i : max_u > max_i;
    const bool own_user = h->chain_own_user;
    int per_cu = 0;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pick_chain_kernel(h->k, own_user), kBlock, 0));
    // progress needs the wave that owns the globally first unfinished rating to be running: HALF of what the occupancy
    // query admits keeps every block of the grid resident even where the hardware admits one block per CU fewer than
    // the API says (kernels with more than 80 SGPRs, MI355X_MICROARCH.md, residency)
    h->chain_grid = device_info(h->device).cus * std::max(1, std::min(per_cu, 8) / 2);
    const int64_t W = (int64_t)h->chain_grid * kWavesPerBlock;
    // owned rows -> waves, heaviest first onto the least loaded wave
    const std::vector<int64_t> &cnt = own_user ? cnt_u : cnt_i;
    const std::vector<int64_t> &own_id = own_user ? h->host_rid : h->host_cid, &sh_id = own_user ? h->host_cid : h->host_rid;
    const int64_t n_own = own_user ? nu : ni, n_sh = own_user ? ni : nu;
    std::vector<int32_t> order((size_t)n_own);
    for (int64_t i = 0; i < n_own; ++i) order[(size_t)i] = (int32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return cnt[(size_t)x] > cnt[(size_t)y]; });
    typedef std::pair<int64_t, int64_t> LW;
    std::priority_queue<LW, std::vector<LW>, std::greater<LW>> heap;
    for (int64_t w = 0; w < W; ++w) heap.push(LW(0, w));
    std::vector<int32_t> owner((size_t)n_own, 0);
    std::vector<int64_t> load((size_t)W, 0);
    for (int32_t i : order) {
        if (cnt[(size_t)i] == 0) continue;
        LW top = heap.top();
        heap.pop();
        owner[(size_t)i] = (int32_t)top.second;
        top.first += cnt[(size_t)i];
        load[(size_t)top.second] = top.first;
        heap.push(top);
    }
    // rows of a wave in the order of their first rating; a row's ratings in stored order
    std::vector<int64_t> first((size_t)n_own, -1);
    for (int64_t s = n - 1; s >= 0; --s) first[(size_t)own_id[(size_t)s]] = s;
    std::vector<std::vector<int32_t>> wrows((size_t)W);
    for (int64_t i = 0; i < n_own; ++i)
        if (cnt[(size_t)i] > 0) wrows[(size_t)owner[(size_t)i]].push_back((int32_t)i);
    std::vector<int64_t> wrow_ptr((size_t)W + 1, 0), row_beg, row_end, row_of((size_t)n_own, -1);
    std::vector<int32_t> row_id;
    int64_t pos = 0;
    for (int64_t w = 0; w < W; ++w) {
        std::sort(wrows[(size_t)w].begin(), wrows[(size_t)w].end(), [&](int32_t x, int32_t y) { return first[(size_t)x] < first[(size_t)y]; });
        for (int32_t i : wrows[(size_t)w]) {
            row_of[(size_t)i] = (int64_t)row_id.size();
            row_id.push_back(i);
            row_beg.push_back(pos);
            pos += cnt[(size_t)i];
            row_end.push_back(pos);
        }
        wrow_ptr[(size_t)w + 1] = (int64_t)row_id.size();
    }
    std::vector<int64_t> cur(row_beg);
    std::vector<int32_t> csid((size_t)n), cseq((size_t)n), seen((size_t)n_sh, 0);
    std::vector<float> cr((size_t)n);
    for (int64_t s = 0; s < n; ++s) {
        const int64_t p = cur[(size_t)row_of[(size_t)own_id[(size_t)s]]]++;
        csid[(size_t)p] = (int32_t)sh_id[(size_t)s];
        cr[(size_t)p] = h->host_val[(size_t)s];
        cseq[(size_t)p] = seen[(size_t)sh_id[(size_t)s]]++;
    }
    const size_t n_rows = row_id.size();
    h->c_sid.alloc((size_t)n); h->c_seq.alloc((size_t)n); h->c_r.alloc((size_t)n);
    h->c_row_id.alloc(n_rows); h->c_row_beg.alloc(n_rows); h->c_row_end.alloc(n_rows); h->c_row_cur.alloc(n_rows);
    h->c_wrow_ptr.alloc((size_t)W + 1);
    h->uver.alloc((size_t)n_sh);
    h->chain_abort.alloc(8);
    h->c_sid.upload(csid.data(), (size_t)n, h->stream);
    h->c_seq.upload(cseq.data(), (size_t)n, h->stream);
    h->c_r.upload(cr.data(), (size_t)n, h->stream);
    h->c_row_id.upload(row_id.data(), n_rows, h->stream);
    h->c_row_beg.upload(row_beg.data(), n_rows, h->stream);
    h->c_row_end.upload(row_end.data(), n_rows, h->stream);
    h->c_wrow_ptr.upload(wrow_ptr.data(), (size_t)W + 1, h->stream);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->chain_built = true;
    h->timing[1] += t.ms();
}

// returns false when the runtime refuses the launch (the whole grid cannot be co-resident): nothing has run, the caller
// takes the level schedule
static bool mf_epoch_chain(cornac_hip_mf_t h, float lr, float reg, float mu, int use_bias, double *loss_slot) {
    mf_build_chain(h);
    HIP_CHECK(hipMemsetAsync(h->uver.p, 0, h->uver.n * sizeof(unsigned int), h->stream));
    HIP_CHECK(hipMemsetAsync(h->chain_abort.p, 0, 8 * sizeof(unsigned int), h->stream));
    HIP_CHECK(hipMemcpyAsync(h->c_row_cur.p, h->c_row_beg.p, h->c_row_beg.n * sizeof(int64_t), hipMemcpyDeviceToDevice, h->stream));
    MfChainArgs a;
    a.wrow_ptr = h->c_wrow_ptr.p; a.row_id = h->c_row_id.p; a.row_end = h->c_row_end.p; a.row_cur = h->c_row_cur.p;
    a.csid = h->c_sid.p; a.cseq = h->c_seq.p; a.cr = h->c_r.p;
    a.ver = h->uver.p; a.abort = h->chain_abort.p;
    a.U = h->U.p; a.V = h->V.p; a.Bu = h->Bu.p; a.Bi = h->Bi.p; a.loss_acc = loss_slot;
    a.k = h->k; a.use_bias = use_bias; a.lr = lr; a.reg = reg; a.mu = mu;
    a.wait_bound_ticks = (long long)prof_env_int("CORNAC_HIP_MF_CHAIN_WAIT_S", 120) * 100000000ll;
    // The waves wait for each other (per-row hand-over counters): every workgroup of the grid has to be resident.  A
    // cooperative launch makes that the runtime's business — it refuses a grid that cannot be co-resident instead of
    // letting the resident half spin on the other half until the time bound (advisor r3).
    {
        void *kargs[] = {(void *)&a};
        const hipError_t st = hipLaunchCooperativeKernel((const void *)pick_chain_kernel(h->k, h->chain_own_user),
                                                         dim3(h->chain_grid), dim3(kBlock), kargs, 0, h->stream);
        if (st == hipErrorCooperativeLaunchTooLarge) {  // the grid cannot be co-resident here: the level schedule takes over
            (void)hipGetLastError();
            return false;
        }
        HIP_CHECK(st);  // any other status is a real launch error, not a refusal (advisor r4)
    }
    unsigned int ab[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_CHECK(hipMemcpyAsync(ab, h->chain_abort.p, sizeof ab, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    if (ab[0])
        fail(CORNAC_HIP_ERR_HIP, "deterministic MF dataflow kernel: wave %u made no progress for its time bound (internal error)", ab[1]);
    return true;
}

typedef void (*MfHogKernel)(const MfHogArgs);

static float hash_key(uint32_t x) {  // position -> pseudo-random key in [0, 1)
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

static MfHogKernel pick_mf_kernel(int k, bool owned) {
    if (k <= 4) return mf_hogwild_rowwise_kernel<4, 1, 2, false>;
    if (k <= 8) return mf_hogwild_rowwise_kernel<8, 1, 2, false>;
    if (k <= 16) return mf_hogwild_rowwise_kernel<16, 1, 2, false>;
    if (k <= 32) return mf_hogwild_rowwise_kernel<32, 1, 4, false>;
    if (owned) {
        if (k <= 64) return mf_hogwild_rowwise_kernel<64, 1, 4, true>;
        if (k <= 128) return mf_hogwild_rowwise_kernel<64, 2, 2, true>;
        if (k <= 192) return mf_hogwild_rowwise_kernel<64, 3, 2, true>;
        return mf_hogwild_rowwise_kernel<64, 4, 1, true>;
    }
    if (k <= 64) return mf_hogwild_rowwise_kernel<64, 1, 4, false>;
    if (k <= 128) return mf_hogwild_rowwise_kernel<64, 2, 2, false>;
    if (k <= 192) return mf_hogwild_rowwise_kernel<64, 3, 2, false>;
    if (k <= 256) return mf_hogwild_rowwise_kernel<64, 4, 1, false>;
    return mf_hogwild_generic_kernel<64>;
}

// ---- split items ("virtual rows", mf_blocks.inc header): which items, how many copies each, the side tables ----
// An item holding more than 0.1 % of the ratings of a LARGE problem (>= 2^20 ratings: below that the launch does not hold
// enough ratings in flight for the staleness to matter) is split into W = ceil(share / 0.1 %) copies.
static void mf_build_split(cornac_hip_mf_t h) {
    if (h->split_built) return;
    const int64_t n = h->nnz, ni = h->n_items;
    h->split_of.assign((size_t)ni, -1);
    h->split_ptr_h.assign(1, 0);
    std::vector<int32_t> items;
    // form 3 (the step handles of the multi-GPU block rotation: few item rows, 10^4..10^7 ratings per launch): the same
    // kind of bound — a few dozen concurrent stale updates per copy — at any size.  Its launch keeps step_inflight ratings in flight
    // together (throttled to ~4 per item row, mf_launch_fused), so item i sees cnt_i x min(n, step_inflight) / n of them:
    // a copy per 32 (twice the big handles' bound: fewer copies, see there).  With even shares no row is split; a row with
    // > 8x the average share is.
    const bool any_size = h->hog_form == 3;
    const int64_t flight = any_size ? std::max<int64_t>(1, std::min<int64_t>(n, h->step_inflight)) : 16000;
    const int64_t per_copy = any_size ? (flight + 31) / 32 : 1000;   // split when cnt_i x per_copy > n
    if (n >= (int64_t(1) << 20) || any_size) {
        std::vector<int64_t> cnt((size_t)ni, 0);
        for (int64_t s = 0; s < n; ++s) ++cnt[(size_t)h->host_cid[(size_t)s]];
        for (int64_t i = 0; i < ni; ++i)
            if (cnt[(size_t)i] * per_copy > n) {
                const int64_t W = std::min<int64_t>(256, (cnt[(size_t)i] * per_copy + n - 1) / n);
                h->split_of[(size_t)i] = (int32_t)items.size();
                items.push_back((int32_t)i);
                h->split_ptr_h.push_back(h->split_ptr_h.back() + (int32_t)W);
            }
    }
    h->mb_n_split = (int)items.size();
    h->mb_n_virtual = h->split_ptr_h.back();
    if (h->mb_n_split) {
        h->mb_split_item.alloc(items.size());
        h->mb_split_ptr.alloc(h->split_ptr_h.size());
        h->mb_split_item.upload(items.data(), items.size(), h->stream);
        h->mb_split_ptr.upload(h->split_ptr_h.data(), h->split_ptr_h.size(), h->stream);
        h->mb_vx.alloc((size_t)h->mb_n_virtual * (size_t)h->k);
        h->mb_bix.alloc((size_t)h->mb_n_virtual * kBiasStride);
        HIP_CHECK(hipStreamSynchronize(h->stream));
    }
    h->split_built = true;
}
// the row a rating of item `it` at COO position s updates: the item, or the copy a hash of the position names
static inline int32_t mf_split_id(cornac_hip_mf_t h, int64_t s, int64_t it) {
    const int32_t j = h->split_of[(size_t)it];
    if (j < 0) return (int32_t)it;
    uint32_t x = (uint32_t)s * 0x85EBCA6Bu + 0x165667B1u;
    x ^= x >> 16; x *= 0x9E3779B1u; x ^= x >> 13;
    const uint32_t W = (uint32_t)(h->split_ptr_h[(size_t)j + 1] - h->split_ptr_h[(size_t)j]);
    return (int32_t)(h->n_items + h->split_ptr_h[(size_t)j] + (int32_t)(x % W));
}

// Users -> waves of the persistent grid, balanced by rating count (LPT); users heavier than half a
// wave's share are shared (their ratings are dealt evenly to all waves, atomics).  Inside a wave the
// ratings of its users are interleaved round-robin.
static void mf_build_ownership(cornac_hip_mf_t h, int64_t W) {
    mf_build_split(h);
    if (h->own_waves == W) return;
    const int64_t nnz = h->nnz, nu = h->n_users;
    std::vector<int64_t> uptr((size_t)nu + 1, 0);
    for (int64_t s = 0; s < nnz; ++s) ++uptr[(size_t)h->host_rid[(size_t)s] + 1];
    for (int64_t u = 0; u < nu; ++u) uptr[(size_t)u + 1] += uptr[(size_t)u];
    std::vector<int32_t> pos((size_t)nnz);  // COO positions grouped by user, stable
    {
        std::vector<int64_t> cur(uptr.begin(), uptr.end() - 1);
        for (int64_t s = 0; s < nnz; ++s) pos[(size_t)cur[(size_t)h->host_rid[(size_t)s]]++] = (int32_t)s;
    }
    const int64_t cap = std::max<int64_t>(1, nnz / W / 2);
    std::vector<int64_t> load((size_t)W, 0);
    int64_t n_shared = 0;
    for (int64_t u = 0; u < nu; ++u)
        if (uptr[(size_t)u + 1] - uptr[(size_t)u] > cap) n_shared += uptr[(size_t)u + 1] - uptr[(size_t)u];
    const int64_t blk = (n_shared + W - 1) / W;
    for (int64_t w = 0; w < W && blk > 0; ++w) load[(size_t)w] = std::max<int64_t>(0, std::min(blk, n_shared - w * blk));
    std::vector<int32_t> order;
    for (int64_t u = 0; u < nu; ++u) {
        const int64_t d = uptr[(size_t)u + 1] - uptr[(size_t)u];
        if (d > 0 && d <= cap) order.push_back((int32_t)u);
    }
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
        return uptr[(size_t)x + 1] - uptr[(size_t)x] > uptr[(size_t)y + 1] - uptr[(size_t)y];
    });
    typedef std::pair<int64_t, int64_t> LW;
    std::priority_queue<LW, std::vector<LW>, std::greater<LW>> heap;
    for (int64_t w = 0; w < W; ++w) heap.push(LW(load[(size_t)w], w));
    std::vector<std::vector<int32_t>> wave_users((size_t)W);
    for (int32_t u : order) {
        LW t = heap.top();
        heap.pop();
        wave_users[(size_t)t.second].push_back(u);
        t.first += uptr[(size_t)u + 1] - uptr[(size_t)u];
        load[(size_t)t.second] = t.first;
        heap.push(t);
    }
    std::vector<int64_t> wptr((size_t)W + 1, 0);
    for (int64_t w = 0; w < W; ++w) wptr[(size_t)w + 1] = wptr[(size_t)w] + load[(size_t)w];
    REQUIRE(wptr[(size_t)W] == nnz, "MF ownership tables do not cover the ratings");
    std::vector<int32_t> ou((size_t)nnz), oi((size_t)nnz);
    std::vector<float> orr((size_t)nnz);
    auto emit = [&](int64_t dst, int32_t s, bool shared) {
        const int32_t u = (int32_t)h->host_rid[(size_t)s];
        ou[(size_t)dst] = shared ? ~u : u;
        oi[(size_t)dst] = mf_split_id(h, s, h->host_cid[(size_t)s]);
        orr[(size_t)dst] = h->host_val[(size_t)s];
    };
    // Shared ratings are dealt round-robin (rating p of the shared sequence -> wave p % W) so that a
    // heavy user's ratings are spread over all waves; inside a wave the ratings are put in a
    // (deterministic, hash-keyed) random order.  Any order that is correlated across waves is
    // poison: with COO input sorted by (user, item), "r-th rating of every user" would make all
    // ~80 k in-flight ratings hit the same narrow band of item rows — hundreds of stale concurrent
    // updates per row, which both serialises the atomics and makes SGD diverge.
    std::vector<std::vector<int32_t>> wave_shared((size_t)W);
    {
        int64_t sp = 0;
        for (int64_t u = 0; u < nu; ++u) {
            if (uptr[(size_t)u + 1] - uptr[(size_t)u] <= cap) continue;
            for (int64_t p = uptr[(size_t)u]; p < uptr[(size_t)u + 1]; ++p, ++sp)
                wave_shared[(size_t)(sp % W)].push_back(pos[(size_t)p]);
        }
    }
    // loads were computed with contiguous blocks of `blk`; recompute them for the round-robin deal
    for (int64_t w = 0; w < W; ++w) {
        int64_t l = (int64_t)wave_shared[(size_t)w].size();
        for (int32_t u : wave_users[(size_t)w]) l += uptr[(size_t)u + 1] - uptr[(size_t)u];
        wptr[(size_t)w + 1] = wptr[(size_t)w] + l;
    }
    REQUIRE(wptr[(size_t)W] == nnz, "MF ownership tables do not cover the ratings");
    std::vector<std::pair<float, int32_t>> keyed;  // (key, COO position | shared flag in the sign of a side array)
    std::vector<char> is_shared;
    for (int64_t w = 0; w < W; ++w) {
        keyed.clear();
        is_shared.clear();
        const std::vector<int32_t> &sh = wave_shared[(size_t)w];
        for (size_t r = 0; r < sh.size(); ++r) {
            keyed.emplace_back(hash_key((uint32_t)sh[r]), (int32_t)is_shared.size());
            is_shared.push_back(1);
        }
        std::vector<int32_t> src;
        src.reserve((size_t)(wptr[(size_t)w + 1] - wptr[(size_t)w]));
        for (int32_t s0 : sh) src.push_back(s0);
        for (int32_t u : wave_users[(size_t)w]) {
            const int64_t d = uptr[(size_t)u + 1] - uptr[(size_t)u];
            for (int64_t r = 0; r < d; ++r) {
                keyed.emplace_back(hash_key((uint32_t)pos[(size_t)(uptr[(size_t)u] + r)]), (int32_t)src.size());
                src.push_back(pos[(size_t)(uptr[(size_t)u] + r)]);
                is_shared.push_back(0);
            }
        }
        std::stable_sort(keyed.begin(), keyed.end(),
                         [](const std::pair<float, int32_t> &x, const std::pair<float, int32_t> &y) { return x.first < y.first; });
        int64_t dst = wptr[(size_t)w];
        for (const auto &kv : keyed) emit(dst++, src[(size_t)kv.second], is_shared[(size_t)kv.second] != 0);
    }
    h->own_u.ensure((size_t)nnz);
    h->own_i.ensure((size_t)nnz);
    h->own_r.ensure((size_t)nnz);
    h->wave_ptr.ensure((size_t)W + 1);
    h->own_u.upload(ou.data(), (size_t)nnz, h->stream);
    h->own_i.upload(oi.data(), (size_t)nnz, h->stream);
    h->own_r.upload(orr.data(), (size_t)nnz, h->stream);
    h->wave_ptr.upload(wptr.data(), (size_t)W + 1, h->stream);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->own_waves = W;
}

// ---- hogwild block rotation (mf_blocks.inc): schedule and launch ----------------------------------------------------
typedef void (*MfBlocksKernel)(const MfBlockArgs);
static MfBlocksKernel pick_blocks_kernel(int k) {
    if (k <= 64) return mf_blocks_kernel<1, 4>;
    if (k <= 128) return mf_blocks_kernel<2, 4>;
    if (k <= 192) return mf_blocks_kernel<3, 2>;
    return mf_blocks_kernel<4, 2>;
}

static size_t mf_blocks_lds(int cap, int k) {
    const int kp = ((k + kWave - 1) / kWave) * kWave;
    return (size_t)cap * (kp + 2) * sizeof(float);
}

// 8 partitions x 32 parts: heaviest first onto the least loaded of the 256 parts
static std::vector<int32_t> mf_lpt_256(const std::vector<int64_t> &cnt, int *max_rows) {
    const int64_t n = (int64_t)cnt.size();
    std::vector<int32_t> order((size_t)n), part((size_t)n, -1);
    for (int64_t i = 0; i < n; ++i) order[(size_t)i] = (int32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return cnt[(size_t)x] > cnt[(size_t)y]; });
    typedef std::pair<int64_t, int32_t> LW;
    std::priority_queue<LW, std::vector<LW>, std::greater<LW>> heap;
    for (int32_t b = 0; b < 256; ++b) heap.push(LW(0, b));
    std::vector<int> rows(256, 0);
    for (int32_t i : order) {
        if (cnt[(size_t)i] == 0) continue;
        LW top = heap.top();
        heap.pop();
        part[(size_t)i] = top.second;
        ++rows[(size_t)top.second];
        top.first += cnt[(size_t)i];
        heap.push(top);
    }
    if (max_rows) *max_rows = *std::max_element(rows.begin(), rows.end());
    return part;
}

static bool mf_uses_blocks(cornac_hip_mf_t h) {
    const DeviceInfo &di = device_info(h->device);
    if (h->blocks_failed || h->hog_form == 1 || h->hog_form == 3 || di.cus != 256 || di.xcds != 8 || h->k <= 32 || h->k > 256 || h->n_items < 256)
        return false;
    return h->hog_form == 2 || h->nnz >= (int64_t(1) << 22);
}

static bool mf_build_blocks(cornac_hip_mf_t h) {
    if (h->blocks_built) return true;
    Timer t;
    const int64_t n = h->nnz, ni = h->n_items, nu = h->n_users;
    std::vector<int64_t> cnt_u((size_t)nu, 0), cnt_i((size_t)ni, 0);
    for (int64_t s = 0; s < n; ++s) {
        ++cnt_u[(size_t)h->host_rid[(size_t)s]];
        ++cnt_i[(size_t)h->host_cid[(size_t)s]];
    }
    // hot items: more than a tenth of a bin's share of the ratings — their LDS row lock would serialise the bin's 16 waves.
    // They stay in global memory under atomics and take no bin.
    std::vector<int64_t> cnt_cold(cnt_i);
    std::vector<char> hot((size_t)ni, 0);
    // ... and a row so popular (> 0.1 % of ALL ratings) that more than ~16 of the ~16 000 ratings in flight at any time name
    // it is SPLIT into W = ceil(share / 0.1 %) virtual rows: summed, hundreds of atomic updates computed from one stale copy
    // overshoot and the factorisation diverges (measured at SURVEY 8d's Zipf(0.8): one row with 3.2 % of the ratings, loss
    // = nan in this form AND in the fused kernel).  The copies are hot rows of their own (ids n_items + v, a side table),
    // merged after every phase (mf_virtual_merge_kernel); a rating takes the copy a hash of its position names.
    mf_build_split(h);
    for (int64_t i = 0; i < ni; ++i)
        if (cnt_i[(size_t)i] * 10 * 256 > n) {
            hot[(size_t)i] = 1;
            cnt_cold[(size_t)i] = 0;
        }
    auto hot_id = [&](int64_t s) -> int32_t { return mf_split_id(h, s, h->host_cid[(size_t)s]); };
    int cap = 0;
    const std::vector<int32_t> ibin = mf_lpt_256(cnt_cold, &cap), ublk = mf_lpt_256(cnt_u, nullptr);
    cap = std::max(cap, 1);
    if (mf_blocks_lds(cap, h->k) > 140 * 1024) {
        h->blocks_failed = true;
        return false;
    }
    // bins: items of bin g in id order; an item's slot = its position there
    std::vector<int32_t> bin_ptr(257, 0), slot((size_t)ni, 0);
    for (int64_t i = 0; i < ni; ++i)
        if (ibin[(size_t)i] >= 0) ++bin_ptr[(size_t)ibin[(size_t)i] + 1];
    for (int g = 0; g < 256; ++g) bin_ptr[(size_t)g + 1] += bin_ptr[(size_t)g];
    std::vector<int32_t> bin_items((size_t)std::max(1, bin_ptr[256])), bcur(bin_ptr.begin(), bin_ptr.end() - 1);
    for (int64_t i = 0; i < ni; ++i) {
        const int g = ibin[(size_t)i];
        if (g < 0) continue;
        slot[(size_t)i] = bcur[(size_t)g] - bin_ptr[(size_t)g];
        bin_items[(size_t)bcur[(size_t)g]++] = (int32_t)i;
    }
    // rating -> block key (phase, xcd, slot, sub-round): item bin (ip, c), user block (up, q):
    //   phase = (ip - up) mod 8, xcd = up, slot = c, sub-round = (q - c) mod 32
    // (a hot item has no bin: its rating takes one of the 256 (ip, c) by a hash of its position)
    const int64_t n_keys = 8 * 8 * 32 * 32;
    auto key_of = [&](int64_t s) -> int64_t {
        const int64_t it = h->host_cid[(size_t)s];
        int g = ibin[(size_t)it];
        if (hot[(size_t)it]) {
            uint32_t x = (uint32_t)s * 0x9E3779B1u;
            x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13;
            g = (int)(x & 255u);
        }
        const int b = ublk[(size_t)h->host_rid[(size_t)s]];
        const int ip = g >> 5, c = g & 31, up = b >> 5, q = b & 31;
        return ((((int64_t)((ip - up) & 7) * 8 + up) * 32 + c) * 32) + ((q - c) & 31);
    };
    std::vector<int64_t> kptr((size_t)n_keys + 1, 0);
    for (int64_t s = 0; s < n; ++s) ++kptr[(size_t)key_of(s) + 1];
    for (int64_t q = 0; q < n_keys; ++q) kptr[(size_t)q + 1] += kptr[(size_t)q];
    std::vector<int64_t> kcur(kptr.begin(), kptr.end() - 1), perm((size_t)n);
    for (int64_t s = 0; s < n; ++s) perm[(size_t)kcur[(size_t)key_of(s)]++] = s;
    // inside a block: the ratings of a user form a run; the runs are dealt to the workgroup's 16 waves (longest first
    // onto the shortest list), so every wave has its own list of ~100 ratings per sub-round, level with the others at the
    // barrier, and a user row belongs to one wave.  Inside a list the k-th ratings of all its users come before the
    // (k+1)-th ones: the ratings a wave has in flight together name distinct users.  Tile (block q, wave w) = q * 16 + w.
    std::vector<int32_t> b_u((size_t)n), b_slot((size_t)n);
    std::vector<float> b_r((size_t)n);
    std::vector<int64_t> tile_ptr((size_t)n_keys * kMbWaves + 1, 0), blk_tile_ptr(1, 0);
    {
        std::vector<std::pair<int64_t, int64_t>> runs;                    // (length, start in perm)
        std::vector<std::vector<std::pair<int64_t, int64_t>>> lists(kMbWaves);  // (occurrence, rating)
        std::vector<int64_t> fill(kMbWaves);
        for (int64_t q = 0; q < n_keys; ++q) {
            const int64_t lo = kptr[(size_t)q], hi = kptr[(size_t)q + 1];
            std::sort(perm.begin() + lo, perm.begin() + hi, [&](int64_t x, int64_t y) {
                const int64_t ux = h->host_rid[(size_t)x], uy = h->host_rid[(size_t)y];
                return ux != uy ? ux < uy : x < y;
            });
            runs.clear();
            for (int64_t p = lo; p < hi;) {
                int64_t e = p + 1;
                while (e < hi && h->host_rid[(size_t)perm[(size_t)e]] == h->host_rid[(size_t)perm[(size_t)p]]) ++e;
                runs.push_back(std::make_pair(e - p, p));
                p = e;
            }
            std::stable_sort(runs.begin(), runs.end(), [](const std::pair<int64_t, int64_t> &x, const std::pair<int64_t, int64_t> &y) { return x.first > y.first; });
            for (int w = 0; w < kMbWaves; ++w) {
                lists[(size_t)w].clear();
                fill[(size_t)w] = 0;
            }
            for (const auto &run : runs) {
                const int w = (int)(std::min_element(fill.begin(), fill.end()) - fill.begin());
                for (int64_t t = 0; t < run.first; ++t) lists[(size_t)w].push_back(std::make_pair(t, perm[(size_t)(run.second + t)]));
                fill[(size_t)w] += run.first;
            }
            int64_t pos = lo;
            for (int w = 0; w < kMbWaves; ++w) {
                tile_ptr[(size_t)(q * kMbWaves + w)] = pos;
                auto &L = lists[(size_t)w];
                std::stable_sort(L.begin(), L.end(), [](const std::pair<int64_t, int64_t> &x, const std::pair<int64_t, int64_t> &y) { return x.first < y.first; });
                for (const auto &e : L) {
                    const int64_t sidx = e.second;
                    b_u[(size_t)pos] = (int32_t)h->host_rid[(size_t)sidx];
                    b_slot[(size_t)pos] = hot[(size_t)h->host_cid[(size_t)sidx]] ? (int32_t)((int64_t)hot_id(sidx) | 0x80000000ll)
                                                                                 : slot[(size_t)h->host_cid[(size_t)sidx]];
                    b_r[(size_t)pos] = h->host_val[(size_t)sidx];
                    ++pos;
                }
            }
        }
        tile_ptr[(size_t)n_keys * kMbWaves] = n;
    }

    // a tile's end is the next tile's start (tiles are contiguous over the whole array)
    h->mb_blk_tile_ptr.alloc(blk_tile_ptr.size());
    h->mb_tile_ptr.alloc(tile_ptr.size());
    h->mb_u.alloc((size_t)n); h->mb_slot.alloc((size_t)n); h->mb_r.alloc((size_t)n);
    h->mb_bin_ptr.alloc(257);
    h->mb_bin_items.alloc(bin_items.size());
    h->mb_sync.alloc(24);
    h->mb_blk_tile_ptr.upload(blk_tile_ptr.data(), blk_tile_ptr.size(), h->stream);
    h->mb_tile_ptr.upload(tile_ptr.data(), tile_ptr.size(), h->stream);
    h->mb_u.upload(b_u.data(), (size_t)n, h->stream);
    h->mb_slot.upload(b_slot.data(), (size_t)n, h->stream);
    h->mb_r.upload(b_r.data(), (size_t)n, h->stream);
    h->mb_bin_ptr.upload(bin_ptr.data(), 257, h->stream);
    h->mb_bin_items.upload(bin_items.data(), bin_items.size(), h->stream);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->mb_cap = cap;
    h->mb_lds = std::max(mf_blocks_lds(cap, h->k), (size_t)82 * 1024);  // >= 82 KB: one workgroup per CU
    h->blocks_built = true;
    h->timing[1] += t.ms();
    return true;
}

// returns false when the form cannot be used (its tables do not fit, or the placement probe did not see 32 workgroups on
// every XCD): the caller runs the epoch with the fused kernel.  A launch that gives up in the middle of an epoch (a
// barrier bound: another kernel held CUs) is reported as an error — part of the epoch has been applied — and the handle
// then stays on the fused kernel.
static bool mf_epoch_blocks(cornac_hip_mf_t h, float lr, float reg, float mu, int use_bias, double *loss_slot) {
    if (!mf_build_blocks(h)) return false;
    MfBlocksKernel kern = pick_blocks_kernel(h->k);
    HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->mb_lds));
    if (!h->blocks_probed) {  // once per handle: does a launch of this shape put 32 workgroups on every XCD?
        HIP_CHECK(hipFuncSetAttribute((const void *)mf_blocks_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)h->mb_lds));
        HIP_CHECK(hipMemsetAsync(h->mb_sync.p, 0, 24 * sizeof(unsigned int), h->stream));
        hipLaunchKernelGGL(mf_blocks_probe_kernel, dim3(256), dim3(kMbBlock), h->mb_lds, h->stream, h->mb_sync.p);
        unsigned int per[8];
        HIP_CHECK(hipMemcpyAsync(per, h->mb_sync.p, sizeof per, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->blocks_probed = true;
        for (int x = 0; x < 8; ++x)
            if (per[x] != 32u) {
                h->blocks_failed = true;   // this device / partition mode: the fused kernel, from the first epoch on
                return false;
            }
    }
    MfBlockArgs a;
    a.blk_tile_ptr = h->mb_blk_tile_ptr.p; a.tile_ptr = h->mb_tile_ptr.p;
    a.b_u = h->mb_u.p; a.b_slot = h->mb_slot.p; a.b_r = h->mb_r.p;
    a.bin_ptr = h->mb_bin_ptr.p; a.bin_items = h->mb_bin_items.p;
    a.U = h->U.p; a.V = h->V.p; a.Bu = h->Bu.p; a.Bi = h->Bi.p; a.loss_acc = loss_slot;
    a.slot_cnt = h->mb_sync.p; a.bar = h->mb_sync.p + 8; a.abort = h->mb_sync.p + 16;
    a.wait_bound_ticks = (long long)prof_env_int("CORNAC_HIP_MF_BLOCKS_WAIT_S", 20) * 100000000ll;
    a.cap = h->mb_cap; a.k = h->k; a.use_bias = use_bias; a.lr = lr; a.reg = reg; a.mu = mu;
    a.Vx = h->mb_vx.p; a.Bix = h->mb_bix.p; a.n_items = (int32_t)h->n_items;
    MfVirtArgs va;
    va.V = h->V.p; va.Bi = h->Bi.p; va.bi_stride = 1; va.Vx = h->mb_vx.p; va.Bix = h->mb_bix.p;
    va.item = h->mb_split_item.p; va.ptr = h->mb_split_ptr.p; va.n_split = h->mb_n_split; va.k = h->k;
    HIP_CHECK(hipMemsetAsync(h->mb_sync.p, 0, 24 * sizeof(unsigned int), h->stream));
    for (int ph = 0; ph < 8; ++ph) {
        a.phase = ph;
        if (ph) HIP_CHECK(hipMemsetAsync(h->mb_sync.p, 0, 16 * sizeof(unsigned int), h->stream));  // (the abort word stays)
        if (h->mb_n_split && ph == 0) {  // the copies start the epoch equal to their rows (the tables may have been set since)
            va.merge = 0;
            hipLaunchKernelGGL(mf_virtual_merge_kernel, dim3(h->mb_n_split), dim3(kWave), 0, h->stream, va);
        }
        h->ktimer.before(h->stream);
        hipLaunchKernelGGL(kern, dim3(256), dim3(kMbBlock), h->mb_lds, h->stream, a);
        h->ktimer.after(h->stream);
        if (h->mb_n_split) {
            va.merge = 1;
            hipLaunchKernelGGL(mf_virtual_merge_kernel, dim3(h->mb_n_split), dim3(kWave), 0, h->stream, va);
        }
    }
    HIP_CHECK(hipGetLastError());
    unsigned int aborted = 0;
    HIP_CHECK(hipMemcpyAsync(&aborted, h->mb_sync.p + 16, sizeof aborted, hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    if (aborted) {
        h->blocks_failed = true;
        fail(CORNAC_HIP_ERR_HIP, "MF block-rotation kernel gave up (workgroup placement or a barrier bound); the handle "
             "uses the fused kernel from now on — re-run the call");
    }
    return true;
}

// The fused atomic kernel over ratings [s0, s0 + n) of the stored order (the whole epoch: s0 = 0, n = nnz, where user-row
// ownership applies too; a slice — the chunks of the multi-GPU driver — runs the all-atomic instantiation).
static void mf_launch_fused(cornac_hip_mf_t h, int64_t s0, int64_t n, float lr, float reg, float mu, int use_bias,
                            double *loss_slot) {
    const DeviceInfo &di = device_info(h->device);
    const int k = h->k;
    const bool whole = s0 == 0 && n == h->nnz;
    const bool owned = whole && k > 32 && k <= 256 && h->nnz >= (int64_t)di.cus * 8 * kWavesPerBlock * kWave;
    MfHogKernel kern = pick_mf_kernel(k, owned);
    if (h->hog_kernel != kern) {
        int per_cu = 0;
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kBlock, 0));
        h->hog_kernel = kern;
        h->hog_blocks_per_cu = std::max(1, std::min(per_cu, 8));
    }
    if (h->hog_form == 3 && !h->step_grid) {
        // A step handle (few item rows): the launch is THROTTLED to ~4 ratings in flight per item row — with every CU full
        // (~16 000 in flight) a block of 300..1 000 rows would take dozens of concurrent stale updates on EVERY row, and
        // training all of them through merged copies lags: the "align" merge of W copies whose steps agree is their MEAN, a
        // row learns at 1 / W of the pace.  Measured (profiles/r06_mf_step_throttle.log: 8 virtual ranks, 300-row blocks,
        // held-out RMSE after 4 epochs where one process has 0.5312; P in flight per row, a copy per C concurrent updates):
        // every row split 0.9647; P, C = 4, 16: 0.5345; 2, 16 and 4, 32 (the same copies): 0.5303; 1, 16: 0.5287; and the
        // Netflix shape's rank share (1 111-row blocks, 0.78 M ratings per step): 2, 16: 18.9 ms per epoch, 4, 32: 13.2 ms,
        // unthrottled with copies 11.4 ms, unthrottled without: inf.  Shipped: 4 per row, a copy per 32.
        const int per_wave = k <= 4 ? 32 : k <= 8 ? 16 : k <= 32 ? 8 : k <= 64 ? 4 : k <= 192 ? 2 : 1;   // TPW x UNR of pick_mf_kernel
        const int64_t per_wg = (int64_t)kWavesPerBlock * per_wave;
        const int64_t full = (int64_t)di.cus * h->hog_blocks_per_cu;
        const int64_t want = std::max<int64_t>(256, 4 * h->n_items);
        h->step_grid = (int)std::max<int64_t>(1, std::min<int64_t>(full, (want + per_wg - 1) / per_wg));
        h->step_inflight = (int64_t)h->step_grid * per_wg;
    }
    // hot items of a large problem train through copies of their rows (mf_blocks.inc, "virtual rows"): thousands of atomic
    // updates of one row computed from one stale copy overshoot and diverge (round 4 raised "diverged" here instead)
    mf_build_split(h);
    const int nv = h->mb_n_virtual;
    if (nv && !h->cid_split.p) {
        std::vector<int64_t> c((size_t)h->nnz);
        for (int64_t s = 0; s < h->nnz; ++s) c[(size_t)s] = mf_split_id(h, s, h->host_cid[(size_t)s]);
        h->cid_split.alloc((size_t)h->nnz);
        h->cid_split.upload(c.data(), (size_t)h->nnz, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
    }
    MfHogArgs a;
    a.rid = h->rid.p + s0; a.cid = (nv ? h->cid_split.p : h->cid.p) + s0; a.val = h->val.p + s0;
    a.own_u = nullptr; a.own_i = nullptr; a.own_r = nullptr; a.wave_ptr = nullptr;
    a.U = h->U.p; a.V = h->V.p; a.Bu = h->Bu.p;
    a.n_items = nv ? (int32_t)h->n_items : INT32_MAX;
    a.Vx_shifted = nv ? h->mb_vx.p - (size_t)h->n_items * (size_t)k : h->V.p;
    h->Bipad.ensure((size_t)(h->n_items + nv) * kBiasStride);
    a.Bi = h->Bipad.p;
    a.bstride = kBiasStride;
    a.loss_acc = loss_slot;
    a.n = n; a.k = k; a.use_bias = use_bias; a.lr = lr; a.reg = reg; a.mu = mu;
    int grid;
    if (owned) {
        grid = h->step_grid ? h->step_grid : di.cus * h->hog_blocks_per_cu;
        mf_build_ownership(h, (int64_t)grid * kWavesPerBlock);
        a.own_u = h->own_u.p; a.own_i = h->own_i.p; a.own_r = h->own_r.p; a.wave_ptr = h->wave_ptr.p;
    } else {
        const int64_t n_tiles = (a.n + kWave - 1) / kWave;
        const int64_t want_blocks = (n_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
        grid = (int)std::max<int64_t>(1, std::min<int64_t>(want_blocks, (int64_t)di.cus * h->hog_blocks_per_cu));
        if (h->step_grid) grid = std::min(grid, h->step_grid);
    }
    const unsigned bgrid = (unsigned)((h->n_items + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(bias_pad_kernel, dim3(bgrid), dim3(kBlock), 0, h->stream, h->Bi.p, h->Bipad.p, h->n_items);
    MfVirtArgs va;
    va.V = h->V.p; va.Bi = h->Bipad.p; va.bi_stride = kBiasStride; va.Vx = h->mb_vx.p;
    va.Bix = h->Bipad.p + (size_t)h->n_items * kBiasStride;   // the copies' bias lines follow the items' in the padded table
    va.item = h->mb_split_item.p; va.ptr = h->mb_split_ptr.p; va.n_split = h->mb_n_split; va.k = k;
    if (nv) {  // the copies start the launch equal to their rows
        va.merge = 0;
        hipLaunchKernelGGL(mf_virtual_merge_kernel, dim3(h->mb_n_split), dim3(kWave), 0, h->stream, va);
    }
    h->ktimer.before(h->stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, h->stream, a);
    h->ktimer.after(h->stream);
    if (nv) {  // ... and are folded back into them after it ("align": sum of the copies' deltas x min(1, sum |d|^2 / |sum d|^2))
        va.merge = 1;
        hipLaunchKernelGGL(mf_virtual_merge_kernel, dim3(h->mb_n_split), dim3(kWave), 0, h->stream, va);
    }
    hipLaunchKernelGGL(bias_unpad_kernel, dim3(bgrid), dim3(kBlock), 0, h->stream, h->Bipad.p, h->Bi.p, h->n_items);
    HIP_CHECK(hipGetLastError());
}

static void mf_epoch_hogwild(cornac_hip_mf_t h, float lr, float reg, float mu, int use_bias, double *loss_slot) {
    if (mf_uses_blocks(h) && mf_epoch_blocks(h, lr, reg, mu, use_bias, loss_slot)) {
        h->hog_form_used = 2;
        return;
    }
    h->hog_form_used = 1;
    mf_launch_fused(h, 0, h->nnz, lr, reg, mu, use_bias, loss_slot);
}

extern "C" {

int cornac_hip_mf_create(cornac_hip_mf_t *out, int device, int64_t n_users, int64_t n_items, int k,
                         const int64_t *rid, const int64_t *cid, const float *val, int64_t nnz) {
    return guarded([&] {
        REQUIRE(out != nullptr, "out handle pointer is NULL");
        *out = nullptr;
        REQUIRE(n_users > 0 && n_items > 0 && k > 0, "n_users, n_items and k must be positive");
        REQUIRE(n_users < (int64_t(1) << 31) && n_items < (int64_t(1) << 31), "index range exceeds int32");
        REQUIRE(nnz > 0 && rid && cid && val, "empty or NULL rating arrays");
        for (int64_t s = 0; s < nnz; ++s)
            REQUIRE(rid[s] >= 0 && rid[s] < n_users && cid[s] >= 0 && cid[s] < n_items,
                    "rating %lld has an out-of-range index", (long long)s);
        use_device(device);
        std::unique_ptr<cornac_hip_mf> h(new cornac_hip_mf());
        h->device = device; h->n_users = n_users; h->n_items = n_items; h->k = k; h->nnz = nnz;
        HIP_CHECK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
        h->stream = h->own_stream;
        h->rid.alloc((size_t)nnz); h->cid.alloc((size_t)nnz); h->val.alloc((size_t)nnz);
        h->rid.upload(rid, (size_t)nnz, h->stream);
        h->cid.upload(cid, (size_t)nnz, h->stream);
        h->val.upload(val, (size_t)nnz, h->stream);
        h->host_rid.assign(rid, rid + nnz);
        h->host_cid.assign(cid, cid + nnz);
        h->host_val.assign(val, val + nnz);
        h->U.alloc((size_t)n_users * k); h->V.alloc((size_t)n_items * k);
        h->Bu.alloc((size_t)n_users); h->Bi.alloc((size_t)n_items);
        HIP_CHECK(hipMemsetAsync(h->U.p, 0, h->U.n * 4, h->stream));
        HIP_CHECK(hipMemsetAsync(h->V.p, 0, h->V.n * 4, h->stream));
        HIP_CHECK(hipMemsetAsync(h->Bu.p, 0, h->Bu.n * 4, h->stream));
        HIP_CHECK(hipMemsetAsync(h->Bi.p, 0, h->Bi.n * 4, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        *out = h.release();
    });
}

int cornac_hip_mf_destroy(cornac_hip_mf_t h) {
    return guarded([&] {
        if (!h) return;
        (void)hipSetDevice(h->device);
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
        delete h;
    });
}

int cornac_hip_mf_set_factors(cornac_hip_mf_t h, const float *U, const float *V, const float *Bu, const float *Bi) {
    return guarded([&] {
        mf_check(h);
        if (U) h->U.upload(U, (size_t)h->n_users * h->k, h->stream);
        if (V) h->V.upload(V, (size_t)h->n_items * h->k, h->stream);
        if (Bu) h->Bu.upload(Bu, (size_t)h->n_users, h->stream);
        if (Bi) h->Bi.upload(Bi, (size_t)h->n_items, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}

int cornac_hip_mf_get_factors(cornac_hip_mf_t h, float *U, float *V, float *Bu, float *Bi) {
    return guarded([&] {
        mf_check(h);
        if (U) h->U.download(U, (size_t)h->n_users * h->k, h->stream);
        if (V) h->V.download(V, (size_t)h->n_items * h->k, h->stream);
        if (Bu) h->Bu.download(Bu, (size_t)h->n_users, h->stream);
        if (Bi) h->Bi.download(Bi, (size_t)h->n_items, h->stream);
        HIP_CHECK(hipStreamSynchronize(h->stream));
    });
}

// ---- multi-GPU driver surface (cornac_amd/dist.py ShardedMfTrainer): item side in caller-owned device memory, a caller
// stream, and epochs enqueued in slices without host synchronisation -----------------------------------------------------
int cornac_hip_mf_bind_items(cornac_hip_mf_t h, float *dV, float *dBi) {
    return guarded([&] {
        mf_check(h);
        REQUIRE(dV && dBi, "dV and dBi are required");
        // (the first bind frees the handle's own tables: wait for whatever still reads them; a re-bind from one caller buffer to
        // another only changes what LATER launches see — the block rotation re-binds before every step without a host wait)
        if (h->V.owned || h->Bi.owned) HIP_CHECK(hipStreamSynchronize(h->stream));
        h->V.bind(dV, (size_t)h->n_items * h->k);
        h->Bi.bind(dBi, (size_t)h->n_items);
    });
}

int cornac_hip_mf_bind_users(cornac_hip_mf_t h, float *dU, float *dBu) {
    return guarded([&] {
        mf_check(h);
        REQUIRE(dU && dBu, "dU and dBu are required");
        if (h->U.owned || h->Bu.owned) HIP_CHECK(hipStreamSynchronize(h->stream));
        h->U.bind(dU, (size_t)h->n_users * h->k);
        h->Bu.bind(dBu, (size_t)h->n_users);
    });
}

int cornac_hip_mf_set_stream(cornac_hip_mf_t h, void *hip_stream) {
    return guarded([&] {
        mf_check(h);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    });
}

int cornac_hip_mf_epoch_enqueue(cornac_hip_mf_t h, int part, int n_parts, float lr, float reg, float mu, int use_bias) {
    return guarded([&] {
        mf_check(h);
        REQUIRE(n_parts >= 1 && part >= 0 && part < n_parts, "part %d of %d", part, n_parts);
        h->loss.ensure(1);
        if (!h->enqueue_open) {
            HIP_CHECK(hipMemsetAsync(h->loss.p, 0, sizeof(double), h->stream));
            h->enqueue_open = true;
        }
        if (n_parts == 1) {
            mf_epoch_hogwild(h, lr, reg, mu, use_bias, h->loss.p);
            return;
        }
        const int64_t s0 = (int64_t)(((unsigned __int128)h->nnz * (unsigned)part) / (unsigned)n_parts);
        const int64_t s1 = (int64_t)(((unsigned __int128)h->nnz * (unsigned)(part + 1)) / (unsigned)n_parts);
        h->hog_form_used = 1;
        if (s1 > s0) mf_launch_fused(h, s0, s1 - s0, lr, reg, mu, use_bias, h->loss.p);
    });
}

int cornac_hip_mf_sync(cornac_hip_mf_t h, double *sq_err_sum) {
    return guarded([&] {
        mf_check(h);
        double l = 0.0;
        if (h->enqueue_open) HIP_CHECK(hipMemcpyAsync(&l, h->loss.p, sizeof l, hipMemcpyDeviceToHost, h->stream));
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->enqueue_open = false;
        if (sq_err_sum) *sq_err_sum = l;
        REQUIRE(std::isfinite(l), "the enqueued MF slices diverged: non-finite sum of squared errors (lower the learning rate)");
    });
}

int cornac_hip_mf_fit(cornac_hip_mf_t h, int max_iter, float lr, float reg, float mu, int use_bias, int early_stop,
                      int mode, float *loss_per_epoch, int *epochs_run) {
    return guarded([&] {
        mf_check(h);
        REQUIRE(max_iter >= 0, "max_iter must be >= 0");
        REQUIRE(mode == CORNAC_HIP_MODE_DETERMINISTIC || mode == CORNAC_HIP_MODE_HOGWILD, "unknown mode %d", mode);
        for (double &t : h->timing) t = 0;
        Timer total;
        h->loss.ensure((size_t)std::max(max_iter, 1));
        HIP_CHECK(hipMemsetAsync(h->loss.p, 0, h->loss.n * sizeof(double), h->stream));
        bool chain = mode == CORNAC_HIP_MODE_DETERMINISTIC && mf_uses_chain(h) && !h->chain_refused &&
                     !prof_env_set("CORNAC_HIP_MF_LEVELS");
        if (mode == CORNAC_HIP_MODE_DETERMINISTIC) {
            if (chain) mf_build_chain(h); else mf_build_schedule(h);
        }
        Timer t_k;
        float loss = 0.f, last_loss = 0.f;
        int e = 0;
        for (; e < max_iter; ++e) {
            if (chain && !mf_epoch_chain(h, lr, reg, mu, use_bias, h->loss.p + e)) {
                chain = false;              // the dataflow launch was refused before anything ran: the level schedule,
                h->chain_refused = true;    // for this and every later fit of the handle
                mf_build_schedule(h);
                mf_epoch_deterministic(h, lr, reg, mu, use_bias, h->loss.p + e);
            } else if (!chain) {
                if (mode == CORNAC_HIP_MODE_DETERMINISTIC) mf_epoch_deterministic(h, lr, reg, mu, use_bias, h->loss.p + e);
                else mf_epoch_hogwild(h, lr, reg, mu, use_bias, h->loss.p + e);
            }
            if (early_stop) {  // needs this epoch's loss on the host (backend_cpu.pyx:89-93)
                double l;
                HIP_CHECK(hipMemcpyAsync(&l, h->loss.p + e, sizeof l, hipMemcpyDeviceToHost, h->stream));
                HIP_CHECK(hipStreamSynchronize(h->stream));
                last_loss = loss;
                loss = (float)(0.5 * l);
                if (fabsf(loss - last_loss) < 1e-5f) {
                    ++e;
                    break;
                }
            }
        }
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->timing[2] = t_k.ms();
        if (e > 0) {
            std::vector<double> l((size_t)e);
            HIP_CHECK(hipMemcpy(l.data(), h->loss.p, sizeof(double) * (size_t)e, hipMemcpyDeviceToHost));
            if (loss_per_epoch)
                for (int t = 0; t < e; ++t) loss_per_epoch[t] = (float)(0.5 * l[(size_t)t]);
            if (epochs_run) *epochs_run = e;
            h->timing[3] = total.ms();
            // The racy form has no sequential counterpart to stay faithful to: a non-finite loss means the run diverged
            // (a very popular row under the atomic kernel takes many stale steps at once: profiles/r03_mf_zipf.log), and a
            // NaN model must not be returned silently.  (The sequential mode reproduces the reference, NaNs included.)
            if (mode == CORNAC_HIP_MODE_HOGWILD)
                for (int t = 0; t < e; ++t)
                    REQUIRE(std::isfinite(l[(size_t)t]),
                            "the hogwild MF fit diverged: non-finite loss in epoch %d of %d (learning rate %g; the tables now hold "
                            "non-finite values — lower the learning rate or use mode = deterministic)", t + 1, e, (double)lr);
        }
        if (epochs_run) *epochs_run = e;
        h->timing[3] = total.ms();
    });
}

int cornac_hip_mf_fit_sgd(int device, const int64_t *rid, const int64_t *cid, const float *val, int64_t nnz, float *U,
                          float *V, float *Bu, float *Bi, int64_t n_users, int64_t n_items, int k, float lr, float reg,
                          float mu, int max_iter, int use_bias, int early_stop, int mode, float *loss_per_epoch,
                          int *epochs_run) {
    cornac_hip_mf_t h = nullptr;
    int rc = cornac_hip_mf_create(&h, device, n_users, n_items, k, rid, cid, val, nnz);
    if (rc != CORNAC_HIP_OK) return rc;
    rc = cornac_hip_mf_set_factors(h, U, V, Bu, Bi);
    if (rc == CORNAC_HIP_OK)
        rc = cornac_hip_mf_fit(h, max_iter, lr, reg, mu, use_bias, early_stop, mode, loss_per_epoch, epochs_run);
    if (rc == CORNAC_HIP_OK) rc = cornac_hip_mf_get_factors(h, U, V, Bu, Bi);
    std::string keep = cornac_hip_last_error();
    cornac_hip_mf_destroy(h);
    if (rc != CORNAC_HIP_OK) chip::set_last_error(keep);
    return rc;
}

int cornac_hip_mf_kernel_timing(cornac_hip_mf_t h, int enable, double *total_ms, int64_t *launches) {
    return guarded([&] {
        mf_check(h);
        HIP_CHECK(hipStreamSynchronize(h->stream));
        h->ktimer.collect(total_ms, launches);
        h->ktimer.enabled = enable != 0;
    });
}

int cornac_hip_mf_hogwild_form(cornac_hip_mf_t h, int form) {
    return guarded([&] {
        mf_check(h);
        REQUIRE(form >= 0 && form <= 3, "form must be 0 (automatic), 1 (fused atomic kernel), 2 (block rotation) or 3 (fused "
                "kernel, popular rows split into copies at any size)");
        REQUIRE(!h->split_built || (form == 3) == (h->hog_form == 3),
                "form 3 changes which item rows train through copies: choose it before the handle's first epoch");
        h->hog_form = form;
        if (form == 2) h->blocks_failed = false;
    });
}

int cornac_hip_mf_hogwild_stats(cornac_hip_mf_t h, int64_t *out4) {
    return guarded([&] {
        mf_check(h);
        REQUIRE(out4 != nullptr, "out4 is NULL");
        out4[0] = h->hog_form_used;
        out4[1] = h->blocks_built ? (int64_t)h->mb_tile_ptr.n - 1 : 0;
        out4[2] = h->blocks_built ? h->mb_cap : 0;
        out4[3] = h->blocks_failed ? 1 : 0;
    });
}

int cornac_hip_mf_last_timing(cornac_hip_mf_t h, double *ms4) {
    return guarded([&] {
        REQUIRE(h && ms4, "NULL argument");
        for (int t = 0; t < 4; ++t) ms4[t] = h->timing[t];
    });
}
}

#include "mf_minibatch.inc"
