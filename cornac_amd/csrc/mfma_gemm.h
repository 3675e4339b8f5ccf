// LDS-tiled fp32 MFMA block GEMM shared by the dense training paths (WMF, VBPR): one 128 x 128 block of
// C = A B per 256-thread workgroup on v_mfma_f32_32x32x2_f32, register-prefetched double-buffered LDS tiles.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace chip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWb = 256;        // threads per workgroup: 4 waves as 2 x 2, each wave a 64 x 64 block of C
constexpr int kBM = 128, kBN = 128, kBK = 16;
constexpr int kLdT = kBM + 4;   // LDS row length of a k-major tile (+4 floats: rows 16-byte aligned, banks skewed)

struct GemmSmem {
    float a[2][kBK][kLdT];
    float b[2][kBK][kLdT];
};

// One 128 x 128 block of  C = A B  over k in [k_begin, k_end).
//   A element (m, kk) at A[m * a_sm + kk * a_sk], B element (kk, n) at B[kk * b_sk + n * b_sn];
//   A_KC: a_sk == 1 (k contiguous) else a_sm == 1;   B_NC: b_sn == 1 (n contiguous) else b_sk == 1.
// Out-of-range rows/cols/k read as zero.  acc[i][j] is the 32 x 32 block (i, j) of this wave's 64 x 64 part:
// register r of lane l holds C[row = wm*64 + i*32 + (r&3) + 8*(r>>2) + 4*(l>>5)][col = wn*64 + j*32 + (l&31)].
//   ZERO = false accumulates into acc instead of starting from zero; A_NT reads A with L1-bypassing (nt) loads —
//   for an A operand the same workgroup has just written to global memory.
template <bool A_KC, bool B_NC, bool ZERO = true, bool A_NT = false>
__device__ __forceinline__ void gemm_block(const float *__restrict__ A, int64_t a_sm, int64_t a_sk,
                                           const float *__restrict__ B, int64_t b_sk, int64_t b_sn, int64_t M,
                                           int64_t N, int64_t m0, int64_t n0, int64_t k_begin, int64_t k_end,
                                           GemmSmem &sm, f32x16 (&acc)[2][2], bool skip_loads = false) {
    // (an opaque copy of the thread index: inside a persistent tile loop the addresses derived from it would otherwise be
    // hoisted out of the loop as invariants and stay live — dozens of registers — across every other phase of the tile)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    if (ZERO) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    f32x4 ra[2], rb[2];
    auto ld4 = [](const float *p) {
        return A_NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p)) : *reinterpret_cast<const f32x4 *>(p);
    };
    auto ld1 = [](const float *p) { return A_NT ? __builtin_nontemporal_load(p) : *p; };
    auto load_tile = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (A_KC) {  // thread: one row, 4 consecutive k
                const int64_t m = m0 + (tid >> 2) + 64 * i, kk = k0 + (tid & 3) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m < M) {
                    const float *p = A + m * a_sm + kk;
                    if (kk + 3 < k_end && ((a_sm | kk) & 3) == 0) {
                        v = ld4(p);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (kk + q < k_end) v[q] = ld1(p + q);
                    }
                }
                ra[i] = v;
            } else {  // thread: one k, 4 consecutive m
                const int64_t kk = k0 + (tid >> 5) + 8 * i, m = m0 + (tid & 31) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kk < k_end) {
                    const float *p = A + kk * a_sk + m;
                    if (m + 3 < M && ((a_sk | m) & 3) == 0) {
                        v = ld4(p);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (m + q < M) v[q] = ld1(p + q);
                    }
                }
                ra[i] = v;
            }
            if (B_NC) {  // thread: one k, 4 consecutive n
                const int64_t kk = k0 + (tid >> 5) + 8 * i, n = n0 + (tid & 31) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kk < k_end) {
                    const float *p = B + kk * b_sk + n;
                    if (n + 3 < N && ((b_sk | n) & 3) == 0) {
                        v = *reinterpret_cast<const f32x4 *>(p);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (n + q < N) v[q] = p[q];
                    }
                }
                rb[i] = v;
            } else {  // thread: one column n, 4 consecutive k
                const int64_t n = n0 + (tid >> 2) + 64 * i, kk = k0 + (tid & 3) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < N) {
                    const float *p = B + n * b_sn + kk;
                    if (kk + 3 < k_end && ((b_sn | kk) & 3) == 0) {
                        v = *reinterpret_cast<const f32x4 *>(p);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (kk + q < k_end) v[q] = p[q];
                    }
                }
                rb[i] = v;
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (A_KC) {
                const int m = (tid >> 2) + 64 * i, kq = (tid & 3) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) sm.a[buf][kq + q][m] = ra[i][q];
            } else {
                const int kk = (tid >> 5) + 8 * i, m = (tid & 31) * 4;
                *reinterpret_cast<f32x4 *>(&sm.a[buf][kk][m]) = ra[i];
            }
            if (B_NC) {
                const int kk = (tid >> 5) + 8 * i, n = (tid & 31) * 4;
                *reinterpret_cast<f32x4 *>(&sm.b[buf][kk][n]) = rb[i];
            } else {
                const int n = (tid >> 2) + 64 * i, kq = (tid & 3) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) sm.b[buf][kq + q][n] = rb[i][q];
            }
        }
    };

    const int64_t n_steps = (k_end - k_begin + kBK - 1) / kBK;
    if (n_steps <= 0) return;
    load_tile(k_begin);
    store_tile(0);
    __syncthreads();
    for (int64_t s = 0; s < n_steps; ++s) {
        const int buf = (int)(s & 1);
        if (s + 1 < n_steps && !skip_loads) load_tile(k_begin + (s + 1) * kBK);   // (skip_loads: profile builds' latency ablation)
        // the fragments of k pair t + 1 are requested before the four MFMAs of pair t issue: one wave keeps the matrix
        // pipe busy on its own (the LDS latency hides behind 256 MFMA cycles instead of stalling every fourth MFMA)
        float fa[2][2], fb[2][2];
        auto frag = [&](int t, int slot) {
            fa[slot][0] = sm.a[buf][2 * t + half][wm * 64 + l31];
            fa[slot][1] = sm.a[buf][2 * t + half][wm * 64 + 32 + l31];
            fb[slot][0] = sm.b[buf][2 * t + half][wn * 64 + l31];
            fb[slot][1] = sm.b[buf][2 * t + half][wn * 64 + 32 + l31];
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
        if (s + 1 < n_steps) store_tile(buf ^ 1);  // the other buffer was last read before the previous barrier
        __syncthreads();
    }
}

// visits every accumulator element of this thread: f(row, col, value&)
template <class F>
__device__ __forceinline__ void for_each_acc(f32x16 (&acc)[2][2], int64_t m0, int64_t n0, F &&f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int64_t col = n0 + wn * 64 + j * 32 + l31;
                f(row, col, acc[i][j][r]);
            }
}

// the same visit in four 16-element blocks the scheduler may not interleave: an epilogue that loads several values per
// element (Adam: U, m, v) otherwise has all 64 x 3 loads hoisted and spills
template <class F>
__device__ __forceinline__ void for_each_acc_blocked(f32x16 (&acc)[2][2], int64_t m0, int64_t n0, F &&f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int64_t col = n0 + wn * 64 + j * 32 + l31;
                f(row, col, acc[i][j][r]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

}  // namespace chip

namespace chip {
// Visit for epilogues inside a PERSISTENT tile loop: f(offset, row, col, value) with offset = row * stride + col as a
// 32-bit element offset rebuilt from a per-lane base the optimiser cannot see through.  Without the opaque base the 64
// per-element addresses are loop-invariant, get hoisted out of the tile loop and stay live in 128+ registers.
template <class F>
__device__ __forceinline__ void for_each_acc_local(f32x16 (&acc)[2][2], int stride, F &&f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    int row_base = wm * 64 + 4 * half, col_base = wn * 64 + l31;
    asm volatile("" : "+v"(row_base), "+v"(col_base));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2), col = col_base + j * 32;
                f(row * stride + col, row, col, acc[i][j][r]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}
}  // namespace chip
