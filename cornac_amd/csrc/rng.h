// Device-side random number generation for the triplet sampler (gfx950).
//
//  (1) Bit-faithful reproduction of the reference's seeded sampler: boost::random::mt19937 +
//      boost 1.72 uniform_int_distribution<long>(0, hi)  (RNGVector, cornac/models/bpr/recom_bpr.pyx:54-62;
//      algorithm: cornac/utils/external/boost/random/uniform_int_distribution.hpp:188-227).
//      One workgroup regenerates the 624-word state in three data-parallel phases
//      ([0,227) | [227,454) | [454,624): each phase only depends on the previous one), tempers all
//      624 words at once, applies the bucket/rejection test and stream-compacts the accepted draws,
//      so sample s receives the s-th ACCEPTED draw exactly like the sequential generator.
//  (2) Counter-based Philox4x32-10 for the hogwild (throughput) mode, where — like the
//      reference's multi-thread mode — no particular sequence is contractual.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace chip {

// ---------------------------------------------------------------- Philox4x32-10 -----------------
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// uniform in [0, n): Lemire multiply-shift; a first word in the biased zone falls back to the second.
__host__ __device__ inline uint32_t lemire_bounded2(uint32_t wa, uint32_t wb, uint32_t n, uint32_t thresh) {
    uint64_t m = (uint64_t)wa * n;
    if ((uint32_t)m < thresh) m = (uint64_t)wb * n;
    return (uint32_t)(m >> 32);
}
__host__ __device__ inline uint32_t lemire_thresh(uint32_t n) { return (uint32_t)(0u - n) % n; }

// ---------------------------------------------------------------- MT19937 -----------------------
struct MtStreamParams {
    uint32_t *state;   // [624] device, persistent generator state
    int32_t *idx;      // [1]   device, next unread word (624 => regenerate first)
    uint32_t *out;     // accepted draws
    int64_t need;      // number of accepted draws to produce
    uint32_t range;    // hi  (draws are in [0, hi])
    uint32_t bucket;   // boost bucket size
    int32_t out_stride;  // write out[t * out_stride + out_offset]  (interleaving for the shared stream)
    int32_t out_offset;
};

constexpr int MT_N = 624;
constexpr int MT_M = 397;
constexpr int MT_THREADS = 640;  // one thread per state word (10 waves)

__device__ inline uint32_t mt_twist_word(uint32_t cur, uint32_t nxt, uint32_t far) {
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ inline uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// grid = number of streams (one workgroup each), block = MT_THREADS
__global__ __launch_bounds__(MT_THREADS) void mt19937_draw_kernel(const MtStreamParams *params) {
    const MtStreamParams P = params[blockIdx.x];
    __shared__ uint32_t st[2][MT_N];
    __shared__ int wave_cnt[MT_THREADS / 64];
    __shared__ int sh_stop_idx;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    int cur = 0;
    if (tid < MT_N) st[0][tid] = P.state[tid];
    int idx = *P.idx;
    int64_t produced = 0;
    __syncthreads();
    while (produced < P.need) {
        if (idx >= MT_N) {
            const uint32_t *o = st[cur];
            uint32_t *nw = st[cur ^ 1];
            if (tid < MT_N - MT_M) nw[tid] = mt_twist_word(o[tid], o[tid + 1], o[tid + MT_M]);
            __syncthreads();
            if (tid >= MT_N - MT_M && tid < 2 * (MT_N - MT_M))
                nw[tid] = mt_twist_word(o[tid], o[tid + 1], nw[tid - (MT_N - MT_M)]);
            __syncthreads();
            if (tid >= 2 * (MT_N - MT_M) && tid < MT_N) {
                const uint32_t nxt = (tid == MT_N - 1) ? nw[0] : o[tid + 1];
                nw[tid] = mt_twist_word(o[tid], nxt, nw[tid - (MT_N - MT_M)]);
            }
            __syncthreads();
            cur ^= 1;
            idx = 0;
        }
        // temper + bucket/rejection test for the unread words [idx, 624)
        uint32_t r = 0;
        bool acc = false;
        if (tid >= idx && tid < MT_N) {
            r = mt_temper(st[cur][tid]) / P.bucket;
            acc = r <= P.range;
        }
        const unsigned long long m = __ballot(acc);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        if (tid == 0) sh_stop_idx = MT_N;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < MT_THREADS / 64; ++w) {
            const int c = wave_cnt[w];
            if (w < wave) before += c;
            total += c;
        }
        const int my_pos = before + __popcll(m & ((1ull << lane) - 1ull));
        const int64_t room = P.need - produced;
        if (acc && (int64_t)my_pos < room) {
            P.out[(produced + my_pos) * P.out_stride + P.out_offset] = r;
            if ((int64_t)my_pos == room - 1) sh_stop_idx = tid + 1;  // last draw consumed by this call
        }
        __syncthreads();
        if ((int64_t)total >= room) {
            produced = P.need;
            idx = sh_stop_idx;
        } else {
            produced += total;
            idx = MT_N;
        }
        __syncthreads();
    }
    if (tid < MT_N) P.state[tid] = st[cur][tid];
    if (tid == 0) *P.idx = idx;
}

}  // namespace chip
