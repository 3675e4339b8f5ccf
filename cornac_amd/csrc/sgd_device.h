// Device building blocks shared by the BPR and MF SGD kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// The SGD arithmetic mirrors the reference's unfused float expressions; never let the compiler
// contract a*b+c into an fma in these kernels (explicit fmaf is used where a fused op is wanted).
#pragma clang fp contract(off)

// Ablation bits of the SGD kernels (hogwild_flags bits 8..15: 1 no membership test, 2 no stores, 4 no row loads,
// 8 no bias traffic, ...) exist in -DCORNAC_PROFILE builds only (make PROFILE=1); the shipped kernels compile the
// branches away.
#ifdef CORNAC_PROFILE
#define HOG_ABLATE(args, bit) (((args).ablate & (bit)) != 0)
#else
#define HOG_ABLATE(args, bit) false
#endif

namespace chip {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// butterfly sum over the G lanes of a lane group (G = power of two <= 64); every lane gets the sum
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// Sum over the 64 lanes of a wave, every lane gets it — without the LDS crossbar.  __shfl_xor lowers to ds_bpermute_b32 on
// gfx950: six DEPENDENT LDS round trips per sum (and the LDS is where the LDS-bin kernel's item rows live); here four
// v_add_f32 with DPP modifiers sum each row of 16 lanes (quad swaps, then the half-row and row mirrors), four v_readlane
// fetch the row sums and three adds finish.  Measured on the ML-20M headline (profiles/r06_headline_ablation.log): the
// six-step shuffle sum cost 0.9 ms of a 5.8 ms epoch.  (Different summation order from group_sum: hogwild forms only.)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    auto f2i = [](float x) { return __builtin_bit_cast(int, x); };
    auto i2f = [](int x) { return __builtin_bit_cast(float, x); };
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0xB1, 0xf, 0xf, false));    // quad_perm:[1,0,3,2]
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x4E, 0xf, 0xf, false));    // quad_perm:[2,3,0,1]
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    v += i2f(__builtin_amdgcn_update_dpp(0, f2i(v), 0x140, 0xf, 0xf, false));   // row_mirror
    const float r0 = i2f(__builtin_amdgcn_readlane(f2i(v), 0)), r1 = i2f(__builtin_amdgcn_readlane(f2i(v), 16));
    const float r2 = i2f(__builtin_amdgcn_readlane(f2i(v), 32)), r3 = i2f(__builtin_amdgcn_readlane(f2i(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// acc = ((acc + p[0]) + p[1]) + ... + p[lim-1] over the lanes of a G-lane group, in lane order (the reference's
// sequential float sum over factors).  With one group per wave every term is read with v_readlane (constant
// lane after unrolling); smaller groups need a per-group source lane and go through the cross-lane network.
template <int G>
__device__ __forceinline__ float ordered_lane_sum(float acc, float p, int lim) {
    if (G == kWave) {
#pragma unroll
        for (int l = 0; l < kWave; ++l)
            if (l < lim) acc = acc + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), l));
    } else {
        for (int l = 0; l < lim; ++l) acc = acc + __shfl(p, l, G);
    }
    return acc;
}

// the same for either table type (double: lane broadcasts of the two halves)
template <int G>
__device__ __forceinline__ float ordered_lane_sum_t(float acc, float p, int lim) { return ordered_lane_sum<G>(acc, p, lim); }
template <int G>
__device__ __forceinline__ double ordered_lane_sum_t(double acc, double p, int lim) {
    for (int l = 0; l < lim; ++l) acc = acc + __shfl(p, l, G);
    return acc;
}

// sigmoid(-score) exactly as the reference evaluates it (cornac/models/bpr/recom_bpr.pyx:250):
// exp on a float, then 1.0/(1.0+e) in double, rounded to float on assignment.
__device__ __forceinline__ float sigmoid_neg_exact(float score) {
    const float e = (float)exp((double)score);
    return (float)(1.0 / (1.0 + (double)e));
}
__device__ __forceinline__ float sigmoid_neg_exact_t(float score) { return sigmoid_neg_exact(score); }
__device__ __forceinline__ double sigmoid_neg_exact_t(double score) { return 1.0 / (1.0 + exp(score)); }  // all-double locals
__device__ __forceinline__ float sigmoid_neg_fast(float score) {
    return __frcp_rn(1.0f + __expf(score));
}

// device-scope fp32 atomic add without return (global_atomic_add_f32)
__device__ __forceinline__ void atomic_add_f32(float *p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// write-through (sc1) store: visible to the other XCDs' L2s without a release fence
__device__ __forceinline__ void store_f32_agent(float *p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// L1-bypassing loads of rows that other CUs update concurrently (hogwild)
__device__ __forceinline__ v4f load_row4_fresh(const float *p) {
    return __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
}
__device__ __forceinline__ float load_f32_fresh(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The hogwild kernels address item biases through a padded table (one bias per 128-byte line): a
// dense 4-byte-per-item table concentrates all bias atomics on a handful of memory channels and
// serialises unrelated items that share a line (measured: half of a BPR epoch at ML-20M shape).
constexpr int kBiasStride = 32;
static __global__ __launch_bounds__(kBlock) void bias_pad_kernel(const float *__restrict__ dense,
                                                                 float *__restrict__ padded, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) padded[i * kBiasStride] = dense[i];
}
static __global__ __launch_bounds__(kBlock) void bias_unpad_kernel(const float *__restrict__ padded,
                                                                   float *__restrict__ dense, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dense[i] = padded[i * kBiasStride];
}

// is `col` present in the sorted CSR row [lo, hi)?  (has_non_zero, recom_bpr.pyx:46-51)
__device__ __forceinline__ bool csr_row_contains(const int32_t *__restrict__ indices, int32_t lo, int32_t hi,
                                                 int32_t col) {
    const int32_t end = hi;
    while (lo < hi) {
        const int32_t mid = lo + ((hi - lo) >> 1);
        if (indices[mid] < col) lo = mid + 1; else hi = mid;
    }
    return lo < end && indices[lo] == col;
}

}  // namespace chip
