// Micro-benchmark (not part of the product): sustained rate of v_mfma_f32_32x32x2_f32 on gfx950 for the issue
// patterns the fused rank kernel can use: one dependent accumulator chain per wave vs two independent chains,
// at 1, 2 and 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void probe(float *out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f32x16 acc0 = {0}, acc1 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (CHAINS == 1 || (t & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 123.456f) out[0] = s;
}

// the rank kernel's mix: a dependent chain with VALU compares (VPM per MFMA) and optional LDS fragment reads
template <int VPM, bool LDS>
__global__ __launch_bounds__(256) void probe_mix(float *out, int iters) {
    __shared__ float tile[32][66];
    for (int i = threadIdx.x; i < 32 * 66; i += 256) (&tile[0][0])[i] = i * 1e-4f;
    __syncthreads();
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const int col = threadIdx.x & 31, half = (threadIdx.x >> 5) & 1;
    f32x16 acc0 = {0}, acc1 = {0};
    float thr = 1e30f, base = 0.5f;
    unsigned hits = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (LDS) b = tile[col][2 * t + half];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            if (t < 16) {
#pragma unroll
                for (int v = 0; v < VPM / 4; ++v) {
                    const float sc = (base + (float)v) + acc1[t];
                    hits |= sc >= thr ? (1u << t) : 0u;
                }
            }
        }
        acc1 = acc0;
        acc0 = f32x16{0};
    }
    float s = (float)hits;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 123.456f) out[0] = s;
}

template <int VPM, bool LDS>
void run_mix(int wgs_per_cu, const char *name) {
    float *out; hipMalloc(&out, 4);
    const int iters = 2000, blocks = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe_mix<VPM, LDS><<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe_mix<VPM, LDS><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 32 * 4096.0;
    printf("%-40s %d waves/SIMD  %7.3f ms  %6.1f TFLOP/s\n", name, wgs_per_cu, ms, flops / ms * 1e-9);
    hipFree(out);
}

template <int CHAINS>
void run(int wgs_per_cu, const char *name) {
    float *out; hipMalloc(&out, 4);
    const int iters = 4000, blocks = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<CHAINS><<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<CHAINS><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 32 * 4096.0;
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 32 * wgs_per_cu);  // per MFMA per SIMD at 2.4 GHz
    printf("%-28s %d waves/SIMD  %7.3f ms  %6.1f TFLOP/s  (%.1f cycles@2.4GHz per MFMA per SIMD)\n", name, wgs_per_cu, ms,
           flops / ms * 1e-9, cyc);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<1>(w, "one dependent chain");
        run<2>(w, "two independent chains");
    }
    for (int w : {1, 2}) {
        run_mix<4, false>(w, "chain + 4 VALU/MFMA (first 16)");
        run_mix<16, false>(w, "chain + 16 VALU/MFMA (first 16)");
        run_mix<4, true>(w, "chain + 4 VALU/MFMA + LDS fragments");
        run_mix<16, true>(w, "chain + 16 VALU/MFMA + LDS fragments");
    }
    return 0;
}
