// Micro-benchmark (not part of the product): throughput of fp32 row updates on gfx950 for the access
// shapes the BPR scatter can use.  Rows of 64 floats (256 B) picked pseudo-randomly from a table.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_probe.hip -o tools/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// mode 0: lane = float, 1 row per wave-instruction (2 full lines), agent atomics
// mode 1: 16 lanes x float4 per row, 4 rows per wave, 4 atomic instrs per lane (current kernel shape)
// mode 2: as 0 but plain load+store RMW
// mode 3: as 1 but plain v4f load+store RMW
// mode 4: as 0, workgroup-scope atomics
// mode 5: as 0 but f64 atomics on 32 lanes x double (same bytes)
// mode 6: as 0 but nt load + sc1 (atomic-store) write-through
// mode 7: load only (nt), lane = float
// PRIV: every XCD (block b runs on XCD b % 8) touches only its own eighth of the table
template <int MODE, bool PRIV = false>
__global__ __launch_bounds__(256) void probe(float *tab, unsigned n_rows_total, int iters, unsigned seed) {
    const unsigned n_rows = PRIV ? n_rows_total / 8 : n_rows_total;
    if (PRIV) tab += (size_t)(blockIdx.x & 7) * n_rows * 64;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 6 || MODE == 7) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned row = hash32(seed + wave * 977u + it * 4 + r) % n_rows;
                float *p = tab + (size_t)row * 64 + lane;
                if (MODE == 0) __hip_atomic_fetch_add(p, 1e-6f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (MODE == 4) __hip_atomic_fetch_add(p, 1e-6f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (MODE == 2) *p = *p + 1e-6f;
                if (MODE == 6) { float v = __builtin_nontemporal_load(p); __hip_atomic_store(p, v + 1e-6f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                if (MODE == 7) acc += __builtin_nontemporal_load(p);
            }
        } else if (MODE == 1 || MODE == 3) {
            const unsigned row = hash32(seed + wave * 977u + it * 4 + (lane >> 4)) % n_rows;
            float *p = tab + (size_t)row * 64 + 4 * (lane & 15);
            if (MODE == 1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) __hip_atomic_fetch_add(p + c, 1e-6f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                v4f v = *reinterpret_cast<v4f *>(p);
                v += 1e-6f;
                *reinterpret_cast<v4f *>(p) = v;
            }
        } else if (MODE == 5) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned row = hash32(seed + wave * 977u + it * 4 + r) % n_rows;
                double *p = reinterpret_cast<double *>(tab + (size_t)row * 64) + (lane & 31);
                if (lane < 32) __hip_atomic_fetch_add(p, 1e-6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (acc == 123.456f) tab[0] = acc;
}

template <int MODE, bool PRIV = false>
void run(float *tab, unsigned n_rows, const char *name) {
    const int blocks = 256 * 8, iters = 64;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE, PRIV><<<blocks, 256>>>(tab, n_rows, 4, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE, PRIV><<<blocks, 256>>>(tab, n_rows, iters, 7u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double rows = (double)blocks * 4 * iters * 4;  // waves * iters * 4 rows
    printf("%-34s rows=%8u  %8.3f ms  %7.2f G row-updates/s  %8.1f GB/s(256B rows)\n", name, n_rows, ms,
           rows / ms * 1e-6, rows * 256 / ms * 1e-6);
}

int main() {
    for (unsigned n_rows : {26744u, 138493u, 4000000u}) {
        float *tab; hipMalloc(&tab, (size_t)n_rows * 256); hipMemset(tab, 0, (size_t)n_rows * 256);
        run<0>(tab, n_rows, "atomic agent, lane=float");
        run<1>(tab, n_rows, "atomic agent, 16 lanes x float4");
        run<4>(tab, n_rows, "atomic workgroup, lane=float");
        run<5>(tab, n_rows, "atomic f64 agent, 32 lanes");
        run<2>(tab, n_rows, "plain RMW, lane=float");
        run<3>(tab, n_rows, "plain RMW, 16 lanes x float4");
        run<6>(tab, n_rows, "nt load + sc1 store, lane=float");
        run<7>(tab, n_rows, "nt load only, lane=float");
        hipFree(tab);
    }
    // XCD-private replicas: 8 x 26 744 rows (8 x 6.85 MB) and 8 x 8 192 rows (8 x 2 MB: fits each 4 MB L2)
    for (unsigned per : {26744u, 8192u}) {
        float *tab; hipMalloc(&tab, (size_t)per * 8 * 256); hipMemset(tab, 0, (size_t)per * 8 * 256);
        run<0, true>(tab, per * 8, "XCD-private: atomic agent");
        run<4, true>(tab, per * 8, "XCD-private: atomic workgroup");
        run<2, true>(tab, per * 8, "XCD-private: plain RMW");
        run<6, true>(tab, per * 8, "XCD-private: nt load + sc1 store");
        run<7, true>(tab, per * 8, "XCD-private: nt load only");
        hipFree(tab);
    }
    return 0;
}
