// Micro-benchmark (not part of the product): random 512-byte row gathers / read-modify-writes over a table far larger than
// the caches — the access pattern of the k = 128 SGD kernels at the configs[4] shape — as a function of the rows a wave
// keeps in flight, the width of a lane's access and the waves per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe && tools/gather_probe [table_MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// WIDTH 1: a row = 2 instructions of 64 lanes x 4 B; 2: 1 instruction of 64 x 8 B; 4: 32 lanes x 16 B (a wave instruction = 2 rows)
// INF: rows per wave in flight per step.  RMW: load, add, plain store.
template <int WIDTH, int INF, bool RMW>
__global__ __launch_bounds__(256) void probe(float *tab, unsigned n_rows, int iters, unsigned seed) {
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (WIDTH == 1) {
            float v[INF][2]; float *p[INF];
#pragma unroll
            for (int r = 0; r < INF; ++r) {
                const unsigned row = hash32(seed + wave * 977u + it * INF + r) % n_rows;
                p[r] = tab + (size_t)row * 128 + lane;
                v[r][0] = __builtin_nontemporal_load(p[r]); v[r][1] = __builtin_nontemporal_load(p[r] + 64);
            }
#pragma unroll
            for (int r = 0; r < INF; ++r) {
                if (RMW) { p[r][0] = v[r][0] + 1e-6f; p[r][64] = v[r][1] + 1e-6f; } else acc += v[r][0] + v[r][1];
            }
        } else if (WIDTH == 2) {
            v2f v[INF]; v2f *p[INF];
#pragma unroll
            for (int r = 0; r < INF; ++r) {
                const unsigned row = hash32(seed + wave * 977u + it * INF + r) % n_rows;
                p[r] = reinterpret_cast<v2f *>(tab + (size_t)row * 128) + lane;
                v[r] = __builtin_nontemporal_load(p[r]);
            }
#pragma unroll
            for (int r = 0; r < INF; ++r) {
                if (RMW) *p[r] = v[r] + 1e-6f; else acc += v[r].x + v[r].y;
            }
        } else {
            v4f v[INF]; v4f *p[INF];  // INF wave-instructions = 2 INF rows
#pragma unroll
            for (int r = 0; r < INF; ++r) {
                const unsigned row = hash32(seed + wave * 977u + (it * INF + r) * 2 + (lane >> 5)) % n_rows;
                p[r] = reinterpret_cast<v4f *>(tab + (size_t)row * 128) + (lane & 31);
                v[r] = __builtin_nontemporal_load(p[r]);
            }
#pragma unroll
            for (int r = 0; r < INF; ++r) {
                if (RMW) *p[r] = v[r] + 1e-6f; else acc += v[r].x + v[r].y + v[r].z + v[r].w;
            }
        }
    }
    if (acc == 123.456f) tab[0] = acc;
}

// the strata kernel's item-row access with its bias: a 512-byte row read-modify-written together with EXTRA more bytes that
// are either contiguous with the row (row stride 512 + EXTRA) or a separate random line of a second table (SEP)
template <int EXTRA, bool SEP>
__global__ __launch_bounds__(256) void probe_bias(float *tab, float *side, unsigned n_rows, int iters, unsigned seed) {
    constexpr int INF = 6, STRIDE = SEP ? 128 : 128 + EXTRA / 4;
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        float v[INF][2], e[INF]; float *p[INF], *q[INF];
#pragma unroll
        for (int r = 0; r < INF; ++r) {
            const unsigned row = hash32(seed + wave * 977u + it * INF + r) % n_rows;
            p[r] = tab + (size_t)row * STRIDE + lane;
            q[r] = SEP ? side + (size_t)row * (EXTRA / 4) : tab + (size_t)row * STRIDE + 128;
            v[r][0] = __builtin_nontemporal_load(p[r]); v[r][1] = __builtin_nontemporal_load(p[r] + 64);
            e[r] = __hip_atomic_load(q[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // one lane's worth is needed
        }
#pragma unroll
        for (int r = 0; r < INF; ++r) {
            p[r][0] = v[r][0] + 1e-6f; p[r][64] = v[r][1] + 1e-6f;
            if (lane < EXTRA / 4) q[r][lane] = lane == 0 ? e[r] + 1e-6f : 0.f;  // the whole extra line
        }
    }
}

template <int EXTRA, bool SEP>
static void run_bias(float *tab, float *side, unsigned n_rows, const char *name) {
    const int grid = 256 * 6;
    const int iters = (int)((1ll << 26) / ((long long)grid * 4 * 6));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe_bias<EXTRA, SEP>), dim3(grid), dim3(256), 0, 0, tab, side, n_rows, iters / 8 + 1, 1u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe_bias<EXTRA, SEP>), dim3(grid), dim3(256), 0, 0, tab, side, n_rows, iters, 7u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double rows = (double)grid * 4 * 6 * iters;
    printf("%-44s %6.2f G rows/s = %5.2f TB/s of 512-byte row payload (read + written)\n", name, rows / ms / 1e6, rows * 1024 / ms / 1e9);
    fflush(stdout);
}

template <int WIDTH, int INF, bool RMW>
static void run(float *tab, unsigned n_rows, int blocks_per_cu, const char *name) {
    const int grid = 256 * blocks_per_cu;
    const long long rows_target = 1ll << 26;  // 32 GiB of row traffic per direction
    const int rows_per_it = (WIDTH == 4 ? 2 : 1) * INF;
    const int iters = (int)(rows_target / ((long long)grid * 4 * rows_per_it));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<WIDTH, INF, RMW>), dim3(grid), dim3(256), 0, 0, tab, n_rows, iters / 8 + 1, 1u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<WIDTH, INF, RMW>), dim3(grid), dim3(256), 0, 0, tab, n_rows, iters, 7u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double rows = (double)grid * 4 * rows_per_it * iters;
    printf("%-28s waves/CU %2d: %6.2f G rows/s = %5.2f TB/s %s\n", name, blocks_per_cu * 4, rows / ms / 1e6,
           rows * 512 * (RMW ? 2 : 1) / ms / 1e9, RMW ? "(read + written)" : "(read)");
    fflush(stdout);
}

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? (size_t)atoll(argv[1]) : 6144;
    const unsigned n_rows = (unsigned)(mib * 2048);
    float *tab;
    if (hipMalloc(&tab, (size_t)n_rows * 512) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(tab, 0, (size_t)n_rows * 512);
    printf("table %zu MiB (%u rows of 512 B)\n", mib, n_rows);
    if (argc > 2) {  // bias placement study: rows of the main table sized for a 640-byte stride
        const unsigned nr = n_rows * 4 / 5;
        float *side;
        if (hipMalloc(&side, (size_t)nr * 128) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemset(side, 0, (size_t)nr * 128);
        run<1, 6, true>(tab, nr, 6, "rmw   dword    6 rows/step");
        run_bias<128, true>(tab, side, nr, "row 512 + separate random 128-byte line");
        run_bias<64, true>(tab, side, nr, "row 512 + separate random 64-byte line");
        run_bias<128, false>(tab, side, nr, "row 640 contiguous (128 extra)");
        run_bias<64, false>(tab, side, nr, "row 576 contiguous (64 extra)");
        return 0;
    }
    for (int b : {2, 4, 8}) {
        run<1, 2, false>(tab, n_rows, b, "load  dword    2 rows/step");
        run<1, 6, false>(tab, n_rows, b, "load  dword    6 rows/step");
        run<2, 6, false>(tab, n_rows, b, "load  dwordx2  6 rows/step");
        run<2, 12, false>(tab, n_rows, b, "load  dwordx2 12 rows/step");
        run<4, 6, false>(tab, n_rows, b, "load  dwordx4 12 rows/step");
        run<1, 6, true>(tab, n_rows, b, "rmw   dword    6 rows/step");
        run<2, 6, true>(tab, n_rows, b, "rmw   dwordx2  6 rows/step");
        run<2, 12, true>(tab, n_rows, b, "rmw   dwordx2 12 rows/step");
        run<4, 6, true>(tab, n_rows, b, "rmw   dwordx4 12 rows/step");
    }
    return 0;
}
