// Runtime plumbing of libcornac_hip: error state, device selection, level scheduler.
#include <algorithm>
#include <mutex>

#include "common.h"

namespace chip {

static thread_local std::string g_last_error;
void set_last_error(const std::string &m) { g_last_error = m; }

static std::mutex g_info_mu;
static std::vector<DeviceInfo> g_info;
static std::vector<char> g_info_ok;

const DeviceInfo &device_info(int device) {
    std::lock_guard<std::mutex> lk(g_info_mu);
    if ((int)g_info.size() <= device) {
        g_info.resize(device + 1);
        g_info_ok.resize(device + 1, 0);
    }
    if (!g_info_ok[device]) {
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        g_info[device].cus = prop.multiProcessorCount;
        g_info[device].arch = prop.gcnArchName;
        g_info[device].xcds = 8;
        g_info_ok[device] = 1;
    }
    return g_info[device];
}

void use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        fail(CORNAC_HIP_ERR_NO_DEVICE, "no HIP device visible (%s); libcornac_hip has no CPU fallback",
             e == hipSuccess ? "count=0" : hipGetErrorString(e));
    REQUIRE(device >= 0 && device < n, "device %d out of range (found %d)", device, n);
    const DeviceInfo &di = device_info(device);
    if (di.arch.rfind("gfx950", 0) != 0)
        fail(CORNAC_HIP_ERR_UNSUPPORTED, "device %d is %s; this library carries gfx950 (MI355X) code objects only",
             device, di.arch.c_str());
    HIP_CHECK(hipSetDevice(device));
}

// Conflict-free level schedule (see common.h).  O(n) integer work on the host; the ordering
// constraint is inherently sequential (longest path in the row-conflict DAG in sample order).
void build_level_schedule(const int32_t *su, const int32_t *si, const int32_t *sj, int64_t n, int64_t n_users,
                          int64_t n_items, int32_t *out_u, int32_t *out_i, int32_t *out_j, LevelSchedule &sched,
                          std::vector<int32_t> &lvl_u, std::vector<int32_t> &lvl_i, std::vector<int32_t> &level) {
    lvl_u.assign((size_t)n_users, 0);
    lvl_i.assign((size_t)n_items, 0);
    level.resize((size_t)n);
    int32_t max_level = 0;
    for (int64_t s = 0; s < n; ++s) {
        const int32_t u = su[s];
        if (u < 0) {
            level[s] = 0;
            continue;
        }
        const int32_t i = si[s], j = sj[s];
        int32_t l = std::max(lvl_u[u], lvl_i[i]);
        if (j >= 0) l = std::max(l, lvl_i[j]);
        ++l;
        lvl_u[u] = l;
        lvl_i[i] = l;
        if (j >= 0) lvl_i[j] = l;
        level[s] = l;
        max_level = std::max(max_level, l);
    }
    sched.level_ptr.assign((size_t)max_level + 2, 0);
    for (int64_t s = 0; s < n; ++s)
        if (level[s] > 0) ++sched.level_ptr[(size_t)level[s] + 1];
    // level_ptr[l+1] currently holds count of level l (levels are 1-based); prefix-sum so that
    // level l occupies [level_ptr[l], level_ptr[l+1]).  Index 0/1 stay 0 (there is no level 0).
    for (size_t l = 1; l < sched.level_ptr.size(); ++l) sched.level_ptr[l] += sched.level_ptr[l - 1];
    std::vector<int64_t> cursor(sched.level_ptr.begin(), sched.level_ptr.end());
    for (int64_t s = 0; s < n; ++s) {
        const int32_t l = level[s];
        if (l == 0) continue;
        const int64_t pos = cursor[(size_t)l]++;
        out_u[pos] = su[s];
        out_i[pos] = si[s];
        out_j[pos] = sj[s];
    }
    sched.n_active = sched.level_ptr.back();
}

void build_levels(const int32_t *su, const int32_t *si, const int32_t *sj, int64_t n, int64_t n_users, int64_t n_items,
                  LevelSchedule &sched, std::vector<int32_t> &lvl_u, std::vector<int32_t> &lvl_i, int32_t *level) {
    lvl_u.assign((size_t)n_users, 0);
    lvl_i.assign((size_t)n_items, 0);
    int32_t max_level = 0;
    for (int64_t s = 0; s < n; ++s) {
        const int32_t u = su[s];
        if (u < 0) {
            level[s] = 0;
            continue;
        }
        const int32_t i = si[s], j = sj[s];
        int32_t l = std::max(lvl_u[u], lvl_i[i]);
        if (j >= 0) l = std::max(l, lvl_i[j]);
        ++l;
        lvl_u[u] = l;
        lvl_i[i] = l;
        if (j >= 0) lvl_i[j] = l;
        level[s] = l;
        max_level = std::max(max_level, l);
    }
    sched.level_ptr.assign((size_t)max_level + 2, 0);
    for (int64_t s = 0; s < n; ++s)
        if (level[s] > 0) ++sched.level_ptr[(size_t)level[s] + 1];
    for (size_t l = 1; l < sched.level_ptr.size(); ++l) sched.level_ptr[l] += sched.level_ptr[l - 1];
    sched.n_active = sched.level_ptr.back();
}

// 4-row variant (VEBPR: user, purchased item, viewed item or -1, negative item)
void build_level_schedule4(const int32_t *su, const int32_t *si, const int32_t *sv, const int32_t *sj, int64_t n,
                           int64_t n_users, int64_t n_items, int32_t *out_u, int32_t *out_i, int32_t *out_v,
                           int32_t *out_j, LevelSchedule &sched, std::vector<int32_t> &lvl_u,
                           std::vector<int32_t> &lvl_i, std::vector<int32_t> &level) {
    lvl_u.assign((size_t)n_users, 0);
    lvl_i.assign((size_t)n_items, 0);
    level.resize((size_t)n);
    int32_t max_level = 0;
    for (int64_t s = 0; s < n; ++s) {
        const int32_t u = su[s];
        if (u < 0) {
            level[s] = 0;
            continue;
        }
        const int32_t i = si[s], v = sv[s], j = sj[s];
        int32_t l = std::max(std::max(lvl_u[u], lvl_i[i]), lvl_i[j]);
        if (v >= 0) l = std::max(l, lvl_i[v]);
        ++l;
        lvl_u[u] = l;
        lvl_i[i] = l;
        lvl_i[j] = l;
        if (v >= 0) lvl_i[v] = l;
        level[s] = l;
        max_level = std::max(max_level, l);
    }
    sched.level_ptr.assign((size_t)max_level + 2, 0);
    for (int64_t s = 0; s < n; ++s)
        if (level[s] > 0) ++sched.level_ptr[(size_t)level[s] + 1];
    for (size_t l = 1; l < sched.level_ptr.size(); ++l) sched.level_ptr[l] += sched.level_ptr[l - 1];
    std::vector<int64_t> cursor(sched.level_ptr.begin(), sched.level_ptr.end());
    for (int64_t s = 0; s < n; ++s) {
        const int32_t l = level[s];
        if (l == 0) continue;
        const int64_t pos = cursor[(size_t)l]++;
        out_u[pos] = su[s];
        out_i[pos] = si[s];
        out_v[pos] = sv[s];
        out_j[pos] = sj[s];
    }
    sched.n_active = sched.level_ptr.back();
}

}  // namespace chip

extern "C" {

const char *cornac_hip_last_error(void) { return chip::g_last_error.c_str(); }

const char *cornac_hip_version(void) { return "cornac_hip 0.1.0 (gfx950)"; }

int cornac_hip_device_count(int *count) {
    return chip::guarded([&] {
        REQUIRE(count != nullptr, "count is NULL");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        *count = (e == hipSuccess) ? n : 0;
    });
}

namespace chip {
// random 512-byte row gathers over a table (the access pattern of the k = 128 SGD kernels): every wave reads whole rows
// chosen by a hash of (row counter, seed) and folds them into a checksum so the loads are not optimised away
__global__ __launch_bounds__(256) void probe_gather_kernel(const float *__restrict__ table, int64_t n_rows, int64_t n_gathers,
                                                           float *__restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * 256) >> 6;
    float acc = 0.f;
    for (int64_t g = wave; g < n_gathers; g += n_waves) {
        uint64_t h = (uint64_t)g * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const int64_t r = (int64_t)(h % (uint64_t)n_rows);
        acc += __builtin_nontemporal_load(table + r * 128 + lane) + __builtin_nontemporal_load(table + r * 128 + 64 + lane);
    }
    if (acc == 12345.678f) sink[0] = acc;
}
typedef float probe_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void probe_stream_kernel(const probe_v4f *__restrict__ src, int64_t n4, float *__restrict__ sink) {
    float acc = 0.f;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n4; t += (int64_t)gridDim.x * 256) {
        const probe_v4f v = __builtin_nontemporal_load(src + t);
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
}  // namespace chip

int cornac_hip_device_probe(int device, int64_t bytes, double *out3) {
    return chip::guarded([&] {
        REQUIRE(out3 != nullptr && bytes >= (int64_t(1) << 20), "bad arguments");
        chip::use_device(device);
        chip::DevBuf<float> a, b;
        const int64_t n = bytes / 4 / 128 * 128;
        a.alloc((size_t)n);
        b.alloc((size_t)n);
        HIP_CHECK(hipMemset(a.p, 0, (size_t)n * 4));
        HIP_CHECK(hipMemset(b.p, 0, (size_t)n * 4));
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        auto timed = [&](auto &&f) {
            f();  // warm-up
            HIP_CHECK(hipEventRecord(e0, nullptr));
            f();
            HIP_CHECK(hipEventRecord(e1, nullptr));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            return (double)ms * 1e-3;
        };
        const int grid = chip::device_info(device).cus * 8;
        const double t_copy = timed([&] { HIP_CHECK(hipMemcpyAsync(b.p, a.p, (size_t)n * 4, hipMemcpyDeviceToDevice, nullptr)); });
        const double t_stream = timed([&] {
            hipLaunchKernelGGL(chip::probe_stream_kernel, dim3(grid), dim3(256), 0, nullptr, (const chip::probe_v4f *)a.p, n / 4, b.p);
        });
        const int64_t n_rows = n / 128, n_gathers = int64_t(1) << 25;
        const double t_gather = timed([&] {
            hipLaunchKernelGGL(chip::probe_gather_kernel, dim3(grid), dim3(256), 0, nullptr, a.p, n_rows, n_gathers, b.p);
        });
        HIP_CHECK(hipGetLastError());
        out3[0] = 2.0 * (double)n * 4 / t_copy / 1e9;            // device-to-device copy, bytes read + written
        out3[1] = (double)n * 4 / t_stream / 1e9;                // streaming read
        out3[2] = (double)n_gathers * 512 / t_gather / 1e9;      // random 512-byte row gathers
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    });
}

int cornac_hip_device_info(int device, char *name, int name_len, int *compute_units, int64_t *hbm_bytes) {
    return chip::guarded([&] {
        int n = 0;
        HIP_CHECK(hipGetDeviceCount(&n));
        REQUIRE(device >= 0 && device < n, "device %d out of range (found %d)", device, n);
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        if (name && name_len > 0) {
            snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
        }
        if (compute_units) *compute_units = prop.multiProcessorCount;
        if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    });
}
}
