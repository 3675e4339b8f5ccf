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
