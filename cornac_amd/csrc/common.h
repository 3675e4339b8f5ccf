// Shared host/device helpers of libcornac_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cornac_hip.h"

namespace chip {

// ---- error plumbing: no exception crosses the C ABI ------------------------------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string &m);

[[noreturn]] inline void fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            ::chip::fail(CORNAC_HIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                                       \
    } while (0)

#define REQUIRE(cond, ...)                                         \
    do {                                                           \
        if (!(cond)) ::chip::fail(CORNAC_HIP_ERR_INVALID, __VA_ARGS__); \
    } while (0)

template <class F>
int guarded(F &&f) {
    try {
        f();
        return CORNAC_HIP_OK;
    } catch (const Error &e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return CORNAC_HIP_ERR_INVALID;
    } catch (...) {
        set_last_error("unknown error");
        return CORNAC_HIP_ERR_INVALID;
    }
}

void use_device(int device);  // validates it is a gfx950 part and makes it current

// ---- profiling switches: environment variables are read in -DCORNAC_PROFILE builds only (make PROFILE=1 ->
// libcornac_hip_profile.so); the shipped library always takes the default ------------------------------------
#ifdef CORNAC_PROFILE
inline int prof_env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
inline bool prof_env_set(const char *name) { return getenv(name) != nullptr; }
#else
constexpr int prof_env_int(const char *, int dflt) { return dflt; }
constexpr bool prof_env_set(const char *) { return false; }
#endif

// ---- device buffer -----------------------------------------------------------------------------
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool owned = true;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p && owned) (void)hipFree(p);
        p = nullptr;
        n = 0;
        owned = true;
    }
    void alloc(size_t count) {
        release();
        n = count;
        if (count) HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
    }
    void ensure(size_t count) {
        if (count > n || !p) alloc(count);
    }
    void bind(T *ext, size_t count) {
        release();
        p = ext;
        n = count;
        owned = false;
    }
    void upload(const T *src, size_t count, hipStream_t s) {
        HIP_CHECK(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void download(T *dst, size_t count, hipStream_t s) const {
        HIP_CHECK(hipMemcpyAsync(dst, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
    }
};

template <class T>
struct PinnedBuf {
    T *p = nullptr;
    size_t n = 0;
    ~PinnedBuf() {
        if (p) (void)hipHostFree(p);
    }
    void ensure(size_t count) {
        if (count <= n && p) return;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = count;
        HIP_CHECK(hipHostMalloc((void **)&p, count * sizeof(T), hipHostMallocDefault));
    }
};

struct Timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
};

// ---- HIP-event timing of the dominant kernel's launches (on the stream they are launched on) ------
struct EventTimer {
    bool enabled = false;
    std::vector<hipEvent_t> ev;  // pairs: [2i] before, [2i+1] after
    size_t used = 0;
    ~EventTimer() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    }
    void before(hipStream_t s) {
        if (!enabled) return;
        if (used + 2 > ev.size()) {
            hipEvent_t a, b;
            HIP_CHECK(hipEventCreate(&a));
            HIP_CHECK(hipEventCreate(&b));
            ev.push_back(a);
            ev.push_back(b);
        }
        HIP_CHECK(hipEventRecord(ev[used], s));
    }
    void after(hipStream_t s) {
        if (!enabled) return;
        HIP_CHECK(hipEventRecord(ev[used + 1], s));
        used += 2;
    }
    // sums the elapsed time of all recorded launches (stream must be idle) and resets
    void collect(double *total_ms, int64_t *launches) {
        double t = 0;
        for (size_t i = 0; i + 1 < used; i += 2) {
            HIP_CHECK(hipEventSynchronize(ev[i + 1]));
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            t += ms;
        }
        if (total_ms) *total_ms = t;
        if (launches) *launches = (int64_t)(used / 2);
        used = 0;
    }
};

// ---- conflict-free level scheduling of an ordered update stream (host, sched.cpp) ---------------
// sample s touches user row su[s] and item rows si[s], sj[s] (sj < 0: none; su < 0: sample skipped).
// level[s] = 1 + max(level of the previous sample touching any of its rows).  Samples of one level
// touch disjoint rows, so a level can run fully parallel while cross-level order reproduces the
// sequential result exactly.  Outputs the samples bucketed by level (stable within a level).
struct LevelSchedule {
    std::vector<int64_t> level_ptr;  // [n_levels + 1] offsets into the sorted arrays
    int64_t n_active = 0;
};
void build_level_schedule(const int32_t *su, const int32_t *si, const int32_t *sj, int64_t n, int64_t n_users,
                          int64_t n_items, int32_t *out_u, int32_t *out_i, int32_t *out_j, LevelSchedule &sched,
                          std::vector<int32_t> &scratch_lvl_u, std::vector<int32_t> &scratch_lvl_i,
                          std::vector<int32_t> &scratch_level);

// levels only (BPR deterministic mode buckets the triplets on the device): level[s] in [1, n_levels] or 0 for a
// skipped sample, sched.level_ptr as above
void build_levels(const int32_t *su, const int32_t *si, const int32_t *sj, int64_t n, int64_t n_users, int64_t n_items,
                  LevelSchedule &sched, std::vector<int32_t> &scratch_lvl_u, std::vector<int32_t> &scratch_lvl_i,
                  int32_t *level);

void build_level_schedule4(const int32_t *su, const int32_t *si, const int32_t *sv, const int32_t *sj, int64_t n,
                           int64_t n_users, int64_t n_items, int32_t *out_u, int32_t *out_i, int32_t *out_v,
                           int32_t *out_j, LevelSchedule &sched, std::vector<int32_t> &scratch_lvl_u,
                           std::vector<int32_t> &scratch_lvl_i, std::vector<int32_t> &scratch_level);

// per-device properties cached at first use
struct DeviceInfo {
    int cus = 256;
    int xcds = 8;
    std::string arch;
};
const DeviceInfo &device_info(int device);

}  // namespace chip
