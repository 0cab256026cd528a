// common.h — error plumbing shared by the C-ABI translation units of libmi355clip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "../../include/mi355clip.h"

namespace mi {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& m);

[[noreturn]] inline void fail(int code, const char* fmt, ...) {
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

inline void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess) {
        int code = (e == hipErrorOutOfMemory) ? MI_ERR_OOM : MI_ERR_HIP;
        fail(code, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    }
}
#define HIP_CHECK(x) ::mi::hip_check((x), #x, __FILE__, __LINE__)

// Every extern "C" body runs inside this: nothing may unwind across the ABI.
template <class F>
int guarded(F&& f) noexcept {
    try {
        f();
        return MI_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("host allocation failed");
        return MI_ERR_OOM;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return MI_ERR_INVALID;
    } catch (...) {
        set_last_error("unknown error");
        return MI_ERR_INVALID;
    }
}

// Select `device` and require it to be a gfx950 part: there is no fallback path.
void use_device(int device);

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int device) {
        (void)hipGetDevice(&prev);
        use_device(device);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per (device, function): one DevOnce per call site
// remembers which devices already have it (a second handle on another GPU of the same process, or two
// threads loading handles at once, must not skip it; setting it twice is harmless).
struct DevOnce {
    std::atomic<uint64_t> mask{0};
};
template <class K>
inline void allow_lds_once(DevOnce& once, K kernel, int bytes) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (once.mask.load(std::memory_order_acquire) & bit) return;
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    once.mask.fetch_or(bit, std::memory_order_release);
}

// Order of the work of ONE handle across streams.  Every entry point that enqueues work which touches
// the handle's buffers calls begin(s) first (s waits for whatever the handle enqueued last, on any
// stream) and end(s) last.  Work on a handle therefore runs in call order whichever streams the
// callers pass; the host never blocks.  sync() is for the few places that free or move buffers.
struct WorkOrder {
    hipEvent_t ev = nullptr;
    bool pending = false;
    void begin(hipStream_t s) {
        if (pending) HIP_CHECK(hipStreamWaitEvent(s, ev, 0));
    }
    void end(hipStream_t s) {
        if (!ev) HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(ev, s));
        pending = true;
    }
    void sync() {
        if (pending) HIP_CHECK(hipEventSynchronize(ev));
        pending = false;
    }
    void destroy() {
        if (ev) (void)hipEventDestroy(ev);
        ev = nullptr;
        pending = false;
    }
};

}  // namespace mi
