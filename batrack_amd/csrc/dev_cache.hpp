// dev_cache.hpp — what the launchers remember per DEVICE: the device's properties and, per kernel, the dynamic-LDS limit that
// hipFuncSetAttribute has raised (the attribute is per device: a process-wide static would leave a second GPU of the process at
// its 64 KB default and hand it the first device's CU count).  Thread-safe: atomics, values only ever raised.
#pragma once
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstddef>

namespace bt {

constexpr int kMaxDevices = 64;

struct DevProps { int dev, n_cu; size_t lds_cu; };

// properties of device `dev` (the plan's: PlanDev::dev_id — no runtime call on the launch path; -1: the current device), read once per device
inline bool device_props(DevProps *out, int dev = -1) {
    static std::atomic<int> n_cu[kMaxDevices];
    static std::atomic<size_t> lds_cu[kMaxDevices];
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return false;
    const int slot = dev >= 0 && dev < kMaxDevices ? dev : -1;
    int n = slot >= 0 ? n_cu[slot].load(std::memory_order_acquire) : 0;
    size_t l = slot >= 0 ? lds_cu[slot].load(std::memory_order_relaxed) : 0;
    if (!n) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
        n = prop.multiProcessorCount;
        l = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 160 * 1024;
        if (slot >= 0) { lds_cu[slot].store(l, std::memory_order_relaxed); n_cu[slot].store(n, std::memory_order_release); }
    }
    out->dev = dev; out->n_cu = n; out->lds_cu = l;
    return true;
}

// one per kernel instantiation (a function-local static of its launcher)
struct LdsLimit {
    std::atomic<size_t> raised[kMaxDevices];
    // make `bytes` of dynamic LDS launchable for `func` on device `dev` (-1: the current device; the attribute is set on the current one)
    bool ensure(const void *func, size_t bytes, int dev = -1) {
        if (bytes <= 48 * 1024) return true;
        if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return false;
        const int slot = dev >= 0 && dev < kMaxDevices ? dev : -1;
        if (slot >= 0 && raised[slot].load(std::memory_order_acquire) >= bytes) return true;
        if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
        if (slot >= 0) {
            size_t cur = raised[slot].load(std::memory_order_relaxed);
            while (cur < bytes && !raised[slot].compare_exchange_weak(cur, bytes, std::memory_order_release)) { }
        }
        return true;
    }
};

}  // namespace bt
