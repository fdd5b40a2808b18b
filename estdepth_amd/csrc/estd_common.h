// Shared helpers for the HIP translation units of libestd_hip.so (gfx950 only).
#ifndef ESTD_COMMON_H
#define ESTD_COMMON_H

#include <hip/hip_runtime.h>

#include "estd_hip.h"

#define ESTD_LAUNCH_CHECK() (hipGetLastError() == hipSuccess ? ESTD_OK : ESTD_ERR_LAUNCH)

static inline hipStream_t estd_stream(estd_stream_t s) { return static_cast<hipStream_t>(s); }

// persistent-grid size of a kernel with `per_cu` resident workgroups per CU (256 CUs, minus the reserve of estd_set_reserved_cus)
extern "C" int estd_get_reserved_cus(void);
static inline int estd_persistent_wgs(int per_cu) { return (256 - estd_get_reserved_cus()) * per_cu; }

static inline int estd_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (function, device): raise it once per device ordinal and
// kernel instantiation.  One atomic bit mask per instantiation -- thread-safe, and correct when a process drives
// several devices (launches under graph capture find the bit already set by the warm-up launch).
#include <atomic>
template <auto Kernel>
static inline void estd_allow_dynamic_lds(int bytes)
{
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        done.fetch_or(bit, std::memory_order_release);
    }
}

#endif
