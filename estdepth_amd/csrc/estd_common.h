// Shared helpers for the HIP translation units of libestd_hip.so (gfx950 only).
#ifndef ESTD_COMMON_H
#define ESTD_COMMON_H

#include <hip/hip_runtime.h>

#include "estd_hip.h"

#define ESTD_LAUNCH_CHECK() (hipGetLastError() == hipSuccess ? ESTD_OK : ESTD_ERR_LAUNCH)

static inline hipStream_t estd_stream(estd_stream_t s) { return static_cast<hipStream_t>(s); }

// Activation as a per-channel floor in the straight-line epilogues: out = fmaxf(v, floor) (one v_max_f32).  ReLU: floor 0.
// Activation "none": the floor is a quiet NaN -- v_max_f32 is IEEE maxNum, max(v, NaN) = v for every v and max(NaN, NaN) = NaN, so
// a NaN produced by the convolution still reaches the output as the reference's conv3d + BatchNorm would deliver it (with -inf as
// the floor a NaN came out as -inf).  Same instruction count.
#define ESTD_NO_FLOOR __builtin_nanf("")

// persistent-grid size of a kernel with `per_cu` resident workgroups per CU: the compute units of the CURRENT device (256 on an
// MI355X; queried once per device ordinal -- partitioned / other gfx950 parts report their own count) minus the reserve of
// estd_set_reserved_cus
extern "C" int estd_get_reserved_cus(void);
#include <atomic>
#include <stdio.h>
static inline int estd_device_cus(void)
{
    static std::atomic<int> cus[64];            // zero-initialised; 0 = not queried yet (relaxed: every thread would store the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int n = 0;
    const bool cached = (unsigned)dev < 64u;    // ordinals beyond the table are queried every time instead of aliasing a slot
    if (cached && (n = cus[dev].load(std::memory_order_relaxed)) > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
        static std::atomic<bool> warned{false};
        if (!warned.exchange(true))
            fprintf(stderr, "libestd_hip: hipDeviceGetAttribute(MultiprocessorCount) failed for device %d: persistent grids sized for 256 CUs\n", dev);
        return 256;                             // not cached: a later call may succeed
    }
    if (cached) cus[dev].store(n, std::memory_order_relaxed);
    return n;
}
static inline int estd_persistent_wgs(int per_cu)
{
    const int cus = estd_device_cus() - estd_get_reserved_cus();
    return (cus > 8 ? cus : 8) * per_cu;
}

// csrc/conv3d_wino2_c16.hip: the 16 -> 16 + head instance behind estd_conv3d_k3_wino2 (arguments validated by the caller)
int estd_wino2_c16_launch(const estd_conv3d_desc& d, hipStream_t stream);

static inline int estd_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (function, device): raise it once per device ordinal and
// kernel instantiation.  One atomic bit mask per instantiation -- thread-safe, and correct when a process drives
// several devices (launches under graph capture find the bit already set by the warm-up launch).
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
