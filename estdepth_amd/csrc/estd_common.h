// Shared helpers for the HIP translation units of libestd_hip.so (gfx950 only).
#ifndef ESTD_COMMON_H
#define ESTD_COMMON_H

#include <hip/hip_runtime.h>

#include "estd_hip.h"

#define ESTD_LAUNCH_CHECK() (hipGetLastError() == hipSuccess ? ESTD_OK : ESTD_ERR_LAUNCH)

static inline hipStream_t estd_stream(estd_stream_t s) { return static_cast<hipStream_t>(s); }

static inline int estd_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

#endif
