// ttstamps.h -- diagnostic builds only (-DTT_STAMPS; tools/step_timeline.py): a log of (begin, end) wall-clock stamps (100 MHz, one
// clock for the chip) of EVERY workgroup of a kernel over many launches, in the order the workgroups finish.  One copy per
// translation unit (the library is built without relocatable device code); the host sorts and clusters the entries into launches.
#pragma once
#ifdef TT_STAMPS
#include <hip/hip_runtime.h>
constexpr int TT_LOG_CAP = 16384;
struct TTLog {
    unsigned long long n;
    unsigned long long e[TT_LOG_CAP][2];
};
__device__ __forceinline__ void tt_log_add(TTLog &L, const unsigned long long t0, const unsigned long long t1) {
    const unsigned long long i = atomicAdd(&L.n, 1ull);
    if (i < (unsigned long long)TT_LOG_CAP) { L.e[i][0] = t0; L.e[i][1] = t1; }
}
#endif
