// Wave-level primitives for the kernel bodies (one workgroup == one 64-lane wavefront).
// Device: DPP/permute based reductions.  Host simulation (NL == 1): identities.
#pragma once
#include "lhip_defs.h"

namespace lhip {

#ifdef LHIP_HOSTSIM
struct Wave { int lane; };
LHIP_DEV void wave_sync() {}
LHIP_DEV int wave_sum(int v) { return v; }
LHIP_DEV int wave_max(int v) { return v; }
LHIP_DEV int wave_min(int v) { return v; }
LHIP_DEV int wave_or(int v) { return v; }
LHIP_DEV uint64_t wave_or64(uint64_t v) { return v; }
LHIP_DEV float wave_maxf(float v) { return v; }
LHIP_DEV double wave_maxd(double v) { return v; }
LHIP_DEV double wave_sumd(double v) { return v; }
LHIP_DEV int wave_bcast(int v, int) { return v; }
LHIP_DEV int wave_any(int p) { return p != 0; }
LHIP_DEV int wave_excl_scan(int v, int lane, int* total) { (void)lane; *total = v; return 0; }
LHIP_DEV void lds_or(uint32_t* p, uint32_t v) { *p |= v; }
#else
struct Wave { int lane; };
// Orders this wave's LDS/global traffic around a role change between lanes.  One wavefront
// executes in lock-step, so no s_barrier is needed; the fences stop the compiler from
// moving memory operations across the hand-off.
LHIP_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
LHIP_DEV int wave_sum(int v) { return __reduce_add_sync(~0ull, v); }
LHIP_DEV int wave_max(int v) { return __reduce_max_sync(~0ull, v); }
LHIP_DEV int wave_min(int v) { return __reduce_min_sync(~0ull, v); }
LHIP_DEV int wave_or(int v) { return (int)__reduce_or_sync(~0ull, (unsigned)v); }
LHIP_DEV uint64_t wave_or64(uint64_t v) {
    const unsigned lo = __reduce_or_sync(~0ull, (unsigned)v), hi = __reduce_or_sync(~0ull, (unsigned)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
LHIP_DEV float wave_maxf(float v) {
    // max is exact and associative for non-NaN operands: any reduction order gives the same bits
    for (int o = 32; o > 0; o >>= 1) { float t = __shfl_xor(v, o); v = (t > v) ? t : v; }
    return v;
}
LHIP_DEV double wave_maxd(double v) {
    for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o); v = (t > v) ? t : v; }
    return v;
}
// tree sum: NOT order-exact; only for order-insensitive decisions
LHIP_DEV double wave_sumd(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
LHIP_DEV int wave_bcast(int v, int src) { return __shfl(v, src); }
LHIP_DEV int wave_any(int p) { return __any(p); }
// exclusive prefix sum over the 64 lanes (integers: exact in any order); *total = sum over all lanes
LHIP_DEV int wave_excl_scan(int v, int lane, int* total) {
    int x = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(x, o); if (lane >= o) x += t; }
    *total = __shfl(x, 63);
    return x - v;
}
LHIP_DEV void lds_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
#endif

}  // namespace lhip
