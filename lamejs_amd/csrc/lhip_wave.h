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
LHIP_DEV float wave_maxf_pos(float v) { return v; }
LHIP_DEV double wave_maxd(double v) { return v; }
LHIP_DEV double wave_sumd(double v) { return v; }
LHIP_DEV int wave_bcast(int v, int) { return v; }
LHIP_DEV int wave_any(int p) { return p != 0; }
LHIP_DEV int wave_excl_scan(int v, int lane, int* total) { (void)lane; *total = v; return 0; }
LHIP_DEV void lds_or(uint32_t* p, uint32_t v) { *p |= v; }
LHIP_DEV void lds_max(int32_t* p, int32_t v) { if (*p < v) *p = v; }
// see the device version
LHIP_DEV uint64_t wave_lane_bits(uint64_t v) { return v; }
LHIP_DEV int uni(int v) { return v; }
LHIP_DEV int fresh_lane(int lane) { return lane; }
LHIP_DEV double unid(double v) { return v; }
LHIP_DEV double wave_shr1d(double v, double first) { (void)v; return first; }
// strictly sequential (index order) f64 sum of LHIP_NL * K values, lane l holding elements [l*K, (l+1)*K)
template <int K> LHIP_DEV double wave_seq_sum(const double (&p)[K]) { double s = 0.0; for (int k = 0; k < K; k++) s += p[k]; return s; }
#else
extern "C" __device__ float __ockl_wfred_max_f32(float);
extern "C" __device__ double __ockl_wfred_max_f64(double);
extern "C" __device__ double __ockl_wfred_add_f64(double);
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
// Float reductions go through the device library's DPP wavefront reductions: the result lands in an SGPR,
// so everything computed from it (branch conditions, loop bounds) is known to be wave-uniform by the
// compiler and runs on the scalar unit.  (A __shfl butterfly leaves a per-lane copy that the compiler must
// treat as divergent -- it turns the whole control flow downstream into exec-masked code.)
// max is exact and associative for non-NaN operands: any reduction order gives the same bits.
LHIP_DEV float wave_maxf(float v) { return __ockl_wfred_max_f32(v); }
// maximum of values that are all >= +0 (no NaN, no -0): IEEE order == integer order of the bit patterns, and the integer
// reduction is one fused DPP max per step where the f32 one spends three more on canonicalising its operands
LHIP_DEV float wave_maxf_pos(float v) { return __int_as_float(__reduce_max_sync(~0ull, __float_as_int(v))); }
LHIP_DEV double wave_maxd(double v) { return __ockl_wfred_max_f64(v); }
// tree sum: NOT order-exact; only for order-insensitive decisions
LHIP_DEV double wave_sumd(double v) { return __ockl_wfred_add_f64(v); }
LHIP_DEV int wave_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }   // src must be wave-uniform
LHIP_DEV int wave_any(int p) { return __any(p); }
// exclusive prefix sum over the 64 lanes (integers: exact in any order); *total = sum over all lanes.
// Kogge-Stone inside each row of 16 lanes with DPP row shifts, then the row totals are broadcast into the later rows
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3): six DPP adds, no LDS round trips.
LHIP_DEV int wave_excl_scan(int v, int lane, int* total) {
    (void)lane;
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    *total = __builtin_amdgcn_readlane(x, 63);
    return x - v;
}
LHIP_DEV void lds_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
LHIP_DEV void lds_max(int32_t* p, int32_t v) { atomicMax(p, v); }
// OR over the lanes of masks in which lane l can only have bit l set (the `for (i = lane; i < n; i += 64)` idiom
// with n <= 64): that is a ballot -- one compare, result in SGPRs, no reduction chain.
LHIP_DEV uint64_t wave_lane_bits(uint64_t v) { return __ballot(v != 0); }
// An opaque copy of the lane index: addresses derived from it cannot be merged with (and hoisted like) the ones
// derived from other copies, which keeps loop-invariant address registers from piling up and spilling.
LHIP_DEV int fresh_lane(int lane) { asm volatile("" : "+v"(lane)); __builtin_assume(lane >= 0 && lane < LHIP_NL); return lane; }   // the copy keeps its range
// asserts to the compiler that v is wave-uniform (moves it to an SGPR)
LHIP_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// value of lane - 1 (lane 0 receives `first`): two DPP moves (wave_shr:1), no LDS round trip
LHIP_DEV double wave_shr1d(double v, double first) {
    union { double d; int i[2]; } a, f, r; a.d = v; f.d = first;
    r.i[0] = __builtin_amdgcn_update_dpp(f.i[0], a.i[0], 0x138, 0xf, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(f.i[1], a.i[1], 0x138, 0xf, 0xf, false);
    return r.d;
}
// Strictly sequential (index order) f64 sum of 64 * K values, lane l holding elements [l*K, (l+1)*K): a systolic
// fold -- every step each lane adds its K values, in order, onto the sum handed over by lane l-1; after step t
// lanes 0..t hold exact prefixes, so 64 steps give the exact left-to-right total in lane 63.  Same additions in
// the same order as a scalar loop, but without funnelling the operands through one lane.
template <int K> LHIP_DEV double wave_seq_sum(const double (&p)[K]) {
    double carry = 0.0, s = 0.0;
    for (int st = 0; st < 64; st++) {
        s = carry;
#pragma unroll
        for (int k = 0; k < K; k++) s += p[k];
        carry = wave_shr1d(s, 0.0);
    }
    union { double d; int i[2]; } u; u.d = s;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], 63); u.i[1] = __builtin_amdgcn_readlane(u.i[1], 63);
    return u.d;
}
LHIP_DEV double unid(double v) { union { double d; int i[2]; } u; u.d = v; u.i[0] = uni(u.i[0]); u.i[1] = uni(u.i[1]); return u.d; }
#endif

}  // namespace lhip
