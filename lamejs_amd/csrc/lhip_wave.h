// Wave-level primitives for the kernel bodies (one workgroup == one 64-lane wavefront).
// Device: DPP/permute based reductions.  Host simulation (NL == 1): identities.
#pragma once
#include "lhip_defs.h"

namespace lhip {

#if defined(LHIP_HOSTSIM) && LHIP_NL == 1
struct Wave { int lane; };
LHIP_DEV void wave_sync() {}
LHIP_DEV void wave_sync_global() {}
LHIP_DEV int wave_sum(int v) { return v; }
template <int N> LHIP_DEV void wave_sum_n(int (&v)[N]) { (void)v; }
template <int N> LHIP_DEV void wave_max_n(int (&v)[N]) { (void)v; }
LHIP_DEV int wave_max(int v) { return v; }
LHIP_DEV int wave_min(int v) { return v; }
LHIP_DEV int wave_or(int v) { return v; }
LHIP_DEV uint64_t wave_or64(uint64_t v) { return v; }
LHIP_DEV float wave_maxf(float v) { return v; }
LHIP_DEV float wave_maxf_pos(float v) { return v; }
LHIP_DEV double wave_maxd(double v) { return v; }
LHIP_DEV double wave_sumd(double v) { return v; }
LHIP_DEV int wave_bcast(int v, int) { return v; }
LHIP_DEV uint64_t wave_ballot(int p) { return p ? 1ull : 0ull; }
LHIP_DEV unsigned mul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
LHIP_DEV int wave_any(int p) { return p != 0; }
LHIP_DEV int wave_excl_scan(int v, int lane, int* total) { (void)lane; *total = v; return 0; }
LHIP_DEV void lds_or(uint32_t* p, uint32_t v) { *p |= v; }
LHIP_DEV void lds_max(int32_t* p, int32_t v) { if (*p < v) *p = v; }
// see the device version
LHIP_DEV uint64_t wave_lane_bits(uint64_t v) { return v; }
LHIP_DEV void wg_barrier() {}                  // multi-wave workgroups only exist on the device and in the wave simulation
LHIP_DEV int uni(int v) { return v; }
LHIP_DEV int fresh_lane(int lane) { return lane; }
LHIP_DEV int lane_anew(int lane) { return lane; }
LHIP_DEV double const_here(double c) { return c; }
LHIP_DEV int uni_here(int v) { return v; }
LHIP_DEV double unid(double v) { return v; }
LHIP_DEV double wave_shr1d(double v, double first) { (void)v; return first; }
// strictly sequential (index order) f64 sum of LHIP_NL * K values, lane l holding elements [l*K, (l+1)*K)
template <int K> LHIP_DEV double wave_seq_sum(const double (&p)[K]) { double s = 0.0; for (int k = 0; k < K; k++) s += p[k]; return s; }
#elif defined(LHIP_HOSTSIM)
// ---------------------------------------------------------------------------------------------------------------------
// TEST-ONLY wave simulator (-DLHIP_HOSTSIM -DLHIP_WAVESIM): the 64 lanes of a wave are fibers (ucontext) of one host
// thread and every wave primitive is a rendezvous: a lane deposits its operand, yields, and is resumed once all 64 have
// deposited (the scheduler runs the lanes round-robin, each up to its next primitive).  Between two primitives a lane
// runs alone, so LDS hand-overs between lanes are only correct where the kernel orders them with wave_sync() -- exactly
// the discipline the device code needs for the compiler's sake.  Primitives must be reached by all lanes in the same
// order (checked by a tag), i.e. under wave-uniform control flow, as on the device.
// ---------------------------------------------------------------------------------------------------------------------
}  // namespace lhip
#include <ucontext.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
namespace lhip {
namespace wsim {
enum { NLANES = 64, MAXWAVES = 8, NFIB = NLANES * MAXWAVES, STACK = 1 << 20 };   // a workgroup of up to eight waves (the frame kernel in joint stereo)
// Context switch between the scheduler and a lane fiber.  x86-64: six callee-saved registers and the stack pointer (glibc's
// swapcontext also saves the signal mask -- a system call per switch, and a wave program switches ~10^4 times per frame);
// elsewhere ucontext.
#if defined(__x86_64__)
#define WSIM_ASM_SWITCH 1
extern "C" void lhip_wsim_switch(void** save_sp, void* load_sp);
asm(".text\n.globl lhip_wsim_switch\n.type lhip_wsim_switch,@function\nlhip_wsim_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size lhip_wsim_switch,.-lhip_wsim_switch\n");
#endif
struct Ctx {
#ifdef WSIM_ASM_SWITCH
    void* main_sp = nullptr; void* lane_sp[NFIB];
#else
    ucontext_t main_ctx, lane_ctx[NFIB];
#endif
    char* stacks = nullptr;
    int cur = 0, nfib = NLANES, done[NFIB];
    unsigned gen[NFIB], bar_gen[NFIB];
    long bar_arrived = 0;
    uint64_t slot[2][NFIB];
    int tag[2][NFIB];
    std::function<void(int)> body;          // argument: fiber index = wave * 64 + lane
};
inline Ctx*& current() { static thread_local Ctx* c = nullptr; return c; }
inline void to_scheduler(Ctx* c, int lane) {
#ifdef WSIM_ASM_SWITCH
    lhip_wsim_switch(&c->lane_sp[lane], c->main_sp);
#else
    swapcontext(&c->lane_ctx[lane], &c->main_ctx);
#endif
}
inline void to_lane(Ctx* c, int lane) {
#ifdef WSIM_ASM_SWITCH
    lhip_wsim_switch(&c->main_sp, c->lane_sp[lane]);
#else
    swapcontext(&c->main_ctx, &c->lane_ctx[lane]);
#endif
}
#ifdef WSIM_ASM_SWITCH
inline void trampoline() { Ctx* c = current(); const int lane = c->cur; c->body(lane); c->done[lane] = 1; to_scheduler(c, lane); abort(); }
#else
inline void trampoline(int lane) { Ctx* c = current(); c->body(lane); c->done[lane] = 1; to_scheduler(c, lane); }
#endif
// run `body(wave, lane)` for the 64 x nwaves lanes of one workgroup; the waves only meet at block_barrier()
template <class F> inline void run_block(int nwaves, F&& body) {
    static thread_local Ctx ctx;
    Ctx* c = &ctx;
    if (!c->stacks) c->stacks = (char*)malloc((size_t)NFIB * STACK);
    current() = c;
    c->nfib = NLANES * nwaves; c->bar_arrived = 0;
    c->body = [&body](int f) { body(f / NLANES, f % NLANES); };
    for (int l = 0; l < c->nfib; l++) {
        c->done[l] = 0; c->gen[l] = 0; c->bar_gen[l] = 0;
#ifdef WSIM_ASM_SWITCH
        // initial frame: six zeroed callee-saved registers, then the "return address" = trampoline; the stack is laid out so
        // that it is 16-byte aligned + 8 at the trampoline's entry, as after a call
        uintptr_t top = ((uintptr_t)(c->stacks + (size_t)(l + 1) * STACK)) & ~(uintptr_t)15;
        void** sp = (void**)(top - 8);
        *--sp = (void*)(void (*)())trampoline;
        for (int r = 0; r < 6; r++) *--sp = nullptr;
        c->lane_sp[l] = sp;
#else
        getcontext(&c->lane_ctx[l]);
        c->lane_ctx[l].uc_stack.ss_sp = c->stacks + (size_t)l * STACK;
        c->lane_ctx[l].uc_stack.ss_size = STACK;
        c->lane_ctx[l].uc_link = &c->main_ctx;
        makecontext(&c->lane_ctx[l], (void (*)())trampoline, 1, l);
#endif
    }
    for (;;) {
        int alive = 0;
        for (int l = 0; l < c->nfib; l++) if (!c->done[l]) { c->cur = l; to_lane(c, l); alive += !c->done[l]; }
        if (!alive) break;
    }
    current() = nullptr;
}
// run `body(lane)` for the 64 lanes of one wave
template <class F> inline void run(F&& body) { run_block(1, [&body](int, int lane) { body(lane); }); }
// deposit v under `tag`, wait for the other lanes, return everybody's deposits
inline void exchange(int tag, uint64_t v, uint64_t (&all)[NLANES]) {
    Ctx* c = current();
    const int me = c->cur, base = me - me % NLANES, par = (int)(c->gen[me] & 1u);
    c->slot[par][me] = v; c->tag[par][me] = tag; c->gen[me]++;
    to_scheduler(c, me);
    c->cur = me;
    for (int l = 0; l < NLANES; l++) {
        // the other lanes of my wave have deposited this primitive; some may already have run on to their next one (one
        // generation ahead, other parity): the lanes before me in the round, or -- after a block barrier -- the ones after me
        const int f = base + l;
        const bool gen_ok = (c->gen[f] == c->gen[me]) || (c->gen[f] == c->gen[me] + 1);
        if (!gen_ok || c->tag[par][f] != tag) {
            fprintf(stderr, "wavesim: lanes diverged at a wave primitive (fiber %d tag %d vs fiber %d tag %d, gen %u/%u, done %d)\n",
                    me, tag, f, c->tag[par][f], c->gen[me], c->gen[f], c->done[f]);
            abort();
        }
        all[l] = c->slot[par][f];
    }
}
// workgroup barrier (__syncthreads): every fiber of the block arrives, then all continue
inline void block_barrier() {
    Ctx* c = current();
    const int me = c->cur;
    c->bar_arrived++;
    const long target = (long)(++c->bar_gen[me]) * c->nfib;
    while (c->bar_arrived < target) { to_scheduler(c, me); c->cur = me; }
}
inline int my_lane() { return current()->cur % NLANES; }
template <class T> inline uint64_t bits_of(T v) { uint64_t u = 0; memcpy(&u, &v, sizeof v); return u; }
template <class T> inline T from_bits(uint64_t u) { T v; memcpy(&v, &u, sizeof v); return v; }
}  // namespace wsim
struct Wave { int lane; };
LHIP_DEV void wave_sync() { uint64_t a[64]; wsim::exchange(1, 0, a); }
LHIP_DEV void wave_sync_global() { wave_sync(); }
LHIP_DEV void wg_barrier() { wsim::block_barrier(); }                  // __syncthreads of a multi-wave workgroup
LHIP_DEV int wave_sum(int v) { uint64_t a[64]; wsim::exchange(2, (uint64_t)(uint32_t)v, a); int s = 0; for (int l = 0; l < 64; l++) s += (int)(uint32_t)a[l]; return s; }
template <int N> LHIP_DEV void wave_sum_n(int (&v)[N]);
template <int N> LHIP_DEV void wave_max_n(int (&v)[N]);
LHIP_DEV int wave_max(int v) { uint64_t a[64]; wsim::exchange(3, (uint64_t)(uint32_t)v, a); int m = (int)(uint32_t)a[0]; for (int l = 1; l < 64; l++) if ((int)(uint32_t)a[l] > m) m = (int)(uint32_t)a[l]; return m; }
template <int N> LHIP_DEV void wave_sum_n(int (&v)[N]) { for (int i = 0; i < N; i++) v[i] = wave_sum(v[i]); }
template <int N> LHIP_DEV void wave_max_n(int (&v)[N]) { for (int i = 0; i < N; i++) v[i] = wave_max(v[i]); }
LHIP_DEV int wave_min(int v) { uint64_t a[64]; wsim::exchange(4, (uint64_t)(uint32_t)v, a); int m = (int)(uint32_t)a[0]; for (int l = 1; l < 64; l++) if ((int)(uint32_t)a[l] < m) m = (int)(uint32_t)a[l]; return m; }
LHIP_DEV int wave_or(int v) { uint64_t a[64]; wsim::exchange(5, (uint64_t)(uint32_t)v, a); uint32_t m = 0; for (int l = 0; l < 64; l++) m |= (uint32_t)a[l]; return (int)m; }
LHIP_DEV uint64_t wave_or64(uint64_t v) { uint64_t a[64]; wsim::exchange(6, v, a); uint64_t m = 0; for (int l = 0; l < 64; l++) m |= a[l]; return m; }
LHIP_DEV float wave_maxf(float v) { uint64_t a[64]; wsim::exchange(7, wsim::bits_of(v), a); float m = wsim::from_bits<float>(a[0]); for (int l = 1; l < 64; l++) { const float x = wsim::from_bits<float>(a[l]); if (x > m) m = x; } return m; }
LHIP_DEV float wave_maxf_pos(float v) { return wave_maxf(v); }
LHIP_DEV double wave_maxd(double v) { uint64_t a[64]; wsim::exchange(8, wsim::bits_of(v), a); double m = wsim::from_bits<double>(a[0]); for (int l = 1; l < 64; l++) { const double x = wsim::from_bits<double>(a[l]); if (x > m) m = x; } return m; }
LHIP_DEV double wave_sumd(double v) { uint64_t a[64]; wsim::exchange(9, wsim::bits_of(v), a); double s = 0; for (int l = 0; l < 64; l++) s += wsim::from_bits<double>(a[l]); return s; }   // order-insensitive uses only
LHIP_DEV int wave_bcast(int v, int src) { uint64_t a[64]; wsim::exchange(10, (uint64_t)(uint32_t)v, a); return (int)(uint32_t)a[src]; }
LHIP_DEV uint64_t wave_ballot(int p) { uint64_t a[64]; wsim::exchange(11, p ? 1u : 0u, a); uint64_t m = 0; for (int l = 0; l < 64; l++) m |= a[l] << l; return m; }
LHIP_DEV int wave_any(int p) { return wave_ballot(p) != 0; }
LHIP_DEV int wave_excl_scan(int v, int lane, int* total) { uint64_t a[64]; wsim::exchange(12, (uint64_t)(uint32_t)v, a); int pre = 0, tot = 0; for (int l = 0; l < 64; l++) { if (l < lane) pre += (int)(uint32_t)a[l]; tot += (int)(uint32_t)a[l]; } *total = tot; return pre; }
LHIP_DEV void lds_or(uint32_t* p, uint32_t v) { *p |= v; }                 // lanes run one at a time: plain read-modify-write
LHIP_DEV void lds_max(int32_t* p, int32_t v) { if (*p < v) *p = v; }
LHIP_DEV uint64_t wave_lane_bits(uint64_t v) { return wave_ballot(v != 0); }
// uni(): on the device v_readfirstlane, i.e. "this value is wave-uniform" -- which silently takes lane 0's value if it is
// not.  With LAMEJS_WAVESIM_CHECK_UNI=1 the simulation verifies the claim (a rendezvous per call: slow, for sweeps).
inline bool wsim_check_uni() { static const bool on = [] { const char* e = getenv("LAMEJS_WAVESIM_CHECK_UNI"); return e && e[0] == '1'; }(); return on; }
LHIP_DEV int uni(int v) {
    if (wsim_check_uni()) {
        uint64_t a[64]; wsim::exchange(15, (uint64_t)(uint32_t)v, a);
        for (int l = 1; l < 64; l++) if (a[l] != a[0]) { fprintf(stderr, "wavesim: uni() of a non-uniform value (lane 0: %d, lane %d: %d)\n", (int)(uint32_t)a[0], l, (int)(uint32_t)a[l]); abort(); }
    }
    return v;
}
LHIP_DEV double unid(double v) { union { double d; int i[2]; } u; u.d = v; u.i[0] = uni(u.i[0]); u.i[1] = uni(u.i[1]); return u.d; }
LHIP_DEV int fresh_lane(int lane) { return lane; }
LHIP_DEV int lane_anew(int lane) { return lane; }
LHIP_DEV double const_here(double c) { return c; }
LHIP_DEV int uni_here(int v) { return v; }
LHIP_DEV unsigned mul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
LHIP_DEV double wave_shr1d(double v, double first) { uint64_t a[64]; wsim::exchange(13, wsim::bits_of(v), a); const int me = wsim::my_lane(); return me == 0 ? first : wsim::from_bits<double>(a[me - 1]); }
// the device's systolic fold, literally: 64 steps of "add my K values onto the sum handed over by lane - 1"
template <int K> LHIP_DEV double wave_seq_sum(const double (&p)[K]) {
    double carry = 0.0, s = 0.0;
    for (int st = 0; st < 64; st++) {
        s = carry;
        for (int k = 0; k < K; k++) s += p[k];
        carry = wave_shr1d(s, 0.0);
    }
    uint64_t a[64]; wsim::exchange(14, wsim::bits_of(s), a);
    return wsim::from_bits<double>(a[63]);
}
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
// the same hand-over point for data that lanes pass to each other through GLOBAL memory (the reservoir's header queue): the stores
// are made visible beyond the CU's vector cache and later loads do not reuse lines cached before
LHIP_DEV void wave_sync_global() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// Integer reductions over the 64 lanes (all lanes active: the kernels call them under wave-uniform control flow): four DPP row
// shifts leave each row's total in its last lane, row_bcast:15 / row_bcast:31 carry the totals into the later rows, lane 63 ends
// up with the wave's total, one v_readlane moves it to an SGPR (wave-uniform for the compiler).  Seven VALU instructions and no
// scalar code -- HIP's __reduce_*_sync wrap the same DPP steps in mask checks (~20 SALU instructions and several branches each),
// and the scalar unit is shared by the whole CU.  `IDENT` is what a lane receives when its DPP source does not exist.
#define LHIP_DPP_REDUCE(OP, IDENT)                                                                     \
    int x = v;                                                                                         \
    x = OP(x, __builtin_amdgcn_update_dpp(IDENT, x, 0x111, 0xf, 0xf, false));   /* row_shr:1 */        \
    x = OP(x, __builtin_amdgcn_update_dpp(IDENT, x, 0x112, 0xf, 0xf, false));   /* row_shr:2 */        \
    x = OP(x, __builtin_amdgcn_update_dpp(IDENT, x, 0x114, 0xf, 0xf, false));   /* row_shr:4 */        \
    x = OP(x, __builtin_amdgcn_update_dpp(IDENT, x, 0x118, 0xf, 0xf, false));   /* row_shr:8 */        \
    x = OP(x, __builtin_amdgcn_update_dpp(IDENT, x, 0x142, 0xa, 0xf, false));   /* row_bcast:15 -> rows 1, 3 */ \
    x = OP(x, __builtin_amdgcn_update_dpp(IDENT, x, 0x143, 0xc, 0xf, false));   /* row_bcast:31 -> rows 2, 3 */ \
    return __builtin_amdgcn_readlane(x, 63);
LHIP_DEV int dpp_op_add(int a, int b) { return a + b; }
LHIP_DEV int dpp_op_max(int a, int b) { return a > b ? a : b; }
LHIP_DEV int dpp_op_min(int a, int b) { return a < b ? a : b; }
LHIP_DEV int dpp_op_or(int a, int b) { return a | b; }
// N independent reductions side by side: step s of every value before step s + 1 of any, so that the two wait states a DPP read
// needs after the VALU write of its source are filled by the other chains instead of s_nop (which costs an issue slot like any
// other instruction).
#define LHIP_DPP_REDUCE_N(OP, IDENT)                                                                   \
    _Pragma("unroll") for (int i = 0; i < N; i++) v[i] = OP(v[i], __builtin_amdgcn_update_dpp(IDENT, v[i], 0x111, 0xf, 0xf, false)); \
    _Pragma("unroll") for (int i = 0; i < N; i++) v[i] = OP(v[i], __builtin_amdgcn_update_dpp(IDENT, v[i], 0x112, 0xf, 0xf, false)); \
    _Pragma("unroll") for (int i = 0; i < N; i++) v[i] = OP(v[i], __builtin_amdgcn_update_dpp(IDENT, v[i], 0x114, 0xf, 0xf, false)); \
    _Pragma("unroll") for (int i = 0; i < N; i++) v[i] = OP(v[i], __builtin_amdgcn_update_dpp(IDENT, v[i], 0x118, 0xf, 0xf, false)); \
    _Pragma("unroll") for (int i = 0; i < N; i++) v[i] = OP(v[i], __builtin_amdgcn_update_dpp(IDENT, v[i], 0x142, 0xa, 0xf, false)); \
    _Pragma("unroll") for (int i = 0; i < N; i++) v[i] = OP(v[i], __builtin_amdgcn_update_dpp(IDENT, v[i], 0x143, 0xc, 0xf, false)); \
    _Pragma("unroll") for (int i = 0; i < N; i++) v[i] = __builtin_amdgcn_readlane(v[i], 63);
template <int N> LHIP_DEV void wave_sum_n(int (&v)[N]) { LHIP_DPP_REDUCE_N(dpp_op_add, 0) }
template <int N> LHIP_DEV void wave_max_n(int (&v)[N]) { LHIP_DPP_REDUCE_N(dpp_op_max, (int)0x80000000) }
LHIP_DEV int wave_sum(int v) { LHIP_DPP_REDUCE(dpp_op_add, 0) }
LHIP_DEV int wave_max(int v) { LHIP_DPP_REDUCE(dpp_op_max, (int)0x80000000) }
LHIP_DEV int wave_min(int v) { LHIP_DPP_REDUCE(dpp_op_min, 0x7fffffff) }
LHIP_DEV int wave_or(int v) { LHIP_DPP_REDUCE(dpp_op_or, 0) }
LHIP_DEV uint64_t wave_or64(uint64_t v) {
    const unsigned lo = (unsigned)wave_or((int)(unsigned)v), hi = (unsigned)wave_or((int)(unsigned)(v >> 32));
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
LHIP_DEV float wave_maxf_pos(float v) { return __int_as_float(wave_max(__float_as_int(v))); }
LHIP_DEV double wave_maxd(double v) { return __ockl_wfred_max_f64(v); }
// tree sum: NOT order-exact; only for order-insensitive decisions
LHIP_DEV double wave_sumd(double v) { return __ockl_wfred_add_f64(v); }
LHIP_DEV int wave_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }   // src must be wave-uniform
LHIP_DEV uint64_t wave_ballot(int p) { return __ballot(p); }
LHIP_DEV unsigned mul24(unsigned a, unsigned b) { return __umul24(a, b); }
LHIP_DEV void wg_barrier() { __syncthreads(); }
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
// The lane index formed anew (two VALU instructions) instead of copied: for the entries of once-per-frame / once-per-granule code, where a copy
// would keep the ORIGINAL index alive across the quantization loop (it was parked in scratch memory and reloaded at every such entry).
LHIP_DEV int lane_anew(int lane) {
    (void)lane;
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    __builtin_assume(l >= 0 && l < LHIP_NL);
    return l;
}
// A double-precision literal materialised where it is used (two scalar moves): the compiler forms 64-bit literals of once-per-frame code
// in the kernel prologue and, register pairs not being rematerialisable for it, parked them in scratch memory for the whole kernel.
LHIP_DEV double const_here(double c) {
    union { double d; uint32_t u[2]; } x; x.d = c;
    asm volatile("" : "+s"(x.u[0]), "+s"(x.u[1]));
    return x.d;
}
// the same for a wave-uniform integer whose CONVERSIONS would otherwise be formed once per kernel and parked ((double)channels, ...)
LHIP_DEV int uni_here(int v) { v = __builtin_amdgcn_readfirstlane(v); asm volatile("" : "+s"(v)); return v; }
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

// ---------------------------------------------------------------------------------------------------------------------
// Signalling between the waves of ONE workgroup through words in LDS (k_quant_tail.h: tail help; k_quant.h: count_bits on a second wave).
// wg_store publishes everything the wave wrote to LDS before it; a wave that has seen the value calls wg_acquire before reading.
// ---------------------------------------------------------------------------------------------------------------------
#if defined(LHIP_HOSTSIM) && LHIP_NL == 1
// the one-lane simulation has no workgroups: nothing ever polls
LHIP_DEV int wg_cas(int* p, int expect, int desired, int lane) { (void)lane; const int r = *p; if (r == expect) *p = desired; return r; }
LHIP_DEV int wg_load(const int* p, int lane) { (void)lane; return *p; }
LHIP_DEV void wg_store(int* p, int v, int lane) { (void)lane; *p = v; }
LHIP_DEV void wg_add(int* p, int v, int lane) { (void)lane; *p += v; }
LHIP_DEV void wg_load2(const int* p, int* a, int* b, int lane) { (void)lane; *a = p[0]; *b = p[1]; }
LHIP_DEV void wg_store2(int* p, int a, int b, int lane) { (void)lane; p[0] = a; p[1] = b; }
LHIP_DEV void wg_idle() {}
LHIP_DEV int wg_wave_id() { return 0; }
LHIP_DEV void wg_pause(int) {}
LHIP_DEV void wg_spin() {}
LHIP_DEV void wg_acquire() {}
#elif defined(LHIP_HOSTSIM)
// lane fibers run one at a time and only switch at wave primitives: plain accesses are atomic; wave_bcast is the switch point that lets the
// other waves of the workgroup run while this one polls
LHIP_DEV int wg_cas(int* p, int expect, int desired, int lane) { int r = 0; if (lane == 0) { r = *p; if (r == expect) *p = desired; } return wave_bcast(r, 0); }
LHIP_DEV int wg_load(const int* p, int lane) { int r = 0; if (lane == 0) r = *(const volatile int*)p; return wave_bcast(r, 0); }
LHIP_DEV void wg_store(int* p, int v, int lane) { wave_sync(); if (lane == 0) *(volatile int*)p = v; wave_sync(); }
LHIP_DEV void wg_add(int* p, int v, int lane) { if (lane == 0) *p += v; wave_sync(); }
LHIP_DEV void wg_load2(const int* p, int* a, int* b, int lane) { int r0 = 0, r1 = 0; if (lane == 0) { r0 = ((const volatile int*)p)[0]; r1 = ((const volatile int*)p)[1]; } *a = wave_bcast(r0, 0); *b = wave_bcast(r1, 0); }
LHIP_DEV void wg_store2(int* p, int a, int b, int lane) { wave_sync(); if (lane == 0) { ((volatile int*)p)[0] = a; ((volatile int*)p)[1] = b; } wave_sync(); }
LHIP_DEV void wg_idle() { wave_sync(); }
LHIP_DEV int wg_wave_id() { return wsim::current()->cur / 64; }       // the calling lane's wave inside its workgroup
LHIP_DEV void wg_pause(int) { wave_sync(); }
LHIP_DEV void wg_spin() { wave_sync(); }
LHIP_DEV void wg_acquire() { wave_sync(); }
#else
LHIP_DEV int wg_cas(int* p, int expect, int desired, int lane) { int r = 0; if (lane == 0) r = atomicCAS(p, expect, desired); return __builtin_amdgcn_readfirstlane(r); }
LHIP_DEV int wg_load(const int* p, int lane) { (void)lane; return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
LHIP_DEV void wg_store(int* p, int v, int lane) {          // everything this wave wrote to LDS before is visible to a wave that sees v
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
LHIP_DEV void wg_add(int* p, int v, int lane) { if (lane == 0) atomicAdd(p, v); }
// two adjacent words (8-byte aligned) in one LDS access: a pair of helpers' state words read / posted at once
LHIP_DEV void wg_load2(const int* p, int* a, int* b, int lane) {
    (void)lane;
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    *a = __builtin_amdgcn_readfirstlane((int)(unsigned)v); *b = __builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
}
LHIP_DEV void wg_store2(int* p, int a, int b, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store((unsigned long long*)p, (unsigned long long)(unsigned)a | ((unsigned long long)(unsigned)b << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
LHIP_DEV void wg_idle() { __builtin_amdgcn_s_sleep(8); }
LHIP_DEV int wg_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
#define wg_pause(N_) do { if ((N_) > 0) __builtin_amdgcn_s_sleep(N_); } while (0)       /* (s_sleep takes an immediate) */
#ifndef LHIP_SPIN_SLEEP
#define LHIP_SPIN_SLEEP 0
#endif
// Pause between the looks of a poll that sits on a frame's critical path (the count helper's hand-overs).  None: the helpers only run while every workgroup
// has a CU to itself (g_frame always -- its LDS fills the CU --, g_resv_stream when the host says so, run_batch), so a spinning wave takes issue slots from
// nobody but idle waves of its own workgroup.  (With two workgroups per CU a helper spinning without a pause cost the other workgroup's searching wave
// 11 %, profiles/r05_pass4_*; -DLHIP_SPIN_SLEEP=2 is the 128-clock pause that cured it before the host stopped enabling helpers there.)
LHIP_DEV void wg_spin() { if (LHIP_SPIN_SLEEP) __builtin_amdgcn_s_sleep(LHIP_SPIN_SLEEP); }
LHIP_DEV void wg_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
#endif
// Spin until the word is neither a nor b; returns what it is then.  For hand-overs on a frame's critical path (the count helper).  A counted loop
// with nothing in it but the load and the compares: the compiler unrolls it into read / wait / compare / branch groups, one LDS latency per look
// (tools/ubench_handoff.hip: what a hand-over costs).  A poll that has gone round 2^27 times (seconds) is a protocol or compiler bug: fault instead of
// hanging the device.
LHIP_DEV int wg_wait_not(const int* p, int a, int b, int lane) {
#if defined(LHIP_HOSTSIM)
    for (unsigned long n = 0; n < (1ul << 34); n++) {
#else
    for (unsigned n = 0; n < (1u << 27); n++) {
#endif
        const int s = wg_load(p, lane);
        if (s != a && s != b) return s;
        wg_spin();
    }
#if defined(LHIP_HOSTSIM)
    abort();
#else
    __builtin_trap();
#endif
}
// A meeting point of SOME waves of a workgroup (the hardware barrier is all or nothing): every participant announces itself on a counter in LDS that
// the caller has zeroed behind an earlier workgroup barrier, then waits until `n` have.  Everything a participant wrote to LDS before is visible after.
LHIP_DEV void wg_meet(int* p, int n, int lane) {
#if defined(LHIP_HOSTSIM) && LHIP_NL == 1
    (void)p; (void)n; (void)lane;
#else
#if !defined(LHIP_HOSTSIM)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#endif
    wg_add(p, 1, lane);
#if defined(LHIP_HOSTSIM)
    for (unsigned long k = 0; k < (1ul << 34); k++) {
#else
    for (unsigned k = 0; k < (1u << 27); k++) {
#endif
        if (wg_load(p, lane) >= n) { wg_acquire(); return; }
        wg_spin();
    }
#if defined(LHIP_HOSTSIM)
    abort();
#else
    __builtin_trap();
#endif
#endif
}
// A poll loop that has gone round this often (seconds; a launch's tail is milliseconds) is a protocol or compiler bug: fault instead of
// hanging the device (lhip_api.cpp, g_fixup: a dispenser loop nested in another loop has been miscompiled into an exec-masked loop on this
// toolchain before).
#if defined(LHIP_HOSTSIM)
#define LHIP_SPIN_GUARD(n) do { if (++(n) > (1l << 34)) abort(); } while (0)
#else
#define LHIP_SPIN_GUARD(n) do { if (++(n) > (1l << 24)) __builtin_trap(); } while (0)
#endif

}  // namespace lhip
