// DEV TOOL (GPU box): VALU / SALU / LDS issue-rate microbenchmark for gfx950.
//
// Question it answers ("weak" item 5): the quantization kernel issues ~485 G wave-instructions/s; is that 79 % of
// the chip's ceiling (4 cycles per wave64 VALU instruction) or 39 % (2 cycles)?  The MI355X guide measures v_fma_f32 at 2
// cycles; g_quant's mix is f64 arithmetic, conversions, 32-bit integer/logic, DPP moves and LDS gathers.  Every class below is
// run as an unrolled block of INDEPENDENT instructions (8 accumulator chains, so dependent-issue latency is not what is
// measured) by W waves per SIMD (grid = 4 W waves x CUs: W = 1, 2, 4 as workgroups of 256 W threads, W = 5, 6, 8 as twice as many
// workgroups of 128 W threads), each launch long enough (2-45 ms) for the clock to settle, and reported in WALL-CLOCK terms from
// hipEvents around the launch:
//     G wave-instructions/s over the whole chip, and shader cycles per wave-instruction per SIMD (clock = s_memtime ticks per
//     s_memrealtime microsecond inside the kernel: s_memtime ticks are shader cycles, the real-time counter runs at 100 MHz).
// (The per-workgroup in-kernel duration is NOT used for rates: workgroups of one launch do not all run concurrently -- with W >= 2
// the launch takes up to 3x one workgroup's duration -- so only the launch's wall time says what the chip sustained.)
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/ubench_issue tools/ubench_issue.hip ; run: tools/_build/ubench_issue [json]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

enum { ITER = 30000, UNROLL = 32 };   // instructions per wave = ITER * UNROLL (* ops per macro)

// 8 independent 32-bit chains a0..a7, 8 independent 64-bit chains d0..d7
#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define REP32(M) REP8(M) REP8(M) REP8(M) REP8(M)

#define KERNEL_BEGIN(name)                                                                                   \
    __global__ __launch_bounds__(1024) void name(unsigned* out, unsigned long long* cyc, int iters) {        \
        __shared__ unsigned lds[4096];                                                                       \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float f0 = a0 * 1e-3f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;     \
        double d0 = a0 * 1e-3, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;     \
        unsigned s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7; \
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;                                     \
        __syncthreads();                                                                                     \
        const unsigned la = (threadIdx.x & 63) * 4;   /* conflict-free LDS address */                        \
        (void)la; (void)s0; (void)s1; (void)s2; (void)s3; (void)s4; (void)s5; (void)s6; (void)s7;                                                  \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                      \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
        for (int it = 0; it < iters; it++) {
#define KERNEL_END()                                                                                         \
        }                                                                                                    \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                      \
        unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;                           \
        r ^= __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);                                        \
        r ^= (unsigned)__double_as_longlong(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);                         \
        if (r == 0x12345678u) out[0] = r;                                                                    \
        if (threadIdx.x == 0) { atomicMax(cyc + 0, t1 - t0); atomicMax(cyc + 1, r1 - r0); }                  \
    }

#define A32(op, i) asm volatile(op " %0, %0, %1" : "+v"(a##i) : "v"(a0 | 1u));
#define F32(op, i) asm volatile(op " %0, %0, %1" : "+v"(f##i) : "v"(1.0001f));
#define F64(op, i) asm volatile(op " %0, %0, %1" : "+v"(d##i) : "v"(1.0000001));

KERNEL_BEGIN(k_add_u32)
#define M(i) A32("v_add_u32", i)
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_and_b32)
#define M(i) A32("v_and_b32", i)
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_lshl_b32)
#define M(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_max_i32)
#define M(i) A32("v_max_i32", i)
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_mul_lo_u32)
#define M(i) A32("v_mul_lo_u32", i)
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_mad_u32_u24)
#define M(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a##i) : "v"(a0 | 1u));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_add3_u32)
#define M(i) asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(a##i) : "v"(a0 | 1u));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_bfe_u32)
#define M(i) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_cndmask)
#define M(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(a0 | 1u) : "vcc");
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_cmp_cndmask)
#define M(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(a0 | 1u) : "vcc");
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_mov_dpp)
#define M(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_add_dpp)
#define M(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_mov_dpp_wave_shr)
#define M(i) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_readlane)
#define M(i) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s0) : "v"(a##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_readfirstlane)
#define M(i) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s0) : "v"(a##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_add_f32)
#define M(i) F32("v_add_f32", i)
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_mul_f32)
#define M(i) F32("v_mul_f32", i)
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_fma_f32)
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f##i) : "v"(1.0001f));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_cvt_i32_f32)
#define M(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(a##i) : "v"(f##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_cvt_f32_i32)
#define M(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f##i) : "v"(a##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_add_f64)
#define M(i) F64("v_add_f64", i)
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_mul_f64)
#define M(i) F64("v_mul_f64", i)
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_fma_f64)
#define M(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d##i) : "v"(1.0000001));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_cvt_f64_f32)
#define M(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d##i) : "v"(f##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_cvt_f32_f64)
#define M(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f##i) : "v"(d##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_cvt_i32_f64)
#define M(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a##i) : "v"(d##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_cvt_f64_i32)
#define M(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d##i) : "v"(a##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_max_f64)
#define M(i) F64("v_max_f64", i)
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_sqrt_f64)
#define M(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_rcp_f64)
#define M(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d##i));
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_sqrt_f32)
#define M(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f##i));
    REP32(M)
#undef M
KERNEL_END()
// scalar ALU alone, and interleaved 1:1 with VALU (does the scalar unit issue beside the vector unit?)
KERNEL_BEGIN(k_salu_add)
#define M(i) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s##i) : : "scc");
    REP8(M) REP8(M) REP8(M) REP8(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_valu_salu_1to1)
#define M(i) asm volatile("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, 3" : "+v"(a##i), "+s"(s##i) : "v"(a0 | 1u) : "scc");
    REP32(M)
#undef M
KERNEL_END()
// f64 and 32-bit integer interleaved 1:1 (do the two classes share one issue port?)
KERNEL_BEGIN(k_f64_int_1to1)
#define M(i) asm volatile("v_mul_f64 %0, %0, %2\n v_add_u32 %1, %1, %3" : "+v"(d##i), "+v"(a##i) : "v"(1.0000001), "v"(a0 | 1u));
    REP32(M)
#undef M
KERNEL_END()
// LDS: independent conflict-free reads / byte gathers / writes (issue rate of DS instructions per SIMD)
KERNEL_BEGIN(k_ds_read_b32)
#define M(i) asm volatile("ds_read_b32 %0, %1 offset:" #i "*256" : "=v"(a##i) : "v"(la));
    REP32(M)
    asm volatile("s_waitcnt lgkmcnt(0)");
#undef M
KERNEL_END()
KERNEL_BEGIN(k_ds_read_u8)
#define M(i) asm volatile("ds_read_u8 %0, %1 offset:" #i "*256" : "=v"(a##i) : "v"(la));
    REP32(M)
    asm volatile("s_waitcnt lgkmcnt(0)");
#undef M
KERNEL_END()
KERNEL_BEGIN(k_ds_read_b64)
#define M(i) asm volatile("ds_read_b64 %0, %1 offset:" #i "*512" : "=v"(d##i) : "v"(la * 2));
    REP32(M)
    asm volatile("s_waitcnt lgkmcnt(0)");
#undef M
KERNEL_END()
KERNEL_BEGIN(k_ds_write_b32)
#define M(i) asm volatile("ds_write_b32 %1, %0 offset:" #i "*256" : : "v"(a##i), "v"(la));
    REP32(M)
    asm volatile("s_waitcnt lgkmcnt(0)");
#undef M
KERNEL_END()
// LDS gather interleaved with VALU 1:3 (the shape of the Huffman length-sum loop)
KERNEL_BEGIN(k_ds_u8_valu_1to3)
#define M(i) asm volatile("ds_read_u8 %0, %2 offset:" #i "*256\n v_add_u32 %1, %1, %3\n v_and_b32 %1, %1, %3\n v_add_u32 %1, %1, %3" : "=v"(a##i), "+v"(a7) : "v"(la), "v"(a0 | 1u));
    REP8(M)
    asm volatile("s_waitcnt lgkmcnt(0)");
#undef M
KERNEL_END()

// The quantization kernel's measured mix (profiles/r02_pmc_*: per 16 VALU = 5 int32, 2 f64, 1 cvt, 8 moves/compares/selects;
// 10.5 SALU, 1.3 LDS reads, ~2 branches), as independent instructions: the issue ceiling of THAT mix at a given occupancy.
KERNEL_BEGIN(k_quant_mix)
#define M(i) asm volatile( \
    "v_add_u32 %0, %0, %4\n s_add_u32 %3, %3, 3\n v_and_b32 %0, %0, %4\n s_lshl_b32 %3, %3, 1\n v_lshlrev_b32 %0, 1, %0\n s_and_b32 %3, %3, 0xffff\n" \
    "v_fma_f64 %1, %1, %5, %1\n s_add_u32 %3, %3, 5\n v_mov_b32 %2, %0\n s_cmp_lt_u32 %3, 77\n v_cmp_lt_u32 vcc, %0, %4\n s_cselect_b32 %3, %3, 9\n" \
    "v_cndmask_b32 %2, %2, %4, vcc\n v_cvt_f64_u32 %1, %2\n s_add_u32 %3, %3, 1\n v_mul_f64 %1, %1, %5\n v_add_u32 %0, %0, %2\n s_xor_b32 %3, %3, 21\n" \
    "v_mov_b32 %2, %0\n v_max_i32 %0, %0, %4\n s_add_u32 %3, %3, 7\n v_cmp_gt_u32 vcc, %0, %4\n v_cndmask_b32 %2, %2, %4, vcc\n s_lshr_b32 %3, %3, 1\n" \
    "v_mov_b32 %2, %2\n v_bfe_u32 %0, %0, 1, 20\n ds_read_b32 %2, %6 offset:" #i "*256" \
    : "+v"(a##i), "+v"(d##i), "+v"(f##i), "+s"(s##i) : "v"(a0 | 1u), "v"(1.0000001), "v"(la) : "vcc", "scc");
    REP8(M)
    asm volatile("s_waitcnt lgkmcnt(0)");
#undef M
KERNEL_END()

// ---- selects: what does v_cndmask cost depending on where its mask comes from?  (k_cndmask above, 16 cycles, reads a VCC that nobody
// has written for thousands of instructions; compiled code mostly selects on masks that are not freshly written either) ----
KERNEL_BEGIN(k_sel_vcc_set_per_iter)
    asm volatile("s_mov_b64 vcc, exec" ::: "vcc");
#define M(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(a0 | 1u) : "vcc");
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_sel_sgpr_mask)
    asm volatile("s_mov_b64 s[40:41], exec" ::: "s40", "s41");
#define M(i) asm volatile("v_cndmask_b32 %0, %0, %1, s[40:41]" : "+v"(a##i) : "v"(a0 | 1u) : "s40", "s41");
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_sel_cmp1_cnd4)
#define M(i) asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" \
    : "+v"(a##i), "+v"(s##i##x), "+v"(s##i##y), "+v"(s##i##z) : "v"(a0 | 1u) : "vcc");
    unsigned s0x = a0, s0y = a0, s0z = a0, s1x = a1, s1y = a1, s1z = a1, s2x = a2, s2y = a2, s2z = a2, s3x = a3, s3y = a3, s3z = a3,
             s4x = a4, s4y = a4, s4z = a4, s5x = a5, s5y = a5, s5z = a5, s6x = a6, s6y = a6, s6z = a6, s7x = a7, s7y = a7, s7z = a7;
    REP8(M)
    a0 ^= s0x ^ s0y ^ s0z ^ s1x ^ s1y ^ s1z ^ s2x ^ s2y ^ s2z ^ s3x ^ s3y ^ s3z ^ s4x ^ s4y ^ s4z ^ s5x ^ s5y ^ s5z ^ s6x ^ s6y ^ s6z ^ s7x ^ s7y ^ s7z;
#undef M
KERNEL_END()
KERNEL_BEGIN(k_sel_cmp_add_cnd)
#define M(i) asm volatile("v_cmp_lt_u32 vcc, %0, %2\n v_add_u32 %1, %1, %2\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a##i), "+v"(f##i) : "v"(a0 | 1u) : "vcc");
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_sel_salu_mask_cnd)
#define M(i) asm volatile("s_and_b64 vcc, exec, exec\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(a0 | 1u) : "vcc", "scc");
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_sel_cmp_sgpr_cnd)
#define M(i) asm volatile("v_cmp_lt_u32 s[40:41], %0, %1\n v_cndmask_b32 %0, %0, %1, s[40:41]" : "+v"(a##i) : "v"(a0 | 1u) : "s40", "s41");
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_sel_cmp_vcc_only)
#define M(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(a0 | 1u) : "vcc");
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_sel_cmp_sgpr_only)
#define M(i) asm volatile("v_cmp_lt_u32 s[40:41], %0, %1\n v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(a0 | 1u) : "s40", "s41");
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_sel_add_sgpr_operand)
    asm volatile("s_mov_b32 s40, 3" ::: "s40");
#define M(i) asm volatile("v_add_u32 %0, s40, %0" : "+v"(a##i) :: "s40");
    REP32(M)
#undef M
KERNEL_END()
KERNEL_BEGIN(k_sel_cmp_far_cnd)
    // the mask is written eight instructions before it is used (the compiler's usual distance when it hoists compares)
#define M(i) asm volatile("v_cmp_lt_u32 s[40:41], %0, %1" :: "v"(a##i), "v"(a0 | 1u) : "s40", "s41");
#define N(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(a0 | 1u));
#define P(i) asm volatile("v_cndmask_b32 %0, %0, %1, s[40:41]" : "+v"(a##i) : "v"(a0 | 1u) : "s40", "s41");
    M(0) N(1) N(2) N(3) N(4) N(5) N(6) N(7) N(1) P(0)  M(1) N(2) N(3) N(4) N(5) N(6) N(7) N(0) N(2) P(1)
    M(2) N(3) N(4) N(5) N(6) N(7) N(0) N(1) N(3) P(2)  M(3) N(4) N(5) N(6) N(7) N(0) N(1) N(2) N(4) P(3)
#undef M
#undef N
#undef P
KERNEL_END()
KERNEL_BEGIN(k_sel_bfi)
    // arithmetic select: (a & m) | (b & ~m) with a lane mask value kept in a VGPR
#define M(i) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a##i) : "v"(a0 | 1u));
    REP32(M)
#undef M
KERNEL_END()

struct Case { const char* name; void (*fn)(unsigned*, unsigned long long*, int); int ops_per_iter; const char* cls; };
#define C(name, ops, cls) {#name, name, ops, cls}
static const Case cases[] = {
    C(k_add_u32, 32, "int32"), C(k_and_b32, 32, "int32"), C(k_lshl_b32, 32, "int32"), C(k_max_i32, 32, "int32"), C(k_bfe_u32, 32, "int32"),
    C(k_add3_u32, 32, "int32"), C(k_mad_u32_u24, 32, "int32"), C(k_mul_lo_u32, 32, "int32-mul"), C(k_cndmask, 32, "int32"), C(k_cmp_cndmask, 64, "int32"),
    C(k_mov_dpp, 32, "dpp"), C(k_add_dpp, 32, "dpp"), C(k_mov_dpp_wave_shr, 32, "dpp"), C(k_readlane, 32, "lane"), C(k_readfirstlane, 32, "lane"),
    C(k_add_f32, 32, "f32"), C(k_mul_f32, 32, "f32"), C(k_fma_f32, 32, "f32"), C(k_cvt_i32_f32, 32, "cvt32"), C(k_cvt_f32_i32, 32, "cvt32"), C(k_sqrt_f32, 32, "trans32"),
    C(k_add_f64, 32, "f64"), C(k_mul_f64, 32, "f64"), C(k_fma_f64, 32, "f64"), C(k_max_f64, 32, "f64"),
    C(k_cvt_f64_f32, 32, "cvt64"), C(k_cvt_f32_f64, 32, "cvt64"), C(k_cvt_i32_f64, 32, "cvt64"), C(k_cvt_f64_i32, 32, "cvt64"),
    C(k_sqrt_f64, 32, "trans64"), C(k_rcp_f64, 32, "trans64"),
    C(k_salu_add, 32, "salu"), C(k_valu_salu_1to1, 64, "mix"), C(k_f64_int_1to1, 64, "mix"),
    C(k_ds_read_b32, 32, "lds"), C(k_ds_read_u8, 32, "lds"), C(k_ds_read_b64, 32, "lds"), C(k_ds_write_b32, 32, "lds"), C(k_ds_u8_valu_1to3, 32, "mix"),
    C(k_quant_mix, 8 * 16, "quantmix(VALU only counted: 16 VALU + 10 SALU + 1 LDS per group)"),
    C(k_sel_vcc_set_per_iter, 32, "select"), C(k_sel_sgpr_mask, 32, "select"), C(k_sel_cmp1_cnd4, 40, "select"), C(k_sel_cmp_add_cnd, 96, "select"),
    C(k_sel_salu_mask_cnd, 32, "select(VALU only counted)"), C(k_sel_cmp_sgpr_cnd, 64, "select"), C(k_sel_cmp_vcc_only, 64, "select"), C(k_sel_cmp_sgpr_only, 64, "select"),
    C(k_sel_add_sgpr_operand, 32, "select"), C(k_sel_cmp_far_cnd, 40, "select"), C(k_sel_bfi, 32, "select"),
};

int main(int argc, char** argv) {
    const char* json = argc > 1 ? argv[1] : nullptr;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 64));
    std::string js = "{\"device\": \"" + std::string(prop.gcnArchName) + "\", \"cus\": " + std::to_string(cus) +
                     ", \"unit\": \"G wave64 instructions per second, whole chip, wall clock (hipEvents around the launch); independent instructions\", \"waves_per_simd\": [1, 2, 4, 5, 6, 8], \"cases\": {";
    const int wps[6] = {1, 2, 4, 5, 6, 8};
    printf("%-22s %-9s  G wave-inst/s at 1, 2, 4, 5, 6, 8 waves/SIMD | cycles/inst/SIMD at 4 and 8 | shader MHz at 4\n", "instruction", "class");
    bool first = true;
    const char* only = argc > 2 ? argv[2] : nullptr;          // run only the cases whose name contains this
    for (const Case& c : cases) {
        if (only && !strstr(c.name, only)) continue;
        double rate[6], nsi[6], tick_mhz[6], ev[6], inker[6];
        for (int k = 0; k < 6; k++) {
            const int w = wps[k];
            const int blocks_per_cu = w <= 4 ? 1 : 2, threads = 64 * 4 * w / blocks_per_cu;
            hipLaunchKernelGGL(c.fn, dim3(cus * blocks_per_cu), dim3(threads), 0, 0, out, cyc, 2000);      // warm-up (clock ramp)
            CK(hipMemset(cyc, 0, 64));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(c.fn, dim3(cus * blocks_per_cu), dim3(threads), 0, 0, out, cyc, (int)ITER);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ev_ms = 0; CK(hipEventElapsedTime(&ev_ms, e0, e1));
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
            ev[k] = ev_ms;
            unsigned long long hc[2] = {0, 0}; CK(hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost));
            const double insts = (double)ITER * c.ops_per_iter;     // per wave
            const double ksecs = (double)hc[1] / 100e6;             // one workgroup's duration (s_memrealtime: 100 MHz)
            const double secs = ev_ms * 1e-3;                       // the launch
            tick_mhz[k] = (double)hc[0] / (ksecs * 1e6);            // shader clock while the kernel ran
            rate[k] = insts * w * 4 * cus / secs / 1e9;
            nsi[k] = secs * tick_mhz[k] * 1e6 / (insts * w);        // shader cycles per wave-instruction per SIMD
            inker[k] = ksecs * 1e3;
        }
        printf("%-22s %-9.9s  %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f | %6.3f %6.3f | %6.0f\n", c.name + 2, c.cls, rate[0], rate[1], rate[2], rate[3], rate[4], rate[5], nsi[2], nsi[5], tick_mhz[2]);
        printf("      kernel ms by hipEvents / by s_memrealtime inside the kernel: ");
        for (int k = 0; k < 6; k++) printf(" %.3f/%.3f", ev[k], inker[k]);
        printf("\n");
        char buf[512];
        snprintf(buf, sizeof buf, "%s\"%s\": {\"class\": \"%s\", \"ginst_per_s\": [%.1f, %.1f, %.1f, %.1f, %.1f, %.1f], \"cycles_per_inst_per_simd\": [%.4f, %.4f, %.4f, %.4f, %.4f, %.4f], \"shader_mhz_w4\": %.0f}",
                 first ? "" : ", ", c.name + 2, c.cls, rate[0], rate[1], rate[2], rate[3], rate[4], rate[5], nsi[0], nsi[1], nsi[2], nsi[3], nsi[4], nsi[5], tick_mhz[2]);
        js += buf; first = false;
    }
    js += "}}\n";
    if (json) { FILE* f = fopen(json, "w"); if (f) { fputs(js.c_str(), f); fclose(f); } }
    return 0;
}
