// DEV TOOL (GPU box): VALU / SALU / LDS issue-rate microbenchmark for gfx950.
//
// Question it answers ("weak" item 5): the quantization kernel issues ~485 G wave-instructions/s; is that 79 % of
// the chip's ceiling (4 cycles per wave64 VALU instruction) or 39 % (2 cycles)?  The MI355X guide measures v_fma_f32 at 2
// cycles; g_quant's mix is f64 arithmetic, conversions, 32-bit integer/logic, DPP moves and LDS gathers.  Every class below is
// run as an unrolled block of INDEPENDENT instructions (8 accumulator chains, so dependent-issue latency is not what is
// measured) by W waves per SIMD on every CU, and reported as
//     cycles per wave-instruction per SIMD = elapsed shader cycles * 1 / (W * instructions per wave)
// with the shader clock taken from s_memtime inside the kernel (not an assumed 2.4 GHz).
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/ubench_issue tools/ubench_issue.hip ; run: tools/_build/ubench_issue [json]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

enum { ITER = 2000, UNROLL = 32 };   // instructions per wave = ITER * UNROLL (* ops per macro)

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
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
        for (int it = 0; it < iters; it++) {
#define KERNEL_END()                                                                                         \
        }                                                                                                    \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
        unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;                           \
        r ^= __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);                                        \
        r ^= (unsigned)__double_as_longlong(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);                         \
        if (r == 0x12345678u) out[0] = r;                                                                    \
        if (threadIdx.x == 0) atomicMax(cyc + 0, t1 - t0);                                                   \
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
};

int main(int argc, char** argv) {
    const char* json = argc > 1 ? argv[1] : nullptr;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 64));
    std::string js = "{\"device\": \"" + std::string(prop.gcnArchName) + "\", \"cus\": " + std::to_string(cus) + ", \"unit\": \"shader cycles per wave64 instruction per SIMD (independent instructions)\", \"cases\": {";
    printf("%-22s %-9s  cycles/inst/SIMD at 1, 2, 4 waves per SIMD   (clock MHz at 4)\n", "instruction", "class");
    bool first = true;
    for (const Case& c : cases) {
        double res[3] = {0, 0, 0}; double mhz = 0;
        const int wps[3] = {1, 2, 4};
        for (int k = 0; k < 3; k++) {
            const int threads = 64 * 4 * wps[k];          // one block per CU, wps waves on each of the 4 SIMDs
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(c.fn, dim3(cus), dim3(threads), 0, 0, out, cyc, 50);      // warm-up
            CK(hipMemset(cyc, 0, 64));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(c.fn, dim3(cus), dim3(threads), 0, 0, out, cyc, (int)ITER);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            unsigned long long hc = 0; CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double insts = (double)ITER * c.ops_per_iter;
            res[k] = (double)hc / (insts * wps[k]);        // s_memtime ticks = shader cycles (MI355X guide)
            mhz = (double)hc / (ms * 1e3);
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
        printf("%-22s %-9s  %6.2f %6.2f %6.2f   (%.0f)\n", c.name + 2, c.cls, res[0], res[1], res[2], mhz);
        char buf[256];
        snprintf(buf, sizeof buf, "%s\"%s\": {\"class\": \"%s\", \"w1\": %.3f, \"w2\": %.3f, \"w4\": %.3f}", first ? "" : ", ", c.name + 2, c.cls, res[0], res[1], res[2]);
        js += buf; first = false;
    }
    js += "}}\n";
    if (json) { FILE* f = fopen(json, "w"); if (f) { fputs(js.c_str(), f); fclose(f); } }
    return 0;
}
