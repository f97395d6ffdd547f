// DEV TOOL (GPU box): what a hand-over between two waves of ONE workgroup through a word in LDS costs on gfx950.
//
// Question it answers: the count helper's hand-over was measured inside g_frame at ~750 cycles per leg (request posted -> seen
// by the helper, reply written -> seen by the owner) with s_memtime stamps on both sides.  Is that the hardware (LDS round trip + poll loop), the
// stamps themselves (s_memtime is a scalar-memory instruction), or something the kernel adds?  One workgroup of eight waves; wave 0 and wave P
// play ping-pong on one LDS word with exactly lhip_wave.h's primitives (release fence + relaxed store; relaxed load + readfirstlane; acquire
// fence), N round trips, cycles per ROUND TRIP from one s_memtime pair around the whole run; the other waves wait at the final barrier.
// Variants: partner on another SIMD (wave 1) / on the same SIMD (wave 4); s_sleep 0 / 1 / 2 between polls; with a payload of 288 words written
// before the store and read after the acquire; the same through a word in global memory (agent scope); and the latency of s_memtime and of a
// dependent LDS read themselves.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/ubench_handoff tools/ubench_handoff.hip ; run: tools/_build/ubench_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ int wg_load(const int* p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ void wg_store(int* p, int v, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void wg_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
template <int SLEEP> __device__ __forceinline__ void wait_for(const int* p, int v) {
    for (;;) {
        if (wg_load(p) == v) return;
        if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
    }
}

// ping-pong: wave 0 writes 2i+1, the partner answers 2i+2
template <int SLEEP, int PAYLOAD>
__global__ __launch_bounds__(512) void k_pingpong(unsigned long long* cyc, int n, int partner, unsigned* sink) {
    __shared__ int flag[64];
    __shared__ unsigned pay[2][320];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) flag[0] = 0;
    for (int i = threadIdx.x; i < 640; i += 512) (&pay[0][0])[i] = i;
    __syncthreads();
    unsigned acc = 0;
    if (wv == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < n; i++) {
            if (PAYLOAD) { for (int k = lane; k < 288; k += 64) pay[0][k] = acc + k + i; }
            wg_store(flag, 2 * i + 1, lane);
            wait_for<SLEEP>(flag, 2 * i + 2);
            if (PAYLOAD) { wg_acquire(); acc += pay[1][lane] + pay[1][lane + 64]; }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) cyc[0] = t1 - t0;
    } else if (wv == partner) {
        for (int i = 0; i < n; i++) {
            wait_for<SLEEP>(flag, 2 * i + 1);
            if (PAYLOAD) { wg_acquire(); for (int k = lane; k < 288; k += 64) acc += pay[0][k]; pay[1][lane] = acc; pay[1][lane + 64] = acc + 1; }
            wg_store(flag, 2 * i + 2, lane);
        }
    }
    __syncthreads();
    if (acc == 0x12345678u) sink[0] = acc;
}

// the same through a word in global memory (agent scope): what the workgroup-scope LDS word is compared with
__global__ __launch_bounds__(512) void k_pingpong_global(unsigned long long* cyc, int n, int partner, int* gflag) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wv == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < n; i++) {
            if (lane == 0) __hip_atomic_store(gflag, 2 * i + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(gflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) != 2 * i + 2) { }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) cyc[0] = t1 - t0;
    } else if (wv == partner) {
        for (int i = 0; i < n; i++) {
            while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(gflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) != 2 * i + 1) { }
            if (lane == 0) __hip_atomic_store(gflag, 2 * i + 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
}

// s_memtime back to back (each waits for the one before), a dependent LDS read chain, and the workgroup barrier with 2 / 8 waves
__global__ __launch_bounds__(512) void k_latencies(unsigned long long* cyc, int n, unsigned* sink) {
    __shared__ unsigned chain[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) chain[i] = (i + 1) & 255;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), t = t0;
        for (int i = 0; i < n; i++) { unsigned long long u = __builtin_amdgcn_s_memtime(); asm volatile("" : "+s"(u)); t = u; }
        if (lane == 0) cyc[1] = t - t0;
        unsigned p = lane;
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < n; i++) p = chain[p];
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) cyc[2] = t1 - t0;
        if (p == 0x12345678u) sink[0] = p;
    }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[3] = t1 - t0;
}

template <typename F> static double run(F launch, unsigned long long* d_cyc, int slot, int n) {
    CK(hipMemset(d_cyc, 0, 64));
    launch();
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, d_cyc, 64, hipMemcpyDeviceToHost));
    return (double)h[slot] / n;
}

int main() {
    unsigned long long* d_cyc; unsigned* d_sink; int* d_flag;
    CK(hipMalloc(&d_cyc, 64)); CK(hipMalloc(&d_sink, 64)); CK(hipMalloc(&d_flag, 64));
    const int N = 20000;
    for (int rep = 0; rep < 2; rep++) {          // (first repetition: warm-up, clocks)
        for (int partner : {1, 4}) {
            const char* where = partner == 1 ? "another SIMD" : "the same SIMD";
            double a = run([&] { k_pingpong<0, 0><<<1, 512>>>(d_cyc, N, partner, d_sink); }, d_cyc, 0, N);
            double b = run([&] { k_pingpong<1, 0><<<1, 512>>>(d_cyc, N, partner, d_sink); }, d_cyc, 0, N);
            double c = run([&] { k_pingpong<2, 0><<<1, 512>>>(d_cyc, N, partner, d_sink); }, d_cyc, 0, N);
            double d = run([&] { k_pingpong<0, 1><<<1, 512>>>(d_cyc, N, partner, d_sink); }, d_cyc, 0, N);
            double e = run([&] { k_pingpong<2, 1><<<1, 512>>>(d_cyc, N, partner, d_sink); }, d_cyc, 0, N);
            CK(hipMemset(d_flag, 0, 64));
            double g = run([&] { k_pingpong_global<<<1, 512>>>(d_cyc, N, partner, d_flag); }, d_cyc, 0, N);
            if (rep == 1)
                printf("LDS word ping-pong, partner on %s: cycles per ROUND TRIP (two hand-overs): no pause %.0f | s_sleep 1 %.0f | s_sleep 2 %.0f | "
                       "with 288-word payload each way: no pause %.0f, s_sleep 2 %.0f | through a global-memory word (agent scope) %.0f\n", where, a, b, c, d, e, g);
        }
        CK(hipMemset(d_cyc, 0, 64));
        k_latencies<<<1, 512>>>(d_cyc, N, d_sink);
        CK(hipDeviceSynchronize());
        unsigned long long h[8];
        CK(hipMemcpy(h, d_cyc, 64, hipMemcpyDeviceToHost));
        if (rep == 1)
            printf("s_memtime back to back %.0f cycles each | dependent LDS read %.0f | __syncthreads of 8 waves %.0f\n", (double)h[1] / N, (double)h[2] / N, (double)h[3] / N);
    }
    return 0;
}
