// realistic hand-over microbench: owner does W0 cycles of LDS-heavy work between posts, helper does W1 of LDS work per request; stamps as in the library
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__device__ __forceinline__ int wg_load(const int* p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ void wg_store(int* p, int v, int lane) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wg_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
template <int SLEEP> __device__ __forceinline__ int wait_not(const int* p, int a, int b) { for (;;) { const int s = wg_load(p); if (s != a && s != b) return s; if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); } }
struct CS { int state; unsigned w[3]; unsigned long long t_post, t_reply; unsigned acc[8]; };
__device__ __forceinline__ unsigned work(unsigned* lds, unsigned x, int rounds, int lane) {       // LDS-latency-bound chain
    for (int i = 0; i < rounds; i++) { x = lds[(x + lane) & 1023] + i; x ^= lds[(x >> 3) & 1023]; }
    return x;
}
template <int SLEEP, int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int n, int w_owner, int w_helper, int busy_others, unsigned* sink) {
    __shared__ CS cs[2];
    __shared__ unsigned lds[8][1024];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8 * 1024; i += 512) (&lds[0][0])[i] = i * 2654435761u;
    if (threadIdx.x < 2) { cs[threadIdx.x].state = 0; for (int i = 0; i < 8; i++) cs[threadIdx.x].acc[i] = 0; }
    __syncthreads();
    unsigned x = threadIdx.x;
    if (wv < 2) {                                  // owners
        CS& c = cs[wv];
        for (int i = 0; i < n; i++) {
            x = work(lds[wv], x, w_owner / 4, lane);                     // "quantize"
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            if (lane == 0) c.t_post = t0;
            wg_store(&c.state, 1 | (i & 3) << 8, lane);
            x = work(lds[wv], x, w_owner, lane);                         // "calc_noise"
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            (void)wait_not<SLEEP>(&c.state, 1 | (i & 3) << 8, 1 | (i & 3) << 8);
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            wg_acquire();
            x += c.w[0] + c.w[1] + c.w[2];
            if (lane == 0) { atomicAdd(&c.acc[0], (unsigned)(t1 - t0)); atomicAdd(&c.acc[1], (unsigned)(t2 - t1)); atomicAdd(&c.acc[2], 1u); atomicAdd(&c.acc[6], (unsigned)(t2 - c.t_reply)); }
        }
        wg_store(&c.state, 9, lane);
    } else if (wv < 4) {                           // helpers
        CS& c = cs[wv - 2];
        for (;;) {
            const int s = wait_not<SLEEP>(&c.state, 0, 2);
            if (s == 9) break;
            const unsigned long long h0 = __builtin_amdgcn_s_memtime();
            wg_acquire();
            x = work(lds[wv - 2], x + s, w_helper, lane);
            if (lane == 0) { c.w[0] = x; c.w[1] = x + 1; c.w[2] = x + 2; }
            const unsigned long long h1 = __builtin_amdgcn_s_memtime();
            if (lane == 0) c.t_reply = h1;
            wg_store(&c.state, 2, lane);
            if (lane == 0) { atomicAdd(&c.acc[3], (unsigned)(h1 - h0)); atomicAdd(&c.acc[4], 1u); atomicAdd(&c.acc[5], (unsigned)(h0 - c.t_post)); }
        }
    } else if (wv < 4 + busy_others) {             // other busy waves (psyB-like: f64 chains + LDS)
        double d = lane;
        for (int i = 0; i < n * (w_owner / 8); i++) { d = d * 1.0000001 + lds[wv][(i + lane) & 1023]; }
        x += (unsigned)d;
    }
    __syncthreads();
    if (threadIdx.x < 8) out[threadIdx.x] = cs[0].acc[threadIdx.x];
    if (x == 0x12345678u) sink[0] = x;
}
int main() {
    unsigned long long* d; unsigned* sink; CK(hipMalloc(&d, 64)); CK(hipMalloc(&sink, 64));
    const int N = 3000;
    for (int rep = 0; rep < 2; rep++)
        for (int busy = 0; busy <= 2; busy += 2)
            for (int wo : {20, 40}) {
                unsigned long long h[8];
                k<2, 0><<<1, 512>>>(d, N, wo, wo, busy, sink); CK(hipDeviceSynchronize()); CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
                if (rep) printf("sleep 2, owner work %d rounds, %d other busy waves: calc %.0f | waited %.0f | helper busy %.0f | posted->seen %.0f | reply->seen %.0f\n", wo, busy,
                                (double)h[0] / h[2], (double)h[1] / h[2], (double)h[3] / h[4], (double)h[5] / h[4], (double)h[6] / h[2]);
                k<0, 0><<<1, 512>>>(d, N, wo, wo, busy, sink); CK(hipDeviceSynchronize()); CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
                if (rep) printf("no pause, owner work %d rounds, %d other busy waves: calc %.0f | waited %.0f | helper busy %.0f | posted->seen %.0f | reply->seen %.0f\n", wo, busy,
                                (double)h[0] / h[2], (double)h[1] / h[2], (double)h[3] / h[4], (double)h[5] / h[4], (double)h[6] / h[2]);
            }
    return 0;
}
