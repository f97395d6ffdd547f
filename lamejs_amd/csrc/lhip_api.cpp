// C-ABI implementation of include/lamejs_hip.h: stream objects, HBM workspace management and
// the kernel pipeline for one batch of frames.  Compiled by hipcc for gfx950 (the product), and
// by g++ with -DLHIP_HOSTSIM for the test-only CPU simulation of the kernel logic (tests/hostsim).
//
// Pipeline per batch:  load carried state -> psyA (parallel over granule-channels)
// -> scan (per stream) -> psyB (parallel) -> polyphase -> MDCT -> quantize (parallel over frames,
// speculative bin-search seed) -> validate seed chain [-> repair flagged frames]* -> bit-pack -> save state.
#include "../../include/lamejs_hip.h"
#include "lhip_defs.h"
#include "lhip_wave.h"
#include "lhip_math.h"
#include "lhip_layout.h"
#include "k_psy.h"
#include "k_fb.h"
#include "k_quant.h"
#if !defined(LHIP_HOSTSIM) || defined(LHIP_WAVESIM)
#include "k_quant_tail.h"      // how a launch of the persistent quantization kernel ends (the one-lane simulation has no workgroups: it runs kb_quant)
#endif
#include "k_bits.h"

#include <string>
#include <vector>
#include <map>
#include <mutex>
#include <thread>
#include <atomic>
#include <memory>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <chrono>

using namespace lhip;

// ===========================================================================================
// runtime shim
// ===========================================================================================
static thread_local std::string g_err;
static thread_local int64_t g_stat_frames = 0, g_stat_repaired = 0, g_stat_iters = 0;
// profiling counters of the dev builds (tests/tools/phase_prof.py, wave_tail.py); the product only allocates and zeroes them
#if defined(LHIP_PHASE_PROF) || defined(LHIP_WAVE_TIMES)
enum { PROF_BYTES = 512 + 16 * 8192 + 512 };   /* + (start, end) of every wave of the last g_quant launch (100 MHz ticks): the launch's tail; + the stage stamps of g_frame */
enum { FRAME_PROF_BASE = 64 + 2 * 8192 };      /* u64 index of g_frame's stage stamps */
#else
enum { PROF_BYTES = 512 };
#endif
struct Context;
static thread_local Context* g_stat_pending = nullptr;      // the last batch was enqueued without synchronisation: its repair statistics are still on the device
static void set_err(const std::string& e) { g_err = e; }

#ifdef LHIP_HOSTSIM
namespace rt {
// LHIP_HOSTSIM_DEVICES=n (tests): the simulation pretends to have n devices -- one Context each, so that lhip_set_devices' round-robin
// placement, lhip_stream_device and host threads batching on different contexts at the same time run in the CPU tier (ASan / TSan)
static int device_count() { static const int n = []() { const char* e = getenv("LHIP_HOSTSIM_DEVICES"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 64 ? v : 1; }(); return n; }
static bool set_device(int) { return true; }
static void* dmalloc(size_t n) { return calloc(1, n ? n : 1); }
static void dfree(void* p) { free(p); }
static bool h2d(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); return true; }
static bool d2h(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); return true; }
static bool d2d(void* d, const void* s, size_t n, void*) { memmove(d, s, n); return true; }
static bool dzero(void* d, size_t n, void*) { memset(d, 0, n); return true; }
static bool sync(void*) { return true; }
// streams and events of the chunked host path: everything is synchronous here, so ordering holds trivially
static bool stream_create(void** s) { *s = (void*)(uintptr_t)1; return true; }
static bool event_create(void** e) { *e = (void*)(uintptr_t)1; return true; }
static bool event_record(void*, void*) { return true; }
static bool stream_wait_event(void*, void*) { return true; }
static void* host_alloc_pinned(size_t n) { return calloc(1, n ? n : 1); }
static void host_free_pinned(void* p) { free(p); }
}  // namespace rt
#else
#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(std::string(#x) + ": " + hipGetErrorString(e_)); return false; } } while (0)
namespace rt {
// LHIP_ALIAS_DEVICES=n (tests only, 2 <= n <= 8): ordinals 0 .. n-1 are n SEPARATE library contexts -- own mutex, own HIP stream, own workspaces, own table
// uploads -- on physical device 0, so that a box with one GPU runs the multi-device paths (lhip_set_devices' round-robin placement, host threads batching
// on two contexts at the same time) against real HIP (tests/test_gpu_parity.py::test_gpu_two_devices_*).  Read at every call: a test sets it for its own duration.
static int alias_n() { const char* e = getenv("LHIP_ALIAS_DEVICES"); const int v = e ? atoi(e) : 0; return v >= 2 && v <= 8 ? v : 0; }
static int phys(int d) { return alias_n() ? 0 : d; }
static int device_count() { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; const int a = alias_n(); return (a && n >= 1) ? a : n; }
static bool set_device(int d) { HIPCK(hipSetDevice(phys(d))); return true; }
static void* dmalloc(size_t n) { void* p = nullptr; if (hipMalloc(&p, n ? n : 1) != hipSuccess) return nullptr; return p; }
static void dfree(void* p) { if (p) (void)hipFree(p); }
static bool h2d(void* d, const void* s, size_t n, void* st) { if (n) HIPCK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, (hipStream_t)st)); return true; }
static bool d2h(void* d, const void* s, size_t n, void* st) { if (n) HIPCK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, (hipStream_t)st)); return true; }
static bool d2d(void* d, const void* s, size_t n, void* st) { if (n) HIPCK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, (hipStream_t)st)); return true; }
static bool dzero(void* d, size_t n, void* st) { if (n) HIPCK(hipMemsetAsync(d, 0, n, (hipStream_t)st)); return true; }
static bool sync(void* st) { HIPCK(hipStreamSynchronize((hipStream_t)st)); return true; }
static bool stream_create(void** s) { hipStream_t t; HIPCK(hipStreamCreateWithFlags(&t, hipStreamNonBlocking)); *s = t; return true; }
static bool event_create(void** e) { hipEvent_t t; HIPCK(hipEventCreateWithFlags(&t, hipEventDisableTiming)); *e = t; return true; }
static bool event_record(void* e, void* st) { HIPCK(hipEventRecord((hipEvent_t)e, (hipStream_t)st)); return true; }
static bool stream_wait_event(void* st, void* e) { HIPCK(hipStreamWaitEvent((hipStream_t)st, (hipEvent_t)e, 0)); return true; }
static void* host_alloc_pinned(size_t n) { void* p = nullptr; if (hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault) != hipSuccess) return nullptr; return p; }
static void host_free_pinned(void* p) { if (p) (void)hipHostFree(p); }
}  // namespace rt
#endif

// ===========================================================================================
// device-resident per-stream state (what the reference carries from frame to frame)
// ===========================================================================================
// load carried state into the stream's carry slots and build its sample segment (tail + new samples)
// (`part` of `nparts`: the copies are dealt round-robin over the waves of a workgroup that calls this with several -- the one-frame launch)
LHIP_DEV void kb_load(const Tables& T, const Workspace& W, const StreamDesc* SD, const StreamIO* IO, int st, int lane, int part = 0, int nparts = 1) {
    const int C = T.channels_out;
    const StreamDesc sd = SD[st];
    const StreamIO io = IO[st];
    const StreamState* S = io.state;
    int job = 0;
#define LOAD_JOB() (nparts == 1 || (job++ % nparts) == part)
    for (int ch = 0; ch < C; ch++) {
        if (T.rs_ratio != 1 && LOAD_JOB()) {                     // resampling: the segment is materialised (PcmSrc::plane)
            float* seg = W.pcm + (int64_t)ch * W.pcm_plane + sd.pcm_off;
            for (int i = lane; i < io.mf_size; i += LHIP_NL) seg[i] = S->pcm_tail[ch][i];
        }
        const int64_t o = (int64_t)sd.gslot0 * C + ch;
        if (LOAD_JOB()) for (int i = lane; i < SB_STRIDE; i += LHIP_NL) W.sb[o * SB_STRIDE + i] = S->sb[ch][i];
        if (LOAD_JOB() && lane == 0) {
            W.loud[o] = S->loud[ch];
            W.tent[o] = S->tent[ch];
            W.blocktype[o] = S->last_bt[ch];
            W.seed[((int64_t)sd.fslot0 * C + ch) * 2 + 0] = S->seed[ch][0];
            W.seed[((int64_t)sd.fslot0 * C + ch) * 2 + 1] = S->seed[ch][1];
        }
    }
    const int Cp = T.psy_channels;
    for (int chn = 0; chn < Cp; chn++) {                         // psy channels: L, R and -- joint stereo -- mid, side
        const int64_t o = (int64_t)sd.gslot0 * Cp + chn;
        if (LOAD_JOB()) for (int i = lane; i < E_STRIDE; i += LHIP_NL) W.E[o * E_STRIDE + i] = S->E[chn][i];
        if (LOAD_JOB()) for (int i = lane; i < EBS_STRIDE; i += LHIP_NL) W.ecb_s[o * EBS_STRIDE + i] = S->ecb_s[chn][i];
        if (LOAD_JOB()) {
            for (int i = lane; i < PK_STRIDE; i += LHIP_NL) W.peaks[o * PK_STRIDE + i] = S->peaks[chn][i];
            if (lane == 0) W.last_attack[o] = S->last_attack[chn];
        }
        if (!T.disable_reservoir && LOAD_JOB()) for (int i = lane; i < EBL_STRIDE; i += LHIP_NL) { W.nb1[o * EBL_STRIDE + i] = S->nb1[chn][i]; W.nb2[o * EBL_STRIDE + i] = S->nb2[chn][i]; }
    }
    if (LOAD_JOB()) {
        if (Cp == 4) for (int i = lane; i < 4; i += LHIP_NL) W.tot_ener[(int64_t)sd.gslot0 * 4 + i] = S->tot_ener[i];
        if (lane == 0) { W.ath_adjust[sd.fslot0] = S->ath_adjust; W.ath_limit[sd.fslot0] = S->ath_limit; }
    }
#undef LOAD_JOB
}

// fill_buffer_resample (Lame.js:1719-1843) for an integer ratio r.  There filter_l = 32, bpc = 1, every clock value
// is an integer, the window offset is 0 and the filter index is always 1, so the reference computes a plain decimating
// FIR that does not depend on how the input was chunked:
//     out[m] = sum_{i=0..32} x[m*r + i - 16] * blackfilt[1][i]        (x[<0] = 0; f64 accumulation in tap order)
// and emits out[m] as soon as m*r + 16 < (samples received so far).  `p0` is the position of tap 0 of this call's
// first output relative to this call's first input sample; positions < 0 are the carried tail of earlier calls.
LHIP_DEV void kb_resample_elem(const Tables& T, float* dst, const int16_t* src, const float* old, int p0, int64_t t) {
    const float* coef = T.rs_blackfilt + T.rs_bpc * RS_TAPS;
    const int64_t p = (int64_t)p0 + t * T.rs_ratio;
    const bool do_scale = !(T.scale == 0.0) && !(T.scale == 1.0);
    double xvalue = 0.0;
    for (int i = 0; i < RS_TAPS; i++) {
        const int64_t q = p + i;
        float y;
        if (q < 0) y = old[(RS_TAPS - 1) + q];
        else { y = (float)src[q]; if (do_scale) y = (float)((double)y * T.scale); }
        xvalue += (double)y * (double)coef[i];
    }
    dst[t] = (float)xvalue;
}

// Resampling configurations only: the new output-rate samples of every stream, grid-stride over (stream, channel, sample).
// (Without resampling nothing is materialised: the consumers convert the caller's Int16 where they stage it, PcmSrc.)
LHIP_DEV void kb_prep_stream(const Tables& T, const Workspace& W, const StreamDesc* SD, const StreamIO* IO, int st, int64_t tid, int64_t nthreads) {
    const int C = T.channels_out;
    const StreamIO io = IO[st];
    const int64_t off = SD[st].pcm_off + io.mf_size;
    for (int ch = 0; ch < C; ch++) {
        float* dst = W.pcm + (int64_t)ch * W.pcm_plane + off;
        for (int64_t i = tid; i < io.n_new; i += nthreads) kb_resample_elem(T, dst, (ch ? io.src[1] : io.src[0]), io.state->rs_old[ch], io.rs_p0, i);
    }
}
LHIP_DEV void kb_prep(const Tables& T, const Workspace& W, const StreamDesc* SD, const StreamIO* IO, int nstreams, int64_t tid, int64_t nthreads) {
    for (int st = 0; st < nstreams; st++) kb_prep_stream(T, W, SD, IO, st, tid, nthreads);
}

LHIP_DEV void kb_save(const Tables& T, const Workspace& W, const StreamDesc* SD, const StreamIO* IO, int st, int lane, int part = 0, int nparts = 1) {
    const int C = T.channels_out;
    const StreamDesc sd = SD[st];
    const StreamIO io = IO[st];
    StreamState* S = io.state;
    const int F = sd.nframes;
    const int frame = 576 * T.mode_gr;
    const int total = io.mf_size + io.n_new, keep = total - frame * F;
    int job = 0;
#define SAVE_JOB() (nparts == 1 || (job++ % nparts) == part)
    for (int ch = 0; ch < C; ch++) {
        // new tail = segment[frame * F ...): read through the same accessor the kernels use.  In place: a chunk of 64 is read
        // completely before it is written, and later chunks only read positions above everything written so far
        if (SAVE_JOB()) {
            const PcmSrc P = pcm_source(T, W, sd, io, ch);
            for (int base = 0; base < keep; base += LHIP_NL) {
                const int i = base + lane;
                float v = 0.f;
                if (i < keep) v = pcm_at(P, frame * F + i);
                wave_sync();
                if (i < keep) S->pcm_tail[ch][i] = v;
                wave_sync();
            }
        }
        if (F == 0) continue;
        const int64_t o = (int64_t)(sd.gslot0 + T.mode_gr * F) * C + ch;
        if (SAVE_JOB()) for (int i = lane; i < SB_STRIDE; i += LHIP_NL) S->sb[ch][i] = W.sb[o * SB_STRIDE + i];
        if (SAVE_JOB() && lane == 0) {
            S->loud[ch] = W.loud[o];
            S->tent[ch] = W.tent[o];
            S->last_bt[ch] = W.blocktype[o];
            Seed s;                                               // bit reservoir: the frames were quantized in order and left their seeds in W.seed
            if (T.disable_reservoir) s = seed_before(W, sd, C, F, 0, ch);
            else { s.start = W.seed[((int64_t)(sd.fslot0 + F) * C + ch) * 2]; s.step = W.seed[((int64_t)(sd.fslot0 + F) * C + ch) * 2 + 1]; }
            S->seed[ch][0] = s.start; S->seed[ch][1] = s.step;
        }
    }
    if (F > 0) {
        const int Cp = T.psy_channels;
        for (int chn = 0; chn < Cp; chn++) {
            const int64_t o = (int64_t)(sd.gslot0 + T.mode_gr * F) * Cp + chn;
            if (SAVE_JOB()) for (int i = lane; i < E_STRIDE; i += LHIP_NL) S->E[chn][i] = W.E[o * E_STRIDE + i];
            if (SAVE_JOB()) for (int i = lane; i < EBS_STRIDE; i += LHIP_NL) S->ecb_s[chn][i] = W.ecb_s[o * EBS_STRIDE + i];
            if (SAVE_JOB()) {
                for (int i = lane; i < PK_STRIDE; i += LHIP_NL) S->peaks[chn][i] = i < 9 ? W.peaks[o * PK_STRIDE + i] : 0.f;   // 9 peaks; the pad words are never written by anybody (stale workspace bytes must not reach the state record)
                if (lane == 0) S->last_attack[chn] = W.last_attack[o];
            }
            if (!T.disable_reservoir && SAVE_JOB()) for (int i = lane; i < EBL_STRIDE; i += LHIP_NL) { S->nb1[chn][i] = W.nb1[o * EBL_STRIDE + i]; S->nb2[chn][i] = W.nb2[o * EBL_STRIDE + i]; }
        }
        if (SAVE_JOB()) {
            if (Cp == 4) for (int i = lane; i < 4; i += LHIP_NL) S->tot_ener[i] = W.tot_ener[(int64_t)(sd.gslot0 + T.mode_gr * F) * 4 + i];
            if (lane == 0) { S->ath_adjust = W.ath_adjust[sd.fslot0 + F]; S->ath_limit = W.ath_limit[sd.fslot0 + F]; }
        }
    }
    if (T.rs_ratio != 1 && SAVE_JOB()) {
        // the last 32 input samples seen so far (carried tail ++ this call's input), as the scaled floats the filter reads
        const bool do_scale = !(T.scale == 0.0) && !(T.scale == 1.0);
        for (int ch = 0; ch < C; ch++)
            for (int base = 0; base < RS_TAPS - 1; base += LHIP_NL) {
                const int i = base + lane;
                float v = 0.f;
                if (i < RS_TAPS - 1) {
                    const int64_t q = (int64_t)io.n_in - (RS_TAPS - 1) + i;
                    if (q < 0) v = S->rs_old[ch][(RS_TAPS - 1) + q];
                    else { v = (float)(ch ? io.src[1] : io.src[0])[q]; if (do_scale) v = (float)((double)v * T.scale); }
                }
                wave_sync();
                if (i < RS_TAPS - 1) S->rs_old[ch][i] = v;
                wave_sync();
            }
    }
#undef SAVE_JOB
}

// op 10: records of 2 doubles [a, b] -> [div_by_f32(a, (float)b, RN(1 / (float)b)), a / (float)b]: calc_noise's division by xmin through the reciprocal
// against the division itself (must be bit-identical for finite a >= 0 and a positive Float32 divisor)
LHIP_DEV void math_op10(const double* in, double* out) {
    const double a = in[0], b = (double)(float)in[1], rb = recip_for_div(b);
    out[0] = rb != 0.0 ? div_by_f32(a, b, rb) : a / b;
    out[1] = a / b;
}

// op 8: records of 21 doubles [istep, xa0..4, xb0..4, adj_a0..4, adj_b0..4] (f32 values) -> [0, floor(x istep) x 10, floor(x istep + adj) x 10]
LHIP_DEV void math_op8(const double* in, double* out) {
    float xa[5], xb[5], ja[5], jb[5]; int ra[5], rb[5], va[5], vb[5];
    const float istep = (float)in[0];
    for (int k = 0; k < 5; k++) { xa[k] = (float)in[1 + k]; xb[k] = (float)in[6 + k]; ja[k] = (float)in[11 + k]; jb[k] = (float)in[16 + k]; }
    q_floor_prod(xa, xb, istep, ra, rb);
    q_floor_fma(xa, xb, istep, ja, jb, va, vb);
    out[0] = 0;
    for (int k = 0; k < 5; k++) { out[1 + k] = ra[k]; out[6 + k] = rb[k]; out[11 + k] = va[k]; out[16 + k] = vb[k]; }
}

// ===========================================================================================
// One frame per stream in ONE launch (small batches: the drop-in's own 1152-sample call pattern, and every launch of the bit-reservoir
// mode).  A batch of one frame per stream is a chain of thirteen tiny kernels otherwise, each waiting for the one before it: the
// gaps between dependent launches cost as much as the frame's quantization.  Here a workgroup of FR_WAVES waves takes a stream
// through all stages, a workgroup barrier between them; every stage is the same kb_* body the separate kernels run, so the bytes
// cannot differ.  The LDS of a wave is a union of the stages' structures.
// ===========================================================================================
// Stages of the frame program (workgroup barriers in between).  Stages that do not depend on each other share a slot on different waves
// (round 5: the launch's critical path is  load | psyA | scans | psyB | quantization | bit packing;  the polyphase filterbank runs beside
// psyA, the MDCT beside psyB, the state save beside the bit packing -- profiles/r05_pass1_frame_prof_*.txt has the stage times this is
// built on).  FR_WAVES waves per workgroup whatever the channel mode.
enum { FS_LOAD, FS_PREP, FS_PSYA_POLY, FS_PSYA_MS, FS_SCAN_RAW, FS_SCAN_ATTACK, FS_SCAN_BT, FS_PSYB0_MDCT, FS_PSYB1, FS_QUANT, FS_BITS_SAVE,
       FR_STAGES, FR_WAVES = 8, FR_LDS_PER_WAVE = (sizeof(PolyLds) + 15) & ~15 };
static_assert(sizeof(PsyALds) <= FR_LDS_PER_WAVE && sizeof(PsyBLds4) <= FR_LDS_PER_WAVE && sizeof(MdctLds) <= FR_LDS_PER_WAVE &&
              sizeof(QuantLds) <= FR_LDS_PER_WAVE && sizeof(BitsLds) <= FR_LDS_PER_WAVE, "frame kernel: the per-wave LDS union is sized by PolyLds");
// a stage nobody has work in for this configuration (wave-uniform: a function of the tables and the instantiation) -- skipped with its barrier
template <int RESV> LHIP_DEV bool frame_stage_empty(int stage, const Tables& T) {
    return (stage == FS_PREP && T.rs_ratio == 1) || (stage == FS_PSYA_MS && T.psy_channels != 4) || stage == FS_PSYB1 ||
           ((stage == FS_SCAN_RAW || stage == FS_SCAN_ATTACK) && !RESV && T.mode != 1);      // (the flow of kb_frame_stage runs these two scans inside FS_PSYA_POLY)
}
// What a one-frame launch does BESIDE the search of the frame's first granule (FS_QUANT, waves 4 .. 7; no reservoir, no joint stereo): wave 4 + j holds the
// energies of (granule, channel) pair j in its LDS (kb_psyA<3>, FS_PSYA_POLY) and finishes that pair's psyA (partitions, tonality, short spreading); when both
// channels of a granule are done the first channel's wave runs the granule's psyB.  The second granule's filterbank and MDCT follow on waves that are free by then
// (two channels: 5 and 7 after their psyA parts; one channel: 6).  Meeting points are counters in LDS among the waves concerned (wg_meet; mbox[6 ..], zeroed in
// FS_PSYB0_MDCT).  The second granule's search needs psyB(granule 0) and its own MDCT: a two-channel frame's waves arrive at the workgroup barrier between the
// granules only after all of this; a one-channel frame's second granule waits for mbox[3], set here once both are done.
// LHIP_FRAME_SPLIT (experiment, round 6): everything of the one-frame kernel that is not the search as real functions -- the search loop then has the kernel's register
// allocation to itself (its scalar registers are full: code added to it is paid for in spills by all of it, DESIGN_ONE_FRAME.md)
#if defined(LHIP_FRAME_SPLIT) && !defined(LHIP_HOSTSIM)
#define LHIP_DEV_COLD static __device__ __attribute__((noinline))
#else
#define LHIP_DEV_COLD LHIP_DEV
#endif
LHIP_DEV_COLD void frame_flow_tail(const Tables& T, const PowBase& pb, const Workspace& W, const StreamDesc* SD, const StreamIO* IO, int g1, int wv, int lane,
                              unsigned char* lds, int* mbox) {
    const int C = T.channels_out, GR = T.mode_gr, np = GR * C, j = wv - 4;
    if (j >= 0 && j < np) {
        const int g = j / C;
        kb_psyA<4>(T, W, SD, IO, g1 + g, j % C, lane, *(PsyALds*)lds);
        if (C == 2) wg_meet(mbox + 6 + g, 2, lane);
        wave_sync();                                          // (the wave's LDS changes its meaning)
        if (j % C == 0) kb_psyB<4>(T, pb, W, SD, g1 + g, lane, *(PsyBLds4*)lds, -1, 0, 0);
    }
    if (GR == 2) {
        const bool fb = C == 2 ? (wv == 5 || wv == 7) : wv == 6;      // the second granule's filterbank: channel 0 on wave 5 (6), channel 1 on wave 7
        if (fb) {
            wave_sync();
            kb_poly_run(T, W, SD, IO, g1 + 1, C == 2 ? (wv - 5) / 2 : 0, 1, lane, *(PolyLds*)lds);
            if (C == 2) wg_meet(mbox + 8, 2, lane);
            wave_sync();
            if (wv != 7) kb_mdct(T, W, SD, g1 + 1, lane, *(MdctLds*)lds);
        }
        if (C == 1 && (wv == 4 || wv == 6)) {
            wg_meet(mbox + 9, 2, lane);
            if (wv == 4) wg_store(mbox + 3, 1, lane);
        }
    }
}
// stage `stage` of the frame program for wave `wv` (of `nw` >= 6) of the workgroup that owns stream `st`.  PAIRQ: stereo quantization by
// two waves (kb_quant<1>, which meets once per granule at a workgroup barrier: the other waves keep the barrier count).
template <int RESV, int PAIRQ>
LHIP_DEV void kb_frame_stage(int stage, const Tables& T, const PowBase& pb, const Workspace& W, const StreamDesc* SD, const StreamIO* IO,
                             int st, int wv, int nw, int lane, unsigned char* lds, QuantTabs& Q, int* mbox, CountShare* cshare = nullptr, CandShare* cand = nullptr, unsigned char* lds0 = nullptr) {
    if (!lds0) lds0 = lds - (size_t)wv * FR_LDS_PER_WAVE;       // (the first wave's LDS: the kernel passes the constant)
    const int C = T.channels_out, Cp = T.psy_channels, GR = T.mode_gr;
    const StreamDesc sd = SD[st];
    const bool has = sd.nframes > 0;                          // this launch completes a frame of the stream (else only the state moves)
    const int g1 = sd.gslot0 + 1, fslot = sd.fslot0 + 1;
    ResvState* rv = RESV ? &IO[st].state->rv : nullptr;       // one-frame launches work on the record in global memory
    const int side0 = nw - 2;                                 // the two waves that run the filterbank beside the psychoacoustics
    const bool psyb_late = !RESV && T.mode != 1;              // psyB beside the quantization of granule 0 (FS_QUANT) instead of in front of it
    const bool flow = PAIRQ && psyb_late && nw == 8;          // ... and with it everything else the first granule's search does not need (frame_flow_tail)
    switch (stage) {
        case FS_LOAD: kb_load(T, W, SD, IO, st, lane, wv, nw); if (wv == 0 && lane == 0) mbox[10] = 0; break;      // ([10]: FS_PSYA_POLY's meeting point)
        case FS_PREP: if (T.rs_ratio != 1) kb_prep_stream(T, W, SD, IO, st, (int64_t)wv * LHIP_NL + lane, (int64_t)nw * LHIP_NL); break;
        case FS_PSYA_POLY:
            // Without the reservoir and outside joint stereo (psyb_late) only what the search of the frame's FIRST granule needs stays in front of it:
            //   here      waves 4 + j: pair j's spectra up to the loudness (kb_psyA<3>);  waves 0 (1): granule 0's filterbank;  waves 2, 3: the high-passes
            //   FS_SCAN_* wave 0;   FS_PSYB0_MDCT  wave 1: granule 0's MDCT
            //   FS_QUANT  beside the search (frame_flow_tail): psyA's partitions / tonality (kb_psyA<4>, on the waves that hold the energies) -> psyB; the
            //             second granule's filterbank -> its MDCT, which the second granule's search waits for
            // (measured: a polyphase granule is 17 us of ONE lane's arithmetic whatever else the wave does, psyA's tail 8.7 us: 25.7 -> 17 us for this stage)
            if (flow) {
                const int np = GR * C;
                if (!has) break;
                if (wv >= 4 && wv - 4 < np) kb_psyA<3>(T, W, SD, IO, g1 + (wv - 4) / C, (wv - 4) % C, lane, *(PsyALds*)lds);
                else if (wv < C) kb_poly_run(T, W, SD, IO, g1, wv, 1, lane, *(PolyLds*)lds);
                else if (wv == 2 || wv == 3) {
                    for (int j = wv - 2; j < np; j += 2) { kb_psyA<1>(T, W, SD, IO, g1 + j / C, j % C, lane, *(PsyALds*)lds); wave_sync(); }
                    // the first two scans (attack flags from the peaks) right behind the high-passes, well inside the stage: FS_SCAN_RAW / _ATTACK are empty then
                    wg_meet(mbox + 10, 2, lane);
                    if (wv == 2) {
                        for (int g = lane; g < GR; g += LHIP_NL) kb_scan_raw(T, W, SD, g1 + g);
                        wave_sync_global();
                        for (int g = lane; g < GR; g += LHIP_NL) kb_scan_attack(T, W, SD, g1 + g);
                    }
                }
                break;
            }
            // One-frame launches on the device: waves [0, GR C) take the spectra and everything after them (kb_psyA<2>, the stage's longest chain); wave GR C + j
            // takes (granule, channel) pair j's polyphase filterbank and then its high-pass + sub-block peaks (kb_psyA<1>: a fifth of psyA, needed by the scans
            // only) -- on one wave per channel the filterbank of both granules was as long as all of psyA
            if (PAIRQ && nw >= 2 * GR * C) {
                const int np = GR * C;
                if (has && wv < np) kb_psyA<2>(T, W, SD, IO, g1 + wv / C, wv % C, lane, *(PsyALds*)lds);
                else if (has && wv < 2 * np) {
                    const int j = wv - np;
                    kb_poly_run(T, W, SD, IO, g1 + j / C, j % C, 1, lane, *(PolyLds*)lds);
                    wave_sync();                                  // (the wave's LDS changes its meaning)
                    kb_psyA<1>(T, W, SD, IO, g1 + j / C, j % C, lane, *(PsyALds*)lds);
                }
                break;
            }
            if (has && wv < GR * C) kb_psyA(T, W, SD, IO, g1 + wv / C, wv % C, lane, *(PsyALds*)lds);
            else if (has && wv >= side0 && wv - side0 < C) kb_poly_run(T, W, SD, IO, g1, wv - side0, GR, lane, *(PolyLds*)lds);
            break;
        case FS_PSYA_MS: if (has && Cp == 4 && wv < GR * 2) kb_psyA(T, W, SD, IO, g1 + wv / 2, 2 + wv % 2, lane, *(PsyALds*)lds); break;
        case FS_SCAN_RAW: if (!flow && has && wv == 0) for (int g = lane; g < GR; g += LHIP_NL) kb_scan_raw(T, W, SD, g1 + g); break;
        case FS_SCAN_ATTACK: if (!flow && has && wv == 0) for (int g = lane; g < GR; g += LHIP_NL) kb_scan_attack(T, W, SD, g1 + g); break;
        case FS_SCAN_BT:
            if (has && wv == 0) {
                for (int g = lane; g < GR; g += LHIP_NL) kb_scan_blocktype(T, W, SD, g1 + g);
                if (lane == 0) {                              // adjust_ATH of the one frame (Encoder.js:166-243)
                    double a = W.ath_adjust[sd.fslot0], l = W.ath_limit[sd.fslot0];
                    ath_step(T, ath_max_pow(T, W, sd, C, 0), a, l);
                    W.ath_adjust[fslot] = a; W.ath_limit[fslot] = l;
                }
            }
            break;
        case FS_PSYB0_MDCT:   // the MDCT needs the block types (scans) and the polyphase output.  Bit reservoir: psyB here too, the frame's granules one after
                              // the other (FS_PSYB1 takes the second) -- their thresholds depend on the reservoir; joint stereo: psyB here as well (the frame's M/S
                              // decision reads granule 0's thresholds before anything is quantized); otherwise psyB runs beside the quantization (psyb_late)
            if (flow) { if (has && wv == 1) kb_mdct(T, W, SD, g1, lane, *(MdctLds*)lds); }       // granule 0 only (waves 4 .. 7 keep psyA's energies in their LDS)
            else if (has && !psyb_late && (RESV ? wv == 0 : wv < GR)) kb_psyB<4>(T, pb, W, SD, g1 + (RESV ? 0 : wv), lane, *(PsyBLds4*)lds, -1, RESV ? rv->ResvSize : 0, RESV ? rv->ResvMax : 0);
            else if (has && wv >= side0 && wv - side0 < GR) kb_mdct(T, W, SD, g1 + (wv - side0), lane, *(MdctLds*)lds);
            if (wv == 0 && lane == 0) for (int i = 0; i < 12; i++) mbox[i] = 0;     // [3] "granule 1 may be searched" (FS_QUANT, one-channel frames); [4], [5] the bit packers' meeting points (FS_BITS_SAVE); [6 ..] frame_flow_tail's
            break;
        case FS_PSYB1: break;     // (the reservoir's second psyB runs beside the quantization now: FS_QUANT)
        case FS_QUANT:
            // waves 0 (1): the channel's search; waves 2 (3): its count helper (q_count_helper: the Huffman count of an evaluation while the owner
            // runs calc_noise); the one-lane simulation (PAIRQ == 0) has neither
            // Without the reservoir (and outside joint stereo) psyB is not on the frame's critical path: granule 0 is quantized against the thresholds the PREVIOUS call left
            // (the carry slot), granule 1 against psyB(granule 0)'s, and psyB(granule 1)'s are only saved for the next call.  Waves 4 (5) run psyB
            // while granule 0 is quantized; a two-channel frame's granules are separated by a workgroup barrier anyway (the channels exchange their
            // bits), a one-channel frame's second granule waits for mbox[3].  The one-lane simulation (PAIRQ == 0, waves one after the other) runs
            // psyB first.
            if (!PAIRQ && psyb_late && has && wv == 0) for (int g = 0; g < GR; g++) kb_psyB<4>(T, pb, W, SD, g1 + g, lane, *(PsyBLds4*)lds, -1, 0, 0);
            // Bit reservoir: psyB of the SECOND granule beside the quantization, on wave 4 -- nothing of this frame reads what it leaves (granule 1 is quantized against
            // psyB(granule 0)'s thresholds, the frame's entropies are those of the maskings in use), the next call does (kb_resv_stage, RS_QUANT: the same)
            if (RESV && GR == 2 && has && wv == 4) kb_psyB<4>(T, pb, W, SD, g1 + 1, lane, *(PsyBLds4*)lds, -1, rv->ResvSize, rv->ResvMax);
            // Candidate helpers (k_quant.h q_cand_helper, round 6; only with count helpers, nw == 8): the evaluation at the NEXT gain beside the owner's.  Two channels: waves
            // 4 + 2 c (role 0: count) and 5 + 2 c (role 1: calc_noise) for channel c, once they are through with what they do beside the first granule's search
            // (frame_flow_tail / psyB); one channel: waves 1 and 3, which have nothing else to do.  They keep the workgroup barriers' count like the count helpers.
            if (PAIRQ && C == 2) {
                unsigned char* const base = lds0;
                const bool cands = LHIP_NL != 1 && cand != nullptr && cshare != nullptr && nw == 8;
                if (has && wv < 2) {
                    kb_quant<1, RESV>(T, pb, W, SD, fslot, RESV ? 2 : 0, lane, *(QuantLds*)lds, Q, wv, mbox, rv, nullptr, cshare ? cshare + wv : nullptr, nullptr,
                                      cands ? cand : nullptr, cands ? lds0 : nullptr, (int)FR_LDS_PER_WAVE, cshare ? 2 * (int)FR_LDS_PER_WAVE : 0);
                    if (cshare) wg_store(&cshare[wv].state, CS_QUIT, lane);
#if LHIP_NL != 1
                    if (cands) q_cand_signal(cand[wv], CS_QUIT, lane);
#endif
                }
#if LHIP_NL != 1
                else if (has && cshare && wv < 4) q_count_helper(T, cshare[wv - 2], *(const QuantLds*)(lds - 2 * FR_LDS_PER_WAVE), *(QuantLds*)lds, Q, lane);
#endif
                else {
                    if (flow && has) frame_flow_tail(T, pb, W, SD, IO, g1, wv, lane, lds, mbox);
                    else if (psyb_late && has && wv >= 4 && wv - 4 < GR) kb_psyB<4>(T, pb, W, SD, g1 + (wv - 4), lane, *(PsyBLds4*)lds, -1, 0, 0);
#if LHIP_NL != 1
#ifndef LHIP_CAND_NOHELPERS
#define LHIP_CAND_NOHELPERS 0       /* 1 (experiment, only together with LHIP_CAND_MARGIN=-1000): the owner's code as with candidate helpers, but no wave plays helper */
#endif
                    if (cands && !LHIP_CAND_NOHELPERS && has && wv >= 4) { wave_sync(); q_cand_helper(T, cand[(wv - 4) >> 1], (wv - 4) & 1, *(const QuantLds*)(base + (size_t)((wv - 4) >> 1) * FR_LDS_PER_WAVE), *(QuantLds*)lds, Q, lane); }
                    else
#endif
                    for (int gr = 0; gr < GR; gr++) wg_barrier();
                }
            } else if (has && wv == 0) {
                const bool cands = LHIP_NL != 1 && PAIRQ && cand != nullptr && cshare != nullptr && nw == 8;
                kb_quant<0, RESV>(T, pb, W, SD, fslot, RESV ? 2 : 0, lane, *(QuantLds*)lds, Q, -1, nullptr, rv, nullptr, PAIRQ ? cshare : nullptr, (PAIRQ && psyb_late) ? mbox + 3 : nullptr,
                                  cands ? cand : nullptr, cands ? lds0 : nullptr, (int)FR_LDS_PER_WAVE, (PAIRQ && cshare) ? 2 * (int)FR_LDS_PER_WAVE : 0);
                if (PAIRQ && cshare) wg_store(&cshare[0].state, CS_QUIT, lane);
#if LHIP_NL != 1
                if (cands) q_cand_signal(cand[0], CS_QUIT, lane);
#endif
            }
#if LHIP_NL != 1
            else if (PAIRQ && has && cshare && wv == 2) q_count_helper(T, cshare[0], *(const QuantLds*)(lds - 2 * FR_LDS_PER_WAVE), *(QuantLds*)lds, Q, lane);
            else if (PAIRQ && !LHIP_CAND_NOHELPERS && has && cshare && cand != nullptr && nw == 8 && (wv == 1 || wv == 3))
                q_cand_helper(T, cand[0], wv >> 1, *(const QuantLds*)lds0, *(QuantLds*)lds, Q, lane);
#endif
            else if (flow && has && wv >= 4) frame_flow_tail(T, pb, W, SD, IO, g1, wv, lane, lds, mbox);      // (sets the flag granule 1 waits for)
            else if (PAIRQ && psyb_late && has && wv == 4) {           // one-channel frame: both granules' psyB on this wave, then the flag granule 1 waits for
                for (int g = 0; g < GR; g++) kb_psyB<4>(T, pb, W, SD, g1 + g, lane, *(PsyBLds4*)lds, -1, 0, 0);
                wg_store(mbox + 3, 1, lane);
            }
            break;
        case FS_BITS_SAVE:   // the state record's reservoir part belongs to the bit packer, everything else to the save: disjoint words
            if (PAIRQ && !RESV && nw > GR * C) {
                // no reservoir: a frame's granule-channels are packed side by side by waves [0, GR C) into wave 0's frame image (kb_bits_mw), the other
                // waves save the state
                const int nb = GR * C;
                uint32_t* wsh = ((BitsLds*)(lds - (size_t)wv * FR_LDS_PER_WAVE))->w;
                if (wv >= nb) kb_save(T, W, SD, IO, st, lane, wv - nb, nw - nb);
                else if (has) kb_bits_mw(T, W, SD, fslot, lane, *(BitsLds*)lds, wsh, wv, nb, mbox + 4);
                break;
            }
            if (wv == 0) { if (has) kb_bits(T, W, SD, fslot, lane, *(BitsLds*)lds, rv, W.out_bytes + st); }
            else kb_save(T, W, SD, IO, st, lane, wv - 1, nw - 1);
            break;
        default: break;
    }
}

// ===========================================================================================
// Bit reservoir: ALL frames of a stream in one launch.  With the reservoir in use a frame's bit budget and -- through pcfact -- the
// second half of its psychoacoustics depend on the bits every earlier frame spent: the frames of a stream are a
// serial chain.  What does NOT depend on the reservoir (load, resampling, psyA, the scans, the ATH recurrence, polyphase, MDCT) runs
// batched over all frames of all streams like any other batch; then ONE workgroup per stream walks the stream's frames in order:
//     psyB(granule 0) | quantization        (a workgroup barrier after each)
// with the stream's reservoir record in LDS for the whole walk, the bit packing of frame k - 1 beside psyB(granule 0) of frame k on another
// wave (the packer commits the record; psyB takes the reservoir fill from what the quantization of frame k - 1 decided, W.fr, so the two do
// not touch the same words), and psyB(granule 1) beside the quantization (only the NEXT frame reads what it leaves; round 5: it was a stage of
// its own, 11 us of a frame's 183).  No launch and no read-back per frame: the byte counts stay on the device until the call ends.
// Waves: 0 (and 1: second channel, kb_quant<1>) quantize, 2 runs psyB, 3 packs bits.
// ===========================================================================================
enum { RS_PSYB0, RS_QUANT, RS_STAGES, RS_WAVES = 4,
       RS_LDS_PER_WAVE = ((sizeof(QuantLds) > sizeof(PsyBLds4) ? (sizeof(QuantLds) > sizeof(BitsLds) ? sizeof(QuantLds) : sizeof(BitsLds))
                                                                : (sizeof(PsyBLds4) > sizeof(BitsLds) ? sizeof(PsyBLds4) : sizeof(BitsLds))) + 15) & ~15 };
// stage `stage` of frame k (of F) of stream st for wave wv; k == F: only the tail (bit packing of the last frame)
// cshare (device, wave simulation): the count helpers' records -- while a frame is quantized the psyB and the bit-packing wave have nothing to do
// and take the Huffman counts of waves 0 / 1 (q_count_helper, k_quant.h)
template <int PAIRQ>
LHIP_DEV void kb_resv_stage(int stage, const Tables& T, const PowBase& pb, const Workspace& W, const StreamDesc* SD, int st, int k, int F,
                            int wv, int lane, unsigned char* lds, QuantTabs& Q, int* mbox, ResvState& RV, int32_t* nout, CountShare* cshare = nullptr) {
    const int C = T.channels_out, GR = T.mode_gr;
    const StreamDesc sd = SD[st];
    const int g1 = sd.gslot0 + 1 + GR * k, fslot = sd.fslot0 + 1 + k, fidx = sd.out_slot0 + k;
    switch (stage) {
        case RS_PSYB0:
            if (cshare && wv < 2 && lane == 0) { cshare[wv].state = CS_IDLE; cshare[wv].here = 0; }      // (the helpers look at it one barrier from here)
            if (wv == 2 && k < F) {       // the reservoir as frame k - 1 left it: decided by that frame's quantization (the packer may still be committing it)
                const int rs = k == 0 ? RV.ResvSize : W.fr[fidx - 1].ResvSize, rm = k == 0 ? RV.ResvMax : W.fr[fidx - 1].ResvMax;
                kb_psyB<4>(T, pb, W, SD, g1, lane, *(PsyBLds4*)lds, -1, rs, rm);
            }
            if (wv == 3 && k > 0) kb_bits(T, W, SD, fslot - 1, lane, *(BitsLds*)lds, &RV, nout);
            break;
        case RS_QUANT:
            if (k >= F) break;
            // psyB of the frame's SECOND granule runs beside the quantization: nothing of frame k reads what it leaves (the psychoacoustics are one
            // granule ahead of their use: granule 1 is quantized against psyB(granule 0)'s thresholds, and the frame's entropies -- q_frame_pe -- are
            // those of the maskings in use), frame k + 1 does, two barriers from here.  Two-channel frames: on wave 2 before it turns count helper (the
            // searches post their first request after their bin searches, which take longer than this); one-channel frames: on wave 3, which has nothing else to do.
            if (GR == 2 && ((PAIRQ && wv == (C == 2 ? 2 : 3)) || (!PAIRQ && wv == 2))) {
                const int rs = k == 0 ? RV.ResvSize : W.fr[fidx - 1].ResvSize, rm = k == 0 ? RV.ResvMax : W.fr[fidx - 1].ResvMax;
                kb_psyB<4>(T, pb, W, SD, g1 + 1, lane, *(PsyBLds4*)lds, -1, rs, rm);
            }
            if (PAIRQ && C == 2) {
                if (wv < 2) { kb_quant<1, 1>(T, pb, W, SD, fslot, 2, lane, *(QuantLds*)lds, Q, wv, mbox, &RV, nullptr, cshare ? cshare + wv : nullptr, nullptr, nullptr, nullptr, 0, cshare ? 2 * (int)RS_LDS_PER_WAVE : 0); if (cshare) wg_store(&cshare[wv].state, CS_QUIT, lane); }
#if LHIP_NL != 1
                else if (cshare) q_count_helper(T, cshare[wv - 2], *(const QuantLds*)(lds - 2 * RS_LDS_PER_WAVE), *(QuantLds*)lds, Q, lane);
#endif
                else for (int gr = 0; gr < GR; gr++) wg_barrier();
            } else if (wv == 0) { kb_quant<0, 1>(T, pb, W, SD, fslot, 2, lane, *(QuantLds*)lds, Q, -1, nullptr, &RV, nullptr, PAIRQ ? cshare : nullptr, nullptr, nullptr, nullptr, 0, (PAIRQ && cshare) ? 2 * (int)RS_LDS_PER_WAVE : 0); if (PAIRQ && cshare) wg_store(&cshare[0].state, CS_QUIT, lane); }
#if LHIP_NL != 1
            else if (PAIRQ && cshare && wv == 2) q_count_helper(T, cshare[0], *(const QuantLds*)(lds - 2 * RS_LDS_PER_WAVE), *(QuantLds*)lds, Q, lane);
#endif
            break;
        default: break;
    }
}

// ===========================================================================================
// kernel launch layer
// ===========================================================================================
#ifndef LHIP_HOSTSIM
// Workgroups are handed to the 8 XCDs round-robin (workgroup b runs on XCD b % 8) and every XCD has its own L2.  Kernels whose
// neighbouring work items read the same data (a granule and its successor: overlapping PCM windows, the polyphase output that
// two MDCT granules share, the carried thresholds) number their items so that neighbours run on ONE XCD, back to back:
// item = (b % 8) * ceil(n / 8) + b / 8.  Launch XCD_GRID(n) workgroups; -1 = no item for this workgroup.
#define XCD_GRID(n) (8 * (((n) + 7) / 8))
static __device__ __forceinline__ int xcd_item(int b, int n) { const int it = (b & 7) * ((n + 7) >> 3) + (b >> 3); return it < n && (b >> 3) < ((n + 7) >> 3) ? it : -1; }
__global__ __launch_bounds__(64) void g_load(Tables T, Workspace W, const StreamDesc* SD, const StreamIO* IO) { kb_load(T, W, SD, IO, blockIdx.x, threadIdx.x); }
__global__ __launch_bounds__(64) void g_save(Tables T, Workspace W, const StreamDesc* SD, const StreamIO* IO) { kb_save(T, W, SD, IO, blockIdx.x, threadIdx.x); }
// psy channels chn0 .. chn0 + nch - 1 of every granule slot: (0, C) for L / R; joint stereo then runs (2, 2) for mid / side, which
// read what the L / R pass left in W.fht / W.hpf
// Waves per workgroup of the two psychoacoustic kernels, whose work item is one wave: four waves of consecutive items per workgroup (they share nothing but the
// launch; neighbouring items -- which read the same windows, twiddles and spreading rows, and overlapping PCM -- stay on one XCD, now on one CU and its L1).  Measured
// (round 6, profiles/r06_ab_waves_per_workgroup.txt): g_psyA 2.78 -> 2.59 ms, g_psyB 1.07 -> 1.01 ms per 1e5 two-channel frames (one channel 1.36 -> 1.28, 0.61 -> 0.58);
// 2 waves half of that, 8 slower than 1.  The filterbank kernels do not move and the bit packer loses 6 % (its waves end at very different times): they stay one wave per
// workgroup.  (Occupancy is not what changed: 4 - 5 resident waves per SIMD before and after -- SQ_WAVE_CYCLES counts in units of four clocks, calibrated on g_quant's known 4.)
#ifndef LHIP_WPB
#define LHIP_WPB 4
#endif
enum { WPB = LHIP_WPB };
#define XCD_GRID_W(n) XCD_GRID(((n) + WPB - 1) / WPB)
static __device__ __forceinline__ int xcd_wave_item(int n, int* lane) {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    *lane = (int)(threadIdx.x & 63);
    const int g = xcd_item(blockIdx.x, (n + WPB - 1) / WPB);
    const int it = g * WPB + wv;
    return (g >= 0 && it < n) ? it : -1;
}
#define WAVE_LDS(TYPE, NAME) __shared__ TYPE NAME##_[WPB]; TYPE& NAME = NAME##_[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))]
__global__ __launch_bounds__(64 * WPB) void g_psyA(Tables T, Workspace W, const StreamDesc* SD, const StreamIO* IO, int chn0, int nch) {
    WAVE_LDS(PsyALds, L);
    int lane;
    const int it = xcd_wave_item(W.ngslots * nch, &lane);
    if (it >= 0) kb_psyA(T, W, SD, IO, it / nch, chn0 + it % nch, lane, L);
}
__global__ __launch_bounds__(256) void g_prep(Tables T, Workspace W, const StreamDesc* SD, const StreamIO* IO, int nstreams) {
    kb_prep(T, W, SD, IO, nstreams, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}
__global__ __launch_bounds__(64) void g_scan_raw(Tables T, Workspace W, const StreamDesc* SD, int ngs) { const int g = blockIdx.x * 64 + threadIdx.x; if (g < ngs) kb_scan_raw(T, W, SD, g); }
__global__ __launch_bounds__(64) void g_scan_attack(Tables T, Workspace W, const StreamDesc* SD, int ngs) { const int g = blockIdx.x * 64 + threadIdx.x; if (g < ngs) kb_scan_attack(T, W, SD, g); }
__global__ __launch_bounds__(64) void g_scan_blocktype(Tables T, Workspace W, const StreamDesc* SD, int ngs) { const int g = blockIdx.x * 64 + threadIdx.x; if (g < ngs) kb_scan_blocktype(T, W, SD, g); }
__global__ __launch_bounds__(ATH_NT) void g_scan_ath(Tables T, Workspace W, const StreamDesc* SD) { __shared__ AthLds L; kb_scan_ath(T, W, SD, blockIdx.x, threadIdx.x, L); }
template <int NCH> __global__ __launch_bounds__(64 * WPB) void g_psyB(Tables T, PowBase pb, Workspace W, const StreamDesc* SD, int par) {
    WAVE_LDS(PsyBLdsT<NCH>, L);
    int lane;
    const int it = xcd_wave_item(W.ngslots, &lane);
    if (it >= 0) kb_psyB<NCH>(T, pb, W, SD, it, lane, L, par);
}
__global__ __launch_bounds__(64, 4) void g_poly(Tables T, Workspace W, const StreamDesc* SD, const StreamIO* IO, int nitems) {
    __shared__ PolyLds L;
    const int it = xcd_item(blockIdx.x, (nitems + POLY_PER_WAVE - 1) / POLY_PER_WAVE);
    if (it >= 0) kb_polyphase(T, W, SD, IO, it, nitems, threadIdx.x, L);
}
__global__ __launch_bounds__(64) void g_mdct(Tables T, Workspace W, const StreamDesc* SD) {
    __shared__ MdctLds L;
    const int it = xcd_item(blockIdx.x, W.ngslots);
    if (it >= 0) kb_mdct(T, W, SD, it, threadIdx.x, L);
}
// quantization kernels: 8 waves (= 8 frames) per workgroup share one copy of the lookup tables in LDS
#ifdef LHIP_PHASE_PROF
enum { QWAVES = 7 };      /* the profiling counters take LDS: 7 waves keep two workgroups per CU */
#else
enum { QWAVES = 8 };
#endif
// (two workgroups must fit in the 160 KB of LDS of a CU, or occupancy silently halves: static_assert in g_quant)
#ifndef LHIP_QOCC
#define LHIP_QOCC 4     /* waves per SIMD the quantization kernels are register-budgeted for */
#endif
// Persistent workgroups: each wave draws the next frame slot from a global dispenser until none is left, so a
// workgroup never idles on its slowest frame (frames differ a lot in the number of quantization rounds they need) and
// the table copy in LDS is made once per workgroup, not once per 8 frames.
LHIP_DEV int next_frame_slot(int32_t* ctr) {
    int v = 0;
    if ((threadIdx.x & 63) == 0) v = atomicAdd(ctr, 1);
    return __builtin_amdgcn_readfirstlane(v);
}
struct QArgs { Tables T; PowBase pb; Workspace W; const StreamDesc* SD; int chain, nfs, ctr; };
#ifdef LHIP_QVGPR      /* experiment builds only: cap g_quant's register budget at what 5 / 6 waves per SIMD would leave it (96 / 80).  The backend doubles an
                          "amdgpu-num-vgpr" request on gfx90a+ (unified VGPR + AGPR file) and clamps it to the range the waves-per-EU bounds imply, so the upper
                          bound must be opened too */
#define LHIP_QUANT_BOUNDS __attribute__((amdgpu_flat_work_group_size(1, 64 * QWAVES), amdgpu_waves_per_eu(LHIP_QOCC, 8), amdgpu_num_vgpr((LHIP_QVGPR) / 2)))
#else
#define LHIP_QUANT_BOUNDS __launch_bounds__(64 * QWAVES, LHIP_QOCC)
#endif
template <int RESV> __global__ LHIP_QUANT_BOUNDS void g_quant(QArgs a_unused) {
    __shared__ QuantTabs Q;
    __shared__ QuantLds L[QWAVES];
    __shared__ TailShare TS;
#ifndef LHIP_PHASE_PROF
    static_assert(QWAVES * sizeof(QuantLds) + sizeof(QuantTabs) + sizeof(TailShare) <= 80 * 1024, "g_quant: LDS budget for 2 workgroups per CU exceeded");
#endif
    const QArgs* A = (const QArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    q_copy_tabs(A->T, Q, threadIdx.x, 64 * QWAVES);
    if (threadIdx.x == 0) TS.drawing = QWAVES;
    if (threadIdx.x < QWAVES) TS.offer[threadIdx.x].state = 0;
    const bool tail_help_on = !RESV && A->T.channels_out == 2;       // a one-channel frame is one chain: nothing to offer
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef LHIP_PHASE_PROF
    L[wv].prof[threadIdx.x & 63] = 0;                 // per-wave cycle sums, flushed once at the end (a flush per frame would
#endif                                                // itself congest the memory pipeline it is trying to observe)
#if defined(LHIP_PHASE_PROF) || defined(LHIP_WAVE_TIMES)
    const unsigned long long wave_t0_ = wall_clock64();
#endif
    int hint[3] = {-1, -1, -1};                       // stream and bin-search results of this wave's previous frame (kb_quant: speculation seed)
    for (;;) {
        const int fslot = next_frame_slot(A->W.work_ctr + A->ctr);
        if (fslot >= A->nfs) break;
        // the speculative pass (chain == 0): the frame program with the second channel of every granule on offer to the workgroup's idle waves
        // (k_quant_tail.h); with the reservoir the frames are a chain and this kernel is not launched (g_resv_stream)
        if constexpr (!RESV) kb_quant_th(A->T, A->pb, A->W, A->SD, fslot, threadIdx.x & 63, L[wv], Q, hint, TS, wv);
        else kb_quant<0, RESV>(A->T, A->pb, A->W, A->SD, fslot, A->chain, threadIdx.x & 63, L[wv], Q, -1, nullptr, nullptr, nullptr);
        hint[0] = __builtin_amdgcn_readfirstlane(hint[0]); hint[1] = __builtin_amdgcn_readfirstlane(hint[1]); hint[2] = __builtin_amdgcn_readfirstlane(hint[2]);
    }
    // the dispenser is empty: stay and take the second channels that this workgroup's waves still have ahead of them
    if (tail_help_on) tail_help(A->T, A->pb, A->W, A->SD, threadIdx.x & 63, L, wv, QWAVES, Q, TS);
#ifdef LHIP_PHASE_PROF
    atomicAdd((unsigned long long*)A->W.prof + (threadIdx.x & 63), (unsigned long long)L[wv].prof[threadIdx.x & 63]);
#endif
#if defined(LHIP_PHASE_PROF) || defined(LHIP_WAVE_TIMES)
    {
        const int wid = blockIdx.x * QWAVES + wv;
        if ((threadIdx.x & 63) == 0 && wid < 8192) { A->W.prof[64 + 2 * wid] = wave_t0_; A->W.prof[64 + 2 * wid + 1] = wall_clock64(); }
    }
#endif
}
// Latency path for small stereo batches: one workgroup of two waves per frame, one wave per channel (kb_quant<1>).  A single
// frame is one wave's serially dependent search; with fewer frames than SIMDs the chip is idle anyway, so the two channels
// of a granule -- independent given the granule's bit budget -- run side by side.
template <int RESV> __global__ __launch_bounds__(128, LHIP_QOCC) void g_quant_pair(QArgs a_unused) {
    __shared__ QuantTabs Q;
    __shared__ QuantLds L[2];
    __shared__ int mbox[4];
    const QArgs* A = (const QArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    q_copy_tabs(A->T, Q, threadIdx.x, 128);
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    kb_quant<1, RESV>(A->T, A->pb, A->W, A->SD, blockIdx.x, A->chain, threadIdx.x & 63, L[wv], Q, wv, mbox);
}
__global__ __launch_bounds__(256) void g_validate_fast(Tables T, Workspace W, const StreamDesc* SD, int nfs) {
    __shared__ ValidateShare S;
    const int t = blockIdx.x * 256 + threadIdx.x, fslot = t >> 2;        // four lanes per frame slot (kb_validate_fast_quad)
    kb_validate_fast_quad(T, W, SD, fslot < nfs ? fslot : 0, t & 3, fslot < nfs, S);
}
// ---- seed-chain validation + repair without the host (persistent, grid barriers) ------------------------------------------
// One launch replaces the host's loop "validate -> read the flagged count back -> repair -> ...": every workgroup walks the same
// phases, separated by grid barriers, until a validation pass flags nothing.  Counters per iteration live in two parity slots of
// W.nflagged (zeroed for the next-but-one iteration by workgroup 0); the verdicts of all waves are the same because they read
// the counters after the barrier.  Launched cooperatively when the grid has more than one workgroup (co-residency is what a grid
// barrier needs); a single workgroup (batches of up to 64 frames) needs no cross-workgroup barrier at all.
enum { FX_NFLAG = 0, FX_NSLOW = 1, FX_WORK_REPAIR = 2, FX_WORK_SLOW = 3, FX_PARITY_STRIDE = 8, FX_BAR = 24, FX_STATS = 32 };
LHIP_DEV void grid_barrier(int32_t* bar, int nblocks) {
    __syncthreads();
    if (nblocks > 1 && threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                    // this workgroup's records / flags -> L2 and beyond
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int gen = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(bar, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // (bounded: 2^26 looks of ~0.12 us = 8 s -- two hundred times the longest launch this can stand behind; a grid that is not co-resident faults instead of hanging the device)
            for (int n = 0; __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen; n++) { if (n > (1 << 26)) __builtin_trap(); __builtin_amdgcn_s_sleep(4); }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                    // this CU's L1 must not serve stale records
    }
    __syncthreads();
}
// Two workgroups per CU at 128 registers (like g_quant) rather than one at 256: the phases of this kernel are spread over all its waves, and twice the
// waves beat the 128 B per lane the smaller budget spills (round 4, profiles/r04_pass9_ab_fixup_two_workgroups_per_cu.txt: validation + repair
// 0.32 -> 0.27 ms per 1e5 two-channel frames, `bursts` 2.2 -> 2.0 ms).  LHIP_FIXUP_OCC=2 builds the old shape.
#ifndef LHIP_FIXUP_OCC
#define LHIP_FIXUP_OCC 4
#endif
__global__ __launch_bounds__(64 * QWAVES, LHIP_FIXUP_OCC) void g_fixup(QArgs a_unused) {
    __shared__ QuantTabs Q;
    __shared__ QuantLds L[QWAVES];
    const QArgs* A = (const QArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const int nfs = A->nfs, nblocks = gridDim.x, nthr = 64 * QWAVES;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int32_t* base = A->W.nflagged;
    bool tabs = false;
    int repaired = 0, iters = 0, failed = 0;
    for (int it = 0;; it++) {
        int32_t* ctr = base + FX_PARITY_STRIDE * (it & 1);
        Workspace W = A->W;
        W.nflagged = ctr;                                                     // kb_validate(_fast) count into this iteration's slots
        if (blockIdx.x == 0 && threadIdx.x < FX_PARITY_STRIDE) base[FX_PARITY_STRIDE * ((it + 1) & 1) + threadIdx.x] = 0;   // next iteration's
        // V: memo-only replay, one thread per frame slot.  The first pass has been made by g_validate_fast, a plain launch in front
        // of this kernel (it needs no LDS, so it runs at full occupancy; the kernel boundary orders it): in the usual case -- nothing
        // flagged -- this kernel reads two counters and ends without a single grid barrier.
        // Later passes only look at the successors of the frames the last repair phase re-quantized (stamped in W.reval): a frame's
        // verdict depends on its own records and on its predecessor's, and nothing else has changed.  (On material where most
        // replays miss the memo -- `bursts`: 98 % -- a second pass over everything cost another 4 ms.)
        if (it > 0) {
            for (int f = blockIdx.x * nthr + threadIdx.x; f < nfs; f += nblocks * nthr) kb_validate_fast(A->T, W, A->SD, f, it);
            grid_barrier(base + FX_BAR, nblocks);
        }
        // (the counters are read by every lane and asserted wave-uniform: every decision below must be scalar control flow)
        const int nslow = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr + FX_NSLOW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        // Static work split over all waves of the grid (the work of these phases is rare and tiny).  NOT an atomic dispenser: a
        // `for (;;) { i = next_frame_slot(ctr); if (i >= n) break; ... }` loop nested in this iteration loop was compiled into an
        // exec-masked loop whose first-active-lane read of the dispensed index span forever on hardware (seen on ROCm 7.2, gfx950).
        const int gw = blockIdx.x * QWAVES + wv, nw = nblocks * QWAVES;
        if (nslow > 0) {                                                      // frames whose replay asked for a gain never evaluated
            if (!tabs) { q_copy_tabs(A->T, Q, threadIdx.x, nthr); __syncthreads(); tabs = true; }
            for (int i = gw; i < nslow; i += nw)
                kb_validate(A->T, A->pb, W, A->SD, __builtin_amdgcn_readfirstlane(W.slow_list[i]), lane, L[wv], Q);
            grid_barrier(base + FX_BAR, nblocks);
        }
        const int nflag = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr + FX_NFLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (nflag == 0) break;
        repaired += nflag; iters++;
        if (iters > nfs + 2) { failed = 1; break; }                           // cannot happen: every pass finalises at least the first flagged frame
        // R: re-quantize the flagged frames with the chain-implied seeds.  Wave gw owns the frame slots congruent to gw modulo the
        // number of waves (lane = every nw-th slot): flagged frames come in runs (a burst upsets the seeds of the frames after it),
        // and a frame is one wave's serial search of 1-2 ms, so a run must land on different waves -- owning 64 CONSECUTIVE slots
        // made one wave re-quantize a whole run back to back (8.7 ms for 49 frames on the `bursts` material).
        if (!tabs) { q_copy_tabs(A->T, Q, threadIdx.x, nthr); __syncthreads(); tabs = true; }
        for (int b0 = gw; b0 < nfs; b0 += 64 * nw) {
            int flagged = 0;
            const int f = b0 + nw * lane;
            if (f < nfs) {
                const StreamDesc* sd = A->SD + W.fslot_stream[f];
                const int k = f - sd->fslot0 - 1;
                if (k >= 0) flagged = W.seed_flag[sd->out_slot0 + k] == 1;
            }
            uint64_t m = __ballot(flagged);
            while (m) {
                const int l = (int)__builtin_ctzll(m);
                m &= m - 1;
                const int fr = b0 + nw * l;
                kb_quant(A->T, A->pb, W, A->SD, fr, 1, lane, L[wv], Q);
                if (lane == 0) {                                              // its successor (same stream) is what the next pass re-checks
                    const StreamDesc* sd = A->SD + W.fslot_stream[fr];
                    const int k = fr - sd->fslot0 - 1;
                    if (k + 1 < sd->nframes) W.reval[sd->out_slot0 + k + 1] = it + 1;
                }
            }
        }
        grid_barrier(base + FX_BAR, nblocks);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { base[FX_STATS + 0] = repaired; base[FX_STATS + 1] = iters; base[FX_STATS + 2] = failed; }
}
__global__ __launch_bounds__(64) void g_bits(Tables T, Workspace W, const StreamDesc* SD) {
    __shared__ BitsLds L;
    kb_bits(T, W, SD, blockIdx.x, threadIdx.x, L);
}
__global__ __launch_bounds__(64) void g_resv_flush(Tables T, Workspace W, const StreamDesc* SD) {
    __shared__ BitsLds L;
    if (SD[blockIdx.x].flush) kb_resv_flush(T, W, blockIdx.x, threadIdx.x, L, W.io[blockIdx.x].state->rv, W.out_bytes + blockIdx.x);
}
#ifndef LHIP_FRAME_PIPE
#define LHIP_FRAME_PIPE 1      /* 0: the Huffman counts of the outer loop on the searching wave itself (A/B builds) */
#endif
static constexpr bool g_frame_pipe = LHIP_FRAME_PIPE != 0;
#ifndef LHIP_FRAME_CAND
#define LHIP_FRAME_CAND 0      /* 1: candidate helpers (the evaluation at the next gain beside the owner's, k_quant.h q_cand_helper).  Built, bit-exact on the device and in the wave
                                  simulation (where it is on), and measured in round 6: an evaluation taken from the helpers saves ~ 3 k of 15.6 k cycles and the search of a
                                  two-channel frame gets 9.5 % shorter when every round posts -- but a launch whose idle waves poll as helpers instead of waiting at the workgroup
                                  barrier is 3.6 % (two channels) / 3 % (one) slower before the first request is posted, so the net is +- 0 (two channels) / - 2 .. - 4 % (one):
                                  profiles/r06_cand_*.txt, DESIGN_ONE_FRAME.md.  Off in the shipped library. */
#endif
static constexpr bool g_frame_cand = LHIP_FRAME_CAND != 0;
// the per-stream reservoir program (kb_resv_stage): one workgroup of RS_WAVES waves per stream
// (two waves per SIMD: 256 registers instead of the 264 an unbounded build takes -- the second workgroup per CU is what lets 512 streams
//  run side by side; the mode's throughput is streams in flight x one frame per 184 us)
__global__ __launch_bounds__(64 * RS_WAVES, 2) void g_resv_stream(QArgs a_unused) {
    __shared__ QuantTabs Q;
    __shared__ __attribute__((aligned(16))) unsigned char U[RS_WAVES][RS_LDS_PER_WAVE];
    __shared__ ResvState RV;
    __shared__ int mbox[4];
    __shared__ int32_t nout;
    __shared__ CountShare CS[2];
    const QArgs* A = (const QArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, st = blockIdx.x;
    q_copy_tabs(A->T, Q, threadIdx.x, 64 * RS_WAVES);
    static_assert(sizeof(ResvState) % 4 == 0, "the reservoir record is copied as words");
    ResvState* grv = &A->W.io[st].state->rv;
    for (int i = threadIdx.x; i < (int)(sizeof(ResvState) / 4); i += 64 * RS_WAVES) ((uint32_t*)&RV)[i] = ((const uint32_t*)grv)[i];
    if (threadIdx.x == 0) nout = 0;
    __syncthreads();
    const int F = __builtin_amdgcn_readfirstlane(A->SD[st].nframes);
    for (int k = 0; k <= F; k++)
        for (int stage = 0; stage < RS_STAGES; stage++) {
            kb_resv_stage<1>(stage, A->T, A->pb, A->W, A->SD, st, k, F, wv, lane, U[wv], Q, mbox, RV, &nout, (g_frame_pipe && A->ctr) ? CS : nullptr);     // A->ctr: the host's verdict on count helpers (run_batch)
            __syncthreads();
        }
    if (wv == 3 && A->SD[st].flush) kb_resv_flush(A->T, A->W, st, lane, *(BitsLds*)U[3], RV, &nout);
    __syncthreads();
    for (int i = threadIdx.x; i < (int)(sizeof(ResvState) / 4); i += 64 * RS_WAVES) ((uint32_t*)grv)[i] = ((const uint32_t*)&RV)[i];
    if (threadIdx.x == 0) A->W.out_bytes[st] = nout;
}
#if defined(LHIP_FRAME_SPLIT) && !defined(LHIP_HOSTSIM)
// every stage but the search, one copy behind a call (LHIP_FRAME_SPLIT)
template <int RESV> static __device__ __attribute__((noinline)) void kb_frame_stage_cold(int stage, const QArgs* A, const StreamIO* IO, int st, int wv, int lane, unsigned char* lds, QuantTabs& Q, int* mbox,
                                                                                        CountShare* cshare, CandShare* cand, unsigned char* lds0) {
    kb_frame_stage<RESV, 1>(stage, A->T, A->pb, A->W, A->SD, IO, st, wv, FR_WAVES, lane, lds, Q, mbox, cshare, cand, lds0);
}
#endif
// one workgroup of FR_WAVES waves per stream, one frame per stream (see kb_frame_stage)
template <int RESV> __global__ __launch_bounds__(64 * FR_WAVES) void g_frame(QArgs a_unused, const StreamIO* IO) {
    __shared__ QuantTabs Q;
    __shared__ __attribute__((aligned(16))) unsigned char U[FR_WAVES][FR_LDS_PER_WAVE];
    __shared__ int mbox[12];
    __shared__ CountShare CS[2];
#if LHIP_FRAME_CAND
    __shared__ CandShare CD[2];
#else
    CandShare* const CD = nullptr;
#endif
    const QArgs* A = (const QArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (threadIdx.x < 2) { CS[threadIdx.x].state = CS_IDLE; if (LHIP_BS_AHEAD) CS[threadIdx.x].here = 0; }
#if LHIP_FRAME_CAND
    if (threadIdx.x < 4) { CD[threadIdx.x >> 1].state[threadIdx.x & 1] = CS_IDLE; CD[threadIdx.x >> 1].present[threadIdx.x & 1] = 0; }
#if defined(LHIP_HANDOFF_PROF)
    if (threadIdx.x < 8) CD[0].acc[threadIdx.x] = 0;
#endif
#endif
#if defined(LHIP_PHASE_PROF) || defined(LHIP_HANDOFF_PROF)
    if (threadIdx.x < 8) CS[0].acc[threadIdx.x] = 0;
#endif
#ifdef LHIP_PHASE_PROF
    if (blockIdx.x == 0 && threadIdx.x == 0) { A->W.prof[FRAME_PROF_BASE + FR_STAGES + 1] = wall_clock64(); A->W.prof[FRAME_PROF_BASE + FR_STAGES + 3] = __builtin_amdgcn_s_memtime(); }
#endif
    q_copy_tabs(A->T, Q, threadIdx.x, 64 * FR_WAVES);       // (first read by the quantization stage: the barriers in between order it)
    // Unrolled: every stage runs once, and as a loop the compiler hoisted the constants and addresses of ALL stage bodies in front of it and carried them across
    // the stages -- 60 to 150 values parked in scratch memory (308 - 1264 bytes of scratch per lane as the bodies grew; past ~0.5 KB the launch itself got 20 us
    // slower: the runtime allocates scratch that large afresh per dispatch).  Unrolled, g_frame<1> needs no scratch at all.
#pragma clang loop unroll(full)
    for (int stage = 0; stage < FR_STAGES; stage++) {
#ifdef LHIP_PHASE_PROF
        // profiling build (tests/tools/frame_prof.py): when every stage of stream 0's frame starts, and the quantization phases of its wave 0
        if (blockIdx.x == 0 && threadIdx.x == 0) A->W.prof[FRAME_PROF_BASE + stage] = __builtin_amdgcn_s_memtime();
        if (stage == FS_QUANT) ((QuantLds*)U[wv])->prof[lane] = 0;
#endif
        if (frame_stage_empty<RESV>(stage, A->T)) continue;
#if defined(LHIP_FRAME_SPLIT) && !defined(LHIP_HOSTSIM)
        if (stage != FS_QUANT) kb_frame_stage_cold<RESV>(stage, A, IO, blockIdx.x, wv, lane, U[wv], Q, mbox, g_frame_pipe ? CS : nullptr, (g_frame_pipe && g_frame_cand) ? CD : nullptr, U[0]);
        else
#endif
        kb_frame_stage<RESV, 1>(stage, A->T, A->pb, A->W, A->SD, IO, blockIdx.x, wv, FR_WAVES, lane, U[wv], Q, mbox, g_frame_pipe ? CS : nullptr, (g_frame_pipe && g_frame_cand) ? CD : nullptr, U[0]);
#ifdef LHIP_PHASE_PROF
        // when each wave finished its part of the two stages whose work is dealt over waves (cycles after the stage's start)
        if ((stage == FS_PSYA_POLY || stage == FS_BITS_SAVE) && blockIdx.x == 0 && lane == 0)
            A->W.prof[FRAME_PROF_BASE + (stage == FS_PSYA_POLY ? 40 : 48) + wv] = __builtin_amdgcn_s_memtime() - A->W.prof[FRAME_PROF_BASE + stage];
        if (stage == FS_QUANT && blockIdx.x == 0 && wv == 0 && (lane < 22 || (lane > 28 && lane != 54))) A->W.prof[lane] = ((QuantLds*)U[wv])->prof[lane];      // (22 .. 28, 54: psyA's phases, PSY_FLUSH)
        if (stage == FS_QUANT && blockIdx.x == 0 && wv == 2 && lane < 5) {     // the count helper of wave 0: its five count phases (cycles, calls)
            A->W.prof[FRAME_PROF_BASE + 16 + lane] = ((QuantLds*)U[wv])->prof[PH_C_LOAD + lane]; A->W.prof[FRAME_PROF_BASE + 24 + lane] = ((QuantLds*)U[wv])->prof[32 + PH_C_LOAD + lane];
        }
        if (stage == FS_QUANT && blockIdx.x == 0 && wv == 0 && lane < 8) A->W.prof[FRAME_PROF_BASE + 32 + lane] = CS[0].acc[lane];      // the hand-over's legs
#endif
        __syncthreads();
    }
#ifdef LHIP_PHASE_PROF
    if (blockIdx.x == 0 && threadIdx.x == 0) { A->W.prof[FRAME_PROF_BASE + FR_STAGES] = __builtin_amdgcn_s_memtime(); A->W.prof[FRAME_PROF_BASE + FR_STAGES + 2] = wall_clock64(); }
#elif defined(LHIP_HANDOFF_PROF)
    // (tests/tools/handoff_prof.py: the legs of wave 0's hand-overs, summed over the calls of the process; the product never zeroes or reads these words)
    if (blockIdx.x == 0 && threadIdx.x < 8) atomicAdd(A->W.prof + 32 + threadIdx.x, (unsigned long long)CS[0].acc[threadIdx.x]);
#if LHIP_FRAME_CAND
    if (blockIdx.x == 0 && threadIdx.x < 8) atomicAdd(A->W.prof + 40 + threadIdx.x, (unsigned long long)CD[0].acc[threadIdx.x]);      // the candidate helpers of channel 0
#endif
#endif
}
// optional per-kernel timing with HIP events on the launch stream (bench.py roofline accounting)
enum { KT_LOAD, KT_PREP, KT_PSYA, KT_SCAN, KT_PSYB, KT_POLY, KT_MDCT, KT_QUANT, KT_VALIDATE, KT_REPAIR, KT_BITS, KT_SAVE, KT_N };
static const char* const g_kt_names[KT_N] = {"load", "prep", "psyA", "scan", "psyB", "polyphase", "mdct", "quant", "validate", "repair", "bits", "save"};
// The switch is process-wide (bench.py turns it on for one extra, untimed step); the events of a batch belong to the calling
// thread (a batch runs entirely inside one run_batch call), the accumulators are shared by all devices and guarded by g_kt_mu.
static std::atomic<bool> g_kt_on{false};
static std::mutex g_kt_mu;
static double g_kt_ms[KT_N];
static int64_t g_kt_calls[KT_N];
struct KtPending { int id; hipEvent_t a, b; };
static thread_local std::vector<KtPending> g_kt_pending;
static void kt_begin(int id, void* st) {
    if (!g_kt_on) return;
    KtPending p; p.id = id;
    hipEventCreate(&p.a); hipEventCreate(&p.b);
    hipEventRecord(p.a, (hipStream_t)st);
    g_kt_pending.push_back(p);
}
static void kt_end(void* st) { if (g_kt_on && !g_kt_pending.empty()) hipEventRecord(g_kt_pending.back().b, (hipStream_t)st); }
static void kt_collect() {
    std::lock_guard<std::mutex> lk(g_kt_mu);
    for (auto& p : g_kt_pending) {
        float ms = 0.f;
        hipEventSynchronize(p.b);
        hipEventElapsedTime(&ms, p.a, p.b);
        g_kt_ms[p.id] += ms; g_kt_calls[p.id]++;
        hipEventDestroy(p.a); hipEventDestroy(p.b);
    }
    g_kt_pending.clear();
}
// LAMEJS_HIP_TRACE=1: synchronise after every launch and name it on stderr (locating a kernel that does not come back)
static const bool g_trace = []() { const char* e = getenv("LAMEJS_HIP_TRACE"); return e && e[0] == '1'; }();
#define TRACE_SYNC(kern, st) do { if (g_trace) { fprintf(stderr, "[lhip] %s launched...", #kern); fflush(stderr); hipError_t t_ = hipStreamSynchronize((hipStream_t)(st)); fprintf(stderr, " %s\n", hipGetErrorString(t_)); } } while (0)
#define LAUNCHB(id, kern, nblk, nthr, st, ...) do { if ((nblk) > 0) { kt_begin(id, st); hipLaunchKernelGGL(kern, dim3(nblk), dim3(nthr), 0, (hipStream_t)(st), __VA_ARGS__); kt_end(st); TRACE_SYNC(kern, st); \
    hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { set_err(std::string(#kern) + ": " + hipGetErrorString(e_)); return false; } } } while (0)
#define LAUNCH(id, kern, nblk, st, ...) do { if ((nblk) > 0) { kt_begin(id, st); hipLaunchKernelGGL(kern, dim3(nblk), dim3(64), 0, (hipStream_t)(st), __VA_ARGS__); kt_end(st); TRACE_SYNC(kern, st); \
    hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { set_err(std::string(#kern) + ": " + hipGetErrorString(e_)); return false; } } } while (0)

// lhip_create: the quantization kernels' tables gathered once into an HBM image (q_copy_tabs)
__global__ __launch_bounds__(256) void g_build_qtabs(Tables T, QuantTabs* img) { q_load_tabs(T, *img, threadIdx.x, 256); }
__global__ void g_math(int op, const double* in, double* out, size_t n, PowBase pb) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (op == 8) { if (21 * i + 21 <= n) math_op8(in + 21 * i, out + 21 * i); return; }
    if (op == 10) { if (2 * i + 2 <= n) math_op10(in + 2 * i, out + 2 * i); return; }
    if (i >= n) return;
    const double x = in[i];
    double r = 0;
    switch (op) {
        case 0: r = v8_log10(x); break;
        case 1: r = v8_pow_base(pb, x); break;
        case 2: r = d_sqrt(x); break;
        case 3: r = 1.0 / x; break;
        case 4: r = (double)(float)x; break;
        case 5: r = (double)js_toint32(x); break;
        case 6: r = x / 3.0 + x * 0.1; break;
        case 7: r = v8_log10_pos(x); break;
        case 11: r = (double)ma_index16(x) + 1000.0 * (double)js_toint32(v8_log10_pos(x) * 16.0); break;   // mask_add's table index: shortcut (-1: none) + 1000 * the logarithm's
        case 9: {   // calc_noise's shortcut: noise_class(x) + 1000 * class from the f64 logarithm + 1e6 * class from its Float32 copy
            const double l = v8_log10_pos(x > 1E-20 ? x : 1E-20);
            r = (double)noise_class(x) + 1000.0 * (double)noise_class_of_log(l) + 1e6 * (double)noise_class_of_log((double)(float)l);
        } break;
    }
    out[i] = r;
}
#endif

// ===========================================================================================
// tables (shared between streams with identical blobs)
// ===========================================================================================
static std::atomic<int> g_spec_start{180}, g_spec_step{4};   // seed assumed by the speculative quantization pass (test hook)
struct lhtb_entry { char name[32]; uint32_t dtype, count, offset, pad; };

struct TableSet {
    Tables T;               // device pointers
    PowBase pb10;
    std::vector<uint8_t> blob;
    void* d_blob = nullptr;
    void* d_extra = nullptr;
    void* d_qtabs = nullptr;
    int device = 0;
    int base_frame_bytes = 0;
    ~TableSet() { rt::dfree(d_blob); rt::dfree(d_extra); rt::dfree(d_qtabs); }
};

static const lhtb_entry* find_entry(const uint8_t* b, const char* name) {
    uint32_t n; memcpy(&n, b + 8, 4);
    const lhtb_entry* e = (const lhtb_entry*)(b + 16);
    for (uint32_t i = 0; i < n; i++) if (strncmp(e[i].name, name, 32) == 0) return &e[i];
    return nullptr;
}

static bool build_tables(TableSet& ts, const void* blob, size_t nbytes, const lhip_config& cfg, void* stream) {
    if (nbytes < 16) { set_err("tables blob too small"); return false; }
    uint32_t magic, total;
    memcpy(&magic, blob, 4); memcpy(&total, (const uint8_t*)blob + 12, 4);
    if (magic != 0x4254484cu || total > nbytes) { set_err("tables blob: bad magic/size"); return false; }
    ts.blob.assign((const uint8_t*)blob, (const uint8_t*)blob + total);
    const uint8_t* b = ts.blob.data();
    ts.d_blob = rt::dmalloc(total);
    if (!ts.d_blob) { set_err("hipMalloc(tables) failed"); return false; }
    if (!rt::h2d(ts.d_blob, b, total, stream)) return false;
    Tables& T = ts.T;
    memset(&T, 0, sizeof T);
    bool ok = true;
    auto arr = [&](const char* name, uint32_t dtype, int* count) -> const void* {
        const lhtb_entry* e = find_entry(b, name);
        if (!e || e->dtype != dtype) { set_err(std::string("tables blob: entry missing: ") + name); ok = false; return nullptr; }
        if (count) *count = (int)e->count;
        return (const uint8_t*)ts.d_blob + e->offset;
    };
    auto host_arr = [&](const char* name) -> const void* { const lhtb_entry* e = find_entry(b, name); return e ? b + e->offset : nullptr; };
    auto named = [&](const char* names_key, const char* key) -> int {
        const lhtb_entry* e = find_entry(b, names_key);
        if (!e) { ok = false; return 0; }
        const int32_t* chars = (const int32_t*)(b + e->offset);
        std::string all;
        for (uint32_t i = 0; i < e->count && chars[i]; i++) all.push_back((char)chars[i]);
        size_t pos = 0; int idx = 0;
        while (pos <= all.size()) {
            size_t c = all.find(',', pos);
            if (c == std::string::npos) c = all.size();
            if (all.compare(pos, c - pos, key) == 0) return idx;
            idx++; pos = c + 1;
        }
        set_err(std::string("tables blob: config key missing: ") + key); ok = false; return 0;
    };
    const int32_t* ci = (const int32_t*)host_arr("cfg_i");
    const double* cd = (const double*)host_arr("cfg_d");
    if (!ci || !cd) { set_err("tables blob: cfg arrays missing"); return false; }
#define CI(f) T.f = ci[named("cfg_i_names", #f)]
#define CD(f) T.f = cd[named("cfg_d_names", #f)]
    CI(channels_out); CI(mode); CI(mode_gr); CI(version); CI(samplerate_index); CI(bitrate_index); CI(brate);
    CI(out_samplerate); CI(sideinfo_len); CI(frac_SpF); CI(noise_shaping); CI(noise_shaping_amp);
    CI(noise_shaping_stop); CI(subblock_gain); CI(use_best_huffman); CI(full_outer_loop); CI(substep_shaping);
    CI(sfb21_extra); CI(quant_comp); CI(quant_comp_short); CI(short_blocks_coupled); CI(useTemporal);
    CI(ATH_useAdjust); CI(athaa_loudapprox); CI(copyright); CI(original); CI(emphasis); CI(extension);
    CI(error_protection); CI(npart_l); CI(npart_s); CI(in_samplerate); CI(rs_filter_l); CI(rs_bpc);
    CI(disable_reservoir);
    CD(resample_ratio);
    CD(scale); CD(attackthre); CD(attackthre_s); CD(interChRatio); CD(masking_lower_long); CD(masking_lower_short);
    CD(ATH_aaSensitivityP); CD(ATH_floor); CD(decay); CD(ma_max_i1); CD(ma_max_i2); CD(ma_max_m); CD(VO_SCALE);
    CD(msfix); CD(ATHlower);
#undef CI
#undef CD
#define AF(f) T.f = (const float*)arr(#f, 2, nullptr)
#define AI(f) T.f = (const int32_t*)arr(#f, 1, nullptr)
#define AD(f) T.f = (const double*)arr(#f, 3, nullptr)
    AF(rs_blackfilt);
    AF(amp_filter); AF(ATH_l); AF(ATH_s); AF(ATH_psfb21); AF(ATH_psfb12); AF(ATH_cb_l); AF(ATH_cb_s); AF(eql_w);
    AF(pow43); AF(adj43); AF(ipow20); AF(pow20); AF(longfact); AF(shortfact); AF(rnumlines_l); AF(bo_l_weight);
    AF(bo_s_weight); AF(s3_ll); AF(s3_ss); AF(window); AF(window_s); AF(mld_l); AF(mld_s);
    AI(sfb_l); AI(sfb_s); AI(psfb21); AI(psfb12); AI(bv_scf); AI(numlines_l); AI(numlines_s); AI(bo_l); AI(bm_l);
    AI(bo_s); AI(bm_s); AI(s3ind); AI(s3ind_s); AI(fft_rv_tbl); AI(mdct_order); AI(pretab); AI(scfsi_band);
    AI(slen1_n); AI(slen2_n); AI(slen1_tab); AI(slen2_tab); AI(scale_short); AI(scale_long); AI(huf_tbl_noESC);
    AI(ht_xlen); AI(ht_linmax); AI(ht_off); AI(ht_code); AI(ht_hlen); AI(largetbl); AI(table23); AI(table56);
    AI(t32l); AI(t33l);
    T.version_bytes = (const int32_t*)arr("version_bytes", 1, &T.n_version_bytes);
    AD(fht_twiddle); AD(fht_costab); AD(enwindow); AD(mdct_win); AD(ma_tab); AD(ma_table1); AD(ma_table2);
    AD(ma_table3); AD(hpf_fircoef);
#undef AF
#undef AI
#undef AD
    if (!ok) return false;
    // MPEGMode: 0 stereo, 1 joint stereo (an extension -- the reference's Mp3Encoder never asks for it, index.js:105), 3 mono
    if (!((T.mode == 0 && T.channels_out == 2) || (T.mode == 1 && T.channels_out == 2) || (T.mode == 3 && T.channels_out == 1))) { set_err("configuration outside the supported envelope (channel mode)"); return false; }
    T.psy_channels = (T.mode == 1) ? 4 : T.channels_out;
    // ---- envelope checks: fail loudly rather than produce different bytes than the reference ----
    if (T.channels_out != (cfg.channels == 1 ? 1 : 2) || T.in_samplerate != cfg.samplerate || T.brate <= 0) { set_err("tables blob does not match the requested configuration"); return false; }
    // resampling (Lame.js:1849): only integer decimation ratios, where the reference's filter is a fixed 33-tap FIR
    T.rs_ratio = 1;
    if (T.resample_ratio < .9999 || T.resample_ratio > 1.0001) {
        const int r = T.out_samplerate > 0 ? T.in_samplerate / T.out_samplerate : 0;
        if (r < 2 || r * T.out_samplerate != T.in_samplerate || T.rs_filter_l != RS_TAPS - 1 || T.rs_bpc != 1) {
            set_err("configuration outside the supported envelope (resampling by a non-integer ratio)"); return false;
        }
        T.rs_ratio = r;
    }
    if ((T.version != 1 && T.version != 0) || T.mode_gr != (T.version == 1 ? 2 : 1) || T.quant_comp != 9 || T.quant_comp_short != 9 || T.error_protection || T.sfb21_extra ||
        T.substep_shaping != 0 || T.noise_shaping_amp > 2 || T.use_best_huffman > 1 || T.athaa_loudapprox != 2 || T.full_outer_loop != 0) {
        set_err("configuration outside the supported envelope (MPEG-1/2/2.5 CBR, quality-3 switches)"); return false;
    }
    // ---- derived index tables ----
    const int32_t* h_s3ind = (const int32_t*)host_arr("s3ind");
    const int32_t* h_s3ind_s = (const int32_t*)host_arr("s3ind_s");
    const int32_t* h_nl = (const int32_t*)host_arr("numlines_l");
    const int32_t* h_ns = (const int32_t*)host_arr("numlines_s");
    const int32_t* h_bo_l = (const int32_t*)host_arr("bo_l");
    const int32_t* h_bo_s = (const int32_t*)host_arr("bo_s");
    std::vector<int32_t> extra(4 * CBANDS, 0);
    int k = 0, j = 0;
    for (int p = 0; p < T.npart_l; p++) { extra[p] = k; k += h_s3ind[2 * p + 1] - h_s3ind[2 * p] + 1; extra[2 * CBANDS + p] = j; j += h_nl[p]; }
    if (j != HBLKSIZE) { set_err("long partitions do not cover 513 lines"); return false; }
    T.n_s3_ll = k;
    if (k > PSYB_S3_LDS) { set_err("configuration outside the supported envelope (spreading table larger than g_psyB's LDS copy)"); return false; }
    k = 0; j = 0;
    for (int p = 0; p < T.npart_s; p++) { extra[CBANDS + p] = k; k += h_s3ind_s[2 * p + 1] - h_s3ind_s[2 * p] + 1; extra[3 * CBANDS + p] = j; j += h_ns[p]; }
    if (j != HBLKSIZE_s) { set_err("short partitions do not cover 129 lines"); return false; }
    // convert_partition2scalefac walks partitions and bands together (PsyModel.js:644-734): band sb adds partitions
    // up to min(bo[sb], npart), then splits the partition it stopped at with band sb+1.  Where bo[] does not grow
    // (8 kHz short blocks) the walk stops at max(entry, bo[sb]) rather than bo[sb]; the kernel works per band from the
    // stopping points, so hand it those instead of the raw bo[] (identical wherever bo[] is strictly increasing).
    auto walk = [&](const int32_t* bo, int nb, int npart, int32_t* stop) {
        int b = 0, sb = 0;
        for (; sb < nb; ++b, ++sb) {
            const int lim = bo[sb] < npart ? bo[sb] : npart;
            if (b < lim) b = lim;
            stop[sb] = b;
            if (b >= npart) { ++sb; break; }
        }
        for (; sb < nb; ++sb) stop[sb] = npart;              // bands the walk never reaches (zero-filled by the kernel)
    };
    extra.resize(4 * CBANDS + SBMAX_l + SBMAX_s);
    walk(h_bo_l, SBMAX_l, T.npart_l, extra.data() + 4 * CBANDS);
    walk(h_bo_s, SBMAX_s, T.npart_s, extra.data() + 4 * CBANDS + SBMAX_l);
    // the polyphase band filter by output index (k_fb.h poly_slot): amp_by_out[order[band]] = amp_filter[band] where it scales at all
    const size_t amp_at = (extra.size() + 1) & ~(size_t)1;                   // 8-byte aligned
    extra.resize(amp_at + 64);
    {
        const float* h_af = (const float*)host_arr("amp_filter");
        const int32_t* h_order = (const int32_t*)host_arr("mdct_order");
        double amp[32];
        for (int i = 0; i < 32; i++) amp[i] = 1.0;
        T.amp_mask = 0;
        for (int band = 0; band < 32; band++) {
            const double af = (double)h_af[band];
            const int ob = h_order[band];
            if (ob < 0 || ob > 31) { set_err("mdct_order is not a permutation of 0..31"); return false; }
            if (!(af < 1e-12) && af < 1.0) { amp[ob] = af; T.amp_mask |= 1 << ob; }
        }
        memcpy(extra.data() + amp_at, amp, sizeof amp);
    }
    // calc_noise's systolic fold (k_quant.h): lane l owns the lines 9 l .. 9 l + 8 of the (re-ordered) spectrum; per lane, bit k =
    // line 9 l + k is the first line of its scalefactor band, bit 16 + k = it is the last one ([0..63] long, [64..127] short blocks);
    // then the widest band among bands 0 .. b: 24 entries for long blocks, 40 for short ones (band = 3 * sfb + window)
    const size_t fold_at = extra.size();
    extra.resize(fold_at + 128 + 24 + 40 + 289);
    {
        const int32_t* h_sl = (const int32_t*)host_arr("sfb_l");
        const int32_t* h_ss = (const int32_t*)host_arr("sfb_s");
        std::vector<int> band_l(576), band_s(576);
        for (int d = 0, sfb = 0; d < 576; d++) { while (sfb < SBMAX_l - 1 && h_sl[sfb + 1] <= d) sfb++; band_l[d] = sfb; }
        for (int d = 0, sfb = 0; d < 576; d++) {
            while (sfb < SBMAX_s - 1 && 3 * h_ss[sfb + 1] <= d) sfb++;
            const int st = h_ss[sfb], w = h_ss[sfb + 1] - st;
            band_s[d] = 3 * sfb + (w > 0 ? (d - 3 * st) / w : 0);
        }
        for (int sh = 0; sh < 2; sh++) {
            const std::vector<int>& b = sh ? band_s : band_l;
            for (int ln = 0; ln < 64; ln++) {
                uint32_t m = 0;
                for (int kk = 0; kk < 9; kk++) {
                    const int jj = 9 * ln + kk;
                    if (jj == 0 || b[jj - 1] != b[jj]) m |= 1u << kk;
                    if (jj == 575 || b[jj + 1] != b[jj]) m |= 1u << (16 + kk);
                }
                extra[fold_at + 64 * sh + ln] = (int32_t)m;
            }
        }
        int mx = 0;
        for (int i = 0; i < 24; i++) { if (i < SBMAX_l) { const int w = h_sl[i + 1] - h_sl[i]; if (mx < w) mx = w; } extra[fold_at + 128 + i] = mx; }
        mx = 0;
        for (int i = 0; i < 40; i++) { if (i < 3 * SBMAX_s) { const int w = h_ss[i / 3 + 1] - h_ss[i / 3]; if (mx < w) mx = w; } extra[fold_at + 128 + 24 + i] = mx; }
        // count_bits, NORM blocks (Takehiro.js:575-590): everything it derives from big_values = 2 e in one word -- the region borders
        // a1 = sfb_l[r0 + 1], a2 = sfb_l[r0 + r1 + 2] (10 bits each), the region counts r0 = bv_scf[i - 2], r1 = bv_scf[i - 1] (4 + 3 bits) and
        // PrevNoise.sfb_count1 = the band of line i - 1, plus one (5 bits)
        {
            const int32_t* h_bv = (const int32_t*)host_arr("bv_scf");
            extra[fold_at + 128 + 24 + 40] = 0;
            for (int e = 1; e <= 288; e++) {
                const int i = 2 * e, r0 = h_bv[i - 2], r1 = h_bv[i - 1];
                if (r0 < 0 || r0 > 15 || r1 < 0 || r1 > 7 || r0 + r1 + 2 > SBMAX_l) { set_err("bv_scf outside the range count_bits' region table is packed for"); return false; }
                const int a1 = h_sl[r0 + 1], a2 = h_sl[r0 + r1 + 2];
                extra[fold_at + 128 + 24 + 40 + e] = (int32_t)((uint32_t)a1 | ((uint32_t)a2 << 10) | ((uint32_t)r0 << 20) | ((uint32_t)r1 << 24) | ((uint32_t)(band_l[i - 1] + 1) << 27));
            }
        }
    }
    // psyA's partition energies as a systolic fold (k_psy.h): lane l owns the FFT lines 8 l .. 8 l + 7; per lane three words -- marks
    // (bit k: line 8 l + k is the first of its partition, bit 8 + k: the last one, counting line 512 as part of the spectrum) and
    // the partition numbers of its eight lines, a byte each
    const size_t psyfold_at = extra.size();
    extra.resize(psyfold_at + 3 * 64);
    {
        std::vector<int> part(514, 0);
        for (int p = 0, jj = 0; p < T.npart_l; p++) for (int i = 0; i < h_nl[p] && jj < 513; i++) part[jj++] = p;
        part[513] = -1;
        int mx = 0;
        for (int p = 0; p < T.npart_l; p++) if (mx < h_nl[p]) mx = h_nl[p];
        T.psy_maxlen_l = mx;
        for (int ln = 0; ln < 64; ln++) {
            uint32_t m = 0, w[2] = {0, 0};
            for (int kk = 0; kk < 8; kk++) {
                const int jj = 8 * ln + kk;
                if (jj == 0 || part[jj - 1] != part[jj]) m |= 1u << kk;
                if (part[jj + 1] != part[jj]) m |= 1u << (8 + kk);
                w[kk >> 2] |= (uint32_t)part[jj] << (8 * (kk & 3));
            }
            extra[psyfold_at + 3 * ln] = (int32_t)m; extra[psyfold_at + 3 * ln + 1] = (int32_t)w[0]; extra[psyfold_at + 3 * ln + 2] = (int32_t)w[1];
        }
        const float* h_eql = (const float*)host_arr("eql_w");
        for (int i = 0; i < BLKSIZE / 2; i++) if (!(h_eql[i] >= 0.f)) { set_err("eql_w has a negative entry (the loudness sum's error bound needs non-negative terms)"); return false; }
    }
    ts.d_extra = rt::dmalloc(extra.size() * 4);
    if (!ts.d_extra) { set_err("hipMalloc failed"); return false; }
    if (!rt::h2d(ts.d_extra, extra.data(), extra.size() * 4, stream)) return false;
    if (!rt::sync(stream)) return false;
    T.s3off_l = (const int32_t*)ts.d_extra; T.s3off_s = T.s3off_l + CBANDS; T.lineoff_l = T.s3off_l + 2 * CBANDS; T.lineoff_s = T.s3off_l + 3 * CBANDS;
    T.bo_l = T.s3off_l + 4 * CBANDS; T.bo_s = T.bo_l + SBMAX_l;
    T.amp_by_out = (const double*)(T.s3off_l + amp_at);
    T.fold_marks = T.s3off_l + fold_at; T.wpre = T.fold_marks + 128; T.bvtab = T.wpre + 64;
    T.psy_fold = T.s3off_l + psyfold_at;
    // the quantization kernels' LDS tables as one image (q_copy_tabs)
    ts.d_qtabs = rt::dmalloc(sizeof(QuantTabs));
    if (!ts.d_qtabs) { set_err("hipMalloc failed"); return false; }
    if (!rt::dzero(ts.d_qtabs, sizeof(QuantTabs), stream)) return false;          // padding bytes: a defined image
#ifdef LHIP_HOSTSIM
    q_load_tabs(T, *(QuantTabs*)ts.d_qtabs, 0, 1);
#else
    hipLaunchKernelGGL(g_build_qtabs, dim3(1), dim3(256), 0, (hipStream_t)stream, T, (QuantTabs*)ts.d_qtabs);
    { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { set_err(std::string("g_build_qtabs: ") + hipGetErrorString(e_)); return false; } }
    if (!rt::sync(stream)) return false;
#endif
    T.qtabs_img = ts.d_qtabs;
    ts.pb10 = pow_log2_parts(10.0);
    ts.base_frame_bytes = (int)((double)((T.version + 1) * 72000 * T.brate) / T.out_samplerate);
    // kb_bits assembles a frame in BitsLds (and zeroes one word past its last one): the largest frame of this configuration must fit
    if ((8 * (ts.base_frame_bytes + (T.frac_SpF != 0 ? 1 : 0)) + 31) / 32 + 1 > (int)BITS_LDS_WORDS) {
        set_err("configuration outside the supported envelope (frame larger than the bit-packing buffer)"); return false;
    }
    return true;
}

// ===========================================================================================
// per-device context: HIP stream + grow-only workspace
// ===========================================================================================
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    bool ensure(size_t n) {
        if (n <= cap) return true;
        rt::dfree(p);
        cap = n + n / 4 + 4096;
        p = rt::dmalloc(cap);
        if (!p) { cap = 0; set_err("hipMalloc(workspace) failed"); return false; }
        return true;
    }
    ~DevBuf() { rt::dfree(p); }
};

// grow-only pinned host staging (small host-buffer calls: one copy in, one copy out -- run_batch)
struct PinBuf {
    void* p = nullptr; size_t cap = 0;
    bool ensure(size_t n) {
        if (n <= cap) return true;
        rt::host_free_pinned(p);
        cap = n + n / 4 + 4096;
        p = rt::host_alloc_pinned(cap);
        if (!p) { cap = 0; return false; }
        return true;
    }
    void release() { rt::host_free_pinned(p); p = nullptr; cap = 0; }
    // (not freed by a destructor: the contexts live in a process-wide map, so that would run during static destruction, after the HIP runtime --
    //  and a profiler hooked into it -- has begun to shut down: `rocprofv3 -- python bench.py` ended in a segmentation fault at exit.  The orderly
    //  way out releases them: lhip_destroy of a context's last stream, while the runtime is certainly still up.)
};

// Everything a batch in flight owns: the workspace arrays, the descriptor / host-I/O staging, and the side stream + events of the ATH scan.
// (Round 4 measured a second set with two batches in flight -- the persistent quantization kernels side by side or one behind the other -- as
// slower than one batch at a time in every form, profiles/r04_pass5_ab_*.txt, and removed it: DESIGN.md, measured and discarded.)
struct WorkSet {
    DevBuf pcm, peaks, loud, eb_l, mask_idx, eb_s, ecb_s, att_raw, uselong, ul_tmp, last_attack, tent, prev_short, blocktype,
        ath_adjust, ath_limit, E, sb, xr, side, l3, seed, seed_flag, nflagged, slow_list, frame_bytes, desc, in16, out8, prof, fht, hpf, tot_ener, reval, att_clean, nb1, nb2, fr, out_bytes, vdig, small;
    PinBuf pin_in, pin_out;     // small host-buffer calls (run_batch): everything that travels in / out, staged once in pinned memory
    // last batch (for debug taps)
    Workspace lastW; int lastC = 0, lastCp = 0; bool have_last = false;
    // side stream for the one kernel that cannot fill the chip (the ATH recurrence: one workgroup per stream); it runs
    // beside the filterbank kernels, which do not depend on it
    void* aux_stream = nullptr; void* ev_fork = nullptr; void* ev_join = nullptr;
};

struct Context {
    int device = 0;
    void* stream = nullptr;
    std::mutex mu;
    void* ev_coop[2] = {nullptr, nullptr};      // aliased contexts: the events that order the cooperative launch (null stream) with the context's own stream
    bool own_stream = false;    // `stream` was created by the library (aliased contexts only) and is destroyed by lhip_debug_release_context
    int live_streams = 0;       // streams created on this context and not yet destroyed (guarded by mu): the last one out releases the pinned staging buffers
    std::map<std::string, std::shared_ptr<TableSet>> tables;
    WorkSet ws;
    int num_cus = 256;
    int fixup_wg_per_cu = 0;    // resident g_fixup workgroups per CU (occupancy query at the first cooperative launch)
    // large host-buffer calls (the drop-in's encodeBuffer with a long Int16Array): chunks of the call are copied in on this stream
    // while the chunk before is being encoded and the one before that is copied out (encode_host_chunked)
    void* copy_stream = nullptr; void* ev_in[2] = {nullptr, nullptr}; void* ev_done[2] = {nullptr, nullptr};
    DevBuf chunk_in, chunk_out, chunk_fx, state_bak;     // staging halves (sized for the largest chunk a call has reached so far), the per-chunk repair verdicts, the stream state a failed call gives back
    std::mutex chunk_mu;        // one chunked call at a time per device (they share the two staging halves); taken BEFORE mu, never inside it
};

static std::mutex g_ctx_mu;
static std::map<int, std::unique_ptr<Context>> g_ctx;

static Context* get_context(int device) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto it = g_ctx.find(device);
    if (it != g_ctx.end()) return it->second.get();
    std::unique_ptr<Context> c(new Context());
    c->device = device;
#ifndef LHIP_HOSTSIM
    { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, rt::phys(device)) == hipSuccess && n > 0) c->num_cus = n; }
    // an aliased context (LHIP_ALIAS_DEVICES) gets a HIP stream of its own: on the null stream two contexts of one physical device would simply queue up
    if (rt::alias_n() && device > 0 && hipSetDevice(0) == hipSuccess) { void* st = nullptr; if (rt::stream_create(&st)) { c->stream = st; c->own_stream = true; } }
#endif
    Context* r = c.get();
    g_ctx[device] = std::move(c);
    return r;
}

struct lhip_stream {
    uint32_t magic = 0x4c484950;
    Context* ctx = nullptr;
    std::shared_ptr<TableSet> ts;
    StreamState* d_state = nullptr;
    int mf_size = MF_INIT;
    int mf_samples_to_encode = 576 + 1152;
    int slot_lag = 0;
    int64_t frame_num = 0;
    int64_t rs_n_in = 0;           // resampling streams: input samples received so far
    ~lhip_stream() { rt::dfree(d_state); magic = 0; }
};

// ===========================================================================================
// batch encode
// ===========================================================================================
struct Job {
    lhip_stream* s; const int16_t* l; const int16_t* r; size_t n; uint8_t* out; size_t cap; int64_t written;
    int F; int64_t bytes;
    int64_t n_out;              // samples this call appends to the encoder's buffer (== n unless resampling)
    bool flush = false;         // bit reservoir: the stream ends with this call (its bitstream is padded to the end of the last frame)
};

// resampling by the integer ratio r: output sample m exists once m*r + 16 < (input samples received) -- see kb_resample_elem
static int64_t rs_outputs(int64_t n_in_total, int r) { return n_in_total > 16 ? (n_in_total - 16 + r - 1) / r : 0; }

static int64_t batch_bytes(const TableSet& ts, int slot_lag, int F) {
    int64_t npad = 0;
    const Tables& T = ts.T;
    if (T.frac_SpF != 0 && F > 0) {
        const int64_t sr = T.out_samplerate;
        int64_t m0 = slot_lag % sr; if (m0 < 0) m0 += sr;
        const int64_t need = (int64_t)F * T.frac_SpF - m0;
        npad = need > 0 ? (need + sr - 1) / sr : 0;
    }
    return (int64_t)F * ts.base_frame_bytes + npad;
}

#ifdef LHIP_PHASE_PROF
// profiling build: where the host side of a batch spends its time (seconds, summed; [7] = batches) -- lhip_debug_read(9)
static double g_call_prof[8];
#define CALL_STAMP(i) do { const auto n_ = std::chrono::steady_clock::now(); g_call_prof[i] += std::chrono::duration<double>(n_ - cp_t_).count(); cp_t_ = n_; } while (0)
#else
#define CALL_STAMP(i) do {} while (0)
#endif
enum { FX_STATS_OFF = 32 * 4 };      // byte offset of the repair statistics inside the counter block (FX_STATS of g_fixup)
#ifndef LHIP_HOSTSIM
static inline bool g_kt_on_() { return g_kt_on; }
static_assert(FX_STATS_OFF == FX_STATS * 4, "counter block layout");
#else
static inline bool g_kt_on_() { return false; }
#endif
// fx_dst (device, optional): where this batch's repair verdict (three words: repaired frames, iterations, "did not converge") is copied on the launch
// stream while ctx->mu is still held -- the chunked host path logs one per unit, and another thread's batch on the same device must not get in between
static bool run_batch(Context* ctx, std::vector<Job>& jobs, bool dev_io, bool want_sync, int32_t* fx_dst = nullptr) {
    if (jobs.empty()) return true;
#ifdef LHIP_PHASE_PROF
    auto cp_t_ = std::chrono::steady_clock::now();
    g_call_prof[7] += 1;
#endif
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!rt::set_device(ctx->device)) return false;
    WorkSet& ws = ctx->ws;
    void* st = ctx->stream;
    TableSet& ts = *jobs[0].s->ts;
    const Tables& T = ts.T;
    const int C = T.channels_out;
    const int GR = T.mode_gr, frame = 576 * GR, mf_needed = 1024 + frame - 272;   // calcNeeded (Lame.js:1517-1530)
    const int S = (int)jobs.size();
    const bool resv = !T.disable_reservoir;       // bit reservoir (extension): the frames of a stream are a serial chain (g_resv_stream), output sizes known to the device only
    // ---- plan ----
    std::vector<StreamDesc> sd(S);
    std::vector<StreamIO> io(S);
    int nfs = 0, ngs = 0, nfr = 0, maxF = 0;
    int64_t pcm_plane = 0, in_total = 0, out_total = 0;
    for (int i = 0; i < S; i++) {
        Job& j = jobs[i];
        lhip_stream* s = j.s;
        if (s->ts.get() != &ts) { set_err("batch: all streams must share one configuration"); return false; }
        j.n_out = T.rs_ratio == 1 ? (int64_t)j.n : rs_outputs(s->rs_n_in + (int64_t)j.n, T.rs_ratio) - rs_outputs(s->rs_n_in, T.rs_ratio);
        const int64_t total = (int64_t)s->mf_size + j.n_out;
        if (total > 0x7fffffff || (int64_t)j.n > 0x7fffffff) { set_err("too many samples in one call"); return false; }
        j.F = total >= mf_needed ? (int)((total - mf_needed) / frame) + 1 : 0;
        j.bytes = batch_bytes(ts, s->slot_lag, j.F);
        if (resv)     // upper bound (the real count comes back from the device): the call's own frames, what earlier frames left in the
                      // reservoir (main data up to 511 bytes ahead of its header) and in the header queue, the flush padding
            j.bytes = (int64_t)j.F * (ts.base_frame_bytes + 1) + (j.F > 0 || j.flush ? 512 + RESV_HQ * RESV_HDR : 0) + (j.flush ? 1440 + RESV_HQ * RESV_HDR : 0);
        if ((size_t)j.bytes > j.cap) { j.written = LHIP_ERR_BUFFER_TOO_SMALL; set_err("output buffer too small"); return false; }
        StreamDesc& d = sd[i];
        memset(&d, 0, sizeof d);
        d.nframes = j.F; d.fslot0 = nfs; d.gslot0 = ngs; d.out_slot0 = nfr;
        d.pcm_off = pcm_plane; d.out_off = out_total; d.seg_len = (int)total; d.first_call = s->frame_num == 0;
        d.slot_lag = s->slot_lag; d.frame_num0 = s->frame_num; d.flush = (resv && j.flush) ? 1 : 0;
        nfs += j.F + 1; ngs += GR * j.F + 1; nfr += j.F;
        if (j.F > maxF) maxF = j.F;
        pcm_plane += (total + 63) & ~(int64_t)63;
        in_total += (int64_t)j.n; out_total += (j.bytes + 15) & ~(int64_t)15;
    }
    // ---- workspace ----
    Workspace W;
    memset(&W, 0, sizeof W);
    W.spec_start = g_spec_start; W.spec_step = g_spec_step; W.mode_gr = T.mode_gr;
    W.nstreams = S; W.nframes_total = nfr; W.nfslots = nfs; W.ngslots = ngs; W.pcm_plane = pcm_plane;
    const int Cp = T.psy_channels;
    const size_t GC = (size_t)ngs * C, GP = (size_t)ngs * Cp, FR = (size_t)(nfr > 0 ? nfr : 1);
#define ENS(buf, bytes) if (!ws.buf.ensure(bytes)) return false
    ENS(pcm, T.rs_ratio != 1 ? (size_t)pcm_plane * C * 4 + 64 : 64);
    ENS(peaks, GP * PK_STRIDE * 4); ENS(loud, GC * 4); ENS(eb_l, GP * EBL_STRIDE * 4); ENS(mask_idx, GP * EBL_STRIDE * 4);
    ENS(eb_s, GP * EBS_STRIDE * 4); ENS(ecb_s, GP * EBS_STRIDE * 4); ENS(att_raw, GP * 4); ENS(uselong, GC * 4); ENS(ul_tmp, GP * 4); ENS(last_attack, GP * 4);
    ENS(tent, GC * 4); ENS(prev_short, GC * 4); ENS(blocktype, GC * 4); ENS(ath_adjust, (size_t)nfs * 8);
    ENS(ath_limit, (size_t)nfs * 8); ENS(E, GP * E_STRIDE * 4); ENS(sb, GC * SB_STRIDE * 4); ENS(xr, GC * 576 * 4);
    ENS(att_clean, GP * 4); ENS(nb1, GP * EBL_STRIDE * 4); ENS(nb2, GP * EBL_STRIDE * 4); ENS(fr, FR * sizeof(FrameResv)); ENS(out_bytes, (size_t)S * 4 + 64);
    ENS(fht, Cp == 4 ? (size_t)ngs * 2 * FHT_STRIDE * 4 : 64); ENS(hpf, Cp == 4 ? (size_t)ngs * 2 * 576 * 4 : 64); ENS(tot_ener, (size_t)ngs * 4 * 4);
    ENS(side, FR * 2 * C * sizeof(GrSide)); ENS(l3, FR * 2 * C * 576 * 2); ENS(seed, (size_t)nfs * C * 2 * 4);
    ENS(vdig, FR * 2 * C * VD_WORDS * 4);
    ENS(seed_flag, FR * 4); ENS(reval, FR * 4); ENS(nflagged, 256); ENS(slow_list, (size_t)nfs * 4); ENS(frame_bytes, FR * 4);
    ENS(prof, PROF_BYTES);
    // Small host-buffer calls (the drop-in's own 1152-sample call pattern, small encodeBatch calls): everything that travels is laid out as ONE
    // device block   [ output bytes | counters (nflagged, out_bytes) | seed_flag | reval | descriptors | Int16 input ]
    // mirrored in pinned host memory, so that a call is ONE copy in (counters arrive as the zeros they must start from), the kernels, ONE copy
    // out (bytes + counters) and one synchronisation -- instead of three pageable copies in, five memsets and two to three copies out, each of
    // which is a stream operation the frame's single launch waits behind (profiles/r05_*frame_prof*.txt).
    enum { SMALL_CALL_BYTES = 1 << 20 };
    auto a16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t desc_sd = 0, desc_io = a16(desc_sd + (size_t)S * sizeof(StreamDesc)), desc_fm = a16(desc_io + (size_t)S * sizeof(StreamIO)),
                 desc_gm = a16(desc_fm + (size_t)nfs * 4), desc_bytes = desc_gm + (size_t)ngs * 4;
    const size_t in_bytes = (size_t)in_total * 2 * C;
    const size_t sm_out = 0, sm_nfl = a16((size_t)out_total + 64), sm_ob = sm_nfl + 256, sm_sf = a16(sm_ob + (size_t)S * 4 + 64), sm_rv = sm_sf + a16(FR * 4),
                 sm_desc = sm_rv + a16(FR * 4), sm_in = a16(sm_desc + desc_bytes), sm_end = sm_in + in_bytes + 64;
    static const bool no_small = []() { const char* e = getenv("LAMEJS_HIP_NO_SMALL_CALLS"); return e && e[0] == '1'; }();
    bool small = !dev_io && !no_small && sm_end <= SMALL_CALL_BYTES;
    if (small && !(ws.pin_in.ensure(sm_end - sm_nfl) && ws.pin_out.ensure(sm_sf))) small = false;       // no pinned memory: the general path
    if (small) { ENS(small, sm_end); }
    else if (!dev_io) { ENS(in16, in_bytes + 64); ENS(out8, (size_t)out_total + 64); }
#undef ENS
    W.pcm = (float*)ws.pcm.p;
    W.peaks = (float*)ws.peaks.p; W.loud = (float*)ws.loud.p; W.eb_l = (float*)ws.eb_l.p; W.mask_idx = (int32_t*)ws.mask_idx.p;
    W.eb_s = (float*)ws.eb_s.p; W.ecb_s = (float*)ws.ecb_s.p; W.att_raw = (int32_t*)ws.att_raw.p; W.uselong = (int32_t*)ws.uselong.p; W.ul_tmp = (int32_t*)ws.ul_tmp.p;
    W.last_attack = (int32_t*)ws.last_attack.p; W.tent = (int32_t*)ws.tent.p; W.prev_short = (int32_t*)ws.prev_short.p;
    W.blocktype = (int32_t*)ws.blocktype.p; W.ath_adjust = (double*)ws.ath_adjust.p; W.ath_limit = (double*)ws.ath_limit.p;
    W.E = (float*)ws.E.p; W.sb = (float*)ws.sb.p; W.xr = (float*)ws.xr.p; W.side = (GrSide*)ws.side.p;
    W.fht = (float*)ws.fht.p; W.hpf = (float*)ws.hpf.p; W.tot_ener = (float*)ws.tot_ener.p;
    W.att_clean = (int32_t*)ws.att_clean.p; W.nb1 = (float*)ws.nb1.p; W.nb2 = (float*)ws.nb2.p; W.fr = (FrameResv*)ws.fr.p; W.out_bytes = (int32_t*)ws.out_bytes.p;
    W.l3 = (int16_t*)ws.l3.p; W.seed = (int32_t*)ws.seed.p; W.seed_flag = (int32_t*)ws.seed_flag.p; W.reval = (int32_t*)ws.reval.p;
    W.vdig = (uint32_t*)ws.vdig.p; W.vdig_n = (int64_t)FR * 2 * C;
    W.nflagged = (int32_t*)ws.nflagged.p; W.work_ctr = (int32_t*)ws.nflagged.p + 16; W.slow_list = (int32_t*)ws.slow_list.p; W.frame_bytes = (int32_t*)ws.frame_bytes.p; W.out = nullptr; W.prof = (unsigned long long*)ws.prof.p;
    uint8_t* const smb = (uint8_t*)ws.small.p;
    uint8_t* const desc_dev = small ? smb + sm_desc : nullptr;        // (the general path sizes ws.desc below)
    if (small) {
        W.nflagged = (int32_t*)(smb + sm_nfl); W.work_ctr = W.nflagged + 16; W.out_bytes = (int32_t*)(smb + sm_ob);
        W.seed_flag = (int32_t*)(smb + sm_sf); W.reval = (int32_t*)(smb + sm_rv);
    }

    CALL_STAMP(0);                                  // plan + workspace
    // ---- descriptors / inputs ----
    std::vector<int32_t> fmap(nfs), gmap(ngs);
    std::vector<int64_t> out_rel(S);              // a stream's output offset inside the output area (sd.out_off becomes the absolute address below)
    int64_t in_off = 0;
    uint8_t* const pin = small ? (uint8_t*)ws.pin_in.p : nullptr;       // mirrors the device block from sm_nfl on
    if (small) memset(pin, 0, sm_desc - sm_nfl);                       // the counters start from zero
    for (int i = 0; i < S; i++) {
        Job& j = jobs[i];
        for (int k = 0; k <= j.F; k++) fmap[sd[i].fslot0 + k] = i;
        for (int k = 0; k <= GR * j.F; k++) gmap[sd[i].gslot0 + k] = i;
        StreamIO& o = io[i];
        o.state = j.s->d_state; o.n_new = (int)j.n_out; o.mf_size = j.s->mf_size; o.n_in = (int)j.n;
        o.rs_p0 = T.rs_ratio == 1 ? 0 : (int)(rs_outputs(j.s->rs_n_in, T.rs_ratio) * T.rs_ratio - 16 - j.s->rs_n_in);
        out_rel[i] = sd[i].out_off;
        if (dev_io) {
            o.src[0] = j.l; o.src[1] = (C == 2 && j.r) ? j.r : j.l; o.out = j.out;
        } else {
            int16_t* base = small ? (int16_t*)(smb + sm_in) : (int16_t*)ws.in16.p;
            int16_t* hbase = small ? (int16_t*)(pin + (sm_in - sm_nfl)) : nullptr;
            o.src[0] = base + in_off;
            if (small) memcpy(hbase + in_off, j.l, j.n * 2);
            else if (!rt::h2d((void*)o.src[0], j.l, j.n * 2, st)) return false;
            in_off += (int64_t)j.n;
            if (C == 2) {
                o.src[1] = base + in_off;
                if (small) memcpy(hbase + in_off, j.r ? j.r : j.l, j.n * 2);
                else if (!rt::h2d((void*)o.src[1], j.r ? j.r : j.l, j.n * 2, st)) return false;
                in_off += (int64_t)j.n;
            } else o.src[1] = o.src[0];
            o.out = (small ? smb + sm_out : (uint8_t*)ws.out8.p) + sd[i].out_off;
        }
    }
    // All descriptors travel in ONE host-to-device copy (a small pageable copy costs ~10 us of host time each, and a 1-frame
    // call is only ~0.4 ms long): [StreamDesc x S | StreamIO x S | frame-slot map | granule-slot map], 16-byte aligned parts.
    // The out pointer per stream is carried in StreamIO; kb_bits reads W.out + sd.out_off, so W.out is a zero base and
    // out_off holds the absolute address (the device address space is 64-bit).
    if (!small && !ws.desc.ensure(desc_bytes)) return false;
    uint8_t* const ddesc = small ? desc_dev : (uint8_t*)ws.desc.p;
    {
        std::vector<uint8_t> stage_v;
        uint8_t* stage = nullptr;
        if (small) stage = pin + (sm_desc - sm_nfl);
        else { stage_v.assign(desc_bytes, 0); stage = stage_v.data(); }
        if (small) memset(stage, 0, sm_in - sm_desc);
        for (int i = 0; i < S; i++) sd[i].out_off = (int64_t)(uintptr_t)io[i].out;
        memcpy(stage + desc_sd, sd.data(), (size_t)S * sizeof(StreamDesc));
        memcpy(stage + desc_io, io.data(), (size_t)S * sizeof(StreamIO));
        memcpy(stage + desc_fm, fmap.data(), (size_t)nfs * 4);
        memcpy(stage + desc_gm, gmap.data(), (size_t)ngs * 4);
        if (small) { if (!rt::h2d(smb + sm_nfl, pin, sm_end - 64 - sm_nfl, st)) return false; }      // counters (zeros) + descriptors + input: one copy from pinned memory
        else if (!rt::h2d(ddesc, stage, desc_bytes, st)) return false;
    }
    W.fslot_stream = (const int32_t*)(ddesc + desc_fm); W.gslot_stream = (const int32_t*)(ddesc + desc_gm);
    if (!small) {
        if (!rt::dzero(ws.seed_flag.p, FR * 4, st)) return false;
        if (!rt::dzero(ws.reval.p, FR * 4, st)) return false;
        if (resv && !rt::dzero(ws.out_bytes.p, (size_t)S * 4, st)) return false;
        if (!rt::dzero(ws.nflagged.p, 256, st)) return false;
    }
#if defined(LHIP_PHASE_PROF) || defined(LHIP_WAVE_TIMES)
    if (!rt::dzero(ws.prof.p, PROF_BYTES, st)) return false;      // (the product never reads these counters)
#endif
    const StreamDesc* dSD = (const StreamDesc*)(ddesc + desc_sd);
    const StreamIO* dIO = (const StreamIO*)(ddesc + desc_io);
    W.io = dIO;
    CALL_STAMP(1);                                  // input copies, descriptors, counters zeroed: enqueued

    int64_t repaired = 0, iters = 0;
    // at most one frame per stream: the whole frame program in one launch (kb_frame_stage); LAMEJS_HIP_NO_FRAME_KERNEL=1 keeps the separate kernels
    static const bool no_frame = []() { const char* e = getenv("LAMEJS_HIP_NO_FRAME_KERNEL"); return e && e[0] == '1'; }();
    // (one workgroup per stream at one wave per SIMD: beyond one stream per CU the separate kernels, each at full occupancy, are faster --
    //  bit-reservoir mode over 2048 streams: 0.7 M frames/s with this kernel, measured; the launch set of the separate kernels costs ~0.5 ms whatever S)
    const bool use_frame = maxF <= 1 && !no_frame && S <= ctx->num_cus;
#ifdef LHIP_HOSTSIM
    {
        // WAVE_RUN: one wave of a kernel body.  Scalar simulation: the body runs once with lane 0 (NL = 1); wave simulation
        // (-DLHIP_WAVESIM): its 64 lanes run as fibers and meet at every wave primitive (lhip_wave.h).
#ifdef LHIP_WAVESIM
#define WAVE_RUN(...) wsim::run([&](int lane_) { __VA_ARGS__; })
#else
#define WAVE_RUN(...) do { const int lane_ = 0; __VA_ARGS__; } while (0)
#endif
        // (thread_local: with LHIP_HOSTSIM_DEVICES > 1 host threads batch on different contexts at the same time)
        static thread_local PsyALds LA; static thread_local PsyBLds4 LB; static thread_local MdctLds LM; static thread_local PolyLds LP; static thread_local QuantLds LQ; static thread_local BitsLds LBi; static thread_local QuantTabs QT;
        q_load_tabs(T, QT, 0, 1);
        if (use_frame) {
            // the one-frame-per-stream program (kb_frame_stage), stage by stage; the wave simulation runs it as a real workgroup
            const int NW = FR_WAVES;
            alignas(16) static thread_local unsigned char UL[FR_WAVES][FR_LDS_PER_WAVE]; static thread_local int fmbox[12]; static thread_local CountShare fcs[2]; static thread_local CandShare fcd[2];
            static const bool sim_cand = []() { const char* e = getenv("LAMEJS_SIM_NO_CAND"); return !(e && e[0] == '1'); }();
            for (int s = 0; s < S; s++) {
                fcs[0].state = CS_IDLE; fcs[1].state = CS_IDLE; fcs[0].here = fcs[1].here = 0;      // per workgroup, as g_frame does (a stream's owners leave CS_QUIT behind)
                for (int c = 0; c < 2; c++) for (int r = 0; r < 2; r++) { fcd[c].state[r] = CS_IDLE; fcd[c].present[r] = 0; }
#ifdef LHIP_WAVESIM
                wsim::run_block(NW, [&](int wave_, int lane_) {
                    for (int stage = 0; stage < FR_STAGES; stage++) {
                        if (resv ? frame_stage_empty<1>(stage, T) : frame_stage_empty<0>(stage, T)) continue;
                        if (resv) kb_frame_stage<1, 1>(stage, T, ts.pb10, W, dSD, dIO, s, wave_, NW, lane_, UL[wave_], QT, fmbox, fcs, sim_cand ? fcd : nullptr);
                        else kb_frame_stage<0, 1>(stage, T, ts.pb10, W, dSD, dIO, s, wave_, NW, lane_, UL[wave_], QT, fmbox, fcs, sim_cand ? fcd : nullptr);
                        wg_barrier();
                    }
                });
#else
                for (int stage = 0; stage < FR_STAGES; stage++)
                    for (int wv = 0; wv < NW; wv++) {
                        if (resv) kb_frame_stage<1, 0>(stage, T, ts.pb10, W, dSD, dIO, s, wv, NW, 0, UL[wv], QT, fmbox);
                        else kb_frame_stage<0, 0>(stage, T, ts.pb10, W, dSD, dIO, s, wv, NW, 0, UL[wv], QT, fmbox);
                    }
#endif
            }
            if (resv) for (int s = 0; s < S; s++) if (sd[s].flush) WAVE_RUN(kb_resv_flush(T, W, s, lane_, LBi, dIO[s].state->rv, W.out_bytes + s));
        } else {
        for (int s = 0; s < S; s++) WAVE_RUN(kb_load(T, W, dSD, dIO, s, lane_));
        if (T.rs_ratio != 1) kb_prep(T, W, dSD, dIO, S, 0, 1);
        for (int b = 0; b < ngs * C; b++) WAVE_RUN(kb_psyA(T, W, dSD, dIO, b / C, b % C, lane_, LA));
        if (T.psy_channels == 4) for (int b = 0; b < ngs * 2; b++) WAVE_RUN(kb_psyA(T, W, dSD, dIO, b / 2, 2 + b % 2, lane_, LA));
        for (int b = 0; b < ngs; b++) kb_scan_raw(T, W, dSD, b);
        for (int b = 0; b < ngs; b++) kb_scan_attack(T, W, dSD, b);
        for (int b = 0; b < ngs; b++) kb_scan_blocktype(T, W, dSD, b);
#ifdef LHIP_WAVESIM
        { static thread_local AthLds LAth; for (int s = 0; s < S; s++) wsim::run_block(ATH_NT / 64, [&](int wave_, int lane_) { kb_scan_ath(T, W, dSD, s, 64 * wave_ + lane_, LAth); }); }
#else
        { static thread_local AthLds LAth; for (int s = 0; s < S; s++) kb_scan_ath(T, W, dSD, s, 0, LAth); }
#endif
        if (!resv) for (int b = 0; b < ngs; b++) WAVE_RUN(kb_psyB<4>(T, ts.pb10, W, dSD, b, lane_, LB, -1));
        for (int b = 0; b < (ngs * C + POLY_PER_WAVE - 1) / POLY_PER_WAVE; b++) WAVE_RUN(kb_polyphase(T, W, dSD, dIO, b, ngs * C, lane_, LP));
        for (int b = 0; b < ngs; b++) WAVE_RUN(kb_mdct(T, W, dSD, b, lane_, LM));
        if (resv) {
            // the per-stream reservoir program (kb_resv_stage), as g_resv_stream runs it; the wave simulation as a real workgroup of four waves
            alignas(16) static thread_local unsigned char RU[RS_WAVES][RS_LDS_PER_WAVE]; static thread_local int rmbox[4]; static thread_local CountShare rcs[2];
            for (int s = 0; s < S; s++) {
                ResvState RV = dIO[s].state->rv;
                int32_t nout = 0;
                const int F = sd[s].nframes;
#ifdef LHIP_WAVESIM
                wsim::run_block(RS_WAVES, [&](int wave_, int lane_) {
                    for (int k = 0; k <= F; k++)
                        for (int stage = 0; stage < RS_STAGES; stage++) { kb_resv_stage<1>(stage, T, ts.pb10, W, dSD, s, k, F, wave_, lane_, RU[wave_], QT, rmbox, RV, &nout, rcs); wg_barrier(); }
                    if (wave_ == 3 && sd[s].flush) kb_resv_flush(T, W, s, lane_, *(BitsLds*)RU[3], RV, &nout);
                });
#else
                for (int k = 0; k <= F; k++)
                    for (int stage = 0; stage < RS_STAGES; stage++)
                        for (int wv = 0; wv < RS_WAVES; wv++) kb_resv_stage<0>(stage, T, ts.pb10, W, dSD, s, k, F, wv, 0, RU[wv], QT, rmbox, RV, &nout);
                if (sd[s].flush) kb_resv_flush(T, W, s, 0, *(BitsLds*)RU[3], RV, &nout);
#endif
                dIO[s].state->rv = RV;
                W.out_bytes[s] = nout;
            }
        } else {
#ifdef LHIP_WAVESIM
        // small stereo batches: the two-waves-per-frame latency kernel (kb_quant<1>), as run_batch chooses on the device
        static thread_local QuantLds LQ2[2]; static thread_local int mbox[4];
        static const int pair_max = []() { const char* e = getenv("LAMEJS_HIP_PAIR_MAX_FRAMES"); return e ? atoi(e) : 12; }();
        const bool pair = (C == 2 && nfs <= pair_max);
#define QUANT_RUN(chain_) do { if (pair) wsim::run_block(2, [&](int wave_, int lane_) { kb_quant<1, 0>(T, ts.pb10, W, dSD, b, chain_, lane_, LQ2[wave_], QT, wave_, mbox); }); \
                               else WAVE_RUN(kb_quant<0, 0>(T, ts.pb10, W, dSD, b, chain_, lane_, LQ, QT)); } while (0)
#else
#define QUANT_RUN(chain_) WAVE_RUN(kb_quant<0, 0>(T, ts.pb10, W, dSD, b, chain_, lane_, LQ, QT))
#endif
#if defined(LHIP_WAVESIM)
        if (!pair) {   // the persistent kernel as a real 8-wave workgroup: frames drawn from a shared counter (kb_quant_th, what g_quant runs for one- and
                        // two-channel streams alike), then -- two channels -- the waves help each other (k_quant_tail.h)
            static thread_local QuantLds LQ8[8]; static thread_local TailShare TS;
            int ctr = 0;
            TS.drawing = 8; for (int w = 0; w < 8; w++) TS.offer[w].state = 0;
            wsim::run_block(8, [&](int wave_, int lane_) {
                int hint[3] = {-1, -1, -1};
                for (;;) {
                    int f = 0;
                    if (lane_ == 0) f = ctr++;
                    f = wave_bcast(f, 0);
                    if (f >= nfs) break;
                    kb_quant_th(T, ts.pb10, W, dSD, f, lane_, LQ8[wave_], QT, hint, TS, wave_);
                }
                if (C == 2) tail_help(T, ts.pb10, W, dSD, lane_, LQ8, wave_, 8, QT, TS);
            });
        } else
#endif
        {   // the speculative pass as three "persistent waves" striding over the frame slots, each with its own seed hint -- what g_quant's
            // waves do with the frames they draw (kb_quant: `hint`), so that the simulations cover that path too
            int hints[3][3] = {{-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}};
            for (int b = 0; b < nfs; b++) {
#ifdef LHIP_WAVESIM
                if (pair) { QUANT_RUN(0); continue; }
                int hl[64][3];                                     // every lane fiber updates its own copy; they must agree (wave-uniform)
                for (int l = 0; l < 64; l++) for (int q = 0; q < 3; q++) hl[l][q] = hints[b % 3][q];
                WAVE_RUN(kb_quant<0, 0>(T, ts.pb10, W, dSD, b, 0, lane_, LQ, QT, -1, nullptr, nullptr, hl[lane_]));
                for (int l = 1; l < 64; l++) for (int q = 0; q < 3; q++) if (hl[l][q] != hl[0][q]) { set_err("wavesim: seed hint not wave-uniform"); return false; }
                for (int q = 0; q < 3; q++) hints[b % 3][q] = hl[0][q];
#else
                WAVE_RUN(kb_quant<0, 0>(T, ts.pb10, W, dSD, b, 0, lane_, LQ, QT, -1, nullptr, nullptr, hints[b % 3]));
#endif
            }
        }
        for (;;) {
            W.nflagged[0] = 0; W.nflagged[1] = 0;
            for (int b = 0; b < nfs; b++) kb_validate_fast(T, W, dSD, b);
            for (int i = 0; i < W.nflagged[1]; i++) WAVE_RUN(kb_validate(T, ts.pb10, W, dSD, W.slow_list[i], lane_, LQ, QT));
            const int nf = W.nflagged[0];
            if (nf == 0) break;
            repaired += nf; iters++;
            for (int b = 0; b < nfs; b++) QUANT_RUN(1);
            if (iters > nfr + 2) { set_err("seed-chain repair did not converge"); return false; }
        }
        for (int b = 0; b < nfs; b++) WAVE_RUN(kb_bits(T, W, dSD, b, lane_, LBi));
        }
        for (int s = 0; s < S; s++) WAVE_RUN(kb_save(T, W, dSD, dIO, s, lane_));
        }
#undef QUANT_RUN
#undef WAVE_RUN
    }
#else
    if (use_frame) {
        QArgs qa; qa.T = T; qa.pb = ts.pb10; qa.W = W; qa.SD = dSD; qa.chain = 0; qa.nfs = nfs; qa.ctr = 0;
        if (resv) LAUNCHB(KT_QUANT, g_frame<1>, S, 64 * FR_WAVES, st, qa, dIO); else LAUNCHB(KT_QUANT, g_frame<0>, S, 64 * FR_WAVES, st, qa, dIO);
        if (resv) { bool any_flush = false; for (int i = 0; i < S; i++) any_flush |= sd[i].flush != 0; if (any_flush) LAUNCH(KT_BITS, g_resv_flush, S, st, T, W, dSD); }
    } else {
    LAUNCH(KT_LOAD, g_load, S, st, T, W, dSD, dIO);
    if (T.rs_ratio != 1) {          // only the resampler materialises samples; otherwise the consumers convert the caller's Int16 themselves
        int64_t nb = (in_total / C + 255) / 256;
        if (nb > 8192) nb = 8192;
        if (nb < 1) nb = 1;
        LAUNCHB(KT_PREP, g_prep, (int)nb, 256, st, T, W, dSD, dIO, S);
    }
    LAUNCHB(KT_PSYA, g_psyA, XCD_GRID_W(ngs * C), 64 * WPB, st, T, W, dSD, dIO, 0, C);
    if (T.psy_channels == 4) LAUNCHB(KT_PSYA, g_psyA, XCD_GRID_W(ngs * 2), 64 * WPB, st, T, W, dSD, dIO, 2, 2);
    LAUNCH(KT_SCAN, g_scan_raw, (ngs + 63) / 64, st, T, W, dSD, ngs);
    LAUNCH(KT_SCAN, g_scan_attack, (ngs + 63) / 64, st, T, W, dSD, ngs);
    LAUNCH(KT_SCAN, g_scan_blocktype, (ngs + 63) / 64, st, T, W, dSD, ngs);
    // g_scan_ath (needs psyA's loudness, feeds psyB and the quantizer) is one workgroup per stream: it runs on a side
    // stream while polyphase + MDCT (which need neither) keep the chip busy.  With per-kernel timing on, everything stays
    // on the launch stream so that the HIP events bracket each kernel.
    bool forked = false;
    struct AuxJoin {           // an error return between fork and join must not leave g_scan_ath running on the shared workspace
        void* aux = nullptr;
        ~AuxJoin() { if (aux) (void)hipStreamSynchronize((hipStream_t)aux); }
    } aux_guard;
    if (!g_kt_on) {
        if (!ws.aux_stream) {
            hipStream_t a; hipEvent_t e1, e2;
            if (hipStreamCreateWithFlags(&a, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&e2, hipEventDisableTiming) == hipSuccess) { ws.aux_stream = a; ws.ev_fork = e1; ws.ev_join = e2; }
        }
        if (ws.aux_stream && hipEventRecord((hipEvent_t)ws.ev_fork, (hipStream_t)st) == hipSuccess &&
            hipStreamWaitEvent((hipStream_t)ws.aux_stream, (hipEvent_t)ws.ev_fork, 0) == hipSuccess) {
            LAUNCHB(KT_SCAN, g_scan_ath, S, ATH_NT, ws.aux_stream, T, W, dSD);
            aux_guard.aux = ws.aux_stream;
            HIPCK(hipEventRecord((hipEvent_t)ws.ev_join, (hipStream_t)ws.aux_stream));
            forked = true;
        }
    }
    if (!forked) LAUNCHB(KT_SCAN, g_scan_ath, S, ATH_NT, st, T, W, dSD);
    LAUNCH(KT_POLY, g_poly, XCD_GRID((ngs * C + POLY_PER_WAVE - 1) / POLY_PER_WAVE), st, T, W, dSD, dIO, ngs * C);
    LAUNCH(KT_MDCT, g_mdct, XCD_GRID(ngs), st, T, W, dSD);
    if (forked) { HIPCK(hipStreamWaitEvent((hipStream_t)st, (hipEvent_t)ws.ev_join, 0)); aux_guard.aux = nullptr; }
    if (resv) {
        // bit reservoir: psyB -> quantization -> bit packing of a stream's frames are a serial chain: one workgroup per stream walks them
        // (qa.ctr = 1: the idle waves of a workgroup count for the quantizing ones -- only while every workgroup has a CU to itself: with two
        //  per CU a helper shares its SIMD with the other workgroup's searching wave and the speculative work costs more than it hides --
        //  512 streams: 2.37 M frames/s without helpers, 2.21 M with, profiles/r05_pass4_* / r05_pass5_*)
        QArgs qa; qa.T = T; qa.pb = ts.pb10; qa.W = W; qa.SD = dSD; qa.chain = 2; qa.nfs = nfs; qa.ctr = S <= ctx->num_cus ? 1 : 0;
        LAUNCHB(KT_QUANT, g_resv_stream, S, 64 * RS_WAVES, st, qa);
    } else {
    if (T.psy_channels == 4) LAUNCHB(KT_PSYB, g_psyB<4>, XCD_GRID_W(ngs), 64 * WPB, st, T, ts.pb10, W, dSD, -1);
    else LAUNCHB(KT_PSYB, g_psyB<2>, XCD_GRID_W(ngs), 64 * WPB, st, T, ts.pb10, W, dSD, -1);
    // persistent quantization kernels: as many workgroups as can be resident (2 per CU), frames dispensed dynamically
    int qgrid = (nfs + QWAVES - 1) / QWAVES;
    if (qgrid > ctx->num_cus * 2) qgrid = ctx->num_cus * 2;
#ifdef LHIP_PHASE_PROF
    const bool pair = false;
#else
    // Two waves per frame while that still leaves SIMDs under-subscribed.  Measured on MI355X (stereo 128 kbps, ms per batch,
    // persistent / pair): 600 frames 3.40 / 2.20, 1000: 3.52 / 2.41, 2000: 3.66 / 3.61, 4000: 4.82 / 5.15 -> cross-over at
    // about 2000 frames = 8 x CUs; LAMEJS_HIP_PAIR_MAX_FRAMES overrides the threshold for experiments.
    static const int pair_max = []() { const char* e = getenv("LAMEJS_HIP_PAIR_MAX_FRAMES"); return e ? atoi(e) : -1; }();
    const bool pair = (C == 2 && nfs <= (pair_max >= 0 ? pair_max : 6 * ctx->num_cus));
#endif
    { QArgs qa; qa.T = T; qa.pb = ts.pb10; qa.W = W; qa.SD = dSD; qa.chain = 0; qa.nfs = nfs; qa.ctr = 0;
      if (pair) LAUNCHB(KT_QUANT, g_quant_pair<0>, nfs, 128, st, qa); else LAUNCHB(KT_QUANT, g_quant<0>, qgrid, 64 * QWAVES, st, qa); }
    if (nfr > 0) {
        // validation of the seed chain + repair of the flagged frames, decided on the device (no host round trip in the pipeline)
        QArgs qa; qa.T = T; qa.pb = ts.pb10; qa.W = W; qa.SD = dSD; qa.chain = 1; qa.nfs = nfs; qa.ctr = 0;
        LAUNCHB(KT_VALIDATE, g_validate_fast, (nfs + 63) / 64, 256, st, T, W, dSD, nfs);
        // as many workgroups as can be resident (two per CU): the memo-miss re-validation (a quarter to a third of the frames of steady
        // material) is spread over all of them -- a quarter-chip grid was tried and doubled this stage's time
        int fgrid = (nfs + 63) / 64;
        if (ctx->fixup_wg_per_cu == 0) {      // once per context: how many of this build's g_fixup workgroups a CU really holds (a grid barrier needs them all resident)
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)g_fixup, 64 * QWAVES, 0) != hipSuccess || nb < 1) nb = 1;
            ctx->fixup_wg_per_cu = nb < LHIP_FIXUP_OCC / 2 ? nb : LHIP_FIXUP_OCC / 2;
        }
        if (fgrid > ctx->num_cus * ctx->fixup_wg_per_cu) fgrid = ctx->num_cus * ctx->fixup_wg_per_cu;
        if (fgrid < 1) fgrid = 1;
        if (fgrid == 1) LAUNCHB(KT_VALIDATE, g_fixup, 1, 64 * QWAVES, st, qa);
        else {
            kt_begin(KT_VALIDATE, st);
            void* kargs[] = {(void*)&qa};
            // An aliased context (tests, LHIP_ALIAS_DEVICES) runs on a stream the library created, and its batches come from worker threads: after a cooperative
            // launch on such a stream from a thread that has since ended, ROCm 7.2's own exit handler crashes inside the HSA runtime (seen on gfx950: every variant
            // without the cooperative launch, or on the null stream, or from the main thread, exits cleanly).  So there the launch goes through the null
            // stream, ordered behind / in front of the context's stream by two events.
            hipStream_t cst = (hipStream_t)st;
            if (ctx->own_stream) {
                if (!ctx->ev_coop[0]) { if (!rt::event_create(&ctx->ev_coop[0]) || !rt::event_create(&ctx->ev_coop[1])) return false; }
                if (!rt::event_record(ctx->ev_coop[0], st) || !rt::stream_wait_event(nullptr, ctx->ev_coop[0])) return false;
                cst = nullptr;
            }
            hipError_t e_ = hipLaunchCooperativeKernel((const void*)g_fixup, dim3(fgrid), dim3(64 * QWAVES), kargs, 0, cst);
            if (ctx->own_stream && e_ == hipSuccess) { if (!rt::event_record(ctx->ev_coop[1], nullptr) || !rt::stream_wait_event(st, ctx->ev_coop[1])) return false; }
            kt_end(st);
            TRACE_SYNC(g_fixup_cooperative, st);
            if (e_ != hipSuccess) { set_err(std::string("g_fixup (cooperative launch): ") + hipGetErrorString(e_)); return false; }
        }
    }
    LAUNCH(KT_BITS, g_bits, nfs, st, T, W, dSD);
    }
    LAUNCH(KT_SAVE, g_save, S, st, T, W, dSD, dIO);
    }
#endif
    CALL_STAMP(2);                                  // kernels enqueued
    // repair statistics live on the device; they travel with the final synchronisation when there is one, else they are fetched
    // when somebody asks (lhip_last_batch_stats)
    int32_t fx[3] = {0, 0, 0};
    const bool fetch_fx = nfr > 0 && (!dev_io || want_sync || g_kt_on_());
    if (fx_dst) {
        if (nfr > 0) { if (!rt::d2d(fx_dst, (const uint8_t*)W.nflagged + FX_STATS_OFF, 12, st)) return false; }
        else if (!rt::dzero(fx_dst, 12, st)) return false;
    }
    // ---- outputs ----
    if (small) {
        // one copy out: [output bytes | counters | out_bytes], then the callers' buffers are filled from the pinned mirror
        const uint8_t* po = (const uint8_t*)ws.pin_out.p;
        if (!rt::d2h(ws.pin_out.p, smb, sm_sf, st) || !rt::sync(st)) return false;
        memcpy(fx, po + sm_nfl + FX_STATS_OFF, sizeof fx);
        if (resv) for (int i = 0; i < S; i++) jobs[i].bytes = ((const int32_t*)(po + sm_ob))[i];
        for (int i = 0; i < S; i++) if (jobs[i].bytes > 0) memcpy(jobs[i].out, po + sm_out + out_rel[i], (size_t)jobs[i].bytes);
    } else {
        if (fetch_fx && !rt::d2h(fx, (const int32_t*)ws.nflagged.p + FX_STATS_OFF / 4, sizeof fx, st)) return false;
        std::vector<int32_t> ob;
        if (resv) {                                   // how much each stream really wrote
            ob.assign((size_t)S, 0);
            if (!rt::d2h(ob.data(), ws.out_bytes.p, (size_t)S * 4, st) || !rt::sync(st)) return false;
            for (int i = 0; i < S; i++) jobs[i].bytes = ob[i];
        }
        if (!dev_io) {
            for (int i = 0; i < S; i++)
                if (jobs[i].bytes > 0 && !rt::d2h(jobs[i].out, io[i].out, (size_t)jobs[i].bytes, st)) return false;
            if (!rt::sync(st)) return false;
        } else if (want_sync) {
            if (!rt::sync(st)) return false;
        }
    }
#ifndef LHIP_HOSTSIM
    if (g_kt_on) { if (!rt::sync(st)) return false; kt_collect(); }
#endif
    CALL_STAMP(3);                                  // output copies + synchronisation
    // ---- host-side stream bookkeeping (Lame.js:1629-1661) ----
    for (int i = 0; i < S; i++) {
        Job& j = jobs[i];
        lhip_stream* s = j.s;
        const int64_t total = (int64_t)s->mf_size + j.n_out;
        if (j.n > 0) {
            if (s->mf_samples_to_encode < 1) s->mf_samples_to_encode = 576 + 1152;
            s->mf_samples_to_encode += (int)j.n_out;
        }
        s->rs_n_in += (int64_t)j.n;
        s->mf_samples_to_encode -= frame * j.F;
        s->mf_size = (int)(total - (int64_t)frame * j.F);
        if (T.frac_SpF != 0 && j.F > 0) {
            int64_t m = ((int64_t)s->slot_lag - (int64_t)j.F * T.frac_SpF) % T.out_samplerate;
            if (m < 0) m += T.out_samplerate;
            s->slot_lag = (int)m;
        }
        s->frame_num += j.F;
        j.written = j.bytes;
    }
#ifndef LHIP_HOSTSIM
    g_stat_pending = nullptr;
    if (fetch_fx) {
        repaired = fx[0]; iters = fx[1];
        if (fx[2]) { set_err("seed-chain repair did not converge"); return false; }
    } else if (nfr > 0) g_stat_pending = ctx;
#endif
    g_stat_frames = nfr; g_stat_repaired = repaired; g_stat_iters = iters;
    ws.lastW = W; ws.lastC = C; ws.lastCp = T.psy_channels; ws.have_last = true;
    return true;
}

// ===========================================================================================
// C ABI
// ===========================================================================================
extern "C" {

int lhip_device_count(void) { return rt::device_count(); }

static std::atomic<uint64_t> g_dev_mask{0};
static std::atomic<unsigned> g_dev_rr{0};
int lhip_set_devices(uint64_t mask) {
    const int n = rt::device_count();
    if (n <= 0) { set_err("no HIP device available (this library has no CPU fallback)"); return LHIP_ERR_INTERNAL; }
    const uint64_t all = n >= 64 ? ~0ull : ((1ull << n) - 1);
    if (mask & ~all) { set_err("lhip_set_devices: mask names a device that does not exist"); return LHIP_ERR_INTERNAL; }
    g_dev_mask = mask; g_dev_rr = 0;
    return mask ? __builtin_popcountll(mask) : n;
}

int lhip_stream_device(const lhip_stream* s) {
    if (!s || s->magic != 0x4c484950) { set_err("bad stream handle"); return LHIP_ERR_BAD_HANDLE; }
    return s->ctx->device;
}

int lhip_device_identity(int device, char* pci_bus_id, char* uuid_hex, size_t cap) {
    if (!pci_bus_id || !uuid_hex || cap < 2) { set_err("null argument"); return LHIP_ERR_INTERNAL; }
    pci_bus_id[0] = 0; uuid_hex[0] = 0;
#ifdef LHIP_HOSTSIM
    snprintf(pci_bus_id, cap, "hostsim:%d", device < 0 ? 0 : device); snprintf(uuid_hex, cap, "hostsim-%d", device < 0 ? 0 : device);
    return 0;
#else
    if (device < 0 && hipGetDevice(&device) != hipSuccess) { set_err("hipGetDevice failed"); return LHIP_ERR_INTERNAL; }
    const int pd = rt::phys(device);
    if (hipDeviceGetPCIBusId(pci_bus_id, (int)cap, pd) != hipSuccess) { set_err("hipDeviceGetPCIBusId failed"); return LHIP_ERR_INTERNAL; }
    hipUUID u;
    if (hipDeviceGetUuid(&u, pd) == hipSuccess) {
        size_t o = 0;
        for (int i = 0; i < 16 && o + 3 <= cap; i++) o += (size_t)snprintf(uuid_hex + o, cap - o, "%02x", (unsigned)(unsigned char)u.bytes[i]);
    }
    return 0;
#endif
}

const char* lhip_last_error(void) { return g_err.c_str(); }
const char* lhip_version(void) {
#ifdef LHIP_HOSTSIM
    return "lamejs_amd 0.1 (HOST SIMULATION - tests only)";
#else
    return "lamejs_amd 0.1 (HIP gfx950)";
#endif
}

int lhip_create(const lhip_config* cfg, const void* tables, size_t tables_bytes, lhip_stream** out) {
    if (!cfg || !tables || !out) { set_err("null argument"); return LHIP_ERR_INTERNAL; }
    *out = nullptr;
    if (rt::device_count() <= 0) { set_err("no HIP device available (this library has no CPU fallback)"); return LHIP_ERR_INTERNAL; }
    int dev = cfg->device;
    const uint64_t mask = g_dev_mask;
    if (mask) {
        if (dev >= 0 && !((mask >> dev) & 1)) { set_err("device excluded by lhip_set_devices"); return LHIP_ERR_INTERNAL; }
        if (dev < 0) {                       // deal streams round-robin over the allowed devices
            unsigned k = g_dev_rr++ % (unsigned)__builtin_popcountll(mask);
            for (dev = 0; dev < 64; dev++) if ((mask >> dev) & 1) { if (k == 0) break; k--; }
        }
    }
#ifndef LHIP_HOSTSIM
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) { set_err("hipGetDevice failed"); return LHIP_ERR_INTERNAL; } }
#else
    if (dev < 0) dev = 0;
#endif
    if (dev >= rt::device_count()) { set_err("no such HIP device"); return LHIP_ERR_INTERNAL; }
    Context* ctx = get_context(dev);
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!rt::set_device(dev)) return LHIP_ERR_INTERNAL;
    std::string key((const char*)tables, tables_bytes);
    std::shared_ptr<TableSet> ts;
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        ts = it->second;
        // the same checks build_tables makes of the blob against the requested configuration
        if (ts->T.channels_out != (cfg->channels == 1 ? 1 : 2) || ts->T.in_samplerate != cfg->samplerate) { set_err("tables blob does not match the requested configuration"); return LHIP_ERR_INTERNAL; }
    } else {
        ts = std::make_shared<TableSet>();
        ts->device = dev;
        if (!build_tables(*ts, tables, tables_bytes, *cfg, ctx->stream)) return LHIP_ERR_INTERNAL;
        ctx->tables[key] = ts;
    }
    std::unique_ptr<lhip_stream> s(new lhip_stream());
    s->ctx = ctx; s->ts = ts;
    s->slot_lag = ts->T.frac_SpF;
    // initial carried state (PsyModel.js:2566-2596 psymodel_init, Lame.js:168-171 lame_init_old)
    std::unique_ptr<StreamState> h(new StreamState());
    memset(h.get(), 0, sizeof(StreamState));
    for (int ch = 0; ch < 4; ch++) {                      // psy state exists for L, R, mid, side (PsyModel.js:2566-2596)
        for (int i = 0; i < E_STRIDE; i++) h->E[ch][i] = 1e20f;
        for (int i = 0; i < EBS_STRIDE; i++) h->ecb_s[ch][i] = 1.0f;
        for (int i = 0; i < 9; i++) h->peaks[ch][i] = 10.f;
        for (int i = 0; i < EBL_STRIDE; i++) h->nb1[ch][i] = h->nb2[ch][i] = 1e20f;
    }
    for (int i = 0; i < 19; i++) h->rv.pefirbuf[i] = (float)(700 * ts->T.mode_gr * ts->T.channels_out);      // Lame.js:1120
    for (int ch = 0; ch < 2; ch++) {
        h->tent[ch] = NORM_TYPE;
        h->last_bt[ch] = -1;
        h->seed[ch][0] = 180; h->seed[ch][1] = 4;
    }
    h->ath_adjust = 0.01; h->ath_limit = 1.0;
    s->d_state = (StreamState*)rt::dmalloc(sizeof(StreamState));
    if (!s->d_state) { set_err("hipMalloc(stream state) failed"); return LHIP_ERR_INTERNAL; }
    if (!rt::h2d(s->d_state, h.get(), sizeof(StreamState), ctx->stream) || !rt::sync(ctx->stream)) return LHIP_ERR_INTERNAL;
    ctx->live_streams++;
    *out = s.release();
    return 0;
}

void lhip_destroy(lhip_stream* s) {
    if (!s || s->magic != 0x4c484950) return;
    Context* ctx = s->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    rt::set_device(ctx->device);
    std::shared_ptr<TableSet> ts = s->ts;
    delete s;
    if (--ctx->live_streams == 0) { (void)rt::sync(ctx->stream); ctx->ws.pin_in.release(); ctx->ws.pin_out.release(); }      // (grow-only while streams live; a later stream allocates them again)
    // the cache entry goes with the last stream that uses it (one reference is the map's, one is `ts` here)
    if (ts.use_count() == 2)
        for (auto it = ctx->tables.begin(); it != ctx->tables.end(); ++it) if (it->second == ts) { ctx->tables.erase(it); break; }
}

size_t lhip_max_output_bytes(const lhip_stream* s, size_t nsamples) {
    if (!s || s->magic != 0x4c484950) return 0;
    const size_t frame = 576 * (size_t)s->ts->T.mode_gr;
    return (nsamples / frame + 3 + (FRAME / frame)) * (size_t)(s->ts->base_frame_bytes + 1) + (s->ts->T.disable_reservoir ? 0 : 4096);      // reservoir: slack for the per-launch bound
}

static int call_frames(const lhip_stream* s, size_t nsamples);
int64_t lhip_encode_output_bytes(const lhip_stream* s, size_t nsamples) {
    if (!s || s->magic != 0x4c484950) { set_err("bad stream handle"); return LHIP_ERR_BAD_HANDLE; }
    if (!s->ts->T.disable_reservoir) return (int64_t)lhip_max_output_bytes(s, nsamples);      // data-dependent: only a bound exists
    return batch_bytes(*s->ts, s->slot_lag, call_frames(s, nsamples));
}

int lhip_output_bytes_is_exact(const lhip_stream* s) {
    if (!s || s->magic != 0x4c484950) { set_err("bad stream handle"); return LHIP_ERR_BAD_HANDLE; }
    return s->ts->T.disable_reservoir ? 1 : 0;
}

static int encode_many(lhip_stream* const* streams, size_t n, const int16_t* const* l, const int16_t* const* r,
                       const size_t* ns, uint8_t* const* out, const size_t* cap, int64_t* written, bool dev_io, bool sync, bool flush_stream = false) {
    if (n == 0) return 0;
    std::vector<Job> jobs(n);
    for (size_t i = 0; i < n; i++) {
        if (!streams[i] || streams[i]->magic != 0x4c484950) { set_err("bad stream handle"); return LHIP_ERR_BAD_HANDLE; }
        if (streams[i]->ctx != streams[0]->ctx) { set_err("batch: streams on different devices"); return LHIP_ERR_INTERNAL; }
        jobs[i] = Job{streams[i], l[i], r ? r[i] : nullptr, ns[i], out[i], cap[i], 0, 0, 0, 0};
    }
    // Bit reservoir (extension): the frames of a stream are a serial chain, walked by one workgroup per stream inside the launch
    // (g_resv_stream); a stream that ends with this call (flush) has its bitstream padded by the same launch -- decided per stream
    // (an already flushed stream in a flush batch has nothing to encode and is not flushed again).  The byte counts are only known
    // on the device, so these calls always synchronise.
    const Tables& T0 = streams[0]->ts->T;
    const bool resv = !T0.disable_reservoir;
    if (resv) for (size_t i = 0; i < n; i++) jobs[i].flush = flush_stream && ns[i] > 0;
    const bool ok = run_batch(streams[0]->ctx, jobs, dev_io, sync || resv);
    for (size_t i = 0; i < n; i++) if (written) written[i] = ok ? jobs[i].written : (jobs[i].written < 0 ? jobs[i].written : LHIP_ERR_INTERNAL);
    if (!ok) { for (auto& j : jobs) if (j.written < 0) return (int)j.written; return LHIP_ERR_INTERNAL; }
    return 0;
}

// Host-buffer calls with many frames (what encodeBuffer() hands over when a caller passes a long Int16Array): the call is cut into
// chunks of whole frames' worth of samples; chunk k + 1 travels to the device (copy stream) while chunk k is encoded (launch stream)
// and the bytes of chunk k - 1 travel back -- PCIe needs about a sixth of the encode time, so it hides behind it.  Any chunking of a
// sample stream gives the same bytes (the library's basic contract), so the result is what one batch gives.  Not for the bit
// reservoir (its byte counts are only known after each launch).
enum { HOST_CHUNK_FRAMES = 8192 };
// chunk schedule: the first chunk is small (its copy is the part of the call nothing overlaps), every later one twice the one before up
// to a cap -- a chunk's copy still fits inside the encode of the chunk before it, and large chunks keep the persistent quantization
// kernel's waves busy (at 8192 frames a wave draws two frames and every launch ends on its slowest one: 1e5 stereo frames took 72.5 ms
// in 8192-frame chunks, 58.3 ms with 8192 doubling to 32768, 60.0 ms as one batch with nothing overlapped; tests/tools/dropin_sweep.py).
// Two-channel streams take chunks of twice the frames (a stereo frame is four to five times the work of a mono frame, so a chunk's fixed
// costs -- the launch tails -- weigh the same at twice the size, and its copy hides as well): 16384 doubling to 65536 measured 52.0 ms
// against 53.8 ms with the mono schedule on the final code of round 3, mono the other way round (11.96 vs 12.52 ms;
// profiles/r03_dropin_host_chunk_sweep.txt).  LAMEJS_HIP_HOST_CHUNK_FRAMES=first[,cap[,growth]] overrides both (tuning, tests).
struct ChunkSchedule { size_t first = HOST_CHUNK_FRAMES, cap = 4 * HOST_CHUNK_FRAMES, growth = 2; bool fixed = false; };
static const ChunkSchedule& host_chunk_schedule() {
    static const ChunkSchedule cs = []() {           // read once (function-local static: initialised exactly once, whichever thread comes first)
        ChunkSchedule c;
        if (const char* e = getenv("LAMEJS_HIP_HOST_CHUNK_FRAMES")) {
            char* end = nullptr;
            const unsigned long v = strtoul(e, &end, 10);
            if (v >= 1 && v <= (1ul << 20)) { c.first = v; c.cap = v; c.fixed = true; }
            if (end && *end == ',') {
                const unsigned long w = strtoul(end + 1, &end, 10); if (w >= c.first && w <= (1ul << 20)) c.cap = w;
                if (end && *end == ',') { const unsigned long gr = strtoul(end + 1, nullptr, 10); if (gr >= 2 && gr <= 8) c.growth = gr; }
            }
        }
        return c;
    }();
    return cs;
}
// whole frames (and their bytes: exact without the bit reservoir) that a call with `nsamples` more input samples completes on this stream
static int call_frames(const lhip_stream* s, size_t nsamples) {
    const Tables& T = s->ts->T;
    const int frame = 576 * T.mode_gr, mf_needed = 1024 + frame - 272;
    const int64_t n_out = T.rs_ratio == 1 ? (int64_t)nsamples : rs_outputs(s->rs_n_in + (int64_t)nsamples, T.rs_ratio) - rs_outputs(s->rs_n_in, T.rs_ratio);
    const int64_t total = (int64_t)s->mf_size + n_out;
    return total >= mf_needed ? (int)((total - mf_needed) / frame) + 1 : 0;
}
// One piece of host input for one stream: `n` samples per channel from l / r; its frames' bytes go to the stream's destination (running).
struct HostPiece { int si; const int16_t* l; const int16_t* r; size_t n; };
// The overlapped host path, in general form: `units` are processed in order, each a batch of pieces (one per stream at most) -- unit k + 1
// travels to the device (copy stream) while unit k is encoded (launch stream) and the bytes of unit k - 1 travel back.  For ONE long stream
// the units are consecutive sample ranges of its input (encode_host_chunked); for MANY streams (lhip_encode_batch with host buffers,
// BASELINE configs[4] through the JavaScript encodeBatch) they are groups of streams.  dst[si] / cap[si]: stream si's output buffer;
// written[si] receives its byte count.  A failed call gives every stream back as it found it.
static int encode_host_pipelined(Context* ctx, const std::vector<lhip_stream*>& strs, const std::vector<std::vector<HostPiece>>& units,
                                 uint8_t* const* dst, int64_t* written) {
    std::lock_guard<std::mutex> chunk_lk(ctx->chunk_mu);
    const size_t NS = strs.size();
    const Tables& T = strs[0]->ts->T;
    const int C = T.channels_out;
    const size_t obytes = (size_t)(strs[0]->ts->base_frame_bytes + 1);
    const size_t spf = (size_t)576 * T.mode_gr * T.rs_ratio;             // input samples per frame
    // staging halves sized for the largest unit of THIS call: samples per channel (pieces back to back, each rounded up to 64) and output bytes
    size_t in_max = 0, out_max = 0;
    for (const auto& u : units) {
        size_t a = 0, o = 0;
        for (const HostPiece& pc : u) { a += (pc.n + 63) & ~(size_t)63; o += ((pc.n / spf + 3) * obytes + 63) & ~(size_t)63; }
        if (a > in_max) in_max = a;
        if (o > out_max) out_max = o;
    }
    const size_t stride = in_max, out_chunk = out_max + 64;
    // what a failed call must give back: the host-side counters and the device-side state record of every stream (a call that fails in unit
    // k > 0 would otherwise leave streams k units further on with `out` half written -- "a failed call consumes nothing" has to hold here too)
    struct Snap { int mf, ste, lag; int64_t fn, rs; };
    std::vector<Snap> snap(NS);
    for (size_t i = 0; i < NS; i++) snap[i] = Snap{strs[i]->mf_size, strs[i]->mf_samples_to_encode, strs[i]->slot_lag, strs[i]->frame_num, strs[i]->rs_n_in};
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (!rt::set_device(ctx->device)) return LHIP_ERR_INTERNAL;
        if (!ctx->copy_stream) {
            void* cs = nullptr; void* e[4] = {nullptr, nullptr, nullptr, nullptr};
            if (!rt::stream_create(&cs)) return LHIP_ERR_INTERNAL;
            for (int i = 0; i < 4; i++) if (!rt::event_create(&e[i])) return LHIP_ERR_INTERNAL;
            ctx->ev_in[0] = e[0]; ctx->ev_in[1] = e[1]; ctx->ev_done[0] = e[2]; ctx->ev_done[1] = e[3]; ctx->copy_stream = cs;
        }
        if (!ctx->chunk_in.ensure(2 * C * stride * 2 + 64) || !ctx->chunk_out.ensure(2 * out_chunk) || !ctx->chunk_fx.ensure(units.size() * 16 + 16) ||
            !ctx->state_bak.ensure(NS * sizeof(StreamState))) return LHIP_ERR_INTERNAL;
        for (size_t i = 0; i < NS; i++)
            if (!rt::d2d((uint8_t*)ctx->state_bak.p + i * sizeof(StreamState), strs[i]->d_state, sizeof(StreamState), ctx->stream)) return LHIP_ERR_INTERNAL;
    }
    void* cs = ctx->copy_stream; void* ks = ctx->stream;
    int64_t frames_all = 0, repaired_all = 0, iters_all = 0;
    std::vector<int64_t> total(NS, 0);
    struct Pending { uint8_t* dst; const uint8_t* src; int64_t bytes; };
    std::vector<Pending> pending; int pending_par = 0; bool have_pending = false;      // the unit whose output is still on the device
    auto fail = [&](const char* what, int64_t code = LHIP_ERR_INTERNAL) -> int {      // wait for everything in flight, then put the streams back where the call found them
        const std::string why = what ? std::string(what) : g_err;
        (void)rt::sync(cs); (void)rt::sync(ks);
        for (size_t i = 0; i < NS; i++) {
            (void)rt::d2d(strs[i]->d_state, (const uint8_t*)ctx->state_bak.p + i * sizeof(StreamState), sizeof(StreamState), ks);
            strs[i]->mf_size = snap[i].mf; strs[i]->mf_samples_to_encode = snap[i].ste; strs[i]->slot_lag = snap[i].lag; strs[i]->frame_num = snap[i].fn; strs[i]->rs_n_in = snap[i].rs;
        }
        (void)rt::sync(ks);
        set_err(why);
        return (int)code;
    };
    auto drain = [&]() -> bool {               // copy the pending unit's bytes out; ALWAYS waits for that unit's kernels (its input half is reused next)
        if (!have_pending) return true;
        have_pending = false;
        if (!rt::stream_wait_event(cs, ctx->ev_done[pending_par])) return false;
        for (const Pending& q : pending) if (q.bytes > 0 && !rt::d2h(q.dst, q.src, (size_t)q.bytes, cs)) return false;
        return rt::sync(cs);
    };
    // LAMEJS_HIP_TRACE_CHUNKS=1: host-side timeline of the call on stderr (ms since the call began: after the input copies were issued, after the
    // kernels were enqueued, after the previous unit's bytes arrived) -- where a slow caller-side buffer shows
    static const bool trace_chunks = []() { const char* e = getenv("LAMEJS_HIP_TRACE_CHUNKS"); return e && e[0] == '1'; }();
    const auto t_call = std::chrono::steady_clock::now();
    auto ms_now = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(); };
    for (size_t k = 0; k < units.size(); k++) {
        const int par = (int)(k & 1);
        const std::vector<HostPiece>& u = units[k];
        const double t_a = trace_chunks ? ms_now() : 0.0;
        int16_t* d_in = (int16_t*)ctx->chunk_in.p + (size_t)par * C * stride;
        uint8_t* d_out = (uint8_t*)ctx->chunk_out.p + (size_t)par * out_chunk;
        // buffer `par` was last used by unit k - 2: its kernels are done (drain() waited for them before unit k - 1 was enqueued)
        // (copies straight from the caller's pageable memory: measured as fast as copies through pinned staging filled by four host
        //  threads -- 72.5 vs 73.2 ms per 1e5 stereo frames at 8192-frame chunks -- so there is no staging layer)
        std::vector<Job> jobs(u.size());
        size_t io = 0, oo = 0;
        for (size_t j = 0; j < u.size(); j++) {
            const HostPiece& pc = u[j];
            if (!rt::h2d(d_in + io, pc.l, pc.n * 2, cs)) return fail(nullptr);
            if (C == 2 && !rt::h2d(d_in + stride + io, pc.r ? pc.r : pc.l, pc.n * 2, cs)) return fail(nullptr);
            const size_t ocap = ((pc.n / spf + 3) * obytes + 63) & ~(size_t)63;
            jobs[j] = Job{strs[pc.si], d_in + io, C == 2 ? d_in + stride + io : nullptr, pc.n, d_out + oo, ocap, 0, 0, 0, 0};
            io += (pc.n + 63) & ~(size_t)63; oo += ocap;
        }
        if (!rt::event_record(ctx->ev_in[par], cs) || !rt::stream_wait_event(ks, ctx->ev_in[par])) return fail(nullptr);
        const double t_b = trace_chunks ? ms_now() : 0.0;
        // (this unit's repair verdict stays on the device until the call ends: run_batch copies it into the call's log -- stream-ordered, under the context's
        //  lock -- and the log is read back once after the last unit)
        if (!run_batch(ctx, jobs, true, false, (int32_t*)ctx->chunk_fx.p + 4 * k)) { int64_t code = LHIP_ERR_INTERNAL; for (const Job& j : jobs) if (j.written < 0) { code = j.written; break; } return fail(nullptr, code); }
#ifdef LHIP_HOSTSIM
        // tests: a failure injected after unit k has been consumed (the streams must come back as the call found them)
        if (const char* e = getenv("LHIP_HOSTSIM_FAIL_CHUNK")) if (e[0] && (size_t)atoi(e) == k) return fail("injected failure (LHIP_HOSTSIM_FAIL_CHUNK)");
        repaired_all += g_stat_repaired; iters_all += g_stat_iters;
#endif
        if (!rt::event_record(ctx->ev_done[par], ks)) return fail(nullptr);
        const double t_c = trace_chunks ? ms_now() : 0.0;
        if (!drain()) return fail(nullptr);             // unit k - 1, while unit k is being encoded
        if (trace_chunks) fprintf(stderr, "[lhip unit %zu: %zu piece(s)] begin %.2f  copies issued %.2f  kernels enqueued %.2f  previous unit's bytes home %.2f ms\n", k, u.size(), t_a, t_b, t_c, ms_now());
        pending.clear();
        for (size_t j = 0; j < u.size(); j++) {
            const int si = u[j].si;
            pending.push_back(Pending{dst[si] + total[si], jobs[j].out, jobs[j].written});
            total[si] += jobs[j].written;
        }
        pending_par = par; have_pending = true;
        frames_all += g_stat_frames;
    }
    if (!drain()) return fail(nullptr);
    if (trace_chunks) fprintf(stderr, "[lhip units] last unit's bytes home %.2f ms\n", ms_now());
#ifndef LHIP_HOSTSIM
    {
        std::vector<int32_t> fx(4 * units.size(), 0);
        if (!rt::d2h(fx.data(), ctx->chunk_fx.p, fx.size() * 4, ks) || !rt::sync(ks)) return fail(nullptr);
        bool bad = false;
        for (size_t k = 0; k < units.size(); k++) { repaired_all += fx[4 * k]; iters_all += fx[4 * k + 1]; bad |= fx[4 * k + 2] != 0; }
        if (bad) return fail("seed-chain repair did not converge");
    }
    g_stat_pending = nullptr;
#endif
    g_stat_frames = frames_all; g_stat_repaired = repaired_all; g_stat_iters = iters_all;     // lhip_last_batch_stats: the whole call
    for (size_t i = 0; i < NS; i++) written[i] = total[i];
    return 0;
}

// ONE long stream: consecutive sample ranges of the call (chunk schedule above)
static int64_t encode_host_chunked(lhip_stream* s, const int16_t* left, const int16_t* right, size_t nsamples, uint8_t* out, size_t out_cap) {
    const Tables& T = s->ts->T;
    const ChunkSchedule& cfg = host_chunk_schedule();
    const size_t mul = (!cfg.fixed && T.channels_out == 2) ? 2 : 1;
    const size_t spf = (size_t)576 * T.mode_gr * T.rs_ratio;
    // the whole call must fit the caller's buffer BEFORE anything is consumed (a failed call consumes nothing)
    if ((size_t)batch_bytes(*s->ts, s->slot_lag, call_frames(s, nsamples)) > out_cap) { set_err("output buffer too small"); return LHIP_ERR_BUFFER_TOO_SMALL; }
    std::vector<std::vector<HostPiece>> units;
    {
        const size_t cap = cfg.cap * mul * spf;
        size_t p = 0, cur = cfg.first * mul * spf;
        while (p < nsamples) {
            size_t m = nsamples - p < cur ? nsamples - p : cur;
            if (nsamples - p - m < m / 4 && nsamples - p <= cap) m = nsamples - p;     // no short chunk at the end: a launch for a few frames costs a whole tail
            units.push_back({HostPiece{0, left + p, right ? right + p : nullptr, m}});
            p += m;
            cur = cfg.growth * cur < cap ? cfg.growth * cur : cap;
        }
    }
    int64_t w = 0;
    uint8_t* d = out;
    const int rc = encode_host_pipelined(s->ctx, {s}, units, &d, &w);
    return rc < 0 ? rc : w;
}

// MANY streams with host buffers (lhip_encode_batch): groups of streams as units -- a group's copies hide behind the encode of the group before
static int encode_host_groups(lhip_stream* const* streams, size_t n, const int16_t* const* l, const int16_t* const* r, const size_t* ns,
                              uint8_t* const* out, const size_t* cap, int64_t* written) {
    const Tables& T = streams[0]->ts->T;
    const size_t spf = (size_t)576 * T.mode_gr * T.rs_ratio;
    for (size_t i = 0; i < n; i++)
        if ((size_t)batch_bytes(*streams[i]->ts, streams[i]->slot_lag, call_frames(streams[i], ns[i])) > cap[i]) {
            for (size_t k = 0; k < n; k++) if (written) written[k] = LHIP_ERR_BUFFER_TOO_SMALL;
            set_err("output buffer too small"); return LHIP_ERR_BUFFER_TOO_SMALL;
        }
    // groups of about 2 x the first chunk of the one-stream schedule (16384 one-channel frames): large enough for the persistent kernel,
    // small enough that the first group's copy -- the part nothing overlaps -- stays short
    const size_t target = 2 * host_chunk_schedule().first * spf;
    std::vector<std::vector<HostPiece>> units(1);
    size_t acc = 0;
    for (size_t i = 0; i < n; i++) {
        if (acc >= target) { units.emplace_back(); acc = 0; }
        units.back().push_back(HostPiece{(int)i, l[i], r ? r[i] : nullptr, ns[i]});
        acc += ns[i];
    }
    std::vector<lhip_stream*> strs(streams, streams + n);
    std::vector<int64_t> w(n, 0);
    const int rc = encode_host_pipelined(streams[0]->ctx, strs, units, out, w.data());
    for (size_t i = 0; i < n; i++) if (written) written[i] = rc < 0 ? (int64_t)rc : w[i];
    return rc;
}

int64_t lhip_encode(lhip_stream* s, const int16_t* left, const int16_t* right, size_t nsamples, uint8_t* out, size_t out_cap) {
    if (!s || s->magic != 0x4c484950) { set_err("bad stream handle"); return LHIP_ERR_BAD_HANDLE; }
    if (nsamples == 0) return 0;
    if (!left) { set_err("null input"); return LHIP_ERR_INTERNAL; }
    {
        static const bool no_chunk = []() { const char* e = getenv("LAMEJS_HIP_NO_HOST_CHUNKS"); return e && e[0] == '1'; }();
        const Tables& T = s->ts->T;
        if (!no_chunk && T.disable_reservoir && nsamples > (size_t)2 * host_chunk_schedule().first * 576 * T.mode_gr * T.rs_ratio) return encode_host_chunked(s, left, right, nsamples, out, out_cap);
    }
    int64_t w = 0;
    const int rc = encode_many(&s, 1, &left, &right, &nsamples, &out, &out_cap, &w, false, true);
    return rc < 0 ? rc : w;
}

static size_t flush_zeros(lhip_stream* s) {
    // Lame.js:1381-1443: the flush loop feeds bunches of at most 1152 zeros (fill_buffer takes them one frame at a
    // time) until `frames_left` bunches have each completed at least one frame; the total number of zeros is what
    // the batch path needs, the frames follow from it
    if (s->mf_samples_to_encode < 1) return 0;
    const Tables& T = s->ts->T;
    const int frame = 576 * T.mode_gr, mf_needed = 1024 + frame - 272, r = T.rs_ratio;
    // doubles where the reference's numbers can be fractional (16/r for r = 3)
    double samples_to_encode = s->mf_samples_to_encode - 1152;
    if (T.in_samplerate != T.out_samplerate) samples_to_encode += 16. * T.out_samplerate / T.in_samplerate;
    double end_padding = frame - fmod(samples_to_encode, (double)frame);
    if (end_padding < 576) end_padding += frame;
    double frames_left = (samples_to_encode + end_padding) / frame;
    int mf = s->mf_size;
    int64_t n_in = s->rs_n_in;
    size_t zeros = 0;
    while (frames_left > 0) {
        int64_t bunch = (int64_t)(mf_needed - mf) * r;           // bunch *= in_samplerate; bunch /= out_samplerate (exact: integer ratio)
        if (bunch > 1152) bunch = 1152;
        if (bunch < 1) bunch = 1;
        int emitted = 0;
        if (r == 1) {
            for (int rem = (int)bunch; rem > 0;) {
                const int n = rem < frame ? rem : frame;
                mf += n; rem -= n;
                if (mf >= mf_needed) { emitted++; mf -= frame; }
            }
        } else {
            // the fill loop adds at most one frame of resampled samples per pass and encodes whenever mf_needed is reached
            mf += (int)(rs_outputs(n_in + bunch, r) - rs_outputs(n_in, r));
            n_in += bunch;
            while (mf >= mf_needed) { emitted++; mf -= frame; }
        }
        zeros += (size_t)bunch;
        if (emitted) frames_left--;
    }
    return zeros;
}

int64_t lhip_flush(lhip_stream* s, uint8_t* out, size_t out_cap) {
    if (!s || s->magic != 0x4c484950) { set_err("bad stream handle"); return LHIP_ERR_BAD_HANDLE; }
    const size_t z = flush_zeros(s);
    if (z == 0) { s->mf_samples_to_encode = 0; return 0; }
    std::vector<int16_t> zeros(z, 0);
    const int16_t* l = zeros.data();
    const int16_t* r = zeros.data();
    int64_t w = 0;
    const int rc = encode_many(&s, 1, &l, &r, &z, &out, &out_cap, &w, false, true, true);
    if (rc < 0) return rc;                 // nothing was consumed (e.g. -1: the call can be repeated with a larger buffer)
    s->mf_samples_to_encode = 0;
    return w;
}

int lhip_encode_batch(lhip_stream* const* streams, size_t nstreams, const int16_t* const* left, const int16_t* const* right,
                      const size_t* nsamples, uint8_t* const* out, const size_t* out_cap, int64_t* written) {
    // many streams with a lot of input: groups of streams go through the overlapped host path (copies behind the encode of the group before)
    if (nstreams > 1 && streams && left && nsamples && out && out_cap) {
        static const bool no_chunk = []() { const char* e = getenv("LAMEJS_HIP_NO_HOST_CHUNKS"); return e && e[0] == '1'; }();
        bool ok = !no_chunk;
        size_t total = 0;
        for (size_t i = 0; i < nstreams && ok; i++) {
            ok = streams[i] && streams[i]->magic == 0x4c484950 && streams[i]->ctx == streams[0]->ctx && streams[i]->ts.get() == streams[0]->ts.get() && left[i] && nsamples[i] > 0;
            if (ok) for (size_t k = 0; k < i; k++) if (streams[k] == streams[i]) { ok = false; break; }      // (a handle twice in one batch: the plain path reports it)
            total += nsamples[i];
        }
        if (ok && streams[0]->ts->T.disable_reservoir) {
            const Tables& T = streams[0]->ts->T;
            if (total > (size_t)4 * host_chunk_schedule().first * 576 * T.mode_gr * T.rs_ratio) return encode_host_groups(streams, nstreams, left, right, nsamples, out, out_cap, written);
        }
    }
    return encode_many(streams, nstreams, left, right, nsamples, out, out_cap, written, false, true);
}

int lhip_flush_batch(lhip_stream* const* streams, size_t nstreams, uint8_t* const* out, const size_t* out_cap, int64_t* written) {
    std::vector<std::vector<int16_t>> zs(nstreams);
    std::vector<const int16_t*> l(nstreams);
    std::vector<size_t> ns(nstreams);
    for (size_t i = 0; i < nstreams; i++) {
        if (!streams[i] || streams[i]->magic != 0x4c484950) { set_err("bad stream handle"); return LHIP_ERR_BAD_HANDLE; }
        ns[i] = flush_zeros(streams[i]);
        zs[i].assign(ns[i] ? ns[i] : 1, 0);
        l[i] = zs[i].data();
    }
    const int rc = encode_many(streams, nstreams, l.data(), l.data(), ns.data(), out, out_cap, written, false, true, true);
    if (rc >= 0) for (size_t i = 0; i < nstreams; i++) streams[i]->mf_samples_to_encode = 0;
    return rc;
}

int lhip_encode_batch_device(lhip_stream* const* streams, size_t nstreams, const int16_t* const* d_left, const int16_t* const* d_right,
                             const size_t* nsamples, uint8_t* const* d_out, const size_t* out_cap, int64_t* written, int sync) {
    return encode_many(streams, nstreams, d_left, d_right, nsamples, d_out, out_cap, written, true, sync != 0);
}

// ---- frame-range sharding of ONE stream (second mode): speculate the state at a cut, verify it, transplant on a miss ----
struct StateHdr { uint32_t magic, bytes; int32_t mf_size, mf_samples_to_encode, slot_lag, config; int64_t frame_num, rs_n_in; };
// what must agree between the stream a state blob came from and the stream it is put into
static int32_t state_config_tag(const Tables& T) {
    // FNV-1a over everything that selects a different table set or state layout (MPEG-2 and MPEG-2.5 share version and
    // samplerate_index, so the rates themselves are part of it)
    const int32_t f[] = {T.channels_out, T.mode, T.disable_reservoir != 0, T.samplerate_index, T.version, T.bitrate_index, T.rs_ratio,
                         T.out_samplerate, T.in_samplerate, T.brate, T.mode_gr, T.psy_channels};
    uint32_t h = 2166136261u;
    for (int32_t v : f) for (int b = 0; b < 4; b++) { h ^= (uint32_t)(v >> (8 * b)) & 0xffu; h *= 16777619u; }
    return (int32_t)h;
}
size_t lhip_state_bytes(const lhip_stream* s) {
    if (!s || s->magic != 0x4c484950) return 0;
    return sizeof(StateHdr) + sizeof(StreamState);
}
int lhip_state_get(lhip_stream* s, void* buf, size_t cap) {
    if (!s || s->magic != 0x4c484950) { set_err("bad stream handle"); return LHIP_ERR_BAD_HANDLE; }
    if (!buf || cap < lhip_state_bytes(s)) { set_err("state buffer too small"); return LHIP_ERR_BUFFER_TOO_SMALL; }
    Context* ctx = s->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!rt::set_device(ctx->device)) return LHIP_ERR_INTERNAL;
    StateHdr h; memset(&h, 0, sizeof h);
    h.magic = 0x5453484cu; h.bytes = (uint32_t)lhip_state_bytes(s);
    h.mf_size = s->mf_size; h.mf_samples_to_encode = s->mf_samples_to_encode; h.slot_lag = s->slot_lag; h.frame_num = s->frame_num; h.rs_n_in = s->rs_n_in;
    h.config = state_config_tag(s->ts->T);
    memcpy(buf, &h, sizeof h);
    if (!rt::set_device(ctx->device) || !rt::d2h((uint8_t*)buf + sizeof h, s->d_state, sizeof(StreamState), ctx->stream) || !rt::sync(ctx->stream)) return LHIP_ERR_INTERNAL;
    // Canonical form: fields no later launch can read are zeroed, so that "equal blobs = equal futures" also holds the other way round
    // for streams that reached the same point through different call sizes (kb_save never clears the samples beyond mf_size, the
    // resampler tail and the reservoir record are dead without resampler / reservoir).
    {
        const Tables& T = s->ts->T;
        StreamState* S = (StreamState*)((uint8_t*)buf + sizeof h);
        for (int ch = 0; ch < 2; ch++) {
            const int live = ch < T.channels_out ? s->mf_size : 0;
            for (int i = live; i < MF_NEEDED; i++) S->pcm_tail[ch][i] = 0.f;
        }
        if (T.rs_ratio == 1) memset(S->rs_old, 0, sizeof S->rs_old);
        if (T.disable_reservoir) { memset(S->nb1, 0, sizeof S->nb1); memset(S->nb2, 0, sizeof S->nb2); memset(&S->rv, 0, sizeof S->rv); }
    }
    return 0;
}
int lhip_state_set(lhip_stream* s, const void* buf, size_t n) {
    if (!s || s->magic != 0x4c484950) { set_err("bad stream handle"); return LHIP_ERR_BAD_HANDLE; }
    StateHdr h;
    if (!buf || n < sizeof h) { set_err("state blob too small"); return LHIP_ERR_INTERNAL; }
    memcpy(&h, buf, sizeof h);
    Context* ctx = s->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (h.magic != 0x5453484cu || h.bytes != lhip_state_bytes(s) || n < h.bytes || h.config != state_config_tag(s->ts->T)) { set_err("state blob does not belong to this build / configuration"); return LHIP_ERR_INTERNAL; }
    if (!rt::set_device(ctx->device)) return LHIP_ERR_INTERNAL;
    if (!rt::set_device(ctx->device) || !rt::h2d(s->d_state, (const uint8_t*)buf + sizeof h, sizeof(StreamState), ctx->stream) || !rt::sync(ctx->stream)) return LHIP_ERR_INTERNAL;
    s->mf_size = h.mf_size; s->mf_samples_to_encode = h.mf_samples_to_encode; s->slot_lag = h.slot_lag; s->frame_num = h.frame_num; s->rs_n_in = h.rs_n_in;
    return 0;
}
size_t lhip_seek_tail_samples(const lhip_stream* s) {
    if (!s || s->magic != 0x4c484950) return 0;
    return (size_t)(MF_INIT + 576 * s->ts->T.mode_gr);
}
// Put a FRESH stream where a stream that has consumed `sample_pos` input samples stands, as far as that is known without encoding:
// the buffered samples (the lhip_seek_tail_samples() samples in front of sample_pos), the frame counter and the padding
// accumulator.  Everything the encoder derives from earlier audio (masking history, filterbank overlap, attack / block-type
// chains, ATH adjustment, bin-search seeds) starts from its initial value and converges while a few warm-up frames are encoded.
int lhip_seek(lhip_stream* s, int64_t sample_pos, const int16_t* tail_left, const int16_t* tail_right) {
    if (!s || s->magic != 0x4c484950) { set_err("bad stream handle"); return LHIP_ERR_BAD_HANDLE; }
    const Tables& T = s->ts->T;
    const int frame = 576 * T.mode_gr, ntail = MF_INIT + frame;
    Context* ctx = s->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (T.rs_ratio != 1 || !T.disable_reservoir) { set_err("lhip_seek: not for resampling or bit-reservoir streams"); return LHIP_ERR_INTERNAL; }
    if (s->frame_num != 0 || s->mf_size != MF_INIT) { set_err("lhip_seek: the stream has been used"); return LHIP_ERR_INTERNAL; }
    if (sample_pos < 2 * frame || sample_pos % frame != 0 || !tail_left) { set_err("lhip_seek: position must be a whole number (>= 2) of frames"); return LHIP_ERR_INTERNAL; }
    if (T.channels_out == 2 && !tail_right) { set_err("lhip_seek: a two-channel stream needs both tails"); return LHIP_ERR_INTERNAL; }
    const bool do_scale = !(T.scale == 0.0) && !(T.scale == 1.0);
    std::vector<float> t((size_t)2 * MF_NEEDED, 0.f);
    for (int ch = 0; ch < T.channels_out; ch++) {
        const int16_t* src = (ch == 1 && tail_right) ? tail_right : tail_left;
        for (int i = 0; i < ntail; i++) { float v = (float)src[i]; if (do_scale) v = (float)((double)v * T.scale); t[(size_t)ch * MF_NEEDED + i] = v; }
    }
    if (!rt::set_device(ctx->device) || !rt::h2d((uint8_t*)s->d_state + offsetof(StreamState, pcm_tail), t.data(), sizeof(float) * 2 * MF_NEEDED, ctx->stream) || !rt::sync(ctx->stream)) return LHIP_ERR_INTERNAL;
    const int64_t k = sample_pos / frame;                     // the stream has emitted k - 1 frames
    s->frame_num = k - 1;
    s->rs_n_in = sample_pos;
    s->mf_size = ntail;
    s->mf_samples_to_encode = 576 + 1152 + frame;
    if (T.frac_SpF != 0) {
        int64_t m = ((int64_t)T.frac_SpF - (k - 1) * (int64_t)T.frac_SpF) % T.out_samplerate;     // slot_lag starts at frac_SpF, one decrement per frame
        if (m < 0) m += T.out_samplerate;
        s->slot_lag = (int)m;
    }
    return 0;
}

// Test hook (aliased contexts, LHIP_ALIAS_DEVICES): gives back what context `device` holds beyond its streams' lifetime -- the HIP stream the library created
// for it, the side stream and events of the ATH scan -- while the runtime is up.  Only with no live stream on the context.
int lhip_debug_release_context(int device) {
#ifndef LHIP_HOSTSIM
    Context* ctx = get_context(device);
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->live_streams != 0) { set_err("lhip_debug_release_context: the context still has streams"); return LHIP_ERR_INTERNAL; }
    if (!rt::set_device(ctx->device)) return LHIP_ERR_INTERNAL;
    (void)hipStreamSynchronize((hipStream_t)ctx->stream);
    WorkSet& ws = ctx->ws;
    if (ws.aux_stream) { (void)hipStreamSynchronize((hipStream_t)ws.aux_stream); (void)hipStreamDestroy((hipStream_t)ws.aux_stream); (void)hipEventDestroy((hipEvent_t)ws.ev_fork); (void)hipEventDestroy((hipEvent_t)ws.ev_join); ws.aux_stream = ws.ev_fork = ws.ev_join = nullptr; }
    if (ctx->own_stream) { (void)hipStreamDestroy((hipStream_t)ctx->stream); ctx->stream = nullptr; ctx->own_stream = false; }
#else
    (void)device;
#endif
    return 0;
}

int lhip_set_hip_stream(int device, void* hip_stream) {
#ifndef LHIP_HOSTSIM
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return LHIP_ERR_INTERNAL; }
#else
    if (device < 0) device = 0;
#endif
    Context* ctx = get_context(device);
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->stream = hip_stream;
    return 0;
}

void lhip_last_batch_stats(int64_t* frames, int64_t* repaired_frames, int64_t* repair_iterations) {
#ifndef LHIP_HOSTSIM
    if (g_stat_pending) {                      // asynchronous batch: wait for it and fetch the device-side counters
        Context* ctx = g_stat_pending;
        g_stat_pending = nullptr;
        std::lock_guard<std::mutex> lk(ctx->mu);
        int32_t fx[3] = {0, 0, 0};
        WorkSet& ws = ctx->ws;
        void* st = ctx->stream;
        if (rt::set_device(ctx->device) && rt::d2h(fx, (const int32_t*)ws.nflagged.p + FX_STATS, sizeof fx, st) && rt::sync(st)) {
            g_stat_repaired = fx[0]; g_stat_iters = fx[1];
            if (fx[2]) set_err("seed-chain repair did not converge");
        }
    }
#endif
    if (frames) *frames = g_stat_frames;
    if (repaired_frames) *repaired_frames = g_stat_repaired;
    if (repair_iterations) *repair_iterations = g_stat_iters;
}

int64_t lhip_debug_read(int what, void* dst, size_t cap) {
#ifdef LHIP_PHASE_PROF
    if (what == 9) { const size_t n = cap < sizeof g_call_prof ? cap : sizeof g_call_prof; memcpy(dst, g_call_prof, n); return (int64_t)n; }
#endif
    Context* ctx = nullptr;
    { std::lock_guard<std::mutex> lk(g_ctx_mu); for (auto& kv : g_ctx) if (kv.second->ws.have_last) ctx = kv.second.get(); }
    if (!ctx) { set_err("no batch has run"); return LHIP_ERR_INTERNAL; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    rt::set_device(ctx->device);
    const WorkSet& ws = ctx->ws;
    const Workspace& W = ws.lastW;
    const size_t GC = (size_t)W.ngslots * ws.lastC;
    const void* src = nullptr; size_t n = 0;
    switch (what) {
        case 0: src = W.xr; n = GC * 576 * 4; break;
        case 1: src = W.blocktype; n = GC * 4; break;
        case 2: src = W.E; n = (size_t)W.ngslots * ws.lastCp * E_STRIDE * 4; break;
        case 3: src = W.ath_adjust; n = (size_t)W.nfslots * 8; break;
        case 4: src = W.side; n = (size_t)W.nframes_total * 2 * ws.lastC * sizeof(GrSide); break;
        case 5: src = W.sb; n = GC * SB_STRIDE * 4; break;
        case 6: src = W.peaks; n = GC * PK_STRIDE * 4; break;
        case 7: src = W.prof; n = PROF_BYTES; break;
        case 8: src = W.nflagged; n = 256; break;           // validation counters of the last batch: [0] frames flagged by the first pass, [1] frames its memo could not decide, [32..34] repair statistics
        default: set_err("unknown tap"); return LHIP_ERR_INTERNAL;
    }
    if (n > cap) n = cap;
    if (!rt::d2h(dst, src, n, ctx->stream) || !rt::sync(ctx->stream)) return LHIP_ERR_INTERNAL;
    return (int64_t)n;
}

int lhip_kernel_timing(int enable) {
#ifndef LHIP_HOSTSIM
    std::lock_guard<std::mutex> lk(g_kt_mu);
    g_kt_on = enable != 0;
    for (int i = 0; i < KT_N; i++) { g_kt_ms[i] = 0; g_kt_calls[i] = 0; }
    return KT_N;
#else
    (void)enable; return 0;
#endif
}

int lhip_kernel_times(int idx, const char** name, double* total_ms, int64_t* launches) {
#ifndef LHIP_HOSTSIM
    if (idx < 0 || idx >= KT_N) return -1;
    std::lock_guard<std::mutex> lk(g_kt_mu);
    if (name) *name = g_kt_names[idx];
    if (total_ms) *total_ms = g_kt_ms[idx];
    if (launches) *launches = g_kt_calls[idx];
    return 0;
#else
    (void)idx; (void)name; (void)total_ms; (void)launches; return -1;
#endif
}

int lhip_debug_set_spec_seed(int start, int step) {
    if (start < 0 || start > 255 || step < 1 || step > 255) return LHIP_ERR_INTERNAL;
    g_spec_start = start; g_spec_step = step;
    return 0;
}

int lhip_debug_math(int op, const double* in, double* out, size_t n) {
#ifndef LHIP_HOSTSIM
    double *din = nullptr, *dout = nullptr;
    if (hipMalloc((void**)&din, n * 8) != hipSuccess || hipMalloc((void**)&dout, n * 8) != hipSuccess) { set_err("hipMalloc failed"); return LHIP_ERR_INTERNAL; }
    hipMemcpy(din, in, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(g_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, din, dout, n, pow_log2_parts(10.0));
    hipError_t e = hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost);
    hipFree(din); hipFree(dout);
    if (e != hipSuccess) { set_err(hipGetErrorString(e)); return LHIP_ERR_INTERNAL; }
    return 0;
#else
    const PowBase pb = pow_log2_parts(10.0);
    if (op == 8) { for (size_t i = 0; 21 * i + 21 <= n; i++) math_op8(in + 21 * i, out + 21 * i); return 0; }
    if (op == 10) { for (size_t i = 0; 2 * i + 2 <= n; i++) math_op10(in + 2 * i, out + 2 * i); return 0; }
    for (size_t i = 0; i < n; i++) {
        const double x = in[i];
        switch (op) {
            case 0: out[i] = v8_log10(x); break;
            case 1: out[i] = v8_pow_base(pb, x); break;
            case 2: out[i] = d_sqrt(x); break;
            case 3: out[i] = 1.0 / x; break;
            case 4: out[i] = (double)(float)x; break;
            case 5: out[i] = (double)js_toint32(x); break;
            case 7: out[i] = v8_log10_pos(x); break;
            case 11: out[i] = (double)ma_index16(x) + 1000.0 * (double)js_toint32(v8_log10_pos(x) * 16.0); break;
            case 9: { const double l = v8_log10_pos(x > 1E-20 ? x : 1E-20);
                      out[i] = (double)noise_class(x) + 1000.0 * (double)noise_class_of_log(l) + 1e6 * (double)noise_class_of_log((double)(float)l); } break;
            default: out[i] = x / 3.0 + x * 0.1; break;
        }
    }
    return 0;
#endif
}

}  // extern "C"
