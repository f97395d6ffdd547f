// Psychoacoustic model kernels (nspsytune), decomposed per SURVEY.md 3.4:
//   kb_psyA  : per (psy call, channel) -- everything that is a pure function of the PCM window:
//              fs/4 high-pass + sub-block peaks, 1024-pt and 3x256-pt windowed FHT, energies,
//              loudness, partition energies / tonality index, short-block spreading.
//   kb_scan  : per stream -- the two tiny recurrences (attack/block-type chain, ATH auto-adjust).
//   kb_psyB  : per psy call -- thresholds that need the scan results (additive masking with the
//              adjusted ATH, short-block limiting, sfb mapping, inter-channel masking).
// Behaviour follows reference PsyModel.js:1000-1383 (L3psycho_anal_ns), FFT.js:31-224 and
// Encoder.js:166-243 (adjust_ATH); line references are given at each step.
#pragma once
#include "lhip_defs.h"
#include "lhip_wave.h"
#include "lhip_math.h"
#include "lhip_layout.h"

namespace lhip {

// ---------------------------------------------------------------------------------------------
// FHT butterflies, one radix-4 pass over n points held in LDS (FFT.js:45-112).
// Work item t in [0, n/8): block m = t / kx, index i = t % kx inside the block.
// ---------------------------------------------------------------------------------------------
// The i == 0 butterfly of block m (no twiddles) and the i >= 1 butterflies are separate work lists, so that a
// wave never executes both code paths for one batch of items.
// LDS layout of a transform buffer while the passes run: one pad word after every 32 (index a lives at a + a / 32).  The butterflies of the
// first passes touch, lane after lane, addresses 16 (k1 = 4) or 64 (k1 = 16) words apart: in a plain array that is 2 resp. 7 of the 32 LDS
// banks for a whole wave -- 72 % of g_psyA's LDS cycles were bank conflicts (profiles/r03_pmc_lds_config3.json) -- with the pad the 64 lanes
// of the k1 = 4 pass spread over all 32 banks.  Everything that reads or writes transform DATA goes through fzp(): the windowing's outputs,
// the passes, the energies' operands, the joint-stereo copies; the other phases use the same memory as plain scratch, and every transition
// between the two views already goes through registers and a wave_sync.
#ifdef LHIP_NO_FZ_PAD      /* A/B builds: the plain layout */
LHIP_DEV int fzp(int a) { return a; }
LHIP_DEV int fzo(int c) { return c; }
#else
LHIP_DEV int fzp(int a) { return a + (a >> 5); }
// offsets inside an item: index + c lands at fzp(index) + c + c / 32 because (index & 31) + (c & 31) < 32 for every operand of every pass
// (k1 = 4: a 16-aligned group; k1 = 16: i < 16 and c & 31 is 0 or 16; k1 >= 64: c is a multiple of 32)
LHIP_DEV int fzo(int c) { return c + (c >> 5); }
#endif
LHIP_DEV void fht_item0(float* fz, int base, int k1, int kx, int m) {
    const int k2 = k1 << 1, k3 = k2 + k1, k4 = k2 << 1;
    const int K1 = fzo(k1), K2 = fzo(k2), K3 = fzo(k3);
    float* fi = fz + fzp(base + m * k4);
    float* gi = fz + fzp(base + m * k4 + kx);
    double f0, f1, f2, f3;
    f1 = (double)fi[0] - (double)fi[K1];
    f0 = (double)fi[0] + (double)fi[K1];
    f3 = (double)fi[K2] - (double)fi[K3];
    f2 = (double)fi[K2] + (double)fi[K3];
    fi[K2] = (float)(f0 - f2);
    fi[0] = (float)(f0 + f2);
    fi[K3] = (float)(f1 - f3);
    fi[K1] = (float)(f1 + f3);
    f1 = (double)gi[0] - (double)gi[K1];
    f0 = (double)gi[0] + (double)gi[K1];
    f3 = LHIP_SQRT2 * (double)gi[K3];
    f2 = LHIP_SQRT2 * (double)gi[K2];
    gi[K2] = (float)(f0 - f2);
    gi[0] = (float)(f0 + f2);
    gi[K3] = (float)(f1 - f3);
    gi[K1] = (float)(f1 + f3);
}
LHIP_DEV void fht_item1(float* fz, int base, int k1, int kx, int m, int i, const double* tw) {
    const int k2 = k1 << 1, k3 = k2 + k1, k4 = k2 << 1;
    const int K1 = fzo(k1), K2 = fzo(k2), K3 = fzo(k3);
    (void)kx;
    float* fi = fz + fzp(base + m * k4 + i);
    float* gi = fz + fzp(base + m * k4 + k1 - i);
    const double c1 = tw[4 * (i - 1) + 0], s1 = tw[4 * (i - 1) + 1], c2 = tw[4 * (i - 1) + 2], s2 = tw[4 * (i - 1) + 3];
    double a, b, g0, f0, f1, g1, f2, g2, f3, g3;
    b = s2 * (double)fi[K1] - c2 * (double)gi[K1];
    a = c2 * (double)fi[K1] + s2 * (double)gi[K1];
    f1 = (double)fi[0] - a;
    f0 = (double)fi[0] + a;
    g1 = (double)gi[0] - b;
    g0 = (double)gi[0] + b;
    b = s2 * (double)fi[K3] - c2 * (double)gi[K3];
    a = c2 * (double)fi[K3] + s2 * (double)gi[K3];
    f3 = (double)fi[K2] - a;
    f2 = (double)fi[K2] + a;
    g3 = (double)gi[K2] - b;
    g2 = (double)gi[K2] + b;
    b = s1 * f2 - c1 * g3;
    a = c1 * f2 + s1 * g3;
    fi[K2] = (float)(f0 - a);
    fi[0] = (float)(f0 + a);
    gi[K3] = (float)(g1 - b);
    gi[K1] = (float)(g1 + b);
    b = c1 * g2 - s1 * f3;
    a = s1 * g2 + c1 * f3;
    gi[K2] = (float)(g0 - a);
    gi[0] = (float)(g0 + a);
    fi[K3] = (float)(f1 - b);
    fi[K1] = (float)(f1 + b);
}
// one radix-4 pass over n points: the n/8 work items are the n/(8 kx) twiddle-free ones and the rest
LHIP_DEV void fht_pass(float* fz, int n, int k1, int kx, int lane, int nl, int base, const double* tw) {
    const int items = n / 8, nz = items / kx, kx1 = kx - 1;
    for (int t = lane - base; t < nz; t += nl) if (t >= 0) fht_item0(fz, 0, k1, kx, t);
    for (int t = lane - base; t < items - nz; t += nl) if (t >= 0) { const int m = t / kx1; fht_item1(fz, 0, k1, kx, m, 1 + t - m * kx1, tw); }
}

// the same pass over `nb` equally long transforms stored back to back (ONE padded buffer), as ONE work list: three 256-point transforms have
// 32 items each per pass -- one list of 96 fills the lanes where three lists of 32 leave half of them idle
LHIP_DEV void fht_pass_blocks(float* fz, int nb, int n, int k1, int kx, int lane, int nl, const double* tw) {
    const int items = n / 8, nz = items / kx, kx1 = kx - 1, n1 = items - nz;
    for (int t = lane; t < nb * nz; t += nl) { const int b = t / nz; fht_item0(fz, b * n, k1, kx, t - b * nz); }
    for (int t = lane; t < nb * n1; t += nl) {
        const int b = t / n1, r = t - b * n1, m = r / kx1;
        fht_item1(fz, b * n, k1, kx, m, 1 + r - m * kx1, tw);
    }
}

// LDS of one psy-A wave.  The energies overwrite the lower halves of the FHT buffers they are computed from
// (fe = fz[0..512], fes[b] = fs[b][0..128]) and the partition arrays live in the then-dead upper half of fz:
// 7.2 KB instead of 12.3 KB per wave, i.e. LDS no longer caps the kernel at 3 waves per SIMD.
struct PsyALds {
    float fz[BLKSIZE + BLKSIZE / 32];              // long FHT buffer (padded layout, fzp); after the energies: [0..512] fe, [516..] eb/mx/av/ebs
    float fs[3][BLKSIZE_s];                        // short FHT buffers: one padded buffer of 3 x 256 points (fzp over the flat index b * 256 + j) ...
    float fs_pad[3 * BLKSIZE_s / 32];              // ... which therefore extends into this; after the energies: fs[b][0..128] = fes
};
static_assert(offsetof(PsyALds, fs_pad) == offsetof(PsyALds, fs) + sizeof(float) * 3 * BLKSIZE_s, "the short transform buffer is one flat padded array");
#define PSYA_FE(L) ((L).fz)
#define PSYA_FES(L, b) ((L).fs[b])
#define PSYA_EB(L) ((L).fz + 516)
#define PSYA_MX(L) ((L).fz + 516 + CBANDS)
#define PSYA_AV(L) ((L).fz + 516 + 2 * CBANDS)
#define PSYA_EBS(L, b) ((L).fz + 516 + (3 + (b)) * CBANDS)

#if defined(LHIP_PHASE_PROF) && !defined(LHIP_HOSTSIM)
#define PSY_STAMP(i) psy_t_[i] = __builtin_amdgcn_s_memtime()      /* stamp 2 sits inside the `ch < 2` branch: the mid / side waves keep stamp 1's time there */
#define PSY_FLUSH() do { if (lane == 0 && ((gslot & 63) == 0 || (W.ngslots <= 4 && ch == 0 && gslot - sd.gslot0 == 1))) {   /* a sample of the waves: a flush per wave congests what it measures */ if (psy_t_[2] == 0) psy_t_[2] = psy_t_[1]; \
    for (int i_ = 0; i_ < 7; i_++) atomicAdd((unsigned long long*)W.prof + 22 + i_, psy_t_[i_ + 1] - psy_t_[i_]); atomicAdd((unsigned long long*)W.prof + 54, 1ull); } } while (0)
#define PSY_DECL() unsigned long long psy_t_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#else
#define PSY_STAMP(i) do {} while (0)
#define PSY_FLUSH() do {} while (0)
#define PSY_DECL() do {} while (0)
#endif
// (float)((sum of the LHIP_NL * K non-negative values, lane l holding [l K, (l + 1) K)) * scale + add) when that Float32 does not depend
// on the order of the additions: *out is set and 1 returned if both ends of the error band of an any-order sum round to the same
// Float32 (wave-uniform verdict); 0: the caller must form the sum in the reference's order.  scale > 0, add >= 0.
template <int K> LHIP_DEV int guarded_f32_of_sum(const double (&p)[K], double scale, double add, float* out) {
#if LHIP_NL == 1
    (void)p; (void)scale; (void)add; (void)out;
    return 0;                                         // the one-lane build always takes the sequential sum
#else
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < K; k++) s += p[k];
    s = wave_sumd(s);
    // any-order sum vs the reference's sequential one: both within (n - 1) u (n = 512, u = 2^-53) of the exact sum, then one rounding
    // each for `* scale` and `+ add` on either side: 2e-13 covers it with a margin of 1.7
    const float lo = (float)((s * (1.0 - 2e-13)) * scale + add), hi = (float)((s * (1.0 + 2e-13)) * scale + add);
    *out = lo;
#ifdef LHIP_WAVESIM
    if (lo == hi) {       // the simulation checks the claim on everything it encodes: the guarded value IS the sequential sum's
        const float ref = (float)(wave_seq_sum<K>(p) * scale + add);
        if (ref != lo) { fprintf(stderr, "wavesim: guarded_f32_of_sum disagrees with the sequential sum (%a vs %a)\n", (double)lo, (double)ref); abort(); }
    }
#endif
    return lo == hi;
#endif
}

// one wave per (granule slot >= 1 of a stream, psy channel).  ch = 0, 1: L, R.  Joint stereo adds ch = 2, 3 (mid, side) in a second
// launch: their high-passed samples and their spectra are linear combinations of the L / R ones (PsyModel.js:1113-1121, 258-273),
// which the L / R waves leave in W.hpf / W.fht; everything from the energies on is the same code for all four.
// PART (one-frame launches, where a wave is alone with its latencies and other waves of the workgroup idle): 1 = only the high-pass + sub-block peaks
// (they need the samples and nothing else, and nothing below needs them), 2 = everything else; 0 = all of it on one wave (the batched kernel);
// 3 = everything else up to the loudness (what the scans and the search of the frame's first granule need), 4 = the rest -- partition energies, tonality,
// short spreading: psyB's inputs -- later, on the same wave and the same LDS record (nothing but LDS carries over).
template <int PART = 0>
LHIP_DEV void kb_psyA(const Tables& T, const Workspace& W, const StreamDesc* SD, const StreamIO* IO, int gslot, int ch, int lane, PsyALds& L) {
    const int C = T.channels_out, Cp = T.psy_channels;
    const int st = W.gslot_stream[gslot];
    const StreamDesc sd = SD[st];
    const int q = gslot - sd.gslot0 - 1;              // local psy call index
    if (q < 0) return;                                // carry slot: nothing to compute
    // The call's 1024-sample window (segment index 576 q + 304 onwards) is converted ONCE into the long FHT buffer: the high-pass,
    // the short windowing and the long windowing all read it from LDS; the long windowing then runs in place (every lane takes its
    // samples into registers before anybody writes).  The caller's Int16 is read coalesced, 2 bytes per sample, exactly once.
    if (PART != 4 && ch < 2) {
        const PcmSrc P = pcm_source(T, W, sd, IO[st], ch);
        const int b0 = 576 * q + 304;
        if (!P.plane && b0 >= P.mf) {
            // the usual case, wave-uniform: the whole window lies in this call's new samples -- no per-sample decisions
            const int16_t* src = P.src + (b0 - P.mf);
            for (int i = lane; i < BLKSIZE; i += LHIP_NL) {
                float v = (float)src[i];
                if (P.do_scale) v = (float)((double)v * P.scale);
                L.fz[i] = v;
            }
        } else {
            for (int i = lane; i < BLKSIZE; i += LHIP_NL) L.fz[i] = pcm_at(P, b0 + i);
        }
        wave_sync();
    }
#define buf(i) L.fz[i]
    const int64_t o = (int64_t)gslot * Cp + ch;
    PSY_DECL();
    PSY_STAMP(0);

    if (PART == 4) goto psya_tail;
    // --- fs/4 high-pass, 9 sub-block peaks (PsyModel.js:1051-1069, 1122-1132) ---
    if (PART == 0 || PART == 1) {
        const float* fir = L.fz + 397;                // 576 - 350 - 21 + 192
        // Each lane filters NINE CONSECUTIVE outputs: they share 30 input samples, read and widened once (a lane that took
        // output lane + 64 k of each sub-block instead would read 198).  The magnitudes go through LDS (the short FHT buffers are
        // free until the windowing) so that lane l then holds value l of each 64-sample sub-block for the nine wave maxima.
        // Per output the sums are formed exactly as in PsyModel.js:1051-1069 (same operands, same order).
        enum { HO = (576 + LHIP_NL - 1) / LHIP_NL };      // outputs per lane: 9
        float* mag = &L.fs[0][0];
        if (ch >= 2) {
            const float* hl = W.hpf + (int64_t)gslot * 2 * 576;
            for (int i = lane; i < 576; i += LHIP_NL) {
                const double l = hl[i], r = hl[576 + i];
                float v = (ch == 2) ? (float)(l + r) : (float)(l - r);
                mag[i] = v < 0 ? -v : v;
            }
        } else {
            double x[HO + 21];
#pragma unroll
            for (int t = 0; t < HO + 21; t++) { const int n = HO * lane + t; x[t] = (n < 576 + 21) ? (double)fir[n] : 0.0; }
#pragma unroll
            for (int k = 0; k < HO; k++) {
                double sum1 = x[k + 10], sum2 = 0.0;
#pragma unroll
                for (int j = 0; j < 9; j += 2) {
                    sum1 += T.hpf_fircoef[j] * (x[k + j] + x[k + 21 - j]);
                    sum2 += T.hpf_fircoef[j + 1] * (x[k + j + 1] + x[k + 21 - j - 1]);
                }
                float v = (float)(sum1 + sum2);
                if (Cp == 4 && HO * lane + k < 576) W.hpf[((int64_t)gslot * 2 + ch) * 576 + HO * lane + k] = v;
                v = v < 0 ? -v : v;
                if (HO * lane + k < 576) mag[HO * lane + k] = v;
            }
        }
        wave_sync();
        enum { KP = (64 + LHIP_NL - 1) / LHIP_NL };
        float pk[9];
#pragma unroll
        for (int sbk = 0; sbk < 9; sbk++) {
            float m = 1.0f;
            for (int u = 0; u < KP; u++) { const float v = mag[sbk * 64 + lane + LHIP_NL * u]; if (m < v) m = v; }
            pk[sbk] = m;
        }
        wave_sync();                                  // the magnitudes are dead: the short windowing refills L.fs
        {   // nine maxima of values >= 1: IEEE order == integer order of the bit patterns; reduced side by side (one DPP step serves all nine)
            int pb[9];
#pragma unroll
            for (int sbk = 0; sbk < 9; sbk++) __builtin_memcpy(&pb[sbk], &pk[sbk], 4);
            wave_max_n(pb);
#pragma unroll
            for (int sbk = 0; sbk < 9; sbk++) __builtin_memcpy(&pk[sbk], &pb[sbk], 4);
        }
        if (lane == 0) {
#pragma unroll
            for (int sbk = 0; sbk < 9; sbk++) W.peaks[o * PK_STRIDE + sbk] = pk[sbk];
        }
    }
    if (PART == 1) return;

    PSY_STAMP(1);
    if (ch >= 2) {
        // mid / side spectra from the L / R ones (compute_ffts, PsyModel.js:258-273): (l + r) * SQRT2 * 0.5, rounded to f32
        const float* fl = W.fht + (int64_t)gslot * 2 * FHT_STRIDE;
        for (int i = lane; i < FHT_STRIDE; i += LHIP_NL) {
            const double l = fl[i], r = fl[FHT_STRIDE + i];
            const float v = (ch == 2) ? (float)((l + r) * LHIP_SQRT2 * 0.5) : (float)((l - r) * LHIP_SQRT2 * 0.5);
            if (i < BLKSIZE) L.fz[fzp(i)] = v; else (&L.fs[0][0])[fzp(i - BLKSIZE)] = v;
        }
        wave_sync();
    } else {
    // --- windowing + first radix-4 stage (FFT.js:185-221 long, 140-180 short) ---
    for (int it = lane; it < 3 * (BLKSIZE_s / 8); it += LHIP_NL) {
        const int b = it / (BLKSIZE_s / 8), j = it - b * (BLKSIZE_s / 8);
        const int k = (576 / 3) * (b + 1);
        const int i = T.fft_rv_tbl[j << 2] & 0xff;
        float* x = &L.fs[0][0] + fzp(b * BLKSIZE_s + 4 * j);        // padded layout: x[0..3] stay together, the second group sits 128 + 4 words on
        double f0, f1, f2, f3, w;
        f0 = (double)T.window_s[i] * (double)buf(i + k);
        w = (double)T.window_s[0x7f - i] * (double)buf(i + k + 0x80);
        f1 = f0 - w; f0 = f0 + w;
        f2 = (double)T.window_s[i + 0x40] * (double)buf(i + k + 0x40);
        w = (double)T.window_s[0x3f - i] * (double)buf(i + k + 0xc0);
        f3 = f2 - w; f2 = f2 + w;
        x[0] = (float)(f0 + f2); x[2] = (float)(f0 - f2); x[1] = (float)(f1 + f3); x[3] = (float)(f1 - f3);
        f0 = (double)T.window_s[i + 0x01] * (double)buf(i + k + 0x01);
        w = (double)T.window_s[0x7e - i] * (double)buf(i + k + 0x81);
        f1 = f0 - w; f0 = f0 + w;
        f2 = (double)T.window_s[i + 0x41] * (double)buf(i + k + 0x41);
        w = (double)T.window_s[0x3e - i] * (double)buf(i + k + 0xc1);
        f3 = f2 - w; f2 = f2 + w;
        x[fzo(BLKSIZE_s / 2) + 0] = (float)(f0 + f2); x[fzo(BLKSIZE_s / 2) + 2] = (float)(f0 - f2);
        x[fzo(BLKSIZE_s / 2) + 1] = (float)(f1 + f3); x[fzo(BLKSIZE_s / 2) + 3] = (float)(f1 - f3);
    }
    {   // long window, in place: all samples of the lane's items first, then -- after everybody has read -- the butterflies
        enum { KL = (BLKSIZE / 8 + LHIP_NL - 1) / LHIP_NL };
        float xin[KL][8];
#pragma unroll
        for (int u = 0; u < KL; u++) {
            const int jj = lane + LHIP_NL * u;
            if (jj < BLKSIZE / 8) {
                const int i = T.fft_rv_tbl[jj] & 0xff;
                xin[u][0] = buf(i); xin[u][1] = buf(i + 0x200); xin[u][2] = buf(i + 0x100); xin[u][3] = buf(i + 0x300);
                xin[u][4] = buf(i + 0x001); xin[u][5] = buf(i + 0x201); xin[u][6] = buf(i + 0x101); xin[u][7] = buf(i + 0x301);
            }
        }
        wave_sync();
#pragma unroll
        for (int u = 0; u < KL; u++) {
            const int jj = lane + LHIP_NL * u;
            if (jj < BLKSIZE / 8) {
                const int i = T.fft_rv_tbl[jj] & 0xff;
                float* x = L.fz + fzp(4 * jj);
                double f0, f1, f2, f3, w;
                f0 = (double)T.window[i] * (double)xin[u][0];
                w = (double)T.window[i + 0x200] * (double)xin[u][1];
                f1 = f0 - w; f0 = f0 + w;
                f2 = (double)T.window[i + 0x100] * (double)xin[u][2];
                w = (double)T.window[i + 0x300] * (double)xin[u][3];
                f3 = f2 - w; f2 = f2 + w;
                x[0] = (float)(f0 + f2); x[2] = (float)(f0 - f2); x[1] = (float)(f1 + f3); x[3] = (float)(f1 - f3);
                f0 = (double)T.window[i + 0x001] * (double)xin[u][4];
                w = (double)T.window[i + 0x201] * (double)xin[u][5];
                f1 = f0 - w; f0 = f0 + w;
                f2 = (double)T.window[i + 0x101] * (double)xin[u][6];
                w = (double)T.window[i + 0x301] * (double)xin[u][7];
                f3 = f2 - w; f2 = f2 + w;
                x[fzo(BLKSIZE / 2) + 0] = (float)(f0 + f2); x[fzo(BLKSIZE / 2) + 2] = (float)(f0 - f2);
                x[fzo(BLKSIZE / 2) + 1] = (float)(f1 + f3); x[fzo(BLKSIZE / 2) + 3] = (float)(f1 - f3);
            }
        }
    }
    wave_sync();

    PSY_STAMP(2);
    // --- remaining FHT passes; twiddle table offsets 0,1,8,39 (pass t has kx-1 entries) ---
    {
        int off = 0;
#pragma unroll
        for (int k1 = 4, kx = 2; k1 < BLKSIZE; k1 <<= 2, kx <<= 2) {
            fht_pass(L.fz, BLKSIZE, k1, kx, lane, LHIP_NL, 0, T.fht_twiddle + 4 * off);
            if (k1 < BLKSIZE_s) fht_pass_blocks(&L.fs[0][0], 3, BLKSIZE_s, k1, kx, lane, LHIP_NL, T.fht_twiddle + 4 * off);
            wave_sync();
            off += kx - 1;
        }
    }
    if (Cp == 4) {                                    // joint stereo: the spectra the mid / side waves combine
        float* fo = W.fht + ((int64_t)gslot * 2 + ch) * FHT_STRIDE;
        for (int i = lane; i < FHT_STRIDE; i += LHIP_NL) fo[i] = (i < BLKSIZE) ? L.fz[fzp(i)] : (&L.fs[0][0])[fzp(i - BLKSIZE)];
    }
    }   // ch < 2

    PSY_STAMP(3);
    // --- energies (PsyModel.js:274-296), written over the lower halves of the transform buffers ---
    {
        enum { KE = (BLKSIZE / 2 + 1 + LHIP_NL - 1) / LHIP_NL, KS = (3 * (BLKSIZE_s / 2 + 1) + LHIP_NL - 1) / LHIP_NL };
        float el[KE], es[KS];
#pragma unroll
        for (int u = 0; u < KE; u++) {
            const int j = lane + LHIP_NL * u;
            el[u] = 0.f;
            if (j == 0) { const float e0 = L.fz[0]; el[u] = (float)((double)e0 * (double)e0); }
            else if (j <= BLKSIZE / 2) { const double re = L.fz[fzp(j)], im = L.fz[fzp(BLKSIZE - j)]; el[u] = (float)((re * re + im * im) * 0.5); }
        }
#pragma unroll
        for (int u = 0; u < KS; u++) {
            const int it = lane + LHIP_NL * u;
            es[u] = 0.f;
            if (it < 3 * (BLKSIZE_s / 2 + 1)) {
                const int b = it / (BLKSIZE_s / 2 + 1), j = it - b * (BLKSIZE_s / 2 + 1);
                const float* fsf = &L.fs[0][0];
                if (j == 0) { const float e0 = fsf[fzp(b * BLKSIZE_s)]; es[u] = (float)((double)e0 * (double)e0); }
                else { const double re = fsf[fzp(b * BLKSIZE_s + j)], im = fsf[fzp(b * BLKSIZE_s + BLKSIZE_s - j)]; es[u] = (float)((re * re + im * im) * 0.5); }
            }
        }
        wave_sync();                                  // every lane has read its operands before anything is overwritten
#pragma unroll
        for (int u = 0; u < KE; u++) { const int j = lane + LHIP_NL * u; if (j <= BLKSIZE / 2) PSYA_FE(L)[j] = el[u]; }
#pragma unroll
        for (int u = 0; u < KS; u++) {
            const int it = lane + LHIP_NL * u;
            if (it < 3 * (BLKSIZE_s / 2 + 1)) { const int b = it / (BLKSIZE_s / 2 + 1), j = it - b * (BLKSIZE_s / 2 + 1); PSYA_FES(L, b)[j] = es[u]; }
        }
    }
    wave_sync();

    PSY_STAMP(4);
    // --- loudness: strictly sequential f64 sum (PsyModel.js:241-249), products in parallel, ordered fold ---
    {
        enum { K = (BLKSIZE / 2) / LHIP_NL };
        double pr[K];
#pragma unroll
        for (int k = 0; k < K; k++) { const int i = K * lane + k; pr[k] = (double)PSYA_FE(L)[i] * (double)T.eql_w[i]; }
        // The reference's sum is strictly sequential, and only its Float32 rounding is ever read.  All terms are >= 0, so a sum in ANY
        // order lies within (n - 1) u of the exact sum, as the reference's own does: the two differ by less than 1.2e-13 (relative).
        // A tree sum (8 adds per lane + a wave reduction instead of 64 hand-over steps) therefore rounds to the reference's Float32
        // whenever both ends of its 2e-13 error band round to the same Float32; when they do not (about one call in 10^5) the
        // sequential fold decides.  (T.eql_w >= 0 is checked where the table is loaded.)
        if (ch < 2) {                                 // no loudness for mid / side (PsyModel.js:319)
            float lf;
            if (!guarded_f32_of_sum<K>(pr, T.VO_SCALE, 0.0, &lf)) lf = (float)(wave_seq_sum<K>(pr) * T.VO_SCALE);
            if (lane == 0) W.loud[(int64_t)gslot * C + ch] = lf;
        }
        if (Cp == 4) {
            // total energy: lines 11 .. 512 summed in ascending order (PsyModel.js:300-307); the leading zeros change nothing
#pragma unroll
            for (int k = 0; k < K; k++) { const int i = K * lane + k; pr[k] = (i >= 11) ? (double)PSYA_FE(L)[i] : 0.0; }
            const double last = (double)PSYA_FE(L)[BLKSIZE / 2];
            float tf;
            if (!guarded_f32_of_sum<K>(pr, 1.0, last, &tf)) tf = (float)(wave_seq_sum<K>(pr) + last);
            if (lane == 0) W.tot_ener[(int64_t)gslot * 4 + ch] = tf;
        }
    }

    PSY_STAMP(5);
    if (PART == 3) return;
psya_tail:
    // --- long partitions: energy, max, average (calc_energy, PsyModel.js:906-928) ---
#if LHIP_NL == 1
    LHIP_LANE_ONCE(b, 0, T.npart_l) {                        // npart_l < CBANDS = 64
        double ebb = 0, m = 0;
        int j = T.lineoff_l[b];
        for (int i = 0; i < T.numlines_l[b]; ++i, ++j) {
            const double el = PSYA_FE(L)[j];
            ebb += el;
            if (m < el) m = el;
        }
        PSYA_EB(L)[b] = (float)ebb;
        PSYA_MX(L)[b] = (float)m;
        PSYA_AV(L)[b] = (float)(ebb * (double)T.rnumlines_l[b]);
    }
#else
    {
        // The partitions tile the 513 lines, and a partition's energy is the f64 sum of its lines in ascending order: a systolic fold
        // (as calc_noise's band sums, k_quant.h) -- lane l owns the lines 8 l .. 8 l + 7 and folds them, in order, onto the running sum
        // lane l - 1 hands over, the chain restarting at every partition's first line; ceil(longest partition / 8) + 1 hand-overs give
        // every partition's strictly sequential sum (one lane per partition walking its lines took as many trips as the longest
        // partition has lines: 83 at 44.1 kHz).  Line 512 is added by lane 63 at the end -- it is the last line of the last partition.
        // The maxima are order-free: every lane reduces its lines per partition and merges through an LDS integer maximum (energies
        // are >= 0: IEEE order == integer order of the bit patterns).
        enum { KF = 8 };
        static_assert(BLKSIZE / 2 == KF * LHIP_NL, "psy_fold is laid out for 8 lines per lane");
        const int32_t* pf = T.psy_fold + 3 * lane;
        const uint32_t marks = (uint32_t)pf[0], pw0 = (uint32_t)pf[1], pw1 = (uint32_t)pf[2];
        LHIP_LANE_ONCE(b, 0, CBANDS) PSYA_MX(L)[b] = 0.f;
        double tq[KF], keep[KF];
        float ev[KF];
#pragma unroll
        for (int k = 0; k < KF; k++) { ev[k] = PSYA_FE(L)[KF * lane + k]; tq[k] = (double)ev[k]; keep[k] = one_unless_bit(marks, k); }
        const float e512 = PSYA_FE(L)[BLKSIZE / 2];
        wave_sync();                                  // the maxima start from zero
        {   // maxima: running maximum inside the lane, flushed where a partition ends (or the lane does)
            float m = 0.f;
#pragma unroll
            for (int k = 0; k < KF; k++) {
                m = ((marks >> k) & 1u) ? ev[k] : fmax_nonneg(m, ev[k]);
                const int bnd = (int)(((k < 4 ? pw0 : pw1) >> (8 * (k & 3))) & 0xffu);
                float mm = m;
                if (lane == LHIP_NL - 1 && k == KF - 1 && !((marks >> (8 + k)) & 1u)) mm = fmax_nonneg(mm, e512);     // line 512 belongs to the same partition
                if (((marks >> (8 + k)) & 1u) || k == KF - 1) { int32_t bits; __builtin_memcpy(&bits, &mm, 4); lds_max((int32_t*)&PSYA_MX(L)[bnd], bits); }
            }
            if (lane == LHIP_NL - 1 && ((marks >> (8 + KF - 1)) & 1u)) {       // line 512 is a partition of its own (the last one)
                int32_t bits; __builtin_memcpy(&bits, &e512, 4); lds_max((int32_t*)&PSYA_MX(L)[T.npart_l - 1], bits);
            }
        }
        const int nsteps = (T.psy_maxlen_l + KF - 1) / KF + 1;
        double carry = 0.0;
        for (int st = 0; st + 1 < nsteps; st++) {
            double sacc = carry;
#pragma unroll
            for (int k = 0; k < KF; k++) sacc = __builtin_fma(sacc, keep[k], tq[k]);
            carry = wave_shr1d(sacc, 0.0);
        }
        {
            double sacc = carry;
#pragma unroll
            for (int k = 0; k < KF; k++) { sacc = __builtin_fma(sacc, keep[k], tq[k]); tq[k] = sacc; }   // tq := running sums of the last step
        }
#pragma unroll
        for (int k = 0; k < KF; k++) {
            const int bnd = (int)(((k < 4 ? pw0 : pw1) >> (8 * (k & 3))) & 0xffu);
            if ((marks >> (8 + k)) & 1u) { PSYA_EB(L)[bnd] = (float)tq[k]; PSYA_AV(L)[bnd] = (float)(tq[k] * (double)T.rnumlines_l[bnd]); }
        }
        if (lane == LHIP_NL - 1) {                    // the partition line 512 closes
            const int bl = T.npart_l - 1;
            const double ebb = ((marks >> (8 + KF - 1)) & 1u) ? (double)e512 : tq[KF - 1] + (double)e512;
            PSYA_EB(L)[bl] = (float)ebb; PSYA_AV(L)[bl] = (float)(ebb * (double)T.rnumlines_l[bl]);
        }
    }
#endif
    // --- short partitions: energy per sub-block (compute_masking_s first loop, 740-750) ---
    for (int it = lane; it < 3 * T.npart_s; it += LHIP_NL) {
        const int sblock = it / T.npart_s, b = it - sblock * T.npart_s;
        double ebb = 0;
        int j = T.lineoff_s[b];
        for (int i = 0; i < T.numlines_s[b]; ++i, ++j) ebb += (double)PSYA_FES(L, sblock)[j];
        PSYA_EBS(L, sblock)[b] = (float)ebb;
    }
    wave_sync();

    PSY_STAMP(6);
    // --- tonality index (calc_mask_index_l, PsyModel.js:930-992) ---
    for (int b = lane; b < CBANDS; b += LHIP_NL) {
        int k = 0;
        float ebv = 0.f;
        if (b < T.npart_l) {
            const int last = T.npart_l - 1;
            const int lo = b > 0 ? b - 1 : b, hi = b < last ? b + 1 : b;
            double a, m;
            int nl;
            if (b == 0) { a = (double)PSYA_AV(L)[0] + (double)PSYA_AV(L)[1]; nl = T.numlines_l[0] + T.numlines_l[1] - 1; }
            else if (b == last) { a = (double)PSYA_AV(L)[b - 1] + (double)PSYA_AV(L)[b]; nl = T.numlines_l[b - 1] + T.numlines_l[b] - 1; }
            else { a = (double)PSYA_AV(L)[b - 1] + (double)PSYA_AV(L)[b] + (double)PSYA_AV(L)[b + 1]; nl = T.numlines_l[b - 1] + T.numlines_l[b] + T.numlines_l[b + 1] - 1; }
            if (a > 0.0) {
                m = PSYA_MX(L)[lo];
                for (int t = lo + 1; t <= hi; t++) if (m < (double)PSYA_MX(L)[t]) m = PSYA_MX(L)[t];
                a = 20.0 * (m * (double)(hi - lo + 1) - a) / (a * nl);
                k = js_toint32(a);
                if (k > 8) k = 8;
            }
            ebv = PSYA_EB(L)[b];
        }
        W.eb_l[o * EBL_STRIDE + b] = ebv;
        W.mask_idx[o * EBL_STRIDE + b] = k;
    }
    // --- short spreading (compute_masking_s second loop before limiting, 752-760) ---
    // lane = partition; its spreading row is fetched once (independent loads) and applied to the three sub-blocks
    for (int b = lane; b < CBANDS; b += LHIP_NL) {
        float ecbv[3] = {0.f, 0.f, 0.f}, ebv[3] = {0.f, 0.f, 0.f};
        if (b < T.npart_s) {
            enum { MT = 12 };                         // rows are at most 12 long for every MPEG-1 rate; longer rows take the tail loop
            const int k0 = T.s3ind_s[2 * b], k1 = T.s3ind_s[2 * b + 1], j0 = T.s3off_s[b];
            float cf[MT];
#pragma unroll
            for (int u = 0; u < MT; u++) cf[u] = (k0 + u <= k1) ? T.s3_ss[j0 + u] : 0.f;
            for (int sblock = 0; sblock < 3; sblock++) {
                const float* e = PSYA_EBS(L, sblock);
                double ecb = (double)cf[0] * (double)e[k0];
#pragma unroll
                for (int u = 1; u < MT; u++) if (k0 + u <= k1) ecb += (double)cf[u] * (double)e[k0 + u];
                for (int kk = k0 + MT; kk <= k1; kk++) ecb += (double)T.s3_ss[j0 + (kk - k0)] * (double)e[kk];
                ecbv[sblock] = (float)ecb;
                ebv[sblock] = e[b];
            }
        }
        for (int sblock = 0; sblock < 3; sblock++) {
            W.ecb_s[o * EBS_STRIDE + sblock * CBANDS + b] = ecbv[sblock];
            W.eb_s[o * EBS_STRIDE + sblock * CBANDS + b] = ebv[sblock];
        }
    }
    PSY_STAMP(7);
    if (PART == 0 || PART == 2) PSY_FLUSH();
#undef buf
}

// ---------------------------------------------------------------------------------------------
// kb_scan: one wave per stream.  Phase 1 (parallel): raw attack flags per (psy call, channel).
// Phase 2 (lane 0): attack clean-up / block-type chain and the ATH auto-adjust recurrence.
// ---------------------------------------------------------------------------------------------
LHIP_DEV int attack_flags_raw(const Tables& T, const float* cur, const float* prev, int chn) {
    // PsyModel.js:1105-1181 under the fractional-index port bug: only i in {0,3,6,9}
    const double thr = (chn == 3) ? T.attackthre_s : T.attackthre;
    float ai[4];
    ai[0] = (float)((double)prev[6] / (double)prev[4]);
    const double refv[3] = {(double)prev[7], (double)cur[1], (double)cur[4]};
    for (int t = 0; t < 3; t++) {
        double p = cur[3 * t];
        const double r = refv[t];
        if (p > r) p = p / r;
        else if (r > p * 10.0) p = r / (p * 10.0);
        else p = 0.0;
        ai[t + 1] = (float)p;
    }
    double en_short[4];
    en_short[0] = 0.0;
    for (int i = 0; i < 3; i++) en_short[0] += (double)prev[i + 6];
    en_short[1] = 0.0 + (double)cur[0]; en_short[2] = 0.0 + (double)cur[3]; en_short[3] = 0.0 + (double)cur[6];
    int a[4];
    for (int j = 0; j < 4; j++) a[j] = ((double)ai[j] > thr) ? 1 : 0;
    for (int i = 1; i < 4; i++) {
        double ratio;
        if (en_short[i - 1] > en_short[i]) ratio = en_short[i - 1] / en_short[i];
        else ratio = en_short[i] / en_short[i - 1];
        if (ratio < 1.7) { a[i] = 0; if (i == 1) a[0] = 0; }
    }
    return a[0] | (a[1] << 1) | (a[2] << 2) | (a[3] << 3);
}

// raw attack flags of one psy call, all channels (parallel over granule slots)
LHIP_DEV void kb_scan_raw(const Tables& T, const Workspace& W, const StreamDesc* SD, int gslot) {
    const int Cp = T.psy_channels;
    const StreamDesc sd = SD[W.gslot_stream[gslot]];
    if (gslot - sd.gslot0 - 1 < 0) return;
    for (int ch = 0; ch < Cp; ch++) {
        const int64_t o = (int64_t)gslot * Cp + ch;
        W.att_raw[o] = attack_flags_raw(T, W.peaks + o * PK_STRIDE, W.peaks + (o - Cp) * PK_STRIDE, ch);
    }
}

// lastAttacks after the psy call in `gslot` (PsyModel.js:1183-1196, 1268), resolved by looking back:
// a0' = a0 & !last, a1' = a1 & !a0', a2' = a2 & !a1'  ->  the value only depends on the previous call
// when a0 = a1 = a2 = 1, so the walk almost always stops at once.
LHIP_DEV int last_attack_after(const Workspace& W, int C, int gcarry, int gslot, int ch) {
    int flips = 0, g = gslot;
    for (;;) {
        if (g == gcarry) { const int v = W.last_attack[(int64_t)g * C + ch] != 0; return flips ? !v : v; }
        const int raw = W.att_raw[(int64_t)g * C + ch];
        const int a0 = raw & 1, a1 = (raw >> 1) & 1, a2 = (raw >> 2) & 1;
        int v;
        if (!a2) v = 0;
        else if (!a1) v = 1;
        else if (!a0) v = 0;
        else { flips ^= 1; g--; continue; }        // a2' = !last(g-1)
        return flips ? !v : v;
    }
}

// per granule slot: attack clean-up -> uselongblock (coupled), lastAttacks (parallel over granule slots)
LHIP_DEV void kb_scan_attack(const Tables& T, const Workspace& W, const StreamDesc* SD, int gslot) {
    const int C = T.channels_out, Cp = T.psy_channels;
    const StreamDesc sd = SD[W.gslot_stream[gslot]];
    if (gslot - sd.gslot0 - 1 < 0) return;
    int ul[2] = {1, 1};
    for (int ch = 0; ch < Cp; ch++) {
        const int last = last_attack_after(W, Cp, sd.gslot0, gslot - 1, ch);
        const int raw = W.att_raw[(int64_t)gslot * Cp + ch];
        int a0 = raw & 1, a1 = (raw >> 1) & 1, a2 = (raw >> 2) & 1, a3 = (raw >> 3) & 1;
        if (a0 != 0 && last != 0) a0 = 0;
        if ((a0 + a1 + a2 + a3) != 0) {            // lastAttacks == 3 never happens
            if (ch < 2) ul[ch] = 0;
            else ul[0] = ul[1] = 0;                 // an attack in mid or side switches both channels (PsyModel.js:1198-1204)
            if (a1 != 0 && a0 != 0) a1 = 0;
            if (a2 != 0 && a1 != 0) a2 = 0;
            if (a3 != 0 && a2 != 0) a3 = 0;
        }
        W.att_clean[(int64_t)gslot * Cp + ch] = a0 | (a1 << 1) | (a2 << 2) | (a3 << 3);
        W.ul_tmp[(int64_t)gslot * Cp + ch] = a2;   // lastAttacks after this call (published below)
    }
    if (T.short_blocks_coupled && !(ul[0] != 0 && ul[1] != 0)) ul[0] = ul[1] = 0;
    for (int ch = 0; ch < C; ch++) W.uselong[(int64_t)gslot * C + ch] = ul[ch];
}

// per granule slot: block_type_set chain from the (coupled) uselongblock flags of this and the two
// previous calls (PsyModel.js:784-826) (parallel over granule slots)
LHIP_DEV void kb_scan_blocktype(const Tables& T, const Workspace& W, const StreamDesc* SD, int gslot) {
    const int C = T.channels_out;
    const StreamDesc sd = SD[W.gslot_stream[gslot]];
    const int q = gslot - sd.gslot0 - 1;
    if (q < 0) return;
    for (int ch = 0; ch < C; ch++) {
        // tentative type left by the previous call
        int old;
        if (q == 0) old = W.tent[(int64_t)sd.gslot0 * C + ch];
        else {
            const int ul1 = W.uselong[(int64_t)(gslot - 1) * C + ch];
            int pshort;
            if (q == 1) pshort = W.tent[(int64_t)sd.gslot0 * C + ch] == SHORT_TYPE;
            else pshort = W.uselong[(int64_t)(gslot - 2) * C + ch] == 0;
            old = ul1 ? (pshort ? STOP_TYPE : NORM_TYPE) : SHORT_TYPE;
        }
        const int ul = W.uselong[(int64_t)gslot * C + ch];
        W.prev_short[(int64_t)gslot * C + ch] = (old == SHORT_TYPE) ? 1 : 0;
        int bt = NORM_TYPE;
        if (ul != 0) { if (old == SHORT_TYPE) bt = STOP_TYPE; }
        else {
            bt = SHORT_TYPE;
            if (old == NORM_TYPE) old = START_TYPE;
            if (old == STOP_TYPE) old = SHORT_TYPE;
        }
        W.blocktype[(int64_t)gslot * C + ch] = old;
        W.tent[(int64_t)gslot * C + ch] = bt;
    }
    const int Cp = T.psy_channels;
    for (int ch = 0; ch < Cp; ch++) W.last_attack[(int64_t)gslot * Cp + ch] = W.ul_tmp[(int64_t)gslot * Cp + ch];
}

// ATH auto-adjust recurrence (Encoder.js:166-243), one 1024-thread workgroup per stream.
// The recurrence (adjust, adjustLimit) <- F_k(adjust, adjustLimit) is serial in general (repeated f64
// multiplications during a loudness descent cannot be re-associated), but two consecutive "loud" frames
// (max_pow > 0.03125) force the state to (1, 1) whatever came before.  Frames are processed in chunks of
// ATH_NT x ATH_SEG; inside a chunk every thread owns a contiguous segment: pass A evaluates each segment from its
// first such reset point onwards (no dependence on other threads), then the segment prefixes are filled in
// as soon as the state at the end of the previous segment is known (at most ATH_NT rounds, 1 in the common case).
// Nothing but the segment end states is staged: max_pow is recomputed from the per-granule loudness (L2-resident).
#if defined(LHIP_HOSTSIM) && defined(LHIP_WAVESIM)
// TEST-ONLY: the wave simulator runs the scan as a workgroup of two waves with two-frame segments, so that a few hundred frames walk
// through several chunks, multi-frame segments, reset points, prefix rounds and the chunk carry (the device: 1024 threads x 16)
enum { ATH_NT = 128, ATH_SEG = 2 };
LHIP_DEV void block_sync() { wg_barrier(); }
LHIP_DEV int block_any(int p) { static int any_ = 0; wg_barrier(); if (p) any_ = 1; wg_barrier(); const int r = any_; wg_barrier(); any_ = 0; return r; }
#elif defined(LHIP_HOSTSIM)
enum { ATH_NT = 1, ATH_SEG = 1 << 20 };
LHIP_DEV void block_sync() {}
LHIP_DEV int block_any(int p) { return p != 0; }
#else
enum { ATH_NT = 1024, ATH_SEG = 16 };
LHIP_DEV void block_sync() { __syncthreads(); }
LHIP_DEV int block_any(int p) { return __syncthreads_or(p); }
#endif
struct AthLds { double e_adj[ATH_NT + 1], e_lim[ATH_NT + 1]; int known[ATH_NT + 1], first_reset[ATH_NT + 1]; double c_adj, c_lim, c_mp; };

LHIP_DEV void ath_step(const Tables& T, double max_pow, double& adj, double& lim) {
    if (T.ATH_useAdjust == 0) { adj = 1.0; return; }
    if (max_pow > 0.03125) {
        if (adj >= 1.0) adj = 1.0;
        else if (adj < lim) adj = lim;
        lim = 1.0;
    } else {
        const double adj_lim_new = 31.98 * max_pow + 0.000625;
        if (adj >= adj_lim_new) {
            adj *= adj_lim_new * 0.075 + 0.925;
            if (adj < adj_lim_new) adj = adj_lim_new;
        } else {
            if (lim >= adj_lim_new) adj = adj_lim_new;
            else if (adj < lim) adj = lim;
        }
        lim = adj_lim_new;
    }
}

// max_pow of frame k of the stream (Encoder.js:420-440): loudness of the two psy calls before the frame's granules
LHIP_DEV double ath_max_pow(const Tables& T, const Workspace& W, const StreamDesc& sd, int C, int k) {
    const int GR = T.mode_gr;
    const int64_t g0 = (int64_t)(sd.gslot0 + GR * k) * C, g1 = g0 + C;
    double max_pow = W.loud[g0], gr2_max = (GR == 2) ? (double)W.loud[g1] : 0.0;
    if (C == 2) { max_pow += (double)W.loud[g0 + 1]; if (GR == 2) gr2_max += (double)W.loud[g1 + 1]; }
    else { max_pow += max_pow; gr2_max += gr2_max; }
    if (GR == 2) max_pow = max_pow > gr2_max ? max_pow : gr2_max;      // Encoder.js:187-189: only with two granules
    max_pow *= 0.5;
    max_pow *= T.ATH_aaSensitivityP;
    return max_pow;
}

LHIP_DEV void kb_scan_ath(const Tables& T, const Workspace& W, const StreamDesc* SD, int st, int tid, AthLds& L) {
    const int C = T.channels_out;
    const StreamDesc sd = SD[st];
    if (tid == 0) { L.c_adj = W.ath_adjust[sd.fslot0]; L.c_lim = W.ath_limit[sd.fslot0]; L.c_mp = 0.0; }   // state carried into the chunk; 0: previous frame unknown / not loud
    block_sync();
    for (int k0 = 0; k0 < sd.nframes; k0 += ATH_NT * ATH_SEG) {
        const int rem = sd.nframes - k0;
        const int n = rem < ATH_NT * ATH_SEG ? rem : ATH_NT * ATH_SEG;
        const int seglen = (n + ATH_NT - 1) / ATH_NT;
        const int s0 = tid * seglen < n ? tid * seglen : n, s1 = (tid + 1) * seglen < n ? (tid + 1) * seglen : n;
        const double cadj = L.c_adj, clim = L.c_lim, prev_mp = L.c_mp;
        // pass A: from the first reset point of the segment to its end
        {
            int fr = -1;
            if (T.ATH_useAdjust != 0) {
                double pm = (s0 < s1) ? (s0 > 0 ? ath_max_pow(T, W, sd, C, k0 + s0 - 1) : prev_mp) : 0.0;
                for (int k = s0; k < s1; k++) {
                    const double m = ath_max_pow(T, W, sd, C, k0 + k);
                    if (pm > 0.03125 && m > 0.03125) { fr = k; break; }
                    pm = m;
                }
            } else if (s0 < s1) fr = s0;
            double a = 1.0, l = 1.0;
            if (fr >= 0) {
                l = (T.ATH_useAdjust != 0) ? 1.0 : clim;
                W.ath_adjust[sd.fslot0 + 1 + k0 + fr] = 1.0; W.ath_limit[sd.fslot0 + 1 + k0 + fr] = l;
                for (int k = fr + 1; k < s1; k++) {
                    ath_step(T, ath_max_pow(T, W, sd, C, k0 + k), a, l);
                    W.ath_adjust[sd.fslot0 + 1 + k0 + k] = a; W.ath_limit[sd.fslot0 + 1 + k0 + k] = l;
                }
            }
            L.first_reset[tid] = fr;
            L.known[tid + 1] = (fr >= 0) || (s0 >= s1);      // empty segments are all at the tail: nobody waits for their state
                                                             // (forwarding it one thread per round cost a 1-frame call 1023 rounds)
            L.e_adj[tid + 1] = a; L.e_lim[tid + 1] = l;
            if (tid == 0) { L.known[0] = 1; L.e_adj[0] = cadj; L.e_lim[0] = clim; }
        }
        block_sync();
        // prefix rounds: a segment's frames before its first reset need the state at the end of the previous segment
        int done = (s0 >= s1);
        for (int round = 0; round < ATH_NT + 1; round++) {
            int progressed = 0;
            if (!done && L.known[tid]) {
                double a = L.e_adj[tid], l = L.e_lim[tid];
                const int fr = L.first_reset[tid];
                const int stop = fr >= 0 ? fr : s1;
                for (int k = s0; k < stop; k++) {
                    ath_step(T, ath_max_pow(T, W, sd, C, k0 + k), a, l);
                    W.ath_adjust[sd.fslot0 + 1 + k0 + k] = a; W.ath_limit[sd.fslot0 + 1 + k0 + k] = l;
                }
                if (fr < 0) { L.e_adj[tid + 1] = a; L.e_lim[tid + 1] = l; }
                done = 1; progressed = 1;
            }
            block_sync();
            if (progressed && L.first_reset[tid] < 0) L.known[tid + 1] = 1;
            // empty segments simply forward the state
            if (s0 >= s1 && L.known[tid] && !L.known[tid + 1]) { L.e_adj[tid + 1] = L.e_adj[tid]; L.e_lim[tid + 1] = L.e_lim[tid]; L.known[tid + 1] = 1; }
            block_sync();
            if (!block_any(!done || !L.known[tid + 1])) break;
        }
        // carry: state after the last frame of the chunk = end state of the last non-empty segment
        if (s0 < s1 && s1 == n) { L.c_adj = L.e_adj[tid + 1]; L.c_lim = L.e_lim[tid + 1]; L.c_mp = ath_max_pow(T, W, sd, C, k0 + n - 1); }
        block_sync();
    }
}

// ---------------------------------------------------------------------------------------------
// kb_psyB: one wave per psy call (all channels).
// ---------------------------------------------------------------------------------------------
// NCH = psy channels the instantiation can hold: 2 (mono / stereo), or 4 for joint stereo (L, R, mid, side) -- a separate kernel so that
// the usual configurations keep their 3.5 KB of LDS per wave
struct PsyBTabs { double mt1[25], mt2[10], mt3[14], mtab[9]; };     // mask_add tables: looked up inside a serially dependent chain
// The long-block spreading is a chain per partition (lane) whose every step needs three operands of ANOTHER partition k -- its energy times
// its tonality factor, its ATH share -- and one entry of the lane's own spreading row.  Fetched from global memory inside the chain (one step
// ahead) they left the wave in s_waitcnt for 70 % of its cycles: a step computes for ~150 cycles, a load from L2 takes several hundred.  They are
// staged in LDS before the chains start: ebm[k] = eb_l[k] * mtab[mask_idx[k]] and athc[k] = ATH_cb_l[k] * ATH.adjust (the products the chain
// formed at every use: same operands, same operation) and the spreading rows s3_ll, which overlay the arrays only the later phases use.
enum { PSYB_S3_LDS = 1024 };
template <int NCH> struct PsyBLdsT : PsyBTabs {
    double ebm[CBANDS], athc[CBANDS];
    float thr_l[NCH][CBANDS + 2];
    union {
        struct { float thr_s[NCH][3][CBANDS + 2]; float E[NCH][E_STRIDE]; };     // short limiting onwards
        float s3[PSYB_S3_LDS];                                                   // long spreading: T.s3_ll
    };
};
typedef PsyBLdsT<2> PsyBLds;
typedef PsyBLdsT<4> PsyBLds4;

// js_toint32(v8_log10_pos(ratio) * 16.0), the table index of mask_add (PsyModel.js:403-473), for 1 <= ratio < 2^20 -- without the
// logarithm where that is safe: the index is a step function of ratio with steps at 10^(k / 16), and away from the steps any
// approximation of 16 log10(ratio) with an error below the distance to the next integer gives the same integer.
// t = log2_f32((float)ratio) * 16 log10(2): error < 2e-5 (one ulp of v_log_f32 at |log2| <= 20, the conversion of ratio, the f32
// multiply); returns -1 within 2e-4 of an integer (the caller then takes the logarithm, as before; about 1 operand in 2500).
LHIP_DEV int ma_index16(double ratio) {
    const float t = fast_log2f((float)ratio) * 4.81647993062369912f;
    const float k = __builtin_floorf(t);
    const float fr = t - k;
    // (written so that a ratio beyond Float32 or a NaN -- t infinite or NaN, fr NaN -- fails the test and takes the logarithm as well)
    int i = -1;
    if (fr >= 2e-4f && fr <= 1.0f - 2e-4f) i = (int)k;
#ifdef LHIP_HOSTSIM
    if (i >= 0 && i != js_toint32(v8_log10_pos(ratio) * 16.0)) { fprintf(stderr, "hostsim: ma_index16 disagrees with the logarithm at %a\n", ratio); abort(); }
#endif
    return i;
}
LHIP_DEV int ma_index16_exact(double ratio) {
    int i = ma_index16(ratio);
    if (i < 0) i = js_toint32(v8_log10_pos(ratio) * 16.0);
    return i;
}

LHIP_DEV double mask_add_l(const Tables& T, const PsyBTabs& L, double ath_cb, double m1, double m2, int b) {
    // PsyModel.js:403-473 (long blocks).  Every logarithm here has a positive, finite, normal operand -- `ratio` lies in
    // [1, ma_max_i2) and m1 / m2 in (1, ma_max_m) on the paths that take it -- so the branch-free v8_log10_pos applies (lhip_math.h)
    double ratio;
    if (m2 > m1) {
        if (m2 < (m1 * T.ma_max_i2)) ratio = m2 / m1;
        else return (m1 + m2);
    } else {
        if (m1 >= (m2 * T.ma_max_i2)) return (m1 + m2);
        ratio = m1 / m2;
    }
    m1 += m2;
    if ((b + 3) <= 3 + 3) {
        if (ratio >= T.ma_max_i1) return m1;
        const int i = ma_index16_exact(ratio);
        return m1 * L.mt2[i];
    }
    const int i = ma_index16_exact(ratio);
    m2 = ath_cb;
    if (m1 < T.ma_max_m * m2) {
        if (m1 > m2) {
            double f = 1.0;
            if (i <= 13) f = L.mt3[i];
            const double r = v8_log10_pos(m1 / m2) * (10.0 / 15.0);
            return m1 * ((L.mt1[i] - f) * r + f);
        }
        if (i > 13) return m1;
        return m1 * L.mt3[i];
    }
    return m1 * L.mt1[i];
}


// NS_INTERP (PsyModel.js:828-842).  r = constant * pcfact; pcfact is 0 with the reservoir disabled, where this returns y unchanged
LHIP_DEV double ns_interp(double x, double y, double r) {
    if (r >= 1.0) return x;
    if (r <= 0.0) return y;
    if (y > 0.0) return v8_pow(x / y, r) * y;
    return 0.0;
}

// par >= 0 (bit reservoir): only the psy calls with q % mode_gr == par -- the second granule's short-block pre-echo control looks at
// the first one's finished thresholds, so the two are launched one after the other
// resv_size / resv_max (bit reservoir only): ResvSize / ResvMax as the previous frame left them -- handed in by the caller, which knows
// where the stream's reservoir record lives (the persistent per-stream kernel keeps it in LDS and pipelines the previous frame's bit
// packing with this call, so this function must not read the record itself)
template <int NCH> LHIP_DEV void kb_psyB(const Tables& T, const PowBase& pb10, const Workspace& W, const StreamDesc* SD, int gslot, int lane, PsyBLdsT<NCH>& L, int par = -1,
                                         int resv_size = 0, int resv_max = 0) {
    const int C = T.channels_out, Cp = T.psy_channels;      // Cp <= NCH: the launch picks the instantiation by Tables::psy_channels
    const int st = W.gslot_stream[gslot];
    const StreamDesc sd = SD[st];
    const int q = gslot - sd.gslot0 - 1;
    if (q < 0) return;
    if (par >= 0 && q % T.mode_gr != par) return;
    // PsyModel.js:1036-1038: share of the reservoir in use, as the previous frame left it (0 with the reservoir disabled)
    double pcfact = 0.0;
    if (!T.disable_reservoir) pcfact = resv_max == 0 ? 0.0 : (double)resv_size / resv_max * 0.5;
    const int fs = sd.fslot0 + q / T.mode_gr;            // ATH.adjust as left by the previous frame
    const double ath_adjust = W.ath_adjust[fs];
    for (int i = lane; i < 25; i += LHIP_NL) {
        L.mt1[i] = T.ma_table1[i];
        if (i < 10) L.mt2[i] = T.ma_table2[i];
        if (i < 14) L.mt3[i] = T.ma_table3[i];
        if (i < 9) L.mtab[i] = T.ma_tab[i];
    }
    wave_sync();

    for (int i = lane; i < T.n_s3_ll; i += LHIP_NL) L.s3[i] = T.s3_ll[i];
    LHIP_LANE_ONCE(k, 0, T.npart_l) L.athc[k] = (double)T.ATH_cb_l[k] * ath_adjust;
    for (int ch = 0; ch < Cp; ch++) {
        const int64_t o = (int64_t)gslot * Cp + ch;
        const float* eb_l = W.eb_l + o * EBL_STRIDE;
        const int32_t* midx = W.mask_idx + o * EBL_STRIDE;
        LHIP_LANE_ONCE(k, 0, T.npart_l) L.ebm[k] = (double)eb_l[k] * L.mtab[midx[k]];
        wave_sync();
        // long-block spreading with additive masking (PsyModel.js:1274-1320); thr = ecb (pcfact == 0)
        // The additive-masking chain is serial in the partition's spreading row; all of its operands are in LDS (see PsyBLdsT).
        LHIP_LANE_ONCE(b, 0, T.npart_l) {                        // npart_l < CBANDS = 64
            const int k0 = T.s3ind[2 * b], k1 = T.s3ind[2 * b + 1], j0 = T.s3off_l[b];
            double ecb = (double)L.s3[j0] * L.ebm[k0];
            for (int kk = k0 + 1; kk <= k1; kk++)
                ecb = mask_add_l(T, L, L.athc[kk], ecb, (double)L.s3[j0 + (kk - k0)] * L.ebm[kk], kk - b);
            ecb *= 0.158489319246111;
            float thr = (float)ecb;
            if (!T.disable_reservoir) {   // long-block pre-echo control (PsyModel.js:1300-1318): dead with the reservoir disabled (pcfact == 0)
                const float n1 = W.nb1[(o - Cp) * EBL_STRIDE + b], n2 = W.nb2[(o - Cp) * EBL_STRIDE + b];
                if (pcfact > 0.0 && !W.prev_short[(int64_t)gslot * C + (ch & 1)]) {
                    const double a = 2 * (double)n1, b2 = 16 * (double)n2, m = a < b2 ? a : b2;
                    thr = (float)ns_interp(ecb < m ? ecb : m, ecb, pcfact);
                }
                W.nb2[o * EBL_STRIDE + b] = n1;
                W.nb1[o * EBL_STRIDE + b] = (float)ecb;
            }
            L.thr_l[ch][b] = thr;
        }
        wave_sync();                                  // the next channel refills ebm
    }
    // (the spreading rows are dead: their LDS is thr_s / E from here on)
    for (int ch = 0; ch < Cp; ch++) {
        const int64_t o = (int64_t)gslot * Cp + ch;
        // short-block limiting by the two previous sub-blocks (compute_masking_s, 762-775)
        const int pshort = W.prev_short[(int64_t)gslot * C + (ch & 1)];      // blocktype_old[chn & 1] (PsyModel.js:767)
        for (int it = lane; it < 3 * T.npart_s; it += LHIP_NL) {
            const int sblock = it / T.npart_s, b = it - sblock * T.npart_s;
            const float* e0 = W.ecb_s + o * EBS_STRIDE;
            const float* em = W.ecb_s + (o - Cp) * EBS_STRIDE;      // previous psy call (or carry)
            const float ecb = e0[sblock * CBANDS + b];
            const float nb1 = sblock >= 1 ? e0[(sblock - 1) * CBANDS + b] : em[2 * CBANDS + b];
            const float nb2 = sblock >= 2 ? e0[(sblock - 2) * CBANDS + b] : em[(sblock + 1) * CBANDS + b];
            double x = 2 * (double)nb1;
            float thr = (float)((double)ecb < x ? (double)ecb : x);
            if (pshort) {
                x = 16 * (double)nb2;
                const double y = thr;
                thr = (float)(x < y ? x : y);
            }
            L.thr_s[ch][sblock][b] = thr;
        }
    }
    wave_sync();

    for (int ch = 0; ch < Cp; ch++) {
        const int64_t o = (int64_t)gslot * Cp + ch;
        const float* eb_l = W.eb_l + o * EBL_STRIDE;
        float* Eo = L.E[ch];
        // convert_partition2scalefac_l (PsyModel.js:692-734): lane per scalefactor band
        for (int sb = lane; sb < SBMAX_l; sb += LHIP_NL) {
            const int npart = T.npart_l;
            const int bstart = sb == 0 ? 0 : T.bo_l[sb - 1];      // partition shared with the previous band
            float en_f = 0.f, thm_f = 0.f;
            // band sb exists only if the walk has not run past npart before reaching it
            if (sb == 0 || bstart < npart) {
                double enn = 0.0, thmm = 0.0;
                int b = bstart;
                if (sb > 0) {
                    const double w_next = 1.0 - (double)T.bo_l_weight[sb - 1];
                    enn = w_next * (double)eb_l[b];
                    thmm = w_next * (double)L.thr_l[ch][b];
                    b++;
                }
                const int bo = T.bo_l[sb];
                const int b_lim = bo < npart ? bo : npart;
                while (b < b_lim) { enn += (double)eb_l[b]; thmm += (double)L.thr_l[ch][b]; b++; }
                en_f = (float)enn; thm_f = (float)thmm;
                if (b < npart) {
                    const double w_curr = T.bo_l_weight[sb];
                    en_f = (float)((double)en_f + w_curr * (double)eb_l[b]);
                    thm_f = (float)((double)thm_f + w_curr * (double)L.thr_l[ch][b]);
                }
            }
            Eo[E_EN_L + sb] = en_f;
            Eo[E_THM_L + sb] = thm_f;
        }
        // convert_partition2scalefac_s (644-687)
        for (int it = lane; it < 3 * SBMAX_s; it += LHIP_NL) {
            const int sblock = it / SBMAX_s, sb = it - sblock * SBMAX_s;
            const float* ebs = W.eb_s + o * EBS_STRIDE + sblock * CBANDS;
            const float* thr = L.thr_s[ch][sblock];
            const int npart = T.npart_s;
            const int bstart = sb == 0 ? 0 : T.bo_s[sb - 1];
            float en_f = 0.f, thm_f = 0.f;
            if (sb == 0 || bstart < npart) {
                double enn = 0.0, thmm = 0.0;
                int b = bstart;
                if (sb > 0) {
                    const double w_next = 1.0 - (double)T.bo_s_weight[sb - 1];
                    enn = w_next * (double)ebs[b];
                    thmm = w_next * (double)thr[b];
                    b++;
                }
                const int bo = T.bo_s[sb];
                const int b_lim = bo < npart ? bo : npart;
                while (b < b_lim) { enn += (double)ebs[b]; thmm += (double)thr[b]; b++; }
                en_f = (float)enn; thm_f = (float)thmm;
                if (b < npart) {
                    const double w_curr = T.bo_s_weight[sb];
                    en_f = (float)((double)en_f + w_curr * (double)ebs[b]);
                    thm_f = (float)((double)thm_f + w_curr * (double)thr[b]);
                }
            }
            Eo[E_EN_S + sb * 3 + sblock] = en_f;
            Eo[E_THM_S + sb * 3 + sblock] = thm_f;                 // as converted; the pre-echo control follows
        }
    }
    wave_sync();
    for (int ch = 0; ch < Cp; ch++) {
        // short-block pre-echo control (PsyModel.js:1226-1267), one lane per band, the three sub-blocks in order: x0.8, the two
        // attack-driven interpolations towards the previous sub-block's finished threshold (sub-block 0: the previous CALL's
        // sub-block 2) -- identities when pcfact == 0 -- and the pulse halving.  ns_attacks holds only 0 / 1, so
        // the reference's `>= 2` and `== 3` alternatives never fire.
        const int64_t o = (int64_t)gslot * Cp + ch;
        const float* pk = W.peaks + o * PK_STRIDE;            // en_subshort[3..11]
        const int att = W.att_clean[o];
        float* Eo = L.E[ch];
        LHIP_LANE_ONCE(sb, 0, SBMAX_s) {
            float prev = pcfact > 0.0 ? W.E[(o - Cp) * E_STRIDE + E_THM_S + sb * 3 + 2] : 0.f;      // (only read where the previous call is complete)
            for (int sblock = 0; sblock < 3; sblock++) {
                double thmm = Eo[E_THM_S + sb * 3 + sblock];
                thmm *= 0.8;
                if ((att >> (sblock + 1)) & 1) { const double p = ns_interp(prev, thmm, 0.6 * pcfact); thmm = thmm < p ? thmm : p; }
                if ((att >> sblock) & 1) { const double p = ns_interp(prev, thmm, 0.3 * pcfact); thmm = thmm < p ? thmm : p; }
                const double e3 = pk[sblock * 3 + 0], e4 = pk[sblock * 3 + 1], e5 = pk[sblock * 3 + 2];
                const double enn = e3 + e4 + e5;
                if (e5 * 6 < enn) {
                    thmm *= 0.5;
                    if (e4 * 6 < enn) thmm *= 0.5;
                }
                prev = (float)thmm;
                Eo[E_THM_S + sb * 3 + sblock] = prev;
            }
        }
    }
    wave_sync();
    // inter-channel masking (PsyModel.js:525-543): stereo mode with ratio > 0
    if ((T.mode == 0 || T.mode == 1) && T.interChRatio > 0.0 && C > 1) {
        const double r_ = T.interChRatio;
        float nl[2] = {0.f, 0.f};
        for (int i = lane; i < SBMAX_l + 3 * SBMAX_s; i += LHIP_NL) {
            const int idx = E_THM_L + i;                       // thm.l then thm.s are contiguous
            const double l = L.E[0][idx], r = L.E[1][idx];
            nl[0] = (float)(l + r * r_);
            nl[1] = (float)(r + l * r_);
            L.E[0][idx] = nl[0];
            L.E[1][idx] = nl[1];
        }
        wave_sync();
    }
    if constexpr (NCH == 4) if (Cp == 4) {
        // joint stereo (PsyModel.js:1336-1342): msfix1 (548-582), then ns_msfix (591-636) with the ATH.adjust the previous frame left.
        // One lane per band (thm.l then thm.s are contiguous in E, and so are en.l / en.s); both steps only touch the band's own
        // four thresholds, so they run back to back in the lane.
        const double msfix = T.msfix;
        const bool do_ns = d_abs(msfix) > 0.0;
        const double athlower_l = do_ns ? v8_pow_base(pb10, T.ATHlower * ath_adjust) : 0.0;
        const double athlower_s = athlower_l * ((double)BLKSIZE_s / BLKSIZE);
        for (int i = lane; i < SBMAX_l + 3 * SBMAX_s; i += LHIP_NL) {
            const int it = E_THM_L + i, ie = E_EN_L + i;
            const bool lng = i < SBMAX_l;
            const int sb = lng ? i : (i - SBMAX_l) / 3;
            float t0 = L.E[0][it], t1 = L.E[1][it], t2 = L.E[2][it], t3 = L.E[3][it];
            if (!((double)t0 > 1.58 * (double)t1 || (double)t1 > 1.58 * (double)t0)) {
                const double m = lng ? (double)T.mld_l[sb] : (double)T.mld_s[sb];
                double mld = m * (double)L.E[3][ie];
                const double lo1 = (double)t3 < mld ? (double)t3 : mld;
                const double rmid = (double)t2 > lo1 ? (double)t2 : lo1;
                mld = m * (double)L.E[2][ie];
                const double lo2 = (double)t2 < mld ? (double)t2 : mld;
                const double rside = (double)t3 > lo2 ? (double)t3 : lo2;
                t2 = (float)rmid; t3 = (float)rside;
            }
            if (do_ns) {
                const double ath = lng ? (double)T.ATH_cb_l[T.bm_l[sb]] * athlower_l : (double)T.ATH_cb_s[T.bm_s[sb]] * athlower_s;
                const double a0 = (double)t0 > ath ? (double)t0 : ath, a1 = (double)t1 > ath ? (double)t1 : ath;
                const double thmLR = a0 < a1 ? a0 : a1;
                double thmM = (double)t2 > ath ? (double)t2 : ath, thmS = (double)t3 > ath ? (double)t3 : ath;
                if (thmLR * (msfix * 2.0) < thmM + thmS) {
                    const double f = thmLR * (msfix * 2.0) / (thmM + thmS);
                    thmM *= f;
                    thmS *= f;
                }
                t2 = (float)(thmM < (double)t2 ? thmM : (double)t2);
                t3 = (float)(thmS < (double)t3 ? thmS : (double)t3);
            }
            L.E[2][it] = t2; L.E[3][it] = t3;
        }
        wave_sync();
    }
    for (int ch = 0; ch < Cp; ch++)
        for (int i = lane; i < E_STRIDE; i += LHIP_NL) W.E[((int64_t)gslot * Cp + ch) * E_STRIDE + i] = L.E[ch][i];
}

}  // namespace lhip
