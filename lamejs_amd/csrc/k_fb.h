// Polyphase analysis filterbank + MDCT kernels.
//   kb_polyphase : per (granule, channel): 18 slots of the 32-band polyphase analysis
//                  (reference NewMDCT.js window_subband 534-914), odd-slot sign compensation
//                  (1074-1076) and the per-band low-pass amplitude scaling (1089-1096).
//   kb_mdct      : per granule: windowing + 36/12-point MDCT per band (mdct_long 981-1051,
//                  mdct_short 927-979, driver 1085-1133) and the alias-reduction butterflies
//                  (1136-1149), block types decided by kb_scan.
// Every store into the 32-slot row / xr rounds to f32 exactly where the reference does.
#pragma once
#include "lhip_defs.h"
#include "lhip_wave.h"
#include "lhip_layout.h"

namespace lhip {

// PCM window of one (granule, channel) work item staged in LDS, transposed: sample n (relative to the first slot
// anchor minus POLY_BIAS) lives at [(n & 31) * POLY_ROW + (n >> 5)].  Lane j (slot j, anchor 32 j) then reads
// anchor-relative offset `off` at [((off + BIAS) & 31) * ROW + ((off + BIAS) >> 5) + j]: consecutive lanes hit
// consecutive banks, and the whole index except `+ j` is a compile-time constant.
enum { POLY_PER_WAVE = 3, POLY_BIAS = 320, POLY_N1 = 32 * 17 + 256 + POLY_BIAS + 1, POLY_N = POLY_N1 + 576 * (POLY_PER_WAVE - 1),
       POLY_ROW = 73 /* odd, >= POLY_N / 32 + 1 */, POLY_ITEM = 32 * POLY_ROW };
#define XT(off) ((double)xt[(((off) + POLY_BIAS) & 31) * ROW + (((off) + POLY_BIAS) >> 5)])

// One 32-band slot.  xt = transposed window base of this lane's slot; a[] lives in registers.
template <int ROW> LHIP_DEV void window_subband(lhip_ctab W, const float* xt, float* a) {
    // reference pointers: p = x - d, q = x - 62 + d with d = i + 15 (NewMDCT.js:540-583)
#pragma unroll
    for (int d = 0; d < 15; d++) {
        const int wp = 10 + 18 * d;
        double w = W[wp - 10];
        double s = XT(-62 + d - 224) * w;
        double t = XT(-d + 224) * w;
#pragma unroll
        for (int k = 1; k < 8; k++) {
            w = W[wp - 10 + k];
            s += XT(-62 + d - 224 + 64 * k) * w;
            t += XT(-d + 224 - 64 * k) * w;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            w = W[wp - 2 + k];
            s += XT(-d - 256 + 64 * k) * w;
            t -= XT(-62 + d + 256 - 64 * k) * w;
        }
        s *= W[wp + 6];
        w = t - s;
        a[2 * d] = (float)(t + s);
        a[2 * d + 1] = (float)(W[wp + 7] * w);
        LHIP_SCHED_FENCE();                       // keep one d's coefficient loads from being hoisted over the others (register pressure)
    }
    const int wp = 10 + 18 * 15;
    {
        // p = x - 15 here (NewMDCT.js:585-622)
#define P(o) XT(-15 + (o))
        double s, t, u, v;
        t = P(-16) * W[wp - 10];
        s = P(-32) * W[wp - 2];
        t += (P(-48) - P(16)) * W[wp - 9];
        s += P(-96) * W[wp - 1];
        t += (P(-80) + P(48)) * W[wp - 8];
        s += P(-160) * W[wp + 0];
        t += (P(-112) - P(80)) * W[wp - 7];
        s += P(-224) * W[wp + 1];
        t += (P(-144) + P(112)) * W[wp - 6];
        s -= P(32) * W[wp + 2];
        t += (P(-176) - P(144)) * W[wp - 5];
        s -= P(96) * W[wp + 3];
        t += (P(-208) + P(176)) * W[wp - 4];
        s -= P(160) * W[wp + 4];
        t += (P(-240) - P(208)) * W[wp - 3];
        s -= P(224);
#undef P
        u = s - t;
        v = s + t;
        t = a[14];
        s = (double)a[15] - t;
        a[31] = (float)(v + t);
        a[30] = (float)(u + s);
        a[15] = (float)(u - s);
        a[14] = (float)(v - t);
    }
    // 32-point butterfly network.  R(i) reads slot i widened to f64, S(i, e) stores e rounded to f32.
    const double c2 = W[wp - 2 * 18 + 7], c4 = W[wp - 4 * 18 + 7], c6 = W[wp - 6 * 18 + 7],
                 c10 = W[wp - 10 * 18 + 7], c12 = W[wp - 12 * 18 + 7], c14 = W[wp - 14 * 18 + 7];
    double z;
#define R(i) ((double)a[i])
#define S(i, e) a[i] = (float)(e)
#define DIFSCALE(hi, lo, cc) { z = R(hi) - R(lo); S(lo, R(lo) + R(hi)); S(hi, z * (cc)); }
    DIFSCALE(28, 0, c2) DIFSCALE(29, 1, c2) DIFSCALE(26, 2, c4) DIFSCALE(27, 3, c4) DIFSCALE(24, 4, c6) DIFSCALE(25, 5, c6)
    z = R(22) - R(6); S(6, R(6) + R(22)); S(22, z * LHIP_SQRT2);
    z = R(23) - R(7); S(7, R(7) + R(23)); S(23, z * LHIP_SQRT2 - R(7));
    S(7, R(7) - R(6)); S(22, R(22) - R(7)); S(23, R(23) - R(22));
    z = R(6);  S(6, R(31) - z);  S(31, R(31) + z);
    z = R(7);  S(7, R(30) - z);  S(30, R(30) + z);
    z = R(22); S(22, R(15) - z); S(15, R(15) + z);
    z = R(23); S(23, R(14) - z); S(14, R(14) + z);
    DIFSCALE(20, 8, c10) DIFSCALE(21, 9, c10) DIFSCALE(18, 10, c12) DIFSCALE(19, 11, c12) DIFSCALE(16, 12, c14) DIFSCALE(17, 13, c14)
    z = -R(20) + R(24); S(20, R(20) + R(24)); S(24, z * c12);
    z = -R(21) + R(25); S(21, R(21) + R(25)); S(25, z * c12);
    z = R(4) - R(8);    S(4, R(4) + R(8));    S(8, z * c12);
    z = R(5) - R(9);    S(5, R(5) + R(9));    S(9, z * c12);
    z = R(0) - R(12);   S(0, R(0) + R(12));   S(12, z * c4);
    z = R(1) - R(13);   S(1, R(1) + R(13));   S(13, z * c4);
    z = R(16) - R(28);  S(16, R(16) + R(28)); S(28, z * c4);
    z = -R(17) + R(29); S(17, R(17) + R(29)); S(29, z * c4);
    z = LHIP_SQRT2 * (R(2) - R(10));   S(2, R(2) + R(10));   S(10, z);
    z = LHIP_SQRT2 * (R(3) - R(11));   S(3, R(3) + R(11));   S(11, z);
    z = LHIP_SQRT2 * (-R(18) + R(26)); S(18, R(18) + R(26)); S(26, z - R(18));
    z = LHIP_SQRT2 * (-R(19) + R(27)); S(19, R(19) + R(27)); S(27, z - R(19));
    z = R(2);  S(19, R(19) - R(3));  S(3, R(3) - z);   S(2, R(31) - z);  S(31, R(31) + z);
    z = R(3);  S(11, R(11) - R(19)); S(18, R(18) - z); S(3, R(30) - z);  S(30, R(30) + z);
    z = R(18); S(27, R(27) - R(11)); S(19, R(19) - z); S(18, R(15) - z); S(15, R(15) + z);
    z = R(19); S(10, R(10) - z); S(19, R(14) - z); S(14, R(14) + z);
    z = R(10); S(11, R(11) - z); S(10, R(23) - z); S(23, R(23) + z);
    z = R(11); S(26, R(26) - z); S(11, R(22) - z); S(22, R(22) + z);
    z = R(26); S(27, R(27) - z); S(26, R(7) - z);  S(7, R(7) + z);
    z = R(27); S(27, R(6) - z);  S(6, R(6) + z);
    z = LHIP_SQRT2 * (R(0) - R(4));   S(0, R(0) + R(4));   S(4, z);
    z = LHIP_SQRT2 * (R(1) - R(5));   S(1, R(1) + R(5));   S(5, z);
    z = LHIP_SQRT2 * (R(16) - R(20)); S(16, R(16) + R(20)); S(20, z);
    z = LHIP_SQRT2 * (R(17) - R(21)); S(17, R(17) + R(21)); S(21, z);
    z = -LHIP_SQRT2 * (R(8) - R(12));  S(8, R(8) + R(12));   S(12, z - R(8));
    z = -LHIP_SQRT2 * (R(9) - R(13));  S(9, R(9) + R(13));   S(13, z - R(9));
    z = -LHIP_SQRT2 * (R(25) - R(29)); S(25, R(25) + R(29)); S(29, z - R(25));
    z = -LHIP_SQRT2 * (R(24) + R(28)); S(24, R(24) - R(28)); S(28, z - R(24));
    // difference chains: the f64 value (not the rounded slot) is carried to the next link
#define LINK0(d, m, s_) { z = R(m) - R(s_); S(d, z); }
#define LINK(d) { z = R(d) - z; S(d, z); }
    LINK0(24, 24, 16) LINK(20) LINK(28)
    LINK0(25, 25, 17) LINK(21) LINK(29)
    LINK0(17, 17, 1) LINK(9) LINK(25) LINK(5) LINK(21) LINK(13) LINK(29)
    LINK0(1, 1, 0) LINK(16) LINK(17) LINK(8) LINK(9) LINK(24) LINK(25) LINK(4) LINK(5) LINK(20) LINK(21) LINK(12) LINK(13) LINK(28) LINK(29)
#define SUMDIF(lo, hi) { z = R(lo); S(lo, R(lo) + R(hi)); S(hi, R(hi) - z); }
    SUMDIF(0, 31) SUMDIF(1, 30) SUMDIF(16, 15) SUMDIF(17, 14) SUMDIF(8, 23) SUMDIF(9, 22) SUMDIF(24, 7) SUMDIF(25, 6)
    SUMDIF(4, 27) SUMDIF(5, 26) SUMDIF(20, 11) SUMDIF(21, 10) SUMDIF(12, 19) SUMDIF(13, 18) SUMDIF(28, 3) SUMDIF(29, 2)
#undef SUMDIF
#undef LINK
#undef LINK0
#undef DIFSCALE
#undef S
#undef R
}

// One wave serves POLY_PER_WAVE work items (granule slot, channel), numbered channel-major so that the items of a
// wave are normally consecutive granules of one channel of one stream: their PCM windows overlap (576-sample hop,
// 1121-sample window), so ONE transposed staging of 2273 samples serves all three and lane u = item * 18 + slot simply
// reads at column offset u (576 = 18 rows of 32).  Waves that straddle a stream boundary take the items one by one.
struct PolyLds { float xs[POLY_ITEM]; };
static_assert(POLY_ITEM >= 18 * POLY_PER_WAVE * 33, "the polyphase output is staged in the PCM area");
// one slot: 32 band values, sign and low-pass scaling applied, stored with stride 1 at `out`
template <int ROW> LHIP_DEV void poly_slot(const Tables& T, const float* xt, float* out, int j) {
    float a[32];
    window_subband<ROW>(LHIP_CTAB(T.enwindow), xt, a);
    if (j & 1)
        for (int band = 1; band < 32; band += 2) a[band] = (float)((double)a[band] * -1);
    // the band filter of the lowpass (NewMDCT.js:1075-1086: `a[order[band]] *= amp_filter[band]` where 1e-12 <= amp_filter < 1): by
    // OUTPUT index, from a table derived at create time -- indexing the register array with a loaded `order[band]` made the
    // compiler search for the register at run time, one memory round trip per band
    {
        lhip_ctab amp = LHIP_CTAB(T.amp_by_out);
#pragma unroll
        for (int i = 0; i < 32; i++)
            if ((T.amp_mask >> i) & 1) a[i] = (float)((double)a[i] * amp[i]);
    }
    for (int i = 0; i < 32; i++) out[i] = a[i];
}
// `cnt` (<= POLY_PER_WAVE) consecutive granule slots gs, gs + 1, ... of ONE stream, channel ch: one transposed staging serves them all
LHIP_DEV void kb_poly_run(const Tables& T, const Workspace& W, const StreamDesc* SD, const StreamIO* IO, int gs, int ch, int cnt, int lane, PolyLds& L) {
    const int C = T.channels_out;
    const int lo = POLY_BIAS - 286;                           // samples before the stream segment are never used
    const int stg = W.gslot_stream[gs];
    const StreamDesc sd = SD[stg];
    const int q = gs - sd.gslot0 - 1;
    if (q < 0) return;                                        // carry slot: nothing to compute
    const PcmSrc P = pcm_source(T, W, sd, IO[stg], ch);
    const int s0 = 576 * q + 286 - POLY_BIAS;                 // segment index of staging slot 0 (slots below `lo` lie before the segment)
    const int n_need = POLY_N1 + 576 * (cnt - 1);
    wave_sync();
    if (!P.plane && s0 + lo >= P.mf) {                        // wave-uniform usual case: everything staged is new Int16 input
        // eight loads in flight per lane: with one load per trip (the trip count is not a compile-time constant) the wave waited
        // out a full memory latency 36 times, which was most of this kernel's time
        const int16_t* src = P.src + (s0 - P.mf);
        enum { STG = 8 };
        for (int n0 = 0; n0 < n_need; n0 += LHIP_NL * STG) {
            int raw[STG];
#pragma unroll
            for (int k = 0; k < STG; k++) {                  // unconditional loads from a clamped index: a predicated load is waited for inside its branch
                const int n = n0 + lane + LHIP_NL * k;
                raw[k] = (int)src[n < lo ? lo : (n < n_need ? n : n_need - 1)];
            }
#pragma unroll
            for (int k = 0; k < STG; k++) LHIP_PIN_LOADED(raw[k]);
#pragma unroll
            for (int k = 0; k < STG; k++) {
                const int n = n0 + lane + LHIP_NL * k;
                float v = (float)raw[k];
                if (P.do_scale) v = (float)((double)v * P.scale);
                if (n < lo) v = 0.f;                            // (0 * scale could be -0 or NaN for an odd scale; the slot is defined as +0)
                if (n < n_need) L.xs[(n & 31) * POLY_ROW + (n >> 5)] = v;
            }
        }
    } else {
        for (int n = lane; n < n_need; n += LHIP_NL) L.xs[(n & 31) * POLY_ROW + (n >> 5)] = (n >= lo) ? pcm_at(P, s0 + n) : 0.f;
    }
    wave_sync();
#if LHIP_NL == 1
    for (int u = lane; u < 18 * cnt; u += LHIP_NL) {
        const int it = u / 18, j = u - 18 * it;
        float a[32];
        poly_slot<POLY_ROW>(T, L.xs + u, a, j);
        float* out = W.sb + ((int64_t)(gs + it) * C + ch) * SB_STRIDE + j * 32;
        for (int i = 0; i < 32; i++) out[i] = a[i];
    }
#else
    // every lane holds one slot (32 values); stored from the registers, a lane would write 128 bytes at a 128-byte stride from
    // its neighbours'.  Through LDS (the PCM window is no longer needed; rows padded to 33 against bank conflicts) the wave
    // writes each item's 2304 bytes as consecutive 256-byte rows instead.
    {
        const int u = lane, it = u / 18, j = u - 18 * it;      // 18 * cnt <= 54 < 64: one slot per lane
        float a[32];
        if (u < 18 * cnt) poly_slot<POLY_ROW>(T, L.xs + u, a, j);
        wave_sync();
        if (u < 18 * cnt) for (int i = 0; i < 32; i++) L.xs[u * 33 + i] = a[i];
        wave_sync();
        for (int k = 0; k < cnt; k++) {
            float* out = W.sb + ((int64_t)(gs + k) * C + ch) * SB_STRIDE;
            for (int i = lane; i < SB_STRIDE; i += LHIP_NL) out[i] = L.xs[(18 * k + (i >> 5)) * 33 + (i & 31)];
        }
    }
#endif
}
LHIP_DEV void kb_polyphase(const Tables& T, const Workspace& W, const StreamDesc* SD, const StreamIO* IO, int wave_idx, int nitems, int lane, PolyLds& L) {
    const int ngs = W.ngslots;
    const int item0 = wave_idx * POLY_PER_WAVE;
    const int nit = (nitems - item0) < POLY_PER_WAVE ? (nitems - item0) : POLY_PER_WAVE;
    // item -> (channel, granule slot), channel-major
    const int ch0 = item0 / ngs, g0 = item0 - ch0 * ngs;
    const int st0 = W.gslot_stream[g0];
    const StreamDesc sd0 = SD[st0];
    const int q0 = g0 - sd0.gslot0 - 1;
    int together = (q0 >= 0);
    for (int it = 1; it < nit; it++) {
        const int item = item0 + it, ch = item / ngs, gs = item - ch * ngs;
        if (ch != ch0 || gs != g0 + it || W.gslot_stream[gs] != st0) together = 0;
    }
    // one pass over all items when they share a staging, else one pass per item (single instance of the slot code)
    const int npass = together ? 1 : nit;
    for (int ps = 0; ps < npass; ps++) {
        const int itA = together ? 0 : ps, cnt = together ? nit : 1;
        const int item = item0 + itA, ch = item / ngs, gs = item - ch * ngs;
        kb_poly_run(T, W, SD, IO, gs, ch, cnt, lane, L);
    }
}

LHIP_DEV void mdct_short3(lhip_ctab ws, float* io) {
    for (int l = 0; l < 3; l++, io++) {
        double tc0, tc1, tc2, ts0, ts1, ts2;
        ts0 = (double)io[6] * ws[0] - (double)io[15];
        tc0 = (double)io[0] * ws[2] - (double)io[9];
        tc1 = ts0 + tc0;
        tc2 = ts0 - tc0;
        ts0 = (double)io[15] * ws[0] + (double)io[6];
        tc0 = (double)io[9] * ws[2] + (double)io[0];
        ts1 = ts0 + tc0;
        ts2 = -ts0 + tc0;
        tc0 = ((double)io[3] * ws[1] - (double)io[12]) * 2.069978111953089e-11;
        ts0 = ((double)io[12] * ws[1] + (double)io[3]) * 2.069978111953089e-11;
        io[0] = (float)(tc1 * 1.907525191737280e-11 + tc0);
        io[15] = (float)(-ts1 * 1.907525191737280e-11 + ts0);
        tc2 = tc2 * 0.86602540378443870761 * 1.907525191737281e-11;
        ts1 = ts1 * 0.5 * 1.907525191737281e-11 + ts0;
        io[3] = (float)(tc2 - ts1);
        io[6] = (float)(tc2 + ts1);
        tc1 = tc1 * 0.5 * 1.907525191737281e-11 - tc0;
        ts2 = ts2 * 0.86602540378443870761 * 1.907525191737281e-11;
        io[9] = (float)(tc1 + ts2);
        io[12] = (float)(tc1 - ts2);
    }
}

LHIP_DEV void mdct_long18(lhip_ctab cx, float* out, const float* in) {
    double ct, st;
#define I(k) ((double)in[k])
    {
        double tc1 = I(17) - I(9), tc3 = I(15) - I(11), tc4 = I(14) - I(12);
        double ts5 = I(0) + I(8), ts6 = I(1) + I(7), ts7 = I(2) + I(6), ts8 = I(3) + I(5);
        out[17] = (float)((ts5 + ts7 - ts8) - (ts6 - I(4)));
        st = (ts5 + ts7 - ts8) * cx[7] + (ts6 - I(4));
        ct = (tc1 - tc3 - tc4) * cx[6];
        out[5] = (float)(ct + st);
        out[6] = (float)(ct - st);
        double tc2 = (I(16) - I(10)) * cx[6];
        ts6 = ts6 * cx[7] + I(4);
        ct = tc1 * cx[0] + tc2 + tc3 * cx[1] + tc4 * cx[2];
        st = -ts5 * cx[4] + ts6 - ts7 * cx[5] + ts8 * cx[3];
        out[1] = (float)(ct + st);
        out[2] = (float)(ct - st);
        ct = tc1 * cx[1] - tc2 - tc3 * cx[2] + tc4 * cx[0];
        st = -ts5 * cx[5] + ts6 - ts7 * cx[3] + ts8 * cx[4];
        out[9] = (float)(ct + st);
        out[10] = (float)(ct - st);
        ct = tc1 * cx[2] - tc2 + tc3 * cx[0] - tc4 * cx[1];
        st = ts5 * cx[3] - ts6 + ts7 * cx[4] - ts8 * cx[5];
        out[13] = (float)(ct + st);
        out[14] = (float)(ct - st);
    }
    {
        double ts1 = I(8) - I(0), ts3 = I(6) - I(2), ts4 = I(5) - I(3);
        double tc5 = I(17) + I(9), tc6 = I(16) + I(10), tc7 = I(15) + I(11), tc8 = I(14) + I(12);
        out[0] = (float)((tc5 + tc7 + tc8) + (tc6 + I(13)));
        ct = (tc5 + tc7 + tc8) * cx[7] - (tc6 + I(13));
        st = (ts1 - ts3 + ts4) * cx[6];
        out[11] = (float)(ct + st);
        out[12] = (float)(ct - st);
        double ts2 = (I(7) - I(1)) * cx[6];
        tc6 = I(13) - tc6 * cx[7];
        ct = tc5 * cx[3] - tc6 + tc7 * cx[4] + tc8 * cx[5];
        st = ts1 * cx[2] + ts2 + ts3 * cx[0] + ts4 * cx[1];
        out[3] = (float)(ct + st);
        out[4] = (float)(ct - st);
        ct = -tc5 * cx[5] + tc6 - tc7 * cx[3] - tc8 * cx[4];
        st = ts1 * cx[1] + ts2 - ts3 * cx[2] - ts4 * cx[0];
        out[7] = (float)(ct + st);
        out[8] = (float)(ct - st);
        ct = -tc5 * cx[4] + tc6 - tc7 * cx[5] - tc8 * cx[3];
        st = ts1 * cx[0] - ts2 + ts3 * cx[1] - ts4 * cx[2];
        out[15] = (float)(ct + st);
        out[16] = (float)(ct - st);
    }
#undef I
}

struct MdctLds { float xr[2][576]; };

// windowing + MDCT of one band of one granule: band0 / band1 = previous / current polyphase output of that band, slot r at
// [r * RS]; enc = the band's 18 spectral lines
template <int RS> LHIP_DEV void mdct_band(const Tables& T, const float* band0, const float* band1, int type, int band, float* enc) {
    lhip_ctab win = LHIP_CTAB(T.mdct_win);
#define B0(r) ((double)band0[(r) * RS])
#define B1(r) ((double)band1[(r) * RS])
    if ((double)T.amp_filter[band] < 1e-12) {
        for (int k = 0; k < 18; k++) enc[k] = 0.f;
    } else if (type == SHORT_TYPE) {
        lhip_ctab ws = win + 2 * 36;
        float v[18];
        for (int k = -3; k < 0; k++) {
            const double w = ws[k + 3];
            v[k * 3 + 9] = (float)(B0(9 + k) * w - B0(8 - k));
            v[k * 3 + 18] = (float)(B0(14 - k) * w + B0(15 + k));
            v[k * 3 + 10] = (float)(B0(15 + k) * w - B0(14 - k));
            v[k * 3 + 19] = (float)(B1(2 - k) * w + B1(3 + k));
            v[k * 3 + 11] = (float)(B1(3 + k) * w - B1(2 - k));
            v[k * 3 + 20] = (float)(B1(8 - k) * w + B1(9 + k));
        }
        mdct_short3(ws, v);
        for (int k = 0; k < 18; k++) enc[k] = v[k];
    } else {
        float work[18], o18[18];
        lhip_ctab wt = win + type * 36;
        lhip_ctab tantab = win + 2 * 36 + 3;
        for (int k = -9; k < 0; k++) {
            const double a = wt[k + 27] * B1(k + 9) + wt[k + 36] * B1(8 - k);
            const double b = wt[k + 9] * B0(k + 9) - wt[k + 18] * B0(8 - k);
            work[k + 9] = (float)(a - b * tantab[k + 9]);
            work[k + 18] = (float)(a * tantab[k + 9] + b);
        }
        mdct_long18(win + 2 * 36 + 12, o18, work);
        for (int k = 0; k < 18; k++) enc[k] = o18[k];
    }
#undef B0
#undef B1
}
// alias-reduction butterflies between band - 1 and band (non-short blocks, band >= 1); enc = the band's lines
LHIP_DEV void alias_band(const Tables& T, float* enc) {
    lhip_ctab ca = LHIP_CTAB(T.mdct_win) + 2 * 36 + 20;
    lhip_ctab cs = LHIP_CTAB(T.mdct_win) + 2 * 36 + 28;
    for (int k = 7; k >= 0; --k) {
        const double bu = (double)enc[k] * ca[k] + (double)enc[-1 - k] * cs[k];
        const double bd = (double)enc[k] * cs[k] - (double)enc[-1 - k] * ca[k];
        enc[-1 - k] = (float)bu;
        enc[k] = (float)bd;
    }
}

// one wave per granule slot >= 1; lane = ch * 32 + band
LHIP_DEV void kb_mdct(const Tables& T, const Workspace& W, const StreamDesc* SD, int gslot, int lane, MdctLds& L) {
    const int C = T.channels_out;
    const int st = W.gslot_stream[gslot];
    const StreamDesc sd = SD[st];
    if (gslot - sd.gslot0 - 1 < 0) return;
    LHIP_LANE_ONCE(it, 0, C * 32) {                            // C <= 2: at most one (channel, band) per lane
        const int ch = it >> 5, band = it & 31;
        const int ob = T.mdct_order[band];
        mdct_band<32>(T, W.sb + ((int64_t)(gslot - 1) * C + ch) * SB_STRIDE + ob, W.sb + ((int64_t)gslot * C + ch) * SB_STRIDE + ob,
                      W.blocktype[(int64_t)gslot * C + ch], band, L.xr[ch] + 18 * band);
    }
    wave_sync();
    LHIP_LANE_ONCE(it, 0, C * 32) {                            // disjoint element pairs
        const int ch = it >> 5, band = it & 31;
        if (W.blocktype[(int64_t)gslot * C + ch] != SHORT_TYPE && band != 0) alias_band(T, L.xr[ch] + 18 * band);
    }
    wave_sync();
    for (int ch = 0; ch < C; ch++)
        for (int i = lane; i < 576; i += LHIP_NL) W.xr[((int64_t)gslot * C + ch) * 576 + i] = L.xr[ch][i];
}

}  // namespace lhip
