/*
 * TEST INFRASTRUCTURE (CPU oracle) -- polyphase analysis + MDCT.
 * Restates reference NewMDCT.js: window_subband (534-914), mdct_short (927-979),
 * mdct_long (981-1051), mdct_sub48 (1053-1161).  Operation order is the contract:
 * f64 arithmetic, an f32 rounding at every store into the 32-slot work row / xr.
 */
#include "lo_common.h"

/* one 32-band polyphase slot: x points at the newest-sample anchor (reference's x1[x1Pos]) */
static void lo_window_subband(const lo_cfg* c, const float* x, float* a) {
    const double* W = c->enwindow;
    const float* p = x;            /* walks down  */
    const float* q = x - 62;       /* walks up (238 - 14 - 286) */
    int wp = 10, i, k;
    for (i = -15; i < 0; i++) {
        double w, s, t;
        w = W[wp - 10];
        s = D(q[-224]) * w;
        t = D(p[224]) * w;
        for (k = 1; k < 8; k++) {
            w = W[wp - 10 + k];
            s += D(q[-224 + 64 * k]) * w;
            t += D(p[224 - 64 * k]) * w;
        }
        for (k = 0; k < 8; k++) {
            w = W[wp - 2 + k];
            s += D(p[-256 + 64 * k]) * w;
            t -= D(q[256 - 64 * k]) * w;
        }
        s *= W[wp + 6];
        w = t - s;
        a[30 + i * 2] = (float)(t + s);
        a[31 + i * 2] = (float)(W[wp + 7] * w);
        wp += 18;
        p--;
        q++;
    }
    {
        double s, t, u, v;
        t = D(p[-16]) * W[wp - 10];
        s = D(p[-32]) * W[wp - 2];
        t += (D(p[-48]) - D(p[16])) * W[wp - 9];
        s += D(p[-96]) * W[wp - 1];
        t += (D(p[-80]) + D(p[48])) * W[wp - 8];
        s += D(p[-160]) * W[wp + 0];
        t += (D(p[-112]) - D(p[80])) * W[wp - 7];
        s += D(p[-224]) * W[wp + 1];
        t += (D(p[-144]) + D(p[112])) * W[wp - 6];
        s -= D(p[32]) * W[wp + 2];
        t += (D(p[-176]) - D(p[144])) * W[wp - 5];
        s -= D(p[96]) * W[wp + 3];
        t += (D(p[-208]) + D(p[176])) * W[wp - 4];
        s -= D(p[160]) * W[wp + 4];
        t += (D(p[-240]) - D(p[208])) * W[wp - 3];
        s -= D(p[224]);
        u = s - t;
        v = s + t;
        t = a[14];
        s = D(a[15]) - t;
        a[31] = (float)(v + t);
        a[30] = (float)(u + s);
        a[15] = (float)(u - s);
        a[14] = (float)(v - t);
    }
    {
        /* in-place 32-point butterfly network; every assignment rounds to f32 */
        double xr;
        const double c2 = W[wp - 2 * 18 + 7], c4 = W[wp - 4 * 18 + 7], c6 = W[wp - 6 * 18 + 7],
            c10 = W[wp - 10 * 18 + 7], c12 = W[wp - 12 * 18 + 7], c14 = W[wp - 14 * 18 + 7];
#define A(i) D(a[i])
#define SET(i, e) a[i] = (float)(e)
        /* stage: (hi - lo) * c ; lo += hi */
#define ROT(hi, lo, cc) do { xr = A(hi) - A(lo); SET(lo, A(lo) + A(hi)); SET(hi, xr * (cc)); } while (0)
        ROT(28, 0, c2);  ROT(29, 1, c2);
        ROT(26, 2, c4);  ROT(27, 3, c4);
        ROT(24, 4, c6);  ROT(25, 5, c6);
        xr = A(22) - A(6); SET(6, A(6) + A(22)); SET(22, xr * SQRT2);
        xr = A(23) - A(7); SET(7, A(7) + A(23)); SET(23, xr * SQRT2 - A(7));
        SET(7, A(7) - A(6));
        SET(22, A(22) - A(7));
        SET(23, A(23) - A(22));

        xr = A(6);  SET(6, A(31) - xr);  SET(31, A(31) + xr);
        xr = A(7);  SET(7, A(30) - xr);  SET(30, A(30) + xr);
        xr = A(22); SET(22, A(15) - xr); SET(15, A(15) + xr);
        xr = A(23); SET(23, A(14) - xr); SET(14, A(14) + xr);

        ROT(20, 8, c10);  ROT(21, 9, c10);
        ROT(18, 10, c12); ROT(19, 11, c12);
        ROT(16, 12, c14); ROT(17, 13, c14);

        xr = -A(20) + A(24); SET(20, A(20) + A(24)); SET(24, xr * c12);
        xr = -A(21) + A(25); SET(21, A(21) + A(25)); SET(25, xr * c12);
        xr = A(4) - A(8);    SET(4, A(4) + A(8));    SET(8, xr * c12);
        xr = A(5) - A(9);    SET(5, A(5) + A(9));    SET(9, xr * c12);
        xr = A(0) - A(12);   SET(0, A(0) + A(12));   SET(12, xr * c4);
        xr = A(1) - A(13);   SET(1, A(1) + A(13));   SET(13, xr * c4);
        xr = A(16) - A(28);  SET(16, A(16) + A(28)); SET(28, xr * c4);
        xr = -A(17) + A(29); SET(17, A(17) + A(29)); SET(29, xr * c4);

        xr = SQRT2 * (A(2) - A(10));   SET(2, A(2) + A(10));   SET(10, xr);
        xr = SQRT2 * (A(3) - A(11));   SET(3, A(3) + A(11));   SET(11, xr);
        xr = SQRT2 * (-A(18) + A(26)); SET(18, A(18) + A(26)); SET(26, xr - A(18));
        xr = SQRT2 * (-A(19) + A(27)); SET(19, A(19) + A(27)); SET(27, xr - A(19));

        xr = A(2);  SET(19, A(19) - A(3)); SET(3, A(3) - xr);   SET(2, A(31) - xr);  SET(31, A(31) + xr);
        xr = A(3);  SET(11, A(11) - A(19)); SET(18, A(18) - xr); SET(3, A(30) - xr);  SET(30, A(30) + xr);
        xr = A(18); SET(27, A(27) - A(11)); SET(19, A(19) - xr); SET(18, A(15) - xr); SET(15, A(15) + xr);

        xr = A(19); SET(10, A(10) - xr); SET(19, A(14) - xr); SET(14, A(14) + xr);
        xr = A(10); SET(11, A(11) - xr); SET(10, A(23) - xr); SET(23, A(23) + xr);
        xr = A(11); SET(26, A(26) - xr); SET(11, A(22) - xr); SET(22, A(22) + xr);
        xr = A(26); SET(27, A(27) - xr); SET(26, A(7) - xr);  SET(7, A(7) + xr);
        xr = A(27); SET(27, A(6) - xr);  SET(6, A(6) + xr);

        xr = SQRT2 * (A(0) - A(4));   SET(0, A(0) + A(4));   SET(4, xr);
        xr = SQRT2 * (A(1) - A(5));   SET(1, A(1) + A(5));   SET(5, xr);
        xr = SQRT2 * (A(16) - A(20)); SET(16, A(16) + A(20)); SET(20, xr);
        xr = SQRT2 * (A(17) - A(21)); SET(17, A(17) + A(21)); SET(21, xr);

        xr = -SQRT2 * (A(8) - A(12));  SET(8, A(8) + A(12));   SET(12, xr - A(8));
        xr = -SQRT2 * (A(9) - A(13));  SET(9, A(9) + A(13));   SET(13, xr - A(9));
        xr = -SQRT2 * (A(25) - A(29)); SET(25, A(25) + A(29)); SET(29, xr - A(25));
        xr = -SQRT2 * (A(24) + A(28)); SET(24, A(24) - A(28)); SET(28, xr - A(24));

        /* running differences: each store feeds the next */
/* the reference keeps the *unrounded* f64 `xr` between links (xr = a - xr; a = xr) */
#define CHAIN0(dst, m, s_) do { xr = A(m) - A(s_); SET(dst, xr); } while (0)
#define CHAIN(dst) do { xr = A(dst) - xr; SET(dst, xr); } while (0)
        CHAIN0(24, 24, 16); CHAIN(20); CHAIN(28);
        CHAIN0(25, 25, 17); CHAIN(21); CHAIN(29);
        CHAIN0(17, 17, 1);  CHAIN(9);  CHAIN(25); CHAIN(5); CHAIN(21); CHAIN(13); CHAIN(29);
        CHAIN0(1, 1, 0);    CHAIN(16); CHAIN(17); CHAIN(8); CHAIN(9);  CHAIN(24); CHAIN(25);
        CHAIN(4); CHAIN(5); CHAIN(20); CHAIN(21); CHAIN(12); CHAIN(13); CHAIN(28); CHAIN(29);

        /* final sum/difference pairs */
#define FIN(lo, hi) do { xr = A(lo); SET(lo, A(lo) + A(hi)); SET(hi, A(hi) - xr); } while (0)
        FIN(0, 31);  FIN(1, 30);  FIN(16, 15); FIN(17, 14);
        FIN(8, 23);  FIN(9, 22);  FIN(24, 7);  FIN(25, 6);
        FIN(4, 27);  FIN(5, 26);  FIN(20, 11); FIN(21, 10);
        FIN(12, 19); FIN(13, 18); FIN(28, 3);  FIN(29, 2);
#undef FIN
#undef CHAIN
#undef CHAIN0
#undef ROT
#undef SET
#undef A
    }
}

/* three 12-point MDCTs in place on xr[pos .. pos+17] (interleaved by 3) */
static void lo_mdct_short(const lo_cfg* c, float* io) {
    const double* ws = c->mdct_win + 2 * 36;   /* win[SHORT_TYPE] */
    int l;
    for (l = 0; l < 3; l++, io++) {
        double tc0, tc1, tc2, ts0, ts1, ts2;
        ts0 = D(io[2 * 3]) * ws[0] - D(io[5 * 3]);
        tc0 = D(io[0 * 3]) * ws[2] - D(io[3 * 3]);
        tc1 = ts0 + tc0;
        tc2 = ts0 - tc0;
        ts0 = D(io[5 * 3]) * ws[0] + D(io[2 * 3]);
        tc0 = D(io[3 * 3]) * ws[2] + D(io[0 * 3]);
        ts1 = ts0 + tc0;
        ts2 = -ts0 + tc0;
        tc0 = (D(io[1 * 3]) * ws[1] - D(io[4 * 3])) * 2.069978111953089e-11;
        ts0 = (D(io[4 * 3]) * ws[1] + D(io[1 * 3])) * 2.069978111953089e-11;
        io[3 * 0] = (float)(tc1 * 1.907525191737280e-11 + tc0);
        io[3 * 5] = (float)(-ts1 * 1.907525191737280e-11 + ts0);
        tc2 = tc2 * 0.86602540378443870761 * 1.907525191737281e-11;
        ts1 = ts1 * 0.5 * 1.907525191737281e-11 + ts0;
        io[3 * 1] = (float)(tc2 - ts1);
        io[3 * 2] = (float)(tc2 + ts1);
        tc1 = tc1 * 0.5 * 1.907525191737281e-11 - tc0;
        ts2 = ts2 * 0.86602540378443870761 * 1.907525191737281e-11;
        io[3 * 3] = (float)(tc1 + ts2);
        io[3 * 4] = (float)(tc1 - ts2);
    }
}

/* 36 -> 18 MDCT on the pre-twiddled work vector in[18] */
static void lo_mdct_long(const lo_cfg* c, float* out, const float* in) {
    const double* cx = c->mdct_win + 2 * 36 + 12;   /* win[SHORT_TYPE][12 + k] */
    double ct, st;
#define I(k) D(in[k])
    {
        double tc1, tc2, tc3, tc4, ts5, ts6, ts7, ts8;
        tc1 = I(17) - I(9);
        tc3 = I(15) - I(11);
        tc4 = I(14) - I(12);
        ts5 = I(0) + I(8);
        ts6 = I(1) + I(7);
        ts7 = I(2) + I(6);
        ts8 = I(3) + I(5);
        out[17] = (float)((ts5 + ts7 - ts8) - (ts6 - I(4)));
        st = (ts5 + ts7 - ts8) * cx[7] + (ts6 - I(4));
        ct = (tc1 - tc3 - tc4) * cx[6];
        out[5] = (float)(ct + st);
        out[6] = (float)(ct - st);
        tc2 = (I(16) - I(10)) * cx[6];
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
        double ts1, ts2, ts3, ts4, tc5, tc6, tc7, tc8;
        ts1 = I(8) - I(0);
        ts3 = I(6) - I(2);
        ts4 = I(5) - I(3);
        tc5 = I(17) + I(9);
        tc6 = I(16) + I(10);
        tc7 = I(15) + I(11);
        tc8 = I(14) + I(12);
        out[0] = (float)((tc5 + tc7 + tc8) + (tc6 + I(13)));
        ct = (tc5 + tc7 + tc8) * cx[7] - (tc6 + I(13));
        st = (ts1 - ts3 + ts4) * cx[6];
        out[11] = (float)(ct + st);
        out[12] = (float)(ct - st);
        ts2 = (I(7) - I(1)) * cx[6];
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

/* w0/w1: per-channel PCM windows (the reference's inbuf, index 0 == mfbuf[0]) */
static void lo_mdct_sub48(lo_enc* e, const float* w0, const float* w1) {
    const lo_cfg* c = &e->c;
    const int32_t* order = c->mdct_order;
    const float* wk = w0;
    int ch, gr, k, band;
    for (ch = 0; ch < c->channels_out; ch++) {
        int wkPos = 286;
        for (gr = 0; gr < c->mode_gr; gr++) {
            lo_gr* gi = &e->tt[gr][ch];
            float* enc = gi->xr;
            float (*samp)[32] = e->sb_sample[ch][1 - gr];
            for (k = 0; k < 9; k++) {
                lo_window_subband(c, wk + wkPos, samp[2 * k]);
                lo_window_subband(c, wk + wkPos + 32, samp[2 * k + 1]);
                wkPos += 64;
                for (band = 1; band < 32; band += 2) samp[2 * k + 1][band] = (float)(D(samp[2 * k + 1][band]) * -1);
            }
            for (band = 0; band < 32; band++, enc += 18) {
                int type = gi->block_type;
                float (*band0)[32] = e->sb_sample[ch][gr];
                float (*band1)[32] = e->sb_sample[ch][1 - gr];
                const int ob = order[band];
                if (gi->mixed_block_flag != 0 && band < 2) type = 0;
                if (D(c->amp_filter[band]) < 1e-12) {
                    memset(enc, 0, 18 * sizeof(float));
                } else {
                    if (D(c->amp_filter[band]) < 1.0)
                        for (k = 0; k < 18; k++) band1[k][ob] = (float)(D(band1[k][ob]) * D(c->amp_filter[band]));
                    if (type == SHORT_TYPE) {
                        const double* ws = c->mdct_win + 2 * 36;
                        for (k = -3; k < 0; k++) {
                            double w = ws[k + 3];
                            enc[k * 3 + 9] = (float)(D(band0[9 + k][ob]) * w - D(band0[8 - k][ob]));
                            enc[k * 3 + 18] = (float)(D(band0[14 - k][ob]) * w + D(band0[15 + k][ob]));
                            enc[k * 3 + 10] = (float)(D(band0[15 + k][ob]) * w - D(band0[14 - k][ob]));
                            enc[k * 3 + 19] = (float)(D(band1[2 - k][ob]) * w + D(band1[3 + k][ob]));
                            enc[k * 3 + 11] = (float)(D(band1[3 + k][ob]) * w - D(band1[2 - k][ob]));
                            enc[k * 3 + 20] = (float)(D(band1[8 - k][ob]) * w + D(band1[9 + k][ob]));
                        }
                        lo_mdct_short(c, enc);
                    } else {
                        float work[18];
                        const double* wt = c->mdct_win + type * 36;
                        const double* tantab = c->mdct_win + 2 * 36 + 3;
                        for (k = -9; k < 0; k++) {
                            double a, b;
                            a = wt[k + 27] * D(band1[k + 9][ob]) + wt[k + 36] * D(band1[8 - k][ob]);
                            b = wt[k + 9] * D(band0[k + 9][ob]) - wt[k + 18] * D(band0[8 - k][ob]);
                            work[k + 9] = (float)(a - b * tantab[k + 9]);
                            work[k + 18] = (float)(a * tantab[k + 9] + b);
                        }
                        lo_mdct_long(c, enc, work);
                    }
                }
                /* aliasing reduction butterflies against the previous band */
                if (type != SHORT_TYPE && band != 0) {
                    const double* ca = c->mdct_win + 2 * 36 + 20;
                    const double* cs = c->mdct_win + 2 * 36 + 28;
                    for (k = 7; k >= 0; --k) {
                        double bu, bd;
                        bu = D(enc[k]) * ca[k] + D(enc[-1 - k]) * cs[k];
                        bd = D(enc[k]) * cs[k] - D(enc[-1 - k]) * ca[k];
                        enc[-1 - k] = (float)bu;
                        enc[k] = (float)bd;
                    }
                }
            }
        }
        wk = w1;
        if (c->mode_gr == 1)                    /* NewMDCT.js:1154-1159: the single granule becomes "previous" */
            memcpy(e->sb_sample[ch][0], e->sb_sample[ch][1], sizeof e->sb_sample[ch][0]);
    }
}
