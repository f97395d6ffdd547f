/*
 * TEST INFRASTRUCTURE (CPU oracle) -- psychoacoustic model (nspsytune), one granule per call.
 * Restates reference PsyModel.js L3psycho_anal_ns (1000-1383) with its helpers
 * compute_ffts (251-324), mask_add (403-473), calc_interchannel_masking (525-543),
 * convert_partition2scalefac_s/l (644-734), compute_masking_s (736-782), block_type_set
 * (784-826), calc_energy / calc_mask_index_l (906-992), FFT.js fht/fft_short/fft_long
 * (31-224), and Encoder.js adjust_ATH (166-243).  Perceptual-entropy (pecalc_*) is not
 * computed: with the reservoir disabled it cannot reach the output.
 */
#include "lo_common.h"

/* ---- FHT (FFT.js:31-115), in place on n2 = 2*n floats ---- */
static void lo_fht(const lo_cfg* c, float* fz, int n) {
    const double* costab = c->fht_costab;
    int tri = 0, k4 = 4;
    float *fi, *gi, *fn;
    n <<= 1;
    fn = fz + n;
    do {
        double s1, c1;
        int i, k1, k2, k3, kx;
        kx = k4 >> 1;
        k1 = k4;
        k2 = k4 << 1;
        k3 = k2 + k1;
        k4 = k2 << 1;
        fi = fz;
        gi = fi + kx;
        do {
            double f0, f1, f2, f3;
            f1 = D(fi[0]) - D(fi[k1]);
            f0 = D(fi[0]) + D(fi[k1]);
            f3 = D(fi[k2]) - D(fi[k3]);
            f2 = D(fi[k2]) + D(fi[k3]);
            fi[k2] = (float)(f0 - f2);
            fi[0] = (float)(f0 + f2);
            fi[k3] = (float)(f1 - f3);
            fi[k1] = (float)(f1 + f3);
            f1 = D(gi[0]) - D(gi[k1]);
            f0 = D(gi[0]) + D(gi[k1]);
            f3 = SQRT2 * D(gi[k3]);
            f2 = SQRT2 * D(gi[k2]);
            gi[k2] = (float)(f0 - f2);
            gi[0] = (float)(f0 + f2);
            gi[k3] = (float)(f1 - f3);
            gi[k1] = (float)(f1 + f3);
            gi += k4;
            fi += k4;
        } while (fi < fn);
        c1 = costab[tri + 0];
        s1 = costab[tri + 1];
        for (i = 1; i < kx; i++) {
            double c2, s2;
            c2 = 1 - (2 * s1) * s1;
            s2 = (2 * s1) * c1;
            fi = fz + i;
            gi = fz + k1 - i;
            do {
                double a, b, g0, f0, f1, g1, f2, g2, f3, g3;
                b = s2 * D(fi[k1]) - c2 * D(gi[k1]);
                a = c2 * D(fi[k1]) + s2 * D(gi[k1]);
                f1 = D(fi[0]) - a;
                f0 = D(fi[0]) + a;
                g1 = D(gi[0]) - b;
                g0 = D(gi[0]) + b;
                b = s2 * D(fi[k3]) - c2 * D(gi[k3]);
                a = c2 * D(fi[k3]) + s2 * D(gi[k3]);
                f3 = D(fi[k2]) - a;
                f2 = D(fi[k2]) + a;
                g3 = D(gi[k2]) - b;
                g2 = D(gi[k2]) + b;
                b = s1 * f2 - c1 * g3;
                a = c1 * f2 + s1 * g3;
                fi[k2] = (float)(f0 - a);
                fi[0] = (float)(f0 + a);
                gi[k3] = (float)(g1 - b);
                gi[k1] = (float)(g1 + b);
                b = c1 * g2 - s1 * f3;
                a = s1 * g2 + c1 * f3;
                gi[k2] = (float)(g0 - a);
                gi[0] = (float)(g0 + a);
                fi[k3] = (float)(f1 - b);
                fi[k1] = (float)(f1 + b);
                gi += k4;
                fi += k4;
            } while (fi < fn);
            c2 = c1;
            c1 = c2 * costab[tri + 0] - s1 * costab[tri + 1];
            s1 = c2 * costab[tri + 1] + s1 * costab[tri + 0];
        }
        tri += 2;
    } while (k4 < n);
}

static void lo_fft_long(const lo_cfg* c, float* y, const float* buf) {
    const float* win = c->window;
    int jj = BLKSIZE / 8 - 1;
    float* x = y + BLKSIZE / 2;
    do {
        double f0, f1, f2, f3, w;
        int i = c->fft_rv_tbl[jj] & 0xff;
        f0 = D(win[i]) * D(buf[i]);
        w = D(win[i + 0x200]) * D(buf[i + 0x200]);
        f1 = f0 - w; f0 = f0 + w;
        f2 = D(win[i + 0x100]) * D(buf[i + 0x100]);
        w = D(win[i + 0x300]) * D(buf[i + 0x300]);
        f3 = f2 - w; f2 = f2 + w;
        x -= 4;
        x[0] = (float)(f0 + f2);
        x[2] = (float)(f0 - f2);
        x[1] = (float)(f1 + f3);
        x[3] = (float)(f1 - f3);
        f0 = D(win[i + 0x001]) * D(buf[i + 0x001]);
        w = D(win[i + 0x201]) * D(buf[i + 0x201]);
        f1 = f0 - w; f0 = f0 + w;
        f2 = D(win[i + 0x101]) * D(buf[i + 0x101]);
        w = D(win[i + 0x301]) * D(buf[i + 0x301]);
        f3 = f2 - w; f2 = f2 + w;
        x[BLKSIZE / 2 + 0] = (float)(f0 + f2);
        x[BLKSIZE / 2 + 2] = (float)(f0 - f2);
        x[BLKSIZE / 2 + 1] = (float)(f1 + f3);
        x[BLKSIZE / 2 + 3] = (float)(f1 - f3);
    } while (--jj >= 0);
    lo_fht(c, x, BLKSIZE / 2);
}

static void lo_fft_short(const lo_cfg* c, float xs[3][BLKSIZE_s], const float* buf) {
    const float* win = c->window_s;
    int b;
    for (b = 0; b < 3; b++) {
        float* x = xs[b] + BLKSIZE_s / 2;
        int k = (576 / 3) * (b + 1);
        int j = BLKSIZE_s / 8 - 1;
        do {
            double f0, f1, f2, f3, w;
            int i = c->fft_rv_tbl[j << 2] & 0xff;
            f0 = D(win[i]) * D(buf[i + k]);
            w = D(win[0x7f - i]) * D(buf[i + k + 0x80]);
            f1 = f0 - w; f0 = f0 + w;
            f2 = D(win[i + 0x40]) * D(buf[i + k + 0x40]);
            w = D(win[0x3f - i]) * D(buf[i + k + 0xc0]);
            f3 = f2 - w; f2 = f2 + w;
            x -= 4;
            x[0] = (float)(f0 + f2);
            x[2] = (float)(f0 - f2);
            x[1] = (float)(f1 + f3);
            x[3] = (float)(f1 - f3);
            f0 = D(win[i + 0x01]) * D(buf[i + k + 0x01]);
            w = D(win[0x7e - i]) * D(buf[i + k + 0x81]);
            f1 = f0 - w; f0 = f0 + w;
            f2 = D(win[i + 0x41]) * D(buf[i + k + 0x41]);
            w = D(win[0x3e - i]) * D(buf[i + k + 0xc1]);
            f3 = f2 - w; f2 = f2 + w;
            x[BLKSIZE_s / 2 + 0] = (float)(f0 + f2);
            x[BLKSIZE_s / 2 + 2] = (float)(f0 - f2);
            x[BLKSIZE_s / 2 + 1] = (float)(f1 + f3);
            x[BLKSIZE_s / 2 + 3] = (float)(f1 - f3);
        } while (--j >= 0);
        lo_fht(c, x, BLKSIZE_s / 2);
    }
}

/* ---- mask_add (PsyModel.js:403-473); long blocks only on this path ---- */
static double lo_mask_add(const lo_enc* e, double m1, double m2, int kk, int b) {
    const lo_cfg* c = &e->c;
    double ratio;
    int i;
    if (m2 > m1) {
        if (m2 < (m1 * c->ma_max_i2)) ratio = m2 / m1;
        else return (m1 + m2);
    } else {
        if (m1 >= (m2 * c->ma_max_i2)) return (m1 + m2);
        ratio = m1 / m2;
    }
    m1 += m2;
    if ((b + 3) <= 3 + 3) {             /* signed compare: true for every b <= 3 */
        if (ratio >= c->ma_max_i1) return m1;
        i = js_toint32(v8_log10(ratio) * 16.0);
        return m1 * c->ma_table2[i];
    }
    i = js_toint32(v8_log10(ratio) * 16.0);
    m2 = D(c->ATH_cb_l[kk]) * e->ATH_adjust;
    if (m1 < c->ma_max_m * m2) {
        if (m1 > m2) {
            double f = 1.0, r;
            if (i <= 13) f = c->ma_table3[i];
            r = v8_log10(m1 / m2) * (10.0 / 15.0);
            return m1 * ((c->ma_table1[i] - f) * r + f);
        }
        if (i > 13) return m1;
        return m1 * c->ma_table3[i];
    }
    return m1 * c->ma_table1[i];
}

static void lo_convert_p2s_s(lo_enc* e, const float* eb, const float* thr, int chn, int sblock) {
    const lo_cfg* c = &e->c;
    int sb, b;
    double enn = 0.0, thmm = 0.0;
    for (sb = b = 0; sb < SBMAX_s; ++b, ++sb) {
        int bo = c->bo_s[sb], npart = c->npart_s;
        int b_lim = bo < npart ? bo : npart;
        while (b < b_lim) { enn += D(eb[b]); thmm += D(thr[b]); b++; }
        e->en[chn].s[sb][sblock] = (float)enn;
        e->thm[chn].s[sb][sblock] = (float)thmm;
        if (b >= npart) { ++sb; break; }
        {
            double w_curr = c->bo_s_weight[sb], w_next = 1.0 - w_curr;
            enn = w_curr * D(eb[b]);
            thmm = w_curr * D(thr[b]);
            e->en[chn].s[sb][sblock] = (float)(D(e->en[chn].s[sb][sblock]) + enn);
            e->thm[chn].s[sb][sblock] = (float)(D(e->thm[chn].s[sb][sblock]) + thmm);
            enn = w_next * D(eb[b]);
            thmm = w_next * D(thr[b]);
        }
    }
    for (; sb < SBMAX_s; ++sb) { e->en[chn].s[sb][sblock] = 0; e->thm[chn].s[sb][sblock] = 0; }
}

static void lo_convert_p2s_l(lo_enc* e, const float* eb, const float* thr, int chn) {
    const lo_cfg* c = &e->c;
    int sb, b;
    double enn = 0.0, thmm = 0.0;
    for (sb = b = 0; sb < SBMAX_l; ++b, ++sb) {
        int bo = c->bo_l[sb], npart = c->npart_l;
        int b_lim = bo < npart ? bo : npart;
        while (b < b_lim) { enn += D(eb[b]); thmm += D(thr[b]); b++; }
        e->en[chn].l[sb] = (float)enn;
        e->thm[chn].l[sb] = (float)thmm;
        if (b >= npart) { ++sb; break; }
        {
            double w_curr = c->bo_l_weight[sb], w_next = 1.0 - w_curr;
            enn = w_curr * D(eb[b]);
            thmm = w_curr * D(thr[b]);
            e->en[chn].l[sb] = (float)(D(e->en[chn].l[sb]) + enn);
            e->thm[chn].l[sb] = (float)(D(e->thm[chn].l[sb]) + thmm);
            enn = w_next * D(eb[b]);
            thmm = w_next * D(thr[b]);
        }
    }
    for (; sb < SBMAX_l; ++sb) { e->en[chn].l[sb] = 0; e->thm[chn].l[sb] = 0; }
}

static void lo_compute_masking_s(lo_enc* e, float fftenergy_s[3][HBLKSIZE_s], float* eb, float* thr, int chn, int sblock) {
    const lo_cfg* c = &e->c;
    int j, b;
    for (b = j = 0; b < c->npart_s; ++b) {
        double ebb = 0;
        int n = c->numlines_s[b], i;
        for (i = 0; i < n; ++i, ++j) ebb += D(fftenergy_s[sblock][j]);
        eb[b] = (float)ebb;
    }
    for (j = b = 0; b < c->npart_s; b++) {
        int kk = c->s3ind_s[2 * b];
        double ecb = D(c->s3_ss[j++]) * D(eb[kk]);
        ++kk;
        while (kk <= c->s3ind_s[2 * b + 1]) { ecb += D(c->s3_ss[j]) * D(eb[kk]); ++j; ++kk; }
        {
            double x = 2 * D(e->nb_s1[chn][b]);               /* rpelev_s */
            thr[b] = (float)(ecb < x ? ecb : x);
        }
        if (e->blocktype_old[chn & 1] == SHORT_TYPE) {
            double x = 16 * D(e->nb_s2[chn][b]);              /* rpelev2_s */
            double y = thr[b];
            thr[b] = (float)(x < y ? x : y);
        }
        e->nb_s2[chn][b] = e->nb_s1[chn][b];
        e->nb_s1[chn][b] = (float)ecb;
    }
    for (; b <= CBANDS; ++b) { eb[b] = 0; thr[b] = 0; }
}

/* One psy call: analyses the granule that starts 576 samples after the one being coded.
 * buf[ch] = mfbuf[ch] + (576 + 576*gr - 272).  Writes masking for (gr,ch) and block types. */

/* perceptual entropy (PsyModel.js:845-905).  Dead for the output of the L/R path (on_pe's additions collapse with the reservoir
 * disabled); live in joint stereo, where the M/S decision compares the sums (Encoder.js:540-560). */
static double lo_pecalc_s(const lo_ratio* mr, double masking_lower) {
    static const double regcoef_s[] = {11.8, 13.6, 17.2, 32, 46.5, 51.3, 57.5, 67.1, 71.5, 84.6, 97.6, 130};
    double pe_s = 1236.28 / 4;
    int sb, sblock;
    for (sb = 0; sb < SBMAX_s - 1; sb++)
        for (sblock = 0; sblock < 3; sblock++) {
            double thm = mr->thm.s[sb][sblock];
            if (thm > 0.0) {
                double x = thm * masking_lower, en = mr->en.s[sb][sblock];
                if (en > x) {
                    if (en > x * 1e10) pe_s += regcoef_s[sb] * (10.0 * 2.30258509299404568402);
                    else pe_s += regcoef_s[sb] * v8_log10(en / x);
                }
            }
        }
    return pe_s;
}
static double lo_pecalc_l(const lo_ratio* mr, double masking_lower) {
    static const double regcoef_l[] = {6.8, 5.8, 5.8, 6.4, 6.5, 9.9, 12.1, 14.4, 15, 18.9, 21.6, 26.9, 34.2, 40.2, 46.8, 56.5,
                                       60.7, 73.9, 85.7, 93.4, 126.1};
    double pe_l = 1124.23 / 4;
    int sb;
    for (sb = 0; sb < SBMAX_l - 1; sb++) {
        double thm = mr->thm.l[sb];
        if (thm > 0.0) {
            double x = thm * masking_lower, en = mr->en.l[sb];
            if (en > x) {
                if (en > x * 1e10) pe_l += regcoef_l[sb] * (10.0 * 2.30258509299404568402);
                else pe_l += regcoef_l[sb] * v8_log10(en / x);
            }
        }
    }
    return pe_l;
}

/* PsyModel.js:828-842 */
static double lo_ns_interp(double x, double y, double r) {
    if (r >= 1.0) return x;
    if (r <= 0.0) return y;
    if (y > 0.0) return v8_pow(x / y, r) * y;
    return 0.0;
}

#define LO_MAX(a, b) ((a) > (b) ? (a) : (b))      /* Math.max / Math.min on non-NaN operands */
#define LO_MIN(a, b) ((a) < (b) ? (a) : (b))
/* M/S thresholds after Johnston & Ferreira (PsyModel.js:548-582) */
static void lo_msfix1(lo_enc* e) {
    const lo_cfg* c = &e->c;
    int sb, sblock;
    for (sb = 0; sb < SBMAX_l; sb++) {
        double mld, rmid, rside;
        if (D(e->thm[0].l[sb]) > 1.58 * D(e->thm[1].l[sb]) || D(e->thm[1].l[sb]) > 1.58 * D(e->thm[0].l[sb])) continue;
        mld = D(c->mld_l[sb]) * D(e->en[3].l[sb]);
        rmid = LO_MAX(D(e->thm[2].l[sb]), LO_MIN(D(e->thm[3].l[sb]), mld));
        mld = D(c->mld_l[sb]) * D(e->en[2].l[sb]);
        rside = LO_MAX(D(e->thm[3].l[sb]), LO_MIN(D(e->thm[2].l[sb]), mld));
        e->thm[2].l[sb] = (float)rmid;
        e->thm[3].l[sb] = (float)rside;
    }
    for (sb = 0; sb < SBMAX_s; sb++)
        for (sblock = 0; sblock < 3; sblock++) {
            double mld, rmid, rside;
            if (D(e->thm[0].s[sb][sblock]) > 1.58 * D(e->thm[1].s[sb][sblock]) || D(e->thm[1].s[sb][sblock]) > 1.58 * D(e->thm[0].s[sb][sblock])) continue;
            mld = D(c->mld_s[sb]) * D(e->en[3].s[sb][sblock]);
            rmid = LO_MAX(D(e->thm[2].s[sb][sblock]), LO_MIN(D(e->thm[3].s[sb][sblock]), mld));
            mld = D(c->mld_s[sb]) * D(e->en[2].s[sb][sblock]);
            rside = LO_MAX(D(e->thm[3].s[sb][sblock]), LO_MIN(D(e->thm[2].s[sb][sblock]), mld));
            e->thm[2].s[sb][sblock] = (float)rmid;
            e->thm[3].s[sb][sblock] = (float)rside;
        }
}
/* PsyModel.js:591-636 (note: the long-block scaling uses msfix2, the short-block one msfix; both are 2 x the setting) */
static void lo_ns_msfix(lo_enc* e, double msfix, double athadjust) {
    const lo_cfg* c = &e->c;
    double msfix2 = msfix, athlower = v8_pow(10, athadjust);
    int sb, sblock;
    msfix *= 2.0;
    msfix2 *= 2.0;
    for (sb = 0; sb < SBMAX_l; sb++) {
        double ath = D(c->ATH_cb_l[c->bm_l[sb]]) * athlower;
        double thmLR = LO_MIN(LO_MAX(D(e->thm[0].l[sb]), ath), LO_MAX(D(e->thm[1].l[sb]), ath));
        double thmM = LO_MAX(D(e->thm[2].l[sb]), ath), thmS = LO_MAX(D(e->thm[3].l[sb]), ath);
        if (thmLR * msfix < thmM + thmS) {
            double f = thmLR * msfix2 / (thmM + thmS);
            thmM *= f;
            thmS *= f;
        }
        e->thm[2].l[sb] = (float)LO_MIN(thmM, D(e->thm[2].l[sb]));
        e->thm[3].l[sb] = (float)LO_MIN(thmS, D(e->thm[3].l[sb]));
    }
    athlower *= (D(BLKSIZE_s) / BLKSIZE);
    for (sb = 0; sb < SBMAX_s; sb++)
        for (sblock = 0; sblock < 3; sblock++) {
            double ath = D(c->ATH_cb_s[c->bm_s[sb]]) * athlower;
            double thmLR = LO_MIN(LO_MAX(D(e->thm[0].s[sb][sblock]), ath), LO_MAX(D(e->thm[1].s[sb][sblock]), ath));
            double thmM = LO_MAX(D(e->thm[2].s[sb][sblock]), ath), thmS = LO_MAX(D(e->thm[3].s[sb][sblock]), ath);
            if (thmLR * msfix < thmM + thmS) {
                double f = thmLR * msfix / (thmM + thmS);
                thmM *= f;
                thmS *= f;
            }
            e->thm[2].s[sb][sblock] = (float)LO_MIN(D(e->thm[2].s[sb][sblock]), thmM);
            e->thm[3].s[sb][sblock] = (float)LO_MIN(D(e->thm[3].s[sb][sblock]), thmS);
        }
}

/* masking_MS / pe / pe_MS / tot_ener are only filled (and only meaningful) in joint stereo; tot_ener[chn] is the total of
 * the PREVIOUS call (PsyModel.js:1206, one granule of delay like the maskings) */
static void lo_psycho_anal(lo_enc* e, const float* const buf[2], int gr_out, lo_ratio masking[2][2], lo_ratio masking_MS[2][2],
                           double pe[2], double pe_MS[2], float tot_ener[4], int blocktype_d[2]) {
    const lo_cfg* c = &e->c;
    float wsamp_L[2][BLKSIZE];
    float wsamp_S[2][3][BLKSIZE_s];
    float eb_l[CBANDS + 1], eb_s[CBANDS + 1], thr[CBANDS + 2];
    int blocktype[2], uselongblock[2] = {0, 0};
    float ns_hpfsmpl[2][576];
    int32_t mask_idx_l[CBANDS + 2];
    int chn, i, j, b, sb, sblock, k;
    const int numchn = (c->mode == 1) ? 4 : c->channels_out;        /* chn 2, 3 = mid, side (PsyModel.js:1031-1034) */
    /* PsyModel.js:1036-1038: 0 with the reservoir disabled (ResvMax == 0), which makes every NS_INTERP below return its second operand */
    const double pcfact = (e->ResvMax == 0) ? 0 : D(e->ResvSize) / e->ResvMax * 0.5;

    /* fs/4 high-pass for attack detection */
    for (chn = 0; chn < c->channels_out; chn++) {
        const float* fir = buf[chn] + 576 - 350 - 21 + 192;
        for (i = 0; i < 576; i++) {
            double sum1 = D(fir[i + 10]), sum2 = 0.0;
            for (j = 0; j < 9; j += 2) {
                sum1 += c->hpf_fircoef[j] * (D(fir[i + j]) + D(fir[i + 21 - j]));
                sum2 += c->hpf_fircoef[j + 1] * (D(fir[i + j + 1]) + D(fir[i + 21 - j - 1]));
            }
            ns_hpfsmpl[chn][i] = (float)(sum1 + sum2);
        }
        masking[gr_out][chn].en = e->en[chn];
        masking[gr_out][chn].thm = e->thm[chn];
        if (numchn > 2) {
            masking_MS[gr_out][chn].en = e->en[chn + 2];
            masking_MS[gr_out][chn].thm = e->thm[chn + 2];
        }
    }

    for (chn = 0; chn < numchn; chn++) {
        float en_subshort[12];
        double en_short[4] = {0, 0, 0, 0};
        float attack_intensity[12];
        int ns_uselongblock = 1;
        double attackThreshold;
        float max[CBANDS], avg[CBANDS];
        int ns_attacks[4] = {0, 0, 0, 0};
        float fftenergy[HBLKSIZE];
        float fftenergy_s[3][HBLKSIZE_s];

        for (i = 0; i < 3; i++) {
            en_subshort[i] = e->last_en_subshort[chn][i + 6];
            attack_intensity[i] = (float)(D(en_subshort[i]) / D(e->last_en_subshort[chn][i + 4]));
            en_short[0] += D(en_subshort[i]);
        }
        if (chn == 2)                                   /* PsyModel.js:1113-1121: mid / side of the high-passed samples */
            for (i = 0; i < 576; i++) {
                const double l = ns_hpfsmpl[0][i], r = ns_hpfsmpl[1][i];
                ns_hpfsmpl[0][i] = (float)(l + r);
                ns_hpfsmpl[1][i] = (float)(l - r);
            }
        {
            const float* pf = ns_hpfsmpl[chn & 1];
            for (i = 0; i < 9; i++) {
                const float* pfe = pf + 576 / 9;
                double p = 1.;
                for (; pf < pfe; pf++)
                    if (p < fabs(D(*pf))) p = fabs(D(*pf));
                e->last_en_subshort[chn][i] = en_subshort[i + 3] = (float)p;
                /* en_short[1 + i/3] only lands on a real slot when i % 3 == 0 (fractional index otherwise) */
                if (i % 3 == 0) en_short[1 + i / 3] += p;
                if (p > D(en_subshort[i + 3 - 2])) p = p / D(en_subshort[i + 3 - 2]);
                else if (D(en_subshort[i + 3 - 2]) > p * 10.0) p = D(en_subshort[i + 3 - 2]) / (p * 10.0);
                else p = 0.0;
                attack_intensity[i + 3] = (float)p;
            }
        }
        attackThreshold = (chn == 3) ? c->attackthre_s : c->attackthre;
        /* ns_attacks[i/3]: only integer indices (i % 3 == 0) exist, and the stored value is (i%3)+1 == 1 */
        for (i = 0; i < 12; i += 3)
            if (0 == ns_attacks[i / 3] && D(attack_intensity[i]) > attackThreshold) ns_attacks[i / 3] = 1;

        for (i = 1; i < 4; i++) {
            double ratio;
            if (en_short[i - 1] > en_short[i]) ratio = en_short[i - 1] / en_short[i];
            else ratio = en_short[i] / en_short[i - 1];
            if (ratio < 1.7) {
                ns_attacks[i] = 0;
                if (i == 1) ns_attacks[0] = 0;
            }
        }
        if (ns_attacks[0] != 0 && e->lastAttacks[chn] != 0) ns_attacks[0] = 0;
        if (e->lastAttacks[chn] == 3 || (ns_attacks[0] + ns_attacks[1] + ns_attacks[2] + ns_attacks[3]) != 0) {
            ns_uselongblock = 0;
            if (ns_attacks[1] != 0 && ns_attacks[0] != 0) ns_attacks[1] = 0;
            if (ns_attacks[2] != 0 && ns_attacks[1] != 0) ns_attacks[2] = 0;
            if (ns_attacks[3] != 0 && ns_attacks[2] != 0) ns_attacks[3] = 0;
        }
        if (chn < 2) uselongblock[chn] = ns_uselongblock;
        else if (ns_uselongblock == 0) uselongblock[0] = uselongblock[1] = 0;      /* PsyModel.js:1198-1204 */
        tot_ener[chn] = e->tot_ener[chn];

        /* FFTs + energies (compute_ffts, PsyModel.js:251-327) */
        if (chn < 2) {
            lo_fft_long(c, wsamp_L[chn & 1], buf[chn]);
            lo_fft_short(c, wsamp_S[chn & 1], buf[chn]);
        } else if (chn == 2) {                          /* mid / side spectra from the L / R ones */
            for (j = BLKSIZE - 1; j >= 0; --j) {
                const double l = wsamp_L[0][j], r = wsamp_L[1][j];
                wsamp_L[0][j] = (float)((l + r) * SQRT2 * 0.5);
                wsamp_L[1][j] = (float)((l - r) * SQRT2 * 0.5);
            }
            for (b = 2; b >= 0; --b)
                for (j = BLKSIZE_s - 1; j >= 0; --j) {
                    const double l = wsamp_S[0][b][j], r = wsamp_S[1][b][j];
                    wsamp_S[0][b][j] = (float)((l + r) * SQRT2 * 0.5);
                    wsamp_S[1][b][j] = (float)((l - r) * SQRT2 * 0.5);
                }
        }
        {
            const float* wl = wsamp_L[chn & 1];
            fftenergy[0] = wl[0];
            fftenergy[0] = (float)(D(fftenergy[0]) * D(fftenergy[0]));
            for (j = BLKSIZE / 2 - 1; j >= 0; --j) {
                double re = wl[BLKSIZE / 2 - j], im = wl[BLKSIZE / 2 + j];
                fftenergy[BLKSIZE / 2 - j] = (float)((re * re + im * im) * 0.5);
            }
            for (b = 2; b >= 0; --b) {
                const float* ws = wsamp_S[chn & 1][b];
                fftenergy_s[b][0] = ws[0];
                fftenergy_s[b][0] = (float)(D(fftenergy_s[b][0]) * D(fftenergy_s[b][0]));
                for (j = BLKSIZE_s / 2 - 1; j >= 0; --j) {
                    double re = ws[BLKSIZE_s / 2 - j], im = ws[BLKSIZE_s / 2 + j];
                    fftenergy_s[b][BLKSIZE_s / 2 - j] = (float)((re * re + im * im) * 0.5);
                }
            }
            {                                           /* total energy (PsyModel.js:300-307) */
                double tot = 0.0;
                for (j = 11; j < HBLKSIZE; j++) tot += D(fftenergy[j]);
                e->tot_ener[chn] = (float)tot;
            }
            /* loudness approximation for the ATH auto-adjust (athaa_loudapprox == 2); not for mid / side */
            if (chn < 2) {
                double lp = 0.0;
                e->loudness_sq[gr_out][chn] = e->loudness_sq_save[chn];
                for (i = 0; i < BLKSIZE / 2; ++i) lp += D(fftenergy[i]) * D(c->eql_w[i]);
                lp *= c->VO_SCALE;
                e->loudness_sq_save[chn] = (float)lp;
            }
        }

        /* partition energies + tonality index */
        for (b = j = 0; b < c->npart_l; ++b) {
            double ebb = 0, m = 0;
            for (i = 0; i < c->numlines_l[b]; ++i, ++j) {
                double el = fftenergy[j];
                ebb += el;
                if (m < el) m = el;
            }
            eb_l[b] = (float)ebb;
            max[b] = (float)m;
            avg[b] = (float)(ebb * D(c->rnumlines_l[b]));
        }
        {
            const int last_tab_entry = 8;
            const int32_t* nl = c->numlines_l;
            double a, m;
            int kk;
            b = 0;
            a = D(avg[b]) + D(avg[b + 1]);
            if (a > 0.0) {
                m = max[b]; if (m < D(max[b + 1])) m = max[b + 1];
                a = 20.0 * (m * 2.0 - a) / (a * (nl[b] + nl[b + 1] - 1));
                kk = js_toint32(a); if (kk > last_tab_entry) kk = last_tab_entry;
                mask_idx_l[b] = kk;
            } else mask_idx_l[b] = 0;
            for (b = 1; b < c->npart_l - 1; b++) {
                a = D(avg[b - 1]) + D(avg[b]) + D(avg[b + 1]);
                if (a > 0.0) {
                    m = max[b - 1];
                    if (m < D(max[b])) m = max[b];
                    if (m < D(max[b + 1])) m = max[b + 1];
                    a = 20.0 * (m * 3.0 - a) / (a * (nl[b - 1] + nl[b] + nl[b + 1] - 1));
                    kk = js_toint32(a); if (kk > last_tab_entry) kk = last_tab_entry;
                    mask_idx_l[b] = kk;
                } else mask_idx_l[b] = 0;
            }
            a = D(avg[b - 1]) + D(avg[b]);
            if (a > 0.0) {
                m = max[b - 1]; if (m < D(max[b])) m = max[b];
                a = 20.0 * (m * 2.0 - a) / (a * (nl[b - 1] + nl[b] - 1));
                kk = js_toint32(a); if (kk > last_tab_entry) kk = last_tab_entry;
                mask_idx_l[b] = kk;
            } else mask_idx_l[b] = 0;
        }

        /* short-block thresholds */
        for (sblock = 0; sblock < 3; sblock++) {
            lo_compute_masking_s(e, fftenergy_s, eb_s, thr, chn, sblock);
            lo_convert_p2s_s(e, eb_s, thr, chn, sblock);
            for (sb = 0; sb < SBMAX_s; sb++) {
                double thmm = e->thm[chn].s[sb][sblock], enn;
                thmm *= 0.8;                                  /* NS_PREECHO_ATT0 */
                /* short-block pre-echo control (PsyModel.js:1235-1253); ns_attacks only ever holds 0 / 1 (see above), so the
                 * `>= 2` and `== 3` alternatives of the reference never fire; with pcfact == 0 both interpolations return thmm */
                if (ns_attacks[sblock + 1] == 1) {
                    const int idx = (sblock != 0) ? sblock - 1 : 2;
                    const double p = lo_ns_interp(e->thm[chn].s[sb][idx], thmm, 0.6 * pcfact);      /* NS_PREECHO_ATT1 */
                    thmm = LO_MIN(thmm, p);
                }
                if (ns_attacks[sblock] == 1) {
                    const int idx = (sblock != 0) ? sblock - 1 : 2;
                    const double p = lo_ns_interp(e->thm[chn].s[sb][idx], thmm, 0.3 * pcfact);      /* NS_PREECHO_ATT2 */
                    thmm = LO_MIN(thmm, p);
                }
                enn = D(en_subshort[sblock * 3 + 3]) + D(en_subshort[sblock * 3 + 4]) + D(en_subshort[sblock * 3 + 5]);
                if (D(en_subshort[sblock * 3 + 5]) * 6 < enn) {
                    thmm *= 0.5;
                    if (D(en_subshort[sblock * 3 + 4]) * 6 < enn) thmm *= 0.5;
                }
                e->thm[chn].s[sb][sblock] = (float)thmm;
            }
        }
        e->lastAttacks[chn] = ns_attacks[2];

        /* long-block spreading with additive masking */
        k = 0;
        for (b = 0; b < c->npart_l; b++) {
            int kk = c->s3ind[2 * b];
            double eb2 = D(eb_l[kk]) * c->ma_tab[mask_idx_l[kk]];
            double ecb = D(c->s3_ll[k++]) * eb2;
            while (++kk <= c->s3ind[2 * b + 1]) {
                eb2 = D(eb_l[kk]) * c->ma_tab[mask_idx_l[kk]];
                ecb = lo_mask_add(e, ecb, D(c->s3_ll[k++]) * eb2, kk, kk - b);
            }
            ecb *= 0.158489319246111;
            /* long-block pre-echo control (PsyModel.js:1300-1318); pcfact == 0 reduces it to ecb */
            if (e->blocktype_old[chn & 1] == SHORT_TYPE) thr[b] = (float)ecb;
            else {
                const double a = 2 * D(e->nb_1[chn][b]), b2 = 16 * D(e->nb_2[chn][b]);      /* rpelev, rpelev2 */
                const double m = LO_MIN(a, b2);
                thr[b] = (float)lo_ns_interp(LO_MIN(ecb, m), ecb, pcfact);
            }
            e->nb_2[chn][b] = e->nb_1[chn][b];
            e->nb_1[chn][b] = (float)ecb;
        }
        for (; b <= CBANDS; ++b) { eb_l[b] = 0; thr[b] = 0; }
        lo_convert_p2s_l(e, eb_l, thr, chn);
    }

    /* inter-channel masking (stereo only, ratio > 0) */
    if ((c->mode == 0 || c->mode == 1) && c->interChRatio > 0.0 && c->channels_out > 1) {
        double r_ = c->interChRatio;
        for (sb = 0; sb < SBMAX_l; sb++) {
            double l = e->thm[0].l[sb], r = e->thm[1].l[sb];
            e->thm[0].l[sb] = (float)(D(e->thm[0].l[sb]) + r * r_);
            e->thm[1].l[sb] = (float)(D(e->thm[1].l[sb]) + l * r_);
        }
        for (sb = 0; sb < SBMAX_s; sb++)
            for (sblock = 0; sblock < 3; sblock++) {
                double l = e->thm[0].s[sb][sblock], r = e->thm[1].s[sb][sblock];
                e->thm[0].s[sb][sblock] = (float)(D(e->thm[0].s[sb][sblock]) + r * r_);
                e->thm[1].s[sb][sblock] = (float)(D(e->thm[1].s[sb][sblock]) + l * r_);
            }
    }

    if (c->mode == 1) {                                 /* PsyModel.js:1336-1342 */
        lo_msfix1(e);
        if (fabs(c->msfix) > 0.0) lo_ns_msfix(e, c->msfix, c->ATHlower * e->ATH_adjust);
    }

    /* block_type_set */
    if (c->short_blocks_coupled && !(uselongblock[0] != 0 && uselongblock[1] != 0))
        uselongblock[0] = uselongblock[1] = 0;
    for (chn = 0; chn < c->channels_out; chn++) {
        blocktype[chn] = NORM_TYPE;
        if (uselongblock[chn] != 0) {
            if (e->blocktype_old[chn] == SHORT_TYPE) blocktype[chn] = STOP_TYPE;
        } else {
            blocktype[chn] = SHORT_TYPE;
            if (e->blocktype_old[chn] == NORM_TYPE) e->blocktype_old[chn] = START_TYPE;
            if (e->blocktype_old[chn] == STOP_TYPE) e->blocktype_old[chn] = SHORT_TYPE;
        }
        blocktype_d[chn] = e->blocktype_old[chn];
        e->blocktype_old[chn] = blocktype[chn];
    }
    /* perceptual entropy of the granule handed back (PsyModel.js:1352-1380): the maskings are the delayed ones */
    for (chn = 0; chn < numchn; chn++) {
        int type;
        const lo_ratio* mr;
        if (chn > 1) {
            type = (blocktype_d[0] == SHORT_TYPE || blocktype_d[1] == SHORT_TYPE) ? SHORT_TYPE : NORM_TYPE;
            mr = &masking_MS[gr_out][chn - 2];
        } else {
            type = blocktype_d[chn];
            mr = &masking[gr_out][chn];
        }
        {
            const double v = (type == SHORT_TYPE) ? lo_pecalc_s(mr, e->masking_lower) : lo_pecalc_l(mr, e->masking_lower);
            if (chn > 1) pe_MS[chn - 2] = v; else pe[chn] = v;
        }
    }
}

/* Encoder.js:166-243 */
static void lo_adjust_ATH(lo_enc* e) {
    const lo_cfg* c = &e->c;
    double gr2_max, max_pow;
    if (c->ATH_useAdjust == 0) { e->ATH_adjust = 1.0; return; }
    max_pow = e->loudness_sq[0][0];
    gr2_max = e->loudness_sq[1][0];
    if (c->channels_out == 2) {
        max_pow += D(e->loudness_sq[0][1]);
        gr2_max += D(e->loudness_sq[1][1]);
    } else {
        max_pow += max_pow;
        gr2_max += gr2_max;
    }
    if (c->mode_gr == 2) max_pow = max_pow > gr2_max ? max_pow : gr2_max;   /* Math.max */
    max_pow *= 0.5;
    max_pow *= c->ATH_aaSensitivityP;
    if (max_pow > 0.03125) {
        if (e->ATH_adjust >= 1.0) e->ATH_adjust = 1.0;
        else if (e->ATH_adjust < e->ATH_adjustLimit) e->ATH_adjust = e->ATH_adjustLimit;
        e->ATH_adjustLimit = 1.0;
    } else {
        double adj_lim_new = 31.98 * max_pow + 0.000625;
        if (e->ATH_adjust >= adj_lim_new) {
            e->ATH_adjust *= adj_lim_new * 0.075 + 0.925;
            if (e->ATH_adjust < adj_lim_new) e->ATH_adjust = adj_lim_new;
        } else {
            if (e->ATH_adjustLimit >= adj_lim_new) e->ATH_adjust = adj_lim_new;
            else if (e->ATH_adjust < e->ATH_adjustLimit) e->ATH_adjust = e->ATH_adjustLimit;
        }
        e->ATH_adjustLimit = adj_lim_new;
    }
}
