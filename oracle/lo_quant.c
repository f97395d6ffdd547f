/*
 * TEST INFRASTRUCTURE (CPU oracle) -- CBR quantization / noise-shaping loop.
 * Restates reference CBRNewIterationLoop.js (25-90), Quantize.js (92-1078, live parts),
 * QuantizePVT.js (421-484, 541-878), Takehiro.js (102-1030, MPEG-1 parts), Reservoir.js.
 * Control flow, caching side effects (prev_noise) and GrInfo copy semantics are part of
 * the contract.
 */
#include "lo_common.h"

#define IPOW20(c, x) D((c)->ipow20[x])
#define POW20(c, x) D((c)->pow20[(x) + Q_MAX2])

/* ------------------------------------------------------------------ */
/* helpers                                                             */
/* ------------------------------------------------------------------ */

/* QuantizePVT.js:541-561 */
static double lo_athAdjust(double a, double x, double athFloor) {
    const double o = 90.30873362, p = 94.82444863;
    double u = v8_log10(x) * 10.0;
    double v = a * a, w = 0.0;
    u -= athFloor;
    if (v > 1E-20) w = 1. + v8_log10(v) * (10.0 / o);
    if (w < 0) w = 0.;
    u *= w;
    u += athFloor + o - p;
    return v8_pow(10., 0.1 * u);
}

/* Quantize.js:147-202 */
static void lo_psfb21_analogsilence(lo_enc* e, lo_gr* gi) {
    const lo_cfg* c = &e->c;
    float* xr = gi->xr;
    int gsfb, j, block;
    if (gi->block_type != SHORT_TYPE) {
        int stop = 0;
        for (gsfb = PSFB21 - 1; gsfb >= 0 && !stop; gsfb--) {
            int start = c->psfb21[gsfb], end = c->psfb21[gsfb + 1];
            double ath21 = lo_athAdjust(e->ATH_adjust, c->ATH_psfb21[gsfb], c->ATH_floor);
            if (D(c->longfact[21]) > 1e-12) ath21 *= D(c->longfact[21]);
            for (j = end - 1; j >= start; j--) {
                if (fabs(D(xr[j])) < ath21) xr[j] = 0;
                else { stop = 1; break; }
            }
        }
    } else {
        for (block = 0; block < 3; block++) {
            int stop = 0;
            for (gsfb = PSFB12 - 1; gsfb >= 0 && !stop; gsfb--) {
                int start = c->sfb_s[12] * 3 + (c->sfb_s[13] - c->sfb_s[12]) * block + (c->psfb12[gsfb] - c->psfb12[0]);
                int end = start + (c->psfb12[gsfb + 1] - c->psfb12[gsfb]);
                double ath12 = lo_athAdjust(e->ATH_adjust, c->ATH_psfb12[gsfb], c->ATH_floor);
                if (D(c->shortfact[12]) > 1e-12) ath12 *= D(c->shortfact[12]);
                for (j = end - 1; j >= start; j--) {
                    if (fabs(D(xr[j])) < ath12) xr[j] = 0;
                    else { stop = 1; break; }
                }
            }
        }
    }
}

/* Quantize.js:204-306 */
static void lo_init_outer_loop(lo_enc* e, lo_gr* gi) {
    const lo_cfg* c = &e->c;
    int sfb, window, l;
    gi->part2_3_length = 0; gi->big_values = 0; gi->count1 = 0; gi->global_gain = 210;
    gi->scalefac_compress = 0;
    gi->table_select[0] = gi->table_select[1] = gi->table_select[2] = 0;
    gi->subblock_gain[0] = gi->subblock_gain[1] = gi->subblock_gain[2] = gi->subblock_gain[3] = 0;
    gi->region0_count = 0; gi->region1_count = 0; gi->preflag = 0; gi->scalefac_scale = 0;
    gi->count1table_select = 0; gi->part2_length = 0;
    gi->sfb_part_tab = 0; gi->sfb_part_row = 0; gi->slen[0] = gi->slen[1] = gi->slen[2] = gi->slen[3] = 0;   /* Quantize.js:292-296 */
    gi->sfb_lmax = SBPSY_l; gi->sfb_smin = SBPSY_s;
    gi->psy_lmax = c->sfb21_extra ? SBMAX_l : SBPSY_l;
    gi->psymax = gi->psy_lmax;
    gi->sfbmax = gi->sfb_lmax;
    gi->sfbdivide = 11;
    for (sfb = 0; sfb < SBMAX_l; sfb++) {
        gi->width[sfb] = c->sfb_l[sfb + 1] - c->sfb_l[sfb];
        gi->window[sfb] = 3;
    }
    if (gi->block_type == SHORT_TYPE) {
        float ixwork[576];
        int ix, j;
        gi->sfb_smin = 0; gi->sfb_lmax = 0;
        gi->psymax = gi->sfb_lmax + 3 * ((c->sfb21_extra ? SBMAX_s : SBPSY_s) - gi->sfb_smin);
        gi->sfbmax = gi->sfb_lmax + 3 * (SBPSY_s - gi->sfb_smin);
        gi->sfbdivide = gi->sfbmax - 18;
        gi->psy_lmax = gi->sfb_lmax;
        ix = c->sfb_l[gi->sfb_lmax];
        memcpy(ixwork, gi->xr, sizeof ixwork);
        for (sfb = gi->sfb_smin; sfb < SBMAX_s; sfb++) {
            int start = c->sfb_s[sfb], end = c->sfb_s[sfb + 1];
            for (window = 0; window < 3; window++)
                for (l = start; l < end; l++) gi->xr[ix++] = ixwork[3 * l + window];
        }
        j = gi->sfb_lmax;
        for (sfb = gi->sfb_smin; sfb < SBMAX_s; sfb++) {
            gi->width[j] = gi->width[j + 1] = gi->width[j + 2] = c->sfb_s[sfb + 1] - c->sfb_s[sfb];
            gi->window[j] = 0; gi->window[j + 1] = 1; gi->window[j + 2] = 2;
            j += 3;
        }
    }
    gi->count1bits = 0;
    gi->max_nonzero_coeff = 575;
    memset(gi->scalefac, 0, sizeof gi->scalefac);
    lo_psfb21_analogsilence(e, gi);
}

/* Quantize.js:92-138 ; returns 1 if there is energy to code */
static int lo_init_xrpow(lo_gr* gi, float* xrpow) {
    double sum = 0;
    int i, upper = gi->max_nonzero_coeff;
    gi->xrpow_max = 0;
    for (i = upper; i < 576; i++) xrpow[i] = 0;
    for (i = 0; i <= upper; ++i) {
        double tmp = fabs(D(gi->xr[i]));
        sum += tmp;
        xrpow[i] = (float)sqrt(tmp * sqrt(tmp));
        if (D(xrpow[i]) > gi->xrpow_max) gi->xrpow_max = xrpow[i];
    }
    if (sum > 1E-20) return 1;
    memset(gi->l3_enc, 0, sizeof gi->l3_enc);
    return 0;
}

/* QuantizePVT.js:569-719 */
static void lo_calc_xmin(lo_enc* e, const lo_ratio* ratio, lo_gr* gi, float* pxmin) {
    const lo_cfg* c = &e->c;
    const float* xr = gi->xr;
    const double masking_lower = e->masking_lower;
    int gsfb, j = 0, sfb, b, p = 0;
    for (gsfb = 0; gsfb < gi->psy_lmax; gsfb++) {
        double en0, xmin, rh1, rh2;
        int width, l;
        xmin = e->ATH_adjust * D(c->ATH_l[gsfb]);
        width = gi->width[gsfb];
        rh1 = xmin / width;
        rh2 = 2.2204460492503131e-016;
        l = width >> 1;
        en0 = 0.0;
        do {
            double xa, xb;
            xa = D(xr[j]) * D(xr[j]); en0 += xa; rh2 += (xa < rh1) ? xa : rh1; j++;
            xb = D(xr[j]) * D(xr[j]); en0 += xb; rh2 += (xb < rh1) ? xb : rh1; j++;
        } while (--l > 0);
        {
            double en = ratio->en.l[gsfb];
            if (en > 0.0) {
                double x = en0 * D(ratio->thm.l[gsfb]) * masking_lower / en;
                if (xmin < x) xmin = x;
            }
        }
        pxmin[p++] = (float)(xmin * D(c->longfact[gsfb]));
    }
    {
        int max_nonzero = 575;
        if (gi->block_type != SHORT_TYPE) {
            int k = 576;
            while (k-- != 0 && D(xr[k]) == 0) max_nonzero = k;   /* BitStream.EQ(x, 0) for x == 0 */
        }
        gi->max_nonzero_coeff = max_nonzero;
    }
    for (sfb = gi->sfb_smin; gsfb < gi->psymax; sfb++, gsfb += 3) {
        int width = gi->width[gsfb];
        double tmpATH = e->ATH_adjust * D(c->ATH_s[sfb]);
        for (b = 0; b < 3; b++) {
            double en0 = 0.0, xmin, rh1, rh2;
            int l = width >> 1;
            rh1 = tmpATH / width;
            rh2 = 2.2204460492503131e-016;
            do {
                double xa, xb;
                xa = D(xr[j]) * D(xr[j]); en0 += xa; rh2 += (xa < rh1) ? xa : rh1; j++;
                xb = D(xr[j]) * D(xr[j]); en0 += xb; rh2 += (xb < rh1) ? xb : rh1; j++;
            } while (--l > 0);
            xmin = tmpATH;
            {
                double en = ratio->en.s[sfb][b];
                if (en > 0.0) {
                    double x = en0 * D(ratio->thm.s[sfb][b]) * masking_lower / en;
                    if (xmin < x) xmin = x;
                }
            }
            pxmin[p++] = (float)(xmin * D(c->shortfact[sfb]));
        }
        if (c->useTemporal) {
            if (D(pxmin[p - 3]) > D(pxmin[p - 3 + 1]))
                pxmin[p - 3 + 1] = (float)(D(pxmin[p - 3 + 1]) + (D(pxmin[p - 3]) - D(pxmin[p - 3 + 1])) * c->decay);
            if (D(pxmin[p - 3 + 1]) > D(pxmin[p - 3 + 2]))
                pxmin[p - 3 + 2] = (float)(D(pxmin[p - 3 + 2]) + (D(pxmin[p - 3 + 1]) - D(pxmin[p - 3 + 2])) * c->decay);
        }
    }
}

/* ------------------------------------------------------------------ */
/* quantization + Huffman bit counting (Takehiro.js)                    */
/* ------------------------------------------------------------------ */

static void lo_quantize_lines_01(int l, double istep, const float* xr, int32_t* ix) {
    const double compareval0 = (1.0 - 0.4054) / istep;
    l = l >> 1;
    while ((l--) != 0) {
        *ix++ = (compareval0 > D(*xr++)) ? 0 : 1;
        *ix++ = (compareval0 > D(*xr++)) ? 0 : 1;
    }
}

static void lo_quantize_lines(const lo_cfg* c, int l, double istep, const float* xr, int32_t* ix) {
    /* processes 2 * (l >> 1) lines; per line: x = xr*istep; ix = (int)(x + adj43[(int)x]) */
    int n = (l >> 1) * 2, i;
    for (i = 0; i < n; i++) {
        double x = D(xr[i]) * istep;
        int rx = js_toint32(x);
        x += D(c->adj43[rx]);
        ix[i] = js_toint32(x);
    }
}

/* Takehiro.js:171-314 */
static void lo_quantize_xrpow(const lo_cfg* c, const float* xp, int32_t* pi, double istep, const lo_gr* gi,
                              const lo_noise_data* prev) {
    int sfb, sfbmax, j = 0, accumulate = 0, accumulate01 = 0;
    int pos = 0, acc_pos = 0;
    const int prev_data_use = (prev != NULL && (gi->global_gain == prev->global_gain));
    sfbmax = (gi->block_type == SHORT_TYPE) ? 38 : 21;
    for (sfb = 0; sfb <= sfbmax; sfb++) {
        int step = -1;
        if (prev_data_use || gi->block_type == NORM_TYPE) {
            step = gi->global_gain
                - ((gi->scalefac[sfb] + (gi->preflag != 0 ? c->pretab[sfb] : 0)) << (gi->scalefac_scale + 1))
                - gi->subblock_gain[gi->window[sfb]] * 8;
        }
        if (prev_data_use && (prev->step[sfb] == step)) {
            if (accumulate != 0) { lo_quantize_lines(c, accumulate, istep, xp + acc_pos, pi + acc_pos); accumulate = 0; }
            if (accumulate01 != 0) { lo_quantize_lines_01(accumulate01, istep, xp + acc_pos, pi + acc_pos); accumulate01 = 0; }
        } else {
            int l = gi->width[sfb];
            if ((j + gi->width[sfb]) > gi->max_nonzero_coeff) {
                int usefullsize = gi->max_nonzero_coeff - j + 1, t;
                for (t = gi->max_nonzero_coeff; t < 576; t++) pi[t] = 0;
                l = usefullsize;
                if (l < 0) l = 0;
                sfb = sfbmax + 1;
            }
            if (0 == accumulate && 0 == accumulate01) acc_pos = pos;
            if (prev != NULL && prev->sfb_count1 > 0 && sfb >= prev->sfb_count1 && prev->step[sfb] > 0
                && step >= prev->step[sfb]) {
                if (accumulate != 0) {
                    lo_quantize_lines(c, accumulate, istep, xp + acc_pos, pi + acc_pos);
                    accumulate = 0;
                    acc_pos = pos;
                }
                accumulate01 += l;
            } else {
                if (accumulate01 != 0) {
                    lo_quantize_lines_01(accumulate01, istep, xp + acc_pos, pi + acc_pos);
                    accumulate01 = 0;
                    acc_pos = pos;
                }
                accumulate += l;
            }
            if (l <= 0) {
                if (accumulate01 != 0) { lo_quantize_lines_01(accumulate01, istep, xp + acc_pos, pi + acc_pos); accumulate01 = 0; }
                if (accumulate != 0) { lo_quantize_lines(c, accumulate, istep, xp + acc_pos, pi + acc_pos); accumulate = 0; }
                break;
            }
        }
        if (sfb <= sfbmax) {
            pos += gi->width[sfb];
            j += gi->width[sfb];
        }
    }
    if (accumulate != 0) lo_quantize_lines(c, accumulate, istep, xp + acc_pos, pi + acc_pos);
    if (accumulate01 != 0) lo_quantize_lines_01(accumulate01, istep, xp + acc_pos, pi + acc_pos);
}

static int lo_ix_max(const int32_t* ix, int pos, int end) {
    int max1 = 0, max2 = 0;
    do {
        int x1 = ix[pos++], x2 = ix[pos++];
        if (max1 < x1) max1 = x1;
        if (max2 < x2) max2 = x2;
    } while (pos < end);
    return max1 < max2 ? max2 : max1;
}

static const int32_t* lo_hlen(const lo_cfg* c, int t) { return c->ht_hlen + c->ht_off[t]; }

/* Takehiro.js:465-516 with count_bit_* ; *s accumulates bits, returns table */
static int lo_choose_table(const lo_cfg* c, const int32_t* ix, int pos, int end, int* s) {
    int max = lo_ix_max(ix, pos, end);
    if (max == 0) return 0;
    if (max == 1) {
        const int32_t* h1 = lo_hlen(c, 1);
        int sum1 = 0;
        do { int x = ix[pos] * 2 + ix[pos + 1]; pos += 2; sum1 += h1[x]; } while (pos < end);
        *s += sum1;
        return 1;
    }
    if (max <= 3) {
        int t1 = c->huf_tbl_noESC[max - 1];
        int xlen = c->ht_xlen[t1];
        const int32_t* hl = (t1 == 2) ? c->table23 : c->table56;
        int sum = 0, sum2;
        do { int x = ix[pos] * xlen + ix[pos + 1]; pos += 2; sum += hl[x]; } while (pos < end);
        sum2 = sum & 0xffff;
        sum >>= 16;
        if (sum > sum2) { sum = sum2; t1++; }
        *s += sum;
        return t1;
    }
    if (max <= 15) {
        int t1 = c->huf_tbl_noESC[max - 1];
        int xlen = c->ht_xlen[t1];
        const int32_t *h1 = lo_hlen(c, t1), *h2 = lo_hlen(c, t1 + 1), *h3 = lo_hlen(c, t1 + 2);
        int sum1 = 0, sum2 = 0, sum3 = 0, t = t1;
        do {
            int x = ix[pos] * xlen + ix[pos + 1];
            pos += 2;
            sum1 += h1[x]; sum2 += h2[x]; sum3 += h3[x];
        } while (pos < end);
        if (sum1 > sum2) { sum1 = sum2; t++; }
        if (sum1 > sum3) { sum1 = sum3; t = t1 + 2; }
        *s += sum1;
        return t;
    }
    if (max > IXMAX_VAL) { *s = LARGE_BITS; return -1; }
    {
        int choice, choice2, linbits, sum = 0, sum2;
        max -= 15;
        for (choice2 = 24; choice2 < 32; choice2++) if (c->ht_linmax[choice2] >= max) break;
        for (choice = choice2 - 8; choice < 24; choice++) if (c->ht_linmax[choice] >= max) break;
        linbits = c->ht_xlen[choice] * 65536 + c->ht_xlen[choice2];
        do {
            int x = ix[pos++], y = ix[pos++];
            if (x != 0) { if (x > 14) { x = 15; sum += linbits; } x *= 16; }
            if (y != 0) { if (y > 14) { y = 15; sum += linbits; } x += y; }
            sum += c->largetbl[x];
        } while (pos < end);
        sum2 = sum & 0xffff;
        sum = (int)((uint32_t)sum >> 16) | 0;   /* JS >>= on a non-negative int */
        if (sum > sum2) { sum = sum2; choice = choice2; }
        *s += sum;
        return choice;
    }
}

/* Takehiro.js:521-628 (use_best_huffman != 2) */
static int lo_noquant_count_bits(const lo_cfg* c, lo_gr* gi, lo_noise_data* prev) {
    const int32_t* ix = gi->l3_enc;
    int i = ((gi->max_nonzero_coeff + 2) >> 1) << 1, a1, a2, bits;
    if (i > 576) i = 576;
    if (prev != NULL) prev->sfb_count1 = 0;
    for (; i > 1; i -= 2)
        if ((ix[i - 1] | ix[i - 2]) != 0) break;
    gi->count1 = i;
    a1 = a2 = 0;
    for (; i > 3; i -= 4) {
        int p;
        if (((ix[i - 1] | ix[i - 2] | ix[i - 3] | ix[i - 4]) & 0x7fffffff) > 1) break;
        p = ((ix[i - 4] * 2 + ix[i - 3]) * 2 + ix[i - 2]) * 2 + ix[i - 1];
        a1 += c->t32l[p];
        a2 += c->t33l[p];
    }
    bits = a1;
    gi->count1table_select = 0;
    if (a1 > a2) { bits = a2; gi->count1table_select = 1; }
    gi->count1bits = bits;
    gi->big_values = i;
    if (i == 0) return bits;

    if (gi->block_type == SHORT_TYPE) {
        a1 = 3 * c->sfb_s[3];
        if (a1 > gi->big_values) a1 = gi->big_values;
        a2 = gi->big_values;
    } else if (gi->block_type == NORM_TYPE) {
        a1 = gi->region0_count = c->bv_scf[i - 2];
        a2 = gi->region1_count = c->bv_scf[i - 1];
        a2 = c->sfb_l[a1 + a2 + 2];
        a1 = c->sfb_l[a1 + 1];
        if (a2 < i) gi->table_select[2] = lo_choose_table(c, ix, a2, i, &bits);
    } else {
        gi->region0_count = 7;
        gi->region1_count = SBMAX_l - 1 - 7 - 1;
        a1 = c->sfb_l[7 + 1];
        a2 = i;
        if (a1 > a2) a1 = a2;
    }
    if (a1 > i) a1 = i;
    if (a2 > i) a2 = i;
    if (0 < a1) gi->table_select[0] = lo_choose_table(c, ix, 0, a1, &bits);
    if (a1 < a2) gi->table_select[1] = lo_choose_table(c, ix, a1, a2, &bits);
    if (prev != NULL && gi->block_type == NORM_TYPE) {
        int sfb = 0;
        while (c->sfb_l[sfb] < gi->big_values) sfb++;
        prev->sfb_count1 = sfb;
    }
    return bits;
}

/* Takehiro.js:630-660 */
static int lo_count_bits(const lo_cfg* c, const float* xrpow, lo_gr* gi, lo_noise_data* prev) {
    double w = (IXMAX_VAL) / IPOW20(c, gi->global_gain);
    if (gi->xrpow_max > w) return LARGE_BITS;
    lo_quantize_xrpow(c, xrpow, gi->l3_enc, IPOW20(c, gi->global_gain), gi, prev);
    return lo_noquant_count_bits(c, gi, prev);
}

/* ------------------------------------------------------------------ */
/* noise                                                               */
/* ------------------------------------------------------------------ */

/* QuantizePVT.js:725-767 */
static double lo_calc_noise_core(const lo_cfg* c, const lo_gr* gi, int* startline, int l, double step) {
    double noise = 0;
    int j = *startline;
    const int32_t* ix = gi->l3_enc;
    if (j > gi->count1) {
        while ((l--) != 0) {
            double t;
            t = gi->xr[j]; j++; noise += t * t;
            t = gi->xr[j]; j++; noise += t * t;
        }
    } else if (j > gi->big_values) {
        float ix01[2];
        ix01[0] = 0; ix01[1] = (float)step;
        while ((l--) != 0) {
            double t;
            t = fabs(D(gi->xr[j])) - D(ix01[ix[j]]); j++; noise += t * t;
            t = fabs(D(gi->xr[j])) - D(ix01[ix[j]]); j++; noise += t * t;
        }
    } else {
        while ((l--) != 0) {
            double t;
            t = fabs(D(gi->xr[j])) - D(c->pow43[ix[j]]) * step; j++; noise += t * t;
            t = fabs(D(gi->xr[j])) - D(c->pow43[ix[j]]) * step; j++; noise += t * t;
        }
    }
    *startline = j;
    return noise;
}

/* QuantizePVT.js:784-878 */
static int lo_calc_noise(const lo_cfg* c, const lo_gr* gi, const float* l3_xmin, float* distort,
                         lo_noise_res* res, lo_noise_data* prev) {
    int sfb, l, over = 0, j = 0;
    double over_noise_db = 0, tot_noise_db = 0, max_noise = -20.0;
    res->over_SSD = 0;
    for (sfb = 0; sfb < gi->psymax; sfb++) {
        int s = gi->global_gain
            - ((gi->scalefac[sfb] + (gi->preflag != 0 ? c->pretab[sfb] : 0)) << (gi->scalefac_scale + 1))
            - gi->subblock_gain[gi->window[sfb]] * 8;
        double noise = 0.0;
        if (prev != NULL && (prev->step[sfb] == s)) {
            noise = prev->noise[sfb];
            j += gi->width[sfb];
            distort[sfb] = (float)(noise / D(l3_xmin[sfb]));
            noise = prev->noise_log[sfb];
        } else {
            double step = POW20(c, s);
            l = gi->width[sfb] >> 1;
            if ((j + gi->width[sfb]) > gi->max_nonzero_coeff) {
                int usefullsize = gi->max_nonzero_coeff - j + 1;
                if (usefullsize > 0) l = usefullsize >> 1;
                else l = 0;
            }
            noise = lo_calc_noise_core(c, gi, &j, l, step);
            if (prev != NULL) { prev->step[sfb] = s; prev->noise[sfb] = (float)noise; }
            noise = noise / D(l3_xmin[sfb]);          /* value of the assignment expression: unrounded */
            distort[sfb] = (float)noise;
            noise = v8_log10(noise > 1E-20 ? noise : 1E-20);
            if (prev != NULL) prev->noise_log[sfb] = (float)noise;
        }
        if (prev != NULL) prev->global_gain = gi->global_gain;
        tot_noise_db += noise;
        if (noise > 0.0) {
            int tmp = js_toint32(noise * 10 + .5);
            if (tmp < 1) tmp = 1;
            res->over_SSD += tmp * tmp;
            over++;
            over_noise_db += noise;
        }
        max_noise = max_noise > noise ? max_noise : noise;
    }
    res->over_count = over;
    res->tot_noise = tot_noise_db;
    res->over_noise = over_noise_db;
    res->max_noise = max_noise;
    return over;
}

/* ------------------------------------------------------------------ */
/* scalefactor coding                                                  */
/* ------------------------------------------------------------------ */

/* Takehiro.js:980-1030 ; returns 1 when no legal scalefac_compress exists */
static int lo_scale_bitcount(const lo_cfg* c, lo_gr* gi) {
    int k, sfb, max_slen1 = 0, max_slen2 = 0;
    const int32_t* tab;
    int32_t* scalefac = gi->scalefac;
    if (gi->block_type == SHORT_TYPE) {
        tab = c->scale_short;
    } else {
        tab = c->scale_long;
        if (0 == gi->preflag) {
            for (sfb = 11; sfb < SBPSY_l; sfb++)
                if (scalefac[sfb] < c->pretab[sfb]) break;
            if (sfb == SBPSY_l) {
                gi->preflag = 1;
                for (sfb = 11; sfb < SBPSY_l; sfb++) scalefac[sfb] -= c->pretab[sfb];
            }
        }
    }
    for (sfb = 0; sfb < gi->sfbdivide; sfb++) if (max_slen1 < scalefac[sfb]) max_slen1 = scalefac[sfb];
    for (; sfb < gi->sfbmax; sfb++) if (max_slen2 < scalefac[sfb]) max_slen2 = scalefac[sfb];
    gi->part2_length = LARGE_BITS;
    for (k = 0; k < 16; k++) {
        if (max_slen1 < c->slen1_n[k] && max_slen2 < c->slen2_n[k] && gi->part2_length > tab[k]) {
            gi->part2_length = tab[k];
            gi->scalefac_compress = k;
        }
    }
    return gi->part2_length == LARGE_BITS;
}

/* ------------------------------------------------------------------ */
/* outer loop (Quantize.js)                                            */
/* ------------------------------------------------------------------ */

static int lo_bin_search_StepSize(lo_enc* e, lo_gr* gi, int desired_rate, int ch, const float* xrpow) {
    const lo_cfg* c = &e->c;
    int nBits, CurrentStep = e->CurrentStep[ch], flagGoneOver = 0;
    const int start = e->OldValue[ch];
    int Direction = 0;  /* 0 none, 1 up, 2 down */
    gi->global_gain = start;
    desired_rate -= gi->part2_length;
    for (;;) {
        int step;
        nBits = lo_count_bits(c, xrpow, gi, NULL);
        if (CurrentStep == 1 || nBits == desired_rate) break;
        if (nBits > desired_rate) {
            if (Direction == 2) flagGoneOver = 1;
            if (flagGoneOver) CurrentStep /= 2;
            Direction = 1;
            step = CurrentStep;
        } else {
            if (Direction == 1) flagGoneOver = 1;
            if (flagGoneOver) CurrentStep /= 2;
            Direction = 2;
            step = -CurrentStep;
        }
        gi->global_gain += step;
        if (gi->global_gain < 0) { gi->global_gain = 0; flagGoneOver = 1; }
        if (gi->global_gain > 255) { gi->global_gain = 255; flagGoneOver = 1; }
    }
    while (nBits > desired_rate && gi->global_gain < 255) {
        gi->global_gain++;
        nBits = lo_count_bits(c, xrpow, gi, NULL);
    }
    e->CurrentStep[ch] = (start - gi->global_gain >= 4) ? 4 : 2;
    e->OldValue[ch] = gi->global_gain;
    gi->part2_3_length = nBits;
    return nBits;
}

static int lo_loop_break(const lo_gr* gi) {
    int sfb;
    for (sfb = 0; sfb < gi->sfbmax; sfb++)
        if (gi->scalefac[sfb] + gi->subblock_gain[gi->window[sfb]] == 0) return 0;
    return 1;
}

/* Quantize.js:481-568, case 9 */
static int lo_quant_compare(const lo_noise_res* best, const lo_noise_res* calc) {
    int better;
    if (best->over_count > 0) {
        better = calc->over_SSD <= best->over_SSD;
        if (calc->over_SSD == best->over_SSD) better = calc->bits < best->bits;
    } else {
        better = ((calc->max_noise < 0) && ((calc->max_noise * 10 + calc->bits) <= (best->max_noise * 10 + best->bits)));
    }
    if (best->over_count == 0) better = better && calc->bits < best->bits;
    return better;
}

/* Quantize.js:597-669, noise_shaping_amp modes 0..2 (3 not reachable via quality 3) */
static void lo_amp_scalefac_bands(const lo_cfg* c, lo_gr* gi, const float* distort, float* xrpow) {
    double ifqstep34, trigger = 0;
    int sfb, j = 0, l;
    ifqstep34 = (gi->scalefac_scale == 0) ? 1.29683955465100964055 : 1.68179283050742922612;
    for (sfb = 0; sfb < gi->sfbmax; sfb++) if (trigger < D(distort[sfb])) trigger = distort[sfb];
    switch (c->noise_shaping_amp) {
        case 2: break;
        case 1:
            if (trigger > 1.0) trigger = sqrt(trigger);   /* Math.pow(x, .5) == sqrt(x) in V8 */
            else trigger *= .95;
            break;
        default:
            if (trigger > 1.0) trigger = 1.0;
            else trigger *= .95;
            break;
    }
    for (sfb = 0; sfb < gi->sfbmax; sfb++) {
        int width = gi->width[sfb];
        j += width;
        if (D(distort[sfb]) < trigger) continue;
        gi->scalefac[sfb]++;
        for (l = -width; l < 0; l++) {
            xrpow[j + l] = (float)(D(xrpow[j + l]) * ifqstep34);
            if (D(xrpow[j + l]) > gi->xrpow_max) gi->xrpow_max = xrpow[j + l];
        }
        if (c->noise_shaping_amp == 2) return;
    }
}

static void lo_inc_scalefac_scale(const lo_cfg* c, lo_gr* gi, float* xrpow) {
    const double ifqstep34 = 1.29683955465100964055;
    int j = 0, sfb, l;
    for (sfb = 0; sfb < gi->sfbmax; sfb++) {
        int width = gi->width[sfb];
        int s = gi->scalefac[sfb];
        if (gi->preflag != 0) s += c->pretab[sfb];
        j += width;
        if ((s & 1) != 0) {
            s++;
            for (l = -width; l < 0; l++) {
                xrpow[j + l] = (float)(D(xrpow[j + l]) * ifqstep34);
                if (D(xrpow[j + l]) > gi->xrpow_max) gi->xrpow_max = xrpow[j + l];
            }
        }
        gi->scalefac[sfb] = s >> 1;
    }
    gi->preflag = 0;
    gi->scalefac_scale = 1;
}

/* Quantize.js:705-778 ; returns 1 on failure */
static int lo_inc_subblock_gain(const lo_cfg* c, lo_gr* gi, float* xrpow) {
    int sfb, window, l;
    int32_t* scalefac = gi->scalefac;
    for (sfb = 0; sfb < gi->sfb_lmax; sfb++) if (scalefac[sfb] >= 16) return 1;
    for (window = 0; window < 3; window++) {
        int s1 = 0, s2 = 0, j;
        for (sfb = gi->sfb_lmax + window; sfb < gi->sfbdivide; sfb += 3) if (s1 < scalefac[sfb]) s1 = scalefac[sfb];
        for (; sfb < gi->sfbmax; sfb += 3) if (s2 < scalefac[sfb]) s2 = scalefac[sfb];
        if (s1 < 16 && s2 < 8) continue;
        if (gi->subblock_gain[window] >= 7) return 1;
        gi->subblock_gain[window]++;
        j = c->sfb_l[gi->sfb_lmax];
        for (sfb = gi->sfb_lmax + window; sfb < gi->sfbmax; sfb += 3) {
            double amp;
            int width = gi->width[sfb];
            int s = scalefac[sfb];
            s = s - (4 >> gi->scalefac_scale);
            if (s >= 0) { scalefac[sfb] = s; j += width * 3; continue; }
            scalefac[sfb] = 0;
            amp = IPOW20(c, 210 + (s << (gi->scalefac_scale + 1)));
            j += width * (window + 1);
            for (l = -width; l < 0; l++) {
                xrpow[j + l] = (float)(D(xrpow[j + l]) * amp);
                if (D(xrpow[j + l]) > gi->xrpow_max) gi->xrpow_max = xrpow[j + l];
            }
            j += width * (3 - window - 1);
        }
        {
            double amp = IPOW20(c, 202);
            j += gi->width[sfb] * (window + 1);
            for (l = -gi->width[sfb]; l < 0; l++) {
                xrpow[j + l] = (float)(D(xrpow[j + l]) * amp);
                if (D(xrpow[j + l]) > gi->xrpow_max) gi->xrpow_max = xrpow[j + l];
            }
        }
    }
    return 0;
}

/* MPEG-2 LSF: scalefactor partition tables of ISO 13818-3 2.4.3.2 as the reference holds them
 * (QuantizePVT.js:116-122 nr_of_sfb_block, Takehiro.js:1035-1037 max_range_sfac_tab, 1138-1139 log2tab) */
const int lo_nr_of_sfb_block[6][3][4] = {
    {{6, 5, 5, 5}, {9, 9, 9, 9}, {6, 9, 9, 9}}, {{6, 5, 7, 3}, {9, 9, 12, 6}, {6, 9, 12, 6}},
    {{11, 10, 0, 0}, {18, 18, 0, 0}, {15, 18, 0, 0}}, {{7, 7, 7, 0}, {12, 12, 12, 0}, {6, 15, 12, 0}},
    {{6, 6, 6, 3}, {12, 9, 9, 6}, {6, 12, 9, 6}}, {{8, 8, 5, 0}, {15, 12, 9, 0}, {6, 18, 9, 0}}};
static const int lo_max_range_sfac_tab[6][4] = {{15, 15, 7, 7}, {15, 15, 7, 0}, {7, 3, 0, 0}, {15, 31, 31, 0}, {7, 7, 7, 0}, {3, 3, 0, 0}};
static const int lo_log2tab[16] = {0, 1, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4};

/* Takehiro.js:1046-1132 scale_bitcount_lsf ; returns 1 when a scalefactor exceeds its partition's range */
static int lo_scale_bitcount_lsf(const lo_cfg* c, lo_gr* gi) {
    int table_number, row_in_table, partition, nr_sfb, window, over, i, sfb, max_sfac[4] = {0, 0, 0, 0};
    const int32_t* scalefac = gi->scalefac;
    const int* partition_table;
    (void)c;
    table_number = (gi->preflag != 0) ? 2 : 0;
    if (gi->block_type == SHORT_TYPE) {
        row_in_table = 1;
        partition_table = lo_nr_of_sfb_block[table_number][row_in_table];
        for (sfb = 0, partition = 0; partition < 4; partition++) {
            nr_sfb = partition_table[partition] / 3;
            for (i = 0; i < nr_sfb; i++, sfb++)
                for (window = 0; window < 3; window++)
                    if (scalefac[sfb * 3 + window] > max_sfac[partition]) max_sfac[partition] = scalefac[sfb * 3 + window];
        }
    } else {
        row_in_table = 0;
        partition_table = lo_nr_of_sfb_block[table_number][row_in_table];
        for (sfb = 0, partition = 0; partition < 4; partition++) {
            nr_sfb = partition_table[partition];
            for (i = 0; i < nr_sfb; i++, sfb++)
                if (scalefac[sfb] > max_sfac[partition]) max_sfac[partition] = scalefac[sfb];
        }
    }
    for (over = 0, partition = 0; partition < 4; partition++)
        if (max_sfac[partition] > lo_max_range_sfac_tab[table_number][partition]) over = 1;
    if (!over) {
        int slen1, slen2, slen3, slen4;
        gi->sfb_part_tab = table_number; gi->sfb_part_row = row_in_table;
        for (partition = 0; partition < 4; partition++) gi->slen[partition] = lo_log2tab[max_sfac[partition]];
        slen1 = gi->slen[0]; slen2 = gi->slen[1]; slen3 = gi->slen[2]; slen4 = gi->slen[3];
        switch (table_number) {
            case 0: gi->scalefac_compress = (((slen1 * 5) + slen2) << 4) + (slen3 << 2) + slen4; break;
            case 1: gi->scalefac_compress = 400 + (((slen1 * 5) + slen2) << 2) + slen3; break;
            case 2: gi->scalefac_compress = 500 + (slen1 * 3) + slen2; break;
            default: break;
        }
        gi->part2_length = 0;
        for (partition = 0; partition < 4; partition++) gi->part2_length += gi->slen[partition] * partition_table[partition];
    }
    return over;
}

static int lo_scale_bitcount_any(const lo_cfg* c, lo_gr* gi) {       /* Quantize.js:814-817, 840-843; Takehiro.js:936-940 */
    return (c->mode_gr == 2) ? lo_scale_bitcount(c, gi) : lo_scale_bitcount_lsf(c, gi);
}

/* Quantize.js:793-846 ; returns 1 to continue, 0 to stop */
static int lo_balance_noise(const lo_cfg* c, lo_gr* gi, const float* distort, float* xrpow) {
    int status;
    lo_amp_scalefac_bands(c, gi, distort, xrpow);
    status = lo_loop_break(gi);
    if (status) return 0;
    status = lo_scale_bitcount_any(c, gi);
    if (!status) return 1;
    if (c->noise_shaping > 1) {
        if (0 == gi->scalefac_scale) {
            lo_inc_scalefac_scale(c, gi, xrpow);
            status = 0;
        } else if (gi->block_type == SHORT_TYPE && c->subblock_gain > 0) {
            status = (lo_inc_subblock_gain(c, gi, xrpow) || lo_loop_break(gi));
        }
    }
    if (!status) status = lo_scale_bitcount_any(c, gi);
    return !status;
}

/* Quantize.js:871-1052 */
static void lo_outer_loop(lo_enc* e, lo_gr* cod_info, const float* l3_xmin, float* xrpow, int ch, int targ_bits) {
    const lo_cfg* c = &e->c;
    static lo_gr cod_info_w;          /* oracle is single-threaded */
    float distort[SFBMAX];
    lo_noise_res best_noise_info;
    lo_noise_data prev_noise;
    int best_part2_3_length = 9999999;
    int age = 0;

    memset(&best_noise_info, 0, sizeof best_noise_info);
    memset(&prev_noise, 0, sizeof prev_noise);
    memset(distort, 0, sizeof distort);

    lo_bin_search_StepSize(e, cod_info, targ_bits, ch, xrpow);
    if (0 == c->noise_shaping) return;

    lo_calc_noise(c, cod_info, l3_xmin, distort, &best_noise_info, &prev_noise);
    best_noise_info.bits = cod_info->part2_3_length;
    cod_info_w = *cod_info;

    do {
        lo_noise_res noise_info;
        const int search_limit = 3;
        int maxggain = 255, huff_bits, better;
        memset(&noise_info, 0, sizeof noise_info);

        if (!lo_balance_noise(c, &cod_info_w, distort, xrpow)) break;
        if (cod_info_w.scalefac_scale != 0) maxggain = 254;
        huff_bits = targ_bits - cod_info_w.part2_length;
        if (huff_bits <= 0) break;

        while ((cod_info_w.part2_3_length = lo_count_bits(c, xrpow, &cod_info_w, &prev_noise)) > huff_bits
               && cod_info_w.global_gain <= maxggain)
            cod_info_w.global_gain++;
        if (cod_info_w.global_gain > maxggain) break;

        if (best_noise_info.over_count == 0) {
            while ((cod_info_w.part2_3_length = lo_count_bits(c, xrpow, &cod_info_w, &prev_noise)) > best_part2_3_length
                   && cod_info_w.global_gain <= maxggain)
                cod_info_w.global_gain++;
            if (cod_info_w.global_gain > maxggain) break;
        }

        lo_calc_noise(c, &cod_info_w, l3_xmin, distort, &noise_info, &prev_noise);
        noise_info.bits = cod_info_w.part2_3_length;

        better = lo_quant_compare(&best_noise_info, &noise_info);
        if (better) {
            best_part2_3_length = cod_info->part2_3_length;   /* read BEFORE the copy */
            best_noise_info = noise_info;
            *cod_info = cod_info_w;
            age = 0;
        } else if (c->full_outer_loop == 0) {
            if (++age > search_limit && best_noise_info.over_count == 0) break;
        }
    } while ((cod_info_w.global_gain + cod_info_w.scalefac_scale) < 255);
}

/* ------------------------------------------------------------------ */
/* final clean-up (Takehiro.js)                                        */
/* ------------------------------------------------------------------ */

static void lo_scfsi_calc(lo_enc* e, int ch) {
    const lo_cfg* c = &e->c;
    lo_gr* gi = &e->tt[1][ch];
    const lo_gr* g0 = &e->tt[0][ch];
    int sfb, i, s1 = 0, c1 = 0, s2 = 0, c2 = 0;
    for (i = 0; i < 4; i++) {
        for (sfb = c->scfsi_band[i]; sfb < c->scfsi_band[i + 1]; sfb++)
            if (g0->scalefac[sfb] != gi->scalefac[sfb] && gi->scalefac[sfb] >= 0) break;
        if (sfb == c->scfsi_band[i + 1]) {
            for (sfb = c->scfsi_band[i]; sfb < c->scfsi_band[i + 1]; sfb++) gi->scalefac[sfb] = -1;
            e->scfsi[ch][i] = 1;
        }
    }
    for (sfb = 0; sfb < 11; sfb++) {
        if (gi->scalefac[sfb] == -1) continue;
        c1++;
        if (s1 < gi->scalefac[sfb]) s1 = gi->scalefac[sfb];
    }
    for (; sfb < SBPSY_l; sfb++) {
        if (gi->scalefac[sfb] == -1) continue;
        c2++;
        if (s2 < gi->scalefac[sfb]) s2 = gi->scalefac[sfb];
    }
    for (i = 0; i < 16; i++) {
        if (s1 < c->slen1_n[i] && s2 < c->slen2_n[i]) {
            int cc = c->slen1_tab[i] * c1 + c->slen2_tab[i] * c2;
            if (gi->part2_length > cc) { gi->part2_length = cc; gi->scalefac_compress = i; }
        }
    }
}

/* Takehiro.js:862-943 */
static void lo_best_scalefac_store(lo_enc* e, int gr, int ch) {
    const lo_cfg* c = &e->c;
    lo_gr* gi = &e->tt[gr][ch];
    int sfb, i, j, l, recalc = 0;
    j = 0;
    for (sfb = 0; sfb < gi->sfbmax; sfb++) {
        int width = gi->width[sfb];
        j += width;
        for (l = -width; l < 0; l++) if (gi->l3_enc[l + j] != 0) break;
        if (l == 0) gi->scalefac[sfb] = recalc = -2;
    }
    if (0 == gi->scalefac_scale && 0 == gi->preflag) {
        int s = 0;
        for (sfb = 0; sfb < gi->sfbmax; sfb++) if (gi->scalefac[sfb] > 0) s |= gi->scalefac[sfb];
        if (0 == (s & 1) && s != 0) {
            for (sfb = 0; sfb < gi->sfbmax; sfb++) if (gi->scalefac[sfb] > 0) gi->scalefac[sfb] >>= 1;
            gi->scalefac_scale = recalc = 1;
        }
    }
    if (0 == gi->preflag && gi->block_type != SHORT_TYPE && c->mode_gr == 2) {
        for (sfb = 11; sfb < SBPSY_l; sfb++)
            if (gi->scalefac[sfb] < c->pretab[sfb] && gi->scalefac[sfb] != -2) break;
        if (sfb == SBPSY_l) {
            for (sfb = 11; sfb < SBPSY_l; sfb++) if (gi->scalefac[sfb] > 0) gi->scalefac[sfb] -= c->pretab[sfb];
            gi->preflag = recalc = 1;
        }
    }
    for (i = 0; i < 4; i++) e->scfsi[ch][i] = 0;
    if (c->mode_gr == 2 && gr == 1 && e->tt[0][ch].block_type != SHORT_TYPE && e->tt[1][ch].block_type != SHORT_TYPE) {
        lo_scfsi_calc(e, ch);
        recalc = 0;
    }
    for (sfb = 0; sfb < gi->sfbmax; sfb++) if (gi->scalefac[sfb] == -2) gi->scalefac[sfb] = 0;
    if (recalc != 0) lo_scale_bitcount_any(c, gi);
}

static void lo_recalc_divide_init(const lo_cfg* c, const lo_gr* gi, const int32_t* ix, int* r01_bits, int* r01_div,
                                  int* r0_tbl, int* r1_tbl) {
    int bigv = gi->big_values, r0, r1;
    for (r0 = 0; r0 <= 7 + 15; r0++) r01_bits[r0] = LARGE_BITS;
    for (r0 = 0; r0 < 16; r0++) {
        int a1 = c->sfb_l[r0 + 1], r0bits = 0, r0t;
        if (a1 >= bigv) break;
        r0t = lo_choose_table(c, ix, 0, a1, &r0bits);
        for (r1 = 0; r1 < 8; r1++) {
            int a2 = c->sfb_l[r0 + r1 + 2], bits, r1t;
            if (a2 >= bigv) break;
            bits = r0bits;
            r1t = lo_choose_table(c, ix, a1, a2, &bits);
            if (r01_bits[r0 + r1] > bits) {
                r01_bits[r0 + r1] = bits;
                r01_div[r0 + r1] = r0;
                r0_tbl[r0 + r1] = r0t;
                r1_tbl[r0 + r1] = r1t;
            }
        }
    }
}

static void lo_recalc_divide_sub(const lo_cfg* c, const lo_gr* cod_info2, lo_gr* gi, const int32_t* ix,
                                 const int* r01_bits, const int* r01_div, const int* r0_tbl, const int* r1_tbl) {
    int bigv = cod_info2->big_values, r2;
    for (r2 = 2; r2 < SBMAX_l + 1; r2++) {
        int a2 = c->sfb_l[r2], bits, r2t;
        if (a2 >= bigv) break;
        bits = r01_bits[r2 - 2] + cod_info2->count1bits;
        if (gi->part2_3_length <= bits) break;
        r2t = lo_choose_table(c, ix, a2, bigv, &bits);
        if (gi->part2_3_length <= bits) continue;
        *gi = *cod_info2;
        gi->part2_3_length = bits;
        gi->region0_count = r01_div[r2 - 2];
        gi->region1_count = r2 - 2 - r01_div[r2 - 2];
        gi->table_select[0] = r0_tbl[r2 - 2];
        gi->table_select[1] = r1_tbl[r2 - 2];
        gi->table_select[2] = r2t;
    }
}

/* Takehiro.js:727-800 (MPEG-1) */
static void lo_best_huffman_divide(const lo_cfg* c, lo_gr* gi) {
    static lo_gr cod_info2;
    int32_t ix[576];
    int r01_bits[7 + 15 + 1], r01_div[7 + 15 + 1], r0_tbl[7 + 15 + 1], r1_tbl[7 + 15 + 1];
    int i, a1, a2;
    if (gi->block_type == SHORT_TYPE && c->mode_gr == 1) return;      /* Takehiro.js:735-737: fails for MPEG-2 short blocks */
    memset(r01_div, 0, sizeof r01_div); memset(r0_tbl, 0, sizeof r0_tbl); memset(r1_tbl, 0, sizeof r1_tbl);
    memcpy(ix, gi->l3_enc, sizeof ix);     /* the reference keeps reading the pre-assign array; contents equal */
    cod_info2 = *gi;
    if (gi->block_type == NORM_TYPE) {
        lo_recalc_divide_init(c, gi, ix, r01_bits, r01_div, r0_tbl, r1_tbl);
        lo_recalc_divide_sub(c, &cod_info2, gi, ix, r01_bits, r01_div, r0_tbl, r1_tbl);
    }
    i = cod_info2.big_values;
    if (i == 0 || (ix[i - 2] | ix[i - 1]) > 1) return;
    i = gi->count1 + 2;
    if (i > 576) return;
    cod_info2 = *gi;
    cod_info2.count1 = i;
    a1 = a2 = 0;
    for (; i > cod_info2.big_values; i -= 4) {
        int p = ((ix[i - 4] * 2 + ix[i - 3]) * 2 + ix[i - 2]) * 2 + ix[i - 1];
        a1 += c->t32l[p];
        a2 += c->t33l[p];
    }
    cod_info2.big_values = i;
    cod_info2.count1table_select = 0;
    if (a1 > a2) { a1 = a2; cod_info2.count1table_select = 1; }
    cod_info2.count1bits = a1;
    if (cod_info2.block_type == NORM_TYPE) {
        lo_recalc_divide_sub(c, &cod_info2, gi, ix, r01_bits, r01_div, r0_tbl, r1_tbl);
    } else {
        cod_info2.part2_3_length = a1;
        a1 = c->sfb_l[7 + 1];
        if (a1 > i) a1 = i;
        if (a1 > 0) cod_info2.table_select[0] = lo_choose_table(c, ix, 0, a1, &cod_info2.part2_3_length);
        if (i > a1) cod_info2.table_select[1] = lo_choose_table(c, ix, a1, i, &cod_info2.part2_3_length);
        if (gi->part2_3_length > cod_info2.part2_3_length) *gi = cod_info2;
    }
}

/* ------------------------------------------------------------------ */
/* per-frame driver (CBRNewIterationLoop.js:25-90 + Reservoir.js + on_pe) */
/* ------------------------------------------------------------------ */

static int lo_frame_bits(const lo_enc* e) {
    const lo_cfg* c = &e->c;
    /* BitStream.js:83-98: bytes = 0 | (version+1)*72000*bit_rate/out_samplerate + padding */
    int bytes = js_toint32(D((c->version + 1) * 72000 * c->brate) / c->out_samplerate + e->padding);
    return 8 * bytes;
}

/* reduce_side (QuantizePVT.js:486-534): M/S granules move bits from the side to the mid channel; targ_bits is an Int32Array there */
static void lo_reduce_side(int32_t targ_bits[2], double ms_ener_ratio, double mean_bits, double max_bits) {
    double fac = .33 * (.5 - ms_ener_ratio) / .5;
    int move_bits;
    if (fac < 0) fac = 0;
    if (fac > .5) fac = .5;
    move_bits = js_toint32(fac * .5 * (targ_bits[0] + targ_bits[1]));
    if (move_bits > MAX_BITS_PER_CHANNEL - targ_bits[0]) move_bits = MAX_BITS_PER_CHANNEL - targ_bits[0];
    if (move_bits < 0) move_bits = 0;
    if (targ_bits[1] >= 125) {
        if (targ_bits[1] - move_bits > 125) {
            if (targ_bits[0] < mean_bits) targ_bits[0] += move_bits;
            targ_bits[1] -= move_bits;
        } else {
            targ_bits[0] += targ_bits[1] - 125;
            targ_bits[1] = 125;
        }
    }
    move_bits = targ_bits[0] + targ_bits[1];
    if (move_bits > max_bits) {
        targ_bits[0] = js_toint32((max_bits * targ_bits[0]) / move_bits);
        targ_bits[1] = js_toint32((max_bits * targ_bits[1]) / move_bits);
    }
}

/* on_pe (QuantizePVT.js:421-484) with ResvMaxBits (Reservoir.js:190-229), the reservoir in use.  The reference computes in JS
 * numbers (doubles) except where it stores into Int32Arrays (targ_bits, add_bits): those stores truncate. */
static double lo_on_pe_resv(lo_enc* e, double pe[2][2], int32_t targ_bits[2], double mean_bits, int gr, int cbr) {
    const lo_cfg* c = &e->c;
    double ResvSize = e->ResvSize, ResvMax = e->ResvMax, tbits, add_b, extra_bits, max_bits, bits;
    int32_t add_bits[2] = {0, 0};
    int ch;
    if (cbr != 0) ResvSize += mean_bits;
    tbits = mean_bits;
    if (ResvSize * 10 > ResvMax * 9) {
        add_b = ResvSize - (ResvMax * 9) / 10;
        tbits += add_b;
    } else {
        add_b = 0;
        if (!c->disable_reservoir) tbits -= .1 * mean_bits;
    }
    extra_bits = (ResvSize < (D(e->ResvMax) * 6) / 10 ? ResvSize : (D(e->ResvMax) * 6) / 10);
    extra_bits -= add_b;
    if (extra_bits < 0) extra_bits = 0;
    max_bits = tbits + extra_bits;
    if (max_bits > MAX_BITS_PER_GRANULE) max_bits = MAX_BITS_PER_GRANULE;
    for (bits = 0, ch = 0; ch < c->channels_out; ++ch) {
        const double t = tbits / c->channels_out;
        targ_bits[ch] = js_toint32(t < MAX_BITS_PER_CHANNEL ? t : MAX_BITS_PER_CHANNEL);
        add_bits[ch] = js_toint32(D(targ_bits[ch]) * pe[gr][ch] / 700.0 - targ_bits[ch]);
        if (add_bits[ch] > mean_bits * 3 / 4) add_bits[ch] = js_toint32(mean_bits * 3 / 4);
        if (add_bits[ch] < 0) add_bits[ch] = 0;
        if (add_bits[ch] + targ_bits[ch] > MAX_BITS_PER_CHANNEL) add_bits[ch] = (MAX_BITS_PER_CHANNEL - targ_bits[ch]) > 0 ? MAX_BITS_PER_CHANNEL - targ_bits[ch] : 0;
        bits += add_bits[ch];
    }
    if (bits > extra_bits)
        for (ch = 0; ch < c->channels_out; ++ch) add_bits[ch] = js_toint32(extra_bits * add_bits[ch] / bits);
    for (ch = 0; ch < c->channels_out; ++ch) {
        targ_bits[ch] += add_bits[ch];
        extra_bits -= add_bits[ch];
    }
    for (bits = 0, ch = 0; ch < c->channels_out; ++ch) bits += targ_bits[ch];
    if (bits > MAX_BITS_PER_GRANULE)
        for (ch = 0; ch < c->channels_out; ++ch) {
            targ_bits[ch] = js_toint32(D(targ_bits[ch]) * MAX_BITS_PER_GRANULE);
            targ_bits[ch] = js_toint32(D(targ_bits[ch]) / bits);
        }
    return max_bits;
}

/* CBRNewIterationLoop.js:25-90 with the reservoir in use: ResvFrameBegin (Reservoir.js:130-180), on_pe per granule, ResvAdjust per
 * granule-channel, ResvFrameEnd (Reservoir.js:243-293).  The frame's bits then go through the continuous stream writer. */
static void lo_iteration_loop_resv(lo_enc* e, double pe[2][2], lo_ratio ratio[2][2], const double ms_ener_ratio[2]) {
    const lo_cfg* c = &e->c;
    float l3_xmin[SFBMAX];
    float xrpow[576];
    int32_t targ_bits[2];
    int gr, ch;
    const int frameLength = lo_frame_bits(e);
    const double mean_bits = D(frameLength - c->sideinfo_len * 8) / c->mode_gr;
    {   /* ResvFrameBegin */
        const int resvLimit = (8 * 256) * c->mode_gr - 8, maxmp3buf = 8 * 1440;       /* brate <= 320, not strict_ISO */
        e->ResvMax = maxmp3buf - frameLength;
        if (e->ResvMax > resvLimit) e->ResvMax = resvLimit;
        if (e->ResvMax < 0 || c->disable_reservoir) e->ResvMax = 0;
        e->resvDrain_pre = 0;
    }
    for (gr = 0; gr < c->mode_gr; gr++) {
        const double max_bits = lo_on_pe_resv(e, pe, targ_bits, mean_bits, gr, gr);
        if (e->mode_ext == 2) {
            int i;
            for (i = 0; i < 576; ++i) {
                const double l = e->tt[gr][0].xr[i], r = e->tt[gr][1].xr[i];
                e->tt[gr][0].xr[i] = (float)((l + r) * (SQRT2 * 0.5));
                e->tt[gr][1].xr[i] = (float)((l - r) * (SQRT2 * 0.5));
            }
            lo_reduce_side(targ_bits, ms_ener_ratio[gr], mean_bits, max_bits);
        }
        for (ch = 0; ch < c->channels_out; ch++) {
            lo_gr* gi = &e->tt[gr][ch];
            e->masking_lower = (gi->block_type != SHORT_TYPE) ? c->masking_lower_long : c->masking_lower_short;
            lo_init_outer_loop(e, gi);
            if (lo_init_xrpow(gi, xrpow)) {
                lo_calc_xmin(e, &ratio[gr][ch], gi, l3_xmin);
                if (e->tap) memcpy(e->tap->l3_xmin[gr][ch], l3_xmin, sizeof l3_xmin);
                lo_outer_loop(e, gi, l3_xmin, xrpow, ch, targ_bits[ch]);
            }
            lo_best_scalefac_store(e, gr, ch);
            if (c->use_best_huffman == 1) lo_best_huffman_divide(c, gi);
            e->ResvSize -= gi->part2_3_length + gi->part2_length;                       /* ResvAdjust */
            if (e->tap) {
                e->tap->global_gain[gr][ch] = gi->global_gain;
                e->tap->part2_3_length[gr][ch] = gi->part2_3_length;
                e->tap->part2_length[gr][ch] = gi->part2_length;
            }
        }
    }
    {   /* ResvFrameEnd */
        int over_bits, stuffingBits = 0;
        double mdb_bytes;
        e->ResvSize += js_toint32(mean_bits * c->mode_gr);
        e->resvDrain_post = 0;
        e->resvDrain_pre = 0;
        if ((over_bits = e->ResvSize % 8) != 0) stuffingBits += over_bits;
        over_bits = (e->ResvSize - stuffingBits) - e->ResvMax;
        if (over_bits > 0) stuffingBits += over_bits;
        mdb_bytes = (e->main_data_begin * 8 < stuffingBits ? e->main_data_begin * 8 : D(stuffingBits)) / 8;
        e->resvDrain_pre += js_toint32(8 * mdb_bytes);
        stuffingBits -= js_toint32(8 * mdb_bytes);
        e->ResvSize -= js_toint32(8 * mdb_bytes);
        e->main_data_begin -= mdb_bytes;
        e->resvDrain_post += stuffingBits;
        e->ResvSize -= stuffingBits;
    }
}

static void lo_iteration_loop(lo_enc* e, lo_ratio ratio[2][2], const double ms_ener_ratio[2]) {
    const lo_cfg* c = &e->c;
    float l3_xmin[SFBMAX];
    float xrpow[576];
    int32_t targ_bits[2];
    int gr, ch;
    const int frameLength = lo_frame_bits(e);
    const int mean_bits = (frameLength - c->sideinfo_len * 8) / c->mode_gr;
    /* ResvFrameBegin with the reservoir disabled: ResvMax = 0 */
    for (gr = 0; gr < c->mode_gr; gr++) {
        /* on_pe + ResvMaxBits(cbr = gr): add_bits/extra_bits collapse to 0 */
        int ResvSize = e->ResvSize, tbits, bits = 0, max_bits;
        if (gr != 0) ResvSize += mean_bits;
        tbits = mean_bits;
        if (ResvSize * 10 > 0) tbits += ResvSize;
        for (ch = 0; ch < c->channels_out; ++ch) {
            double t = D(tbits) / c->channels_out;
            targ_bits[ch] = js_toint32(t < MAX_BITS_PER_CHANNEL ? t : MAX_BITS_PER_CHANNEL);
            bits += targ_bits[ch];
        }
        if (bits > MAX_BITS_PER_GRANULE) {
            for (ch = 0; ch < c->channels_out; ++ch) {
                targ_bits[ch] = js_toint32(D(targ_bits[ch]) * MAX_BITS_PER_GRANULE);
                targ_bits[ch] = js_toint32(D(targ_bits[ch]) / bits);
            }
        }
        if (e->mode_ext == 2) {
            /* ms_convert (Quantize.js:76-83) */
            int i;
            for (i = 0; i < 576; ++i) {
                const double l = e->tt[gr][0].xr[i], r = e->tt[gr][1].xr[i];
                e->tt[gr][0].xr[i] = (float)((l + r) * (SQRT2 * 0.5));
                e->tt[gr][1].xr[i] = (float)((l - r) * (SQRT2 * 0.5));
            }
            max_bits = tbits < MAX_BITS_PER_GRANULE ? tbits : MAX_BITS_PER_GRANULE;      /* on_pe: tbits + extra_bits (= 0), capped */
            lo_reduce_side(targ_bits, ms_ener_ratio[gr], mean_bits, max_bits);
        }
        for (ch = 0; ch < c->channels_out; ch++) {
            lo_gr* gi = &e->tt[gr][ch];
            e->masking_lower = (gi->block_type != SHORT_TYPE) ? c->masking_lower_long : c->masking_lower_short;
            lo_init_outer_loop(e, gi);
            if (lo_init_xrpow(gi, xrpow)) {
                lo_calc_xmin(e, &ratio[gr][ch], gi, l3_xmin);
                if (e->tap) memcpy(e->tap->l3_xmin[gr][ch], l3_xmin, sizeof l3_xmin);
                lo_outer_loop(e, gi, l3_xmin, xrpow, ch, targ_bits[ch]);
            }
            /* iteration_finish_one */
            lo_best_scalefac_store(e, gr, ch);
            if (c->use_best_huffman == 1) lo_best_huffman_divide(c, gi);
            e->ResvSize -= gi->part2_3_length + gi->part2_length;
            if (e->tap) {
                e->tap->global_gain[gr][ch] = gi->global_gain;
                e->tap->part2_3_length[gr][ch] = gi->part2_3_length;
                e->tap->part2_length[gr][ch] = gi->part2_length;
            }
        }
    }
    /* ResvFrameEnd: everything left is drained into this frame's ancillary data */
    e->ResvSize += mean_bits * c->mode_gr;
    e->resvDrain_post = e->ResvSize;
    e->ResvSize = 0;
}
