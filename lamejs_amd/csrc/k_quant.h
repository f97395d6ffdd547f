// CBR quantization / noise-shaping loop as a wave program: one 64-lane wavefront owns one frame
// and runs granule 0 (all channels) then granule 1, exactly as the reference's
// CBRNewIterationLoop.iteration_loop does (CBRNewIterationLoop.js:25-90).  Inside a
// granule-channel the *control flow* of the reference (Quantize.js outer_loop 871-1052,
// bin_search_StepSize 322-381, balance_noise 793-846, ...) is executed uniformly by all lanes,
// while every loop over the 576 spectral lines / the scalefactor bands is spread over the lanes:
//   - quantize x^(3/4) lines (Takehiro.js:102-314): 9 lines per lane, per-sfb mode table in LDS
//   - Huffman bit counting (Takehiro.js:319-628): region maxima and packed length sums by
//     integer wave reductions (exact, order-free)
//   - noise per scalefactor band (QuantizePVT.js:725-878): one lane per band, lines summed in the
//     reference's order (f64 sums are order-sensitive, so they are never tree-reduced)
// State (GrInfo scalars) is wave-uniform and lives in registers; spectra and per-band arrays live in LDS.
#pragma once
#include "lhip_defs.h"
#include "lhip_wave.h"
#include "lhip_math.h"
#include "lhip_layout.h"

namespace lhip {

struct GI {   // wave-uniform scalar part of the reference's GrInfo
    double xrpow_max;
    int part2_3_length, big_values, count1, global_gain, scalefac_compress, block_type;
    int table_select[3], subblock_gain[4];
    int region0_count, region1_count, preflag, scalefac_scale, count1table_select;
    int part2_length, sfb_lmax, sfb_smin, psy_lmax, sfbmax, psymax, sfbdivide;
    int count1bits, max_nonzero_coeff;
};

struct NoiseRes { double max_noise; int over_count, over_SSD, bits; };

// optional phase profiling (build with -DLHIP_PHASE_PROF; never in the product library)
#if defined(LHIP_PHASE_PROF) && !defined(LHIP_HOSTSIM)
#define PH_BEGIN() const unsigned long long ph_t0_ = __builtin_amdgcn_s_memtime()
#define PH_END(L, id) do { if (lane == 0) { (L).prof[id] += __builtin_amdgcn_s_memtime() - ph_t0_; (L).prof[16 + id] += 1; } } while (0)
#else
#define PH_BEGIN() do {} while (0)
#define PH_END(L, id) do {} while (0)
#endif
enum { PH_INIT, PH_XRPOW, PH_XMIN, PH_QUANTIZE, PH_COUNT, PH_NOISE, PH_BALANCE, PH_SFSTORE, PH_HUFFDIV, PH_PUBLISH, PH_COPY, PH_TOTAL, PH_N };

struct QuantLds {
    float xr[576];
    float xrpow[576];
    int32_t ixw[576];            // l3_enc of the working copy (cod_info_w)
    int32_t ixb[576];            // l3_enc of the best/kept copy (cod_info)
    int32_t sfw[SFBMAX + 1], sfb[SFBMAX + 1];     // scalefac working / kept
    int32_t width[SFBMAX + 1], window[SFBMAX + 1], start[SFBMAX + 2];
    float xmin[SFBMAX + 1], distort[SFBMAX + 1];
    int32_t pn_step[SFBMAX + 1];
    float pn_noise[SFBMAX + 1], pn_noise_log[SFBMAX + 1];
    int32_t qmode[SFBMAX + 1], qlen[SFBMAX + 1];
    int32_t nstart[SFBMAX + 1], npairs[SFBMAX + 1], ncached[SFBMAX + 1];
    int32_t sf_gr0[2][SFBMAX + 1];                // final gr0 scalefactors per channel (for scfsi)
    int32_t bstat[16][SBMAX_l + 2];               // per-band Huffman statistics (best_huffman_divide)
    int32_t r01_bits[24], r01_div[24], r0_tbl[24], r1_tbl[24];
    uint8_t line2sfb[576];
    double ath_pseudo[6];
#ifdef LHIP_PHASE_PROF
    unsigned long long prof[32];
#endif
};

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
LHIP_DEV double ipow20(const Tables& T, int x) { return (double)T.ipow20[x]; }
LHIP_DEV double pow20(const Tables& T, int x) { return (double)T.pow20[x + Q_MAX2]; }

// QuantizePVT.js:541-561
LHIP_DEV double athAdjust(const Tables& T, const PowBase& pb10, double a, double x, double athFloor) {
    const double o = 90.30873362, p = 94.82444863;
    double u = v8_log10(x) * 10.0;
    const double v = a * a;
    double w = 0.0;
    u -= athFloor;
    if (v > 1E-20) w = 1. + v8_log10(v) * (10.0 / o);
    if (w < 0) w = 0.;
    u *= w;
    u += athFloor + o - p;
    return v8_pow_from_parts(0.1 * u, pb10.t1, pb10.t2);
}

LHIP_DEV int sbgain(const GI& g, int w) {   // subblock_gain[w] without a dynamically indexed register array
    return w == 0 ? g.subblock_gain[0] : w == 1 ? g.subblock_gain[1] : w == 2 ? g.subblock_gain[2] : g.subblock_gain[3];
}

LHIP_DEV int sf_step(const Tables& T, const GI& g, const int32_t* scalefac, const int32_t* window, int sfb) {
    return g.global_gain - ((scalefac[sfb] + (g.preflag != 0 ? T.pretab[sfb] : 0)) << (g.scalefac_scale + 1))
           - sbgain(g, window[sfb]) * 8;
}

// ---------------------------------------------------------------------------------------------
// init_outer_loop (Quantize.js:204-306) incl. psfb21_analogsilence (147-202)
// xr_g: this granule-channel's MDCT output in HBM (natural order)
// ---------------------------------------------------------------------------------------------
LHIP_DEV void q_init_outer_loop(const Tables& T, const PowBase& pb10, double ath_adjust, GI& g, int block_type,
                                const float* xr_g, int lane, QuantLds& L) {
    g.part2_3_length = 0; g.big_values = 0; g.count1 = 0; g.global_gain = 210; g.scalefac_compress = 0;
    g.block_type = block_type;
    g.table_select[0] = g.table_select[1] = g.table_select[2] = 0;
    g.subblock_gain[0] = g.subblock_gain[1] = g.subblock_gain[2] = g.subblock_gain[3] = 0;
    g.region0_count = 0; g.region1_count = 0; g.preflag = 0; g.scalefac_scale = 0; g.count1table_select = 0;
    g.part2_length = 0; g.sfb_lmax = SBPSY_l; g.sfb_smin = SBPSY_s;
    g.psy_lmax = T.sfb21_extra ? SBMAX_l : SBPSY_l;
    g.psymax = g.psy_lmax; g.sfbmax = g.sfb_lmax; g.sfbdivide = 11;
    g.count1bits = 0; g.max_nonzero_coeff = 575; g.xrpow_max = 0;
    int nsfb;
    if (block_type == SHORT_TYPE) {
        g.sfb_smin = 0; g.sfb_lmax = 0;
        g.psymax = 3 * ((T.sfb21_extra ? SBMAX_s : SBPSY_s));
        g.sfbmax = 3 * SBPSY_s;
        g.sfbdivide = g.sfbmax - 18;
        g.psy_lmax = 0;
        nsfb = 3 * SBMAX_s;
        for (int i = lane; i < nsfb; i += LHIP_NL) {
            const int sfb = i / 3, win = i - 3 * sfb;
            const int w = T.sfb_s[sfb + 1] - T.sfb_s[sfb];
            L.width[i] = w; L.window[i] = win; L.start[i] = 3 * T.sfb_s[sfb] + win * w;
        }
        if (lane == 0) L.start[nsfb] = 576;
        // re-order: within each short sfb the three windows become consecutive runs
        for (int d = lane; d < 576; d += LHIP_NL) {
            int sfb = 0;
            while (3 * T.sfb_s[sfb + 1] <= d) sfb++;
            const int st = T.sfb_s[sfb], w = T.sfb_s[sfb + 1] - st;
            const int r = d - 3 * st, win = r / w, l = st + (r - win * w);
            L.xr[d] = xr_g[3 * l + win];
            L.line2sfb[d] = (uint8_t)(3 * sfb + win);
        }
    } else {
        nsfb = SBMAX_l;
        for (int i = lane; i < SBMAX_l; i += LHIP_NL) {
            L.width[i] = T.sfb_l[i + 1] - T.sfb_l[i]; L.window[i] = 3; L.start[i] = T.sfb_l[i];
        }
        if (lane == 0) L.start[SBMAX_l] = 576;
        for (int d = lane; d < 576; d += LHIP_NL) {
            int sfb = 0;
            while (T.sfb_l[sfb + 1] <= d) sfb++;
            L.xr[d] = xr_g[d];
            L.line2sfb[d] = (uint8_t)sfb;
        }
    }
    for (int i = lane; i <= SFBMAX; i += LHIP_NL) { L.sfw[i] = 0; L.sfb[i] = 0; }
    wave_sync();

    // analog silence in the pseudo bands above sfb21 / sfb12: zero trailing lines below the adjusted ATH
    if (block_type != SHORT_TYPE) {
        for (int gsfb = lane; gsfb < PSFB21; gsfb += LHIP_NL) {
            double a = athAdjust(T, pb10, ath_adjust, T.ATH_psfb21[gsfb], T.ATH_floor);
            if ((double)T.longfact[21] > 1e-12) a *= (double)T.longfact[21];
            L.ath_pseudo[gsfb] = a;
        }
        wave_sync();
        const int lo = T.psfb21[0];
        int top = lo - 1;                       // highest line that is NOT below its threshold
        for (int j = lo + lane; j < 576; j += LHIP_NL) {
            int gsfb = 0;
            while (T.psfb21[gsfb + 1] <= j) gsfb++;
            if (!(d_abs((double)L.xr[j]) < L.ath_pseudo[gsfb])) top = j;   // ascending j per lane
        }
        top = wave_max(top);
        for (int j = lo + lane; j < 576; j += LHIP_NL) if (j > top) L.xr[j] = 0;
    } else {
        for (int gsfb = lane; gsfb < PSFB12; gsfb += LHIP_NL) {
            double a = athAdjust(T, pb10, ath_adjust, T.ATH_psfb12[gsfb], T.ATH_floor);
            if ((double)T.shortfact[12] > 1e-12) a *= (double)T.shortfact[12];
            L.ath_pseudo[gsfb] = a;
        }
        wave_sync();
        const int w12 = T.sfb_s[13] - T.sfb_s[12];
        for (int block = 0; block < 3; block++) {
            const int lo = T.sfb_s[12] * 3 + w12 * block;
            int top = lo - 1;
            for (int j = lo + lane; j < lo + w12; j += LHIP_NL) {
                const int rel = j - lo + T.psfb12[0];
                int gsfb = 0;
                while (T.psfb12[gsfb + 1] <= rel) gsfb++;
                if (!(d_abs((double)L.xr[j]) < L.ath_pseudo[gsfb])) top = j;
            }
            top = wave_max(top);
            for (int j = lo + lane; j < lo + w12; j += LHIP_NL) if (j > top) L.xr[j] = 0;
        }
    }
    wave_sync();
}

// init_xrpow (Quantize.js:92-138); returns 1 if the granule has energy
LHIP_DEV int q_init_xrpow(GI& g, int lane, QuantLds& L) {
    float m = 0.f;
    double sum = 0;
    for (int i = lane; i < 576; i += LHIP_NL) {
        const double tmp = d_abs((double)L.xr[i]);
        sum += tmp;
        const float v = (float)d_sqrt(tmp * d_sqrt(tmp));
        L.xrpow[i] = v;
        if (v > m) m = v;
    }
    m = wave_maxf(m);
    g.xrpow_max = m;
    // `sum > 1e-20` only separates digital silence from signal; the reduction order cannot change the verdict
    // except within 1e-14 (relative) of the threshold itself
    const int has = wave_sumd(sum) > 1E-20;
    wave_sync();
    return has;
}

// calc_xmin (QuantizePVT.js:569-719), CBR flavour
LHIP_DEV void q_calc_xmin(const Tables& T, double ath_adjust, double masking_lower, const float* ratio /*E layout*/,
                          GI& g, int lane, QuantLds& L) {
    if (g.block_type != SHORT_TYPE) {
        for (int gsfb = lane; gsfb < g.psy_lmax; gsfb += LHIP_NL) {
            double xmin = ath_adjust * (double)T.ATH_l[gsfb];
            double en0 = 0.0;
            const int st = L.start[gsfb], w = L.width[gsfb];
            for (int j = st; j < st + w; j++) { const double x = L.xr[j]; en0 += x * x; }
            const double en = ratio[E_EN_L + gsfb];
            if (en > 0.0) {
                const double x = en0 * (double)ratio[E_THM_L + gsfb] * masking_lower / en;
                if (xmin < x) xmin = x;
            }
            L.xmin[gsfb] = (float)(xmin * (double)T.longfact[gsfb]);
        }
        int t = -1;
        for (int k = lane; k < 576; k += LHIP_NL) if (!((double)L.xr[k] == 0)) t = k;
        t = wave_max(t);
        g.max_nonzero_coeff = (t >= 575) ? 575 : t + 1;
    } else {
        for (int sfb = lane; sfb < SBPSY_s; sfb += LHIP_NL) {       // psymax/3 bands, 3 windows each
            const double tmpATH = ath_adjust * (double)T.ATH_s[sfb];
            float px[3];
            for (int b = 0; b < 3; b++) {
                const int gs = 3 * sfb + b, st = L.start[gs], w = L.width[gs];
                double en0 = 0.0, xmin = tmpATH;
                for (int j = st; j < st + w; j++) { const double x = L.xr[j]; en0 += x * x; }
                const double en = ratio[E_EN_S + sfb * 3 + b];
                if (en > 0.0) {
                    const double x = en0 * (double)ratio[E_THM_S + sfb * 3 + b] * masking_lower / en;
                    if (xmin < x) xmin = x;
                }
                px[b] = (float)(xmin * (double)T.shortfact[sfb]);
            }
            if (T.useTemporal) {
                if ((double)px[0] > (double)px[1]) px[1] = (float)((double)px[1] + ((double)px[0] - (double)px[1]) * T.decay);
                if ((double)px[1] > (double)px[2]) px[2] = (float)((double)px[2] + ((double)px[1] - (double)px[2]) * T.decay);
            }
            L.xmin[3 * sfb] = px[0]; L.xmin[3 * sfb + 1] = px[1]; L.xmin[3 * sfb + 2] = px[2];
        }
        g.max_nonzero_coeff = 575;
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------------
// quantize_xrpow (Takehiro.js:171-314) -> ix ; `prev` = use the prev_noise cache (pn_* in LDS)
// ---------------------------------------------------------------------------------------------
LHIP_DEV void q_quantize(const Tables& T, const GI& g, const int32_t* scalefac, int32_t* ix, int use_prev,
                         int pn_gain, int pn_sfb_count1, int lane, QuantLds& L) {
    const double istep = ipow20(T, g.global_gain);
    const int sfbmax = (g.block_type == SHORT_TYPE) ? 38 : 21;
    const int prev_data_use = use_prev && (g.global_gain == pn_gain);
    const int mnz = g.max_nonzero_coeff;
    // per-band decision: 0 keep cached values, 1 full quantization, 2 the 0/1 shortcut
    int cand = 99;
    for (int sfb = lane; sfb <= sfbmax; sfb += LHIP_NL) {
        int step = -1;
        if (prev_data_use || g.block_type == NORM_TYPE) step = sf_step(T, g, scalefac, L.window, sfb);
        int mode;
        if (prev_data_use && L.pn_step[sfb] == step) mode = 0;
        else {
            mode = 1;
            if (use_prev && pn_sfb_count1 > 0 && sfb >= pn_sfb_count1 && L.pn_step[sfb] > 0 && step >= L.pn_step[sfb]) mode = 2;
            if (L.start[sfb] + L.width[sfb] > mnz) cand = sfb;   // first such band ends the walk
        }
        L.qmode[sfb] = mode;
        L.qlen[sfb] = L.width[sfb];
    }
    const int sstar = wave_min(cand);
    wave_sync();
    int fill_from = 576;
    if (sstar <= sfbmax) {
        int l = mnz - L.start[sstar] + 1;
        if (l < 0) l = 0;
        fill_from = mnz;
        if (lane == 0) { L.qmode[sstar] = 1; L.qlen[sstar] = l & ~1; }
    }
    wave_sync();
    const double compareval0 = (1.0 - 0.4054) / istep;
    for (int i = lane; i < 576; i += LHIP_NL) {
        const int sfb = L.line2sfb[i];
        int have = 0, v = 0;
        if (i >= fill_from) { have = 1; v = 0; }
        if (sfb <= sfbmax && sfb <= sstar) {
            const int mode = L.qmode[sfb];
            if (mode != 0 && (i - L.start[sfb]) < L.qlen[sfb]) {
                have = 1;
                const double xv = L.xrpow[i];
                if (mode == 2) v = (compareval0 > xv) ? 0 : 1;
                else {
                    double x = xv * istep;
                    const int rx = js_toint32(x);
                    x += (double)T.adj43[rx];
                    v = js_toint32(x);
                }
            }
        }
        if (have) ix[i] = v;
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------------
// choose_table over pairs [a, b) (Takehiro.js:465-516): cooperative; adds to *bits, returns table
// ---------------------------------------------------------------------------------------------
LHIP_DEV int q_choose_table(const Tables& T, const int32_t* ix, int a, int b, int* bits, int lane) {
    int mx = 0;
    for (int p = a + 2 * lane; p < b; p += 2 * LHIP_NL) {
        const int x1 = ix[p], x2 = ix[p + 1];
        if (mx < x1) mx = x1;
        if (mx < x2) mx = x2;
    }
    mx = wave_max(mx);
    if (mx == 0) return 0;
    if (mx == 1) {
        const int32_t* h1 = T.ht_hlen + T.ht_off[1];
        int s = 0;
        for (int p = a + 2 * lane; p < b; p += 2 * LHIP_NL) s += h1[ix[p] * 2 + ix[p + 1]];
        *bits += wave_sum(s);
        return 1;
    }
    if (mx <= 3) {
        int t1 = T.huf_tbl_noESC[mx - 1];
        const int xlen = T.ht_xlen[t1];
        const int32_t* hl = (t1 == 2) ? T.table23 : T.table56;
        int s = 0;
        for (int p = a + 2 * lane; p < b; p += 2 * LHIP_NL) s += hl[ix[p] * xlen + ix[p + 1]];   // two 16-bit sums, no carry (<= 288*19)
        s = wave_sum(s);
        int s2 = s & 0xffff;
        s >>= 16;
        if (s > s2) { s = s2; t1++; }
        *bits += s;
        return t1;
    }
    if (mx <= 15) {
        const int t1 = T.huf_tbl_noESC[mx - 1];
        const int xlen = T.ht_xlen[t1];
        const int32_t *h1 = T.ht_hlen + T.ht_off[t1], *h2 = T.ht_hlen + T.ht_off[t1 + 1], *h3 = T.ht_hlen + T.ht_off[t1 + 2];
        int s1 = 0, s2 = 0, s3 = 0;
        for (int p = a + 2 * lane; p < b; p += 2 * LHIP_NL) {
            const int x = ix[p] * xlen + ix[p + 1];
            s1 += h1[x]; s2 += h2[x]; s3 += h3[x];
        }
        s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
        int t = t1;
        if (s1 > s2) { s1 = s2; t++; }
        if (s1 > s3) { s1 = s3; t = t1 + 2; }
        *bits += s1;
        return t;
    }
    if (mx > IXMAX_VAL) { *bits = LARGE_BITS; return -1; }
    mx -= 15;
    int choice2, choice;
    for (choice2 = 24; choice2 < 32; choice2++) if (T.ht_linmax[choice2] >= mx) break;
    for (choice = choice2 - 8; choice < 24; choice++) if (T.ht_linmax[choice] >= mx) break;
    const int lb1 = T.ht_xlen[choice], lb2 = T.ht_xlen[choice2];
    int sa = 0, sb2 = 0;        // the two halves of the reference's packed sum, kept apart
    for (int p = a + 2 * lane; p < b; p += 2 * LHIP_NL) {
        int x = ix[p], y = ix[p + 1], n = 0;
        if (x != 0) { if (x > 14) { x = 15; n++; } x *= 16; }
        if (y != 0) { if (y > 14) { y = 15; n++; } x += y; }
        const int lt = T.largetbl[x];
        sa += (lt >> 16) + n * lb1;
        sb2 += (lt & 0xffff) + n * lb2;
    }
    sa = wave_sum(sa); sb2 = wave_sum(sb2);
    // reference: sum = sa*65536 + sb2 packed; sum2 = sum & 0xffff; sum >>= 16  (sb2 < 65536 on this path)
    if (sa > sb2) { sa = sb2; choice = choice2; }
    *bits += sa;
    return choice;
}

// noquant_count_bits (Takehiro.js:521-628); updates g, returns bits.  pn_sfb_count1 as in/out.
LHIP_DEV int q_noquant_count_bits(const Tables& T, GI& g, const int32_t* ix, int use_prev, int* pn_sfb_count1, int lane) {
    int i = ((g.max_nonzero_coeff + 2) >> 1) << 1;
    if (i > 576) i = 576;
    if (use_prev) *pn_sfb_count1 = 0;
    // count1 boundary: highest pair with a non-zero value
    int top = 0;
    for (int p = 2 * lane; p < i; p += 2 * LHIP_NL) if ((ix[p] | ix[p + 1]) != 0) top = p + 2;
    i = wave_max(top);
    g.count1 = i;
    // quadruples (walking down from count1 in steps of 4) with all |v| <= 1
    int a1 = 0, a2 = 0;
    // quad k covers lines [i-4(k+1), i-4k); the scan stops at the first quad holding a value > 1, or at i <= 3
    const int nq = i >> 2;
    int firstbig = nq;                                   // index of the first (topmost) quad that breaks the scan
    for (int k = lane; k < nq; k += LHIP_NL) {
        const int e = i - 4 * k;
        if (((ix[e - 1] | ix[e - 2] | ix[e - 3] | ix[e - 4]) & 0x7fffffff) > 1) { if (k < firstbig) firstbig = k; }
    }
    firstbig = wave_min(firstbig);
    for (int k = lane; k < firstbig; k += LHIP_NL) {
        const int e = i - 4 * k;
        const int p = ((ix[e - 4] * 2 + ix[e - 3]) * 2 + ix[e - 2]) * 2 + ix[e - 1];
        a1 += T.t32l[p];
        a2 += T.t33l[p];
    }
    a1 = wave_sum(a1); a2 = wave_sum(a2);
    i -= 4 * firstbig;
    int bits = a1;
    g.count1table_select = 0;
    if (a1 > a2) { bits = a2; g.count1table_select = 1; }
    g.count1bits = bits;
    g.big_values = i;
    if (i == 0) return bits;
    if (g.block_type == SHORT_TYPE) {
        a1 = 3 * T.sfb_s[3];
        if (a1 > g.big_values) a1 = g.big_values;
        a2 = g.big_values;
    } else if (g.block_type == NORM_TYPE) {
        a1 = g.region0_count = T.bv_scf[i - 2];
        a2 = g.region1_count = T.bv_scf[i - 1];
        a2 = T.sfb_l[a1 + a2 + 2];
        a1 = T.sfb_l[a1 + 1];
        if (a2 < i) g.table_select[2] = q_choose_table(T, ix, a2, i, &bits, lane);
    } else {
        g.region0_count = 7;
        g.region1_count = SBMAX_l - 1 - 7 - 1;
        a1 = T.sfb_l[7 + 1];
        a2 = i;
        if (a1 > a2) a1 = a2;
    }
    if (a1 > i) a1 = i;
    if (a2 > i) a2 = i;
    if (0 < a1) g.table_select[0] = q_choose_table(T, ix, 0, a1, &bits, lane);
    if (a1 < a2) g.table_select[1] = q_choose_table(T, ix, a1, a2, &bits, lane);
    if (use_prev && g.block_type == NORM_TYPE) {
        int sfb = 0;
        while (T.sfb_l[sfb] < g.big_values) sfb++;
        *pn_sfb_count1 = sfb;
    }
    return bits;
}

struct PrevNoise { int gain, sfb_count1; };   // scalar part of CalcNoiseData (arrays are L.pn_*)

// count_bits (Takehiro.js:630-660)
LHIP_DEV int q_count_bits(const Tables& T, GI& g, const int32_t* scalefac, int32_t* ix, PrevNoise* pn, int lane, QuantLds& L) {
    const double w = (double)IXMAX_VAL / ipow20(T, g.global_gain);
    if (g.xrpow_max > w) return LARGE_BITS;
    { PH_BEGIN(); q_quantize(T, g, scalefac, ix, pn != nullptr, pn ? pn->gain : 0, pn ? pn->sfb_count1 : 0, lane, L); PH_END(L, PH_QUANTIZE); }
    int dummy = 0;
    PH_BEGIN();
    const int r = q_noquant_count_bits(T, g, ix, pn != nullptr, pn ? &pn->sfb_count1 : &dummy, lane);
    PH_END(L, PH_COUNT);
    return r;
}

// ---------------------------------------------------------------------------------------------
// calc_noise (QuantizePVT.js:784-878); distort -> L.distort, cache -> L.pn_*
// ---------------------------------------------------------------------------------------------
LHIP_DEV void q_calc_noise_(const Tables& T, const GI& g, const int32_t* scalefac, const int32_t* ix, NoiseRes* res,
                           PrevNoise* pn, int lane, QuantLds& L) {
    // 1) the (sequential) start-line walk: where each band begins and how many pairs it sums
    if (lane == 0) {
        int j = 0;
        for (int sfb = 0; sfb < g.psymax; sfb++) {
            const int s = sf_step(T, g, scalefac, L.window, sfb);
            if (pn != nullptr && L.pn_step[sfb] == s) {
                L.ncached[sfb] = 1;
                j += L.width[sfb];
            } else {
                int l = L.width[sfb] >> 1;
                if ((j + L.width[sfb]) > g.max_nonzero_coeff) {
                    const int usefullsize = g.max_nonzero_coeff - j + 1;
                    l = usefullsize > 0 ? usefullsize >> 1 : 0;
                }
                L.ncached[sfb] = 0; L.nstart[sfb] = j; L.npairs[sfb] = l;
                j += 2 * l;
            }
        }
    }
    wave_sync();
    // 2) per band: noise sum in line order, distortion ratio, log10
    int over = 0, ssd = 0;
    double max_noise = -20.0;
    for (int sfb = lane; sfb < g.psymax; sfb += LHIP_NL) {
        const int s = sf_step(T, g, scalefac, L.window, sfb);
        double noise;
        if (L.ncached[sfb]) {
            noise = L.pn_noise[sfb];
            L.distort[sfb] = (float)(noise / (double)L.xmin[sfb]);
            noise = L.pn_noise_log[sfb];
        } else {
            const double step = pow20(T, s);
            int j = L.nstart[sfb], l = L.npairs[sfb];
            noise = 0;
            if (j > g.count1) {
                for (int t = 0; t < 2 * l; t++, j++) { const double x = L.xr[j]; noise += x * x; }
            } else if (j > g.big_values) {
                const float ix01_1 = (float)step;
                for (int t = 0; t < 2 * l; t++, j++) {
                    const double x = d_abs((double)L.xr[j]) - (ix[j] == 0 ? 0.0 : (double)ix01_1);
                    noise += x * x;
                }
            } else {
                for (int t = 0; t < 2 * l; t++, j++) {
                    const double x = d_abs((double)L.xr[j]) - (double)T.pow43[ix[j]] * step;
                    noise += x * x;
                }
            }
            if (pn != nullptr) { L.pn_step[sfb] = s; L.pn_noise[sfb] = (float)noise; }
            noise = noise / (double)L.xmin[sfb];
            L.distort[sfb] = (float)noise;
            noise = v8_log10(noise > 1E-20 ? noise : 1E-20);
            if (pn != nullptr) L.pn_noise_log[sfb] = (float)noise;
        }
        if (noise > 0.0) {
            int tmp = js_toint32(noise * 10 + .5);
            if (tmp < 1) tmp = 1;
            ssd += tmp * tmp;
            over++;
        }
        if (noise > max_noise) max_noise = noise;
    }
    if (pn != nullptr) pn->gain = g.global_gain;
    res->over_count = wave_sum(over);
    res->over_SSD = wave_sum(ssd);
    res->max_noise = wave_maxd(max_noise);
    wave_sync();
}

LHIP_DEV void q_calc_noise(const Tables& T, const GI& g, const int32_t* scalefac, const int32_t* ix, NoiseRes* res,
                           PrevNoise* pn, int lane, QuantLds& L) {
    PH_BEGIN();
    q_calc_noise_(T, g, scalefac, ix, res, pn, lane, L);
    PH_END(L, PH_NOISE);
}

// ---------------------------------------------------------------------------------------------
// scale_bitcount (Takehiro.js:980-1030), MPEG-1, no mixed blocks.  returns 1 on failure
// ---------------------------------------------------------------------------------------------
LHIP_DEV int q_scale_bitcount(const Tables& T, GI& g, int32_t* scalefac, int lane) {
    const int32_t* tab;
    if (g.block_type == SHORT_TYPE) tab = T.scale_short;
    else {
        tab = T.scale_long;
        if (0 == g.preflag) {
            int bad = 0;
            for (int sfb = 11 + lane; sfb < SBPSY_l; sfb += LHIP_NL) if (scalefac[sfb] < T.pretab[sfb]) bad = 1;
            if (!wave_any(bad)) {
                g.preflag = 1;
                wave_sync();
                for (int sfb = 11 + lane; sfb < SBPSY_l; sfb += LHIP_NL) scalefac[sfb] -= T.pretab[sfb];
                wave_sync();
            }
        }
    }
    int m1 = 0, m2 = 0;
    for (int sfb = lane; sfb < g.sfbmax; sfb += LHIP_NL) {
        const int v = scalefac[sfb];
        if (sfb < g.sfbdivide) { if (m1 < v) m1 = v; } else { if (m2 < v) m2 = v; }
    }
    m1 = wave_max(m1); m2 = wave_max(m2);
    g.part2_length = LARGE_BITS;
    for (int k = 0; k < 16; k++)
        if (m1 < T.slen1_n[k] && m2 < T.slen2_n[k] && g.part2_length > tab[k]) { g.part2_length = tab[k]; g.scalefac_compress = k; }
    return g.part2_length == LARGE_BITS;
}

// ---------------------------------------------------------------------------------------------
// amplification helpers (Quantize.js:453-460, 597-778)
// ---------------------------------------------------------------------------------------------
LHIP_DEV int q_loop_break(const GI& g, const int32_t* scalefac, int lane, QuantLds& L) {
    int z = 0;
    for (int sfb = lane; sfb < g.sfbmax; sfb += LHIP_NL)
        if (scalefac[sfb] + sbgain(g, L.window[sfb]) == 0) z = 1;
    return !wave_any(z);
}

// multiply xrpow of the flagged bands (L.qmode[sfb] = 1) by `amp`, tracking xrpow_max
LHIP_DEV void q_amplify_flagged(GI& g, double amp, int lane, QuantLds& L) {
    float m = 0.f;
    for (int i = lane; i < 576; i += LHIP_NL) {
        const int sfb = L.line2sfb[i];
        if (L.qmode[sfb]) {
            const float v = (float)((double)L.xrpow[i] * amp);
            L.xrpow[i] = v;
            if (v > m) m = v;
        }
    }
    m = wave_maxf(m);
    if ((double)m > g.xrpow_max) g.xrpow_max = m;
    wave_sync();
}

LHIP_DEV void q_amp_scalefac_bands(const Tables& T, GI& g, int32_t* scalefac, int lane, QuantLds& L) {
    const double ifqstep34 = (g.scalefac_scale == 0) ? 1.29683955465100964055 : 1.68179283050742922612;
    float tr = 0.f;
    for (int sfb = lane; sfb < g.sfbmax; sfb += LHIP_NL) if (tr < L.distort[sfb]) tr = L.distort[sfb];
    double trigger = wave_maxf(tr);
    switch (T.noise_shaping_amp) {
        case 2: break;
        case 1:
            if (trigger > 1.0) trigger = d_sqrt(trigger);
            else trigger *= .95;
            break;
        default:
            if (trigger > 1.0) trigger = 1.0;
            else trigger *= .95;
            break;
    }
    int first = 99;
    for (int sfb = lane; sfb <= SFBMAX; sfb += LHIP_NL) {
        int f = 0;
        if (sfb < g.sfbmax && !((double)L.distort[sfb] < trigger)) { f = 1; if (sfb < first) first = sfb; }
        L.qmode[sfb] = f;
    }
    first = wave_min(first);
    wave_sync();
    if (T.noise_shaping_amp == 2) {          // amplify exactly one band
        for (int sfb = lane; sfb <= SFBMAX; sfb += LHIP_NL) L.qmode[sfb] = (sfb == first) ? 1 : 0;
        wave_sync();
    }
    for (int sfb = lane; sfb < g.sfbmax; sfb += LHIP_NL) if (L.qmode[sfb]) scalefac[sfb]++;
    q_amplify_flagged(g, ifqstep34, lane, L);
}

LHIP_DEV void q_inc_scalefac_scale(const Tables& T, GI& g, int32_t* scalefac, int lane, QuantLds& L) {
    for (int sfb = lane; sfb <= SFBMAX; sfb += LHIP_NL) {
        int f = 0;
        if (sfb < g.sfbmax) {
            int s = scalefac[sfb];
            if (g.preflag != 0) s += T.pretab[sfb];
            if ((s & 1) != 0) { s++; f = 1; }
            scalefac[sfb] = s >> 1;
        }
        L.qmode[sfb] = f;
    }
    wave_sync();
    g.preflag = 0;
    g.scalefac_scale = 1;
    q_amplify_flagged(g, 1.29683955465100964055, lane, L);
}

// inc_subblock_gain (Quantize.js:705-778); returns 1 on failure.  Short blocks only (sfb_lmax == 0).
LHIP_DEV int q_inc_subblock_gain(const Tables& T, GI& g, int32_t* scalefac, int lane, QuantLds& L) {
    for (int window = 0; window < 3; window++) {
        int s1 = 0, s2 = 0;
        for (int sfb = window + 3 * lane; sfb < g.sfbmax; sfb += 3 * LHIP_NL) {
            const int v = scalefac[sfb];
            if (sfb < g.sfbdivide) { if (s1 < v) s1 = v; } else { if (s2 < v) s2 = v; }
        }
        s1 = wave_max(s1); s2 = wave_max(s2);
        if (s1 < 16 && s2 < 8) continue;
        if (sbgain(g, window) >= 7) return 1;
        if (window == 0) g.subblock_gain[0]++; else if (window == 1) g.subblock_gain[1]++; else g.subblock_gain[2]++;
        // per band of this window: either lower the scalefactor or scale xrpow by IPOW20(210 + (s << ..))
        // (bands are handled one amplitude at a time; at most 13 bands + the sfb12 tail per window)
        wave_sync();
        for (int sfb = window; sfb < g.sfbmax + 3; sfb += 3) {   // last iteration: sfb == sfbmax + window, the sfb12 tail
            double amp;
            int doamp = 0;
            if (sfb < g.sfbmax) {
                int s = scalefac[sfb];
                s = s - (4 >> g.scalefac_scale);
                if (s >= 0) { if (lane == 0) scalefac[sfb] = s; }
                else {
                    if (lane == 0) scalefac[sfb] = 0;
                    amp = ipow20(T, 210 + (s << (g.scalefac_scale + 1)));
                    doamp = 1;
                }
            } else { amp = ipow20(T, 202); doamp = 1; }
            wave_sync();
            if (doamp) {
                float m = 0.f;
                const int st = L.start[sfb], w = L.width[sfb];
                for (int i = st + lane; i < st + w; i += LHIP_NL) {
                    const float v = (float)((double)L.xrpow[i] * amp);
                    L.xrpow[i] = v;
                    if (v > m) m = v;
                }
                m = wave_maxf(m);
                if ((double)m > g.xrpow_max) g.xrpow_max = m;
            }
        }
        wave_sync();
    }
    return 0;
}

// balance_noise (Quantize.js:793-846); returns 1 to continue the outer loop
LHIP_DEV int q_balance_noise(const Tables& T, GI& g, int32_t* scalefac, int lane, QuantLds& L) {
    q_amp_scalefac_bands(T, g, scalefac, lane, L);
    int status = q_loop_break(g, scalefac, lane, L);
    if (status) return 0;
    status = q_scale_bitcount(T, g, scalefac, lane);
    if (!status) return 1;
    if (T.noise_shaping > 1) {
        if (0 == g.scalefac_scale) {
            q_inc_scalefac_scale(T, g, scalefac, lane, L);
            status = 0;
        } else if (g.block_type == SHORT_TYPE && T.subblock_gain > 0) {
            status = (q_inc_subblock_gain(T, g, scalefac, lane, L) || q_loop_break(g, scalefac, lane, L));
        }
    }
    if (!status) status = q_scale_bitcount(T, g, scalefac, lane);
    return !status;
}

// bin_search_StepSize (Quantize.js:322-381) on the kept copy (ixb / sfb arrays)
LHIP_DEV int q_bin_search(const Tables& T, GI& g, int desired_rate, int start, int CurrentStep, int* step_out,
                          int lane, QuantLds& L) {
    int nBits, flagGoneOver = 0, Direction = 0;
    g.global_gain = start;
    desired_rate -= g.part2_length;
    for (;;) {
        int step;
        nBits = q_count_bits(T, g, L.sfb, L.ixb, nullptr, lane, L);
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
        g.global_gain += step;
        if (g.global_gain < 0) { g.global_gain = 0; flagGoneOver = 1; }
        if (g.global_gain > 255) { g.global_gain = 255; flagGoneOver = 1; }
    }
    while (nBits > desired_rate && g.global_gain < 255) {
        g.global_gain++;
        nBits = q_count_bits(T, g, L.sfb, L.ixb, nullptr, lane, L);
    }
    *step_out = (start - g.global_gain >= 4) ? 4 : 2;
    g.part2_3_length = nBits;
    return nBits;
}

LHIP_DEV int q_quant_compare(const NoiseRes& best, const NoiseRes& calc) {   // Quantize.js:481-568 case 9
    int better;
    if (best.over_count > 0) {
        better = calc.over_SSD <= best.over_SSD;
        if (calc.over_SSD == best.over_SSD) better = calc.bits < best.bits;
    } else {
        better = ((calc.max_noise < 0) && ((calc.max_noise * 10 + calc.bits) <= (best.max_noise * 10 + best.bits)));
    }
    if (best.over_count == 0) better = better && calc.bits < best.bits;
    return better;
}

// outer_loop (Quantize.js:871-1052).  g = kept copy (cod_info); seeds in/out via start/step.
LHIP_DEV void q_outer_loop(const Tables& T, GI& g, int targ_bits, int bs_start, int bs_step, int* bs_gain_out,
                           int lane, QuantLds& L) {
    int step_unused;
    q_bin_search(T, g, targ_bits, bs_start, bs_step, &step_unused, lane, L);
    *bs_gain_out = g.global_gain;                        // OldValue[ch] after this granule
    if (0 == T.noise_shaping) return;
    NoiseRes best; PrevNoise pn; pn.gain = 0; pn.sfb_count1 = 0;
    for (int i = lane; i <= SFBMAX; i += LHIP_NL) { L.pn_step[i] = 0; L.pn_noise[i] = 0.f; L.pn_noise_log[i] = 0.f; L.distort[i] = 0.f; }
    wave_sync();
    q_calc_noise(T, g, L.sfb, L.ixb, &best, &pn, lane, L);
    best.bits = g.part2_3_length;
    GI w = g;                                            // cod_info_w.assign(cod_info)
    for (int i = lane; i < 576; i += LHIP_NL) L.ixw[i] = L.ixb[i];
    for (int i = lane; i <= SFBMAX; i += LHIP_NL) L.sfw[i] = L.sfb[i];
    wave_sync();
    int best_part2_3_length = 9999999, age = 0;
    do {
        NoiseRes ni;
        const int search_limit = 3;
        int maxggain = 255;
        int bal_;
        { PH_BEGIN(); bal_ = q_balance_noise(T, w, L.sfw, lane, L); PH_END(L, PH_BALANCE); }
        if (!bal_) break;
        if (w.scalefac_scale != 0) maxggain = 254;
        const int huff_bits = targ_bits - w.part2_length;
        if (huff_bits <= 0) break;
        while ((w.part2_3_length = q_count_bits(T, w, L.sfw, L.ixw, &pn, lane, L)) > huff_bits && w.global_gain <= maxggain)
            w.global_gain++;
        if (w.global_gain > maxggain) break;
        if (best.over_count == 0) {
            while ((w.part2_3_length = q_count_bits(T, w, L.sfw, L.ixw, &pn, lane, L)) > best_part2_3_length && w.global_gain <= maxggain)
                w.global_gain++;
            if (w.global_gain > maxggain) break;
        }
        q_calc_noise(T, w, L.sfw, L.ixw, &ni, &pn, lane, L);
        ni.bits = w.part2_3_length;
        if (q_quant_compare(best, ni)) {
            best_part2_3_length = g.part2_3_length;       // value BEFORE the copy (reference quirk)
            best = ni;
            g = w;
            for (int i = lane; i < 576; i += LHIP_NL) L.ixb[i] = L.ixw[i];
            for (int i = lane; i <= SFBMAX; i += LHIP_NL) L.sfb[i] = L.sfw[i];
            wave_sync();
            age = 0;
        } else if (T.full_outer_loop == 0) {
            if (++age > search_limit && best.over_count == 0) break;
        }
    } while ((w.global_gain + w.scalefac_scale) < 255);
}

// ---------------------------------------------------------------------------------------------
// iteration_finish_one: best_scalefac_store (Takehiro.js:862-943), scfsi_calc (809-855),
// best_huffman_divide (727-800)
// ---------------------------------------------------------------------------------------------
LHIP_DEV void q_best_scalefac_store(const Tables& T, GI& g, int gr, int ch, int gr0_block_type, int* scfsi /*[4]*/,
                                    int lane, QuantLds& L) {
    int32_t* sf = L.sfb;
    int recalc = 0;
    {
        int any = 0;
        for (int sfb = lane; sfb < g.sfbmax; sfb += LHIP_NL) {
            const int st = L.start[sfb], w = L.width[sfb];
            int nz = 0;
            for (int j = st; j < st + w; j++) if (L.ixb[j] != 0) { nz = 1; break; }
            if (!nz) { sf[sfb] = -2; any = 1; }
        }
        if (wave_any(any)) recalc = -2;
        wave_sync();
    }
    if (0 == g.scalefac_scale && 0 == g.preflag) {
        int s = 0;
        for (int sfb = lane; sfb < g.sfbmax; sfb += LHIP_NL) if (sf[sfb] > 0) s |= sf[sfb];
        s = wave_or(s);
        if (0 == (s & 1) && s != 0) {
            for (int sfb = lane; sfb < g.sfbmax; sfb += LHIP_NL) if (sf[sfb] > 0) sf[sfb] >>= 1;
            g.scalefac_scale = recalc = 1;
            wave_sync();
        }
    }
    if (0 == g.preflag && g.block_type != SHORT_TYPE && T.mode_gr == 2) {
        int bad = 0;
        for (int sfb = 11 + lane; sfb < SBPSY_l; sfb += LHIP_NL) if (sf[sfb] < T.pretab[sfb] && sf[sfb] != -2) bad = 1;
        if (!wave_any(bad)) {
            for (int sfb = 11 + lane; sfb < SBPSY_l; sfb += LHIP_NL) if (sf[sfb] > 0) sf[sfb] -= T.pretab[sfb];
            g.preflag = recalc = 1;
            wave_sync();
        }
    }
    for (int i = 0; i < 4; i++) scfsi[i] = 0;
    if (T.mode_gr == 2 && gr == 1 && gr0_block_type != SHORT_TYPE && g.block_type != SHORT_TYPE) {
        // scfsi_calc: uniform scalar code over 21 bands, reading gr0's final scalefactors
        const int32_t* g0 = L.sf_gr0[ch];
        for (int i = 0; i < 4; i++) {
            int sfb, same = 1;
            for (sfb = T.scfsi_band[i]; sfb < T.scfsi_band[i + 1]; sfb++)
                if (g0[sfb] != sf[sfb] && sf[sfb] >= 0) { same = 0; break; }
            if (same) {
                wave_sync();
                if (lane == 0) for (sfb = T.scfsi_band[i]; sfb < T.scfsi_band[i + 1]; sfb++) sf[sfb] = -1;
                wave_sync();
                scfsi[i] = 1;
            }
        }
        int s1 = 0, c1 = 0, s2 = 0, c2 = 0, sfb;
        for (sfb = 0; sfb < 11; sfb++) { if (sf[sfb] == -1) continue; c1++; if (s1 < sf[sfb]) s1 = sf[sfb]; }
        for (; sfb < SBPSY_l; sfb++) { if (sf[sfb] == -1) continue; c2++; if (s2 < sf[sfb]) s2 = sf[sfb]; }
        for (int i = 0; i < 16; i++)
            if (s1 < T.slen1_n[i] && s2 < T.slen2_n[i]) {
                const int c = T.slen1_tab[i] * c1 + T.slen2_tab[i] * c2;
                if (g.part2_length > c) { g.part2_length = c; g.scalefac_compress = i; }
            }
        recalc = 0;
    }
    wave_sync();
    for (int sfb = lane; sfb < g.sfbmax; sfb += LHIP_NL) if (sf[sfb] == -2) sf[sfb] = 0;
    wave_sync();
    if (recalc != 0) q_scale_bitcount(T, g, sf, lane);
}

// Huffman statistics of scalefactor band `band` over pairs below `limit`:
// row 0 max, 1 t1, 2 table23(packed), 3 table56(packed), 4..6 t7-9, 7..9 t10-12, 10..12 t13-15, 13 largetbl hi, 14 lo, 15 #esc
LHIP_DEV void q_band_stats(const Tables& T, const int32_t* ix, int limit, int lane, QuantLds& L) {
    for (int band = lane; band < SBMAX_l; band += LHIP_NL) {
        int a = T.sfb_l[band], b = T.sfb_l[band + 1];
        if (b > limit) b = limit;
        int mx = 0;
        for (int p = a; p < b; p++) if (mx < ix[p]) mx = ix[p];
        int s[16];
        for (int k = 0; k < 16; k++) s[k] = 0;
        s[0] = mx;
        const int32_t *h1 = T.ht_hlen + T.ht_off[1];
        const int32_t *h7 = T.ht_hlen + T.ht_off[7], *h8 = T.ht_hlen + T.ht_off[8], *h9 = T.ht_hlen + T.ht_off[9];
        const int32_t *h10 = T.ht_hlen + T.ht_off[10], *h11 = T.ht_hlen + T.ht_off[11], *h12 = T.ht_hlen + T.ht_off[12];
        const int32_t *h13 = T.ht_hlen + T.ht_off[13], *h14 = T.ht_hlen + T.ht_off[14], *h15 = T.ht_hlen + T.ht_off[15];
        for (int p = a; p < b; p += 2) {
            const int x = ix[p], y = ix[p + 1];
            if (mx <= 1) s[1] += h1[x * 2 + y];
            if (mx <= 2) s[2] += T.table23[x * 3 + y];
            if (mx <= 3) s[3] += T.table56[x * 4 + y];
            if (mx <= 5) { const int q = x * 6 + y; s[4] += h7[q]; s[5] += h8[q]; s[6] += h9[q]; }
            if (mx <= 7) { const int q = x * 8 + y; s[7] += h10[q]; s[8] += h11[q]; s[9] += h12[q]; }
            if (mx <= 15) { const int q = x * 16 + y; s[10] += h13[q]; s[11] += h14[q]; s[12] += h15[q]; }
            {
                int xx = x, yy = y, n = 0;
                if (xx != 0) { if (xx > 14) { xx = 15; n++; } xx *= 16; }
                if (yy != 0) { if (yy > 14) { yy = 15; n++; } xx += yy; }
                const int lt = T.largetbl[xx];
                s[13] += lt >> 16; s[14] += lt & 0xffff; s[15] += n;
            }
        }
        for (int k = 0; k < 16; k++) L.bstat[k][band] = s[k];
    }
    wave_sync();
}

// choose_table for the union of whole bands [b0, b1) from the statistics above
LHIP_DEV int q_choose_from_stats(const Tables& T, int b0, int b1, int* bits, const QuantLds& L) {
    int mx = 0, s[16];
    for (int k = 1; k < 16; k++) s[k] = 0;
    for (int b = b0; b < b1; b++) {
        if (mx < L.bstat[0][b]) mx = L.bstat[0][b];
        for (int k = 1; k < 16; k++) s[k] += L.bstat[k][b];
    }
    if (mx == 0) return 0;
    if (mx == 1) { *bits += s[1]; return 1; }
    if (mx <= 3) {
        int t1 = T.huf_tbl_noESC[mx - 1];
        int sum = (mx == 2) ? s[2] : s[3];
        int sum2 = sum & 0xffff;
        sum >>= 16;
        if (sum > sum2) { sum = sum2; t1++; }
        *bits += sum;
        return t1;
    }
    if (mx <= 15) {
        const int t1 = T.huf_tbl_noESC[mx - 1];
        const int o = (t1 == 7) ? 4 : (t1 == 10) ? 7 : 10;
        int s1 = s[o], s2 = s[o + 1], s3 = s[o + 2], t = t1;
        if (s1 > s2) { s1 = s2; t++; }
        if (s1 > s3) { s1 = s3; t = t1 + 2; }
        *bits += s1;
        return t;
    }
    if (mx > IXMAX_VAL) { *bits = LARGE_BITS; return -1; }
    mx -= 15;
    int choice2, choice;
    for (choice2 = 24; choice2 < 32; choice2++) if (T.ht_linmax[choice2] >= mx) break;
    for (choice = choice2 - 8; choice < 24; choice++) if (T.ht_linmax[choice] >= mx) break;
    int sa = s[13] + s[15] * T.ht_xlen[choice], sb = s[14] + s[15] * T.ht_xlen[choice2];
    if (sa > sb) { sa = sb; choice = choice2; }
    *bits += sa;
    return choice;
}

LHIP_DEV void q_recalc_divide_sub(const Tables& T, const GI& c2, GI& g, const int32_t* ix, int lane, const QuantLds& L) {
    const int bigv = c2.big_values;
    for (int r2 = 2; r2 < SBMAX_l + 1; r2++) {
        const int a2 = T.sfb_l[r2];
        if (a2 >= bigv) break;
        int bits = L.r01_bits[r2 - 2] + c2.count1bits;
        if (g.part2_3_length <= bits) break;
        const int r2t = q_choose_table(T, ix, a2, bigv, &bits, lane);
        if (g.part2_3_length <= bits) continue;
        g = c2;
        g.part2_3_length = bits;
        g.region0_count = L.r01_div[r2 - 2];
        g.region1_count = r2 - 2 - L.r01_div[r2 - 2];
        g.table_select[0] = L.r0_tbl[r2 - 2];
        g.table_select[1] = L.r1_tbl[r2 - 2];
        g.table_select[2] = r2t;
    }
}

LHIP_DEV void q_best_huffman_divide(const Tables& T, GI& g, int lane, QuantLds& L) {
    const int32_t* ix = L.ixb;
    GI c2 = g;
    if (g.block_type == NORM_TYPE) {
        // recalc_divide_init: every (region0, region1) split evaluated from per-band statistics
        q_band_stats(T, ix, g.big_values, lane, L);
        const int bigv = g.big_values;
        for (int s = lane; s < 24; s += LHIP_NL) { L.r01_bits[s] = LARGE_BITS; L.r01_div[s] = 0; L.r0_tbl[s] = 0; L.r1_tbl[s] = 0; }
        wave_sync();
        // lane s owns the sum index s = r0 + r1 (the only slots recalc_divide_sub can read are 0..20)
        for (int s = lane; s <= 20; s += LHIP_NL) {
            int bb = LARGE_BITS, bd = 0, bt0 = 0, bt1 = 0;
            for (int r0 = 0; r0 < 16 && r0 <= s; r0++) {
                const int r1 = s - r0;
                if (r1 >= 8) continue;
                const int a1 = T.sfb_l[r0 + 1];
                if (a1 >= bigv) break;
                const int a2 = T.sfb_l[r0 + r1 + 2];
                if (a2 >= bigv) continue;            // the r1 loop of the reference has ended for this r0
                int r0bits = 0;
                const int r0t = q_choose_from_stats(T, 0, r0 + 1, &r0bits, L);
                int bits = r0bits;
                const int r1t = q_choose_from_stats(T, r0 + 1, r0 + r1 + 2, &bits, L);
                if (bb > bits) { bb = bits; bd = r0; bt0 = r0t; bt1 = r1t; }
            }
            L.r01_bits[s] = bb; L.r01_div[s] = bd; L.r0_tbl[s] = bt0; L.r1_tbl[s] = bt1;
        }
        wave_sync();
        q_recalc_divide_sub(T, c2, g, ix, lane, L);
    }
    int i = c2.big_values;
    if (i == 0 || (ix[i - 2] | ix[i - 1]) > 1) return;
    i = g.count1 + 2;
    if (i > 576) return;
    c2 = g;
    c2.count1 = i;
    int a1 = 0, a2 = 0;
    for (; i > c2.big_values; i -= 4) {
        const int p = ((ix[i - 4] * 2 + ix[i - 3]) * 2 + ix[i - 2]) * 2 + ix[i - 1];
        a1 += T.t32l[p];
        a2 += T.t33l[p];
    }
    c2.big_values = i;
    c2.count1table_select = 0;
    if (a1 > a2) { a1 = a2; c2.count1table_select = 1; }
    c2.count1bits = a1;
    if (c2.block_type == NORM_TYPE) q_recalc_divide_sub(T, c2, g, ix, lane, L);
    else {
        c2.part2_3_length = a1;
        a1 = T.sfb_l[7 + 1];
        if (a1 > i) a1 = i;
        if (a1 > 0) c2.table_select[0] = q_choose_table(T, ix, 0, a1, &c2.part2_3_length, lane);
        if (i > a1) c2.table_select[1] = q_choose_table(T, ix, a1, i, &c2.part2_3_length, lane);
        if (g.part2_3_length > c2.part2_3_length) g = c2;
    }
}

// ---------------------------------------------------------------------------------------------
// Bin-search seed chain (Quantize.js:324-326, 377-378: gfc.OldValue / gfc.CurrentStep).
// The seed of a granule-channel is a function of the bin-search results of the two most recent
// *active* granules of that channel: start = gain[p1], step = (start_of_p1 - gain[p1] >= 4) ? 4 : 2
// with start_of_p1 = gain[p2].  Frames are quantized speculatively (kb_quant with chain == 0 uses the
// reset seed), then kb_validate re-runs every bin search with the chain-implied seed and flags
// frames whose result differs; flagged frames are re-quantized with chain == 1 until none is left.
// ---------------------------------------------------------------------------------------------
struct Seed { int start, step; };

// seed seen by granule (k, gr) of channel ch, derived from the records of all earlier granules
LHIP_DEV Seed seed_before(const Workspace& W, const StreamDesc& sd, int C, int k, int gr, int ch) {
    const int32_t* carry = W.seed + ((int64_t)sd.fslot0 * C + ch) * 2;
    int g1 = -1, g2 = -1;                                    // gains of the last / second-to-last active granule
    for (int q = 2 * k + gr - 1; q >= 0; q--) {
        const GrSide* r = W.side + ((int64_t)sd.out_slot0 * 2 + q) * C + ch;
        if (r->active) {
            if (g1 < 0) g1 = r->bs_gain;
            else { g2 = r->bs_gain; break; }
        }
    }
    Seed s;
    if (g1 < 0) { s.start = carry[0]; s.step = carry[1]; }
    else {
        const int prev_start = (g2 < 0) ? carry[0] : g2;
        s.start = g1;
        s.step = (prev_start - g1 >= 4) ? 4 : 2;
    }
    return s;
}

LHIP_DEV int frame_bits_of(const Tables& T, int padding) {
    return 8 * js_toint32((double)((T.version + 1) * 72000 * T.brate) / T.out_samplerate + padding);
}

// padding bit of the k-th frame of this batch (Encoder.js:442-446): slot_lag is decremented by frac_SpF
// every frame and wrapped by out_samplerate whenever it drops below zero
LHIP_DEV int frame_padding(const Tables& T, const StreamDesc& sd, int k) {
    if (T.frac_SpF == 0) return 0;
    // value held before frame k:  lag0 - k*frac  (mod out_samplerate), representative in [0, out_samplerate)
    int64_t m = ((int64_t)sd.slot_lag - (int64_t)k * T.frac_SpF) % T.out_samplerate;
    if (m < 0) m += T.out_samplerate;
    return (m - T.frac_SpF) < 0 ? 1 : 0;
}

LHIP_DEV void targ_bits_for(const Tables& T, int mean_bits, int gr, int ResvSize, int* targ) {
    // on_pe + ResvMaxBits(cbr = gr) with the reservoir disabled (QuantizePVT.js:421-484, Reservoir.js:190-229)
    const int C = T.channels_out;
    int rs = ResvSize, tbits, bits = 0;
    if (gr != 0) rs += mean_bits;
    tbits = mean_bits;
    if (rs * 10 > 0) tbits += rs;
    for (int ch = 0; ch < C; ++ch) {
        const double t = (double)tbits / C;
        targ[ch] = js_toint32(t < MAX_BITS_PER_CHANNEL ? t : (double)MAX_BITS_PER_CHANNEL);
        bits += targ[ch];
    }
    if (bits > MAX_BITS_PER_GRANULE)
        for (int ch = 0; ch < C; ++ch) {
            targ[ch] = js_toint32((double)targ[ch] * MAX_BITS_PER_GRANULE);
            targ[ch] = js_toint32((double)targ[ch] / bits);
        }
}

// One wave per frame slot.  chain == 0: speculative reset seed (exact for the first frame of a stream
// batch, whose seed is the carried one); chain == 1: chain-implied seed (repair pass, flagged frames only).
LHIP_DEV void kb_quant(const Tables& T, const PowBase& pb10, const Workspace& W, const StreamDesc* SD, int fslot,
                       int chain, int lane, QuantLds& L) {
    const int C = T.channels_out;
    const int st = W.fslot_stream[fslot];
    const StreamDesc sd = SD[st];
    const int k = fslot - sd.fslot0 - 1;
    if (k < 0) return;
    const int fidx = sd.out_slot0 + k;                    // dense frame index
    if (chain && !W.seed_flag[fidx]) return;
#ifdef LHIP_PHASE_PROF
    if (lane == 0) for (int i = 0; i < 32; i++) L.prof[i] = 0;
    const unsigned long long ph_total0_ = __builtin_amdgcn_s_memtime();
#endif
    const double ath_adjust = W.ath_adjust[fslot];        // after adjust_ATH of this frame
    const int padding = frame_padding(T, sd, k);
    const int mean_bits = (frame_bits_of(T, padding) - T.sideinfo_len * 8) / T.mode_gr;
    Seed seed[2];
    for (int ch = 0; ch < C; ch++) {
        if (chain || k == 0) seed[ch] = seed_before(W, sd, C, k, 0, ch);
        else { seed[ch].start = 180; seed[ch].step = 4; }
    }
    int ResvSize = 0;
    int gr0_bt[2] = {0, 0};
    for (int gr = 0; gr < 2; gr++) {
        const int gslot = sd.gslot0 + 1 + 2 * k + gr;
        int targ[2];
        targ_bits_for(T, mean_bits, gr, ResvSize, targ);
        for (int ch = 0; ch < C; ch++) {
            GI g;
            const int bt = W.blocktype[(int64_t)gslot * C + ch];
            const double masking_lower = (bt != SHORT_TYPE) ? T.masking_lower_long : T.masking_lower_short;
            const float* ratio = W.E + ((int64_t)(gslot - 1) * C + ch) * E_STRIDE;   // thresholds of the previous psy call
            { PH_BEGIN(); q_init_outer_loop(T, pb10, ath_adjust, g, bt, W.xr + ((int64_t)gslot * C + ch) * 576, lane, L); PH_END(L, PH_INIT); }
            int active = 0, bs_gain = 0;
            const Seed used = seed[ch];
            if (q_init_xrpow(g, lane, L)) {
                active = 1;
                { PH_BEGIN(); q_calc_xmin(T, ath_adjust, masking_lower, ratio, g, lane, L); PH_END(L, PH_XMIN); }
                q_outer_loop(T, g, targ[ch], used.start, used.step, &bs_gain, lane, L);
                seed[ch].step = (used.start - bs_gain >= 4) ? 4 : 2;
                seed[ch].start = bs_gain;
            } else {
                for (int i = lane; i < 576; i += LHIP_NL) L.ixb[i] = 0;
                wave_sync();
            }
            int scfsi[4];
            { PH_BEGIN(); q_best_scalefac_store(T, g, gr, ch, gr0_bt[ch], scfsi, lane, L); PH_END(L, PH_SFSTORE); }
            if (T.use_best_huffman == 1) { PH_BEGIN(); q_best_huffman_divide(T, g, lane, L); PH_END(L, PH_HUFFDIV); }
            ResvSize -= g.part2_3_length + g.part2_length;
            if (gr == 0) {
                gr0_bt[ch] = g.block_type;
                for (int i = lane; i <= SFBMAX; i += LHIP_NL) L.sf_gr0[ch][i] = L.sfb[i];
            }
            // ---- publish the record and the signed quantized spectrum ----
            GrSide* out = W.side + ((int64_t)fidx * 2 + gr) * C + ch;
            if (lane == 0) {
                out->part2_3_length = g.part2_3_length; out->part2_length = g.part2_length; out->big_values = g.big_values;
                out->count1 = g.count1; out->global_gain = g.global_gain; out->scalefac_compress = g.scalefac_compress;
                out->block_type = g.block_type;
                for (int i = 0; i < 3; i++) { out->table_select[i] = g.table_select[i]; out->subblock_gain[i] = g.subblock_gain[i]; }
                out->region0_count = g.region0_count; out->region1_count = g.region1_count; out->preflag = g.preflag;
                out->scalefac_scale = g.scalefac_scale; out->count1table_select = g.count1table_select;
                out->sfbmax = g.sfbmax; out->sfbdivide = g.sfbdivide;
                out->active = active; out->bs_start = used.start; out->bs_step_in = used.step; out->bs_gain = bs_gain;
                out->targ_bits = targ[ch];
                out->scfsi = scfsi[0] | (scfsi[1] << 1) | (scfsi[2] << 2) | (scfsi[3] << 3);
            }
            for (int i = lane; i < SFBMAX; i += LHIP_NL) out->scalefac[i] = L.sfb[i];
            int16_t* l3o = W.l3 + (((int64_t)fidx * 2 + gr) * C + ch) * 576;
            for (int i = lane; i < 576; i += LHIP_NL) {
                const int v = L.ixb[i];
                l3o[i] = (int16_t)(((double)L.xr[i] < 0) ? -v : v);
            }
            wave_sync();
        }
    }
    if (chain && lane == 0) W.seed_flag[fidx] = 0;
#ifdef LHIP_PHASE_PROF
    if (lane == 0) {
        L.prof[PH_TOTAL] = __builtin_amdgcn_s_memtime() - ph_total0_; L.prof[16 + PH_TOTAL] = 1;
        for (int i = 0; i < 32; i++) atomicAdd((unsigned long long*)W.prof + i, L.prof[i]);
    }
#endif
}

// Re-run the bin searches of a frame with the chain-implied seeds; flag the frame if any result differs.
LHIP_DEV void kb_validate(const Tables& T, const PowBase& pb10, const Workspace& W, const StreamDesc* SD, int fslot,
                          int lane, QuantLds& L) {
    const int C = T.channels_out;
    const int st = W.fslot_stream[fslot];
    const StreamDesc sd = SD[st];
    const int k = fslot - sd.fslot0 - 1;
    if (k < 0) return;
    const int fidx = sd.out_slot0 + k;
    const double ath_adjust = W.ath_adjust[fslot];
    int bad = 0;
    for (int gr = 0; gr < 2 && !bad; gr++) {
        const int gslot = sd.gslot0 + 1 + 2 * k + gr;
        for (int ch = 0; ch < C && !bad; ch++) {
            const GrSide* rec = W.side + ((int64_t)fidx * 2 + gr) * C + ch;
            if (!rec->active) continue;
            const Seed s = seed_before(W, sd, C, k, gr, ch);
            if (s.start == rec->bs_start && s.step == rec->bs_step_in) continue;     // already quantized with this seed
            GI g;
            q_init_outer_loop(T, pb10, ath_adjust, g, W.blocktype[(int64_t)gslot * C + ch],
                              W.xr + ((int64_t)gslot * C + ch) * 576, lane, L);
            q_init_xrpow(g, lane, L);
            // max_nonzero_coeff is set by calc_xmin in the reference before the bin search
            if (g.block_type != SHORT_TYPE) {
                int t = -1;
                for (int i = lane; i < 576; i += LHIP_NL) if (!((double)L.xr[i] == 0)) t = i;
                t = wave_max(t);
                g.max_nonzero_coeff = (t >= 575) ? 575 : t + 1;
            }
            int step_unused;
            q_bin_search(T, g, rec->targ_bits, s.start, s.step, &step_unused, lane, L);
            if (g.global_gain != rec->bs_gain) bad = 1;
        }
    }
    if (lane == 0 && bad) {
        W.seed_flag[fidx] = 1;
#ifdef LHIP_HOSTSIM
        W.nflagged[0] += 1;
#else
        atomicAdd(W.nflagged, 1);
#endif
    }
}

}  // namespace lhip
