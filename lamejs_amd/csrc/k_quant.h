// CBR quantization / noise-shaping loop as a wave program: one 64-lane wavefront owns one frame
// and runs granule 0 (all channels) then granule 1, exactly as the reference's
// CBRNewIterationLoop.iteration_loop does (CBRNewIterationLoop.js:25-90).  Inside a
// granule-channel the *control flow* of the reference (Quantize.js outer_loop 871-1052,
// bin_search_StepSize 322-381, balance_noise 793-846, ...) is executed uniformly by all lanes,
// while every loop over the 576 spectral lines / the scalefactor bands is spread over the lanes:
//   - quantize x^(3/4) lines (Takehiro.js:102-314): five pairs per lane, per-band decisions as ballots
//   - Huffman bit counting (Takehiro.js:319-628): region maxima and packed length sums by
//     integer wave reductions (exact, order-free)
//   - noise per scalefactor band (QuantizePVT.js:725-878): nine consecutive lines per lane, the bands' f64 sums in the
//     reference's line order as a systolic fold over the lanes (f64 sums are order-sensitive, so they are never tree-reduced);
//     the per-band rest (distortion ratio, class, cache) one lane per band
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
    int firstcut;               // first band reaching past max_nonzero_coeff (64: none): fixed once max_nonzero_coeff is (q_set_firstcut)
};

struct NoiseRes { double max_noise; int over_count, over_SSD, bits; };

// Re-asserts that the granule state is wave-uniform (it is by construction: every lane executes the same
// decisions).  The compiler's divergence analysis loses that fact at joins of lane-guarded code, and once one
// loop-carried scalar counts as divergent every decision of the loop turns into exec-masked vector code.
LHIP_DEV void uni_gi(GI& g) {
    g.xrpow_max = unid(g.xrpow_max);
    g.part2_3_length = uni(g.part2_3_length); g.big_values = uni(g.big_values); g.count1 = uni(g.count1); g.global_gain = uni(g.global_gain);
    g.scalefac_compress = uni(g.scalefac_compress); g.block_type = uni(g.block_type);
    for (int i = 0; i < 3; i++) g.table_select[i] = uni(g.table_select[i]);
    for (int i = 0; i < 4; i++) g.subblock_gain[i] = uni(g.subblock_gain[i]);
    g.region0_count = uni(g.region0_count); g.region1_count = uni(g.region1_count); g.preflag = uni(g.preflag); g.scalefac_scale = uni(g.scalefac_scale);
    g.count1table_select = uni(g.count1table_select); g.part2_length = uni(g.part2_length); g.sfb_lmax = uni(g.sfb_lmax); g.sfb_smin = uni(g.sfb_smin);
    g.psy_lmax = uni(g.psy_lmax); g.sfbmax = uni(g.sfbmax); g.psymax = uni(g.psymax); g.sfbdivide = uni(g.sfbdivide);
    g.count1bits = uni(g.count1bits); g.max_nonzero_coeff = uni(g.max_nonzero_coeff); g.firstcut = uni(g.firstcut);
}

// optional phase profiling (build with -DLHIP_PHASE_PROF; never in the product library)
#if defined(LHIP_PHASE_PROF) && !defined(LHIP_HOSTSIM)
#define PH_BEGIN() const unsigned long long ph_t0_ = __builtin_amdgcn_s_memtime()
#define PH_END(L, id) do { if (lane == 0) { (L).prof[id] += (unsigned int)(__builtin_amdgcn_s_memtime() - ph_t0_); (L).prof[32 + id] += 1; } } while (0)
#define PH_MARK(L, id, t) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (lane == 0) { (L).prof[id] += (unsigned int)(n_ - (t)); (L).prof[32 + id] += 1; } (t) = n_; } while (0)
#define PH_NOW() __builtin_amdgcn_s_memtime()
#else
#define PH_BEGIN() do {} while (0)
#define PH_END(L, id) do {} while (0)
#define PH_MARK(L, id, t) do {} while (0)
#define PH_NOW() 0ull
#endif
// balance_noise sub-phase marks share the accumulator slots of psyA's stamps: only with -DLHIP_PROF_BALANCE
#ifdef LHIP_PROF_BALANCE
#define PH_MARKB(L, id, t) PH_MARK(L, id, t)
#else
#define PH_MARKB(L, id, t) do {} while (0)
#endif
enum { PH_INIT, PH_XRPOW, PH_XMIN, PH_QUANTIZE, PH_COUNT, PH_NOISE, PH_BALANCE, PH_SFSTORE, PH_HUFFDIV, PH_PUBLISH, PH_COPY, PH_TOTAL,
       PH_C_LOAD, PH_C_QUADS, PH_C_MAX, PH_C_SUMS, PH_C_FIN, PH_N_WALK, PH_N_TERMS, PH_N_SUMS, PH_Q_MASK, PH_Q_LINES, PH_N, PH_DRAIN = 29 /* 22..28 belong to psyA's stamps */,
       PH_N_LINES = 30, PH_N_FOLD = 31 /* calc_noise: per-line terms / systolic fold; PH_N_TERMS is then the per-band part */,
       PH_B_TRIG = 22, PH_B_AMP = 23, PH_B_BREAK = 24, PH_B_BITCOUNT = 25 /* balance_noise parts (profiling builds: these slots are psyA's otherwise) */ };

// Read-only tables staged once per workgroup in LDS (shared by the waves of the block): everything the
// inner loops gather from -- avoids ~1 us HBM/L2 round trips inside serially dependent code.
enum { QT_N = 256 };
// Layout of the Huffman code-length pool (bytes): the table sizes are fixed by ISO 11172-3, so the offsets are
// compile-time constants (tables 4 and 14 do not exist in the standard; the reference keeps a 256-entry row 14).
enum { HL_T1 = 0, HL_T2 = 4, HL_T3 = 13, HL_T5 = 22, HL_T6 = 38, HL_T7 = 54, HL_T8 = 90, HL_T9 = 126, HL_T10 = 162, HL_T11 = 226,
       HL_T12 = 290, HL_T13 = 354, HL_T14 = 610, HL_T15 = 866, HL_EHI = 1122, HL_ELO = 1378, HL_END = 1634 };
LHIP_DEV int hl_off(int t) {
    switch (t) {
        case 1: return HL_T1; case 2: return HL_T2; case 3: return HL_T3; case 5: return HL_T5; case 6: return HL_T6;
        case 7: return HL_T7; case 8: return HL_T8; case 9: return HL_T9; case 10: return HL_T10; case 11: return HL_T11;
        case 12: return HL_T12; case 13: return HL_T13; case 14: return HL_T14; case 15: return HL_T15; default: return 0;
    }
}
struct alignas(16) QuantTabs {
    float pow43[QT_N], adj43[QT_N];
    float ipow20[Q_MAX], pow20[Q_MAX + Q_MAX2 + 1];
    int32_t sfb_l[SBMAX_l + 1], sfb_s[SBMAX_s + 1], pretab[SBMAX_l];
    uint16_t hoff[16];
    uint8_t hlen[HL_END];        // code-length pool: tables 1-3, 5-15, then the two ESC length tables (layout: hl_off)
    uint8_t t32l[16], t33l[16];
    uint8_t l2s_long[576], l2s_short[576];
    uint32_t bvtab[289];         // count_bits, NORM blocks, by big_values / 2: region borders, region counts, sfb_count1 (Tables::bvtab)
    uint8_t huf_tbl_noESC[16], ht_xlen[34];
    uint16_t ht_linmax[34];
    // Huffman region plans by region maximum (count_bits): entries 0..15 = the maximum itself, 16..28 = bit length 1..13
    // of (maximum - 15), 29 = beyond IXMAX_VAL.  d0/d1 = pool offsets of the candidate length tables | row stride (what
    // the pairs gather with), pw = kind | candidate table numbers | linbits of the two ESC candidates (plan_word)
    struct PlanEnt { uint32_t d0, d1, pw; } plan[30];
    // scale_bitcount candidates (Takehiro.js:980-1030): row k = slen1_n | slen2_n << 8 | scale_long << 16 | scale_short << 24
    uint32_t sbc[16];
    // calc_noise's systolic fold (lane l owns the lines 9 l .. 9 l + 8): bit k = line 9 l + k is the first of its scalefactor band,
    // bit 16 + k = it is the last one; [0] long blocks, [1] short blocks.  wpre: widest band among bands 0 .. b (long / short).
    uint32_t fold_marks[2][64];
    uint16_t wpre_long[24], wpre_short[40];
    // analog silence (Quantize.js:147-202): pseudo-band borders above sfb21 / sfb12, their ATH values ([0..5] long, [6..11] short) and the
    // two band factors they are scaled with
    int16_t psfb21[PSFB21 + 1], psfb12[PSFB12 + 1];
    float ath_psfb[PSFB21 + PSFB12], fact_psfb[2];
};

LHIP_DEV void q_fill_plans(QuantTabs& Q, int tid, int nthr);

LHIP_DEV void q_load_tabs(const Tables& T, QuantTabs& Q, int tid, int nthr) {
    q_fill_plans(Q, tid, nthr);
    for (int i = tid; i < QT_N; i += nthr) { Q.pow43[i] = T.pow43[i]; Q.adj43[i] = T.adj43[i]; }
    for (int i = tid; i < Q_MAX; i += nthr) Q.ipow20[i] = T.ipow20[i];
    for (int i = tid; i < Q_MAX + Q_MAX2 + 1; i += nthr) Q.pow20[i] = T.pow20[i];
    for (int i = tid; i < 16; i += nthr) { Q.t32l[i] = (uint8_t)T.t32l[i]; Q.t33l[i] = (uint8_t)T.t33l[i]; }
    for (int i = tid; i < SBMAX_l + 1; i += nthr) Q.sfb_l[i] = T.sfb_l[i];
    for (int i = tid; i < SBMAX_s + 1; i += nthr) Q.sfb_s[i] = T.sfb_s[i];
    for (int i = tid; i < SBMAX_l; i += nthr) Q.pretab[i] = T.pretab[i];
    for (int i = tid; i < 16; i += nthr)
        Q.sbc[i] = (uint32_t)T.slen1_n[i] | ((uint32_t)T.slen2_n[i] << 8) | ((uint32_t)T.scale_long[i] << 16) | ((uint32_t)T.scale_short[i] << 24);
    for (int i = tid; i < 289; i += nthr) Q.bvtab[i] = (uint32_t)T.bvtab[i];
    for (int i = tid; i < 15; i += nthr) Q.huf_tbl_noESC[i] = (uint8_t)T.huf_tbl_noESC[i];
    for (int i = tid; i < 34; i += nthr) { Q.ht_xlen[i] = (uint8_t)T.ht_xlen[i]; Q.ht_linmax[i] = (uint16_t)T.ht_linmax[i]; }
    // Huffman code-length pool
    if (tid == 0) { Q.hoff[0] = 0; Q.hoff[4] = 0; }
    for (int t = 1; t < 16; t++) {
        if (t == 4) continue;
        const int xl = T.ht_xlen[t];
        const int n = (t == 14) ? 256 : xl * xl, off = hl_off(t);
        if (tid == 0) Q.hoff[t] = (uint16_t)off;
        for (int i = tid; i < n; i += nthr) Q.hlen[off + i] = (uint8_t)T.ht_hlen[T.ht_off[t] + i];
    }
    for (int i = tid; i < 256; i += nthr) { Q.hlen[HL_EHI + i] = (uint8_t)(T.largetbl[i] >> 16); Q.hlen[HL_ELO + i] = (uint8_t)(T.largetbl[i] & 0xffff); }
    for (int d = tid; d < 576; d += nthr) {
        int sfb = 0;
        while (T.sfb_l[sfb + 1] <= d) sfb++;
        Q.l2s_long[d] = (uint8_t)sfb;
        sfb = 0;
        while (3 * T.sfb_s[sfb + 1] <= d) sfb++;
        const int st = T.sfb_s[sfb], w = T.sfb_s[sfb + 1] - st;
        const int r = d - 3 * st, win = r / w, l = st + (r - win * w);
        Q.l2s_short[d] = (uint8_t)(3 * sfb + win);
        (void)l;
    }
    for (int e = tid; e < PSFB21 + 1; e += nthr) Q.psfb21[e] = (int16_t)T.psfb21[e];
    for (int e = tid; e < PSFB12 + 1; e += nthr) Q.psfb12[e] = (int16_t)T.psfb12[e];
    for (int e = tid; e < PSFB21 + PSFB12; e += nthr) Q.ath_psfb[e] = e < PSFB21 ? T.ATH_psfb21[e] : T.ATH_psfb12[e - PSFB21];
    if (tid == 0) { Q.fact_psfb[0] = T.longfact[21]; Q.fact_psfb[1] = T.shortfact[12]; }
    for (int e = tid; e < 2 * 64; e += nthr) Q.fold_marks[e >> 6][e & 63] = (uint32_t)T.fold_marks[e];
    for (int e = tid; e < 24; e += nthr) Q.wpre_long[e] = (uint16_t)T.wpre[e];
    for (int e = tid; e < 40; e += nthr) Q.wpre_short[e] = (uint16_t)T.wpre[24 + e];
}

// The tables do not change after lhip_create: they are gathered ONCE per configuration into an image in HBM (g_build_qtabs) and a workgroup
// copies that image flat -- 16 bytes per thread and step -- instead of repeating q_load_tabs' gathers from fifteen source tables, which cost a
// one-frame launch 19 us of dependent global loads before its first stage (profiles/r05_pass1_frame_prof_*.txt).
LHIP_DEV void q_copy_tabs(const Tables& T, QuantTabs& Q, int tid, int nthr) {
    static_assert(sizeof(QuantTabs) % 16 == 0, "QuantTabs is copied in 16-byte pieces");
    struct alignas(16) V4 { uint32_t a, b, c, d; };
    const V4* src = (const V4*)T.qtabs_img;
    V4* dst = (V4*)&Q;
    for (int i = tid; i < (int)(sizeof(QuantTabs) / 16); i += nthr) dst[i] = src[i];
}

struct QuantLds {
    float xr[576];
    union {                      // xrpow is dead once the outer loop has finished; the Huffman-split scratch reuses it
        float xrpow[576];
        struct { int32_t bstat[16][SBMAX_l + 2]; int16_t cand[21][16]; } hd;   // cand: bits clipped to 0x7fff (LARGE_BITS marker)
    };
    int16_t ixw[576];            // l3_enc of the working copy (cod_info_w)
    int32_t sfw[SFBMAX + 1], sfb[SFBMAX + 1];     // scalefac working / kept
    int16_t width[SFBMAX + 1], window[SFBMAX + 1], start[SFBMAX + 2];
    float xmin[SFBMAX + 1], distort[SFBMAX + 1];
    double rxmin[SFBMAX + 1];                    // 1 / xmin, correctly rounded (0: xmin outside the range div_by_f32 is proven for): calc_noise divides by xmin in every call
    int32_t pn_step[SFBMAX + 1];
    float pn_dist[SFBMAX + 1];                   // the cache of the reference's Float32 noise, as what every later call derives from it: (float)(noise / xmin)
    // the cache of the reference's Float32 noise_log, without the logarithm: pn_x = the band's noise / xmin (f64) as last evaluated,
    // pn_cls = noise_class of the Float32 copy of its logarithm (lhip_math.h); the logarithm itself is only formed when max_noise is read
    union {                      // pn_x is dead once the outer loop has finished; the Huffman split's range-maximum tables reuse it
        double pn_x[SFBMAX + 1];
        int16_t hmax[5][24];     // hmax[t][b] = largest quantized value in the bands b .. b + 2^t - 1 (q_band_max_tables)
    };
    int16_t pn_cls[SFBMAX + 1];
    int32_t qmode[SFBMAX + 1];
    float bstep[SFBMAX + 1];                     // calc_noise: POW20 of the band's step (what the per-line pass reads)
    int8_t sf_gr0[2][SFBMAX + 1];                 // final gr0 scalefactors per channel (for scfsi): -2..15
    union {                      // calc_noise band sums live only inside the outer loop, the split tables only after it
        struct { int32_t r01_bits[24], r01_div[24], r0_tbl[24], r1_tbl[24], r2_bits[24], r2_tbl[24]; };
        double nsum[SFBMAX + 1];
        struct { int32_t bs_tab[BS_TAB_MAX], bs_asg[BS_TAB_MAX]; } memo;   // bin-search memo, flushed to the side record before the first calc_noise
    };
    alignas(8) uint32_t rdesc[4][2];   // per Huffman region: offsets of its candidate length tables | row stride
    int32_t gkeep[24];                 // the mutable scalars of the kept quantization (cod_info) while the outer loop runs on cod_info_w
    double ath_pseudo[PSFB21 + PSFB12];   // this frame's analog-silence thresholds (q_ath_pseudo): [0..5] long blocks, [6..11] short blocks
#ifdef LHIP_PHASE_PROF
    unsigned int prof[64];           // per-frame cycle sums fit 32 bits
#endif
};

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
LHIP_DEV double ipow20(const QuantTabs& Q, int x) { return (double)Q.ipow20[x]; }
LHIP_DEV double pow20(const QuantTabs& Q, int x) { return (double)Q.pow20[x + Q_MAX2]; }
LHIP_DEV double pow43v(const Tables& T, const QuantTabs& Q, int i) {
    float v = Q.pow43[i < QT_N ? i : QT_N - 1];
    if (i >= QT_N) v = T.pow43[i];          // rare: large quantized values
    return (double)v;
}
LHIP_DEV double adj43v(const Tables& T, const QuantTabs& Q, int i) {
    float v = Q.adj43[i < QT_N ? i : QT_N - 1];
    if (i >= QT_N) v = T.adj43[i];
    return (double)v;
}
LHIP_DEV const uint8_t* line2sfb(const QuantTabs& Q, int block_type) { return block_type == SHORT_TYPE ? Q.l2s_short : Q.l2s_long; }
LHIP_DEV const uint8_t* hlen_of(const QuantTabs& Q, int t) { return Q.hlen + Q.hoff[t]; }

// QuantizePVT.js:541-561
LHIP_DEV double athAdjust(const Tables& T, const PowBase& pb10, double a, double x, double athFloor) {
    const double o = 90.30873362, p = 94.82444863;
    double u = v8_log10(x) * 10.0;
    const double v = a * a;
    double w = 0.0;
    u -= athFloor;
    if (v > 1E-20) w = 1. + v8_log10(v) * const_here(10.0 / o);
    if (w < 0) w = 0.;
    u *= w;
    u += athFloor + const_here(o) - const_here(p);
    return v8_pow_base(pb10, 0.1 * u);
}

LHIP_DEV int sbgain(const GI& g, int w) {   // subblock_gain[w] without a dynamically indexed register array
    return w == 0 ? g.subblock_gain[0] : w == 1 ? g.subblock_gain[1] : w == 2 ? g.subblock_gain[2] : g.subblock_gain[3];
}

// the band's sub-block gain: only short blocks have one (every other block type keeps its bands in "window 3", whose gain stays 0 --
// nothing ever increments subblock_gain[3]), so long blocks neither fetch the window nor select among the gains
LHIP_DEV int band_sbgain(const GI& g, const int16_t* window, int sfb) { return g.block_type == SHORT_TYPE ? sbgain(g, window[sfb]) : 0; }
LHIP_DEV int sf_step(const QuantTabs& Q, const GI& g, const int32_t* scalefac, const int16_t* window, int sfb) {
    return g.global_gain - ((scalefac[sfb] + (g.preflag != 0 ? Q.pretab[sfb] : 0)) << (g.scalefac_scale + 1))
           - band_sbgain(g, window, sfb) * 8;
}

// ---------------------------------------------------------------------------------------------
// init_outer_loop (Quantize.js:204-306) incl. psfb21_analogsilence (147-202)
// xr_g: this granule-channel's MDCT output in HBM (natural order)
// ---------------------------------------------------------------------------------------------
// xr_wb != nullptr: lines zeroed by the analog-silence rule are also zeroed in HBM, so that a later pass over the
// same granule (kb_validate) can skip the rule (`skip_silence`); the rule is idempotent.
// Where the lines come from: the channel's own MDCT output, or -- frames that joint stereo codes M/S -- mid / side of both
// channels' outputs (ms_convert, Quantize.js:76-83).  M/S granules are never written back (the L / R lines must survive for the
// other channel and for later passes), so they never skip the silence rule either.
struct XrSrc { const float* a; const float* b; int side; };       // b == nullptr: a is the spectrum; else a = left, b = right
LHIP_DEV XrSrc xr_source(const Workspace& W, int C, int gslot, int ch, int ms) {
    XrSrc X;
    if (!ms) { X.a = W.xr + ((int64_t)gslot * C + ch) * 576; X.b = nullptr; X.side = 0; }
    else { X.a = W.xr + (int64_t)gslot * C * 576; X.b = X.a + 576; X.side = ch; }
    return X;
}
LHIP_DEV float xr_at(const XrSrc& X, int i) {
    if (!X.b) return X.a[i];
    const double l = X.a[i], r = X.b[i];
    return X.side ? (float)((l - r) * (LHIP_SQRT2 * 0.5)) : (float)((l + r) * (LHIP_SQRT2 * 0.5));
}
// The analog-silence thresholds of a frame (Quantize.js:147-202: athAdjust of the pseudo bands' ATH, scaled by the band factor of
// sfb21 / sfb12): functions of the frame's ATH adjustment alone, so they are formed once per frame -- [0..5] for long blocks,
// [6..11] for short ones -- and not once per granule and channel.
LHIP_DEV void q_ath_pseudo(const Tables& T, const PowBase& pb10, double ath_adjust, int lane, QuantLds& L, const QuantTabs& Q) {
    lane = lane_anew(lane);
    wave_sync();
    LHIP_LANE_ONCE(e, 0, PSFB21 + PSFB12) {
        double a = athAdjust(T, pb10, ath_adjust, Q.ath_psfb[e], T.ATH_floor);
        const double f = (double)Q.fact_psfb[e < PSFB21 ? 0 : 1];
        if (f > 1e-12) a *= f;
        L.ath_pseudo[e] = a;
    }
    wave_sync();
}

// the lines of a granule-channel from HBM, in the order the quantizer keeps them (short blocks: the three windows of a band as
// consecutive runs).  All loads of a lane are issued before the first one is needed (a lane-strided loop compiles to one load per
// trip, each waited for by itself: nine round trips to memory).
LHIP_DEV void q_load_lines(const XrSrc& X, int is_short, int lane, QuantLds& L, const QuantTabs& Q) {
    lane = fresh_lane(lane);
#if LHIP_NL == 1
    for (int d = 0; d < 576; d++) {
        int src = d;
        if (is_short) {
            const int sw = Q.l2s_short[d], sfb = sw / 3, win = sw - 3 * sfb;
            const int st = Q.sfb_s[sfb], w = Q.sfb_s[sfb + 1] - st;
            src = 3 * (st + (d - 3 * st - win * w)) + win;
        }
        L.xr[d] = xr_at(X, src);
    }
    (void)lane;
#else
    enum { NV = 576 / LHIP_NL };
    int src[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const int d = lane + LHIP_NL * k;
        src[k] = d;
        if (is_short) {
            const int sw = Q.l2s_short[d], sfb = sw / 3, win = sw - 3 * sfb;
            const int st = Q.sfb_s[sfb], w = Q.sfb_s[sfb + 1] - st;
            src[k] = 3 * (st + (d - 3 * st - win * w)) + win;
        }
    }
    float va[NV], vb[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) { va[k] = X.a[src[k]]; vb[k] = 0.f; }
    if (X.b) {
#pragma unroll
        for (int k = 0; k < NV; k++) vb[k] = X.b[src[k]];
    }
#pragma unroll
    for (int k = 0; k < NV; k++) { LHIP_PIN_LOADED(va[k]); LHIP_PIN_LOADED(vb[k]); }
#pragma unroll
    for (int k = 0; k < NV; k++) {
        float v = va[k];
        if (X.b) { const double l = va[k], r = vb[k]; v = X.side ? (float)((l - r) * (LHIP_SQRT2 * 0.5)) : (float)((l + r) * (LHIP_SQRT2 * 0.5)); }
        L.xr[lane + LHIP_NL * k] = v;
    }
#endif
}

// ath_ready: L.ath_pseudo holds this frame's thresholds (q_ath_pseudo); only read when the silence rule runs
LHIP_DEV void q_init_outer_loop(const Tables& T, const PowBase& pb10, double ath_adjust, GI& g, int block_type,
                                const XrSrc& xr_g, float* xr_wb, int skip_silence, int lane, QuantLds& L, const QuantTabs& Q) {
    lane = fresh_lane(lane);
    unsigned long long tmi_ = PH_NOW(); (void)tmi_;
    (void)pb10; (void)ath_adjust;
#if defined(LHIP_PHASE_PROF) && !defined(LHIP_HOSTSIM)
    __builtin_amdgcn_s_waitcnt(0);           // profiling build only: everything this wave still has in flight (stores of the previous granule)
    PH_MARK(L, PH_DRAIN, tmi_);
#endif
    g.part2_3_length = 0; g.big_values = 0; g.count1 = 0; g.global_gain = 210; g.scalefac_compress = 0;
    g.block_type = block_type;
    g.table_select[0] = g.table_select[1] = g.table_select[2] = 0;
    g.subblock_gain[0] = g.subblock_gain[1] = g.subblock_gain[2] = g.subblock_gain[3] = 0;
    g.region0_count = 0; g.region1_count = 0; g.preflag = 0; g.scalefac_scale = 0; g.count1table_select = 0;
    g.part2_length = 0; g.sfb_lmax = SBPSY_l; g.sfb_smin = SBPSY_s;
    g.psy_lmax = T.sfb21_extra ? SBMAX_l : SBPSY_l;
    g.psymax = g.psy_lmax; g.sfbmax = g.sfb_lmax; g.sfbdivide = 11;
    g.count1bits = 0; g.max_nonzero_coeff = 575; g.xrpow_max = 0; g.firstcut = 64;
    int nsfb;
    if (block_type == SHORT_TYPE) {
        g.sfb_smin = 0; g.sfb_lmax = 0;
        g.psymax = 3 * ((T.sfb21_extra ? SBMAX_s : SBPSY_s));
        g.sfbmax = 3 * SBPSY_s;
        g.sfbdivide = g.sfbmax - 18;
        g.psy_lmax = 0;
        nsfb = 3 * SBMAX_s;
        LHIP_LANE_ONCE(i, 0, nsfb) {
            const int sfb = i / 3, win = i - 3 * sfb;
            const int w = Q.sfb_s[sfb + 1] - Q.sfb_s[sfb];
            L.width[i] = (int16_t)w; L.window[i] = (int16_t)win; L.start[i] = (int16_t)(3 * Q.sfb_s[sfb] + win * w);
        }
        if (lane == 0) L.start[nsfb] = 576;
        q_load_lines(xr_g, 1, lane, L, Q);       // re-ordered: within each short sfb the three windows become consecutive runs
    } else {
        nsfb = SBMAX_l;
        LHIP_LANE_ONCE(i, 0, SBMAX_l) {
            L.width[i] = (int16_t)(Q.sfb_l[i + 1] - Q.sfb_l[i]); L.window[i] = 3; L.start[i] = (int16_t)Q.sfb_l[i];
        }
        if (lane == 0) L.start[SBMAX_l] = 576;
        q_load_lines(xr_g, 0, lane, L, Q);
    }
    LHIP_LANE_ONCE(i, 0, (SFBMAX) + 1) { L.sfw[i] = 0; L.sfb[i] = 0; }
    wave_sync();
    PH_MARK(L, PH_COPY, tmi_);

    // analog silence in the pseudo bands above sfb21 / sfb12: zero trailing lines below the adjusted ATH (thresholds: q_ath_pseudo)
    if (skip_silence) return;
    if (block_type != SHORT_TYPE) {
        const int lo = Q.psfb21[0];
        int top = lo - 1;                       // highest line that is NOT below its threshold
        for (int j = lo + lane; j < 576; j += LHIP_NL) {
            int gsfb = 0;                       // pseudo band of line j: the number of inner borders at or below it
#pragma unroll
            for (int q = 1; q < PSFB21; q++) gsfb += (Q.psfb21[q] <= j);
            if (!(d_abs((double)L.xr[j]) < L.ath_pseudo[gsfb])) top = j;   // ascending j per lane
        }
        top = wave_max(top);
        for (int j = lo + lane; j < 576; j += LHIP_NL) if (j > top) { L.xr[j] = 0; if (xr_wb) xr_wb[j] = 0; }
        PH_MARK(L, PH_PUBLISH, tmi_);
    } else {
        const int s12 = Q.sfb_s[12], w12 = Q.sfb_s[13] - s12;
        for (int block = 0; block < 3; block++) {
            const int lo = s12 * 3 + w12 * block;
            int top = lo - 1;
            for (int j = lo + lane; j < lo + w12; j += LHIP_NL) {
                const int rel = j - lo + Q.psfb12[0];
                int gsfb = 0;
#pragma unroll
                for (int q = 1; q < PSFB12; q++) gsfb += (Q.psfb12[q] <= rel);
                if (!(d_abs((double)L.xr[j]) < L.ath_pseudo[PSFB21 + gsfb])) top = j;
            }
            top = wave_max(top);
            for (int j = lo + lane; j < lo + w12; j += LHIP_NL) if (j > top) { L.xr[j] = 0; if (xr_wb) xr_wb[3 * (s12 + (j - lo)) + block] = 0; }
        }
    }
    wave_sync();
}

// init_xrpow (Quantize.js:92-138); returns 1 if the granule has energy
LHIP_DEV int q_init_xrpow(GI& g, int lane, QuantLds& L, const QuantTabs& Q) {
    lane = fresh_lane(lane);
    float m = 0.f;
    double sum = 0;
    for (int i = lane; i < 576; i += LHIP_NL) {
        const double tmp = d_abs((double)L.xr[i]);
        sum += tmp;
        const float v = (float)d_sqrt(tmp * d_sqrt(tmp));
        L.xrpow[i] = v;
        m = fmax_nonneg(m, v);
    }
    m = wave_maxf_pos(m);
    g.xrpow_max = m;
    // `sum > 1e-20` only separates digital silence from signal; the reduction order cannot change the verdict
    // except within 1e-14 (relative) of the threshold itself
    const int has = wave_sumd(sum) > 1E-20;
    wave_sync();
    return has;
}

// the first band that reaches past max_nonzero_coeff (quantize_xrpow's walk ends inside it, Takehiro.js:212-250); 64 if none does
LHIP_DEV void q_set_firstcut(GI& g, int lane, const QuantLds& L) {
    const int sfbmax = (g.block_type == SHORT_TYPE) ? 38 : 21;
    uint64_t m = 0;
    LHIP_LANE_ONCE(sfb, 0, (sfbmax) + 1) if (L.start[sfb] + L.width[sfb] > g.max_nonzero_coeff) m |= 1ull << sfb;
    m = wave_lane_bits(m);
    g.firstcut = m ? (int)__builtin_ctzll(m) : 64;
}

// Ordered (line order) f64 sums of per-line terms over scalefactor bands, all bands at once, as a systolic fold:
// lane l owns NLN consecutive lines and folds their terms, in order, onto the running sum handed over by lane l-1
// (reset at band starts); ceil(longest band / NLN) + 1 hand-overs reproduce the strictly sequential sums exactly.
// Result: L.nsum[band] for every band whose length is <= maxlen.
enum { NLN_FOLD = 576 / LHIP_NL };
LHIP_DEV void fold_band_sums(double (&tq)[NLN_FOLD], const uint8_t* l2s, int maxlen, int lane, QuantLds& L) {
#if LHIP_NL == 1
    (void)maxlen; (void)lane;
    double sacc = 0.0;
    for (int j = 0; j < 576; j++) {
        if (j == 0 || l2s[j - 1] != l2s[j]) sacc = 0.0;
        sacc += tq[j];
        if (j == 575 || l2s[j + 1] != l2s[j]) L.nsum[l2s[j]] = sacc;
    }
#else
    unsigned endm = 0;
    double keep[NLN_FOLD]; int bnd[NLN_FOLD];
    int prevb = (lane == 0) ? -1 : (int)l2s[NLN_FOLD * lane - 1];
#pragma unroll
    for (int k = 0; k < NLN_FOLD; k++) {
        const int j = NLN_FOLD * lane + k;
        bnd[k] = l2s[j];
        keep[k] = (bnd[k] != prevb) ? 0.0 : 1.0;       // fma(sum, keep, t): `sum + t` or a fresh `t`, one rounding either way
        prevb = bnd[k];
    }
    const int nextb = (lane == LHIP_NL - 1) ? -1 : (int)l2s[NLN_FOLD * (lane + 1)];
#pragma unroll
    for (int k = 0; k < NLN_FOLD; k++) if (bnd[k] != (k + 1 < NLN_FOLD ? bnd[k + 1 < NLN_FOLD ? k + 1 : k] : nextb)) endm |= 1u << k;
    const int nsteps = (maxlen + NLN_FOLD - 1) / NLN_FOLD + 1;
    double carry = 0.0;
    for (int st = 0; st + 1 < nsteps; st++) {
        double sacc = carry;
#pragma unroll
        for (int k = 0; k < NLN_FOLD; k++) sacc = __builtin_fma(sacc, keep[k], tq[k]);
        carry = wave_shr1d(sacc, 0.0);
    }
    {
        double sacc = carry;
#pragma unroll
        for (int k = 0; k < NLN_FOLD; k++) { sacc = __builtin_fma(sacc, keep[k], tq[k]); tq[k] = sacc; }
    }
#pragma unroll
    for (int k = 0; k < NLN_FOLD; k++) if ((endm >> k) & 1u) L.nsum[bnd[k]] = tq[k];
#endif
    wave_sync();
}

// calc_xmin (QuantizePVT.js:569-719), CBR flavour
LHIP_DEV void q_calc_xmin(const Tables& T, double ath_adjust, double masking_lower, const float* ratio /*E layout*/,
                          GI& g, int lane, QuantLds& L, const QuantTabs& Q) {
    lane = fresh_lane(lane);
    // band energies en0 = sum of xr^2 in line order: terms with all lanes busy, then the ordered fold
    {
        double tq[NLN_FOLD];
#pragma unroll
        for (int k = 0; k < NLN_FOLD; k++) { const double x = L.xr[NLN_FOLD * lane + k]; tq[k] = x * x; }
        int maxw = 0;
        const int nb = (g.block_type != SHORT_TYPE) ? g.psy_lmax : 3 * SBPSY_s;
        LHIP_LANE_ONCE(b, 0, nb) if (maxw < L.width[b]) maxw = L.width[b];
        maxw = wave_max(maxw);
        fold_band_sums(tq, line2sfb(Q, g.block_type), maxw, lane, L);
    }
    if (g.block_type != SHORT_TYPE) {
        LHIP_LANE_ONCE(gsfb, 0, g.psy_lmax) {
            double xmin = ath_adjust * (double)T.ATH_l[gsfb];
            const double en0 = L.nsum[gsfb];
            const double en = ratio[E_EN_L + gsfb];
            if (en > 0.0) {
                const double x = en0 * (double)ratio[E_THM_L + gsfb] * masking_lower / en;
                if (xmin < x) xmin = x;
            }
            const float xm = (float)(xmin * (double)T.longfact[gsfb]);
            L.xmin[gsfb] = xm; L.rxmin[gsfb] = recip_for_div((double)xm);
        }
        int t = -1;
        for (int k = lane; k < 576; k += LHIP_NL) if (!((double)L.xr[k] == 0)) t = k;
        t = wave_max(t);
        g.max_nonzero_coeff = (t >= 575) ? 575 : t + 1;
    } else {
        LHIP_LANE_ONCE(sfb, 0, SBPSY_s) {       // psymax/3 bands, 3 windows each
            const double tmpATH = ath_adjust * (double)T.ATH_s[sfb];
            float px[3];
            for (int b = 0; b < 3; b++) {
                const int gs = 3 * sfb + b;
                double xmin = tmpATH;
                const double en0 = L.nsum[gs];
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
            L.rxmin[3 * sfb] = recip_for_div((double)px[0]); L.rxmin[3 * sfb + 1] = recip_for_div((double)px[1]); L.rxmin[3 * sfb + 2] = recip_for_div((double)px[2]);
        }
        g.max_nonzero_coeff = 575;
    }
    q_set_firstcut(g, lane, L);
    wave_sync();
}

// ---------------------------------------------------------------------------------------------
// quantize_xrpow (Takehiro.js:171-314) -> ix ; `use_prev` = the prev_noise cache (pn_* in LDS) is live
// ---------------------------------------------------------------------------------------------
// Lane l owns the PAIRS 2(l + 64 j), +1 (bands start and end on even lines, so a pair never straddles a band):
// xrpow comes in as one 8-byte load and ix goes out as one 32-bit store per pair, and the quantized values stay in
// registers (vx, vy) for the Huffman bit count that follows -- no LDS round trip between the two.
enum { NPL = (288 + LHIP_NL - 1) / LHIP_NL };
// ix_old (candidate helpers, q_cand_helper): where the kept values of cached bands are read from when that is not `ix` itself; `safe`: the record may be
// changing under this wave (a helper working on a request its owner has already left behind): every table index is clamped, whatever the lines hold
LHIP_DEV void q_quantize(const Tables& T, const GI& g, const int32_t* scalefac, int16_t* ix, int use_prev,
                         int pn_gain, int pn_sfb_count1, int (&vx)[NPL], int (&vy)[NPL], int lane, QuantLds& L, const QuantTabs& Q,
                         const int16_t* ix_old = nullptr, int safe = 0) {
    lane = fresh_lane(lane);
    if (!ix_old) ix_old = ix;
    const double istep = ipow20(Q, g.global_gain);
    // every truncated product is <= xrpow_max * istep: below QT_N no lane can need the part of adj43 that is not staged in LDS
    const int may_big = !(g.xrpow_max * istep < (double)QT_N);
    // (safe: where no legitimate value reaches QT_N, a clamp into the staged part of the tables is all that lines changing under this wave need -- it never
    //  changes a legitimate result; where big values are possible the clamp is IXMAX_VAL, the end of the tables in global memory)
    const int clamp_hi = may_big ? (int)IXMAX_VAL : QT_N - 1;
    const int sfbmax = (g.block_type == SHORT_TYPE) ? 38 : 21;
    const int prev_data_use = use_prev && (g.global_gain == pn_gain);
    // per-band decision as wave-uniform bit masks: cached (keep old values) / 0-1 shortcut; the first
    // non-cached band reaching past max_nonzero_coeff (sstar) is quantized partially and ends the walk
    unsigned long long tm_ = PH_NOW(); (void)tm_;
    // the lines first: their LDS latency passes while the band masks are formed
    float xa[NPL], xb[NPL];
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        const int p = 2 * (lane + LHIP_NL * j);
        xa[j] = 0.f; xb[j] = 0.f;
        if (p < 576) { struct F2 { float x, y; }; const F2 xx = *(const F2*)(L.xrpow + p); xa[j] = xx.x; xb[j] = xx.y; }   // 8-byte aligned: p is even
    }
    uint64_t m_cached = 0, m_zo = 0;                     // bit sfb, produced by lane sfb (sfbmax < 64)
    if (use_prev) {                                      // bin-search rounds have no cache and no 0/1 shortcut: nothing to decide
        LHIP_LANE_ONCE(sfb, 0, (sfbmax) + 1) {
            int step = -1;
            const int pstep = L.pn_step[sfb];
            if (prev_data_use || g.block_type == NORM_TYPE) step = sf_step(Q, g, scalefac, L.window, sfb);
            if (prev_data_use && pstep == step) m_cached |= 1ull << sfb;
            else if (pn_sfb_count1 > 0 && sfb >= pn_sfb_count1 && pstep > 0 && step >= pstep) m_zo |= 1ull << sfb;
        }
    }
    m_cached = wave_lane_bits(m_cached); m_zo = wave_lane_bits(m_zo);
    // the bands reaching past max_nonzero_coeff are firstcut .. sfbmax; the walk ends in the first of them that is not cached
    const uint64_t m_cut = (use_prev && g.firstcut <= sfbmax) ? ((~0ull << g.firstcut) & ((2ull << sfbmax) - 1) & ~m_cached) : 0;
    const int sstar = m_cut ? (int)__builtin_ctzll(m_cut) : 99;
    // The reference stops quantizing inside band sstar (at max_nonzero_coeff, rounded to a pair) and fills the rest with
    // zeros.  Every line from max_nonzero_coeff on has xrpow == 0 (it is the first of the spectrum's trailing zeros, or
    // 575 with a full last band) and quantizes to 0 by either formula, and the kept values of cached bands are zero there
    // by induction, so that walk needs no masks here; what remains of it is that the partial band never takes the 0/1
    // shortcut.
    if (sstar <= sfbmax) m_zo &= ~(1ull << sstar);
    PH_MARK(L, PH_Q_MASK, tm_);
    const uint8_t* l2s = line2sfb(Q, g.block_type);
    const int need_old = (m_cached != 0);
    const float istep_f = Q.ipow20[g.global_gain];     // the Float32Array value itself; `istep` above is its f64 image
    // staged, branch-light form: all loads of a stage are independent so they overlap (LDS latency is the cost here); the two
    // truncations of every line are q_floor_prod / q_floor_fma (lhip_math.h)
    int ra[NPL], rb[NPL];
    q_floor_prod(xa, xb, istep_f, ra, rb);                                 // 0 <= x <= 8206: truncation == ToInt32
    if (safe) {
#pragma unroll
        for (int j = 0; j < NPL; j++) { ra[j] = ra[j] < 0 ? 0 : (ra[j] > clamp_hi ? clamp_hi : ra[j]); rb[j] = rb[j] < 0 ? 0 : (rb[j] > clamp_hi ? clamp_hi : rb[j]); }
    }
    float aa[NPL], ab[NPL];
    if (!may_big) {                                                        // every product is below QT_N (see may_big): nothing to clamp
#pragma unroll
        for (int j = 0; j < NPL; j++) { aa[j] = Q.adj43[ra[j]]; ab[j] = Q.adj43[rb[j]]; }
    } else {                                                               // rare: large quantized values
#pragma unroll
        for (int j = 0; j < NPL; j++) { aa[j] = Q.adj43[ra[j] < QT_N ? ra[j] : QT_N - 1]; ab[j] = Q.adj43[rb[j] < QT_N ? rb[j] : QT_N - 1]; }
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            if (ra[j] >= QT_N) aa[j] = T.adj43[ra[j]];      // (safe: clamped above)
            if (rb[j] >= QT_N) ab[j] = T.adj43[rb[j]];
        }
    }
    // no masking at the end of the spectrum: zero xrpow quantizes to (int)(0 + adj43[0]) = 0 (see above)
    q_floor_fma(xa, xb, istep_f, aa, ab, vx, vy);
    if (safe) {
#pragma unroll
        for (int j = 0; j < NPL; j++) { vx[j] = vx[j] < 0 ? 0 : (vx[j] > clamp_hi + 1 ? clamp_hi + 1 : vx[j]); vy[j] = vy[j] < 0 ? 0 : (vy[j] > clamp_hi + 1 ? clamp_hi + 1 : vy[j]); }      // (a product just below QT_N may round up to QT_N: legitimate)
    }
    if (!need_old && m_zo == 0) {
        // the common round (every bin-search round and most others): no cached band, no 0/1 shortcut
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            const int p = 2 * (lane + LHIP_NL * j);
            if (p < 576) *(uint32_t*)(ix + p) = (uint32_t)vx[j] | ((uint32_t)vy[j] << 16);
        }
    } else if (m_zo == 0) {
        // cached bands keep their values, nothing else
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            const int p = 2 * (lane + LHIP_NL * j);
            int sf = 0; uint32_t oldw = 0;
            if (p < 576) { sf = l2s[p]; oldw = *(const uint32_t*)(ix_old + p); }
            const int cached = (int)((m_cached >> sf) & 1);
            const int va = cached ? (int)(oldw & 0xffffu) : vx[j], vb = cached ? (int)(oldw >> 16) : vy[j];
            vx[j] = va; vy[j] = vb;
            if (p < 576) *(uint32_t*)(ix + p) = (uint32_t)va | ((uint32_t)vb << 16);
        }
    } else {
        // 0/1 shortcut (Takehiro.js:187-210): ix = (compareval0 > xrpow) ? 0 : 1 with compareval0 = (1 - 0.4054) / istep in f64.  xrpow is a
        // Float32 value, so the f64 comparison is a Float32 one against zo_thr, the smallest Float32 not below compareval0:
        // compareval0 > x  <=>  x < zo_thr
        const double compareval0 = (1.0 - 0.4054) / istep;
        float zo_thr = (float)compareval0;
        if ((double)zo_thr < compareval0) zo_thr = f32_next_up(zo_thr);
        // the previous values are only fetched when some band is cached
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            const int p = 2 * (lane + LHIP_NL * j);
            int sf = 0; uint32_t oldw = 0;
            if (p < 576) { sf = l2s[p]; if (need_old) oldw = *(const uint32_t*)(ix_old + p); }
            const int cached = (int)((m_cached >> sf) & 1), zo = (int)((m_zo >> sf) & 1);
            int va = vx[j], vb = vy[j];
            if (zo) { va = (xa[j] < zo_thr) ? 0 : 1; vb = (xb[j] < zo_thr) ? 0 : 1; }
            const int oa = (int)(oldw & 0xffffu), ob = (int)(oldw >> 16);
            va = cached ? oa : va;
            vb = cached ? ob : vb;
            vx[j] = va; vy[j] = vb;
            if (p < 576) *(uint32_t*)(ix + p) = (uint32_t)va | ((uint32_t)vb << 16);
        }
    }
    wave_sync();
    PH_MARK(L, PH_Q_LINES, tm_);
    (void)safe;
}

// ---------------------------------------------------------------------------------------------
// Huffman table choice for a set of pairs (Takehiro.js:336-516).  Region description for the one-pass counter.
// ---------------------------------------------------------------------------------------------
struct RegionPlan { int kind, t1, xlen, lb1, lb2, choice, choice2, o0, o1, o2; };   // kind 0 empty/zero, 1 t1, 2 table23/56, 4 triple, 5 ESC, 6 overflow

// ESC table pair for a maximum > 15 (Takehiro.js:479-497): linmax of table t is 2^linbits - 1, so "linmax >= mx"
// is "linbits >= bit length of mx"; linbits of tables 16..23 = 1,2,3,4,6,8,10,13 and of 24..31 = 4,5,6,7,8,9,11,13.
LHIP_DEV void esc_choice(int mx15, int* choice, int* choice2, int* lb1, int* lb2) {
    const int bl = 32 - __builtin_clz((unsigned)mx15);            // 1..13
    const int c24 = bl <= 4 ? 0 : bl <= 9 ? bl - 4 : bl <= 11 ? 6 : 7;
    int f16 = bl <= 4 ? bl - 1 : 4 + ((bl - 5) >> 1);
    if (f16 > 7) f16 = 7;
    int c16 = c24 - 8 + 8;                                         // choice2 - 8 - 16 == c24
    if (c16 < f16) c16 = f16;
    // c16 can reach 8 only if c24 == 8, which cannot happen (c24 <= 7)
    *choice2 = 24 + c24; *choice = 16 + c16;
    *lb1 = (int)((0xDA864321u >> (4 * c16)) & 15u);
    *lb2 = (int)((0xDB987654u >> (4 * c24)) & 15u);
}

// plan of a Huffman region whose largest value is m (Takehiro.js:336-346 choose_table / 479-497 ESC pair), as stored in
// QuantTabs::plan.  plan word: kind | t0 << 3 | t1 << 9 | t2 << 15 | lbA << 21 | lbB << 25
LHIP_DEV int plan_index(int m) { return m <= 15 ? m : (m <= IXMAX_VAL ? 15 + (32 - __builtin_clz((unsigned)(m - 15) | 1u)) : 29); }
LHIP_DEV void q_fill_plans(QuantTabs& Q, int tid, int nthr) {
    for (int e = tid; e < 30; e += nthr) {
        const int m = e <= 15 ? e : (e < 29 ? 15 + (1 << (e - 16)) : IXMAX_VAL + 1);
        // first candidate table for maxima 0..15 (huf_tbl_noESC, Takehiro.js:336-346), one nibble per value
        const int mc = m < 15 ? m : 15;
        const int tn = (int)((mc < 8 ? (0xAA775210u >> (4 * mc)) : (0xDDDDDDDDu >> (4 * (mc - 8)))) & 15u);
        const int kind = (m == 0) ? 0 : (m == 1) ? 1 : (m <= 3) ? 2 : (m <= 15) ? 4 : (m <= IXMAX_VAL) ? 5 : 6;
        int choice, choice2, lb1, lb2;
        esc_choice(m > 15 ? m - 15 : 1, &choice, &choice2, &lb1, &lb2);
        const int esc = (kind == 5);
        int xl = (tn == 1) ? 2 : (tn == 2) ? 3 : (tn == 5) ? 4 : (tn == 7) ? 6 : (tn == 10) ? 8 : 16;
        int oA = hl_off(tn), oB = hl_off(tn + 1), oC = hl_off(tn + 2 > 15 ? 15 : tn + 2);
        if (esc) { oA = HL_EHI; oB = HL_ELO; oC = HL_EHI; xl = 16; }
        if (kind == 0 || kind == 6) { oA = oB = oC = 0; xl = 0; }
        const int t0 = esc ? choice : tn, t1 = esc ? choice2 : tn + 1, t2 = tn + 2;
        Q.plan[e].d0 = (uint32_t)oA | ((uint32_t)oB << 16);
        Q.plan[e].d1 = (uint32_t)oC | ((uint32_t)xl << 16);
        Q.plan[e].pw = (uint32_t)kind | ((uint32_t)t0 << 3) | ((uint32_t)t1 << 9) | ((uint32_t)t2 << 15) |
                       ((uint32_t)(esc ? lb1 : 0) << 21) | ((uint32_t)(esc ? lb2 : 0) << 25);
    }
}

LHIP_DEV RegionPlan plan_region_(const QuantTabs& Q, int mx) {
    (void)Q;
    RegionPlan r; r.kind = 0; r.t1 = 0; r.xlen = 0; r.lb1 = r.lb2 = 0; r.choice = r.choice2 = 0; r.o0 = r.o1 = r.o2 = 0;
    if (mx == 0) return r;
    if (mx == 1) { r.kind = 1; r.t1 = 1; r.xlen = 2; r.o0 = HL_T1; return r; }
    if (mx <= 3) { r.kind = 2; r.t1 = (mx == 2) ? 2 : 5; r.xlen = (mx == 2) ? 3 : 4; return r; }
    if (mx <= 15) {
        r.kind = 4;
        if (mx <= 5) { r.t1 = 7; r.xlen = 6; r.o0 = HL_T7; r.o1 = HL_T8; r.o2 = HL_T9; }
        else if (mx <= 7) { r.t1 = 10; r.xlen = 8; r.o0 = HL_T10; r.o1 = HL_T11; r.o2 = HL_T12; }
        else { r.t1 = 13; r.xlen = 16; r.o0 = HL_T13; r.o1 = HL_T14; r.o2 = HL_T15; }
        return r;
    }
    if (mx > IXMAX_VAL) { r.kind = 6; return r; }
    r.kind = 5;
    esc_choice(mx - 15, &r.choice, &r.choice2, &r.lb1, &r.lb2);
    return r;
}

// contribution of one pair to the (up to three) length sums of its region
LHIP_DEV void pair_bits(const QuantTabs& Q, const RegionPlan& r, int x, int y, int& s0, int& s1, int& s2) {
    switch (r.kind) {
        case 1: s0 += Q.hlen[r.o0 + x * 2 + y]; break;
        case 2: { const int q = (r.t1 == 2) ? x * 3 + y : x * 4 + y, oa = (r.t1 == 2) ? HL_T2 : HL_T5, ob = (r.t1 == 2) ? HL_T3 : HL_T6;
                  s0 += ((int)Q.hlen[oa + q] << 16) | (int)Q.hlen[ob + q]; } break;      // packed t1 | t1 + 1 as count_bit_noESC_from2 does
        case 4: { const int q = x * r.xlen + y; s0 += Q.hlen[r.o0 + q]; s1 += Q.hlen[r.o1 + q]; s2 += Q.hlen[r.o2 + q]; } break;
        case 5: {
            int n = 0;
            if (x != 0) { if (x > 14) { x = 15; n++; } x *= 16; }
            if (y != 0) { if (y > 14) { y = 15; n++; } x += y; }
            s0 += (int)Q.hlen[HL_EHI + x] + n * r.lb1; s1 += (int)Q.hlen[HL_ELO + x] + n * r.lb2;
        } break;
        default: break;
    }
}

// final table + bits from the wave-reduced sums (same tie-breaking as count_bit_* in the reference)
LHIP_DEV int finish_region(const RegionPlan& r, int s0, int s1, int s2, int* bits) {
    switch (r.kind) {
        case 0: return 0;
        case 1: *bits += s0; return 1;
        case 2: { int t1 = r.t1, sum2 = s0 & 0xffff, sum = s0 >> 16; if (sum > sum2) { sum = sum2; t1++; } *bits += sum; return t1; }
        case 4: { int t = r.t1; if (s0 > s1) { s0 = s1; t++; } if (s0 > s2) { s0 = s2; t = r.t1 + 2; } *bits += s0; return t; }
        case 5: { int c = r.choice; if (s0 > s1) { s0 = s1; c = r.choice2; } *bits += s0; return c; }
        default: *bits = LARGE_BITS; return -1;
    }
}

// choose_table over pairs [a, b): cooperative; adds to *bits, returns table
LHIP_DEV int q_choose_table(const Tables& T, const int16_t* ix, int a, int b, int* bits, int lane, const QuantLds& L, const QuantTabs& Q) {
    int mx = 0;
    for (int p = a + 2 * lane; p < b; p += 2 * LHIP_NL) { const int x1 = ix[p], x2 = ix[p + 1]; if (mx < x1) mx = x1; if (mx < x2) mx = x2; }
    mx = wave_max(mx);
    const RegionPlan r = plan_region_(Q, mx);
    int s0 = 0, s1 = 0, s2 = 0;
    if (r.kind >= 1 && r.kind <= 5)
        for (int p = a + 2 * lane; p < b; p += 2 * LHIP_NL) pair_bits(Q, r, ix[p], ix[p + 1], s0, s1, s2);
    s0 = wave_sum(s0);
    if (r.kind >= 4) s1 = wave_sum(s1);
    if (r.kind == 4) s2 = wave_sum(s2);
    return finish_region(r, s0, s1, s2, bits);
}

// Conditionally assigned GrInfo fields (Takehiro.js:566-612: table_select[r] only for a non-empty region, the region
// counts only for long blocks with big_values > 0).  Their values after a bin search depend on the gains it visited, so
// they are part of what the seed-chain validation must reproduce.  mask bits: 1/2/4 table_select[0/1/2], 8 region counts.
LHIP_DEV int pack_cond_fields(const GI& g, int mask) {
    return mask | ((g.table_select[0] + 1) << 4) | ((g.table_select[1] + 1) << 10) | ((g.table_select[2] + 1) << 16) |
           (g.region0_count << 22) | (g.region1_count << 26);
}
LHIP_DEV int apply_cond_fields(int state, int asg) {
    if (asg & 1) state = (state & ~(63 << 4)) | (asg & (63 << 4));
    if (asg & 2) state = (state & ~(63 << 10)) | (asg & (63 << 10));
    if (asg & 4) state = (state & ~(63 << 16)) | (asg & (63 << 16));
    if (asg & 8) state = (state & ~(0x3ff << 22)) | (asg & (0x3ff << 22));
    return state & ~15;
}   // scalar part of CalcNoiseData (arrays are L.pn_*)

// noquant_count_bits (Takehiro.js:521-628); updates g, returns bits.  pn_sfb_count1 as in/out.
// *asg_mask: which conditionally assigned fields this call wrote (see pack_cond_fields).
LHIP_DEV int q_noquant_count_bits(const Tables& T, GI& g, const int16_t* ix, int (&vx)[NPL], int (&vy)[NPL], int use_prev, int* pn_sfb_count1, int* asg_mask, int lane, QuantLds& L, const QuantTabs& Q) {
    *asg_mask = 0;
    lane = fresh_lane(lane);
    unsigned long long tm_ = PH_NOW(); (void)tm_;
    int i = ((g.max_nonzero_coeff + 2) >> 1) << 1;
    if (i > 576) i = 576;
    if (use_prev) *pn_sfb_count1 = 0;
    // the lane's pairs arrive in registers from q_quantize; pairs at or above the max_nonzero_coeff bound are zero already
    // (xr, hence xrpow, is zero from max_nonzero_coeff on -- see q_quantize)
    // count1 boundary (highest pair with a non-zero value) and the end of the big-values region (the quad scan of
    // Takehiro.js:540-560 stops at the first quad, counted from the top, that holds a value > 1).
    int firstbig;
#if LHIP_NL == 1
    {
        int top = 0;
        for (int j = 0; j < NPL; j++) if ((vx[j] | vy[j]) != 0) top = 2 * (lane + LHIP_NL * j) + 2;
        i = top;
        const int nq = i >> 2;
        firstbig = nq;
        for (int k = 0; k < nq; k++) {
            const int e = i - 4 * k;
            if (((ix[e - 1] | ix[e - 2] | ix[e - 3] | ix[e - 4]) & 0x7fff) > 1) { firstbig = k; break; }
        }
    }
#else
    {
        // Pair lane + 64 j is bit `lane` of ballot j, so the two boundaries are "highest set bit" questions
        // answered on the scalar unit: no reduction chain, no second pass over the spectrum.
        int tp = -1, bp = -1;                    // highest non-zero pair / highest pair with a value > 1
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            const uint64_t nz = wave_ballot((vx[j] | vy[j]) != 0), bg = wave_ballot((vx[j] | vy[j]) > 1);
            if (nz) tp = 64 * j + 63 - (int)__builtin_clzll(nz);
            if (bg) bp = 64 * j + 63 - (int)__builtin_clzll(bg);
        }
        i = 2 * tp + 2;
        const int P = tp + 1, nq = i >> 2;
        firstbig = (P - 1 - bp) >> 1;
        if (firstbig > nq) firstbig = nq;
    }
#endif
    g.count1 = i;
    PH_MARK(L, PH_C_LOAD, tm_);
    int a12 = 0;
    for (int k = lane; k < firstbig; k += LHIP_NL) {
        const int e = i - 4 * k;                 // even: the two pairs of the quad as two aligned 32-bit words; values are 0/1 here
        const uint32_t w0 = *(const uint32_t*)(ix + e - 4), w1 = *(const uint32_t*)(ix + e - 2);
        const int p = (int)(((w0 & 1u) << 3) | ((w0 >> 16) << 2) | ((w1 & 1u) << 1) | (w1 >> 16));
        a12 += Q.t32l[p] + (Q.t33l[p] << 16);
    }
    // the quads' two candidate lengths (packed) are reduced together with the region sums below: one reduction chain less per call
    i -= 4 * firstbig;
    g.big_values = i;
    int a1, a2, bits;
#define COUNT1_FINISH(A12) do { const int c1_ = (A12) & 0xffff, c2_ = (int)((unsigned)(A12) >> 16); bits = c1_; g.count1table_select = 0; \
                                if (c1_ > c2_) { bits = c2_; g.count1table_select = 1; } g.count1bits = bits; } while (0)
    PH_MARK(L, PH_C_QUADS, tm_);
    if (i == 0) { const int t = wave_sum(a12); COUNT1_FINISH(t); return bits; }
    int use2 = 0, cnt1_norm = 0;
    if (g.block_type == SHORT_TYPE) {
        a1 = 3 * Q.sfb_s[3];
        if (a1 > g.big_values) a1 = g.big_values;
        a2 = g.big_values;
    } else if (g.block_type == NORM_TYPE) {
        const uint32_t bt = Q.bvtab[i >> 1];               // one look-up: region counts, region borders, sfb_count1
        g.region0_count = (int)((bt >> 20) & 15u); g.region1_count = (int)((bt >> 24) & 7u);
        a1 = (int)(bt & 1023u); a2 = (int)((bt >> 10) & 1023u);
        cnt1_norm = (int)(bt >> 27);
        if (a2 < i) use2 = 1;
    } else {
        g.region0_count = 7;
        g.region1_count = SBMAX_l - 1 - 7 - 1;
        a1 = Q.sfb_l[7 + 1];
        a2 = i;
        if (a1 > a2) a1 = a2;
    }
    if (a1 > i) a1 = i;
    if (a2 > i) a2 = i;
    *asg_mask = (g.block_type != SHORT_TYPE ? 8 : 0) | (use2 ? 4 : 0) | (0 < a1 ? 1 : 0) | (a1 < a2 ? 2 : 0);
    // region maxima: region of a pair = number of boundaries at or below it
    int m0 = 0, m1 = 0, m2 = 0;
    int rj[NPL];                                                    // region of the lane's pairs (3: at or beyond big_values), kept for the length sums
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        rj[j] = 3;
        if (2 * LHIP_NL * j >= i) continue;                         // wave-uniform: every pair of this round of lanes lies beyond big_values
        const int p = 2 * (lane + LHIP_NL * j);
        const int r = (p >= a1) + (p >= a2) + (p >= i);             // a1 <= a2 <= i
        const int m = vx[j] > vy[j] ? vx[j] : vy[j];
        const int mr0 = (r == 0) ? m : 0, mr1 = (r == 1) ? m : 0, mr2 = (r == 2) ? m : 0;
        m0 = m0 > mr0 ? m0 : mr0; m1 = m1 > mr1 ? m1 : mr1; m2 = m2 > mr2 ? m2 : mr2;
        rj[j] = r;
    }
    { int mm[3] = {m0, m1, m2}; wave_max_n(mm); m0 = mm[0]; m1 = mm[1]; m2 = mm[2]; }
    // Data-driven length sums: every region publishes the pool offsets of its (up to three) candidate tables and
    // its row stride; a pair then costs three byte gathers whatever its region's table group is, and all regions
    // are handled in ONE pass.  Region r is planned (and later finished) by lane r, branch-free: there is no
    // scalar per-region control flow left.
    struct LanePlan { int kind, t0, t1, t2, lbA, lbB; };
    // one plan per lane on the device (a scalar struct: an array indexed by r / 64 would live in scratch memory and cost a
    // memory round trip per access); the one-lane host simulation holds all three
#if LHIP_NL == 1
    LanePlan lp[3]; int pv[3];
#define LP_(r) lp[r]
#define PV_(r) pv[r]
#else
    LanePlan lp1; int pv1 = 0;
    lp1.kind = 0; lp1.t0 = lp1.t1 = lp1.t2 = lp1.lbA = lp1.lbB = 0;
#define LP_(r) lp1
#define PV_(r) pv1
#endif
#ifndef LHIP_EXP_UNCOND
#define LHIP_EXP_UNCOND 0      /* 1 (A/B builds, round 6): no exec-masked code around per-pair / per-line work in the hot loops -- every lane works, lanes
                                  without work write to a slot nobody reads (profiles/r06_ab_gquant_scalar_side.txt) */
#endif
    LHIP_LANE_ONCE(r, 0, LHIP_EXP_UNCOND ? 4 : 3) {
        const int m = (r == 0) ? m0 : (r == 1) ? m1 : (r == 2) ? m2 : 0;       // (region "3" = beyond big_values: the empty plan, all-zero descriptor)
        const QuantTabs::PlanEnt pe = Q.plan[plan_index(m)];          // the whole plan is a function of the maximum: one look-up
        L.rdesc[r][0] = pe.d0;
        L.rdesc[r][1] = pe.d1;
        LanePlan& q = LP_(r);
        q.kind = (int)(pe.pw & 7u); q.t0 = (int)((pe.pw >> 3) & 63u); q.t1 = (int)((pe.pw >> 9) & 63u); q.t2 = (int)((pe.pw >> 15) & 63u);
        q.lbA = (int)((pe.pw >> 21) & 15u); q.lbB = (int)((pe.pw >> 25) & 15u);
    }
    wave_sync();
    PH_MARK(L, PH_C_MAX, tm_);
    // field width: a lane owns at most NPL pairs (5 x 21 bits < 2^10); the one-lane host simulation owns all 288
#if LHIP_NL == 1
    typedef uint64_t acc_t; enum { FB = 21 };
#else
    typedef uint32_t acc_t; enum { FB = 10 };
#endif
    const acc_t FM = ((acc_t)1 << FB) - 1;
    acc_t accA = 0, accB = 0, accC = 0, accN = 0;
    const int any_esc = (m0 > 15) | (m1 > 15) | (m2 > 15);          // escaped values only cost extra bits in ESC regions
#if LHIP_NL == 1
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        const int p = 2 * (lane + LHIP_NL * j);
        if (p < i) {
            const int r = (p >= a1) + (p >= a2);
            const uint64_t d = *(const uint64_t*)L.rdesc[r];
            const uint32_t d0 = (uint32_t)d, d1 = (uint32_t)(d >> 32);
            const int x = vx[j], y = vy[j];
            const int idx = (x < 15 ? x : 15) * (int)(d1 >> 16) + (y < 15 ? y : 15);
            const int sh = FB * r;
            accA += (acc_t)Q.hlen[(d0 & 0xffffu) + idx] << sh;
            accB += (acc_t)Q.hlen[(d0 >> 16) + idx] << sh;
            accC += (acc_t)Q.hlen[(d1 & 0xffffu) + idx] << sh;
            if (any_esc) accN += (acc_t)((x > 14) + (y > 14)) << sh;
        }
    }
    (void)rj;
#else
    // a pair at a time (descriptor of its region, three byte gathers, three multiply-adds): measured 1 % faster per launch than the
    // same work in stages (all descriptors, all gathers, all sums), which keeps fifteen more values live
    if (any_esc) {
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            if (LHIP_EXP_UNCOND || rj[j] < 3) {
                const int r = rj[j];
                const uint64_t d = *(const uint64_t*)L.rdesc[r];
                const uint32_t d0 = (uint32_t)d, d1 = (uint32_t)(d >> 32);
                const int x = vx[j], y = vy[j];
                const int idx = (x < 15 ? x : 15) * (int)(d1 >> 16) + (y < 15 ? y : 15);
                const unsigned mult = 1u << (FB * r);                   // field of region r; 24-bit multiply-add accumulates in one instruction
                accA = mul24((unsigned)Q.hlen[(d0 & 0xffffu) + idx], mult) + accA;
                accB = mul24((unsigned)Q.hlen[(d0 >> 16) + idx], mult) + accB;
                accC = mul24((unsigned)Q.hlen[(d1 & 0xffffu) + idx], mult) + accC;
                accN = mul24((unsigned)((x > 14) + (y > 14)), mult) + accN;
            }
        }
    } else {
        // no region holds a value above 15 (three calls in four at stereo 128 kbps): nothing to clamp, no escapes to count
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            if (LHIP_EXP_UNCOND || rj[j] < 3) {
                const int r = rj[j];
                const uint64_t d = *(const uint64_t*)L.rdesc[r];
                const uint32_t d0 = (uint32_t)d, d1 = (uint32_t)(d >> 32);
                const int idx = vx[j] * (int)(d1 >> 16) + vy[j];
                const unsigned mult = 1u << (FB * r);
                accA = mul24((unsigned)Q.hlen[(d0 & 0xffffu) + idx], mult) + accA;
                accB = mul24((unsigned)Q.hlen[(d0 >> 16) + idx], mult) + accB;
                accC = mul24((unsigned)Q.hlen[(d1 & 0xffffu) + idx], mult) + accC;
            }
        }
    }
#endif
    // unpack to (A|B<<16), (C|N<<16) per region: wave totals stay below 2^16 (<= 288 pairs x 21 bits = 6048)
#define FLD(A, R) ((uint32_t)(((A) >> (FB * (R))) & FM))
    int qq[7] = {(int)(FLD(accA, 0) | (FLD(accB, 0) << 16)), (int)(FLD(accC, 0) | (FLD(accN, 0) << 16)), (int)(FLD(accA, 1) | (FLD(accB, 1) << 16)),
                 (int)(FLD(accC, 1) | (FLD(accN, 1) << 16)), (int)(FLD(accA, 2) | (FLD(accB, 2) << 16)), (int)(FLD(accC, 2) | (FLD(accN, 2) << 16)), a12};
    wave_sum_n(qq);                       // six packed region sums and the count1 quads' pair, reduced side by side
    const int q0 = qq[0], q1 = qq[1], q2 = qq[2], q3 = qq[3], q4 = qq[4], q5 = qq[5];
    COUNT1_FINISH(qq[6]);
#undef COUNT1_FINISH
#undef FLD
    PH_MARK(L, PH_C_SUMS, tm_);
    // finish (Takehiro.js count_bit_noESC / _from2 / _from3 / count_bit_ESC tie-breaking): lane r picks the cheapest
    // admissible candidate of region r; result packed as table | overflow << 6 | bits << 8
    LHIP_LANE_ONCE(r, 0, 3) {
        const LanePlan& q = LP_(r);
        const int qa = (r == 0) ? q0 : (r == 1) ? q2 : q4, qc = (r == 0) ? q1 : (r == 1) ? q3 : q5;
        const int n = (int)((unsigned)qc >> 16);
        const int c0 = (qa & 0xffff) + n * q.lbA, c1 = (int)((unsigned)qa >> 16) + n * q.lbB, c2 = qc & 0xffff;
        int b = c0, t = q.t0;
        if ((q.kind == 2 || q.kind == 4 || q.kind == 5) && b > c1) { b = c1; t = q.t1; }
        if (q.kind == 4 && b > c2) { b = c2; t = q.t2; }
        if (q.kind == 0) { b = 0; t = 0; }
        if (q.kind == 6) { b = 0; t = 63; }              // table_select := -1 (never emitted: overflow cannot pass count_bits)
        PV_(r) = t | ((q.kind == 6) << 6) | (b << 8);
    }
#if LHIP_NL == 1
    const int pv0 = pv[0], pv1_ = pv[1], pv2 = pv[2];
#else
    const int pv0 = wave_bcast(pv1, 0), pv1_ = wave_bcast(pv1, 1), pv2 = wave_bcast(pv1, 2);
#endif
    // the reference evaluates region 2 first (NORM only), then 0, then 1; an overflowing region *sets* bits
#define APPLY(PV, SLOT) do { const int t_ = (PV) & 63; if ((PV) & 64) bits = LARGE_BITS; else bits += (PV) >> 8; g.table_select[SLOT] = (t_ == 63) ? -1 : t_; } while (0)
    if (use2) APPLY(pv2, 2);
    if (0 < a1) APPLY(pv0, 0);
    if (a1 < a2) APPLY(pv1_, 1);
#undef APPLY
#undef LP_
#undef PV_
    // first band whose start is >= big_values (PrevNoise.sfb_count1): one table look-up instead of a walk
    if (use_prev && g.block_type == NORM_TYPE) *pn_sfb_count1 = cnt1_norm;       // big_values > 0 here (the table's entry)
    PH_MARK(L, PH_C_FIN, tm_);
    return bits;
}

struct PrevNoise { int gain, sfb_count1; };

// count_bits (Takehiro.js:630-660)
// `use_pn` selects the prev_noise cache; pn is passed by reference with a flag (never as a nullable pointer to a
// local: a select between private addresses is a per-lane value for the compiler and makes the control flow divergent)
LHIP_DEV int q_count_bits(const Tables& T, GI& g, const int32_t* scalefac, int16_t* ix, int use_pn, PrevNoise& pn, int* asg, int lane, QuantLds& L, const QuantTabs& Q) {
    *asg = 0;
    {   // Takehiro.js:635-638: `xrpow_max > IXMAX_VAL / IPOW20(gain)`.  The division is only needed when the product is
        // within rounding of the bound: fl(x * ip) <= 8206 (1 - 2^-50) implies x < fl(8206 / ip)
        const double ip = ipow20(Q, g.global_gain);
        if (g.xrpow_max * ip > (double)IXMAX_VAL * (1.0 - 0x1p-50)) {
            const double w = (double)IXMAX_VAL / ip;
            if (g.xrpow_max > w) return LARGE_BITS;
        }
    }
    int vx[NPL], vy[NPL];
    { PH_BEGIN(); q_quantize(T, g, scalefac, ix, use_pn, use_pn ? pn.gain : 0, use_pn ? pn.sfb_count1 : 0, vx, vy, lane, L, Q); PH_END(L, PH_QUANTIZE); }
    int cnt1 = pn.sfb_count1;
    PH_BEGIN();
    int amask = 0;
    const int r = q_noquant_count_bits(T, g, ix, vx, vy, use_pn, &cnt1, &amask, lane, L, Q);
    *asg = pack_cond_fields(g, amask);
    if (use_pn) pn.sfb_count1 = cnt1;
    PH_END(L, PH_COUNT);
    return r;
}

// ---------------------------------------------------------------------------------------------
// calc_noise (QuantizePVT.js:784-878); distort -> L.distort, cache -> L.pn_*
// The bands' sums in the reference's line order (f64 sums are order-sensitive) as a systolic fold; all gathers hit LDS.
// ---------------------------------------------------------------------------------------------
// need_max: the caller will look at max_noise even if some band is over its threshold (quant_compare only reads max_noise of
// results with over_count == 0, and of `best` only while best.over_count == 0), so the f64 wave maximum is skipped otherwise
// What a calc_noise call would write into the noise cache (PrevNoise / L.pn_*), held by lane = band: a call made SPECULATIVELY (count_bits
// of the same quantization still running on another wave, q_count_bits_piped) must not touch the cache before its evaluation is known to
// be the one the reference makes -- q_noise_commit writes it then; a discarded call leaves no trace the reference could see.
struct NoiseCommit { int fresh, step, cls; float dist; double x; };
LHIP_DEV void q_noise_commit(const GI& g, const NoiseCommit& nc, PrevNoise& pn, int lane, QuantLds& L) {
    lane = fresh_lane(lane);
    LHIP_LANE_ONCE(sfb, 0, g.psymax) {
        if (nc.fresh) { L.pn_step[sfb] = nc.step; L.pn_dist[sfb] = nc.dist; L.pn_x[sfb] = nc.x; L.pn_cls[sfb] = (int16_t)nc.cls; }
    }
    pn.gain = g.global_gain;
    wave_sync();
}
// Sp (candidate helpers): the record that receives what the call WRITES (bstep, nsum, distort) when `L` -- the spectrum, the thresholds, the noise cache --
// belongs to another wave; `safe`: see q_quantize
LHIP_DEV void q_calc_noise_(const Tables& T, const GI& g, const int32_t* scalefac, const int16_t* ix, NoiseRes* res,
                           int use_pn, PrevNoise& pn, int need_max, int lane, QuantLds& L, const QuantTabs& Q, NoiseCommit* defer = nullptr,
                           QuantLds* Sp = nullptr, int safe = 0) {
    QuantLds& S = Sp ? *Sp : L;
    lane = fresh_lane(lane);
    unsigned long long tm_ = PH_NOW(); (void)tm_;
    // 1) per band: its step and whether the cache answers for it.  The reference walks a start line from band to band and sums the
    //    first band that reaches past max_nonzero_coeff over its useful part only, every later band over nothing
    //    (QuantizePVT.js:806-830).  Summing every evaluated band over its WHOLE width gives the same sums: from
    //    max_nonzero_coeff on xr is zero (it is the first of the spectrum's trailing zeros) and so is the quantized value
    //    ((int)(0 * istep + adj43[0]) = 0, cached values by induction), hence every skipped line would add (|0| - pow43[0] * step)^2 = +0
    //    to a non-negative sum -- so there is no range bookkeeping here at all.
    // One error formula serves the three branches of calc_noise_core (QuantizePVT.js:725-767): a band that starts above count1 holds
    // only zeros and one that starts above big_values only 0/1, and pow43[0] = 0, pow43[1] = 1, so |xr| - pow43[ix] * step is bit for
    // bit `xr` (squared), `|xr| - ix01[ix]` and `|xr| - pow43[ix] * step` there.
    const int is_short = g.block_type == SHORT_TYPE;
    // quantized values are <= xrpow_max * ipow20(gain) + 1 (the working copy was quantized at this gain): below QT_N - 1
    // no line can need the part of pow43 that is not staged in LDS
    // (safe: the lines are the helper's own, clamped by its quantization to QT_N - 1 unless big values are possible, in which case this test says so too)
    const int may_big = !(g.xrpow_max * ipow20(Q, g.global_gain) < (double)(QT_N - 1));
#if LHIP_NL == 1
    // one-lane build: the same band by band (a band next to a step of the class function takes the logarithm by itself; where the
    // shortcut is taken it equals the logarithm's verdict, so the wave-wide fallback of the 64-lane program gives the same numbers)
    const uint8_t* l2s = line2sfb(Q, g.block_type);
    int over = 0, ssd = 0;
    uint64_t m_fresh = 0;
    (void)may_big; (void)is_short; (void)S;
    for (int sfb = 0; sfb < g.psymax; sfb++) {
        const int s = sf_step(Q, g, scalefac, L.window, sfb);
        if (!(use_pn && L.pn_step[sfb] == s)) m_fresh |= 1ull << sfb;
        L.qmode[sfb] = s;
        L.bstep[sfb] = Q.pow20[s + Q_MAX2];
    }
    PH_MARK(L, PH_N_WALK, tm_);
    {
        double sacc = 0.0;
        for (int j = 0; j < 576; j++) {
            const int bnd = l2s[j];
            if (j == 0 || l2s[j - 1] != bnd) sacc = 0.0;
            const double x = __builtin_fma(-pow43v(T, Q, ix[j]), (double)L.bstep[bnd < SFBMAX + 1 ? bnd : SFBMAX], d_abs((double)L.xr[j]));
            sacc += x * x;
            if (j == 575 || l2s[j + 1] != bnd) L.nsum[bnd] = sacc;
        }
    }
    PH_MARK(L, PH_N_FOLD, tm_);
    for (int sfb = 0; sfb < g.psymax; sfb++) {
        int cls;
        if (!((m_fresh >> sfb) & 1)) { L.distort[sfb] = L.pn_dist[sfb]; cls = L.pn_cls[sfb]; }
        else {
            const double noise = L.nsum[sfb], b = (double)L.xmin[sfb], rb = L.rxmin[sfb];
            const double x = rb != 0.0 ? div_by_f32(noise, b, rb) : noise / b;
            L.distort[sfb] = (float)x;
            if (use_pn) { L.pn_step[sfb] = L.qmode[sfb]; const double nf = (double)(float)noise; L.pn_dist[sfb] = (float)(rb != 0.0 ? div_by_f32(nf, b, rb) : nf / b); }
            cls = need_max ? -1 : noise_class(x);          // the logarithm will be formed anyway: no shortcut
            int cls_cache = cls;
            if (cls < 0) {
                const double l = v8_log10_pos(x > 1E-20 ? x : 1E-20);
                cls = noise_class_of_log(l); cls_cache = noise_class_of_log((double)(float)l);
            }
            if (use_pn) { L.pn_x[sfb] = x; L.pn_cls[sfb] = (int16_t)cls_cache; }
            else L.nsum[sfb] = x;                          // (no cache in this call: the value for the max_noise pass below)
        }
        if (cls > 0) { ssd += cls * cls; over++; }
    }
    PH_MARK(L, PH_N_TERMS, tm_);
    if (use_pn) pn.gain = g.global_gain;
    res->over_count = over; res->over_SSD = ssd;
    res->max_noise = 0.0;
    if (need_max || over == 0) {
        double max_noise = -20.0;
        for (int sfb = 0; sfb < g.psymax; sfb++) {
            const bool fresh = (m_fresh >> sfb) & 1;
            const double x = (fresh && !use_pn) ? L.nsum[sfb] : L.pn_x[sfb];
            const double l = v8_log10_pos(x > 1E-20 ? x : 1E-20);
            const double nl = fresh ? l : (double)(float)l;
            if (nl > max_noise) max_noise = nl;
        }
        res->max_noise = max_noise;
    }
#else
    int my_step = 0, my_fresh = 0;                        // lane = band: kept in registers for the per-band part after the fold
    LHIP_LANE_ONCE(sfb, 0, g.psymax) {
        my_step = sf_step(Q, g, scalefac, L.window, sfb);
        my_fresh = !(use_pn && L.pn_step[sfb] == my_step);
        S.bstep[sfb] = Q.pow20[my_step + Q_MAX2];
    }
    // the longest chain the fold must carry: the widest evaluated band (an upper bound -- the widest band up to the highest evaluated one)
    const uint64_t m_fresh = wave_ballot(my_fresh);
    const int hib = m_fresh ? 63 - (int)__builtin_clzll(m_fresh) : 0;
    const int maxlen = m_fresh ? (int)(is_short ? Q.wpre_short[hib] : Q.wpre_long[hib]) : 0;
    wave_sync();
    PH_MARK(L, PH_N_WALK, tm_);
    // 2) squared errors summed per band in the reference's line order (f64 sums are order-sensitive) as a
    //    systolic fold: lane l owns NLN consecutive lines and folds their terms, in order, onto the running sum
    //    handed over by lane l-1 (reset at band starts).  One hand-over step extends every band's chain by one
    //    lane, so ceil(longest band / NLN) + 1 steps reproduce the strictly sequential sums exactly, while the
    //    terms themselves are computed once, all lanes busy.
    enum { NLN = 576 / LHIP_NL };
    static_assert(NLN == 9, "QuantTabs::fold_marks is laid out for 9 lines per lane");
    {
        double tq[NLN], keep[NLN];
        int lastb[NLN];
        const uint32_t marks = Q.fold_marks[is_short][lane];
        const uint8_t* l2s = line2sfb(Q, g.block_type);
        // Terms are computed for EVERY line, evaluated band or not: a cached band's sum is never read.
        // keep[k] = 0 at a band start, 1 elsewhere: fma(sum, keep, t) is `sum + t` or `t` with ONE rounding, i.e.
        // exactly the reference's `noise += t` (or a fresh sum); no contraction is involved, the fma is explicit.
        // |xr| - pow43 * step: the product of two Float32 values is exact in f64, so the explicit fma's single rounding is the
        // reference's (a product, then a difference).
        // !may_big (the usual case): every quantized value is below QT_N (see above), nothing to clamp; else the clamped look-up is
        // replaced in the pass that follows
#define NOISE_TERMS(IDX) _Pragma("unroll") for (int k = 0; k < NLN; k++) { \
            const int j = NLN * lane + k; \
            const int bnd = l2s[j]; \
            const float bstep = S.bstep[bnd]; \
            const float xa = L.xr[j]; const int iv = ix[j]; \
            const float pw = Q.pow43[IDX]; \
            const double x = __builtin_fma(-(double)pw, (double)bstep, d_abs((double)xa)); \
            tq[k] = x * x; \
            keep[k] = one_unless_bit(marks, k); \
            lastb[k] = bnd; \
        }
        if (!may_big) { NOISE_TERMS(iv) } else { NOISE_TERMS(iv < QT_N ? iv : QT_N - 1) }
#undef NOISE_TERMS
        if (may_big) {                                   // rare: quantized values beyond the part of pow43 staged in LDS
#pragma unroll
            for (int k = 0; k < NLN; k++) {
                const int j = NLN * lane + k;
                const int iv = ix[j];
                if (iv >= QT_N) {
                    const double x = __builtin_fma(-(double)T.pow43[(!safe || iv <= IXMAX_VAL) ? iv : IXMAX_VAL], (double)S.bstep[lastb[k]], d_abs((double)L.xr[j]));
                    tq[k] = x * x;
                }
            }
        }
        PH_MARK(L, PH_N_LINES, tm_);
        const int nsteps = (maxlen + NLN - 1) / NLN + 1;
        double carry = 0.0;
        for (int st = 0; st + 1 < nsteps; st++) {
            double sacc = carry;
#pragma unroll
            for (int k = 0; k < NLN; k++) sacc = __builtin_fma(sacc, keep[k], tq[k]);
            carry = wave_shr1d(sacc, 0.0);
        }
        {
            double sacc = carry;
#pragma unroll
            for (int k = 0; k < NLN; k++) { sacc = __builtin_fma(sacc, keep[k], tq[k]); tq[k] = sacc; }   // tq := running sums of the last step
        }
#pragma unroll
#if LHIP_EXP_UNCOND
        for (int k = 0; k < NLN; k++) S.nsum[((marks >> (16 + k)) & 1u) ? lastb[k] : SFBMAX] = tq[k];      // nsum[SFBMAX]: no band
#else
        for (int k = 0; k < NLN; k++) if ((marks >> (16 + k)) & 1u) S.nsum[lastb[k]] = tq[k];
#endif
        wave_sync();
        PH_MARK(L, PH_N_FOLD, tm_);
    }
    // 3) per band: distortion ratio x = noise / xmin (div_by_f32: the division's result from the reciprocal calc_xmin left), its class
    //    (over? tmp) -- without the logarithm wherever that is safe (noise_class, lhip_math.h) -- and the cache.  The reference caches
    //    the Float32 copy of the noise and divides it by xmin again in every later call; that quotient is formed here, once.  Lanes whose
    //    band sits next to a step of the class function (or outside the shortcut's range) make the whole wave take the logarithm for
    //    this call (about one call in a hundred); the logarithm is also taken when the caller will read max_noise (need_max) or the
    //    result turns out to have no distorted band.
    double x = 1.0;                                   // this lane's band: noise / xmin (cached bands: as last evaluated)
    int cls = 0;                                      // class used by THIS call
    const int fresh = my_fresh;
    LHIP_LANE_ONCE(sfb, 0, g.psymax) {
        if (!fresh) {
            S.distort[sfb] = L.pn_dist[sfb];
            x = L.pn_x[sfb];
            cls = L.pn_cls[sfb];
        } else {
            const double noise = S.nsum[sfb], b = (double)L.xmin[sfb], rb = L.rxmin[sfb];
            const double nf = (double)(float)noise;
            x = div_by_f32(noise, b, rb);
            double xf = div_by_f32(nf, b, rb);
            if (rb == 0.0) { x = noise / b; xf = nf / b; }         // never on sane material: an xmin outside div_by_f32's proof
            S.distort[sfb] = (float)x;
            if (defer) { defer->step = my_step; defer->dist = (float)xf; }
            else if (use_pn) { L.pn_step[sfb] = my_step; L.pn_dist[sfb] = (float)xf; }
            cls = need_max ? -1 : noise_class(x);          // the logarithm will be formed anyway: no shortcut
        }
    }
    PH_MARK(L, PH_N_TERMS, tm_);
    int cls_cache = cls;                              // class the reference will derive from its Float32 copy in later calls
    int have_log = 0;
    double nl = -20.0;                                // noise_log as this call sees it (only valid when have_log)
    if (wave_any(cls < 0)) {
        LHIP_LANE_ONCE(sfb, 0, g.psymax) {
            const double l = v8_log10_pos(x > 1E-20 ? x : 1E-20);       // operand >= 1e-20 (a NaN compares false and becomes 1e-20 too)
            const double lf = (double)(float)l;
            nl = fresh ? l : lf;
            if (fresh) { cls = noise_class_of_log(l); cls_cache = noise_class_of_log(lf); }
        }
        have_log = 1;
    }
    int over = 0, ssd = 0;
    if (defer) { defer->fresh = 0; LHIP_LANE_ONCE(sfb, 0, g.psymax) { defer->fresh = use_pn && fresh; defer->x = x; defer->cls = cls_cache; } }
    LHIP_LANE_ONCE(sfb, 0, g.psymax) {
        if (!defer && use_pn && fresh) { L.pn_x[sfb] = x; L.pn_cls[sfb] = (int16_t)cls_cache; }
        if (cls > 0) { ssd = cls * cls; over = 1; }
    }
    if (use_pn && !defer) pn.gain = g.global_gain;
    {   // over <= 39 bands and over_SSD <= 39 * 400^2 < 2^25: one packed integer reduction
        const int os = wave_sum((ssd << 6) | over);
        res->over_count = os & 63; res->over_SSD = os >> 6;
    }
    res->max_noise = 0.0;
    if (need_max || res->over_count == 0) {
        double max_noise = -20.0;
        if (!have_log) {
            LHIP_LANE_ONCE(sfb, 0, g.psymax) {
                const double l = v8_log10_pos(x > 1E-20 ? x : 1E-20);
                nl = fresh ? l : (double)(float)l;
            }
        }
        LHIP_LANE_ONCE(sfb, 0, g.psymax) max_noise = nl;
        res->max_noise = wave_maxd(max_noise);
    }
#endif
    wave_sync();
    PH_MARK(L, PH_N_SUMS, tm_);
}

LHIP_DEV void q_calc_noise(const Tables& T, const GI& g, const int32_t* scalefac, const int16_t* ix, NoiseRes* res,
                           int use_pn, PrevNoise& pn, int need_max, int lane, QuantLds& L, const QuantTabs& Q) {
    PH_BEGIN();
    q_calc_noise_(T, g, scalefac, ix, res, use_pn, pn, need_max, lane, L, Q);
    PH_END(L, PH_NOISE);
}

// ---------------------------------------------------------------------------------------------
// count_bits on a second wave (latency kernels: a workgroup per frame with idle waves -- g_frame).
// One evaluation of the outer loop is  quantize -> count_bits -> [fits?] -> calc_noise  and calc_noise needs the quantized values but
// nothing count_bits produces.  So the owner of a granule-channel quantizes, hands the Huffman count of that quantization (noquant_count_bits,
// Takehiro.js:521-628: a pure function of the quantized values, the block type and the region fields it may leave untouched) to a
// helper wave through a record in LDS, and runs calc_noise itself MEANWHILE -- speculatively: its cache writes are held back
// (NoiseCommit) until the count is in and the loop's decisions say that this evaluation is the one calc_noise follows
// (Quantize.js:986-1018); otherwise the result is dropped and nothing of it remains (calc_noise's other outputs -- distort, the band sums
// -- are rewritten by the next call before anybody reads them).  Same values, same order of the reference's decisions; only the clock changes.
// CountShare::state: CS_REQ request posted (owner), CS_DONE reply posted (helper), CS_BARRIER the owner is about to wait at a workgroup
// barrier and the helper must keep the count, CS_QUIT the owner's frame is finished.
// ---------------------------------------------------------------------------------------------
// The hand-over is on the search's critical path, so it is three LDS words each way: the request is the state word itself (CS_REQ | block
// type << 8 -- the count reads nothing else of the granule's state: the quantized values are in the owner's record), the reply packs what
// noquant_count_bits assigns (w[0] bits | count1 << 17; w[1] big_values | count1bits << 10 | count1table_select << 23 | region0_count << 24 |
// region1_count << 28; w[2] table_select[0..2] + 1 at 6 bits each | sfb_count1 << 18 | the mask of conditionally assigned fields << 24: the
// owner takes the region counts and table_select[r] only where the reference's count assigns them, Takehiro.js:566-612).
struct CountShare { int state; uint32_t w[3];
    int here;                   // the helper is listening (set when it enters q_count_helper, cleared where the record is reset): look-ahead requests are only posted then
    int32_t xmax[2];            // the granule's xrpow_max (posted once per search): what a look-ahead evaluation of the bin search needs beside the gain (CS_REQ2)
    int16_t kept[576];          // the search's kept copy of the quantized spectrum (q_outer_loop's `kept`): in LDS where a wave is alone with its memory latencies
                                // (the batched kernel keeps it in HBM -- the granule-channel's slot of W.l3 -- and reads it back once per search)
#if defined(LHIP_PHASE_PROF) || defined(LHIP_HANDOFF_PROF)
    unsigned long long t_post, t_reply; unsigned int acc[8];      // profiling builds: the hand-over's legs (q_count_helper)
#endif
};
// -DLHIP_HANDOFF_PROF (without LHIP_PHASE_PROF): the production code with four clock reads per counted evaluation and fire-and-forget LDS adds --
// what the hand-over costs when nothing else is instrumented (tests/tools/handoff_prof.py).  acc: 0 owner: posted -> calc_noise finished, 1 owner: waited,
// 2 evaluations, 3 helper: request seen -> reply written, 4 helper's evaluations, 5 posted -> seen by the helper, 6 reply written -> seen by the owner
#if defined(LHIP_HANDOFF_PROF) && !defined(LHIP_PHASE_PROF)
#define HO_NOW() __builtin_amdgcn_s_memtime()
#define HO_ADD(CS_, I_, V_) do { if (lane == 0) __hip_atomic_fetch_add(&(CS_).acc[I_], (unsigned int)(V_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
#define HO_ON 1
#else
#define HO_ON 0
#endif
enum { CS_IDLE = 0, CS_REQ = 1, CS_DONE = 2, CS_REQ2 = 3, CS_BARRIER = 8, CS_QUIT = 9 };
#ifndef LHIP_BS_AHEAD
#define LHIP_BS_AHEAD 0       /* 1: look-ahead evaluations of the bin search on the count helper (q_outer_loop).  Built in round 6, bit-exact on the device (1050 cases) and in the wave
                                 simulation (which is compiled with it, tests/hostsim/Makefile); a taken evaluation costs the owner ~ 1.5 k cycles instead of 6.4 k and the predictor is right for
                                 96 / 78 / 53 - 60 % of the evaluations (steady two-channel / one-channel / moving material) -- but the one-frame kernel with this code in it is 5 - 11 % SLOWER
                                 on every line: the owner's unchanged calc_noise takes 4 - 8 % longer, the helper's unchanged count 3 - 10 % (tests/tools/handoff_prof.py): the kernel's scalar
                                 registers are full, and whatever is added to the search loop is paid for in spills by all of it.  profiles/r06_bs_lookahead_ab.txt.  Off. */
#endif
#if defined(LHIP_WAVESIM)
// how often each way was taken (printed at exit with LAMEJS_PIPE_STATS=1: the simulation must exercise both)
struct PipeStats { long piped = 0, committed = 0, posted = 0, taken = 0, bs_posted = 0, bs_taken = 0; ~PipeStats() { if (getenv("LAMEJS_PIPE_STATS")) fprintf(stderr, "count helper: %ld evaluations counted on the helper wave, %ld of the calc_noise calls made beside them committed\ncandidate helpers: %ld next-gain evaluations posted, %ld taken\nbin-search look-ahead: %ld evaluations posted to the count helper, %ld taken\n", piped, committed, posted, taken, bs_posted, bs_taken); } };
inline PipeStats& pipe_stats() { static PipeStats t; return t; }
#define LHIP_PIPE_COUNT(f) do { if (lane == 0) pipe_stats().f++; } while (0)
#else
#define LHIP_PIPE_COUNT(f) do { } while (0)
#endif
// ---------------------------------------------------------------------------------------------
// The NEXT gain of an outer-loop evaluation, evaluated beside it (one-frame launches, round 6).
// After balance_noise has amplified bands, the first evaluation of a round often does not fit its budget and the reference steps the gain up until one does
// (`while (bits > huff_bits) gain++`, Quantize.js:986-1013): on the bench materials 39 - 44 % of the rounds of a two-channel frame make exactly one such step,
// 3 - 6 % more (tools/search_stats.py).  Every step is a full evaluation -- quantize, count, calc_noise -- behind the one before it, because the 0/1 shortcut of
// quantize_xrpow at gain g + 1 is steered by PrevNoise.sfb_count1 as the count at gain g left it.  Two otherwise idle waves make that second evaluation WHILE the
// owner makes the first: each repeats the owner's quantization at g on its own (a pure function of what the owner posts and of arrays nobody writes during a
// round), derives sfb_count1 from it (q_cnt1_after: big_values is a matter of two ballots), quantizes at g + 1 and then -- role 0 -- counts the Huffman bits
// or -- role 1 -- runs calc_noise with the cache writes held back (NoiseCommit, left in its own record).  If the owner's evaluation fits, nothing of this is
// looked at; if it does not, the evaluation at g + 1 is already there: the owner takes the count's reply exactly as it takes its count helper's, the quantized
// lines and the noise results from role 1's record, and goes on as if it had made the evaluation itself.  Same values, same order of the reference's
// decisions.  A request the owner has left behind (its round fitted) is finished on arrays that may be changing under the helper: `safe` mode clamps every
// table index, and nobody reads the result.
// state[role]: CS_IDLE / CS_REQ | gain << 8 | sfb_count1 << 16 | need_max << 22 | pn.gain << 23 (owner) / CS_DONE (helper) / CS_BARRIER / CS_QUIT as for CountShare.
// ---------------------------------------------------------------------------------------------
struct alignas(8) CandShare {
    int state[2], present[2];   // (state: one 8-byte word for the owner -- both helpers get the same request, and "are both free" is one look)
    int32_t gi[14];             // the owner's granule state that changes at most once per round (q_cand_sync): see q_cand_helper
    uint32_t cw[4];             // role 0's reply: CountShare::w[0 .. 2] of the evaluation at gain + 1; [3] sfb_count1 after the evaluation at gain (cross-check)
    int32_t nres[4];            // role 1's reply: max_noise (two words), over_count, over_SSD
#if defined(LHIP_HANDOFF_PROF)
    unsigned int acc[8];        // profiling build (tests/tools/handoff_prof.py): 0 posted, 1 taken, 2 helpers not free when a post was due, 3 cycles waited for a taken evaluation,
                                // 4 / 5 role 0's busy cycles / requests, 6 role 1's busy cycles, 7 cycles from post to take
    unsigned long long t_post;
#endif
};
#if defined(LHIP_HANDOFF_PROF) && !defined(LHIP_HOSTSIM)
#define CD_ADD(CD_, I_, V_) do { if (lane == 0) __hip_atomic_fetch_add(&(CD_).acc[I_], (unsigned int)(V_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
#define CD_NOW() __builtin_amdgcn_s_memtime()
#else
#define CD_ADD(CD_, I_, V_) do { } while (0)
#define CD_NOW() 0ull
#endif
#ifndef LHIP_CAND_SLEEP
#define LHIP_CAND_SLEEP 2       /* x 64 clocks */
#endif
#if LHIP_NL != 1
LHIP_DEV int q_cnt1_after(const GI& g, const int (&vx)[NPL], const int (&vy)[NPL], const QuantTabs& Q) {
    // PrevNoise.sfb_count1 as noquant_count_bits leaves it (Takehiro.js:521-560, 618-628): 0 unless the block is a long one with big_values > 0
    if (g.block_type != NORM_TYPE) return 0;
    int tp = -1, bp = -1;
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        const uint64_t nz = wave_ballot((vx[j] | vy[j]) != 0), bg = wave_ballot((vx[j] | vy[j]) > 1);
        if (nz) tp = 64 * j + 63 - (int)__builtin_clzll(nz);
        if (bg) bp = 64 * j + 63 - (int)__builtin_clzll(bg);
    }
    int i = 2 * tp + 2;
    const int P = tp + 1, nq = i >> 2;
    int firstbig = (P - 1 - bp) >> 1;
    if (firstbig > nq) firstbig = nq;
    i -= 4 * firstbig;
    return i == 0 ? 0 : (int)(Q.bvtab[i >> 1] >> 27);
}
// the owner's side: the part of its state a helper needs and that only balance_noise changes
LHIP_DEV void q_cand_sync(CandShare& cd, const GI& w, int lane) {
    lane = lane_anew(lane);
    if (lane == 0) {
        int32_t* k = cd.gi;
        k[0] = w.block_type; k[1] = w.preflag; k[2] = w.scalefac_scale; k[3] = w.subblock_gain[0]; k[4] = w.subblock_gain[1]; k[5] = w.subblock_gain[2]; k[6] = w.subblock_gain[3];
        k[7] = w.max_nonzero_coeff; k[8] = w.firstcut; k[9] = w.psymax; k[10] = w.sfbmax;
        union { double d; int32_t i[2]; } u; u.d = w.xrpow_max; k[11] = u.i[0]; k[12] = u.i[1];
    }
}
LHIP_DEV int q_cand_present(const CandShare& cd, int lane) { int a, b; wg_load2(cd.present, &a, &b, lane); return a & b; }       // both helpers have arrived (they never leave before CS_QUIT)
LHIP_DEV int q_cand_free(const CandShare& cd, int lane) {       // ... and both are waiting for work: one look at the pair of state words
    int s0, s1;
    wg_load2(cd.state, &s0, &s1, lane);
    return (s0 == CS_IDLE || s0 == CS_DONE) && (s1 == CS_IDLE || s1 == CS_DONE);
}
LHIP_DEV void q_cand_post(CandShare& cd, int gain, int cnt1, int need_max, int pn_gain, int lane) {
    const int req = CS_REQ | (gain << 8) | (cnt1 << 16) | ((need_max != 0) << 22) | (pn_gain << 23);
    wg_store2(cd.state, req, req, lane);
}
// a word for both helpers (CS_BARRIER / CS_QUIT): only once they are not working on a request -- their CS_DONE would overwrite it
LHIP_DEV void q_cand_signal(CandShare& cd, int v, int lane) {
    for (int r = 0; r < 2; r++) {
        long n = 0;
        for (;;) { const int s = wg_load(&cd.state[r], lane); if (s == CS_IDLE || s == CS_DONE) break; wg_spin(); LHIP_SPIN_GUARD(n); }
        wg_store(&cd.state[r], v, lane);
    }
}
// a helper: role 0 counts, role 1 runs calc_noise.  Lo = the owner's record, L = this wave's own.
// (LHIP_CAND_NOINLINE, experiment: the helper as a real function -- its code then takes no part in the register allocation of the kernel it is called from)
#if defined(LHIP_CAND_NOINLINE) && !defined(LHIP_HOSTSIM)
static __device__ __attribute__((noinline)) void q_cand_helper(const Tables& T, CandShare& cd, int role, const QuantLds& Lo, QuantLds& L, const QuantTabs& Q, int lane) {
#else
LHIP_DEV void q_cand_helper(const Tables& T, CandShare& cd, int role, const QuantLds& Lo, QuantLds& L, const QuantTabs& Q, int lane) {
#endif
    wg_store(&cd.present[role], 1, lane);
    for (;;) {
        // (a pause between the looks: an idle helper shares its SIMD with a searching wave or its count helper, and a request picked up ~100 clocks later costs
        //  nothing -- the owner only asks for the result a whole evaluation later)
        int s;
        for (long n = 0;;) { s = wg_load(&cd.state[role], lane); if (s != CS_IDLE && s != CS_DONE) break; wg_pause(LHIP_CAND_SLEEP); LHIP_SPIN_GUARD(n); }
        if (s == CS_QUIT) break;
        if (s == CS_BARRIER) { wg_store(&cd.state[role], CS_IDLE, lane); wg_barrier(); continue; }
        const unsigned long long hb0_ = CD_NOW(); (void)hb0_;
        wg_acquire();
        lane = lane_anew(lane);
        GI g;
        const int32_t* k = cd.gi;
        g.part2_3_length = 0; g.big_values = 0; g.count1 = 0; g.scalefac_compress = 0; g.table_select[0] = g.table_select[1] = g.table_select[2] = 0;
        g.region0_count = 0; g.region1_count = 0; g.count1table_select = 0; g.part2_length = 0; g.sfb_lmax = 0; g.sfb_smin = 0; g.psy_lmax = 0; g.sfbdivide = 0; g.count1bits = 0;
        g.global_gain = uni((s >> 8) & 255);
        const int c0 = uni((s >> 16) & 63), need_max = uni((s >> 22) & 1), pn_gain = uni((int)((unsigned)s >> 23));
        g.block_type = uni(k[0]); g.preflag = uni(k[1]); g.scalefac_scale = uni(k[2]);
        g.subblock_gain[0] = uni(k[3]); g.subblock_gain[1] = uni(k[4]); g.subblock_gain[2] = uni(k[5]); g.subblock_gain[3] = uni(k[6]);
        g.max_nonzero_coeff = uni(k[7]); g.firstcut = uni(k[8]); g.psymax = uni(k[9]); g.sfbmax = uni(k[10]);
        { union { double d; int32_t i[2]; } u; u.i[0] = uni(k[11]); u.i[1] = uni(k[12]); g.xrpow_max = u.d; }
        QuantLds& Lin = const_cast<QuantLds&>(Lo);        // (read-only use: the functions below only write through `ix`, `L` / `&L`)
        int vx[NPL], vy[NPL];
        q_quantize(T, g, Lo.sfw, L.ixw, 1, pn_gain, c0, vx, vy, lane, Lin, Q, Lo.ixw, 1);          // the owner's evaluation at `gain`, repeated
        const int c1 = uni(q_cnt1_after(g, vx, vy, Q));
        g.global_gain = g.global_gain < 255 ? g.global_gain + 1 : 255;
        q_quantize(T, g, Lo.sfw, L.ixw, 1, pn_gain, c1, vx, vy, lane, Lin, Q, L.ixw, 1);           // (gain + 1 > pn.gain: no band is cached)
        if (role == 0) {
            int cnt1 = 0, amask = 0;
            const int bits = uni(q_noquant_count_bits(T, g, L.ixw, vx, vy, 1, &cnt1, &amask, lane, L, Q));
            uni_gi(g);
            const uint32_t w0 = (uint32_t)bits | ((uint32_t)g.count1 << 17);
            const uint32_t w1 = (uint32_t)g.big_values | ((uint32_t)g.count1bits << 10) | ((uint32_t)g.count1table_select << 23) | ((uint32_t)g.region0_count << 24) | ((uint32_t)g.region1_count << 28);
            const uint32_t w2 = (uint32_t)(g.table_select[0] + 1) | ((uint32_t)(g.table_select[1] + 1) << 6) | ((uint32_t)(g.table_select[2] + 1) << 12) | ((uint32_t)uni(cnt1) << 18) | ((uint32_t)uni(amask) << 24);
            if (lane == 0) { cd.cw[0] = w0; cd.cw[1] = w1; cd.cw[2] = w2; cd.cw[3] = (uint32_t)c1; }
        } else {
            NoiseRes ni; NoiseCommit nc; nc.fresh = 0; nc.step = 0; nc.cls = 0; nc.dist = 0.f; nc.x = 0.0;
            PrevNoise pn; pn.gain = pn_gain; pn.sfb_count1 = c1;
            q_calc_noise_(T, g, Lo.sfw, L.ixw, &ni, 1, pn, need_max, lane, Lin, Q, &nc, &L, 1);
            lane = lane_anew(lane);
            LHIP_LANE_ONCE(sfb, 0, (SFBMAX) + 1) { L.qmode[sfb] = nc.fresh; L.pn_step[sfb] = nc.step; L.pn_dist[sfb] = nc.dist; L.pn_x[sfb] = nc.x; L.pn_cls[sfb] = (int16_t)nc.cls; }
            if (lane == 0) { union { double d; int32_t i[2]; } u; u.d = ni.max_noise; cd.nres[0] = u.i[0]; cd.nres[1] = u.i[1]; cd.nres[2] = ni.over_count; cd.nres[3] = ni.over_SSD; }
        }
        CD_ADD(cd, role == 0 ? 4 : 6, CD_NOW() - hb0_); if (role == 0) CD_ADD(cd, 5, 1);
        wg_store(&cd.state[role], CS_DONE, lane);
    }
}
// the owner takes the evaluation at w.global_gain (already stepped up) that the helpers made beside its last one.  Returns the bit count, or -1 if the helpers'
// view of sfb_count1 is not the owner's (cannot happen: both derive it from the same quantization; the caller then evaluates by itself)
LHIP_DEV int q_cand_take(CandShare& cd, const QuantLds& Lh, GI& g, PrevNoise& pn, NoiseRes* ni, NoiseCommit* nc, int lane, QuantLds& L) {
    for (int r = 0; r < 2; r++) {
        long n = 0;
        while (wg_load(&cd.state[r], lane) != CS_DONE) { wg_spin(); LHIP_SPIN_GUARD(n); }
    }
    wg_acquire();
    lane = lane_anew(lane);
    const uint32_t w0 = (uint32_t)uni((int)cd.cw[0]), w1 = (uint32_t)uni((int)cd.cw[1]), w2 = (uint32_t)uni((int)cd.cw[2]);
    const int c1 = uni((int)cd.cw[3]);
    if (c1 != pn.sfb_count1) return -1;
    const int bits = (int)(w0 & 0x1ffffu), amask = (int)((w2 >> 24) & 15u);
    g.count1 = (int)(w0 >> 17); g.big_values = (int)(w1 & 1023u); g.count1bits = (int)((w1 >> 10) & 0x1fffu); g.count1table_select = (int)((w1 >> 23) & 1u);
    if (amask & 8) { g.region0_count = (int)((w1 >> 24) & 15u); g.region1_count = (int)(w1 >> 28); }
    if (amask & 1) g.table_select[0] = (int)(w2 & 63u) - 1;
    if (amask & 2) g.table_select[1] = (int)((w2 >> 6) & 63u) - 1;
    if (amask & 4) g.table_select[2] = (int)((w2 >> 12) & 63u) - 1;
    pn.sfb_count1 = (int)((w2 >> 18) & 63u);
    { union { double d; int32_t i[2]; } u; u.i[0] = uni(cd.nres[0]); u.i[1] = uni(cd.nres[1]); ni->max_noise = u.d; ni->over_count = uni(cd.nres[2]); ni->over_SSD = uni(cd.nres[3]); ni->bits = 0; }
    // what calc_noise left for this evaluation: the cache entries it would write (lane = band), the distortions, and the quantized lines themselves
    nc->fresh = 0; nc->step = 0; nc->cls = 0; nc->dist = 0.f; nc->x = 0.0;
    LHIP_LANE_ONCE(sfb, 0, (SFBMAX) + 1) { nc->fresh = Lh.qmode[sfb]; nc->step = Lh.pn_step[sfb]; nc->dist = Lh.pn_dist[sfb]; nc->x = Lh.pn_x[sfb]; nc->cls = Lh.pn_cls[sfb]; L.distort[sfb] = Lh.distort[sfb]; }
    {
        uint32_t kw[NPL];
#pragma unroll
        for (int j = 0; j < NPL; j++) { const int i = lane + LHIP_NL * j; kw[j] = ((const uint32_t*)Lh.ixw)[i < 288 ? i : 287]; }
#pragma unroll
        for (int j = 0; j < NPL; j++) { const int i = lane + LHIP_NL * j; if (i < 288) ((uint32_t*)L.ixw)[i] = kw[j]; }
    }
    wave_sync();
    return bits;
}
#endif
#if LHIP_NL != 1
// the helper: serves one owner's requests until it is told to leave.  Lo = the owner's LDS record (the quantized values), L = this wave's own (scratch).
LHIP_DEV void q_count_helper(const Tables& T, CountShare& cs, const QuantLds& Lo, QuantLds& L, const QuantTabs& Q, int lane) {
    if (LHIP_BS_AHEAD) wg_store(&cs.here, 1, lane);
    for (;;) {
        const int s = wg_wait_not(&cs.state, CS_IDLE, CS_DONE, lane);       // a request, a barrier to keep, or the end
        if (s == CS_QUIT) break;
        if (s == CS_BARRIER) { wg_store(&cs.state, CS_IDLE, lane); wg_barrier(); continue; }     // (the owner posts again only after the barrier)
#ifdef LHIP_PHASE_PROF
        const unsigned long long hp_seen_ = __builtin_amdgcn_s_memtime();
#endif
#if HO_ON
        const unsigned long long ho_h0_ = HO_NOW();
#endif
        wg_acquire();
        lane = lane_anew(lane);
        GI g;
        const int ahead = LHIP_BS_AHEAD ? uni((s & 255) == CS_REQ2) : 0;
        g.block_type = uni(ahead ? (s >> 16) & 255 : s >> 8); g.max_nonzero_coeff = 0;
        g.region0_count = 0; g.region1_count = 0; g.table_select[0] = g.table_select[1] = g.table_select[2] = 0;
        g.count1table_select = 0; g.count1bits = 0; g.count1 = 0; g.big_values = 0;
        int cnt1 = 0, amask = 0;
        int vx[NPL], vy[NPL];
        int large = 0;
        if (ahead) {
            // a look-ahead evaluation of the owner's bin search (q_outer_loop): count_bits at the requested gain -- a pure function of the gain, the block type
            // and arrays nobody writes during a bin search (xrpow; no scalefactors, no noise cache yet): the owner's own evaluation, made beside it.  The
            // quantized lines stay in this wave's record: the owner copies them if this evaluation ends its search.
            g.global_gain = uni((s >> 8) & 255); g.firstcut = 64; g.max_nonzero_coeff = 575; g.preflag = 0; g.scalefac_scale = 0;
            g.subblock_gain[0] = g.subblock_gain[1] = g.subblock_gain[2] = g.subblock_gain[3] = 0;
            { union { double d; int32_t i[2]; } u; u.i[0] = uni(cs.xmax[0]); u.i[1] = uni(cs.xmax[1]); g.xrpow_max = u.d; }
            const double ip = ipow20(Q, g.global_gain);          // Takehiro.js:635-638, as q_count_bits
            if (g.xrpow_max * ip > (double)IXMAX_VAL * (1.0 - 0x1p-50)) { const double w = (double)IXMAX_VAL / ip; if (g.xrpow_max > w) large = 1; }
            if (!large) q_quantize(T, g, Lo.sfw, L.ixw, 0, 0, 0, vx, vy, lane, const_cast<QuantLds&>(Lo), Q);
        } else {
#pragma unroll
        for (int j = 0; j < NPL; j++) {
            const int i = lane + LHIP_NL * j;
            const uint32_t w = i < 288 ? ((const uint32_t*)Lo.ixw)[i] : 0u;
            vx[j] = (int)(w & 0xffffu); vy[j] = (int)(w >> 16);
        }
        }
#ifdef LHIP_PHASE_PROF
        const unsigned long long hp_start_ = __builtin_amdgcn_s_memtime();
#endif
        int bits_ = (int)LARGE_BITS;
        if (!large) bits_ = q_noquant_count_bits(T, g, ahead ? L.ixw : Lo.ixw, vx, vy, !ahead, &cnt1, &amask, lane, L, Q);      // (one instance of the count for both kinds of request)
        const int bits = uni(bits_);
#ifdef LHIP_PHASE_PROF
        const unsigned long long hp_end_ = __builtin_amdgcn_s_memtime();
#endif
        uni_gi(g);
        const uint32_t w0 = (uint32_t)bits | ((uint32_t)g.count1 << 17);
        const uint32_t w1 = (uint32_t)g.big_values | ((uint32_t)g.count1bits << 10) | ((uint32_t)g.count1table_select << 23) | ((uint32_t)g.region0_count << 24) | ((uint32_t)g.region1_count << 28);
        const uint32_t w2 = (uint32_t)(g.table_select[0] + 1) | ((uint32_t)(g.table_select[1] + 1) << 6) | ((uint32_t)(g.table_select[2] + 1) << 12) | ((uint32_t)uni(cnt1) << 18) | ((uint32_t)uni(amask) << 24);
        if (lane == 0) { cs.w[0] = w0; cs.w[1] = w1; cs.w[2] = w2; }
#ifdef LHIP_PHASE_PROF
        if (lane == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); cs.acc[0] += (unsigned int)(hp_seen_ - cs.t_post); cs.acc[1] += (unsigned int)(hp_start_ - hp_seen_);
                         cs.acc[2] += (unsigned int)(hp_end_ - hp_start_); cs.acc[3] += (unsigned int)(now_ - hp_end_); cs.acc[7] += 1; cs.t_reply = now_; }
#endif
#if HO_ON
        const unsigned long long ho_h1_ = HO_NOW();
        if (lane == 0) cs.t_reply = ho_h1_;
#endif
        wg_store(&cs.state, CS_DONE, lane);
#if HO_ON
        HO_ADD(cs, 3, ho_h1_ - ho_h0_); HO_ADD(cs, 4, 1); HO_ADD(cs, 5, ho_h0_ - cs.t_post);
#endif
    }
}
// the owner's side: q_count_bits (use_pn = 1) with the count on the helper and calc_noise run beside it.  *spec = 1: `ni` / `nc` hold a calc_noise
// of this quantization that has not been committed (q_noise_commit)
LHIP_DEV int q_count_bits_piped(const Tables& T, GI& g, const int32_t* scalefac, int16_t* ix, PrevNoise& pn, int* asg, CountShare& cs,
                                NoiseRes* ni, NoiseCommit* nc, int need_max, int* spec, int lane, QuantLds& L, const QuantTabs& Q) {
    *asg = 0; *spec = 0;
    {   // Takehiro.js:635-638 (see q_count_bits)
        const double ip = ipow20(Q, g.global_gain);
        if (g.xrpow_max * ip > (double)IXMAX_VAL * (1.0 - 0x1p-50)) {
            const double w = (double)IXMAX_VAL / ip;
            if (g.xrpow_max > w) return LARGE_BITS;
        }
    }
    {
        int vx[NPL], vy[NPL];
        PH_BEGIN(); q_quantize(T, g, scalefac, ix, 1, pn.gain, pn.sfb_count1, vx, vy, lane, L, Q); PH_END(L, PH_QUANTIZE);
        (void)vx; (void)vy;                          // the helper reads the pairs back from LDS
    }
#ifdef LHIP_PHASE_PROF
    if (lane == 0) cs.t_post = __builtin_amdgcn_s_memtime();
#endif
#if HO_ON
    const unsigned long long ho_t0_ = HO_NOW();
    if (lane == 0) cs.t_post = ho_t0_;
#endif
    wg_store(&cs.state, CS_REQ | (g.block_type << 8), lane);
    LHIP_PIPE_COUNT(piped);
    { PH_BEGIN(); q_calc_noise_(T, g, scalefac, ix, ni, 1, pn, need_max, lane, L, Q, nc); PH_END(L, PH_NOISE); }
    *spec = 1;
#if HO_ON
    const unsigned long long ho_t1_ = HO_NOW();
#endif
    PH_BEGIN();
#ifdef LHIP_PHASE_PROF
    const unsigned long long op_wait0_ = __builtin_amdgcn_s_memtime();
#endif
    (void)wg_wait_not(&cs.state, CS_REQ | (g.block_type << 8), CS_REQ | (g.block_type << 8), lane);       // the only thing it can become is CS_DONE
#ifdef LHIP_PHASE_PROF
    if (lane == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); cs.acc[4] += (unsigned int)(now_ - cs.t_reply); cs.acc[5] += (unsigned int)(op_wait0_ - cs.t_post); cs.acc[6] += (unsigned int)(now_ - op_wait0_); }
#endif
#if HO_ON
    const unsigned long long ho_t2_ = HO_NOW();
#endif
    wg_acquire();
    const uint32_t w0 = (uint32_t)uni((int)cs.w[0]), w1 = (uint32_t)uni((int)cs.w[1]), w2 = (uint32_t)uni((int)cs.w[2]);
#if HO_ON
    HO_ADD(cs, 0, ho_t1_ - ho_t0_); HO_ADD(cs, 1, ho_t2_ - ho_t1_); HO_ADD(cs, 2, 1); HO_ADD(cs, 6, ho_t2_ - cs.t_reply);
#endif
    const int bits = (int)(w0 & 0x1ffffu), amask = (int)((w2 >> 24) & 15u);
    static_assert(LARGE_BITS < (1 << 17), "the reply packs the bit count in 17 bits");
    g.count1 = (int)(w0 >> 17); g.big_values = (int)(w1 & 1023u); g.count1bits = (int)((w1 >> 10) & 0x1fffu); g.count1table_select = (int)((w1 >> 23) & 1u);
    if (amask & 8) { g.region0_count = (int)((w1 >> 24) & 15u); g.region1_count = (int)(w1 >> 28); }       // (region1_count is 13 for start / stop blocks: four bits)
    if (amask & 1) g.table_select[0] = (int)(w2 & 63u) - 1;
    if (amask & 2) g.table_select[1] = (int)((w2 >> 6) & 63u) - 1;
    if (amask & 4) g.table_select[2] = (int)((w2 >> 12) & 63u) - 1;
    *asg = pack_cond_fields(g, amask);
    pn.sfb_count1 = (int)((w2 >> 18) & 63u);
    PH_END(L, PH_COUNT);                             // (profiling builds: the part of the count the owner still had to wait for)
    return bits;
}
#endif

// ---------------------------------------------------------------------------------------------
// scale_bitcount (Takehiro.js:980-1030), MPEG-1, no mixed blocks.  returns 1 on failure
// ---------------------------------------------------------------------------------------------
LHIP_DEV int q_scale_bitcount(const QuantTabs& Q, GI& g, int32_t* scalefac, int lane) {
    lane = fresh_lane(lane);
    const int sh = (g.block_type == SHORT_TYPE) ? 24 : 16;       // scale_short / scale_long byte of Q.sbc
    if (g.block_type != SHORT_TYPE && 0 == g.preflag) {
        int bad = 0;
        LHIP_LANE_ONCE(sfb, 11, SBPSY_l) if (scalefac[sfb] < Q.pretab[sfb]) bad = 1;
        if (!wave_any(bad)) {
            g.preflag = 1;
            wave_sync();
            LHIP_LANE_ONCE(sfb, 11, SBPSY_l) scalefac[sfb] -= Q.pretab[sfb];
            wave_sync();
        }
    }
    // slen1_n / slen2_n are powers of two, and "every value < 2^b" is "the OR of the values < 2^b": the two maxima
    // of the reference become one OR reduction of (part 1 | part 2 << 8)
    int m12 = 0;
    LHIP_LANE_ONCE(sfb, 0, g.sfbmax) {
        const int v = scalefac[sfb];
        m12 |= (sfb < g.sfbdivide) ? v : (v << 8);
    }
    m12 = wave_or(m12);
    const int m1 = m12 & 0xff, m2 = m12 >> 8;
    // first k with the smallest tab[k] among the admissible ones == minimum of (tab[k], k) pairs; one lane per k
    int best = 0x7fffffff;
    LHIP_LANE_ONCE(k, 0, 16) {
        const uint32_t e = Q.sbc[k];
        if (m1 < (int)(e & 0xffu) && m2 < (int)((e >> 8) & 0xffu)) { const int v = (int)((e >> sh) & 0xffu) * 16 + k; if (v < best) best = v; }
    }
    best = wave_min(best);
    g.part2_length = LARGE_BITS;
    if (best != 0x7fffffff) { g.part2_length = best >> 4; g.scalefac_compress = best & 15; }
    return g.part2_length == LARGE_BITS;
}

// scale_bitcount_lsf (Takehiro.js:1046-1132), MPEG-2/2.5, no mixed blocks, no intensity stereo.  The four
// partitions of nr_of_sfb_block[table][row] are runs of the linear scalefactor index (a short band contributes its
// three windows), every max_range_sfac_tab entry is 2^n-1 and log2tab is the bit length, so the four partition maxima
// of the reference become one OR reduction with a byte per partition.  On failure nothing is written (the reference
// leaves scalefac_compress / part2_length as they were); returns 1 on failure.
LHIP_DEV int q_scale_bitcount_lsf(GI& g, const int32_t* scalefac, int lane) {
    lane = fresh_lane(lane);
    const bool pre = g.preflag != 0, sh = g.block_type == SHORT_TYPE;
    // nr_of_sfb_block[0][0..1] = {6,5,5,5} {9,9,9,9}; nr_of_sfb_block[2][0..1] = {11,10,0,0} {18,18,0,0}
    const int n0 = pre ? (sh ? 18 : 11) : (sh ? 9 : 6), n1 = pre ? (sh ? 18 : 10) : (sh ? 9 : 5);
    const int n2 = pre ? 0 : (sh ? 9 : 5), n3 = n2;
    const uint32_t range = pre ? 0x00000307u : 0x07070f0fu;      // max_range_sfac_tab[0] / [2], partition p in byte p
    const int b1 = n0, b2 = n0 + n1, b3 = b2 + n2, end = b3 + n3;
    uint32_t m = 0;
    LHIP_LANE_ONCE(i, 0, end) {
        int v = scalefac[i];
        if (v < 0) v = 0;
        const int part = (i >= b1) + (i >= b2) + (i >= b3);
        m |= (uint32_t)v << (8 * part);
    }
    m = (uint32_t)wave_or((int)m);
    int over = 0, slen[4];
    for (int p = 0; p < 4; p++) {
        const uint32_t mp = (m >> (8 * p)) & 0xffu;
        if (mp > ((range >> (8 * p)) & 0xffu)) over = 1;
        slen[p] = mp ? 32 - __builtin_clz(mp) : 0;
    }
    if (!over) {
        g.scalefac_compress = pre ? 500 + slen[0] * 3 + slen[1] : ((slen[0] * 5 + slen[1]) << 4) + (slen[2] << 2) + slen[3];
        g.part2_length = slen[0] * n0 + slen[1] * n1 + slen[2] * n2 + slen[3] * n3;
    }
    return over;
}

LHIP_DEV int q_scale_bitcount_any(const Tables& T, const QuantTabs& Q, GI& g, int32_t* scalefac, int lane) {   // Quantize.js:814-817, 840-843
    return T.mode_gr == 2 ? q_scale_bitcount(Q, g, scalefac, lane) : q_scale_bitcount_lsf(g, scalefac, lane);
}

// ---------------------------------------------------------------------------------------------
// amplification helpers (Quantize.js:453-460, 597-778)
// ---------------------------------------------------------------------------------------------
LHIP_DEV int q_loop_break(const GI& g, const int32_t* scalefac, int lane, QuantLds& L, const QuantTabs& Q) {
    lane = fresh_lane(lane);
    int z = 0;
    LHIP_LANE_ONCE(sfb, 0, g.sfbmax)
        if (scalefac[sfb] + band_sbgain(g, L.window, sfb) == 0) z = 1;
    return !wave_any(z);
}

// multiply xrpow of the flagged bands (bit sfb of m_amp) by `amp`, tracking xrpow_max.  Branch-free: an unflagged pair is
// multiplied by 1.0 (exact through the f32 -> f64 -> f32 round trip) and takes part in the maximum like the flagged ones --
// xrpow_max already bounds every unflagged line, so the update `if (max > xrpow_max)` sees the same verdict.
LHIP_DEV void q_amplify_flagged(GI& g, double amp, uint64_t m_amp, int lane, QuantLds& L, const QuantTabs& Q) {
    lane = fresh_lane(lane);
    float m = 0.f;
    const uint8_t* l2s = line2sfb(Q, g.block_type);
    struct F2 { float x, y; };
    F2 xx[NPL]; int bnd[NPL];
#pragma unroll
    for (int j = 0; j < NPL; j++) {                             // pairs never straddle a band
        const int p = 2 * (lane + LHIP_NL * j);
        xx[j].x = 0.f; xx[j].y = 0.f; bnd[j] = 0;
        if (p < 576) { xx[j] = *(const F2*)(L.xrpow + p); bnd[j] = l2s[p]; }
    }
#pragma unroll
    for (int j = 0; j < NPL; j++) {
        const int p = 2 * (lane + LHIP_NL * j);
        const double a = ((m_amp >> bnd[j]) & 1) ? amp : 1.0;
        xx[j].x = (float)((double)xx[j].x * a); xx[j].y = (float)((double)xx[j].y * a);
        if (p < 576) *(F2*)(L.xrpow + p) = xx[j];
        m = fmax_nonneg(m, fmax_nonneg(xx[j].x, xx[j].y));      // xrpow values are >= +0
    }
    m = wave_maxf_pos(m);
    if ((double)m > g.xrpow_max) g.xrpow_max = m;
    wave_sync();
}

// amp_scalefac_bands (Quantize.js:597-669) + loop_break (453-460) + the scalefactor statistics scale_bitcount starts with
// (Takehiro.js:980-1005), in ONE pass over the bands: the amplification decision, the incremented scalefactor, "is any band
// still at zero" and, for MPEG-1 long blocks, "could the pretab be subtracted" / the two range maxima all look at the same
// scalefac[sfb].  *all_nonzero = loop_break's verdict; *pre_bad / *m12 feed q_scale_bitcount_from (values BEFORE a preflag
// subtraction, which that function applies itself when it is due).
LHIP_DEV void q_amp_scalefac_bands(const Tables& T, GI& g, int32_t* scalefac, int* all_nonzero, int* pre_bad, int* m12_out,
                                   int lane, QuantLds& L, const QuantTabs& Q) {
    lane = fresh_lane(lane);
    const double ifqstep34 = (g.scalefac_scale == 0) ? 1.29683955465100964055 : 1.68179283050742922612;
    float tr = 0.f;
    LHIP_LANE_ONCE(sfb, 0, g.sfbmax) if (tr < L.distort[sfb]) tr = L.distort[sfb];
    double trigger = wave_maxf_pos(tr);
    // noise_shaping_amp == 1 with a distorted band: the trigger is sqrt(max distortion) (Quantize.js:612-617).  For Float32 values d and M,
    // d < RN(sqrt(M)) <=> d * d < M with the square formed exactly in f64: a Float32 can only equal the correctly rounded root if it IS
    // the root (a 53-bit neighbour of an irrational root is no 24-bit number, and d * d = M means d = sqrt(M)), so no Float32 lies between
    // the root and its rounding -- the comparison needs no square root (a dozen f64 instructions per call, on every lane)
    const int by_square = T.noise_shaping_amp == 1 && trigger > 1.0;
    const double max_dist = trigger;
    switch (T.noise_shaping_amp) {
        case 2: break;
        case 1:
            if (!(trigger > 1.0)) trigger *= .95;
            break;
        default:
            if (trigger > 1.0) trigger = 1.0;
            else trigger *= .95;
            break;
    }
    uint64_t m_amp = 0;                                  // bit sfb: band is amplified
    LHIP_LANE_ONCE(sfb, 0, g.sfbmax) {
        const double d = (double)L.distort[sfb];
        const int below = by_square ? (d * d < max_dist) : (d < trigger);
        if (!below) m_amp |= 1ull << sfb;
    }
    m_amp = wave_lane_bits(m_amp);
    if (T.noise_shaping_amp == 2 && m_amp) m_amp = 1ull << __builtin_ctzll(m_amp);     // amplify exactly one band
    int z = 0, bad = 0, m12 = 0;
    LHIP_LANE_ONCE(sfb, 0, g.sfbmax) {
        const int v = scalefac[sfb] + (int)((m_amp >> sfb) & 1);
        scalefac[sfb] = v;
        if (v + band_sbgain(g, L.window, sfb) == 0) z = 1;
        if (sfb >= 11 && sfb < SBPSY_l && v < Q.pretab[sfb]) bad = 1;
        m12 |= (sfb < g.sfbdivide) ? v : (v << 8);
    }
    {   // three one-bit / small-field facts in one OR reduction: bits 0-15 m12, bit 16 z, bit 17 bad
        const int r = wave_or(m12 | (z << 16) | (bad << 17));
        *m12_out = r & 0xffff; *all_nonzero = !((r >> 16) & 1); *pre_bad = (r >> 17) & 1;
    }
    { unsigned long long tt_ = PH_NOW(); (void)tt_; q_amplify_flagged(g, ifqstep34, m_amp, lane, L, Q); PH_MARKB(L, PH_B_TRIG, tt_); }
}

// scale_bitcount (MPEG-1) continuing from the statistics of q_amp_scalefac_bands: pre_bad = some band 11..20 is below its
// pretab entry, m12 = OR of the scalefactors of part 1 | part 2 << 8.  Same result as q_scale_bitcount on the same scalefactors.
LHIP_DEV int q_scale_bitcount_from(const QuantTabs& Q, GI& g, int32_t* scalefac, int pre_bad, int m12, int lane) {
    lane = fresh_lane(lane);
    const int sh = (g.block_type == SHORT_TYPE) ? 24 : 16;
    if (g.block_type != SHORT_TYPE && 0 == g.preflag && !pre_bad) {
        g.preflag = 1;
        wave_sync();
        LHIP_LANE_ONCE(sfb, 11, SBPSY_l) scalefac[sfb] -= Q.pretab[sfb];
        wave_sync();
        m12 = 0;                                              // the statistics change with the subtraction: take them again
        LHIP_LANE_ONCE(sfb, 0, g.sfbmax) { const int v = scalefac[sfb]; m12 |= (sfb < g.sfbdivide) ? v : (v << 8); }
        m12 = wave_or(m12);
    }
    const int m1 = m12 & 0xff, m2 = m12 >> 8;
    int best = 0x7fffffff;
    LHIP_LANE_ONCE(k, 0, 16) {
        const uint32_t e = Q.sbc[k];
        if (m1 < (int)(e & 0xffu) && m2 < (int)((e >> 8) & 0xffu)) { const int v = (int)((e >> sh) & 0xffu) * 16 + k; if (v < best) best = v; }
    }
    best = wave_min(best);
    g.part2_length = LARGE_BITS;
    if (best != 0x7fffffff) { g.part2_length = best >> 4; g.scalefac_compress = best & 15; }
    return g.part2_length == LARGE_BITS;
}

LHIP_DEV void q_inc_scalefac_scale(const Tables& T, GI& g, int32_t* scalefac, int lane, QuantLds& L, const QuantTabs& Q) {
    lane = fresh_lane(lane);
    uint64_t m_amp = 0;
    LHIP_LANE_ONCE(sfb, 0, g.sfbmax) {
        int s = scalefac[sfb];
        if (g.preflag != 0) s += Q.pretab[sfb];
        if ((s & 1) != 0) { s++; m_amp |= 1ull << sfb; }
        scalefac[sfb] = s >> 1;
    }
    m_amp = wave_lane_bits(m_amp);
    wave_sync();
    g.preflag = 0;
    g.scalefac_scale = 1;
    q_amplify_flagged(g, 1.29683955465100964055, m_amp, lane, L, Q);
}

// inc_subblock_gain (Quantize.js:705-778); returns 1 on failure.  Short blocks only (sfb_lmax == 0).
LHIP_DEV int q_inc_subblock_gain(const Tables& T, GI& g, int32_t* scalefac, int lane, QuantLds& L, const QuantTabs& Q) {
    lane = fresh_lane(lane);
    for (int window = 0; window < 3; window++) {
        int s1 = 0, s2 = 0;
        LHIP_LANE_ONCE3(sfb, window, g.sfbmax) {
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
            int s = (sfb < g.sfbmax) ? scalefac[sfb] : 0;
            wave_sync();                              // every lane has read the scalefactor before lane 0 rewrites it
            if (sfb < g.sfbmax) {
                s = s - (4 >> g.scalefac_scale);
                if (s >= 0) { if (lane == 0) scalefac[sfb] = s; }
                else {
                    if (lane == 0) scalefac[sfb] = 0;
                    amp = ipow20(Q, 210 + s * (1 << (g.scalefac_scale + 1)));      // s < 0 here: a multiplication, not a shift of a negative value
                    doamp = 1;
                }
            } else { amp = ipow20(Q, 202); doamp = 1; }
            wave_sync();
            if (uni(doamp)) {
                float m = 0.f;
                const int st = L.start[sfb], w = L.width[sfb];
                for (int i = st + lane; i < st + w; i += LHIP_NL) {
                    const float v = (float)((double)L.xrpow[i] * amp);
                    L.xrpow[i] = v;
                    m = fmax_nonneg(m, v);
                }
                m = wave_maxf_pos(m);
                if ((double)m > g.xrpow_max) g.xrpow_max = m;
            }
        }
        wave_sync();
    }
    return 0;
}

// balance_noise (Quantize.js:793-846); returns 1 to continue the outer loop
LHIP_DEV int q_balance_noise(const Tables& T, GI& g, int32_t* scalefac, int lane, QuantLds& L, const QuantTabs& Q) {
    lane = fresh_lane(lane);
    unsigned long long tb_ = PH_NOW(); (void)tb_;
    int all_nonzero, pre_bad, m12;
    q_amp_scalefac_bands(T, g, scalefac, &all_nonzero, &pre_bad, &m12, lane, L, Q);
    PH_MARKB(L, PH_B_AMP, tb_);
    int status = all_nonzero;                                  // loop_break (Quantize.js:453-460)
    if (status) return 0;
    status = T.mode_gr == 2 ? q_scale_bitcount_from(Q, g, scalefac, pre_bad, m12, lane) : q_scale_bitcount_lsf(g, scalefac, lane);
    PH_MARKB(L, PH_B_BITCOUNT, tb_);
    if (!status) return 1;
    if (T.noise_shaping > 1) {
        if (0 == g.scalefac_scale) {
            q_inc_scalefac_scale(T, g, scalefac, lane, L, Q);
            status = 0;
        } else if (g.block_type == SHORT_TYPE && T.subblock_gain > 0) {
            status = (q_inc_subblock_gain(T, g, scalefac, lane, L, Q) || q_loop_break(g, scalefac, lane, L, Q));
        }
    }
    if (!status) status = q_scale_bitcount_any(T, Q, g, scalefac, lane);
    return !status;
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

// The kept quantization's scalars live in LDS while the outer loop runs (only part2_3_length is read there): the loop carries
// one GrInfo (the working copy) in scalar registers instead of two, which is most of what used to spill.  Fields the loop never
// changes (block type, band limits, max_nonzero_coeff) are the same in both copies and are not stored.
LHIP_DEV void gi_keep_store(QuantLds& L, const GI& w, int lane) {
    lane = fresh_lane(lane);
    if (lane == 0) {
        int32_t* k = L.gkeep;
        k[0] = w.part2_3_length; k[1] = w.big_values; k[2] = w.count1; k[3] = w.global_gain; k[4] = w.scalefac_compress;
        k[5] = w.table_select[0]; k[6] = w.table_select[1]; k[7] = w.table_select[2];
        k[8] = w.subblock_gain[0]; k[9] = w.subblock_gain[1]; k[10] = w.subblock_gain[2]; k[11] = w.subblock_gain[3];
        k[12] = w.region0_count; k[13] = w.region1_count; k[14] = w.preflag; k[15] = w.scalefac_scale; k[16] = w.count1table_select;
        k[17] = w.part2_length; k[18] = w.count1bits;
        union { double d; int32_t i[2]; } u; u.d = w.xrpow_max; k[19] = u.i[0]; k[20] = u.i[1];
    }
}
LHIP_DEV void gi_keep_load(const QuantLds& L, GI& g) {
    const int32_t* k = L.gkeep;
    g.part2_3_length = uni(k[0]); g.big_values = uni(k[1]); g.count1 = uni(k[2]); g.global_gain = uni(k[3]); g.scalefac_compress = uni(k[4]);
    g.table_select[0] = uni(k[5]); g.table_select[1] = uni(k[6]); g.table_select[2] = uni(k[7]);
    g.subblock_gain[0] = uni(k[8]); g.subblock_gain[1] = uni(k[9]); g.subblock_gain[2] = uni(k[10]); g.subblock_gain[3] = uni(k[11]);
    g.region0_count = uni(k[12]); g.region1_count = uni(k[13]); g.preflag = uni(k[14]); g.scalefac_scale = uni(k[15]); g.count1table_select = uni(k[16]);
    g.part2_length = uni(k[17]); g.count1bits = uni(k[18]);
    union { double d; int32_t i[2]; } u; u.i[0] = uni(k[19]); u.i[1] = uni(k[20]); g.xrpow_max = u.d;
}

// outer_loop (Quantize.js:871-1052).  g = kept copy (cod_info); seeds in/out via start/step.
// bin_search_StepSize (Quantize.js:322-381) + outer_loop (Quantize.js:871-1052) as ONE state machine, so that the
// three big building blocks -- count_bits, calc_noise, balance_noise -- are each instantiated exactly once
// (code size decides instruction-cache behaviour here).  Everything runs on the working copy `w`
// (ixw / sfw); `g` (kept spectrum in HBM / sfb) is the kept quantization, exactly the reference's cod_info / cod_info_w pair
// with the roles of the first copy swapped (bin search on w, then g = w).
// `kept` (HBM, this granule-channel's slot of W.l3) receives the quantized spectrum of the kept copy whenever a better
// quantization is found; it is read back into L.ixw once the loop has finished (the working copy is dead then).
// `dig` / `dn`: the granule-channel's validation digest (word w at dig[w * dn]; lhip_layout.h VD_*): the bin-search memo in the compact,
// coalesced form the one-thread-per-frame validation reads
// `cs` (latency kernels only): the record through which a helper wave takes the Huffman count of the outer loop's evaluations while this wave
// runs calc_noise beside it (q_count_bits_piped); nullptr: everything on this wave
// bin_search_StepSize's transition (Quantize.js:340-381) as a pure function: the gain evaluated after `gain`, given whether that evaluation's bits exceeded the
// budget (`over`; `equal`: they met it exactly); -1: the search ends with this evaluation.  st_bsup: the step-up loop behind the search (Quantize.js:375-379).
LHIP_DEV int bs_next_gain(int st_bsup, int gain, int CurrentStep, int flagGoneOver, int Direction, int over, int equal) {
    if (!st_bsup) {
        if (!(CurrentStep == 1 || equal)) {
            int step;
            if (over) { if (Direction == 2) flagGoneOver = 1; if (flagGoneOver) CurrentStep /= 2; step = CurrentStep; }
            else { if (Direction == 1) flagGoneOver = 1; if (flagGoneOver) CurrentStep /= 2; step = -CurrentStep; }
            int g = gain + step;
            if (g < 0) g = 0;
            if (g > 255) g = 255;
            return g;
        }
    }
    return (over && gain < 255) ? gain + 1 : -1;
}
#if defined(LHIP_HOSTSIM)
// LAMEJS_SEARCH_STATS=1 (host simulations; tools/search_stats.py): the shape of the search per granule-channel -- evaluations of the bin search and of the
// step-up after it, and, per outer-loop round, the length of the `while (bits > huff_bits) gain++` run in front of the evaluation that fits, with how often
// PrevNoise.sfb_count1 (what the next evaluation's 0/1 shortcut is decided with) changed inside a run.  Prices evaluating the gains of a run side by side.
struct SearchStats {
    long bs[40] = {0}, bsup[40] = {0}, run[40] = {0}, run_cnt1_moved[40] = {0}, rounds = 0, evals = 0, searches = 0, run_steps = 0, run_steps_cnt1_moved = 0, run_steps_zo = 0;
    int prev_nbs = 3;                  // (rule 3: the length of the previous search decides between "the answer is the seed" and "keep walking")
    long pred[4][2] = {{0}};            // bin search: predictions of the next gain, rule x (miss, hit): 0 "over iff gain < seed", 1 "over iff the last move was down", 2 "over iff gain <= seed"
    long bsdir[8][3][2] = {{{0}}};      // bin search: evaluation index (0 .. 7) x direction of the last move (0 none, 1 up, 2 down) x this evaluation's verdict (0: bits <= desired, 1: over)
    long mfail[24] = {0}, mfit[24] = {0};      // first evaluation of a round by its margin (huff_bits - the last evaluation's bits), buckets of 8 bits
    bool on = getenv("LAMEJS_SEARCH_STATS") != nullptr;
    ~SearchStats() {
        if (!on) return;
        fprintf(stderr, "search stats: %ld granule-channel searches, %ld evaluations, %ld outer-loop rounds; %ld evaluations inside gain++ runs (%ld of them with sfb_count1 moved by the evaluation before, %ld with a live 0/1 shortcut)\n", searches, evals, rounds, run_steps, run_steps_cnt1_moved, run_steps_zo);
        auto pr = [](const char* nm, const long* h) { fprintf(stderr, "  %s:", nm); for (int i = 0; i < 40; i++) if (h[i]) fprintf(stderr, " %d:%ld", i, h[i]); fprintf(stderr, "\n"); };
        pr("bin-search evaluations per search", bs); pr("step-up evaluations after it", bsup); pr("gain++ run length per outer-loop round (0 = the first evaluation fits)", run);
        pr("runs in which sfb_count1 moved, by run length", run_cnt1_moved);
        fprintf(stderr, "  next-gain predictions (miss/hit): over iff gain < seed %ld/%ld | over iff last move down %ld/%ld | over iff gain <= seed %ld/%ld | previous search <= 4 evaluations ? gain < seed : same direction again %ld/%ld\n", pred[0][0], pred[0][1], pred[1][0], pred[1][1], pred[2][0], pred[2][1], pred[3][0], pred[3][1]);
        fprintf(stderr, "  bin-search evaluations, index / last move (- up down): under/over:"); for (int i = 0; i < 8; i++) for (int d = 0; d < 3; d++) if (bsdir[i][d][0] + bsdir[i][d][1]) fprintf(stderr, " %d%c:%ld/%ld", i, "-ud"[d], bsdir[i][d][0], bsdir[i][d][1]); fprintf(stderr, "\n");
        fprintf(stderr, "  first evaluation of a round, by margin huff_bits - previous bits (bucket of 8 bits: fails / fits):"); for (int i = 0; i < 24; i++) if (mfail[i] + mfit[i]) fprintf(stderr, " %d:%ld/%ld", 8 * (i - 4), mfail[i], mfit[i]); fprintf(stderr, "\n");
    }
};
inline SearchStats& search_stats() { static SearchStats t; return t; }
#define LHIP_SS(...) do { if (lane == 0 && search_stats().on) { SearchStats& ss_ = search_stats(); __VA_ARGS__; } } while (0)
#else
#define LHIP_SS(...) do { } while (0)
#endif
LHIP_DEV void q_outer_loop(const Tables& T, GI& g, int targ_bits, int bs_start, int bs_step, int* bs_gain_out,
                           int16_t* kept, GrSide* rec, uint32_t* dig, int64_t dn, int lane, QuantLds& L, const QuantTabs& Q, CountShare* cs = nullptr,
                           CandShare* cd = nullptr, const unsigned char* lds0 = nullptr, int lds_stride = 0, int cs_stride = 0) {
    // (cs_stride: the count helper's LDS record lies cs_stride bytes behind this wave's -- where a look-ahead evaluation of the bin search leaves its quantized lines)
    // (candidate helpers: `cd` is the workgroup's array of records and `lds0` its first wave's LDS -- constants of the kernel, not values this loop has to keep alive;
    //  what depends on the wave is formed from its index where it is needed: record cd[wave], helper record of wave 5 + 2 wave (two channels) / 3 (one))
#define CAND_REC() (cd[wg_wave_id()])
#define CAND_LH() (*(const QuantLds*)(lds0 + (size_t)(T.channels_out == 2 ? 5 + 2 * wg_wave_id() : 3) * (size_t)lds_stride))
    lane = fresh_lane(lane);
    enum { ST_BS, ST_BSUP, ST_A, ST_B };
    NoiseRes best, ni;
    PrevNoise pn; pn.gain = 0; pn.sfb_count1 = 0;
    best.max_noise = 0; best.over_count = 0; best.over_SSD = 0; best.bits = 0;
    LHIP_LANE_ONCE(i, 0, (SFBMAX) + 1) { L.pn_step[i] = 0; L.pn_dist[i] = 0.f; L.pn_x[i] = 1.0; L.pn_cls[i] = 0; L.distort[i] = 0.f; }
    wave_sync();
    targ_bits = uni(targ_bits); bs_start = uni(bs_start); bs_step = uni(bs_step);
    uni_gi(g);
    GI w = g;
    // bin-search state
    int CurrentStep = bs_step, flagGoneOver = 0, Direction = 0;
    const int desired_rate = targ_bits - w.part2_length;
    w.global_gain = bs_start;
    // outer-loop state
    int best_part2_3_length = 9999999, age = 0, maxggain = 255, huff_bits = 0, first = 1;
    int kept_p23 = g.part2_3_length;                        // cod_info.part2_3_length (the one kept field the loop reads)
    const int search_limit = 3;
    int st = ST_BS, nbs = 0;
    // Bin search with a look-ahead on the count helper (latency kernels; LHIP_BS_AHEAD): while this wave evaluates a gain, the helper -- idle during the bin search --
    // evaluates the gain the search is most likely to ask for next (bs_next_gain under a predicted verdict: "the answer is the seed" after a short search, "keep walking"
    // after a long one; right for 96 % of the evaluations of a steady two-channel stream, 78 % one channel, 53 - 60 % on material whose gains move, tools/search_stats.py).
    // count_bits during the bin search is a pure function of the gain, so a prediction that comes true IS the reference's evaluation: its reply is taken like a count
    // helper's, its quantized lines are copied if it ends the search.  ahead: the gain under evaluation on the helper + 1 (0: none); ahead_ix: the last evaluation's lines are the helper's.
    const bool bs_ahead = LHIP_BS_AHEAD && LHIP_NL != 1 && cs != nullptr && cs_stride != 0;
    // (one word of loop state -- every scalar this loop carries is paid for in every iteration: bits 0-8 the gain under evaluation on the helper + 1 (0: none), 9 the last
    //  evaluation's lines are the helper's, 10 the previous search of this wave was a long one ("keep walking"), 11 the helper is listening)
    enum { AF_GAIN = 511, AF_IX = 512, AF_WALK = 1024, AF_HERE = 2048 };
    int aflags = 0;
#if LHIP_NL != 1
    if (bs_ahead) { if (lane == 0) { union { double d; int32_t i[2]; } u; u.d = w.xrpow_max; cs->xmax[0] = u.i[0]; cs->xmax[1] = u.i[1]; } if (uni(L.gkeep[23]) > 4) aflags |= AF_WALK; }      // (gkeep[23]: the length of this wave's previous search)
#endif
    int ss_nbs = 0, ss_nup = 0, ss_run = 0, ss_moved = 0, ss_first_ = 0, ss_prev_bits_ = 0; (void)ss_nbs; (void)ss_nup; (void)ss_run; (void)ss_moved; (void)ss_first_; (void)ss_prev_bits_;
    LHIP_SS(ss_.searches++);
#ifndef LHIP_CAND_MARGIN
#define LHIP_CAND_MARGIN 32
#endif
    // candidate helpers (q_cand_*), one word of loop state: bits 0-8 the gain they are evaluating beside this wave's evaluation + 1 (0: none posted), 9 the slow
    // state is posted (q_cand_sync), 10 both helpers have arrived, 11 the next evaluation is the first of its round
    enum { CF_GAIN = 511, CF_SYNCED = 512, CF_HERE = 1024, CF_FIRST = 2048 };
    int cflags = 0;
    for (;;) {
#ifndef LHIP_NO_FORCE_UNI
        uni_gi(w); kept_p23 = uni(kept_p23); st = uni(st); CurrentStep = uni(CurrentStep); flagGoneOver = uni(flagGoneOver); Direction = uni(Direction);
        best_part2_3_length = uni(best_part2_3_length); age = uni(age); maxggain = uni(maxggain); huff_bits = uni(huff_bits); first = uni(first);
        pn.gain = uni(pn.gain); pn.sfb_count1 = uni(pn.sfb_count1);
        best.max_noise = unid(best.max_noise); best.over_count = uni(best.over_count); best.over_SSD = uni(best.over_SSD); best.bits = uni(best.bits);
#endif
        int asg = 0;
        const int cnt1_seen = pn.sfb_count1;                  // what this evaluation's 0/1 shortcut is decided with
        int nBits, spec = 0;
        NoiseCommit nc; nc.fresh = 0; nc.step = 0; nc.cls = 0; nc.dist = 0.f; nc.x = 0.0;
#if LHIP_NL != 1
        int taken = 0;
        if (bs_ahead) {
            aflags = uni(aflags);
            if (st <= ST_BSUP) {
                if ((aflags & AF_GAIN) == w.global_gain + 1) {     // the evaluation at this gain was made beside the last one
                    long n_ = 0;
                    while (wg_load(&cs->state, lane) != CS_DONE) { wg_spin(); LHIP_SPIN_GUARD(n_); }
                    wg_acquire();
                    const uint32_t w0 = (uint32_t)uni((int)cs->w[0]), w1 = (uint32_t)uni((int)cs->w[1]), w2 = (uint32_t)uni((int)cs->w[2]);
                    const int amask = (int)((w2 >> 24) & 15u);
                    nBits = (int)(w0 & 0x1ffffu);
                    aflags &= ~AF_GAIN;
                    if (nBits != (int)LARGE_BITS) {       // (an evaluation that overflows touches neither the fields nor the lines: this wave makes that one itself, q_count_bits)
                        w.count1 = (int)(w0 >> 17); w.big_values = (int)(w1 & 1023u); w.count1bits = (int)((w1 >> 10) & 0x1fffu); w.count1table_select = (int)((w1 >> 23) & 1u);
                        if (amask & 8) { w.region0_count = (int)((w1 >> 24) & 15u); w.region1_count = (int)(w1 >> 28); }
                        if (amask & 1) w.table_select[0] = (int)(w2 & 63u) - 1;
                        if (amask & 2) w.table_select[1] = (int)((w2 >> 6) & 63u) - 1;
                        if (amask & 4) w.table_select[2] = (int)((w2 >> 12) & 63u) - 1;
                        asg = pack_cond_fields(w, amask);
                        taken = 1; aflags |= AF_IX;
                        LHIP_PIPE_COUNT(bs_taken);
                    }
                }
                if (!taken) {
                    // (a request the search did not come to is finished before the next is posted: the helper's reply words are one set)
                    if (aflags & AF_GAIN) { long n_ = 0; while (wg_load(&cs->state, lane) != CS_DONE) { wg_spin(); LHIP_SPIN_GUARD(n_); } }
                    aflags &= ~(AF_GAIN | AF_IX);
                    const int over_ = st == ST_BSUP ? 1 : ((!(aflags & AF_WALK) || Direction == 0) ? (w.global_gain < bs_start) : (Direction == 1));
                    const int nx = bs_next_gain(st == ST_BSUP, w.global_gain, CurrentStep, flagGoneOver, Direction, over_, 0);
                    if (!(aflags & AF_HERE) && wg_load(&cs->here, lane)) aflags |= AF_HERE;       // (a helper still busy with its stage's other work would make this wave wait for it)
                    if ((aflags & AF_HERE) && nx >= 0 && nx != w.global_gain) { wg_store(&cs->state, CS_REQ2 | (nx << 8) | (w.block_type << 16), lane); aflags |= nx + 1; LHIP_PIPE_COUNT(bs_posted); }
                }
            } else if (aflags & AF_GAIN) {        // the first evaluation of the outer loop: the helper must be free before it is asked to count
                long n_ = 0; while (wg_load(&cs->state, lane) != CS_DONE) { wg_spin(); LHIP_SPIN_GUARD(n_); }
                aflags &= ~AF_GAIN;
            }
        }
        if (cd != nullptr && st >= ST_A) {
            cflags = uni(cflags);
            // the evaluation at this gain was made beside the last one (this iteration was reached by `gain++; continue`: nothing but the gain has changed)
            if ((cflags & CF_GAIN) == w.global_gain + 1) {
                CandShare& cdr = CAND_REC();
                const unsigned long long tk0_ = CD_NOW(); (void)tk0_;
                const int b = q_cand_take(cdr, CAND_LH(), w, pn, &ni, &nc, lane, L);
                if (b >= 0) { nBits = b; spec = 1; taken = 1; LHIP_PIPE_COUNT(taken); CD_ADD(cdr, 1, 1); CD_ADD(cdr, 3, CD_NOW() - tk0_); CD_ADD(cdr, 7, CD_NOW() - cdr.t_post); }
            }
            // Posted for the FIRST evaluation of a round only, and only when it is likely not to fit: the bits of the last evaluation, which balance_noise's amplification
            // can only have raised, within LHIP_CAND_MARGIN of the budget (tools/search_stats.py: 85 % of such evaluations fail within 8 bits of it, 40 % at 24 - 32,
            // under 20 % beyond 56; a request nobody takes costs the owner ~ 1 k cycles of contention, a taken one saves ~ 3 k)
            const int cand_due = (cflags & CF_FIRST) && (huff_bits - w.part2_3_length) < LHIP_CAND_MARGIN;
            cflags &= ~(CF_GAIN | CF_FIRST);
            if (!taken && cand_due && w.global_gain < 255 && !(w.xrpow_max * ipow20(Q, w.global_gain) > (double)IXMAX_VAL * (1.0 - 0x1p-50))) {
                CandShare& cdr = CAND_REC();
                if (!(cflags & CF_HERE) && q_cand_present(cdr, lane)) cflags |= CF_HERE;
                if ((cflags & CF_HERE) && q_cand_free(cdr, lane)) {
                    if (!(cflags & CF_SYNCED)) { q_cand_sync(cdr, w, lane); cflags |= CF_SYNCED; }
#if defined(LHIP_HANDOFF_PROF) && !defined(LHIP_HOSTSIM)
                    if (lane == 0) cdr.t_post = CD_NOW();
#endif
                    q_cand_post(cdr, w.global_gain, pn.sfb_count1, best.over_count == 0, pn.gain, lane);
                    cflags |= w.global_gain + 2;
                    LHIP_PIPE_COUNT(posted); CD_ADD(cdr, 0, 1);
                } else CD_ADD(cdr, 2, 1);
            }
        }
        if (taken) { }
        else if (cs != nullptr && st >= ST_A) nBits = q_count_bits_piped(T, w, L.sfw, L.ixw, pn, &asg, *cs, &ni, &nc, best.over_count == 0, &spec, lane, L, Q);
        else
#endif
        nBits = q_count_bits(T, w, L.sfw, L.ixw, st >= ST_A, pn, &asg, lane, L, Q);   // the only call site (but for the two-wave form above)
        LHIP_SS(ss_.evals++; if (st == ST_BS) ss_nbs++; else if (st == ST_BSUP) ss_nup++;
                if (st >= ST_A && ss_run > 0) { ss_.run_steps++; if (ss_moved) ss_.run_steps_cnt1_moved++; if (cnt1_seen > 0) ss_.run_steps_zo++; });
        // memo of the bin search: collected in LDS and written to the side record in one burst when the search ends (a global
        // store per step would sit in front of every later memory wait of the wave -- the VMEM counter retires in order)
        if (st <= ST_BSUP && nbs < BS_TAB_MAX) { if (lane == 0) { L.memo.bs_tab[nbs] = (w.global_gain << 24) | nBits; L.memo.bs_asg[nbs] = asg; } nbs++; }
        LHIP_SS(if (st <= ST_BSUP) { const int ov_ = nBits > desired_rate, eq_ = nBits == desired_rate;
                    const int act_ = bs_next_gain(st == ST_BSUP, w.global_gain, CurrentStep, flagGoneOver, Direction, ov_, eq_);
                    const int p0_ = bs_next_gain(st == ST_BSUP, w.global_gain, CurrentStep, flagGoneOver, Direction, w.global_gain < bs_start, 0);
                    const int p1_ = bs_next_gain(st == ST_BSUP, w.global_gain, CurrentStep, flagGoneOver, Direction, Direction == 2, 0);
                    const int p2_ = bs_next_gain(st == ST_BSUP, w.global_gain, CurrentStep, flagGoneOver, Direction, w.global_gain <= bs_start, 0);
                    const int ov3_ = (ss_.prev_nbs <= 4 || Direction == 0) ? (w.global_gain < bs_start) : (Direction == 1);
                    const int p3_ = bs_next_gain(st == ST_BSUP, w.global_gain, CurrentStep, flagGoneOver, Direction, st == ST_BSUP ? 1 : ov3_, 0);
                    if (act_ >= 0) { ss_.pred[0][p0_ == act_]++; ss_.pred[1][p1_ == act_]++; ss_.pred[2][p2_ == act_]++; ss_.pred[3][p3_ == act_]++; } });
        LHIP_SS(if (st <= ST_BSUP) { const int i_ = (ss_nbs + ss_nup - 1) < 7 ? (ss_nbs + ss_nup - 1) : 7; ss_.bsdir[i_ < 0 ? 0 : i_][st == ST_BSUP ? 1 : Direction][nBits > desired_rate]++; });
        if (st == ST_BS) {
            if (CurrentStep == 1 || nBits == desired_rate) st = ST_BSUP;
            else {
                int step;
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
                w.global_gain += step;
                if (w.global_gain < 0) { w.global_gain = 0; flagGoneOver = 1; }
                if (w.global_gain > 255) { w.global_gain = 255; flagGoneOver = 1; }
                continue;
            }
        }
        if (st == ST_BSUP) {
            if (nBits > desired_rate && w.global_gain < 255) { w.global_gain++; continue; }
            w.part2_3_length = nBits;
#if LHIP_NL != 1
            if (bs_ahead) {
                if ((aflags & AF_IX) && nBits != (int)LARGE_BITS) {        // the search's last evaluation was the helper's: its quantized lines are what the loop goes on with
                    const QuantLds& Lh1 = *(const QuantLds*)((const unsigned char*)&L + cs_stride);
                    uint32_t kw[NPL];
#pragma unroll
                    for (int j = 0; j < NPL; j++) { const int i = lane + LHIP_NL * j; kw[j] = ((const uint32_t*)Lh1.ixw)[i < 288 ? i : 287]; }
#pragma unroll
                    for (int j = 0; j < NPL; j++) { const int i = lane + LHIP_NL * j; if (i < 288) ((uint32_t*)L.ixw)[i] = kw[j]; }
                    wave_sync();
                }
                if (lane == 0) L.gkeep[23] = nbs;                   // the next search of this wave: "walk" or "the answer is the seed"
            }
#endif
            LHIP_SS(ss_.bs[ss_nbs < 39 ? ss_nbs : 39]++; ss_.bsup[ss_nup < 39 ? ss_nup : 39]++; ss_.prev_nbs = ss_nbs);
            *bs_gain_out = w.global_gain;                    // OldValue[ch] after this granule
            wave_sync();
            LHIP_LANE_ONCE(i, 0, nbs) {
                const int32_t e = L.memo.bs_tab[i], a = L.memo.bs_asg[i];
                rec->bs_tab[i] = e; rec->bs_asg[i] = a;
                if (i < VD_ENT) { dig[(VD_TAB + 2 * i) * dn] = (uint32_t)e; dig[(VD_TAB + 2 * i + 1) * dn] = (uint32_t)a; }
            }
            if (lane == 0) {
                const int state = pack_cond_fields(w, 0);
                rec->bs_ntab = nbs; rec->bs_state = state;
                dig[VD_TARG * dn] = (uint32_t)targ_bits | ((uint32_t)nbs << 24); dig[VD_STATE * dn] = (uint32_t)state;
            }
            wave_sync();
            if (0 == T.noise_shaping) {
#if LHIP_NL != 1
                if (bs_ahead && (uni(aflags) & AF_GAIN)) { long n_ = 0; while (wg_load(&cs->state, lane) != CS_DONE) { wg_spin(); LHIP_SPIN_GUARD(n_); } }
#endif
                g = w;
                for (int i = lane; i < 288; i += LHIP_NL) ((uint32_t*)kept)[i] = ((const uint32_t*)L.ixw)[i];
                LHIP_LANE_ONCE(i, 0, (SFBMAX) + 1) L.sfb[i] = L.sfw[i];
                wave_sync();
                return;
            }
        } else if (st == ST_A) {
            w.part2_3_length = nBits;
            LHIP_SS(if (ss_run == 0 && ss_first_) { int b_ = (huff_bits - ss_prev_bits_) / 8 + 4; b_ = b_ < 0 ? 0 : b_ > 23 ? 23 : b_; if (nBits > huff_bits) ss_.mfail[b_]++; else ss_.mfit[b_]++; } ss_first_ = 0);
            if (nBits > huff_bits && w.global_gain <= maxggain) { LHIP_SS(ss_run++; ss_moved = (pn.sfb_count1 != cnt1_seen)); w.global_gain++; continue; }
            if (w.global_gain > maxggain) break;
            if (best.over_count == 0) {
                // The reference's second loop (Quantize.js:1004-1013) starts by counting again at the SAME gain.  That evaluation
                // has the same inputs as the one just made -- spectrum, gain, scalefactors, the noise cache (only calc_noise
                // writes it) -- except PrevNoise.sfb_count1, which the one just made has rewritten and which steers the 0/1
                // shortcut of quantize_xrpow.  If it was rewritten with the value it had, the second evaluation is the first
                // one again (same bits, same assignments, same spectrum) and is not made.
                st = ST_B;
                if (pn.sfb_count1 != cnt1_seen) continue;
                if (nBits > best_part2_3_length && w.global_gain <= maxggain) { LHIP_SS(ss_run++; ss_moved = (pn.sfb_count1 != cnt1_seen)); w.global_gain++; continue; }
                if (w.global_gain > maxggain) break;
            }
        } else {   // ST_B
            w.part2_3_length = nBits;
            if (nBits > best_part2_3_length && w.global_gain <= maxggain) { LHIP_SS(ss_run++; ss_moved = (pn.sfb_count1 != cnt1_seen)); w.global_gain++; continue; }
            if (w.global_gain > maxggain) break;
        }
        LHIP_SS(if (st >= ST_A) { ss_.rounds++; ss_.run[ss_run < 39 ? ss_run : 39]++; }  ss_run = 0; ss_moved = 0);
        if (spec) { q_noise_commit(w, nc, pn, lane, L); LHIP_PIPE_COUNT(committed); }          // made beside the count, on exactly these inputs: now it counts
        else q_calc_noise(T, w, L.sfw, L.ixw, &ni, 1, pn, best.over_count == 0, lane, L, Q);                 // the only call site
        ni.bits = w.part2_3_length;
        int keep;
        if (first) keep = 1;
        else keep = q_quant_compare(best, ni);
        if (keep) {
            if (!first) best_part2_3_length = kept_p23;           // value BEFORE the copy (reference quirk)
            best = ni;
            gi_keep_store(L, w, lane);                            // cod_info = cod_info_w
            kept_p23 = w.part2_3_length;
            for (int i = lane; i < 288; i += LHIP_NL) ((uint32_t*)kept)[i] = ((const uint32_t*)L.ixw)[i];
            LHIP_LANE_ONCE(i, 0, (SFBMAX) + 1) L.sfb[i] = L.sfw[i];
            wave_sync();
            age = 0;
        } else if (T.full_outer_loop == 0) {
            if (++age > search_limit && best.over_count == 0) break;
        }
        if (!first && !((w.global_gain + w.scalefac_scale) < 255)) break;
        first = 0;
        int bal_;
        { PH_BEGIN(); bal_ = q_balance_noise(T, w, L.sfw, lane, L, Q); PH_END(L, PH_BALANCE); }       // the only call site
        if (!bal_) break;
        cflags = (cflags & ~CF_SYNCED) | CF_FIRST;                // (preflag, scalefac_scale, subblock gains, xrpow_max may have moved)
        maxggain = (w.scalefac_scale != 0) ? 254 : 255;
        huff_bits = targ_bits - w.part2_length;
        if (huff_bits <= 0) break;
        LHIP_SS(ss_first_ = 1; ss_prev_bits_ = w.part2_3_length);
        st = ST_A;
    }
    wave_sync();
#if LHIP_NL != 1
    if (bs_ahead && (uni(aflags) & AF_GAIN)) { long n_ = 0; while (wg_load(&cs->state, lane) != CS_DONE) { wg_spin(); LHIP_SPIN_GUARD(n_); } }      // (nobody may post to a busy helper)
#endif
    { const GI inv = w; g = inv; gi_keep_load(L, g); }       // the invariant fields are the working copy's, the rest comes back from LDS
#undef CAND_REC
#undef CAND_LH
}

// ---------------------------------------------------------------------------------------------
// iteration_finish_one: best_scalefac_store (Takehiro.js:862-943), scfsi_calc (809-855),
// best_huffman_divide (727-800)
// ---------------------------------------------------------------------------------------------
LHIP_DEV void q_best_scalefac_store(const Tables& T, GI& g, int gr, int ch, int gr0_block_type, int* scfsi /*[4]*/,
                                    int lane, QuantLds& L, const QuantTabs& Q) {
    lane = fresh_lane(lane);
    int32_t* sf = L.sfb;
    int recalc = 0;
    {
        // bands whose quantized lines are all zero get scalefactor -2 (Takehiro.js:870-882): every lane flags the
        // bands of its non-zero pairs (a pair never straddles a band), then one lane per band reads its flag
        LHIP_LANE_ONCE(sfb, 0, (SFBMAX) + 1) L.qmode[sfb] = 0;
        wave_sync();
        const uint8_t* l2s = line2sfb(Q, g.block_type);
        for (int p = 2 * lane; p < 576; p += 2 * LHIP_NL)
            if (*(const uint32_t*)(L.ixw + p) != 0) L.qmode[l2s[p]] = 1;
        wave_sync();
        int any = 0;
        LHIP_LANE_ONCE(sfb, 0, g.sfbmax) if (!L.qmode[sfb]) { sf[sfb] = -2; any = 1; }
        if (wave_any(any)) recalc = -2;
        wave_sync();
    }
    if (0 == g.scalefac_scale && 0 == g.preflag) {
        int s = 0;
        LHIP_LANE_ONCE(sfb, 0, g.sfbmax) if (sf[sfb] > 0) s |= sf[sfb];
        s = wave_or(s);
        if (0 == (s & 1) && s != 0) {
            LHIP_LANE_ONCE(sfb, 0, g.sfbmax) if (sf[sfb] > 0) sf[sfb] >>= 1;
            g.scalefac_scale = recalc = 1;
            wave_sync();
        }
    }
    if (0 == g.preflag && g.block_type != SHORT_TYPE && T.mode_gr == 2) {
        int bad = 0;
        LHIP_LANE_ONCE(sfb, 11, SBPSY_l) if (sf[sfb] < T.pretab[sfb] && sf[sfb] != -2) bad = 1;
        if (!wave_any(bad)) {
            LHIP_LANE_ONCE(sfb, 11, SBPSY_l) if (sf[sfb] > 0) sf[sfb] -= T.pretab[sfb];
            g.preflag = recalc = 1;
            wave_sync();
        }
    }
    for (int i = 0; i < 4; i++) scfsi[i] = 0;
    if (T.mode_gr == 2 && gr == 1 && uni(gr0_block_type) != SHORT_TYPE && g.block_type != SHORT_TYPE) {
        // scfsi_calc (Takehiro.js:809-855), one lane per scalefactor band: a group is shared with granule 0 when no
        // band of it differs (bands already marked negative do not count as different)
        const int8_t* g0 = L.sf_gr0[ch];
        uint64_t m_diff = 0;
        LHIP_LANE_ONCE(sfb, 0, SBPSY_l) if ((int)g0[sfb] != sf[sfb] && sf[sfb] >= 0) m_diff |= 1ull << sfb;
        m_diff = wave_lane_bits(m_diff);
        uint64_t m_same = 0;
        for (int i = 0; i < 4; i++) {
            const int b0 = T.scfsi_band[i], b1 = T.scfsi_band[i + 1];
            const uint64_t rng = ((1ull << b1) - 1) & ~((1ull << b0) - 1);
            if ((m_diff & rng) == 0) { scfsi[i] = 1; m_same |= rng; }
        }
        wave_sync();
        LHIP_LANE_ONCE(sfb, 0, SBPSY_l) if ((m_same >> sfb) & 1) sf[sfb] = -1;
        wave_sync();
        // slen1_n / slen2_n are powers of two: the two maxima of the reference reduce to ORs (negative markers count as 0)
        uint64_t m_cnt = 0;
        int or12 = 0;
        LHIP_LANE_ONCE(sfb, 0, SBPSY_l) {
            const int v = sf[sfb];
            if (v != -1) m_cnt |= 1ull << sfb;
            const int vp = v > 0 ? v : 0;
            or12 |= (sfb < 11) ? vp : (vp << 8);
        }
        m_cnt = wave_lane_bits(m_cnt);
        or12 = wave_or(or12);
        const int c1 = __builtin_popcountll(m_cnt & 0x7ffull), c2 = __builtin_popcountll(m_cnt >> 11);
        const int s1 = or12 & 0xff, s2 = or12 >> 8;
        int best = 0x7fffffff;                         // first i with the smallest cost == minimum of (cost, i)
        LHIP_LANE_ONCE(i, 0, 16)
            if (s1 < T.slen1_n[i] && s2 < T.slen2_n[i]) { const int c = (T.slen1_tab[i] * c1 + T.slen2_tab[i] * c2) * 16 + i; if (c < best) best = c; }
        best = wave_min(best);
        if (best != 0x7fffffff && g.part2_length > (best >> 4)) { g.part2_length = best >> 4; g.scalefac_compress = best & 15; }
        recalc = 0;
    }
    wave_sync();
    LHIP_LANE_ONCE(sfb, 0, g.sfbmax) if (sf[sfb] == -2) sf[sfb] = 0;
    wave_sync();
    if (uni(recalc) != 0) q_scale_bitcount_any(T, Q, g, sf, lane);
}

// Huffman statistics per scalefactor band over the pairs below `limit`, then turned into prefix sums over
// bands so that any union of whole bands [b0, b1) costs O(1):
// row 0 max (kept per band), 1 t1, 2 table23 (packed), 3 table56 (packed), 4..6 t7-9, 7..9 t10-12, 10..12 t13-15,
// 13 largetbl hi, 14 largetbl lo, 15 number of escaped values.  A row is only meaningful for regions whose
// maximum admits the table group, which is exactly when the reference would look at it.
LHIP_DEV void q_band_max_tables(int lane, QuantLds& L);
LHIP_DEV void q_band_stats(const Tables& T, const int16_t* ix, int limit, int lane, QuantLds& L, const QuantTabs& Q) {
    lane = fresh_lane(lane);
    // One lane per run of consecutive pairs (5 on the device): every pair contributes its 17 code lengths, the
    // lane keeps running sums, an exclusive wave scan of the lane totals turns them into prefix sums over ALL
    // pairs, and the lanes owning the last pair of a band publish the prefix there.  Sums are packed two per
    // word (a total is at most 288 x 21 < 2^16).  A length is only added when the PAIR admits the table; a row
    // is only read for regions whose maximum admits it, where that is the same thing.
    enum { PPL = (288 + LHIP_NL - 1) / LHIP_NL, NW = 9 };
    LHIP_LANE_ONCE(bnd, 0, (SBMAX_l + 1) + 1) L.hd.bstat[0][bnd] = 0;
    wave_sync();
    uint32_t pre[PPL][NW];
    uint32_t run[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) run[w] = 0;
#pragma unroll
    for (int j = 0; j < PPL; j++) {
        const int q = PPL * lane + j, p = 2 * q;
        if (q < 288 && p < limit) {
            const uint32_t w2 = *(const uint32_t*)(ix + p);
            const int x = (int)(w2 & 0xffffu), y = (int)(w2 >> 16), m = x > y ? x : y;
            if (m > 0) lds_max(&L.hd.bstat[0][Q.l2s_long[p] + 1], m);
            const int xc = x < 15 ? x : 15, yc = y < 15 ? y : 15, e = xc * 16 + yc;
            uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0;
            if (m <= 1) c0 = Q.hlen[HL_T1 + x * 2 + y];
            if (m <= 2) { const int k = x * 3 + y; c0 |= (uint32_t)Q.hlen[HL_T2 + k] << 16; c1 = Q.hlen[HL_T3 + k]; }
            if (m <= 3) { const int k = x * 4 + y; c1 |= (uint32_t)Q.hlen[HL_T5 + k] << 16; c2 = Q.hlen[HL_T6 + k]; }
            if (m <= 5) { const int k = x * 6 + y; c2 |= (uint32_t)Q.hlen[HL_T7 + k] << 16; c3 = Q.hlen[HL_T8 + k] | ((uint32_t)Q.hlen[HL_T9 + k] << 16); }
            if (m <= 7) { const int k = x * 8 + y; c4 = Q.hlen[HL_T10 + k] | ((uint32_t)Q.hlen[HL_T11 + k] << 16); c5 = Q.hlen[HL_T12 + k]; }
            if (m <= 15) { c5 |= (uint32_t)Q.hlen[HL_T13 + e] << 16; c6 = Q.hlen[HL_T14 + e] | ((uint32_t)Q.hlen[HL_T15 + e] << 16); }
            const uint32_t c7 = Q.hlen[HL_EHI + e] | ((uint32_t)Q.hlen[HL_ELO + e] << 16);
            const uint32_t c8 = (uint32_t)((x > 14) + (y > 14));
            run[0] += c0; run[1] += c1; run[2] += c2; run[3] += c3; run[4] += c4; run[5] += c5; run[6] += c6; run[7] += c7; run[8] += c8;
        }
#pragma unroll
        for (int w = 0; w < NW; w++) pre[j][w] = run[w];
    }
    uint32_t base[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) { int tot; base[w] = (uint32_t)wave_excl_scan((int)run[w], lane, &tot); }
    // word layout: 0: t1 | t2<<16   1: t3 | t5<<16   2: t6 | t7<<16   3: t8 | t9<<16   4: t10 | t11<<16
    //              5: t12 | t13<<16  6: t14 | t15<<16  7: esc_hi | esc_lo<<16   8: escaped values
#pragma unroll
    for (int j = 0; j < PPL; j++) {
        const int q = PPL * lane + j, p = 2 * q;
        if (q < 288 && (p + 2 == 576 || Q.l2s_long[p] != Q.l2s_long[p + 2])) {
            const int col = Q.l2s_long[p] + 1;
            uint32_t v[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) v[w] = base[w] + pre[j][w];
            L.hd.bstat[1][col] = (int)(v[0] & 0xffffu);
            L.hd.bstat[2][col] = (int)((v[0] & 0xffff0000u) | (v[1] & 0xffffu));          // t2 << 16 | t3 (count_bit_noESC_from2 packing)
            L.hd.bstat[3][col] = (int)((v[1] & 0xffff0000u) | (v[2] & 0xffffu));          // t5 << 16 | t6
            L.hd.bstat[4][col] = (int)(v[2] >> 16); L.hd.bstat[5][col] = (int)(v[3] & 0xffffu); L.hd.bstat[6][col] = (int)(v[3] >> 16);
            L.hd.bstat[7][col] = (int)(v[4] & 0xffffu); L.hd.bstat[8][col] = (int)(v[4] >> 16); L.hd.bstat[9][col] = (int)(v[5] & 0xffffu);
            L.hd.bstat[10][col] = (int)(v[5] >> 16); L.hd.bstat[11][col] = (int)(v[6] & 0xffffu); L.hd.bstat[12][col] = (int)(v[6] >> 16);
            L.hd.bstat[13][col] = (int)(v[7] & 0xffffu); L.hd.bstat[14][col] = (int)(v[7] >> 16); L.hd.bstat[15][col] = (int)v[8];
        }
    }
    LHIP_LANE_ONCE(k, 1, 16) L.hd.bstat[k][0] = 0;
    wave_sync();
    q_band_max_tables(lane, L);
}

// Range maxima over whole bands in O(1): hmax[t][b] = maximum of the band maxima (row 0 of the statistics) of the bands
// b .. b + 2^t - 1, bands past SBMAX_l counting as 0; any range is covered by two (overlapping) power-of-two windows.
LHIP_DEV void q_band_max_tables(int lane, QuantLds& L) {
    LHIP_LANE_ONCE(b, 0, 24) L.hmax[0][b] = (int16_t)(b < SBMAX_l ? L.hd.bstat[0][b + 1] : 0);     // quantized values are <= IXMAX_VAL
    wave_sync();
#pragma unroll
    for (int t = 1; t < 5; t++) {
        LHIP_LANE_ONCE(b, 0, 24) {
            const int h = 1 << (t - 1);
            const int a = L.hmax[t - 1][b], c = (b + h < 24) ? (int)L.hmax[t - 1][b + h] : 0;
            L.hmax[t][b] = (int16_t)(a > c ? a : c);
        }
        wave_sync();
    }
}
LHIP_DEV int q_band_range_max(const QuantLds& L, int lo, int hi) {      // bands [lo, hi); 0 for an empty range
    const int len = hi - lo;
    const int t = 31 - (int)__builtin_clz((unsigned)(len > 0 ? len : 1));
    const int a = L.hmax[t][lo < 23 ? lo : 23], c = L.hmax[t][hi - (1 << t) >= 0 ? hi - (1 << t) : 0];
    const int m = a > c ? a : c;
    return len > 0 ? m : 0;
}

// choose_table for the union of whole bands [b0, b1) from the statistics above (bits are added to *bits).
// Called with lane-varying ranges, so it is written branch-free (selects), like the region planning of count_bits.
LHIP_DEV int q_choose_from_stats(const Tables& T, int b0, int b1, int* bits, const QuantLds& L, const QuantTabs& Q) {
    (void)T; (void)Q;
    const int mx = q_band_range_max(L, b0, b1);
    const int kind = (mx == 0) ? 0 : (mx == 1) ? 1 : (mx <= 3) ? 2 : (mx <= 15) ? 4 : (mx <= IXMAX_VAL) ? 5 : 6;
    const int t1 = (mx <= 1) ? mx : (mx == 2) ? 2 : (mx == 3) ? 5 : (mx <= 5) ? 7 : (mx <= 7) ? 10 : 13;
    const int rowA = (kind == 1) ? 1 : (kind == 2) ? (mx == 2 ? 2 : 3) : (kind == 4) ? (mx <= 5 ? 4 : mx <= 7 ? 7 : 10) : 13;
    int choice, choice2, lb1, lb2;
    esc_choice(mx > 15 ? mx - 15 : 1, &choice, &choice2, &lb1, &lb2);
    const int sA = L.hd.bstat[rowA][b1] - L.hd.bstat[rowA][b0];
    const int sB = L.hd.bstat[rowA + 1][b1] - L.hd.bstat[rowA + 1][b0];
    const int sC = L.hd.bstat[rowA + 2][b1] - L.hd.bstat[rowA + 2][b0];
    int c0 = sA, c1 = sB, c2 = sC, ta = t1, tb = t1 + 1;
    if (kind == 2) { c0 = (int)((unsigned)sA >> 16); c1 = sA & 0xffff; }
    if (kind == 5) { c0 = sA + sC * lb1; c1 = sB + sC * lb2; ta = choice; tb = choice2; }
    int bsum = c0, t = ta;
    if ((kind == 2 || kind == 4 || kind == 5) && bsum > c1) { bsum = c1; t = tb; }
    if (kind == 4 && bsum > c2) { bsum = c2; t = t1 + 2; }
    if (kind == 0) { bsum = 0; t = 0; }
    if (kind == 6) { *bits = LARGE_BITS; return -1; }
    *bits += bsum;
    return t;
}

// recalc_divide_sub (Takehiro.js:698-725); region 2 = bands r2.. up to big_values, from the band statistics.
// The cost of region 2 for every split point r2 is independent of the loop state, so lane r2 evaluates it; the
// reference's running comparison (its `break`s depend on the best length so far) is then replayed over those.
LHIP_DEV void q_recalc_divide_sub(const Tables& T, const GI& c2, GI& g, int lane, QuantLds& L, const QuantTabs& Q) {
    const int bigv = c2.big_values;
    LHIP_LANE_ONCE(r2, 2, SBMAX_l + 1) {
        int bits2 = 0;
        const int tbl = q_choose_from_stats(T, r2, SBMAX_l, &bits2, L, Q);
        L.r2_bits[r2] = bits2; L.r2_tbl[r2] = tbl;
    }
    wave_sync();
    int best = -1, cur = g.part2_3_length;
    for (int r2 = 2; r2 < SBMAX_l + 1; r2++) {
        if (Q.sfb_l[r2] >= bigv) break;
        int bits = L.r01_bits[r2 - 2] + c2.count1bits;
        if (cur <= bits) break;
        bits += L.r2_bits[r2];
        if (cur <= bits) continue;
        cur = bits; best = r2;
    }
    best = uni(best);
    if (best >= 0) {
        g = c2;
        g.part2_3_length = uni(cur);
        g.region0_count = L.r01_div[best - 2];
        g.region1_count = best - 2 - L.r01_div[best - 2];
        g.table_select[0] = L.r0_tbl[best - 2];
        g.table_select[1] = L.r1_tbl[best - 2];
        g.table_select[2] = L.r2_tbl[best];
    }
    wave_sync();
}

LHIP_DEV void q_best_huffman_divide(const Tables& T, GI& g, int lane, QuantLds& L, const QuantTabs& Q) {
    if (g.block_type == SHORT_TYPE && T.mode_gr == 1) return;     // Takehiro.js:735-737
    lane = fresh_lane(lane);
    const int16_t* ix = L.ixw;
    GI c2 = g;
    if (g.block_type == NORM_TYPE) {
        // recalc_divide_init: every (region0, region1) split evaluated from the per-band statistics
        q_band_stats(T, ix, g.big_values, lane, L, Q);
        const int bigv = g.big_values;
        LHIP_LANE_ONCE(s, 0, 24) { L.r01_bits[s] = LARGE_BITS; L.r01_div[s] = 0; L.r0_tbl[s] = 0; L.r1_tbl[s] = 0; }
        // all 16 x 8 splits in parallel; candidate bits go to cand[r0 + r1][r0]
        for (int c = lane; c < 128; c += LHIP_NL) {
            const int r0 = c >> 3, r1 = c & 7, sidx = r0 + r1;
            if (sidx > 20) continue;
            int bits = LARGE_BITS;
            if (Q.sfb_l[r0 + 1] < bigv && Q.sfb_l[r0 + r1 + 2] < bigv) {
                bits = 0;
                q_choose_from_stats(T, 0, r0 + 1, &bits, L, Q);
                q_choose_from_stats(T, r0 + 1, r0 + r1 + 2, &bits, L, Q);
            }
            L.hd.cand[sidx][r0] = (int16_t)(bits > 0x7fff ? 0x7fff : bits);
        }
        wave_sync();
        // lane s: the first r0 (ascending) with the strictly smallest bits wins, as in the reference's loop order
        LHIP_LANE_ONCE(sidx, 0, (20) + 1) {
            int bb = 0x7fff, bd = -1;               // 0x7fff == no valid split (the reference keeps LARGE_BITS there)
            for (int r0 = (sidx > 7 ? sidx - 7 : 0); r0 < 16 && r0 <= sidx; r0++)
                if (bb > L.hd.cand[sidx][r0]) { bb = L.hd.cand[sidx][r0]; bd = r0; }
            if (bd >= 0) {
                int dummy = 0;
                L.r01_bits[sidx] = bb; L.r01_div[sidx] = bd;
                L.r0_tbl[sidx] = q_choose_from_stats(T, 0, bd + 1, &dummy, L, Q);
                L.r1_tbl[sidx] = q_choose_from_stats(T, bd + 1, sidx + 2, &dummy, L, Q);
            }
        }
        wave_sync();
        q_recalc_divide_sub(T, c2, g, lane, L, Q);
    }
    int i = c2.big_values;
    if (i == 0 || (ix[i - 2] | ix[i - 1]) > 1) return;
    i = g.count1 + 2;
    if (i > 576) return;
    c2 = g;
    c2.count1 = i;
    // quads from count1 + 2 down to the old big_values (all values <= 1 there)
    const int nq = (i - c2.big_values + 3) >> 2;
    int a12 = 0;
    for (int k = lane; k < nq; k += LHIP_NL) {
        const int e = i - 4 * k;
        const int p = ((ix[e - 4] * 2 + ix[e - 3]) * 2 + ix[e - 2]) * 2 + ix[e - 1];
        a12 += Q.t32l[p] + (Q.t33l[p] << 16);
    }
    a12 = wave_sum(a12);
    int a1 = a12 & 0xffff, a2 = a12 >> 16;
    i -= 4 * nq;
    c2.big_values = i;
    c2.count1table_select = 0;
    if (a1 > a2) { a1 = a2; c2.count1table_select = 1; }
    c2.count1bits = a1;
    if (c2.block_type == NORM_TYPE) {
        if (c2.big_values != g.big_values) q_band_stats(T, ix, c2.big_values, lane, L, Q);   // statistics must honour the new limit
        q_recalc_divide_sub(T, c2, g, lane, L, Q);
    } else {
        c2.part2_3_length = a1;
        a1 = Q.sfb_l[7 + 1];
        if (a1 > i) a1 = i;
        if (a1 > 0) c2.table_select[0] = q_choose_table(T, ix, 0, a1, &c2.part2_3_length, lane, L, Q);
        if (i > a1) c2.table_select[1] = q_choose_table(T, ix, a1, i, &c2.part2_3_length, lane, L, Q);
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
    const int GR = W.mode_gr;                                 // side records are laid out [frame][2][C] whatever GR is
    for (int q = GR * k + gr - 1; q >= 0; q--) {
        const GrSide* r = W.side + (((int64_t)sd.out_slot0 + q / GR) * 2 + q % GR) * C + ch;
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

// perceptual entropy of one channel of one psy call (pecalc_l / pecalc_s, PsyModel.js:845-905) from the maskings E (layout E_*)
LHIP_DEV double q_pecalc(const float* E, int is_short, double masking_lower) {
    const double regcoef_s[12] = {11.8, 13.6, 17.2, 32, 46.5, 51.3, 57.5, 67.1, 71.5, 84.6, 97.6, 130};
    const double regcoef_l[21] = {6.8, 5.8, 5.8, 6.4, 6.5, 9.9, 12.1, 14.4, 15, 18.9, 21.6, 26.9, 34.2, 40.2, 46.8, 56.5, 60.7, 73.9, 85.7, 93.4, 126.1};
    const double LOG10 = 2.30258509299404568402;
    double pe;
    if (is_short) {
        pe = const_here(1236.28 / 4);
        for (int sb = 0; sb < SBMAX_s - 1; sb++)
            for (int sblock = 0; sblock < 3; sblock++) {
                const double thm = E[E_THM_S + sb * 3 + sblock];
                if (thm > 0.0) {
                    const double x = thm * masking_lower, en = E[E_EN_S + sb * 3 + sblock];
                    if (en > x) {
                        if (en > x * 1e10) pe += regcoef_s[sb] * const_here(10.0 * LOG10);
                        else pe += regcoef_s[sb] * v8_log10(en / x);
                    }
                }
            }
    } else {
        pe = const_here(1124.23 / 4);
        for (int sb = 0; sb < SBMAX_l - 1; sb++) {
            const double thm = E[E_THM_L + sb];
            if (thm > 0.0) {
                const double x = thm * masking_lower, en = E[E_EN_L + sb];
                if (en > x) {
                    if (en > x * 1e10) pe += regcoef_l[sb] * const_here(10.0 * LOG10);
                    else pe += regcoef_l[sb] * v8_log10(en / x);
                }
            }
        }
    }
    return pe;
}

// Joint stereo: M/S or L/R for frame k of the stream (Encoder.js:520-561).  M/S when the perceptual entropy of the mid / side pair,
// summed over the frame's granules, is not larger than that of left / right, and both channels have the same block type in the
// first and in the last granule.  The entropies belong to the psy calls of this frame: maskings of the call before (slot - 1),
// block types of the granule, and the masking_lower the previous frame's last channel left behind (gfc.masking_lower; 1 before
// the first frame -- Lame.js:175).  Returns mode_ext: 0 or 2 (wave-uniform).
// The frame's perceptual entropies into L.nsum[granule * 4 + psy channel] (PsyModel.js:1352-1380): maskings of the psy call before
// (slot - 1), block types of the granule, and the masking_lower the previous frame's last channel left behind (gfc.masking_lower;
// 1 before the first frame -- Lame.js:175).
LHIP_DEV void q_frame_pe(const Tables& T, const Workspace& W, const StreamDesc& sd, int k, int lane, QuantLds& L) {
    const int GR = T.mode_gr, C = T.channels_out, Cp = T.psy_channels;
    const int bt_prev = W.blocktype[(int64_t)(sd.gslot0 + GR * k) * C + (C - 1)];    // carry slot for k == 0: -1 on a fresh stream
    const double ml = bt_prev < 0 ? 1.0 : (bt_prev != SHORT_TYPE ? T.masking_lower_long : T.masking_lower_short);
    wave_sync();
    LHIP_LANE_ONCE(it, 0, 4 * GR) {                                                  // lane = granule * 4 + psy channel
        const int gr = it >> 2, chn = it & 3;
        if (chn < Cp) {
            const int gs = sd.gslot0 + 1 + GR * k + gr;
            const int bt0 = W.blocktype[(int64_t)gs * C], bt1 = W.blocktype[(int64_t)gs * C + (C - 1)];
            const int type = chn < 2 ? (chn ? bt1 : bt0) : ((bt0 == SHORT_TYPE || bt1 == SHORT_TYPE) ? SHORT_TYPE : NORM_TYPE);
            L.nsum[it] = q_pecalc(W.E + ((int64_t)(gs - 1) * Cp + chn) * E_STRIDE, type == SHORT_TYPE, ml);
        }
    }
    wave_sync();
}

// Joint stereo: M/S or L/R for frame k of the stream (Encoder.js:520-561).  M/S when the perceptual entropy of the mid / side pair,
// summed over the frame's granules, is not larger than that of left / right, and both channels have the same block type in the
// first and in the last granule.  Returns mode_ext: 0 or 2 (wave-uniform); the entropies stay in L.nsum.
LHIP_DEV int q_ms_decision(const Tables& T, const Workspace& W, const StreamDesc& sd, int k, int lane, QuantLds& L) {
    const int GR = T.mode_gr;
    q_frame_pe(T, W, sd, k, lane, L);
    double sum_ms = 0., sum_lr = 0.;
    for (int gr = 0; gr < GR; gr++)
        for (int ch = 0; ch < 2; ch++) { sum_ms += L.nsum[4 * gr + 2 + ch]; sum_lr += L.nsum[4 * gr + ch]; }
    int ms = 0;
    if (sum_ms <= 1.00 * sum_lr) {
        const int g0 = sd.gslot0 + 1 + GR * k, g1 = g0 + GR - 1;
        if (W.blocktype[(int64_t)g0 * 2] == W.blocktype[(int64_t)g0 * 2 + 1] && W.blocktype[(int64_t)g1 * 2] == W.blocktype[(int64_t)g1 * 2 + 1]) ms = 2;
    }
    return uni(ms);
}

// on_pe (QuantizePVT.js:421-484) with ResvMaxBits (Reservoir.js:190-229), the bit reservoir in use.  The reference computes in JS
// numbers (doubles) except where it stores into Int32Arrays (targ_bits, add_bits): those stores truncate.  Returns max_bits.
LHIP_DEV double q_on_pe_resv(const Tables& T, const double* pe, int* targ, int mean_bits, int cbr, int ResvSize_i, int ResvMax_i) {
    const int C = T.channels_out;
    double ResvSize = ResvSize_i, ResvMax = ResvMax_i, tbits, add_b, extra_bits, max_bits, bits;
    int add_bits[2] = {0, 0};
    if (cbr != 0) ResvSize += mean_bits;
    tbits = mean_bits;
    if (ResvSize * 10 > ResvMax * 9) { add_b = ResvSize - (ResvMax * 9) / 10; tbits += add_b; }
    else { add_b = 0; tbits -= .1 * mean_bits; }
    extra_bits = (ResvSize < (ResvMax * 6) / 10 ? ResvSize : (ResvMax * 6) / 10);
    extra_bits -= add_b;
    if (extra_bits < 0) extra_bits = 0;
    max_bits = tbits + extra_bits;
    if (max_bits > MAX_BITS_PER_GRANULE) max_bits = MAX_BITS_PER_GRANULE;
    bits = 0;
    for (int ch = 0; ch < C; ++ch) {
        const double t = tbits / C;
        targ[ch] = js_toint32(t < MAX_BITS_PER_CHANNEL ? t : (double)MAX_BITS_PER_CHANNEL);
        add_bits[ch] = js_toint32((double)targ[ch] * pe[ch] / 700.0 - targ[ch]);
        if (add_bits[ch] > mean_bits * 3 / 4) add_bits[ch] = mean_bits * 3 / 4;
        if (add_bits[ch] < 0) add_bits[ch] = 0;
        if (add_bits[ch] + targ[ch] > MAX_BITS_PER_CHANNEL) add_bits[ch] = (MAX_BITS_PER_CHANNEL - targ[ch]) > 0 ? MAX_BITS_PER_CHANNEL - targ[ch] : 0;
        bits += add_bits[ch];
    }
    if (bits > extra_bits)
        for (int ch = 0; ch < C; ++ch) add_bits[ch] = js_toint32(extra_bits * add_bits[ch] / bits);
    for (int ch = 0; ch < C; ++ch) { targ[ch] += add_bits[ch]; extra_bits -= add_bits[ch]; }
    bits = 0;
    for (int ch = 0; ch < C; ++ch) bits += targ[ch];
    if (bits > MAX_BITS_PER_GRANULE)
        for (int ch = 0; ch < C; ++ch) {
            targ[ch] = js_toint32((double)targ[ch] * MAX_BITS_PER_GRANULE);
            targ[ch] = js_toint32((double)targ[ch] / bits);
        }
    return max_bits;
}

// reduce_side (QuantizePVT.js:486-534): M/S granules move bits from the side to the mid channel by the energy ratio; the
// reference's targ_bits is an Int32Array, so every store truncates
LHIP_DEV void q_reduce_side(int* targ, double ms_ener_ratio, int mean_bits, double max_bits) {
    double fac = .33 * (.5 - ms_ener_ratio) / .5;
    if (fac < 0) fac = 0;
    if (fac > .5) fac = .5;
    int move_bits = js_toint32(fac * .5 * (targ[0] + targ[1]));
    if (move_bits > MAX_BITS_PER_CHANNEL - targ[0]) move_bits = MAX_BITS_PER_CHANNEL - targ[0];
    if (move_bits < 0) move_bits = 0;
    if (targ[1] >= 125) {
        if (targ[1] - move_bits > 125) {
            if (targ[0] < mean_bits) targ[0] += move_bits;
            targ[1] -= move_bits;
        } else {
            targ[0] += targ[1] - 125;
            targ[1] = 125;
        }
    }
    move_bits = targ[0] + targ[1];
    if (move_bits > max_bits) {
        targ[0] = js_toint32((max_bits * targ[0]) / move_bits);
        targ[1] = js_toint32((max_bits * targ[1]) / move_bits);
    }
}

// returns max_bits of on_pe (tbits + extra_bits, extra_bits = 0 here, capped at the per-granule limit)
LHIP_DEV int targ_bits_for(const Tables& T, int mean_bits, int gr, int ResvSize, int* targ) {
    // on_pe + ResvMaxBits(cbr = gr) with the reservoir disabled (QuantizePVT.js:421-484, Reservoir.js:190-229)
    const int C = uni_here(T.channels_out);
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
    return tbits < MAX_BITS_PER_GRANULE ? tbits : MAX_BITS_PER_GRANULE;
}

// One granule-channel of a frame, from the spectrum to the published record: init_outer_loop .. best_huffman_divide (the body of the
// reference's per-channel loop, Quantize.js:1406-1466 CBR_iteration_loop + iteration_finish_one) on the calling wave's LDS record.
// `used` = the bin-search seed, `gr0_bt` = this channel's block type in granule 0 (scfsi); L.sf_gr0[ch] is written by granule 0 and read
// by granule 1 of the same channel.  Returns the bits spent (part2_3 + part2), the seed the next granule of this channel starts from
// (valid if `active`) and the block type.
struct UnitOut { int bits; Seed next; int block_type; int active; };
// Inlined at every call site: behind a call (one copy of the code for kb_quant's, the owner's and the helper's site) g_quant was a third slower -- spills around the call.
LHIP_DEV UnitOut q_unit(const Tables& T, const PowBase& pb10, const Workspace& W, int C, int Cp, int fidx, int gslot, int gr, int ch, int mode_ext,
                        double ath_adjust, int targ_ch, Seed used, int gr0_bt, int lane, QuantLds& L, const QuantTabs& Q, CountShare* cs = nullptr,
                        CandShare* cd = nullptr, const unsigned char* lds0 = nullptr, int lds_stride = 0, int cs_stride = 0) {
    lane = lane_anew(lane);        // (here and below: lane-derived LDS / HBM addresses are formed where they are used, not parked in scratch across the search)
    UnitOut u; u.next = used;
    GI g;
    const int bt = W.blocktype[(int64_t)gslot * C + ch];
    const double masking_lower = (bt != SHORT_TYPE) ? T.masking_lower_long : T.masking_lower_short;
    const float* ratio = W.E + ((int64_t)(gslot - 1) * Cp + ch + mode_ext) * E_STRIDE;   // thresholds of the previous psy call (mid / side: channels 2, 3)
    { PH_BEGIN(); q_init_outer_loop(T, pb10, ath_adjust, g, bt, xr_source(W, C, gslot, ch, mode_ext == 2), mode_ext == 2 ? nullptr : W.xr + ((int64_t)gslot * C + ch) * 576, 0, lane, L, Q); PH_END(L, PH_INIT); }
    int active = 0, bs_gain = 0;
    if (q_init_xrpow(g, lane, L, Q)) {
        active = 1;
        { PH_BEGIN(); q_calc_xmin(T, ath_adjust, masking_lower, ratio, g, lane, L, Q); PH_END(L, PH_XMIN); }
        int16_t* kept = cs ? cs->kept : W.l3 + (((int64_t)fidx * 2 + gr) * C + ch) * 576;
        { PH_BEGIN(); q_outer_loop(T, g, targ_ch, used.start, used.step, &bs_gain, kept, W.side + ((int64_t)fidx * 2 + gr) * C + ch, W.vdig + ((int64_t)fidx * 2 + gr) * C + ch, W.vdig_n, lane, L, Q, cs, cd, lds0, lds_stride, cs_stride); PH_END(L, PH_XRPOW); }   // (profiling builds: the whole search in the otherwise unused slot)
        uni_gi(g); bs_gain = uni(bs_gain);
        lane = lane_anew(lane);
        wave_sync();                                    // the kept spectrum was written by other lanes of this wave
#if LHIP_NL == 1
        for (int i = lane; i < 288; i += LHIP_NL) ((uint32_t*)L.ixw)[i] = ((const uint32_t*)kept)[i];
#else
        {   // all of a lane's words in flight at once (a lane-strided loop waits for every load by itself)
            uint32_t kw[NPL];
#pragma unroll
            for (int j = 0; j < NPL; j++) { const int i = lane + LHIP_NL * j; kw[j] = ((const uint32_t*)kept)[i < 288 ? i : 287]; }
#pragma unroll
            for (int j = 0; j < NPL; j++) LHIP_PIN_LOADED(kw[j]);
#pragma unroll
            for (int j = 0; j < NPL; j++) { const int i = lane + LHIP_NL * j; if (i < 288) ((uint32_t*)L.ixw)[i] = kw[j]; }
        }
#endif
        wave_sync();
        Seed nx; nx.step = (used.start - bs_gain >= 4) ? 4 : 2; nx.start = bs_gain;
        u.next = nx;
    } else {
        for (int i = lane; i < 576; i += LHIP_NL) L.ixw[i] = 0;
        wave_sync();
    }
    int scfsi[4];
    { PH_BEGIN(); q_best_scalefac_store(T, g, gr, ch, gr0_bt, scfsi, lane, L, Q); PH_END(L, PH_SFSTORE); }
    uni_gi(g);
    if (T.use_best_huffman == 1) { PH_BEGIN(); q_best_huffman_divide(T, g, lane, L, Q); PH_END(L, PH_HUFFDIV); }
    uni_gi(g);
    u.bits = g.part2_3_length + g.part2_length;
    u.block_type = g.block_type;
    if (gr == 0) { LHIP_LANE_ONCE(i, 0, (SFBMAX) + 1) L.sf_gr0[ch][i] = (int8_t)L.sfb[i]; }
    // ---- publish the record and the signed quantized spectrum ----
    lane = lane_anew(lane);
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
        W.vdig[((int64_t)fidx * 2 + gr) * C + ch] = vd_head(active, used.start, used.step, bs_gain);      // digest word VD_HEAD
        out->targ_bits = targ_ch;
        out->mode_ext = mode_ext;
        out->scfsi = scfsi[0] | (scfsi[1] << 1) | (scfsi[2] << 2) | (scfsi[3] << 3);
    }
    LHIP_LANE_ONCE(i, 0, SFBMAX) out->scalefac[i] = L.sfb[i];
    if (!active && lane == 0) { out->bs_ntab = 0; out->bs_state = 0; }
    int16_t* l3o = W.l3 + (((int64_t)fidx * 2 + gr) * C + ch) * 576;
    for (int i = lane; i < 576; i += LHIP_NL) {
        const int v = L.ixw[i];
        l3o[i] = (int16_t)(((double)L.xr[i] < 0) ? -v : v);
    }
    wave_sync();
    u.active = active;
    return u;
}

// One wave per frame slot.  chain == 0: speculative reset seed (exact for the first frame of a stream
// batch, whose seed is the carried one); chain == 1: chain-implied seed (repair pass, flagged frames only);
// chain == 2: the frames of the stream are quantized in order (bit reservoir: the persistent per-stream kernel), the seed is what the
// frame before left in W.seed[fslot - 1] and this frame leaves its own in W.seed[fslot] (no walk back through the side records).
// rvp (RESV only): the stream's reservoir record -- in global memory (one-frame launches) or in the LDS of the per-stream kernel.
// PAIR == 1 (latency path for small stereo batches, g_quant_pair): the workgroup is two waves, wave `my_ch` does that
// channel only -- the channels of a granule are independent given the granule's bit budget -- and the two meet once per
// granule to exchange the bits they used (ResvSize feeds the next granule's budget) through `mbox` in LDS.
// RESV == 1: the bit-reservoir extension (Tables::disable_reservoir == 0) -- a separate instantiation, so that the usual path carries none
// of its state (the entropies, the reservoir record) through the frame
template <int PAIR = 0, int RESV = 0>
LHIP_DEV void kb_quant(const Tables& T, const PowBase& pb10, const Workspace& W, const StreamDesc* SD, int fslot,
                       int chain, int lane, QuantLds& L, const QuantTabs& Q, int my_ch = -1, int* mbox = nullptr, const ResvState* rvp = nullptr,
                       int* hint = nullptr, CountShare* cs = nullptr, const int* later_granules_ready = nullptr, CandShare* cd = nullptr, const unsigned char* lds0 = nullptr, int lds_stride = 0, int cs_stride = 0) {
    const int C = T.channels_out;
    const int st = W.fslot_stream[fslot];
    const StreamDesc sd = SD[st];
    const int k = fslot - sd.fslot0 - 1;
    if (k < 0) return;
    const int fidx = sd.out_slot0 + k;                    // dense frame index
    if (chain == 1 && !W.seed_flag[fidx]) return;
#ifdef LHIP_PHASE_PROF
    const unsigned long long ph_total0_ = __builtin_amdgcn_s_memtime();     // L.prof is zeroed / flushed once per wave by g_quant
#endif
    const double ath_adjust = W.ath_adjust[fslot];        // after adjust_ATH of this frame
    const int padding = frame_padding(T, sd, k);
    const int mean_bits = (frame_bits_of(T, padding) - T.sideinfo_len * 8) / T.mode_gr;
    // per-channel state as named scalars: a dynamically indexed local array would live in scratch memory,
    // and everything read back from scratch counts as divergent for the compiler
    Seed seed0, seed1;
    seed0.start = seed1.start = W.spec_start; seed0.step = seed1.step = W.spec_step;
    if (chain == 2) {
        const int32_t* ps = W.seed + (int64_t)(fslot - 1) * C * 2;
        seed0.start = uni(ps[0]); seed0.step = uni(ps[1]);
        if (C > 1) { seed1.start = uni(ps[2]); seed1.step = uni(ps[3]); }
    } else if (chain || k == 0) {
        seed0 = seed_before(W, sd, C, k, 0, 0);
        if (C > 1) seed1 = seed_before(W, sd, C, k, 0, 1);
    }
#ifndef LHIP_NO_SEED_HINT
    else if (hint && hint[0] == st) {
        // speculation with a better guess than the reset seed: the gains this wave's previous frame OF THE SAME STREAM ended on (any seed
        // is legitimate here -- the validation replays the search with the chain-implied one); a steady stream's chain seeds look like this
        if (hint[1] >= 0) { seed0.start = hint[1]; seed0.step = 2; }
        if (C > 1 && hint[2] >= 0) { seed1.start = hint[2]; seed1.step = 2; }
    }
#endif
    int gr0_bt0 = 0, gr0_bt1 = 0;
    const int Cp = T.psy_channels;
    const int mode_ext = (T.mode == 1) ? q_ms_decision(T, W, sd, k, lane, L) : 0;      // joint stereo: this frame M/S (2) or L/R (0)
    // Bit reservoir (extension): the frame starts from the reservoir the previous frame left (one frame per stream and launch), its
    // budget follows the perceptual entropies (on_pe), and ResvFrameEnd's verdict goes to the bit packer, which commits it.
    constexpr bool resv = RESV != 0;
    const ResvState* rv = resv ? rvp : nullptr;
    int ResvSize = resv ? uni(rv->ResvSize) : 0;
    int ResvMax = 0;
    double pe_use[2][2] = {{0., 0.}, {0., 0.}};
    float pefir_new = 0.f;
    if constexpr (resv) {
        // ResvFrameBegin (Reservoir.js:130-180; brate <= 320, not strict_ISO)
        const int frameLength = frame_bits_of(T, padding), resvLimit = (8 * 256) * T.mode_gr - 8, maxmp3buf = 8 * 1440;
        ResvMax = maxmp3buf - frameLength;
        if (ResvMax > resvLimit) ResvMax = resvLimit;
        if (ResvMax < 0) ResvMax = 0;
        // the entropies of the maskings in use, scaled by the 19-frame FIR of their sums (Encoder.js:600-626; pefirbuf is a Float32Array)
        if (T.mode != 1) q_frame_pe(T, W, sd, k, lane, L);
        double f = 0.0;
        for (int gr = 0; gr < T.mode_gr; gr++)
            for (int ch = 0; ch < C; ch++) { pe_use[gr][ch] = L.nsum[4 * gr + ch + mode_ext]; f += pe_use[gr][ch]; }
        pefir_new = (float)f;
        {
            const double fircoef[9] = {-0.0207887 * 5, -0.0378413 * 5, -0.0432472 * 5, -0.031183 * 5, 7.79609e-18 * 5, 0.0467745 * 5,
                                       0.10091 * 5, 0.151365 * 5, 0.187098 * 5};
            // buffer after the shift: b[i] = old[i + 1] for i < 18, b[18] = the new sum
            f = rv->pefirbuf[10];
            for (int i = 0; i < 9; i++) f += ((double)rv->pefirbuf[i + 1] + (double)(i == 0 ? pefir_new : rv->pefirbuf[19 - i])) * fircoef[i];
        }
        f = (670 * 5 * T.mode_gr * C) / f;
        for (int gr = 0; gr < T.mode_gr; gr++)
            for (int ch = 0; ch < C; ch++) pe_use[gr][ch] *= f;
        wave_sync();
    }
    q_ath_pseudo(T, pb10, ath_adjust, lane, L, Q);          // the frame's analog-silence thresholds, both block kinds
    for (int gr = 0; gr < T.mode_gr; gr++) {
        const int gslot = sd.gslot0 + 1 + T.mode_gr * k + gr;
        // one-frame launch: the thresholds granule 1 quantizes against (psyB of granule 0) are computed by other waves WHILE granule 0 is
        // quantized; a one-channel frame has no workgroup barrier between its granules to order that, so it waits for their flag here
        if (gr > 0 && later_granules_ready) { (void)wg_wait_not(later_granules_ready, 0, 0, lane); wg_acquire(); }
        int targ[2] = {0, 0};
        const double max_bits = resv ? q_on_pe_resv(T, pe_use[gr], targ, mean_bits, gr, ResvSize, ResvMax) : (double)targ_bits_for(T, mean_bits, gr, ResvSize, targ);
        if (mode_ext == 2) {
            // Encoder.js:482-486: side / (mid + side) of the total energies the psy call of this granule handed back (one call of delay)
            const float* te = W.tot_ener + (int64_t)(gslot - 1) * 4;
            double r = (double)te[2] + (double)te[3];
            if (r > 0) r = (double)te[3] / r;
            q_reduce_side(targ, r, mean_bits, max_bits);
        }
        const int targ0 = uni(targ[0]), targ1 = uni(targ[1]);
        for (int ch = 0; ch < C; ch++) {
            if (PAIR && ch != my_ch) continue;
            const UnitOut u = q_unit(T, pb10, W, C, Cp, fidx, gslot, gr, ch, mode_ext, ath_adjust, ch == 0 ? targ0 : targ1, ch == 0 ? seed0 : seed1,
                                     ch == 0 ? gr0_bt0 : gr0_bt1, lane, L, Q, cs, cd, lds0, lds_stride, cs_stride);
            if (u.active) { if (ch == 0) seed0 = u.next; else seed1 = u.next; }
            if (!PAIR) ResvSize = uni(ResvSize - u.bits);
            else if (lane == 0) mbox[2 * gr + ch] = u.bits;
            if (gr == 0) { if (ch == 0) gr0_bt0 = u.block_type; else gr0_bt1 = u.block_type; }
        }
        if (PAIR) {                                   // both waves have published this granule: take the other channel's bits
            if (cs) wg_store(&cs->state, CS_BARRIER, lane);     // (this wave's count helper keeps the barrier's count)
#if LHIP_NL != 1
            if (cd) q_cand_signal(cd[wg_wave_id()], CS_BARRIER, lane);       // (so do its candidate helpers)
#endif
            wg_barrier();
            ResvSize = uni(ResvSize - (mbox[2 * gr] + mbox[2 * gr + 1]));
        }
    }
    if constexpr (resv) if (lane == 0 && (!PAIR || my_ch == 0)) {
        // ResvFrameEnd (Reservoir.js:243-293); main_data_begin is a double there (fractions of a byte survive in it)
        const double mdb0 = rv->main_data_begin;
        int rs = ResvSize + mean_bits * T.mode_gr, over_bits, stuffingBits = 0;
        if ((over_bits = rs % 8) != 0) stuffingBits += over_bits;
        over_bits = (rs - stuffingBits) - ResvMax;
        if (over_bits > 0) stuffingBits += over_bits;
        const double mdb_bytes = (mdb0 * 8 < stuffingBits ? mdb0 * 8 : (double)stuffingBits) / 8;
        FrameResv fr;
        fr.drain_pre = js_toint32(8 * mdb_bytes);
        stuffingBits -= fr.drain_pre;
        rs -= fr.drain_pre;
        fr.main_data_begin = mdb0 - mdb_bytes;
        fr.drain_post = stuffingBits;
        rs -= stuffingBits;
        fr.ResvSize = rs; fr.ResvMax = ResvMax; fr.pefir_new = pefir_new; fr.pad_ = 0;
        W.fr[fidx] = fr;
    }
    if (hint) { hint[0] = st; hint[1] = seed0.start; hint[2] = C > 1 ? seed1.start : -1; }
    if (chain == 1 && lane == 0) W.seed_flag[fidx] = 0;
    if (chain == 2 && lane == 0) {                            // OldValue / CurrentStep after this frame (every wave of a pair its own channel)
        int32_t* ps = W.seed + (int64_t)fslot * C * 2;
        if (!PAIR || my_ch == 0) { ps[0] = seed0.start; ps[1] = seed0.step; }
        if (C > 1 && (!PAIR || my_ch == 1)) { ps[2] = seed1.start; ps[3] = seed1.step; }
    }
#ifdef LHIP_PHASE_PROF
    if (lane == 0) { L.prof[PH_TOTAL] += (unsigned int)(__builtin_amdgcn_s_memtime() - ph_total0_); L.prof[32 + PH_TOTAL] += 1; }
#endif
}

// Replay the bin searches of a frame with the chain-implied seeds; flag the frame if any result differs.
// The search only asks for count_bits(gain) with all-zero scalefactors; the speculative pass left a memo of every
// such evaluation (GrSide::bs_tab), so most replays are pure scalar look-ups and the spectrum is only re-quantized
// on a miss.
LHIP_DEV void kb_validate(const Tables& T, const PowBase& pb10, const Workspace& W, const StreamDesc* SD, int fslot,
                          int lane, QuantLds& L, const QuantTabs& Q) {
    const int C = T.channels_out;
    const int st = W.fslot_stream[fslot];
    const StreamDesc sd = SD[st];
    const int k = fslot - sd.fslot0 - 1;
    if (k < 0) return;
    const int fidx = sd.out_slot0 + k;
    if (uni(W.seed_flag[fidx]) != 2) return;                  // only frames the memo-only pass could not decide
    const double ath_adjust = W.ath_adjust[fslot];
    int bad = 0;
    for (int gr = 0; gr < T.mode_gr && !bad; gr++) {
        const int gslot = sd.gslot0 + 1 + T.mode_gr * k + gr;
        for (int ch = 0; ch < C && !bad; ch++) {
            const GrSide* rec = W.side + ((int64_t)fidx * 2 + gr) * C + ch;
            if (!rec->active) continue;
            const Seed s = seed_before(W, sd, C, k, gr, ch);
            if (s.start == rec->bs_start && s.step == rec->bs_step_in) continue;     // already quantized with this seed
            const int ntab = uni(rec->bs_ntab);
            // memo entry per lane (device) / linear search (host simulation)
            int my_ent = 0, my_asg = 0;
            LHIP_LANE_ONCE(i, 0, ntab) { my_ent = rec->bs_tab[i]; my_asg = rec->bs_asg[i]; }
            GI g;
            PrevNoise pn_none; pn_none.gain = 0; pn_none.sfb_count1 = 0;
            int inited = 0;
            // bin_search_StepSize (Quantize.js:322-381), part2_length == 0 at this point
            const int desired_rate = uni(rec->targ_bits);
            int gain = uni(s.start), CurrentStep = uni(s.step), flagGoneOver = 0, Direction = 0, up = 0;
            GI g0;                                          // conditionally assigned fields as init_outer_loop leaves them
            g0.table_select[0] = g0.table_select[1] = g0.table_select[2] = 0; g0.region0_count = 0; g0.region1_count = 0;
            int cstate = pack_cond_fields(g0, 0);
            for (;;) {
                int nBits = -1, asg = 0;
#if LHIP_NL == 1
                for (int i = 0; i < ntab; i++) if ((int)((uint32_t)rec->bs_tab[i] >> 24) == gain) { nBits = rec->bs_tab[i] & 0xffffff; asg = rec->bs_asg[i]; break; }
                (void)my_ent; (void)my_asg;
#else
                {
                    const uint64_t hit = wave_ballot(lane < ntab && (int)((uint32_t)my_ent >> 24) == gain);
                    if (hit) {
                        const int src = (int)__builtin_ctzll(hit);
                        nBits = wave_bcast(my_ent, src) & 0xffffff;
                        asg = wave_bcast(my_asg, src);
                    }
                }
#endif
                if (nBits < 0) {
                    if (!inited) {
                        const int ms = uni(rec->mode_ext) == 2;        // M/S granules were not written back: the silence rule runs again
                        if (ms) q_ath_pseudo(T, pb10, ath_adjust, lane, L, Q);
                        q_init_outer_loop(T, pb10, ath_adjust, g, W.blocktype[(int64_t)gslot * C + ch],
                                          xr_source(W, C, gslot, ch, ms), nullptr, ms ? 0 : 1, lane, L, Q);
                        q_init_xrpow(g, lane, L, Q);
                        // max_nonzero_coeff is set by calc_xmin in the reference before the bin search
                        if (g.block_type != SHORT_TYPE) {
                            int t = -1;
                            for (int i = lane; i < 576; i += LHIP_NL) if (!((double)L.xr[i] == 0)) t = i;
                            t = wave_max(t);
                            g.max_nonzero_coeff = (t >= 575) ? 575 : t + 1;
                        }
                        q_set_firstcut(g, lane, L);
                        inited = 1;
                    }
                    g.global_gain = gain;
                    nBits = q_count_bits(T, g, L.sfb, L.ixw, 0, pn_none, &asg, lane, L, Q);
                }
                nBits = uni(nBits);
                cstate = apply_cond_fields(cstate, uni(asg));
                if (!up) {
                    if (CurrentStep == 1 || nBits == desired_rate) up = 1;
                    else {
                        int step;
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
                        gain += step;
                        if (gain < 0) { gain = 0; flagGoneOver = 1; }
                        if (gain > 255) { gain = 255; flagGoneOver = 1; }
                        continue;
                    }
                }
                if (nBits > desired_rate && gain < 255) { gain++; continue; }
                break;
            }
            // the gain AND the path-dependent leftovers (table_select of empty regions, region counts) must be what the
            // speculative pass produced
            if (gain != rec->bs_gain || cstate != (rec->bs_state & ~15)) bad = 1;
        }
    }
    if (lane == 0) {
        W.seed_flag[fidx] = bad ? 1 : 0;
        if (bad) {
#ifdef LHIP_HOSTSIM
            W.nflagged[0] += 1;
#else
            atomicAdd(W.nflagged, 1);
#endif
        }
    }
}

// The chain-implied seed from the digests (what seed_before derives from the side records): gains of the last two active granules of the channel.
LHIP_DEV Seed seed_before_dig(const Workspace& W, const StreamDesc& sd, int C, int k, int gr, int ch) {
    const int32_t* carry = W.seed + ((int64_t)sd.fslot0 * C + ch) * 2;
    int g1 = -1, g2 = -1;
    const int GR = W.mode_gr;
    for (int q = GR * k + gr - 1; q >= 0; q--) {
        const uint32_t h = W.vdig[(((int64_t)sd.out_slot0 + q / GR) * 2 + q % GR) * C + ch];
        if (vd_active(h)) {
            if (g1 < 0) g1 = vd_gain(h);
            else { g2 = vd_gain(h); break; }
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

// Memo-only replay of ONE granule-channel's bin search with the chain-implied seed: a few scalar look-ups in the granule-channels' DIGESTS
// (W.vdig: seed used, resulting gain, target, the first VD_ENT memo entries; word-major, so the threads of a wave read neighbouring
// words) -- the side records, 464 bytes apart with the memo at their far end, are only read by a search with more than VD_ENT memoised
// evaluations.  Returns 0 consistent, 1 re-quantize the frame with the chain-implied seed, 2 undecided (a gain the speculative pass never
// evaluated) -> kb_validate decides.
LHIP_DEV int validate_fast_gc(const Tables& T, const Workspace& W, const StreamDesc& sd, int k, int fidx, int gr, int ch) {
    const int C = T.channels_out;
    const int64_t dn = W.vdig_n;
    const int64_t gc = ((int64_t)fidx * 2 + gr) * C + ch;
    const uint32_t* dig = W.vdig + gc;
    const uint32_t h = dig[0];
    if (!vd_active(h)) return 0;
    const Seed s = seed_before_dig(W, sd, C, k, gr, ch);
    if (s.start == vd_start(h) && s.step == vd_step(h)) return 0;
    const uint32_t tw = dig[VD_TARG * dn];
    const int ntab = (int)(tw >> 24), desired_rate = (int)(tw & 0xffffffu);
    const GrSide* rec = W.side + gc;
    int32_t et[VD_ENT], ea[VD_ENT];
#pragma unroll
    for (int i = 0; i < VD_ENT; i++) { et[i] = -1; ea[i] = 0; if (i < ntab) { et[i] = (int32_t)dig[(VD_TAB + 2 * i) * dn]; ea[i] = (int32_t)dig[(VD_TAB + 2 * i + 1) * dn]; } }
    int gain = s.start, CurrentStep = s.step, flagGoneOver = 0, Direction = 0, up = 0;
    GI g0;
    g0.table_select[0] = g0.table_select[1] = g0.table_select[2] = 0; g0.region0_count = 0; g0.region1_count = 0;
    int cstate = pack_cond_fields(g0, 0);
    for (;;) {
        int nBits = -1, asg = 0;
#pragma unroll
        for (int i = VD_ENT - 1; i >= 0; i--) if (i < ntab && (int)((uint32_t)et[i] >> 24) == gain) { nBits = et[i] & 0xffffff; asg = ea[i]; }   // the FIRST entry with that gain, as the record walk finds it
        if (nBits < 0 && ntab > VD_ENT)      // a long search: the entries beyond the digest are in the side record
            for (int i = VD_ENT; i < ntab; i++) { const int e = rec->bs_tab[i]; if ((int)((uint32_t)e >> 24) == gain) { nBits = e & 0xffffff; asg = rec->bs_asg[i]; break; } }
        if (nBits < 0) return 2;
        cstate = apply_cond_fields(cstate, asg);
        if (!up) {
            if (CurrentStep == 1 || nBits == desired_rate) up = 1;
            else {
                int step;
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
                gain += step;
                if (gain < 0) { gain = 0; flagGoneOver = 1; }
                if (gain > 255) { gain = 255; flagGoneOver = 1; }
                continue;
            }
        }
        if (nBits > desired_rate && gain < 255) { gain++; continue; }
        break;
    }
    return (gain != vd_gain(h) || cstate != (int)(dig[VD_STATE * dn] & ~15u)) ? 1 : 0;
}

// The frame's verdict from its granule-channels' (a definite "re-quantize" wins over "undecided": the re-quantization settles every
// granule-channel of the frame) -> W.seed_flag, the counters and the list of undecided frames.
LHIP_DEV void validate_fast_publish(const Workspace& W, int fslot, int fidx, int any1, int any2) {
    const int verdict = any1 ? 1 : any2 ? 2 : 0;
    W.seed_flag[fidx] = verdict;
    if (verdict) {
#ifdef LHIP_HOSTSIM
        const int pos = W.nflagged[verdict - 1]; W.nflagged[verdict - 1] += 1;
#else
        const int pos = atomicAdd(W.nflagged + (verdict - 1), 1);
#endif
        if (verdict == 2) W.slow_list[pos] = fslot;
    }
}

// One thread per frame (the persistent repair kernel's later passes, the host simulation).
// only_pass > 0: just the frames stamped for that pass (successors of frames the previous repair pass re-quantized)
LHIP_DEV void kb_validate_fast(const Tables& T, const Workspace& W, const StreamDesc* SD, int fslot, int only_pass = 0) {
    const int C = T.channels_out;
    const StreamDesc sd = SD[W.fslot_stream[fslot]];
    const int k = fslot - sd.fslot0 - 1;
    if (k < 0) return;
    const int fidx = sd.out_slot0 + k;
    if (only_pass > 0 && W.reval[fidx] != only_pass) return;
    int any1 = 0, any2 = 0;
    for (int gr = 0; gr < T.mode_gr && !any1; gr++)
        for (int ch = 0; ch < C && !any1; ch++) { const int v = validate_fast_gc(T, W, sd, k, fidx, gr, ch); any1 |= (v == 1); any2 |= (v == 2); }
    validate_fast_publish(W, fslot, fidx, any1, any2);
}

#ifndef LHIP_HOSTSIM
// The first pass over a whole batch (g_validate_fast): FOUR lanes per frame, one per granule-channel -- a frame's check is a chain of
// dependent look-ups (the two previous granules' digests, then the memo), and one thread per frame left the chip with a wave and a half
// per SIMD waiting on them (0.27 ms per 1e5 two-channel frames); the quad's verdicts meet through DPP and its first lane publishes.
struct ValidateShare { int total[2], base[2], woff[4][2]; };      // g_validate_fast: 256 threads = 4 waves
LHIP_DEV void kb_validate_fast_quad(const Tables& T, const Workspace& W, const StreamDesc* SD, int fslot, int sub, bool live, ValidateShare& S) {
    const int C = T.channels_out;
    int v = 0, fidx = 0;
    bool has = false;
    if (live) {
        const StreamDesc sd = SD[W.fslot_stream[fslot]];
        const int k = fslot - sd.fslot0 - 1;
        if (k >= 0) {
            has = true;
            fidx = sd.out_slot0 + k;
            const int gr = sub / C, ch = sub - gr * C;
            if (gr < T.mode_gr) v = validate_fast_gc(T, W, sd, k, fidx, gr, ch);
        }
    }
    // OR over the four lanes of the quad (bit 0: some granule-channel says "re-quantize", bit 1: some is undecided)
    int m = (v == 1 ? 1 : 0) | (v == 2 ? 2 : 0);
    m |= __builtin_amdgcn_mov_dpp(m, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
    m |= __builtin_amdgcn_mov_dpp(m, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
    // The quad's first lane publishes.  The counters are bumped ONCE PER WORKGROUP: on steady material a quarter to a third of the frames are
    // "undecided" (profiles/r04_pass6_validate_stats.txt), and 23-32 k returning atomics on one address serialise in the L2 -- that, not the
    // replay, was this kernel's time (0.26 ms stereo / 0.37 ms mono per 1e5 frames, in proportion to the undecided frames).  Every wave ranks
    // its publishing lanes with ballots, the waves meet in LDS, one thread per counter talks to global memory, and the list slots follow from
    // the ranks.  (The order of W.slow_list is immaterial: g_fixup spreads its entries over its waves.)
    const bool pub = has && sub == 0;
    const int verdict = pub ? ((m & 1) ? 1 : (m & 2) ? 2 : 0) : 0;
    if (pub) W.seed_flag[fidx] = verdict;
    const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
    const uint64_t b1 = __ballot(verdict == 1), b2 = __ballot(verdict == 2);
    if (threadIdx.x < 2) S.total[threadIdx.x] = 0;
    __syncthreads();
    if (lane == 0) {                                 // this wave's offsets inside the workgroup's share (LDS atomics: cheap, returning)
        S.woff[wv][0] = b1 ? atomicAdd(&S.total[0], (int)__builtin_popcountll(b1)) : 0;
        S.woff[wv][1] = b2 ? atomicAdd(&S.total[1], (int)__builtin_popcountll(b2)) : 0;
    }
    __syncthreads();
    if (threadIdx.x < 2) { const int n = S.total[threadIdx.x]; S.base[threadIdx.x] = n ? atomicAdd(W.nflagged + threadIdx.x, n) : 0; }
    __syncthreads();
    if (verdict == 2) W.slow_list[S.base[1] + S.woff[wv][1] + (int)__builtin_popcountll(b2 & ((1ull << lane) - 1))] = fslot;
}
#endif

}  // namespace lhip
