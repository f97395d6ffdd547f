/*
 * TEST INFRASTRUCTURE -- CPU oracle for the lamejs per-frame encode path.
 * Plain-C sequential restatement of the reference's algorithm (one stateful stream,
 * frame after frame, exactly the order the reference executes).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product
 * (lamejs_amd/csrc) never links or calls anything under oracle/.
 *
 * Number model: every expression is IEEE f64; values held in the
 * reference's Float32Array objects are `float` here (rounded on store, widened on
 * load); Int32Array stores truncate (js_toint32).  Build with -ffp-contract=off.
 */
#ifndef LO_COMMON_H
#define LO_COMMON_H
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <math.h>
#include "v8math.h"

#define SBMAX_l 22
#define SBMAX_s 13
#define SBPSY_l 21
#define SBPSY_s 12
#define PSFB21 6
#define PSFB12 6
#define CBANDS 64
#define BLKSIZE 1024
#define HBLKSIZE 513
#define BLKSIZE_s 256
#define HBLKSIZE_s 129
#define SFBMAX 39
#define NORM_TYPE 0
#define START_TYPE 1
#define SHORT_TYPE 2
#define STOP_TYPE 3
#define IXMAX_VAL 8206
#define PRECALC_SIZE (IXMAX_VAL + 2)
#define Q_MAX 257
#define Q_MAX2 116
#define LARGE_BITS 100000
#define MAX_BITS_PER_CHANNEL 4095
#define MAX_BITS_PER_GRANULE 7680
#define MFSIZE (3 * 1152 + 576 - 48)
#define SQRT2 1.41421356237309504880
#define D(x) ((double)(x))

/* ---- configuration + tables, filled from the LHTB blob built by lamejs_amd/js/tables.js ---- */
typedef struct {
    /* ints */
    int channels_out, mode, mode_gr, version, samplerate_index, bitrate_index, brate, out_samplerate,
        sideinfo_len, frac_SpF, noise_shaping, noise_shaping_amp, noise_shaping_stop, subblock_gain,
        use_best_huffman, full_outer_loop, substep_shaping, sfb21_extra, quant_comp, quant_comp_short,
        short_blocks_coupled, useTemporal, ATH_useAdjust, athaa_loudapprox, copyright, original, emphasis,
        extension, error_protection, npart_l, npart_s,
        disable_reservoir;                      /* 1 on the Mp3Encoder path (index.js:108); 0: the reservoir extension */
    /* doubles */
    int in_samplerate, rs_filter_l, rs_bpc;      /* resampler (Lame.js:1719-1763); rs_filter_l == 0: no resampling */
    double resample_ratio;
    const float* rs_blackfilt;                  /* [2*bpc+1][filter_l+1] */
    double scale, attackthre, attackthre_s, interChRatio, masking_lower_long, masking_lower_short,
        ATH_aaSensitivityP, ATH_floor, decay, ma_max_i1, ma_max_i2, ma_max_m, VO_SCALE,
        msfix, ATHlower;                          /* joint stereo only (PsyModel.js:1336-1341) */
    /* tables */
    const float *amp_filter, *ATH_l, *ATH_s, *ATH_psfb21, *ATH_psfb12, *ATH_cb_l, *ATH_cb_s, *eql_w,
        *pow43, *adj43, *ipow20, *pow20, *longfact, *shortfact, *rnumlines_l, *bo_l_weight, *bo_s_weight,
        *s3_ll, *s3_ss, *window, *window_s, *mld_l, *mld_s;
    const int32_t *sfb_l, *sfb_s, *psfb21, *psfb12, *bv_scf, *numlines_l, *numlines_s, *bo_l, *bm_l, *bo_s,
        *bm_s, *s3ind, *s3ind_s, *fft_rv_tbl, *mdct_order, *pretab, *scfsi_band, *slen1_n, *slen2_n,
        *slen1_tab, *slen2_tab, *scale_short, *scale_long, *huf_tbl_noESC, *version_bytes, *ht_xlen,
        *ht_linmax, *ht_off, *ht_code, *ht_hlen, *largetbl, *table23, *table56, *t32l, *t33l;
    int n_version_bytes;
    const double *fht_twiddle, *fht_costab, *enwindow, *mdct_win, *ma_tab, *ma_table1, *ma_table2, *ma_table3,
        *hpf_fircoef;
    void* blob_copy;
} lo_cfg;

typedef struct {
    float xr[576];
    int32_t l3_enc[576];
    int32_t scalefac[SFBMAX];
    double xrpow_max;
    int part2_3_length, big_values, count1, global_gain, scalefac_compress, block_type, mixed_block_flag;
    int table_select[3], subblock_gain[4];
    int region0_count, region1_count, preflag, scalefac_scale, count1table_select;
    int part2_length, sfb_lmax, sfb_smin, psy_lmax, sfbmax, psymax, sfbdivide;
    int width[SFBMAX], window[SFBMAX];
    int count1bits, max_nonzero_coeff;
    int slen[4], sfb_part_tab, sfb_part_row;   /* MPEG-2 LSF scalefactor partitions (Takehiro.js:1046-1132) */
} lo_gr;

typedef struct { float l[SBMAX_l]; float s[SBMAX_s][3]; } lo_xmin;
typedef struct { lo_xmin en, thm; } lo_ratio;

typedef struct {
    int global_gain, sfb_count1;
    int32_t step[39];
    float noise[39], noise_log[39];
} lo_noise_data;

typedef struct { double over_noise, tot_noise, max_noise; int over_count, over_SSD, bits; } lo_noise_res;

typedef struct lo_enc {
    lo_cfg c;
    /* stream buffering (Lame.js) */
    float mfbuf[2][MFSIZE];
    int mf_size, mf_samples_to_encode;
    double itime[2];                            /* resampler clock (Lame.js:1755-1756, 1813) */
    float inbuf_old[2][40];                     /* last BLACKSIZE input samples */
    long frame_num;
    /* filterbank state */
    float sb_sample[2][2][18][32];
    int frame_init_done;
    /* psy state */
    lo_xmin en[4], thm[4];
    float nb_s1[4][CBANDS], nb_s2[4][CBANDS];
    float last_en_subshort[4][9];
    int lastAttacks[4];
    int blocktype_old[2];
    float loudness_sq[2][2], loudness_sq_save[2];
    float tot_ener[4];                          /* LameInternalFlags.js:271; chn 2, 3 = mid, side */
    int mode_ext;                               /* Encoder.js:520-561: 0 = L/R, 2 = M/S, per frame */
    double ATH_adjust, ATH_adjustLimit;
    /* quantizer state */
    int OldValue[2], CurrentStep[2];
    double masking_lower;
    lo_gr tt[2][2];
    int scfsi[2][4];
    int ResvSize, resvDrain_post;
    /* bit reservoir (disable_reservoir == 0): Reservoir.js, BitStream.js:100-215, 300-330, 710-780, 836-900 */
    int ResvMax, resvDrain_pre, ancillary_flag;
    double main_data_begin;                     /* a double: the reference's arithmetic leaves fractions in it (Reservoir.js:281-286) */
    float pefirbuf[19];                         /* NsPsy.js:30, Encoder.js:600-626 */
    float nb_1[4][CBANDS], nb_2[4][CBANDS];     /* long-block pre-echo control (PsyModel.js:1300-1318): live once pcfact != 0 */
    struct lo_stream* bs;                       /* the continuous bitstream writer (headers are inserted where their frame starts) */
    int slot_lag, padding;
    /* optional per-frame taps for differential debugging (tests only) */
    struct lo_tap* tap;
} lo_enc;

/* tap: intermediate values of the most recent frame, filled when e->tap != NULL */
typedef struct lo_tap {
    float xr[2][2][576];           /* [gr][ch] MDCT output (before short-block reorder) */
    int block_type[2][2];
    lo_ratio ratio[2][2];          /* masking handed to the quantizer */
    double ath_adjust;             /* after adjust_ATH of this frame */
    float l3_xmin[2][2][SFBMAX];
    int global_gain[2][2], part2_3_length[2][2], part2_length[2][2];
    int mode_ext;                  /* joint stereo: the frame's M/S decision */
    double pe[2][2], pe_MS[2][2], ms_ener_ratio[2];
} lo_tap;

#endif
