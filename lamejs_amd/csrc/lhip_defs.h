// lamejs_amd -- MI355X-native MP3 frame-encode path: shared definitions.
//
// Number model (the bit-exactness contract): every arithmetic expression is
// IEEE f64; values the reference keeps in Float32Array are `float` here and are rounded
// exactly at the reference's store points; Int32Array stores truncate (js_toint32).
// All device code is compiled with -ffp-contract=off (no FMA may ever be formed).
//
// The kernel bodies in this directory are written as *wave programs*: NL lanes execute the
// same uniform control flow, `for (i = lane; i < n; i += NL)` loops have independent
// iterations, and cross-lane traffic goes through LDS or the wave_* helpers.  NL is 64 on
// gfx950.  Two test-only builds (tests/hostsim) compile the same bodies for a CPU: -DLHIP_HOSTSIM with NL = 1 (the
// scalar variants of the few places that differ), and -DLHIP_HOSTSIM -DLHIP_WAVESIM with NL = 64, where the 64 lanes of
// a wave run as fibers and every wave_* primitive is a rendezvous -- the wave programs exactly as the GPU executes
// them, minus the hardware.  The product library contains the HIP build only.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef LHIP_HOSTSIM
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#define LHIP_DEV static inline
#ifdef LHIP_WAVESIM
#define LHIP_NL 64      /* test-only: the 64-lane wave programs themselves, lanes as fibers (lhip_wave.h) */
#else
#define LHIP_NL 1
#endif
#else
#include <hip/hip_runtime.h>
#define LHIP_DEV static __device__ __forceinline__
#define LHIP_NL 64
#endif

// Immutable f64 tables read at wave-uniform, compile-time-constant offsets (filterbank windows): through a pointer to the
// constant address space the compiler may use the scalar unit's loads (s_load, operands in SGPRs) instead of one vector
// memory load per lane -- the tables are written once, before any kernel that reads them is launched.
#ifdef LHIP_HOSTSIM
typedef const double* lhip_ctab;
#define LHIP_CTAB(p) (p)
#define LHIP_SCHED_FENCE() ((void)0)
#define LHIP_PIN_LOADED(x) ((void)0)
#else
// the value of a load is needed HERE: keeps the compiler from sinking a batch of independent loads into the (conditional) code
// that uses them one by one, where each would be waited for on its own
#define LHIP_PIN_LOADED(x) asm volatile("" : "+v"(x))
#define LHIP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)     /* the instruction scheduler moves nothing across this point */
typedef const double __attribute__((address_space(4)))* lhip_ctab;
#define LHIP_CTAB(p) ((lhip_ctab)(uintptr_t)(p))
#endif

// `for (v = lo + lane; v < n; v += NL)` where the author knows n - lo <= NL (band loops: at most 39 bands) but the
// compiler does not: at most one iteration per lane.  Left as a loop, the compiler emits a generic loop nest (with a
// 16-fold unrolled body and the spills that come with it) for every one of them.  `lane` must be in scope.
#if LHIP_NL == 1
#define LHIP_LANE_ONCE(v, lo, n) for (int v = (lo); v < (n); v++)
#define LHIP_LANE_ONCE3(v, lo, n) for (int v = (lo); v < (n); v += 3)
#else
#define LHIP_LANE_ONCE(v, lo, n) if (const int v = (lo) + lane; v < (n))
#define LHIP_LANE_ONCE3(v, lo, n) if (const int v = (lo) + 3 * lane; v < (n))
#endif

namespace lhip {

enum {
    SBMAX_l = 22, SBMAX_s = 13, SBPSY_l = 21, SBPSY_s = 12, PSFB21 = 6, PSFB12 = 6,
    CBANDS = 64, BLKSIZE = 1024, HBLKSIZE = 513, BLKSIZE_s = 256, HBLKSIZE_s = 129, SFBMAX = 39,
    NORM_TYPE = 0, START_TYPE = 1, SHORT_TYPE = 2, STOP_TYPE = 3,
    IXMAX_VAL = 8206, PRECALC_SIZE = IXMAX_VAL + 2, Q_MAX = 257, Q_MAX2 = 116, LARGE_BITS = 100000,
    MAX_BITS_PER_CHANNEL = 4095, MAX_BITS_PER_GRANULE = 7680,
    FRAME = 1152, GRAN = 576, MF_NEEDED = 1904, MF_INIT = 528
};

#define LHIP_SQRT2 1.41421356237309504880

// Read-only configuration + lookup tables, resident in HBM; built once per stream
// configuration from the LHTB blob of lamejs_amd/js/tables.js.  Plain pointers only.
struct Tables {
    // scalars (ints)
    int channels_out, mode, mode_gr, version, samplerate_index, bitrate_index, brate, out_samplerate,
        sideinfo_len, frac_SpF, noise_shaping, noise_shaping_amp, noise_shaping_stop, subblock_gain,
        use_best_huffman, full_outer_loop, substep_shaping, sfb21_extra, quant_comp, quant_comp_short,
        short_blocks_coupled, useTemporal, ATH_useAdjust, athaa_loudapprox, copyright, original, emphasis,
        extension, error_protection, npart_l, npart_s, n_version_bytes,
        n_s3_ll,                            // entries of s3_ll (the long-block spreading rows back to back): g_psyB stages them in LDS
       
        in_samplerate, rs_filter_l, rs_bpc,
        rs_ratio,                           // integer decimation factor (1 = no resampling), derived at create time
        psy_channels,                       // channels the psychoacoustic model analyses: channels_out, or 4 (L, R, mid, side) in joint stereo (mode == 1)
        disable_reservoir;                  // 1 on the Mp3Encoder path (index.js:108); 0 = the bit-reservoir extension: one frame per stream and launch
    // scalars (doubles)
    double scale, attackthre, attackthre_s, interChRatio, masking_lower_long, masking_lower_short,
        ATH_aaSensitivityP, ATH_floor, decay, ma_max_i1, ma_max_i2, ma_max_m, VO_SCALE, resample_ratio,
        msfix, ATHlower;                    // joint stereo only (PsyModel.js:1336-1341)
    // arrays
    const float *rs_blackfilt;              // [2*bpc+1][filter_l+1]; row bpc (= 1) is the only one an integer ratio uses
    const float *amp_filter, *ATH_l, *ATH_s, *ATH_psfb21, *ATH_psfb12, *ATH_cb_l, *ATH_cb_s, *eql_w,
        *pow43, *adj43, *ipow20, *pow20, *longfact, *shortfact, *rnumlines_l, *bo_l_weight, *bo_s_weight,
        *s3_ll, *s3_ss, *window, *window_s, *mld_l, *mld_s;
    const int32_t *sfb_l, *sfb_s, *psfb21, *psfb12, *bv_scf, *numlines_l, *numlines_s, *bo_l, *bm_l, *bo_s,
        *bm_s, *s3ind, *s3ind_s, *fft_rv_tbl, *mdct_order, *pretab, *scfsi_band, *slen1_n, *slen2_n,
        *slen1_tab, *slen2_tab, *scale_short, *scale_long, *huf_tbl_noESC, *version_bytes, *ht_xlen,
        *ht_linmax, *ht_off, *ht_code, *ht_hlen, *largetbl, *table23, *table56, *t32l, *t33l;
    const double *fht_twiddle, *fht_costab, *enwindow, *mdct_win, *ma_tab, *ma_table1, *ma_table2, *ma_table3,
        *hpf_fircoef;
    // derived on the host at create time (device pointers)
    const int32_t *s3off_l, *s3off_s;       // start offset of partition b inside s3_ll / s3_ss
    const int32_t *lineoff_l, *lineoff_s;   // first FFT line of partition b
    const double *amp_by_out;               // [32] polyphase output i is scaled by this (1.0: not at all): amp_filter through mdct_order
    int amp_mask;                           // bit i: amp_by_out[i] != 1.0
    const int32_t *psy_fold; int psy_maxlen_l; // psyA's partition fold: [64][3] marks | partition numbers of the lane's 8 lines; longest long partition
    const int32_t *bvtab;                   // count_bits: [289] by big_values / 2: a1 | a2 << 10 | region0_count << 20 | region1_count << 24 | sfb_count1 << 27
    const void *qtabs_img;                  // the quantization kernels' LDS tables (QuantTabs, k_quant.h) as one prebuilt image: a workgroup copies it flat instead of gathering it from the source tables
    const int32_t *fold_marks, *wpre;       // calc_noise's fold: band-start / band-end marks per lane [2][64]; widest band among bands 0 .. b [24 long | 40 short]
};

// Per-granule-channel side information produced by the quantization kernel and consumed by the
// bit-packing kernel (the subset of the reference's GrInfo that reaches the bitstream).
enum { BS_TAB_MAX = 24 };
struct GrSide {
    int32_t part2_3_length, part2_length, big_values, count1, global_gain, scalefac_compress, block_type;
    int32_t table_select[3], subblock_gain[3];
    int32_t region0_count, region1_count, preflag, scalefac_scale, count1table_select;
    int32_t sfbmax, sfbdivide;
    int32_t active;                          // 1 if the granule had energy (bin search / outer loop ran)
    int32_t bs_start, bs_step_in, bs_gain;   // bin-search seed used + resulting gain (seed-chain validation)
    int32_t targ_bits;
    int32_t scfsi;                           // gr1 only: bit i = scfsi[ch][i]
    int32_t scalefac[SFBMAX];
    // bin-search memo: (gain << 24 | bits) of every count_bits evaluation made with all-zero scalefactors;
    // the seed-chain validation replays the search from these and only recomputes on a miss
    int32_t bs_ntab, bs_tab[BS_TAB_MAX];
    // what each memoised evaluation assigned (table_select / region counts are only written for non-empty regions, so
    // their final values depend on the search path) and the resulting state at the end of the bin search
    int32_t bs_asg[BS_TAB_MAX], bs_state;
    int32_t mode_ext;                        // joint stereo: the frame's M/S decision (0 = L/R, 2 = M/S; Encoder.js:520-561), same in all records of a frame
};

}  // namespace lhip
