/*
 * Host-side parameter resolution and table construction for the MI355X encode path.
 *
 * Given the three public knobs of the drop-in API -- (channels, samplerate, kbps) --
 * this module resolves every scalar the hot path needs and builds every lookup
 * table with the host's own Math.* (V8), so the numbers are bit-identical to what
 * the reference computes at `new Mp3Encoder(...)` time.  The result is packed into
 * one "LHTB" blob (see packBlob) that the C-ABI library uploads to HBM once.
 *
 * Behaviour restated (not copied) from the reference's one-time init:
 *   index.js:100-111        fixed Mp3Encoder configuration
 *   Lame.js:747-1371        lame_init_params (CBR, vbr_off branch only)
 *   Lame.js:470-558         polyphase low-pass amplitudes (amp_filter)
 *   Lame.js:560-690         quality map (quality 3)
 *   Presets.js:226-358      ABR preset row applied for CBR
 *   QuantizePVT.js:247-414  ATH per sfb, pow43/adj43/ipow20/pow20, long/shortfact
 *   Takehiro.js:1141-1172   bv_scf region split table
 *   PsyModel.js:2249-2822   partitions, spreading functions, ATH per partition, eql_w
 *   FFT.js:226-242          analysis windows; FFT.js:31-115 twiddle recurrence
 *
 * Supported envelope (everything else throws): CBR, every MPEG-1 / MPEG-2 / MPEG-2.5 sample rate
 * 1 or 2 channels, with out_samplerate == in_samplerate or an integer multiple below it (resampling by a non-integer
 * ratio makes the reference feed itself NaN samples; refused).
 */
'use strict';

const C = require('./constants.json');

const SBMAX_l = 22, SBMAX_s = 13, SBPSY_l = 21, SBPSY_s = 12, PSFB21 = 6, PSFB12 = 6;
const CBANDS = 64, BLKSIZE = 1024, BLKSIZE_s = 256, HBLKSIZE = 513;
const IXMAX_VAL = 8206, PRECALC_SIZE = IXMAX_VAL + 2, Q_MAX = 256 + 1, Q_MAX2 = 116;
const FLOAT_MAX = 3.4028235e+38;
const MODE_STEREO = 0, MODE_JOINT_STEREO = 1, MODE_MONO = 3;

function f32(n) { return new Float32Array(n); }
function i32(n) { return new Int32Array(n); }

/* ------------------------------------------------------------------ */
/* parameter resolution                                                */
/* ------------------------------------------------------------------ */

function nearestBitrateFullIndex(bitrate) {
    const t = C.full_bitrate_table;
    let upper = 16, lower = 16, upper_k = t[16], lower_k = t[16];
    for (let b = 0; b < 16; b++) {
        if (Math.max(bitrate, t[b + 1]) != bitrate) {
            upper_k = t[b + 1]; upper = b + 1; lower_k = t[b]; lower = b;
            break;
        }
    }
    return (upper_k - bitrate) > (bitrate - lower_k) ? lower : upper;
}

function suggestedSampleFreq(lowpassfreq, in_rate) {
    /* same decision table as the reference's out_samplerate chooser, Lame.js:285-364 */
    const grid = [48000, 44100, 32000, 24000, 22050, 16000, 12000, 11025, 8000];
    let s = 44100;
    for (const g of grid) { if (in_rate >= g) { s = g; break; } }
    if (lowpassfreq == -1) return s;
    const cuts = [[15960, 44100], [15250, 32000], [11220, 24000], [9970, 22050],
        [7230, 16000], [5420, 12000], [4510, 11025], [3970, 8000]];
    for (const [lim, fs] of cuts) if (lowpassfreq <= lim) s = fs;
    if (in_rate < s) {
        const up = [[44100, 48000], [32000, 44100], [24000, 32000], [22050, 24000],
            [16000, 22050], [12000, 16000], [11025, 12000], [8000, 11025]];
        for (const [lim, fs] of up) if (in_rate > lim) return fs;
        return 8000;
    }
    return s;
}

function resolveParams(channels, samplerate, kbps, opts) {
    const p = {};
    p.channels_in = channels;
    p.mode = (channels == 1) ? MODE_MONO : MODE_STEREO;      /* index.js:105, Lame.js:759-761 */
    /* extension (SURVEY.md 8f #3): the reference's joint-stereo path, which its Mp3Encoder never selects (index.js:105) */
    if (opts && opts.jointStereo && channels == 2) p.mode = MODE_JOINT_STEREO;
    /* extension (SURVEY.md 8f #4): the bit reservoir, which index.js:108 switches off */
    p.disable_reservoir = (opts && opts.reservoir) ? 0 : 1;
    p.channels_out = (p.mode == MODE_MONO) ? 1 : 2;
    p.in_samplerate = samplerate;
    let brate = kbps;

    /* lowpass from bitrate (Lame.js:838-885) */
    let lowpass = C.lowpass_freq_map[nearestBitrateFullIndex(brate)][1];
    if (p.mode == MODE_MONO) lowpass *= 1.5;
    let lowpassfreq = lowpass | 0;
    if (2 * lowpassfreq > samplerate) lowpassfreq = samplerate / 2;
    const out_samplerate = suggestedSampleFreq(lowpassfreq | 0, samplerate);
    lowpassfreq = Math.min(20500, lowpassfreq);
    lowpassfreq = Math.min(out_samplerate / 2, lowpassfreq);
    p.out_samplerate = out_samplerate;
    p.lowpassfreq = lowpassfreq;

    /* resampler set-up (Lame.js:943, 1719-1763).  Only integer ratios are in the envelope: for any other ratio the
     * reference's filter_l / 2 is 15.5, its buffer positions become fractional, typed-array reads return undefined and
     * the encoder is fed NaN samples (at the latest in flush()) -- there is no well-defined output to reproduce. */
    p.resample_ratio = samplerate / out_samplerate;
    p.rs_filter_l = 0; p.rs_bpc = 0; p.rs_blackfilt = f32(1);
    if (p.resample_ratio < .9999 || p.resample_ratio > 1.0001) {
        const ratio = p.resample_ratio;
        const intratio = (Math.abs(ratio - Math.floor(.5 + ratio)) < .0001) ? 1 : 0;
        if (!intratio)
            throw new Error('lamejs_amd: (' + channels + ',' + samplerate + ',' + kbps + ') would resample to ' + out_samplerate +
                ' Hz by the non-integer ratio ' + ratio + '; the reference feeds itself NaN samples there (fractional buffer positions), not supported');
        const gcd = (i, j) => (j != 0 ? gcd(j, i % j) : i);
        let bpc = out_samplerate / gcd(out_samplerate, samplerate);
        if (bpc > 320) bpc = 320;                                            /* LameInternalFlags.BPC */
        let fcn = 1.00 / ratio;
        if (fcn > 1.00) fcn = 1.00;
        let filter_l = 31;
        if (0 == filter_l % 2) --filter_l;
        filter_l += intratio;
        const BLACKSIZE = filter_l + 1;
        const blackman = function (x, fcn, l) {                               /* Lame.js:1698-1717 */
            const wcn = (Math.PI * fcn);
            x /= l;
            if (x < 0) x = 0;
            if (x > 1) x = 1;
            const x2 = x - .5;
            const bkwn = 0.42 - 0.5 * Math.cos(2 * x * Math.PI) + 0.08 * Math.cos(4 * x * Math.PI);
            if (Math.abs(x2) < 1e-9) return (wcn / Math.PI);
            else return (bkwn * Math.sin(l * wcn * x2) / (Math.PI * l * x2));
        };
        const bf = f32((2 * bpc + 1) * BLACKSIZE);
        for (let j = 0; j <= 2 * bpc; j++) {
            let sum = 0.;
            const offset = (j - bpc) / (2. * bpc);
            const row = bf.subarray(j * BLACKSIZE, (j + 1) * BLACKSIZE);
            for (let i = 0; i <= filter_l; i++) sum += row[i] = blackman(i - offset, fcn, filter_l);   /* adds the unrounded f64 */
            for (let i = 0; i <= filter_l; i++) row[i] /= sum;
        }
        p.rs_filter_l = filter_l; p.rs_bpc = bpc; p.rs_blackfilt = bf;
    }
    /* SmpFrqIndex (Lame.js:369-403): MPEG-1 for 32/44.1/48 kHz, "version 0" (LSF) for MPEG-2 and MPEG-2.5 rates */
    const SFI = { 44100: [1, 0], 48000: [1, 1], 32000: [1, 2], 22050: [0, 0], 24000: [0, 1], 16000: [0, 2],
        11025: [0, 0], 12000: [0, 1], 8000: [0, 2] };
    if (!SFI[out_samplerate])
        throw new Error('lamejs_amd: unsupported sample rate ' + out_samplerate);
    p.version = SFI[out_samplerate][0];
    const sr_idx = SFI[out_samplerate][1];
    p.samplerate_index = sr_idx;
    p.mode_gr = out_samplerate <= 24000 ? 1 : 2;            /* Lame.js:936 */
    p.framesize = 576 * p.mode_gr;

    /* nearest legal bitrate + index (Lame.js:408-444); MPEG-2.5 rates use their own bitrate row */
    {
        const bt = C.bitrate_table[out_samplerate < 16000 ? 2 : p.version];
        let best = bt[1];
        for (let i = 2; i <= 14; i++)
            if (bt[i] > 0 && Math.abs(bt[i] - brate) < Math.abs(best - brate)) best = bt[i];
        brate = best;
        p.bitrate_index = bt.indexOf(brate);
    }
    p.brate = brate;

    /* polyphase lowpass: amp_filter (Lame.js:1011-1040, 470-558); no highpass */
    let lowpass2 = 2. * lowpassfreq, lowpass1 = (1 - 0.00) * 2. * lowpassfreq;
    lowpass1 /= out_samplerate; lowpass2 /= out_samplerate;
    {
        let lowpass_band = 32, minband = 999;
        if (lowpass1 > 0) {
            for (let band = 0; band <= 31; band++) {
                const freq = band / 31.0;
                if (freq >= lowpass2) lowpass_band = Math.min(lowpass_band, band);
                if (lowpass1 < freq && freq < lowpass2) minband = Math.min(minband, band);
            }
            lowpass1 = ((minband == 999 ? lowpass_band : minband) - .75) / 31.0;
            lowpass2 = lowpass_band / 31.0;
        }
        p.amp_filter = f32(32);
        for (let band = 0; band < 32; band++) {
            const freq = band / 31.0;
            let fc2 = 1.0;
            if (lowpass2 > lowpass1) {
                const x = (freq - lowpass1) / (lowpass2 - lowpass1 + 1e-20);
                fc2 = x > 1.0 ? 0.0 : (x <= 0.0 ? 1.0 : Math.cos(Math.PI / 2 * x));
            }
            p.amp_filter[band] = 1.0 * fc2;
        }
    }

    /* scalefactor band edges (Lame.js:1079-1101); note the fractional pseudo-band
     * starts truncated by the Int32 store -- that truncation is the reference's behaviour */
    {
        const j = sr_idx + 3 * p.version + 6 * (out_samplerate < 16000 ? 1 : 0);
        p.sfb_l = i32(SBMAX_l + 1); p.sfb_s = i32(SBMAX_s + 1);
        p.psfb21 = i32(PSFB21 + 1); p.psfb12 = i32(PSFB12 + 1);
        for (let i = 0; i < SBMAX_l + 1; i++) p.sfb_l[i] = C.sfBandIndex[j].l[i];
        for (let i = 0; i < PSFB21 + 1; i++) {
            const size = (p.sfb_l[22] - p.sfb_l[21]) / PSFB21;
            p.psfb21[i] = p.sfb_l[21] + i * size;
        }
        p.psfb21[PSFB21] = 576;
        for (let i = 0; i < SBMAX_s + 1; i++) p.sfb_s[i] = C.sfBandIndex[j].s[i];
        for (let i = 0; i < PSFB12 + 1; i++) {
            const size = (p.sfb_s[13] - p.sfb_s[12]) / PSFB12;
            p.psfb12[i] = p.sfb_s[12] + i * size;
        }
        p.psfb12[PSFB12] = 192;
    }
    /* Lame.js:1103-1110 */
    if (p.version == 1) p.sideinfo_len = (p.channels_out == 1) ? 4 + 17 : 4 + 32;
    else p.sideinfo_len = (p.channels_out == 1) ? 4 + 9 : 4 + 17;

    /* CBR: ABR preset row for this bitrate (Lame.js:1202-1230, Presets.js:270-357) */
    const row = C.abr_switch_map[nearestBitrateFullIndex(brate)];
    /* kbps quant q_s safejoint nsmsfix st_lrm st_s ns-bass scale msk ath_lwr ath_curve interch sfscale */
    let exp_nspsytune = 0;
    if (row[3] > 0) exp_nspsytune |= 2;
    p.noise_shaping = (row[13] > 0) ? 2 : 0;
    if (Math.abs(row[7]) > 0) throw new Error('ns-bass presets not supported');
    p.quant_comp = row[1];
    p.quant_comp_short = row[2];
    p.attackthre = row[5];
    p.attackthre_s = row[6];
    p.scale = row[8];
    const maskingadjust = row[9];
    const maskingadjust_short = (row[9] > 0) ? row[9] * .9 : row[9] * 1.1;
    p.ATHlower = -row[10] / 10.;
    p.ATHcurve = row[11];
    p.interChRatio = row[12];
    /* Presets.js:288-292 (nsmsfix); Lame.js:1322: < 0 -> 0; PsyModel.js:2738-2744: 0 -> NS_MSFIX (3.5), or 1.0 with safejoint */
    p.msfix = row[4];
    if (p.msfix < 0) p.msfix = 0;
    if (!(Math.abs(p.msfix) > 0.0)) p.msfix = ((exp_nspsytune & 2) != 0) ? 1.0 : 3.5;
    p.mask_adjust = maskingadjust;
    p.mask_adjust_short = maskingadjust_short;
    /* CBRNewIterationLoop.js:56-65: masking_lower = 10^(0.1 * mask_adjust[_short]) */
    p.masking_lower_long = Math.pow(10.0, p.mask_adjust * 0.1);
    p.masking_lower_short = Math.pow(10.0, p.mask_adjust_short * 0.1);

    /* quality 3 (Lame.js:626-636) */
    p.psymodel = 1;
    if (p.noise_shaping == 0) p.noise_shaping = 1;
    p.noise_shaping_amp = 1;
    p.noise_shaping_stop = 1;
    p.subblock_gain = 1;
    p.use_best_huffman = 1;
    p.full_outer_loop = 0;
    p.substep_shaping = 0;
    p.sfb21_extra = 0;

    p.ATH_useAdjust = 3;
    p.ATH_aaSensitivityP = Math.pow(10.0, 0.0 / -10.0);
    /* short_block_allowed -> coupled for (joint) stereo (Lame.js:1302-1317) */
    p.short_blocks_coupled = (p.mode == MODE_STEREO || p.mode == MODE_JOINT_STEREO) ? 1 : 0;
    p.exp_nspsytune = exp_nspsytune | 1;
    p.ATHtype = 4;
    p.athaa_loudapprox = 2;
    if (p.interChRatio < 0) p.interChRatio = 0;
    p.useTemporal = 1;
    p.frac_SpF = (((p.version + 1) * 72000 * brate) % out_samplerate) | 0;

    /* header constants (LameGlobalFlags defaults + lame_init_old: original=1) */
    p.copyright = 0; p.original = 1; p.emphasis = 0; p.extension = 0; p.error_protection = 0;
    return p;
}

/* ------------------------------------------------------------------ */
/* ATH                                                                 */
/* ------------------------------------------------------------------ */

function ATHformula_GB(f, value) {
    if (f < -.3) f = 3410;
    f /= 1000;
    f = Math.max(0.1, f);
    return 3.640 * Math.pow(f, -0.8) - 6.800 * Math.exp(-0.6 * Math.pow(f - 3.4, 2.0))
        + 6.000 * Math.exp(-0.15 * Math.pow(f - 8.7, 2.0))
        + (0.6 + 0.04 * value) * 0.001 * Math.pow(f, 4.0);
}

function buildTables(p) {
    const T = {};
    const sfreq0 = p.out_samplerate;
    const ATHformula = (f) => ATHformula_GB(f, p.ATHcurve);     /* ATHtype 4 */
    const ATHmdct = (f) => {
        let ath = ATHformula(f);
        ath -= 100;                                             /* NSATHSCALE */
        return Math.pow(10.0, ath / 10.0 + p.ATHlower);
    };

    /* --- ATH per scalefactor band (QuantizePVT.js:247-328) --- */
    T.ATH_l = f32(SBMAX_l); T.ATH_s = f32(SBMAX_s); T.ATH_psfb21 = f32(PSFB21); T.ATH_psfb12 = f32(PSFB12);
    const bandMin = (dst, k, start, end, div) => {
        dst[k] = FLOAT_MAX;
        for (let i = start; i < end; i++) dst[k] = Math.min(dst[k], ATHmdct(i * sfreq0 / div));
    };
    for (let sfb = 0; sfb < SBMAX_l; sfb++) bandMin(T.ATH_l, sfb, p.sfb_l[sfb], p.sfb_l[sfb + 1], 2 * 576);
    for (let sfb = 0; sfb < PSFB21; sfb++) bandMin(T.ATH_psfb21, sfb, p.psfb21[sfb], p.psfb21[sfb + 1], 2 * 576);
    for (let sfb = 0; sfb < SBMAX_s; sfb++) {
        bandMin(T.ATH_s, sfb, p.sfb_s[sfb], p.sfb_s[sfb + 1], 2 * 192);
        T.ATH_s[sfb] *= (p.sfb_s[sfb + 1] - p.sfb_s[sfb]);
    }
    for (let sfb = 0; sfb < PSFB12; sfb++) {
        bandMin(T.ATH_psfb12, sfb, p.psfb12[sfb], p.psfb12[sfb + 1], 2 * 192);
        T.ATH_psfb12[sfb] *= (p.sfb_s[13] - p.sfb_s[12]);
    }
    T.ATH_floor = 10. * Math.log10(ATHmdct(-1.));

    /* --- quantizer power tables (QuantizePVT.js:343-358) --- */
    T.pow43 = f32(PRECALC_SIZE); T.adj43 = f32(PRECALC_SIZE);
    T.ipow20 = f32(Q_MAX); T.pow20 = f32(Q_MAX + Q_MAX2 + 1);
    T.pow43[0] = 0.0;
    for (let i = 1; i < PRECALC_SIZE; i++) T.pow43[i] = Math.pow(i, 4.0 / 3.0);
    {
        let i;
        for (i = 0; i < PRECALC_SIZE - 1; i++)
            T.adj43[i] = ((i + 1) - Math.pow(0.5 * (T.pow43[i] + T.pow43[i + 1]), 0.75));
        T.adj43[i] = 0.5;
    }
    for (let i = 0; i < Q_MAX; i++) T.ipow20[i] = Math.pow(2.0, (i - 210) * -0.1875);
    for (let i = 0; i <= Q_MAX + Q_MAX2; i++) T.pow20[i] = Math.pow(2.0, (i - 210 - Q_MAX2) * 0.25);

    /* --- region split table (Takehiro.js:1141-1172) --- */
    T.bv_scf = i32(576);
    for (let i = 2; i <= 576; i += 2) {
        let scfb_anz = 0, bv;
        while (p.sfb_l[++scfb_anz] < i);
        bv = C.subdv_table[scfb_anz][0];
        while (p.sfb_l[bv + 1] > i) bv--;
        if (bv < 0) bv = C.subdv_table[scfb_anz][0];
        T.bv_scf[i - 2] = bv;
        bv = C.subdv_table[scfb_anz][1];
        while (p.sfb_l[bv + T.bv_scf[i - 2] + 2] > i) bv--;
        if (bv < 0) bv = C.subdv_table[scfb_anz][1];
        T.bv_scf[i - 1] = bv;
    }

    /* --- long/shortfact (QuantizePVT.js:362-409) --- */
    {
        const dec = (sh) => { let i = (p.exp_nspsytune >> sh) & 63; if (i >= 32) i -= 64; return i; };
        const bass = Math.pow(10, dec(2) / 4.0 / 10.0);
        const alto = Math.pow(10, dec(8) / 4.0 / 10.0);
        const treble = Math.pow(10, dec(14) / 4.0 / 10.0);
        const sfb21 = treble * Math.pow(10, dec(20) / 4.0 / 10.0);
        T.longfact = f32(SBMAX_l); T.shortfact = f32(SBMAX_s);
        for (let i = 0; i < SBMAX_l; i++) T.longfact[i] = i <= 6 ? bass : i <= 13 ? alto : i <= 20 ? treble : sfb21;
        for (let i = 0; i < SBMAX_s; i++) T.shortfact[i] = i <= 5 ? bass : i <= 10 ? alto : i <= 11 ? treble : sfb21;
    }

    /* --- psychoacoustic partitions (PsyModel.js:2363-2460, 2537-2760) --- */
    const LN_TO_LOG10 = 0.2302585093, LOG10 = 2.30258509299404568402, DELBARK = .34;
    const freq2bark = (freq) => {
        if (freq < 0) freq = 0;
        freq = freq * 0.001;
        return 13.0 * Math.atan(.76 * freq) + 3.5 * Math.atan(freq * freq / (7.5 * 7.5));
    };
    const s3_func = (bark) => {
        let tempx = bark, x, tempy;
        if (tempx >= 0) tempx *= 3; else tempx *= 1.5;
        if (tempx >= 0.5 && tempx <= 2.5) { const t = tempx - 0.5; x = 8.0 * (t * t - 2.0 * t); } else x = 0.0;
        tempx += 0.474;
        tempy = 15.811389 + 7.5 * tempx - 17.5 * Math.sqrt(1.0 + tempx * tempx);
        if (tempy <= -60.0) return 0.0;
        tempx = Math.exp((x + tempy) * LN_TO_LOG10);
        tempx /= .6609193;
        return tempx;
    };
    const bval = f32(CBANDS), bval_width = f32(CBANDS), norm = f32(CBANDS);

    function init_numline(numlines, bo, bm, bo_w, mld, blksize, scalepos, deltafreq, sbmax) {
        const b_frq = f32(CBANDS + 1);
        const sample_freq_frac = sfreq0 / (sbmax > 15 ? 2 * 576 : 2 * 192);
        const partition = i32(HBLKSIZE);
        const sfreq = sfreq0 / blksize;
        let i, j = 0, ni = 0;
        for (i = 0; i < CBANDS; i++) {
            const bark1 = freq2bark(sfreq * j);
            b_frq[i] = sfreq * j;
            let j2;
            for (j2 = j; freq2bark(sfreq * j2) - bark1 < DELBARK && j2 <= blksize / 2; j2++);
            numlines[i] = j2 - j;
            ni = i + 1;
            while (j < j2) partition[j++] = i;
            if (j > blksize / 2) { j = blksize / 2; ++i; break; }
        }
        b_frq[i] = sfreq * j;
        for (let sfb = 0; sfb < sbmax; sfb++) {
            const start = scalepos[sfb], end = scalepos[sfb + 1];
            let i1 = 0 | Math.floor(.5 + deltafreq * (start - .5));
            if (i1 < 0) i1 = 0;
            let i2 = 0 | Math.floor(.5 + deltafreq * (end - .5));
            if (i2 > blksize / 2) i2 = blksize / 2;
            bm[sfb] = (partition[i1] + partition[i2]) / 2;
            bo[sfb] = partition[i2];
            const f_tmp = sample_freq_frac * end;
            bo_w[sfb] = (f_tmp - b_frq[bo[sfb]]) / (b_frq[bo[sfb] + 1] - b_frq[bo[sfb]]);
            if (bo_w[sfb] < 0) bo_w[sfb] = 0; else if (bo_w[sfb] > 1) bo_w[sfb] = 1;
            /* stereo demasking threshold of the band (PsyModel.js:2432-2438) */
            let arg = freq2bark(sfreq * scalepos[sfb] * deltafreq);
            arg = (Math.min(arg, 15.5) / 15.5);
            mld[sfb] = Math.pow(10.0, 1.25 * (1 - Math.cos(Math.PI * arg)) - 2.5);
        }
        j = 0;
        for (let k = 0; k < ni; k++) {
            const w = numlines[k];
            bval[k] = .5 * (freq2bark(sfreq * j) + freq2bark(sfreq * (j + w - 1)));
            bval_width[k] = freq2bark(sfreq * (j + w - .5)) - freq2bark(sfreq * (j - .5));
            j += w;
        }
        return ni;
    }

    function init_s3_values(s3ind, npart) {
        const s3 = []; for (let i = 0; i < CBANDS; i++) s3.push(f32(CBANDS));
        for (let i = 0; i < npart; i++)
            for (let j = 0; j < npart; j++) {
                const v = s3_func(bval[i] - bval[j]) * bval_width[j];
                s3[i][j] = v * norm[i];
            }
        let nz = 0;
        for (let i = 0; i < npart; i++) {
            let j;
            for (j = 0; j < npart; j++) if (s3[i][j] > 0.0) break;
            s3ind[2 * i] = j;
            for (j = npart - 1; j > 0; j--) if (s3[i][j] > 0.0) break;
            s3ind[2 * i + 1] = j;
            nz += (s3ind[2 * i + 1] - s3ind[2 * i] + 1);
        }
        const out = f32(nz);
        let k = 0;
        for (let i = 0; i < npart; i++)
            for (let j = s3ind[2 * i]; j <= s3ind[2 * i + 1]; j++) out[k++] = s3[i][j];
        return out;
    }

    T.numlines_l = i32(CBANDS); T.numlines_s = i32(CBANDS); T.rnumlines_l = f32(CBANDS);
    T.bo_l = i32(SBMAX_l); T.bm_l = i32(SBMAX_l); T.bo_s = i32(SBMAX_s); T.bm_s = i32(SBMAX_s);
    T.bo_l_weight = f32(SBMAX_l); T.bo_s_weight = f32(SBMAX_s);
    T.mld_l = f32(SBMAX_l); T.mld_s = f32(SBMAX_s);
    T.s3ind = i32(2 * CBANDS); T.s3ind_s = i32(2 * CBANDS);
    T.ATH_cb_l = f32(CBANDS); T.ATH_cb_s = f32(CBANDS);

    const bvl_a = 13, bvl_b = 24, snr_l_a = 0, snr_l_b = 0, snr_s_a = -8.25, snr_s_b = -4.5;
    T.npart_l = init_numline(T.numlines_l, T.bo_l, T.bm_l, T.bo_l_weight, T.mld_l, BLKSIZE, p.sfb_l,
        BLKSIZE / (2.0 * 576), SBMAX_l);
    for (let i = 0; i < T.npart_l; i++) {
        let snr = snr_l_a;
        if (bval[i] >= bvl_a)
            snr = snr_l_b * (bval[i] - bvl_a) / (bvl_b - bvl_a) + snr_l_a * (bvl_b - bval[i]) / (bvl_b - bvl_a);
        norm[i] = Math.pow(10.0, snr / 10.0);
        T.rnumlines_l[i] = T.numlines_l[i] > 0 ? 1.0 / T.numlines_l[i] : 0;
    }
    T.s3_ll = init_s3_values(T.s3ind, T.npart_l);
    {
        let j = 0;
        for (let i = 0; i < T.npart_l; i++) {
            let x = FLOAT_MAX;
            for (let k = 0; k < T.numlines_l[i]; k++, j++) {
                const freq = sfreq0 * j / (1000.0 * BLKSIZE);
                let level = ATHformula(freq * 1000) - 20;
                level = Math.pow(10., 0.1 * level);
                level *= T.numlines_l[i];
                if (x > level) x = level;
            }
            T.ATH_cb_l[i] = x;
        }
    }
    T.npart_s = init_numline(T.numlines_s, T.bo_s, T.bm_s, T.bo_s_weight, T.mld_s, BLKSIZE_s, p.sfb_s,
        BLKSIZE_s / (2.0 * 192), SBMAX_s);
    {
        let j = 0;
        for (let i = 0; i < T.npart_s; i++) {
            let snr = snr_s_a;
            if (bval[i] >= bvl_a)
                snr = snr_s_b * (bval[i] - bvl_a) / (bvl_b - bvl_a) + snr_s_a * (bvl_b - bval[i]) / (bvl_b - bvl_a);
            norm[i] = Math.pow(10.0, snr / 10.0);
            let x = FLOAT_MAX;
            for (let k = 0; k < T.numlines_s[i]; k++, j++) {
                const freq = sfreq0 * j / (1000.0 * BLKSIZE_s);
                let level = ATHformula(freq * 1000) - 20;
                level = Math.pow(10., 0.1 * level);
                level *= T.numlines_s[i];
                if (x > level) x = level;
            }
            T.ATH_cb_s[i] = x;
        }
    }
    T.s3_ss = init_s3_values(T.s3ind_s, T.npart_s);
    /* spread only from npart_l bands (PsyModel.js:2718-2721) */
    for (let b = 0; b < T.npart_l; b++)
        if (T.s3ind[2 * b + 1] > T.npart_l - 1) T.s3ind[2 * b + 1] = T.npart_l - 1;

    /* mask_add shortcuts (PsyModel.js:376-380) */
    T.ma_max_i1 = Math.pow(10, (8 + 1) / 16.0);
    T.ma_max_i2 = Math.pow(10, (23 + 1) / 16.0);
    T.ma_max_m = Math.pow(10, 15 / 10.0);

    /* FFT analysis windows (FFT.js:226-242) */
    T.window = f32(BLKSIZE); T.window_s = f32(BLKSIZE_s / 2);
    for (let i = 0; i < BLKSIZE; i++)
        T.window[i] = (0.42 - 0.5 * Math.cos(2 * Math.PI * (i + .5) / BLKSIZE)
            + 0.08 * Math.cos(4 * Math.PI * (i + .5) / BLKSIZE));
    for (let i = 0; i < BLKSIZE_s / 2; i++)
        T.window_s[i] = (0.5 * (1.0 - Math.cos(2.0 * Math.PI * (i + 0.5) / BLKSIZE_s)));

    /* FHT twiddles: the f64 recurrence of FFT.js:72-111, unrolled into a table.
     * pass t (k4 = 4,16,64,256 -> kx = 2,8,32,128), entry i = 1..kx-1 holds (c1,s1,c2,s2) as the
     * butterflies of that i see them.  Offsets: pass0 @0 (1 entry), pass1 @1 (7), pass2 @8 (31), pass3 @39 (127) */
    T.fht_twiddle = new Float64Array(4 * 166);
    {
        let off = 0, kx = 2;
        for (let t = 0; t < 4; t++, kx <<= 2) {
            const ct = C.fht_costab[2 * t], st = C.fht_costab[2 * t + 1];
            let c1 = ct, s1 = st;
            for (let i = 1; i < kx; i++) {
                const c2 = 1 - (2 * s1) * s1;
                const s2 = (2 * s1) * c1;
                T.fht_twiddle.set([c1, s1, c2, s2], 4 * off); off++;
                const c2b = c1;
                c1 = c2b * ct - s1 * st;
                s1 = c2b * st + s1 * ct;
            }
        }
    }

    /* temporal masking decay for short-block xmin smoothing (PsyModel.js:2701-2703) */
    T.decay = Math.exp(-1.0 * LOG10 / (0.01 * sfreq0 / 192.0));

    /* equal-loudness weights (PsyModel.js:2735-2752) */
    T.eql_w = f32(BLKSIZE / 2);
    {
        const freq_inc = p.out_samplerate / BLKSIZE;
        let eql_balance = 0.0, freq = 0.0;
        for (let i = 0; i < BLKSIZE / 2; ++i) {
            freq += freq_inc;
            T.eql_w[i] = 1. / Math.pow(10, ATHformula(freq) / 10);
            eql_balance += T.eql_w[i];
        }
        eql_balance = 1.0 / eql_balance;
        for (let i = BLKSIZE / 2; --i >= 0;) T.eql_w[i] *= eql_balance;
    }
    T.VO_SCALE = (1. / (14752 * 14752) / (BLKSIZE / 2));
    return T;
}

/* ------------------------------------------------------------------ */
/* blob packing                                                        */
/* ------------------------------------------------------------------ */

const DT_I32 = 1, DT_F32 = 2, DT_F64 = 3;
const ENTRY_BYTES = 48;           /* char name[32]; u32 dtype,count,offset,pad */

function packBlob(entries) {
    /* entries: [name, typedArray] ; layout: header(16) | directory | 8-byte aligned payloads */
    const n = entries.length;
    let off = 16 + n * ENTRY_BYTES;
    off = (off + 7) & ~7;
    const offs = [];
    for (const [, a] of entries) { offs.push(off); off += (a.byteLength + 7) & ~7; }
    const buf = Buffer.alloc(off);
    buf.writeUInt32LE(0x4254484c, 0);  /* 'LHTB' */
    buf.writeUInt32LE(1, 4);
    buf.writeUInt32LE(n, 8);
    buf.writeUInt32LE(off, 12);
    entries.forEach(([name, a], k) => {
        const e = 16 + k * ENTRY_BYTES;
        if (name.length > 31) throw new Error('name too long: ' + name);
        buf.write(name, e, 'ascii');
        const dt = (a instanceof Int32Array) ? DT_I32 : (a instanceof Float32Array) ? DT_F32 :
            (a instanceof Float64Array) ? DT_F64 : 0;
        if (!dt) throw new Error('bad array type for ' + name);
        buf.writeUInt32LE(dt, e + 32);
        buf.writeUInt32LE(a.length, e + 36);
        buf.writeUInt32LE(offs[k], e + 40);
        Buffer.from(a.buffer, a.byteOffset, a.byteLength).copy(buf, offs[k]);
    });
    return buf;
}

function I(v) { return Int32Array.from(Array.isArray(v) ? v : [v]); }
function D(v) { return Float64Array.from(Array.isArray(v) ? v : [v]); }

function huffmanEntries() {
    /* flatten ht[0..33]: per table xlen, linmax, offset into shared code/len pools (-1 = absent) */
    const xlen = [], linmax = [], off = [], code = [], hlen = [];
    const seen = new Map();
    C.ht.forEach((h, t) => {
        xlen.push(h.xlen); linmax.push(h.linmax);
        if (!h.hlen) { off.push(-1); return; }
        const key = JSON.stringify(h.hlen) + JSON.stringify(h.table);
        if (seen.has(key)) { off.push(seen.get(key)); return; }
        const o = hlen.length;
        seen.set(key, o); off.push(o);
        for (let i = 0; i < h.hlen.length; i++) { hlen.push(h.hlen[i]); code.push(h.table ? h.table[i] : 0); }
    });
    return [['ht_xlen', I(xlen)], ['ht_linmax', I(linmax)], ['ht_off', I(off)],
        ['ht_code', I(code)], ['ht_hlen', I(hlen)],
        ['largetbl', I(C.largetbl)], ['table23', I(C.table23)], ['table56', I(C.table56)],
        ['t32l', I(C.ht[32].hlen)], ['t33l', I(C.ht[33].hlen)]];
}

function buildBlob(channels, samplerate, kbps, opts) {
    const p = resolveParams(channels, samplerate, kbps, opts);
    const T = buildTables(p);
    const cfg_i = {
        channels_out: p.channels_out, mode: p.mode, mode_gr: p.mode_gr, version: p.version,
        samplerate_index: p.samplerate_index, bitrate_index: p.bitrate_index, brate: p.brate,
        out_samplerate: p.out_samplerate, sideinfo_len: p.sideinfo_len, frac_SpF: p.frac_SpF,
        noise_shaping: p.noise_shaping, noise_shaping_amp: p.noise_shaping_amp,
        noise_shaping_stop: p.noise_shaping_stop, subblock_gain: p.subblock_gain,
        use_best_huffman: p.use_best_huffman, full_outer_loop: p.full_outer_loop,
        substep_shaping: p.substep_shaping, sfb21_extra: p.sfb21_extra,
        quant_comp: p.quant_comp, quant_comp_short: p.quant_comp_short,
        short_blocks_coupled: p.short_blocks_coupled, useTemporal: p.useTemporal,
        ATH_useAdjust: p.ATH_useAdjust, athaa_loudapprox: p.athaa_loudapprox,
        copyright: p.copyright, original: p.original, emphasis: p.emphasis, extension: p.extension,
        error_protection: p.error_protection, npart_l: T.npart_l, npart_s: T.npart_s,
        in_samplerate: p.in_samplerate, rs_filter_l: p.rs_filter_l, rs_bpc: p.rs_bpc,
        disable_reservoir: p.disable_reservoir
    };
    const cfg_d = {
        scale: p.scale, attackthre: p.attackthre, attackthre_s: p.attackthre_s,
        interChRatio: p.interChRatio, masking_lower_long: p.masking_lower_long,
        masking_lower_short: p.masking_lower_short, ATH_aaSensitivityP: p.ATH_aaSensitivityP,
        ATH_floor: T.ATH_floor, decay: T.decay, ma_max_i1: T.ma_max_i1, ma_max_i2: T.ma_max_i2,
        ma_max_m: T.ma_max_m, VO_SCALE: T.VO_SCALE, resample_ratio: p.resample_ratio,
        msfix: p.msfix, ATHlower: p.ATHlower
    };
    const entries = [];
    entries.push(['cfg_i_names', Int32Array.from(Buffer.from(Object.keys(cfg_i).join(',') + '\0', 'ascii'))]);
    entries.push(['cfg_i', I(Object.values(cfg_i))]);
    entries.push(['cfg_d_names', Int32Array.from(Buffer.from(Object.keys(cfg_d).join(',') + '\0', 'ascii'))]);
    entries.push(['cfg_d', D(Object.values(cfg_d))]);
    const push = (n, a) => entries.push([n, a]);
    push('amp_filter', p.amp_filter);
    push('rs_blackfilt', p.rs_blackfilt);
    push('sfb_l', p.sfb_l); push('sfb_s', p.sfb_s); push('psfb21', p.psfb21); push('psfb12', p.psfb12);
    push('ATH_l', T.ATH_l); push('ATH_s', T.ATH_s); push('ATH_psfb21', T.ATH_psfb21); push('ATH_psfb12', T.ATH_psfb12);
    push('ATH_cb_l', T.ATH_cb_l); push('ATH_cb_s', T.ATH_cb_s); push('eql_w', T.eql_w);
    push('pow43', T.pow43); push('adj43', T.adj43); push('ipow20', T.ipow20); push('pow20', T.pow20);
    push('bv_scf', T.bv_scf); push('longfact', T.longfact); push('shortfact', T.shortfact);
    push('numlines_l', T.numlines_l); push('numlines_s', T.numlines_s); push('rnumlines_l', T.rnumlines_l);
    push('bo_l', T.bo_l); push('bm_l', T.bm_l); push('bo_s', T.bo_s); push('bm_s', T.bm_s);
    push('bo_l_weight', T.bo_l_weight); push('bo_s_weight', T.bo_s_weight);
    push('mld_l', T.mld_l); push('mld_s', T.mld_s);
    push('s3ind', T.s3ind); push('s3ind_s', T.s3ind_s); push('s3_ll', T.s3_ll); push('s3_ss', T.s3_ss);
    push('window', T.window); push('window_s', T.window_s); push('fht_twiddle', T.fht_twiddle);
    push('fht_costab', D(C.fht_costab)); push('fft_rv_tbl', I(C.fft_rv_tbl));
    push('enwindow', D(C.enwindow));
    push('mdct_win', D([].concat(C.mdct_win[0], C.mdct_win[1], C.mdct_win[2], C.mdct_win[3])));
    push('mdct_order', I(C.mdct_order));
    push('ma_tab', D(C.ma_tab)); push('ma_table1', D(C.ma_table1)); push('ma_table2', D(C.ma_table2));
    push('ma_table3', D(C.ma_table3)); push('hpf_fircoef', D(C.hpf_fircoef));
    push('pretab', I(C.pretab)); push('scfsi_band', I(C.scfsi_band));
    push('slen1_n', I(C.slen1_n)); push('slen2_n', I(C.slen2_n));
    push('slen1_tab', I(C.slen1_tab)); push('slen2_tab', I(C.slen2_tab));
    push('scale_short', I(C.scale_short)); push('scale_long', I(C.scale_long));
    push('huf_tbl_noESC', I(C.huf_tbl_noESC));
    /* ancillary "version" bytes as the reference emits them: string chars coerced by >> (BitStream.js:180-204) */
    push('version_bytes', I(C.lame_short_version.split('').map(ch => (ch >> 0) & 0xff)));
    huffmanEntries().forEach(e => entries.push(e));
    /* what the blob was generated FROM: the first 8 bytes of sha256(tables.js ++ constants.json) -- a cached blob is stale when its generator
     * has changed, whatever the files' modification times say after a checkout (lamejs_amd/__init__.py tables_blob, __graft_entry__.py) */
    push('src_sha256_64', sourceHash());
    return { blob: packBlob(entries), params: p, tables: T };
}

function sourceHash() {
    const fs = require('fs'), path = require('path');
    const h = require('crypto').createHash('sha256');
    h.update(fs.readFileSync(__filename)); h.update(fs.readFileSync(path.join(__dirname, 'constants.json')));
    const d = h.digest();
    return Int32Array.from([d.readInt32LE(0), d.readInt32LE(4)]);
}

module.exports = { buildBlob, resolveParams, buildTables, packBlob, sourceHash };

if (require.main === module) {
    /* CLI: node tables.js <channels> <samplerate> <kbps> <out.bin> [joint] [reservoir] */
    const [ch, sr, kb, out] = process.argv.slice(2), flags = process.argv.slice(6);
    const r = buildBlob(+ch, +sr, +kb, { jointStereo: flags.includes('joint'), reservoir: flags.includes('reservoir') });
    require('fs').writeFileSync(out, r.blob);
    console.log('wrote', out, r.blob.length, 'bytes');
}
