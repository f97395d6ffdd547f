/*
 * TEST TOOL: compare every table/scalar built by lamejs_amd/js/tables.js with the value
 * held by a live reference encoder instance (where the reference exposes it).
 * usage: node tests/tools/check_tables_vs_ref.js   -> prints "OK n checks" or the mismatches, exit code 0/1
 */
'use strict';
const { refEncoder } = require('./ref_harness.js');
const tables = require('../../lamejs_amd/js/tables.js');

let nchk = 0, bad = 0;
function same(a, b) { return Object.is(a, b) || (a === b); }
function cmp(name, mine, ref, n) {
    n = n === undefined ? mine.length : n;
    for (let i = 0; i < n; i++) {
        nchk++;
        if (!same(mine[i], ref[i])) {
            if (bad < 40) console.log('MISMATCH', name, i, mine[i], ref[i]);
            bad++;
        }
    }
}
function cmpS(name, mine, ref) { cmp(name, [mine], [ref]); }

const configs = [[1, 44100, 128], [2, 44100, 128], [2, 44100, 320], [1, 44100, 64], [2, 48000, 192],
    [1, 32000, 96], [2, 44100, 160], [2, 44100, 256], [1, 48000, 320], [2, 32000, 224],
    /* MPEG-2 / MPEG-2.5 (LSF) rates */
    [1, 22050, 64], [2, 24000, 96], [1, 16000, 32], [2, 16000, 48], [1, 11025, 24], [1, 8000, 16], [2, 12000, 32], [1, 22050, 160], [2, 22050, 128],
    /* integer-ratio resampling (in_samplerate = k * out_samplerate) */
    [1, 44100, 32], [2, 44100, 48], [1, 48000, 24], [2, 48000, 64], [1, 32000, 16], [2, 32000, 8], [1, 16000, 8], [2, 24000, 16], [1, 48000, 40], [1, 48000, 8],
    /* joint stereo (extension flag; reference driven with gfp.mode = JOINT_STEREO) */
    [2, 44100, 128, 1], [2, 44100, 96, 1], [2, 48000, 192, 1], [2, 32000, 64, 1], [2, 44100, 320, 1], [2, 22050, 64, 1], [2, 16000, 32, 1]];
for (const [ch, sr, kb, joint] of configs) {
    let r;
    const opts = { jointStereo: !!joint };
    try { r = tables.buildBlob(ch, sr, kb, opts); } catch (e) { console.log('skip', ch, sr, kb, e.message); continue; }
    const e = refEncoder(ch, sr, kb, opts), gfp = e.gfp, gfc = e.gfc, p = r.params, T = r.tables;
    const tag = ch + '/' + sr + '/' + kb + (joint ? '/joint ' : ' ');
    cmpS(tag + 'msfix', p.msfix, gfp.msfix);
    cmp(tag + 'mld_l', T.mld_l, gfc.mld_l);
    cmp(tag + 'mld_s', T.mld_s, gfc.mld_s);
    cmpS(tag + 'out_samplerate', p.out_samplerate, gfp.out_samplerate);
    cmpS(tag + 'mode', p.mode, gfp.mode.ordinal());
    cmpS(tag + 'channels_out', p.channels_out, gfc.channels_out);
    cmpS(tag + 'bitrate_index', p.bitrate_index, gfc.bitrate_index);
    cmpS(tag + 'samplerate_index', p.samplerate_index, gfc.samplerate_index);
    cmpS(tag + 'sideinfo_len', p.sideinfo_len, gfc.sideinfo_len);
    cmpS(tag + 'mode_gr', p.mode_gr, gfc.mode_gr);
    cmpS(tag + 'version', p.version, gfp.version);
    cmpS(tag + 'brate', p.brate, gfp.brate);
    cmpS(tag + 'frac_SpF', p.frac_SpF, gfc.frac_SpF);
    cmpS(tag + 'scale', p.scale, gfp.scale);
    cmpS(tag + 'attackthre', p.attackthre, gfc.nsPsy.attackthre);
    cmpS(tag + 'attackthre_s', p.attackthre_s, gfc.nsPsy.attackthre_s);
    cmpS(tag + 'interChRatio', p.interChRatio, gfp.interChRatio);
    cmpS(tag + 'mask_adjust', p.mask_adjust, gfc.PSY.mask_adjust);
    cmpS(tag + 'mask_adjust_short', p.mask_adjust_short, gfc.PSY.mask_adjust_short);
    cmpS(tag + 'noise_shaping', p.noise_shaping, gfc.noise_shaping);
    cmpS(tag + 'noise_shaping_amp', p.noise_shaping_amp, gfc.noise_shaping_amp);
    cmpS(tag + 'noise_shaping_stop', p.noise_shaping_stop, gfc.noise_shaping_stop);
    cmpS(tag + 'subblock_gain', p.subblock_gain, gfc.subblock_gain);
    cmpS(tag + 'use_best_huffman', p.use_best_huffman, gfc.use_best_huffman);
    cmpS(tag + 'full_outer_loop', p.full_outer_loop, gfc.full_outer_loop);
    cmpS(tag + 'substep_shaping', p.substep_shaping, gfc.substep_shaping);
    cmpS(tag + 'sfb21_extra', !!p.sfb21_extra, !!gfc.sfb21_extra);
    cmpS(tag + 'quant_comp', p.quant_comp, gfp.quant_comp);
    cmpS(tag + 'quant_comp_short', p.quant_comp_short, gfp.quant_comp_short);
    cmpS(tag + 'short_coupled', p.short_blocks_coupled, gfp.short_blocks.ordinal == 1 ? 1 : 0);
    cmpS(tag + 'useTemporal', !!p.useTemporal, !!gfp.useTemporal);
    cmpS(tag + 'useAdjust', p.ATH_useAdjust, gfc.ATH.useAdjust);
    cmpS(tag + 'aaSens', p.ATH_aaSensitivityP, gfc.ATH.aaSensitivityP);
    cmpS(tag + 'loudapprox', p.athaa_loudapprox, gfp.athaa_loudapprox);
    cmpS(tag + 'ATHlower', p.ATHlower, gfp.ATHlower);
    cmpS(tag + 'ATHcurve', p.ATHcurve, gfp.ATHcurve);
    cmpS(tag + 'original', p.original, gfp.original);
    cmpS(tag + 'psymodel', p.psymodel, gfc.psymodel);
    cmp(tag + 'amp_filter', p.amp_filter, gfc.amp_filter);
    cmp(tag + 'sfb_l', p.sfb_l, gfc.scalefac_band.l);
    cmp(tag + 'sfb_s', p.sfb_s, gfc.scalefac_band.s);
    cmp(tag + 'psfb21', p.psfb21, gfc.scalefac_band.psfb21);
    cmp(tag + 'psfb12', p.psfb12, gfc.scalefac_band.psfb12);
    cmp(tag + 'ATH_l', T.ATH_l, gfc.ATH.l);
    cmp(tag + 'ATH_s', T.ATH_s, gfc.ATH.s);
    cmp(tag + 'ATH_psfb21', T.ATH_psfb21, gfc.ATH.psfb21);
    cmp(tag + 'ATH_psfb12', T.ATH_psfb12, gfc.ATH.psfb12);
    cmp(tag + 'ATH_cb_l', T.ATH_cb_l, gfc.ATH.cb_l, T.npart_l);
    cmp(tag + 'ATH_cb_s', T.ATH_cb_s, gfc.ATH.cb_s, T.npart_s);
    cmp(tag + 'eql_w', T.eql_w, gfc.ATH.eql_w);
    cmpS(tag + 'ATH_floor', T.ATH_floor, gfc.ATH.floor);
    cmp(tag + 'adj43', T.adj43, e.qupvt.adj43);
    for (let i = 0; i < 257; i++) cmpS(tag + 'ipow20', T.ipow20[i], e.qupvt.IPOW20(i));
    cmp(tag + 'bv_scf', T.bv_scf, gfc.bv_scf);
    cmp(tag + 'longfact', T.longfact, gfc.nsPsy.longfact);
    cmp(tag + 'shortfact', T.shortfact, gfc.nsPsy.shortfact);
    cmpS(tag + 'npart_l', T.npart_l, gfc.npart_l);
    cmpS(tag + 'npart_s', T.npart_s, gfc.npart_s);
    cmp(tag + 'numlines_l', T.numlines_l, gfc.numlines_l, T.npart_l);
    cmp(tag + 'numlines_s', T.numlines_s, gfc.numlines_s, T.npart_s);
    cmp(tag + 'rnumlines_l', T.rnumlines_l, gfc.rnumlines_l, T.npart_l);
    cmp(tag + 'bo_l', T.bo_l, gfc.bo_l); cmp(tag + 'bm_l', T.bm_l, gfc.bm_l);
    cmp(tag + 'bo_s', T.bo_s, gfc.bo_s); cmp(tag + 'bm_s', T.bm_s, gfc.bm_s);
    cmp(tag + 'bo_l_weight', T.bo_l_weight, gfc.PSY.bo_l_weight);
    cmp(tag + 'bo_s_weight', T.bo_s_weight, gfc.PSY.bo_s_weight);
    for (let b = 0; b < T.npart_l; b++) { cmpS(tag + 's3ind0', T.s3ind[2 * b], gfc.s3ind[b][0]); cmpS(tag + 's3ind1', T.s3ind[2 * b + 1], gfc.s3ind[b][1]); }
    for (let b = 0; b < T.npart_s; b++) { cmpS(tag + 's3inds0', T.s3ind_s[2 * b], gfc.s3ind_s[b][0]); cmpS(tag + 's3inds1', T.s3ind_s[2 * b + 1], gfc.s3ind_s[b][1]); }
    cmpS(tag + 's3_ll.len', T.s3_ll.length, gfc.s3_ll.length); cmp(tag + 's3_ll', T.s3_ll, gfc.s3_ll);
    cmpS(tag + 's3_ss.len', T.s3_ss.length, gfc.s3_ss.length); cmp(tag + 's3_ss', T.s3_ss, gfc.s3_ss);
    cmpS(tag + 'resample_ratio', p.resample_ratio, gfc.resample_ratio);
    if (p.rs_filter_l) {       /* the reference builds its filters on the first fill_buffer_resample call */
        e.encodeBuffer(new Int16Array(64), new Int16Array(64));
        const BL = p.rs_filter_l + 1;
        for (let j = 0; j <= 2 * p.rs_bpc; j++) cmp(tag + 'blackfilt' + j, p.rs_blackfilt.subarray(j * BL, (j + 1) * BL), gfc.blackfilt[j], BL);
        cmpS(tag + 'inbuf_old.len', BL, gfc.inbuf_old[0].length);
    }
    cmpS(tag + 'decay', T.decay, gfc.decay);
    cmpS(tag + 'ATH.adjust0', 0.01, gfc.ATH.adjust);
    cmpS(tag + 'OldValue', 180, gfc.OldValue[0]);
    cmpS(tag + 'CurrentStep', 4, gfc.CurrentStep[0]);
    cmpS(tag + 'slot_lag', p.frac_SpF, gfc.slot_lag);
}
if (bad) { console.log('FAILED:', bad, 'mismatches of', nchk); process.exit(1); }
console.log('OK', nchk, 'checks');
