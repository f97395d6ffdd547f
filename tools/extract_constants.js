#!/usr/bin/env node
/*
 * Build-time tool (runs only where /root/reference is mounted).
 *
 * Extracts the *numeric constant tables* of the MP3 standard / LAME tuning that the
 * hot path needs (ISO Huffman code books, scalefactor band edges, the polyphase
 * analysis window, MDCT windows, mask_add tables, ABR preset rows ...) out of the
 * reference's modules and writes them as plain JSON data to
 * lamejs_amd/js/constants.json.  No reference *code* is copied: arrays are either
 * read from exported module objects or evaluated from their literal initialisers.
 *
 * usage: node tools/extract_constants.js [/root/reference]
 */
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');

const REF = process.argv[2] || '/root/reference';
const SRC = path.join(REF, 'src', 'js');
const OUT = path.join(__dirname, '..', 'lamejs_amd', 'js', 'constants.json');

function src(name) { return fs.readFileSync(path.join(SRC, name), 'utf8'); }

/* evaluate the literal initialiser `var <name> = [ ... ];` found in a source text */
function literalArray(text, name, ctx) {
    const re = new RegExp('var\\s+' + name + '\\s*=\\s*\\[');
    const m = re.exec(text);
    if (!m) throw new Error('literal ' + name + ' not found');
    let i = m.index + m[0].length - 1, depth = 0, j = i;
    for (; j < text.length; j++) {
        if (text[j] === '[') depth++;
        else if (text[j] === ']') { depth--; if (depth === 0) break; }
    }
    const base = { Util: { SQRT2: 1.41421356237309504880 }, Math: Math };
    return vm.runInNewContext('(' + text.slice(i, j + 1) + ')', Object.assign(base, ctx || {}));
}

const out = {};

/* ---- Tables.js: exported object, read directly ---- */
const Tables = require(path.join(SRC, 'Tables.js'));
out.ht = Tables.ht.map(h => ({
    xlen: h.xlen, linmax: h.linmax,
    table: h.table ? Array.from(h.table) : null,
    hlen: h.hlen ? Array.from(h.hlen) : null
}));
out.largetbl = Array.from(Tables.largetbl);
out.table23 = Array.from(Tables.table23);
out.table56 = Array.from(Tables.table56);
out.bitrate_table = Tables.bitrate_table.map(r => Array.from(r));
out.samplerate_table = Tables.samplerate_table.map(r => Array.from(r));
out.scfsi_band = Array.from(Tables.scfsi_band);

/* ---- NewMDCT.js: closure-local literals ---- */
{
    const t = src('NewMDCT.js');
    out.enwindow = literalArray(t, 'enwindow');
    out.mdct_win = literalArray(t, 'win');
    out.mdct_order = literalArray(t, 'order');
}

/* ---- QuantizePVT.js: instance fields ---- */
{
    const QuantizePVT = require(path.join(SRC, 'QuantizePVT.js'));
    const q = new QuantizePVT();
    out.sfBandIndex = q.sfBandIndex.map(s => ({ l: Array.from(s.l), s: Array.from(s.s) }));
    out.pretab = Array.from(q.pretab);
}

/* ---- PsyModel.js: mask_add tables, HPF taps ---- */
{
    const t = src('PsyModel.js');
    out.ma_tab = literalArray(t, 'tab');
    out.ma_table1 = literalArray(t, 'table1');
    out.ma_table2 = literalArray(t, 'table2');
    out.ma_table3 = literalArray(t, 'table3');
    /* the 10-tap fs/4 high-pass (second `fircoef`, the one inside PsyModel) */
    out.hpf_fircoef = literalArray(t, 'fircoef');
}

/* ---- FFT.js ---- */
{
    const t = src('FFT.js');
    out.fht_costab = literalArray(t, 'costab');
    out.fft_rv_tbl = literalArray(t, 'rv_tbl');
}

/* ---- Takehiro.js small integer tables ---- */
{
    const t = src('Takehiro.js');
    out.subdv_table = literalArray(t, 'subdv_table');
    out.huf_tbl_noESC = literalArray(t, 'huf_tbl_noESC');
    out.slen1_n = literalArray(t, 'slen1_n');
    out.slen2_n = literalArray(t, 'slen2_n');
    out.slen1_tab = literalArray(t, 'slen1_tab');
    out.slen2_tab = literalArray(t, 'slen2_tab');
    out.scale_short = literalArray(t, 'scale_short');
    out.scale_mixed = literalArray(t, 'scale_mixed');
    out.scale_long = literalArray(t, 'scale_long');
}

/* ---- Presets.js: ABR switch map rows (numbers only) ---- */
{
    const t = src('Presets.js');
    function ABRPresets() { return Array.prototype.slice.call(arguments); }
    const rows = literalArray(t, 'abr_switch_map', { ABRPresets: ABRPresets });
    /* kbps quant q_s safejoint nsmsfix st_lrm st_s ns-bass scale msk ath_lwr ath_curve interch sfscale */
    out.abr_switch_map = rows.map(r => Array.from(r));
}

/* ---- Lame.js: bitrate -> lowpass map ---- */
{
    const t = src('Lame.js');
    function BandPass(kbps, lp) { return [kbps, lp]; }
    out.lowpass_freq_map = literalArray(t, 'freq_map', { BandPass: BandPass }).map(r => Array.from(r));
    out.full_bitrate_table = literalArray(t, 'full_bitrate_table');
}

/* ---- Version ---- */
{
    const Version = require(path.join(SRC, 'Version.js'));
    out.lame_short_version = new Version().getLameShortVersion();
}

fs.mkdirSync(path.dirname(OUT), { recursive: true });
fs.writeFileSync(OUT, JSON.stringify(out));
console.log('wrote', OUT, fs.statSync(OUT).size, 'bytes; keys:', Object.keys(out).join(','));
