/*
 * TEST TOOL (needs /root/reference): generates tests/golden/* by running the UNMODIFIED reference
 * encoder (src/js/index.js via tests/tools/ref_harness.js refPublic) on
 *   - the reference's own fixtures testdata/Left44100.wav / Right44100.wav (the 287 x 1152 samples Tests.js encodes, stored as
 *     raw s16le so that the GPU box, which has no /root/reference, can read them; the 60-frame excerpt cases are a prefix), and
 *   - the synthetic corpora of tests/tools/pcm_gen.js (regenerated bit-identically by tests/pcm.py).
 * Outputs: golden.json (per case: parameters, md5 of PCM, md5 + length of the MP3, and for the small
 * cases the MP3 bytes themselves as files) .  usage: node tests/tools/gen_golden.js
 */
'use strict';
const fs = require('fs'), path = require('path'), crypto = require('crypto');
const { refPublic, REF } = require('./ref_harness.js');
const gen = require('./pcm_gen.js');
const OUT = path.join(__dirname, '..', 'golden');
fs.mkdirSync(OUT, { recursive: true });
const md5 = (b) => crypto.createHash('md5').update(b).digest('hex');
const lamejs = refPublic();

function encode(L, R, ch, kbps, chunk, sr) {
    const enc = new lamejs.Mp3Encoder(ch, sr || 44100, kbps);
    const parts = [];
    for (let i = 0; i < L.length; i += chunk) {
        const l = L.subarray(i, i + chunk), r = R ? R.subarray(i, i + chunk) : undefined;
        const b = ch == 2 ? enc.encodeBuffer(l, r) : enc.encodeBuffer(l);
        if (b.length) parts.push(Buffer.from(b.buffer, b.byteOffset, b.length));
    }
    const f = enc.flush();
    if (f.length) parts.push(Buffer.from(f.buffer, f.byteOffset, f.length));
    return Buffer.concat(parts);
}
function pcmMd5(L, R) { const h = crypto.createHash('md5'); h.update(Buffer.from(L.buffer, L.byteOffset, L.byteLength)); if (R) h.update(Buffer.from(R.buffer, R.byteOffset, R.byteLength)); return h.digest('hex'); }

const cases = [];
/* 1. reference fixtures (excerpt committed; full files hashed for the container-only test) */
const WL = gen.readWav(fs.readFileSync(path.join(REF, 'testdata/Left44100.wav'))).samples;
const WR = gen.readWav(fs.readFileSync(path.join(REF, 'testdata/Right44100.wav'))).samples;
const NEX = 60 * 1152;
const NFULL = Math.floor(WL.length / 1152) * 1152;
fs.writeFileSync(path.join(OUT, 'left44100_full.s16'), Buffer.from(WL.buffer, WL.byteOffset, NFULL * 2));
fs.writeFileSync(path.join(OUT, 'right44100_full.s16'), Buffer.from(WR.buffer, WR.byteOffset, NFULL * 2));
for (const [ch, kbps] of [[1, 128], [2, 128], [2, 320]]) {
    const L = WL.subarray(0, NEX), R = ch == 2 ? WR.subarray(0, NEX) : null;
    const mp3 = encode(L, R, ch, kbps, 1152);
    const name = `wavexcerpt_${ch}_${kbps}.mp3`;
    fs.writeFileSync(path.join(OUT, name), mp3);
    cases.push({ corpus: 'wavexcerpt', channels: ch, kbps, nsamples: NEX, chunk: 1152, pcm_md5: pcmMd5(L, R), mp3_md5: md5(mp3), mp3_len: mp3.length, mp3_file: name });
    const LF = WL.subarray(0, NFULL), RF = ch == 2 ? WR.subarray(0, NFULL) : null;
    const full = encode(LF, RF, ch, kbps, 1152);
    cases.push({ corpus: 'wavfull', channels: ch, kbps, nsamples: NFULL, chunk: 1152, pcm_md5: pcmMd5(LF, RF), mp3_md5: md5(full), mp3_len: full.length });
}
/* 2. synthetic corpora */
const synth = [
    ['sine', 1, 128, 300, 1152], ['sine', 2, 128, 300, 1152], ['sine', 2, 320, 300, 1152],
    ['bursts', 1, 128, 400, 1152], ['bursts', 2, 128, 400, 1152], ['bursts', 2, 320, 400, 1152],
    ['bursts', 2, 128, 250, 777], ['sine', 1, 128, 250, 4096], ['bursts', 1, 64, 200, 1152], ['sine', 2, 192, 200, 1152],
    ['sine', 1, 128, 2000, 1152 * 2000], ['bursts', 2, 128, 2000, 1152 * 2000], ['sine', 2, 320, 1000, 1152 * 1000],
    ['sine', 1, 128, 1, 1152], ['sine', 2, 128, 2, 100], ['bursts', 1, 128, 3, 1],
    /* the other MPEG-1 sample rates and bitrates of the envelope */
    ['sine', 2, 128, 150, 1152, 48000], ['bursts', 1, 128, 150, 1152, 48000], ['bursts', 2, 192, 150, 1152, 48000], ['bursts', 2, 224, 120, 999, 48000],
    ['sine', 1, 64, 150, 1152, 32000], ['bursts', 2, 128, 150, 1152, 32000], ['sine', 1, 320, 100, 1152, 32000],
    ['sine', 2, 256, 100, 1152], ['bursts', 2, 160, 100, 1152], ['sine', 1, 96, 100, 1152], ['bursts', 1, 320, 100, 1152],
    ['sine', 1, 32, 100, 1152], ['bursts', 2, 64, 100, 1152], ['bursts', 2, 96, 100, 1152], ['sine', 1, 48, 60, 1152], ['bursts', 1, 160, 60, 500],
    /* MPEG-2 (22.05 / 24 / 16 kHz) and MPEG-2.5 (11.025 / 12 / 8 kHz): one 576-sample granule per frame */
    ['bursts', 1, 64, 120, 1152, 22050], ['bursts', 2, 64, 120, 1152, 22050], ['sine', 2, 96, 100, 777, 22050], ['bursts', 2, 160, 80, 1152, 22050],
    ['bursts', 2, 128, 100, 1152, 24000], ['sine', 1, 80, 100, 576, 24000], ['bursts', 1, 32, 100, 1152, 16000], ['bursts', 2, 64, 100, 1000, 16000],
    ['sine', 2, 48, 100, 1152, 16000], ['bursts', 1, 24, 100, 1152, 11025], ['bursts', 2, 64, 100, 1152, 11025], ['sine', 1, 40, 100, 333, 12000],
    ['bursts', 2, 48, 100, 1152, 12000], ['bursts', 1, 8, 100, 1152, 8000], ['bursts', 2, 24, 100, 1152, 8000], ['sine', 1, 64, 100, 4096, 8000],
    ['bursts', 1, 64, 1000, 1152 * 1000, 22050], ['bursts', 2, 32, 600, 1152 * 600, 16000],
    ['sine', 1, 64, 1, 1152, 22050], ['bursts', 2, 32, 2, 1, 16000], ['sine', 1, 16, 1, 575, 8000],
    /* resampling by an integer ratio (fill_buffer_resample, Lame.js:1719-1843): 44.1->22.05, 48->24/16/8, 32->16/8, 24->8, 16->8 kHz */
    ['bursts', 2, 48, 150, 1152, 44100], ['bursts', 1, 24, 100, 777, 48000], ['bursts', 2, 64, 100, 4096, 48000], ['sine', 1, 16, 100, 100, 32000],
    ['bursts', 2, 8, 100, 1152, 32000], ['bursts', 1, 8, 100, 1152, 16000], ['sine', 2, 16, 100, 5000, 24000], ['bursts', 1, 40, 100, 1152, 48000],
    ['bursts', 1, 8, 60, 33, 48000], ['bursts', 2, 24, 3, 1, 48000], ['sine', 1, 32, 600, 1152 * 600, 44100], ['bursts', 2, 40, 1, 17, 32000]
];
for (const [corpus, ch, kbps, nframes, chunk, sr] of synth) {
    const n = nframes * 1152;
    const [L, R] = gen[corpus](n, ch);
    const mp3 = encode(L, R, ch, kbps, chunk, sr);
    const c = { corpus, channels: ch, kbps, nsamples: n, chunk, pcm_md5: pcmMd5(L, R), mp3_md5: md5(mp3), mp3_len: mp3.length };
    if (sr) c.samplerate = sr;
    try { require('../../lamejs_amd/js/tables.js').buildBlob(ch, sr || 44100, kbps); } catch (e) { c.outside_envelope = String(e.message); }   /* the reference would resample by a non-integer ratio (NaN samples there): DESIGN.md 0 */
    if (mp3.length < 30000) { c.mp3_file = `${corpus}_${ch}_${kbps}_${nframes}_${chunk}${sr ? '_' + sr : ''}.mp3`; fs.writeFileSync(path.join(OUT, c.mp3_file), mp3); }
    cases.push(c);
    console.log(corpus, ch, kbps, nframes, chunk, mp3.length, c.mp3_md5);
}
fs.writeFileSync(path.join(OUT, 'golden.json'), JSON.stringify({ generator: 'tests/tools/gen_golden.js', reference: 'zhuker/lamejs v1.2.1 src/js/index.js under node ' + process.version, cases }, null, 1));
console.log('wrote', cases.length, 'cases');
