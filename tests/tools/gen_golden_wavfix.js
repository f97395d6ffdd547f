/*
 * TEST TOOL (needs /root/reference): goldens from the reference's REMAINING fixtures (SURVEY.md 8f #2): testdata/Left.wav + Right.wav
 * (48 kHz) as mono / stereo streams at 128 and 320 kbps, and testdata/Stereo44100.wav de-interleaved -- encoded by the UNMODIFIED
 * reference (tests/tools/ref_harness.js refPublic) with the 1152-sample call pattern of Tests.js and as one large call.
 * The PCM is stored raw (s16le, a whole number of frames) so that the GPU box, which has no /root/reference, can read it.
 * usage: node tests/tools/gen_golden_wavfix.js  ->  tests/golden/golden_wavfix.json, left48000_full.s16, right48000_full.s16
 */
'use strict';
const fs = require('fs'), path = require('path'), crypto = require('crypto');
const { refPublic, REF } = require('./ref_harness.js');
const gen = require('./pcm_gen.js');
const OUT = path.join(__dirname, '..', 'golden');
const md5 = (b) => crypto.createHash('md5').update(b).digest('hex');
const lamejs = refPublic();
function encode(L, R, ch, kbps, chunk, sr) {
    const enc = new lamejs.Mp3Encoder(ch, sr, kbps);
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
const wl = gen.readWav(fs.readFileSync(path.join(REF, 'testdata/Left.wav'))), wr = gen.readWav(fs.readFileSync(path.join(REF, 'testdata/Right.wav')));
if (wl.fmt.rate != 48000 || wr.fmt.rate != 48000 || wl.fmt.channels != 1) throw new Error('unexpected fixture format');
const N48 = Math.floor(Math.min(wl.samples.length, wr.samples.length) / 1152) * 1152;
const L48 = wl.samples.subarray(0, N48), R48 = wr.samples.subarray(0, N48);
fs.writeFileSync(path.join(OUT, 'left48000_full.s16'), Buffer.from(L48.buffer, L48.byteOffset, N48 * 2));
fs.writeFileSync(path.join(OUT, 'right48000_full.s16'), Buffer.from(R48.buffer, R48.byteOffset, N48 * 2));
const st = gen.readWav(fs.readFileSync(path.join(REF, 'testdata/Stereo44100.wav')));
if (st.fmt.rate != 44100 || st.fmt.channels != 2) throw new Error('unexpected Stereo44100.wav format');
const NS = Math.floor(st.samples.length / 2 / 1152) * 1152;
const SL = new Int16Array(NS), SR = new Int16Array(NS);
for (let i = 0; i < NS; i++) { SL[i] = st.samples[2 * i]; SR[i] = st.samples[2 * i + 1]; }
/* Stereo44100.wav is Left44100.wav / Right44100.wav interleaved: the committed left44100_full.s16 / right44100_full.s16 then serve it too */
const L44 = new Int16Array(fs.readFileSync(path.join(OUT, 'left44100_full.s16')).buffer.slice(0)), R44 = new Int16Array(fs.readFileSync(path.join(OUT, 'right44100_full.s16')).buffer.slice(0));
let same = NS == L44.length;
for (let i = 0; same && i < NS; i++) if (SL[i] != L44[i] || SR[i] != R44[i]) same = false;
const cases = [];
for (const [ch, kbps, chunk] of [[1, 128, 1152], [2, 128, 1152], [2, 320, 1152], [1, 320, 1152], [2, 128, N48], [1, 128, 7777], [2, 192, 1152], [1, 96, 1152], [2, 64, 1152]]) {
    const R = ch == 2 ? R48 : null;
    const mp3 = encode(L48, R, ch, kbps, chunk, 48000);
    cases.push({ corpus: 'wav48000', fixture: 'testdata/Left.wav' + (ch == 2 ? ' + Right.wav' : ''), channels: ch, samplerate: 48000, kbps, nsamples: N48, chunk, pcm_md5: pcmMd5(L48, R), mp3_md5: md5(mp3), mp3_len: mp3.length });
    console.log('48k', ch, kbps, chunk, mp3.length, md5(mp3));
}
for (const [kbps, chunk] of [[128, 1152], [320, 1152], [128, NS], [192, 4096]]) {
    const mp3 = encode(SL, SR, 2, kbps, chunk, 44100);
    cases.push({ corpus: 'wavstereo44100', fixture: 'testdata/Stereo44100.wav (de-interleaved)', same_pcm_as_left_right_44100: same, channels: 2, samplerate: 44100, kbps, nsamples: NS, chunk, pcm_md5: pcmMd5(SL, SR), mp3_md5: md5(mp3), mp3_len: mp3.length });
    console.log('st44', kbps, chunk, mp3.length, md5(mp3));
}
if (!same) { fs.writeFileSync(path.join(OUT, 'stereo44100_left.s16'), Buffer.from(SL.buffer)); fs.writeFileSync(path.join(OUT, 'stereo44100_right.s16'), Buffer.from(SR.buffer)); }
fs.writeFileSync(path.join(OUT, 'golden_wavfix.json'), JSON.stringify({ generator: 'tests/tools/gen_golden_wavfix.js', reference: 'zhuker/lamejs v1.2.1 src/js/index.js under node ' + process.version, stereo44100_is_left_right_44100: same, cases }, null, 1));
console.log('wrote', cases.length, 'cases; Stereo44100.wav == Left44100 + Right44100:', same);
