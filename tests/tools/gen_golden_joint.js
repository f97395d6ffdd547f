/*
 * TEST TOOL (needs /root/reference): goldens for the joint-stereo extension (SURVEY.md 8f #3).
 *
 * The reference's public Mp3Encoder hard-codes gfp.mode = STEREO (index.js:105), so its joint-stereo path -- M/S psychoacoustics,
 * the per-frame M/S decision, ms_convert, reduce_side -- is only reachable through its internal modules.  This generator drives
 * the UNMODIFIED reference modules wired exactly as index.js:73-111 wires them (tests/tools/ref_harness.js), with the single
 * difference gfp.mode = JOINT_STEREO, on
 *   - the reference's own fixtures (testdata/Left44100.wav / Right44100.wav, excerpt and full),
 *   - the synthetic corpora of pcm_gen.js, and their "centre" variants L = A + (B >> 3), R = A - (B >> 3) (strongly
 *     correlated channels, so that M/S frames occur; tests/pcm.py has the same transform).
 * Output: tests/golden/golden_joint.json (+ the small MP3s); per case the number of frames coded M/S is recorded so that the
 * tests can insist that the M/S path is exercised.   usage: node tests/tools/gen_golden_joint.js
 */
'use strict';
const fs = require('fs'), path = require('path'), crypto = require('crypto');
const { refEncoder, REF } = require('./ref_harness.js');
const gen = require('./pcm_gen.js');
const OUT = path.join(__dirname, '..', 'golden');
const md5 = (b) => crypto.createHash('md5').update(b).digest('hex');

function centre(A, B) {
    const L = new Int16Array(A.length), R = new Int16Array(A.length);
    for (let i = 0; i < A.length; i++) {
        const d = B[i] >> 3;
        L[i] = Math.max(-32768, Math.min(32767, A[i] + d));
        R[i] = Math.max(-32768, Math.min(32767, A[i] - d));
    }
    return [L, R];
}
function encode(L, R, kbps, chunk, sr) {
    const enc = refEncoder(2, sr || 44100, kbps, { jointStereo: true });
    const parts = [];
    for (let i = 0; i < L.length; i += chunk) {
        const b = enc.encodeBuffer(L.subarray(i, i + chunk), R.subarray(i, i + chunk));
        if (b.length) parts.push(Buffer.from(b.buffer, b.byteOffset, b.length));
    }
    const f = enc.flush();
    if (f.length) parts.push(Buffer.from(f.buffer, f.byteOffset, f.length));
    return Buffer.concat(parts);
}
/* walk the frames of a CBR stream: [frames, frames with mode_ext == 2 (M/S)], checking that every header says joint stereo */
function msFrames(mp3) {
    const BR1 = [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320], BR2 = [0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160];
    const SR = { 3: [44100, 48000, 32000], 2: [22050, 24000, 16000], 0: [11025, 12000, 8000] };
    let pos = 0, n = 0, ms = 0;
    while (pos + 4 <= mp3.length) {
        const h = mp3.readUInt32BE(pos);
        if ((h >>> 21) != 0x7ff) throw new Error('lost sync at ' + pos);
        const ver = (h >>> 19) & 3, bri = (h >>> 12) & 15, sri = (h >>> 10) & 3, pad = (h >>> 9) & 1, mode = (h >>> 6) & 3, ext = (h >>> 4) & 3;
        if (mode != 1) throw new Error('frame ' + n + ' is not joint stereo');
        const kb = (ver == 3 ? BR1 : BR2)[bri], sr = SR[ver][sri];
        const len = Math.floor((ver == 3 ? 144000 : 72000) * kb / sr) + pad;
        n++; if (ext == 2) ms++; else if (ext != 0) throw new Error('unexpected mode_ext ' + ext);
        pos += len;
    }
    return [n, ms];
}
function pcmMd5(L, R) { const h = crypto.createHash('md5'); h.update(Buffer.from(L.buffer, L.byteOffset, L.byteLength)); h.update(Buffer.from(R.buffer, R.byteOffset, R.byteLength)); return h.digest('hex'); }

const cases = [];
const WL = gen.readWav(fs.readFileSync(path.join(REF, 'testdata/Left44100.wav'))).samples;
const WR = gen.readWav(fs.readFileSync(path.join(REF, 'testdata/Right44100.wav'))).samples;
const NEX = 60 * 1152, NFULL = Math.floor(WL.length / 1152) * 1152;
function add(corpus, L, R, kbps, chunk, sr, extra) {
    const mp3 = encode(L, R, kbps, chunk, sr);
    const [nfr, nms] = msFrames(mp3);
    const c = Object.assign({ corpus, channels: 2, joint: 1, kbps, nsamples: L.length, chunk, pcm_md5: pcmMd5(L, R), mp3_md5: md5(mp3), mp3_len: mp3.length, frames: nfr, ms_frames: nms }, extra || {});
    if (sr) c.samplerate = sr;
    if (mp3.length < 30000) {
        c.mp3_file = `joint_${corpus}_${kbps}_${L.length / 1152 | 0}_${chunk}${sr ? '_' + sr : ''}.mp3`;
        fs.writeFileSync(path.join(OUT, c.mp3_file), mp3);
    }
    cases.push(c);
    console.log(corpus, kbps, sr || 44100, chunk, 'frames', nfr, 'M/S', nms, c.mp3_md5);
}
for (const kbps of [128, 320, 192]) {
    add('wavexcerpt', WL.subarray(0, NEX), WR.subarray(0, NEX), kbps, 1152);
    add('wavfull', WL.subarray(0, NFULL), WR.subarray(0, NFULL), kbps, 1152);
}
const synth = [
    ['sine', 128, 300, 1152], ['bursts', 128, 400, 1152], ['centre_sine', 128, 300, 1152], ['centre_bursts', 128, 400, 1152],
    ['centre_bursts', 320, 200, 1152], ['centre_sine', 160, 200, 777], ['centre_bursts', 96, 150, 1152, 48000], ['centre_sine', 256, 100, 4096],
    ['centre_bursts', 128, 2000, 1152 * 2000], ['centre_sine', 128, 1500, 1152 * 1500], ['bursts', 320, 1000, 1152 * 1000],
    ['centre_sine', 128, 1, 1152], ['centre_bursts', 128, 2, 100], ['centre_bursts', 192, 3, 1],
    ['centre_bursts', 128, 150, 1152, 32000], ['centre_sine', 224, 100, 999, 48000],
    /* MPEG-2 / 2.5: one granule per frame */
    ['centre_bursts', 64, 150, 1152, 22050], ['centre_sine', 96, 100, 777, 24000], ['centre_bursts', 32, 100, 1152, 16000],
    ['centre_bursts', 24, 100, 1152, 8000], ['centre_sine', 48, 100, 1000, 12000], ['centre_bursts', 64, 600, 1152 * 600, 22050],
    /* integer-ratio resampling in front */
    ['centre_bursts', 48, 150, 1152, 44100], ['centre_sine', 64, 100, 4096, 48000]
];
for (const [corpus, kbps, nframes, chunk, sr] of synth) {
    const n = nframes * 1152, base = corpus.replace('centre_', '');
    let [L, R] = gen[base](n, 2);
    if (corpus.startsWith('centre_')) [L, R] = centre(L, R);
    try { require('../../lamejs_amd/js/tables.js').buildBlob(2, sr || 44100, kbps, { jointStereo: true }); } catch (e) { console.log('skip (outside the envelope)', corpus, kbps, sr); continue; }
    add(corpus, L, R, kbps, chunk, sr);
}
fs.writeFileSync(path.join(OUT, 'golden_joint.json'), JSON.stringify({ generator: 'tests/tools/gen_golden_joint.js', reference: 'zhuker/lamejs v1.2.1 modules wired as index.js:73-111 with gfp.mode = JOINT_STEREO, under node ' + process.version, cases }, null, 1));
console.log('wrote', cases.length, 'cases;', cases.reduce((a, c) => a + c.ms_frames, 0), 'M/S frames of', cases.reduce((a, c) => a + c.frames, 0));
