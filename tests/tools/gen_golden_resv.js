/*
 * TEST TOOL (needs /root/reference): goldens for the bit-reservoir extension (SURVEY.md 8f #4).
 *
 * The reference's public Mp3Encoder sets gfp.disable_reservoir = true (index.js:108), so its reservoir path -- ResvFrameBegin / ResvMaxBits /
 * ResvFrameEnd, on_pe spending perceptual entropy, the reservoir-dependent pre-echo control of the psychoacoustic model, main_data_begin
 * and the continuous bitstream -- is only reachable through its internal modules.  This generator drives the UNMODIFIED reference
 * modules wired exactly as index.js:73-111 wires them (tests/tools/ref_harness.js) with the single difference
 * gfp.disable_reservoir = false (and, for the `joint` cases, gfp.mode = JOINT_STEREO as well: what the LAME defaults combine).
 * Output: tests/golden/golden_resv.json (+ the small MP3s).   usage: node tests/tools/gen_golden_resv.js
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
function encode(L, R, ch, kbps, chunk, sr, joint) {
    const enc = refEncoder(ch, sr || 44100, kbps, { jointStereo: !!joint, reservoir: true });
    const parts = [];
    for (let i = 0; i < L.length; i += chunk) {
        const b = ch == 2 ? enc.encodeBuffer(L.subarray(i, i + chunk), R.subarray(i, i + chunk)) : enc.encodeBuffer(L.subarray(i, i + chunk));
        if (b.length) parts.push(Buffer.from(b.buffer, b.byteOffset, b.length));
    }
    const f = enc.flush();
    if (f.length) parts.push(Buffer.from(f.buffer, f.byteOffset, f.length));
    return Buffer.concat(parts);
}
/* walk the frames: [frames, largest main_data_begin]; the stream must be a whole number of frames */
function walk(mp3, ch) {
    const BR1 = [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320], BR2 = [0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160];
    const SR = { 3: [44100, 48000, 32000], 2: [22050, 24000, 16000], 0: [11025, 12000, 8000] };
    let pos = 0, n = 0, mdbmax = 0;
    while (pos + 6 <= mp3.length) {
        const h = mp3.readUInt32BE(pos);
        if ((h >>> 21) != 0x7ff) throw new Error('lost sync at ' + pos);
        const ver = (h >>> 19) & 3, bri = (h >>> 12) & 15, sri = (h >>> 10) & 3, pad = (h >>> 9) & 1;
        const mdb = ver == 3 ? ((mp3[pos + 4] << 1) | (mp3[pos + 5] >> 7)) : mp3[pos + 4];
        if (mdb > mdbmax) mdbmax = mdb;
        pos += Math.floor((ver == 3 ? 144000 : 72000) * (ver == 3 ? BR1 : BR2)[bri] / SR[ver][sri]) + pad;
        n++;
    }
    if (pos != mp3.length) throw new Error('stream does not end on a frame boundary');
    return [n, mdbmax];
}
function pcmMd5(L, R) { const h = crypto.createHash('md5'); h.update(Buffer.from(L.buffer, L.byteOffset, L.byteLength)); if (R) h.update(Buffer.from(R.buffer, R.byteOffset, R.byteLength)); return h.digest('hex'); }

const cases = [];
const WL = gen.readWav(fs.readFileSync(path.join(REF, 'testdata/Left44100.wav'))).samples;
const WR = gen.readWav(fs.readFileSync(path.join(REF, 'testdata/Right44100.wav'))).samples;
const NEX = 60 * 1152, NFULL = Math.floor(WL.length / 1152) * 1152;
function add(corpus, L, R, ch, kbps, chunk, sr, joint) {
    const mp3 = encode(L, R, ch, kbps, chunk, sr, joint);
    const [nfr, mdb] = walk(mp3, ch);
    const c = { corpus, channels: ch, reservoir: 1, kbps, nsamples: L.length, chunk, pcm_md5: pcmMd5(L, R), mp3_md5: md5(mp3), mp3_len: mp3.length, frames: nfr, max_main_data_begin: mdb };
    if (joint) c.joint = 1;
    if (sr) c.samplerate = sr;
    if (mp3.length < 30000) {
        c.mp3_file = `resv_${joint ? 'joint_' : ''}${corpus}_${ch}_${kbps}_${L.length / 1152 | 0}_${chunk}${sr ? '_' + sr : ''}.mp3`;
        fs.writeFileSync(path.join(OUT, c.mp3_file), mp3);
    }
    cases.push(c);
    console.log(corpus, ch, kbps, sr || 44100, chunk, joint ? 'joint' : '', 'frames', nfr, 'max main_data_begin', mdb, c.mp3_md5);
}
for (const [ch, kbps, joint] of [[1, 128, 0], [2, 128, 0], [2, 320, 0], [2, 128, 1]]) {
    add('wavexcerpt', WL.subarray(0, NEX), ch == 2 ? WR.subarray(0, NEX) : null, ch, kbps, 1152, undefined, joint);
    add('wavfull', WL.subarray(0, NFULL), ch == 2 ? WR.subarray(0, NFULL) : null, ch, kbps, 1152, undefined, joint);
}
const synth = [
    ['sine', 1, 128, 300, 1152], ['sine', 2, 128, 300, 1152], ['bursts', 1, 128, 400, 1152], ['bursts', 2, 128, 400, 1152], ['bursts', 2, 320, 200, 1152],
    ['bursts', 2, 128, 250, 777], ['sine', 1, 128, 250, 4096], ['bursts', 1, 64, 200, 1152], ['sine', 2, 192, 200, 1152],
    ['bursts', 2, 128, 2000, 1152 * 2000], ['sine', 1, 128, 2000, 1152 * 2000], ['sine', 2, 320, 1000, 1152 * 1000],
    ['sine', 1, 128, 1, 1152], ['sine', 2, 128, 2, 100], ['bursts', 1, 128, 3, 1],
    ['bursts', 2, 192, 150, 1152, 48000], ['sine', 1, 64, 150, 1152, 32000], ['bursts', 2, 160, 100, 999, 48000], ['sine', 1, 32, 100, 1152],
    /* MPEG-2 / 2.5 */
    ['bursts', 1, 64, 120, 1152, 22050], ['bursts', 2, 64, 120, 1152, 22050], ['sine', 2, 96, 100, 777, 24000], ['bursts', 1, 32, 100, 1152, 16000],
    ['bursts', 2, 24, 100, 1152, 8000], ['sine', 1, 40, 100, 333, 12000], ['bursts', 2, 64, 600, 1152 * 600, 22050],
    /* resampling in front */
    ['bursts', 2, 48, 150, 1152, 44100], ['bursts', 1, 24, 100, 777, 48000],
    /* joint stereo + reservoir: the combination LAME itself defaults to */
    ['bursts', 2, 128, 400, 1152, undefined, 1], ['centre_sine', 2, 128, 300, 1152, undefined, 1], ['centre_bursts', 2, 192, 150, 4096, undefined, 1],
    ['centre_bursts', 2, 64, 150, 1152, 22050, 1], ['centre_bursts', 2, 128, 1500, 1152 * 1500, undefined, 1]
];
for (const [corpus, ch, kbps, nframes, chunk, sr, joint] of synth) {
    const n = nframes * 1152, base = corpus.replace('centre_', '');
    let [L, R] = gen[base](n, ch);
    if (corpus.startsWith('centre_')) [L, R] = centre(L, R);
    try { require('../../lamejs_amd/js/tables.js').buildBlob(ch, sr || 44100, kbps, { jointStereo: !!joint, reservoir: true }); } catch (e) { console.log('skip (outside the envelope)', corpus, kbps, sr); continue; }
    add(corpus, L, R, ch, kbps, chunk, sr, joint);
}
fs.writeFileSync(path.join(OUT, 'golden_resv.json'), JSON.stringify({ generator: 'tests/tools/gen_golden_resv.js', reference: 'zhuker/lamejs v1.2.1 modules wired as index.js:73-111 with gfp.disable_reservoir = false (joint cases: gfp.mode = JOINT_STEREO too), under node ' + process.version, cases }, null, 1));
console.log('wrote', cases.length, 'cases');
