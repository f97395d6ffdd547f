/*
 * TEST TOOL (needs /root/reference): md5 + length of the UNMODIFIED reference encoder's output on the full-size BASELINE
 * workloads (SURVEY.md 8d configs 2-5), i.e. what bench.py's streams must reproduce byte for byte on the GPU:
 *   one Mp3Encoder per stream, the whole PCM fed through encodeBuffer (no flush: bench.py does not flush either), md5 of the
 *   concatenated output.  PCM comes from the same generators as tests/tools/pcm_gen.js / tests/pcm.py, produced in chunks.
 * Every job prints one JSON line; tests/tools/gen_full_md5.sh runs the job list in parallel and merges the lines into
 * tests/golden/full_md5.json.
 *
 *   node tests/tools/gen_full_md5.js <corpus> <channels> <kbps> <frames> <seed0> [nseeds] [joint] [reservoir]
 * corpus centre_<x>: L = A + (B >> 3), R = A - (B >> 3) of corpus x; joint: the reference's modules driven with gfp.mode = JOINT_STEREO
 * (ref_harness.js refEncoder) -- the joint-stereo extension, SURVEY.md 8f #3.
 */
'use strict';
const crypto = require('crypto');
const { refPublic, refEncoder } = require('./ref_harness.js');
const gen = require('./pcm_gen.js');
const lamejs = refPublic();

const corpus = process.argv[2], ch = parseInt(process.argv[3]), kbps = parseInt(process.argv[4]), frames = parseInt(process.argv[5]);
const seed0 = parseInt(process.argv[6]), nseeds = parseInt(process.argv[7] || '1'), flags = process.argv.slice(8);
const joint = flags.includes('joint'), resv = flags.includes('reservoir');      /* reservoir: gfp.disable_reservoir = false (SURVEY.md 8f #4) */
const base = corpus.replace('centre_', ''), centre = corpus.startsWith('centre_');
const CHUNK = 1152 * 500;

/* chunked twins of pcm_gen.sine / pcm_gen.bursts: same expressions, the sample index runs over the whole stream */
function makeSource(corpus, ch, seed) {
    const u = gen.lcg(seed);
    let i = 0;
    return function next(n) {
        const L = new Int16Array(n), R = ch == 2 ? new Int16Array(n) : null;
        for (let k = 0; k < n; k++, i++) {
            if (base == 'sine') {
                L[k] = Math.round(8000 * Math.sin(2 * Math.PI * 440 * i / 44100) + 2000 * (2 * u() - 1));
                if (R) R[k] = Math.round(6000 * Math.sin(2 * Math.PI * 660 * i / 44100) + 2000 * (2 * u() - 1));
            } else {
                const inBurst = (i % 22050) >= 11000 && (i % 22050) < 13000;
                L[k] = Math.round((inBurst ? 20000 : 30) * (2 * u() - 1));
                if (R) { const inBurstR = ((i + 5000) % 22050) >= 11000 && ((i + 5000) % 22050) < 13000; R[k] = Math.round((inBurstR ? 20000 : 30) * (2 * u() - 1)); }
            }
        }
        if (centre) for (let k = 0; k < n; k++) { const a = L[k], d = R[k] >> 3; L[k] = Math.max(-32768, Math.min(32767, a + d)); R[k] = Math.max(-32768, Math.min(32767, a - d)); }
        return [L, R];
    };
}

for (let s = seed0; s < seed0 + nseeds; s++) {
    const src = makeSource(corpus, ch, s), enc = (joint || resv) ? refEncoder(ch, 44100, kbps, { jointStereo: joint, reservoir: resv }) : new lamejs.Mp3Encoder(ch, 44100, kbps), h = crypto.createHash('md5');
    let bytes = 0;
    for (let left = 1152 * frames; left > 0;) {
        const n = Math.min(left, CHUNK);
        const [L, R] = src(n);
        const out = ch == 2 ? enc.encodeBuffer(L, R) : enc.encodeBuffer(L);
        h.update(Buffer.from(out.buffer, out.byteOffset, out.length));
        bytes += out.length; left -= n;
    }
    const row = { corpus, channels: ch, samplerate: 44100, kbps, frames, seed: s, flush: false, bytes, md5: h.digest('hex') };
    if (joint) row.joint = 1;
    if (resv) row.reservoir = 1;
    console.log(JSON.stringify(row));
}
