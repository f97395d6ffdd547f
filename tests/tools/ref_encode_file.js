/*
 * TEST TOOL (needs /root/reference): encode an interleaved s16le PCM file with the UNMODIFIED reference encoder.
 * usage: node tests/tools/ref_encode_file.js in.pcm out.mp3 <channels> <samplerate> <kbps> [chunk] [joint] [reservoir]
 * `reservoir`: the same wiring with gfp.disable_reservoir = false (index.js:108 sets it true).
 * `joint`: the reference's modules wired as index.js does, with gfp.mode = JOINT_STEREO (ref_harness.js refEncoder); prints the
 * number of frames coded M/S.
 */
'use strict';
const fs = require('fs');
const { refPublic, refEncoder } = require('./ref_harness.js');
const [inF, outF, chS, srS, kbS, chunkS] = process.argv.slice(2), flags = process.argv.slice(8);
const jointS = flags.includes('joint') ? 'joint' : '', resv = flags.includes('reservoir');
const ch = +chS, sr = +srS, kbps = +kbS, chunk = +(chunkS || 1152);
const raw = fs.readFileSync(inF);
const inter = new Int16Array(raw.buffer, raw.byteOffset, raw.length >> 1);
const n = Math.floor(inter.length / ch);
const L = new Int16Array(n), R = ch == 2 ? new Int16Array(n) : null;
for (let i = 0; i < n; i++) { L[i] = inter[i * ch]; if (R) R[i] = inter[i * ch + 1]; }
const enc = (jointS === 'joint' || resv) ? refEncoder(ch, sr, kbps, { jointStereo: jointS === 'joint', reservoir: resv }) : new (refPublic().Mp3Encoder)(ch, sr, kbps);
const parts = [];
for (let i = 0; i < n; i += chunk) {
    const b = ch == 2 ? enc.encodeBuffer(L.subarray(i, i + chunk), R.subarray(i, i + chunk)) : enc.encodeBuffer(L.subarray(i, i + chunk));
    if (b.length) parts.push(Buffer.from(b.buffer, b.byteOffset, b.length));
}
const f = enc.flush();
if (f.length) parts.push(Buffer.from(f.buffer, f.byteOffset, f.length));
fs.writeFileSync(outF, Buffer.concat(parts));
