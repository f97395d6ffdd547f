/* TEST TOOL: drives the JavaScript drop-in (lamejs_amd/js) exactly like the reference's Tests.js drives
 * lamejs (1152-sample subarrays + flush) and prints the MD5 of the produced MP3.
 * usage: node tests/js_dropin_check.js <corpus> <channels> <kbps> <nframes> [chunk] [samplerate] [joint] [reservoir]
 * corpus centre_<x>: L = A + (B >> 3), R = A - (B >> 3) of corpus x (tests/tools/gen_golden_joint.js); joint: the jointStereo extension */
'use strict';
const path = require('path'), crypto = require('crypto');
const lamejs = require(path.join(__dirname, '..', 'lamejs_amd', 'js'));
const gen = require('./tools/pcm_gen.js');
const [corpus, chS, kbS, nfS, chunkS, srS] = process.argv.slice(2), flags = process.argv.slice(8);
const jointS = flags.includes('joint') ? 'joint' : '', resvS = flags.includes('reservoir');
const ch = +chS, kbps = +kbS, n = +nfS * 1152, chunk = +(chunkS || 1152);
let [L, R] = gen[corpus.replace('centre_', '')](n, ch);
if (corpus.startsWith('centre_')) {
    const A = L, B = R;
    L = new Int16Array(n); R = new Int16Array(n);
    for (let i = 0; i < n; i++) { const d = B[i] >> 3; L[i] = Math.max(-32768, Math.min(32767, A[i] + d)); R[i] = Math.max(-32768, Math.min(32767, A[i] - d)); }
}
const enc = (jointS === 'joint' || resvS) ? new lamejs.Mp3Encoder(ch, +(srS || 44100), kbps, { jointStereo: jointS === 'joint', reservoir: resvS }) : new lamejs.Mp3Encoder(ch, +(srS || 44100), kbps);
const parts = [];
for (let i = 0; i < n; i += chunk) {
    const b = ch == 2 ? enc.encodeBuffer(L.subarray(i, i + chunk), R.subarray(i, i + chunk)) : enc.encodeBuffer(L.subarray(i, i + chunk));
    if (!(b instanceof Int8Array)) throw new Error('encodeBuffer must return an Int8Array');
    parts.push(Buffer.from(b.buffer, b.byteOffset, b.length));
}
const f = enc.flush();
parts.push(Buffer.from(f.buffer, f.byteOffset, f.length));
if (enc.flush().length !== 0) throw new Error('second flush must be empty');
const all = Buffer.concat(parts);
console.log(JSON.stringify({ bytes: all.length, md5: crypto.createHash('md5').update(all).digest('hex') }));
