/* TEST TOOL: drives the JavaScript drop-in (lamejs_amd/js) exactly like the reference's Tests.js drives
 * lamejs (1152-sample subarrays + flush) and prints the MD5 of the produced MP3.
 * usage: node tests/js_dropin_check.js <corpus> <channels> <kbps> <nframes> [chunk] [samplerate] */
'use strict';
const path = require('path'), crypto = require('crypto');
const lamejs = require(path.join(__dirname, '..', 'lamejs_amd', 'js'));
const gen = require('./tools/pcm_gen.js');
const [corpus, chS, kbS, nfS, chunkS, srS] = process.argv.slice(2);
const ch = +chS, kbps = +kbS, n = +nfS * 1152, chunk = +(chunkS || 1152);
const [L, R] = gen[corpus](n, ch);
const enc = new lamejs.Mp3Encoder(ch, +(srS || 44100), kbps);
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
