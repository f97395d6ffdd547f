/* TEST TOOL: the JavaScript batch extension (encodeBatch / flushBatch) must give every stream the bytes its own
 * encodeBuffer()/flush() sequence gives.  N independent streams (seed 1000 + s, BASELINE config 5 shape), fed in
 * `chunk`-sample calls; prints per-stream MD5s of both ways.
 * usage: node tests/js_batch_check.js <channels> <kbps> <nstreams> <nframes> <chunk> */
'use strict';
const path = require('path'), crypto = require('crypto');
const lamejs = require(path.join(__dirname, '..', 'lamejs_amd', 'js'));
const gen = require('./tools/pcm_gen.js');
const [chS, kbS, nsS, nfS, chunkS] = process.argv.slice(2);
const ch = +chS, kbps = +kbS, NS = +nsS, n = +nfS * 1152, chunk = +chunkS;
const pcm = [];
for (let s = 0; s < NS; s++) pcm.push(gen.sine(n - 37 * s, ch, 1000 + s));          // ragged lengths
const md5 = (parts) => crypto.createHash('md5').update(Buffer.concat(parts)).digest('hex');
const buf = (b) => { if (!(b instanceof Int8Array)) throw new Error('Int8Array expected'); return Buffer.from(b.buffer, b.byteOffset, b.length); };
/* one by one */
const single = [];
for (let s = 0; s < NS; s++) {
    const enc = new lamejs.Mp3Encoder(ch, 44100, kbps), parts = [], [L, R] = pcm[s];
    for (let i = 0; i < L.length; i += chunk) parts.push(buf(ch == 2 ? enc.encodeBuffer(L.subarray(i, i + chunk), R.subarray(i, i + chunk)) : enc.encodeBuffer(L.subarray(i, i + chunk))));
    parts.push(buf(enc.flush()));
    single.push(md5(parts));
}
/* batched */
const encs = [], parts = [];
for (let s = 0; s < NS; s++) { encs.push(new lamejs.Mp3Encoder(ch, 44100, kbps)); parts.push([]); }
for (let i = 0; i < n; i += chunk) {
    const out = lamejs.encodeBatch(encs, pcm.map((p) => p[0].subarray(Math.min(i, p[0].length), i + chunk)), ch == 2 ? pcm.map((p) => p[1].subarray(Math.min(i, p[1].length), i + chunk)) : null);
    if (out.length != NS) throw new Error('encodeBatch must return one array per encoder');
    out.forEach((b, s) => parts[s].push(buf(b)));
}
lamejs.flushBatch(encs).forEach((b, s) => parts[s].push(buf(b)));
if (lamejs.flushBatch(encs).some((b) => b.length !== 0)) throw new Error('second flushBatch must be empty');
console.log(JSON.stringify({ single: single, batch: parts.map(md5) }));
