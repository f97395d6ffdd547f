/*
 * FIXTURE GENERATOR (needs the reference: /root/reference or oracle/_ref/lame.all.js): md5 + length of what the UNMODIFIED reference
 * returns for the bench's 1152-sample call pattern -- one Mp3Encoder, 1152 samples per encodeBuffer() call, flush() -- on the first
 * <frames> frames of the `sine` stream (seed 12345), mono and stereo 128 kbps.  Output: tests/golden/calls_md5.json
 * (bench.py's dropin_node_1152* lines check against it).   node tests/tools/gen_calls_md5.js [frames]
 */
'use strict';
const fs = require('fs'), path = require('path'), crypto = require('crypto');
const gen = require('./pcm_gen.js');
const { refPublic } = require('./ref_harness.js');
const frames = +(process.argv[2] || 2000);
const out = { generator: 'tests/tools/gen_calls_md5.js (unmodified reference under node: 1152-sample encodeBuffer calls + flush)', entries: [] };
for (const ch of [1, 2]) {
    const [L, R] = gen.sine(1152 * frames, ch, 12345);
    const enc = new (refPublic().Mp3Encoder)(ch, 44100, 128);
    const h = crypto.createHash('md5'); let n = 0;
    for (let i = 0; i < L.length; i += 1152) {
        const b = ch == 2 ? enc.encodeBuffer(L.subarray(i, i + 1152), R.subarray(i, i + 1152)) : enc.encodeBuffer(L.subarray(i, i + 1152));
        h.update(Buffer.from(b.buffer, b.byteOffset, b.length)); n += b.length;
    }
    const f = enc.flush();
    h.update(Buffer.from(f.buffer, f.byteOffset, f.length)); n += f.length;
    out.entries.push({ corpus: 'sine', channels: ch, samplerate: 44100, kbps: 128, frames, seed: 12345, chunk: 1152, flush: true, bytes: n, md5: h.digest('hex') });
}
fs.writeFileSync(path.join(__dirname, '..', 'golden', 'calls_md5.json'), JSON.stringify(out, null, 1) + '\n');
console.log(JSON.stringify(out));
