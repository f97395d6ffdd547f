/*
 * TEST: the { pendingFrames: N } extension of lamejs_amd/js (input held back until N frames' worth is pending, then ONE launch): fed 1152 samples per
 * call it must return the same byte STREAM as the encoder without it (and hence as the reference), with far fewer non-empty returns.
 * usage: node js_pending_check.js <corpus> <channels> <kbps> <frames>    -> one JSON line
 */
'use strict';
const path = require('path'), crypto = require('crypto');
const gen = require('./tools/pcm_gen.js');
const lamejs = require(path.join(__dirname, '..', 'lamejs_amd', 'js', 'index.js'));
const [corpus, ch, kbps, nfr] = [process.argv[2] || 'sine', +(process.argv[3] || 2), +(process.argv[4] || 128), +(process.argv[5] || 150)];
const [L, R] = gen[corpus](1152 * nfr + 77, ch, 4242);
function run(opts, chunk) {
    const e = opts ? new lamejs.Mp3Encoder(ch, 44100, kbps, opts) : new lamejs.Mp3Encoder(ch, 44100, kbps);
    const h = crypto.createHash('md5'); let n = 0, nonempty = 0;
    for (let i = 0; i < L.length; i += chunk) {
        const b = ch == 2 ? e.encodeBuffer(L.subarray(i, i + chunk), R.subarray(i, i + chunk)) : e.encodeBuffer(L.subarray(i, i + chunk));
        if (b.length) nonempty++;
        h.update(Buffer.from(b.buffer, b.byteOffset, b.length)); n += b.length;
    }
    const f = e.flush();
    h.update(Buffer.from(f.buffer, f.byteOffset, f.length));
    const again = e.flush().length;
    return { md5: h.digest('hex'), bytes: n + f.length, nonempty, second_flush_bytes: again };
}
console.log(JSON.stringify({ plain: run(null, 1152), pending64: run({ pendingFrames: 64 }, 1152), pending7: run({ pendingFrames: 7 }, 1152), pending5_odd_chunks: run({ pendingFrames: 5 }, 3001),
    pending3_big_chunks: run({ pendingFrames: 3 }, 10000) }));
