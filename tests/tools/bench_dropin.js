/*
 * BENCH TOOL (GPU): the drop-in's own throughput through the JavaScript surface -- one Mp3Encoder.encodeBuffer(left, right) call with
 * host Int16Arrays of N frames (lamejs_amd/js -> N-API addon -> lhip_encode, which cuts the call into chunks and overlaps their PCIe
 * copies with the encode of the chunk before), then flush().  Prints one JSON line: frames/s of the encodeBuffer call (wall clock
 * around it: H2D + encode + D2H + the copy into the returned Int8Array), md5 and length of encodeBuffer + flush, and of the encodeBuffer output alone (what tests/golden/full_md5.json holds).
 * usage: node tests/tools/bench_dropin.js <corpus> <channels> <kbps> <frames> <seed>
 */
'use strict';
const path = require('path'), crypto = require('crypto');
const gen = require('./pcm_gen.js');
const lamejs = require(path.join(__dirname, '..', '..', 'lamejs_amd', 'js', 'index.js'));
const [corpus, ch, kbps, nfr, seed] = [process.argv[2] || 'sine', +(process.argv[3] || 2), +(process.argv[4] || 128), +(process.argv[5] || 100000), +(process.argv[6] || 12345)];
const [L, R] = gen[corpus](1152 * nfr, ch, seed);
{   /* warm-up: library, tables, kernels, staging buffers of the chunked path */
    const w = new lamejs.Mp3Encoder(ch, 44100, kbps);
    const m = Math.min(L.length, 1152 * 20000);
    w.encodeBuffer(L.subarray(0, m), R ? R.subarray(0, m) : undefined); w.flush();
}
const enc = new lamejs.Mp3Encoder(ch, 44100, kbps);
const t0 = process.hrtime.bigint();
const a = ch == 2 ? enc.encodeBuffer(L, R) : enc.encodeBuffer(L);
const dt = Number(process.hrtime.bigint() - t0) / 1e9;
const b = enc.flush();
const h = crypto.createHash('md5'), ha = crypto.createHash('md5');
h.update(Buffer.from(a.buffer, a.byteOffset, a.length)); h.update(Buffer.from(b.buffer, b.byteOffset, b.length));
ha.update(Buffer.from(a.buffer, a.byteOffset, a.length));
console.log(JSON.stringify({ what: 'Mp3Encoder.encodeBuffer, one call, host Int16Arrays (node ' + process.version + ', N-API addon)', corpus, channels: ch, kbps, frames: nfr - 1,
    seconds: +dt.toFixed(4), frames_per_s: +((nfr - 1) / dt).toFixed(1), md5: h.digest('hex'), bytes: a.length + b.length,
    md5_encode_buffer: ha.digest('hex'), bytes_encode_buffer: a.length }));
