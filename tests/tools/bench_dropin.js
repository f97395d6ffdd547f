/*
 * BENCH TOOL (GPU): the drop-in's own throughput through the JavaScript surface (lamejs_amd/js -> N-API addon -> C ABI).
 *
 *   node bench_dropin.js <corpus> <channels> <kbps> <frames> <seed> [reps]
 *       ONE Mp3Encoder.encodeBuffer(left, right) call with host Int16Arrays of <frames> frames (lhip_encode cuts it into chunks and
 *       overlaps their PCIe copies with the encode of the chunk before), then flush().  The clock is around encodeBuffer: H2D + encode
 *       + D2H + the allocation of the returned Int8Array (the library writes into it: no copy).
 *   node bench_dropin.js batch <streams> <frames> <seed0> [reps]
 *       BASELINE configs[4]'s shape through the batch extension: <streams> mono 128 kbps encoders (seeds seed0 ...), ONE
 *       encodeBatch(encoders, lefts) call; the constructors are timed separately (the table blob is built once per configuration).
 *
 * One full-size warm-up (library, tables, kernels, the staging buffers of the chunked path at their final size), then `reps` timed
 * repetitions (default 5), each on fresh encoders; reported: the MEDIAN and all samples.  Prints one JSON line with md5 + length of
 * the encodeBuffer output (what tests/golden/full_md5.json holds) and of encodeBuffer + flush.
 */
'use strict';
const path = require('path'), crypto = require('crypto');
const gen = require('./pcm_gen.js');
const lamejs = require(path.join(__dirname, '..', '..', 'lamejs_amd', 'js', 'index.js'));
const md5 = (...arrs) => { const h = crypto.createHash('md5'); for (const a of arrs) h.update(Buffer.from(a.buffer, a.byteOffset, a.length)); return h.digest('hex'); };
const now = () => Number(process.hrtime.bigint()) / 1e9;
const median = (v) => { const s = v.slice().sort((a, b) => a - b); return s.length % 2 ? s[(s.length - 1) / 2] : 0.5 * (s[s.length / 2 - 1] + s[s.length / 2]); };

if (process.argv[2] == 'batch') {
    const [ns, nfr, seed0, reps] = [+(process.argv[3] || 128), +(process.argv[4] || 1000), +(process.argv[5] || 1000), +(process.argv[6] || 5)];
    const lefts = [];
    for (let i = 0; i < ns; i++) lefts.push(gen.sine(1152 * nfr, 1, seed0 + i)[0]);
    const times = [], ctor = [];
    let outs = null, tails = null;
    for (let r = 0; r <= reps; r++) {                    // r = 0: warm-up
        const c0 = now();
        const encs = [];
        for (let i = 0; i < ns; i++) encs.push(new lamejs.Mp3Encoder(1, 44100, 128));
        const c1 = now();
        const t0 = now();
        outs = lamejs.encodeBatch(encs, lefts);
        const dt = now() - t0;
        tails = lamejs.flushBatch(encs);
        if (r > 0) { times.push(dt); ctor.push(c1 - c0); }
    }
    const dt = median(times);
    console.log(JSON.stringify({ what: 'encodeBatch(encoders, lefts): ONE call for ' + ns + ' mono 128 kbps streams x ' + nfr + ' frames, host Int16Arrays (node ' + process.version + ', N-API addon)',
        streams: ns, frames_per_stream: nfr - 1, seconds: +dt.toFixed(5), samples_s: times.map((t) => +t.toFixed(5)), frames_per_s: +(ns * (nfr - 1) / dt).toFixed(1),
        constructors_s: +median(ctor).toFixed(5), constructor_ms_each: +(1000 * median(ctor) / ns).toFixed(4),
        seeds: lefts.map((_, i) => seed0 + i), md5_encode_buffer: outs.map((a) => md5(a)), bytes_encode_buffer: outs.map((a) => a.length), bytes_flush: tails.map((a) => a.length) }));
} else {
    const [corpus, ch, kbps, nfr, seed, reps] = [process.argv[2] || 'sine', +(process.argv[3] || 2), +(process.argv[4] || 128), +(process.argv[5] || 100000), +(process.argv[6] || 12345), +(process.argv[7] || 5)];
    const [L, R] = gen[corpus](1152 * nfr, ch, seed);
    const times = [], ctor = [];
    let a = null, b = null;
    for (let r = 0; r <= reps; r++) {                    // r = 0: warm-up at full size
        const c0 = now();
        const enc = new lamejs.Mp3Encoder(ch, 44100, kbps);
        const c1 = now();
        const t0 = now();
        a = ch == 2 ? enc.encodeBuffer(L, R) : enc.encodeBuffer(L);
        const dt = now() - t0;
        b = enc.flush();
        if (r > 0) { times.push(dt); ctor.push(c1 - c0); }
    }
    const dt = median(times);
    console.log(JSON.stringify({ what: 'Mp3Encoder.encodeBuffer, one call, host Int16Arrays (node ' + process.version + ', N-API addon); median of ' + reps + ' calls after a full-size warm-up', corpus, channels: ch, kbps, frames: nfr - 1,
        seconds: +dt.toFixed(5), samples_s: times.map((t) => +t.toFixed(5)), frames_per_s: +((nfr - 1) / dt).toFixed(1), constructor_ms: +(1000 * median(ctor)).toFixed(3),
        md5: md5(a, b), bytes: a.length + b.length, md5_encode_buffer: md5(a), bytes_encode_buffer: a.length,
        result_is_exact_arraybuffer: a.buffer.byteLength == a.length && a.byteOffset == 0 }));
}
