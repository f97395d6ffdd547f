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
 *   node bench_dropin.js calls <channels> <kbps> fixture|sine [frames] [reps] [pendingFrames]
 *       THE REFERENCE'S DOCUMENTED CALL PATTERN (README.md:69-74, 103-108; Tests.js:19-33): one Mp3Encoder, 1152 samples per encodeBuffer()
 *       call (left.subarray(i, i + 1152)), every non-empty return collected, flush() at the end.  `fixture`: the reference's own test
 *       material (testdata/Left44100.wav [+ Right44100.wav] = tests/golden/*44100_full.s16, 287 calls; the md5 of the collected bytes is the
 *       one Tests.js' outputs have, SURVEY.md 8c); `sine`: <frames> frames (default 2000) of the bench stream, md5 in tests/golden/calls_md5.json
 *       (made from the unmodified reference by tests/tools/gen_calls_md5.js).  The clock is around the whole loop incl. flush().
 *   node bench_dropin.js callsbatch <streams> <frames> <seed0> [reps]
 *       the same pattern over many streams: <streams> mono 128 kbps encoders, ONE encodeBatch() call per 1152 samples of every stream
 *       (<frames> calls), the streams of BASELINE configs[4] (md5 per stream in tests/golden/full_md5.json).
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

if (process.argv[2] == 'calls') {
    const fs = require('fs');
    const [ch, kbps, src, nfrArg, reps, pending] = [+(process.argv[3] || 2), +(process.argv[4] || 128), process.argv[5] || 'fixture', +(process.argv[6] || 2000), +(process.argv[7] || 3), +(process.argv[8] || 0)];
    let L, R = null;
    if (src == 'fixture') {
        const rd = (f) => { const b = fs.readFileSync(path.join(__dirname, '..', 'golden', f)); return new Int16Array(b.buffer, b.byteOffset, b.length >> 1); };
        L = rd('left44100_full.s16'); if (ch == 2) R = rd('right44100_full.s16');
    } else [L, R] = gen.sine(1152 * nfrArg, ch, 12345);
    const ncalls = Math.ceil(L.length / 1152);
    const run = () => {
        const enc = pending > 1 ? new lamejs.Mp3Encoder(ch, 44100, kbps, { pendingFrames: pending }) : new lamejs.Mp3Encoder(ch, 44100, kbps);
        const parts = [], per = new Float64Array(ncalls);
        const t0 = now();
        for (let i = 0, c = 0; i < L.length; i += 1152, c++) {
            const t1 = now();
            const b = ch == 2 ? enc.encodeBuffer(L.subarray(i, i + 1152), R.subarray(i, i + 1152)) : enc.encodeBuffer(L.subarray(i, i + 1152));
            per[c] = now() - t1;
            if (b.length > 0) parts.push(b);
        }
        const f = enc.flush();
        if (f.length > 0) parts.push(f);
        return { dt: now() - t0, parts, per };
    };
    run();                                                   // warm-up: library, tables, kernels
    const times = []; let last = null;
    for (let r = 0; r < reps; r++) { last = run(); times.push(last.dt); }
    const dt = median(times), bytes = last.parts.reduce((a, b) => a + b.length, 0);
    const per = Array.from(last.per.slice(2)).sort((a, b) => a - b);
    console.log(JSON.stringify({ what: 'the reference\'s documented call pattern: one Mp3Encoder, 1152 samples per encodeBuffer() call, flush() at the end' + (pending > 1 ? ' -- with the extension { pendingFrames: ' + pending + ' }: input held back until that many frames are pending, then ONE launch (same byte stream; the bytes arrive in later calls)' : '') + ' (node ' + process.version + ', N-API addon); median of ' + reps + ' runs after a warm-up run',
        source: src == 'fixture' ? 'the reference\'s own testdata/Left44100.wav' + (ch == 2 ? ' + Right44100.wav' : '') : 'sine, seed 12345', channels: ch, kbps, calls: ncalls, frames: ncalls + 1,
        seconds: +dt.toFixed(5), samples_s: times.map((t) => +t.toFixed(5)), frames_per_s: +((ncalls + 1) / dt).toFixed(1), ms_per_call: +(1000 * dt / ncalls).toFixed(4),
        call_us_median: +(1e6 * per[per.length >> 1]).toFixed(1), call_us_p95: +(1e6 * per[Math.floor(per.length * 0.95)]).toFixed(1), call_us_min: +(1e6 * per[0]).toFixed(1),
        md5: md5(...last.parts), bytes }));
} else if (process.argv[2] == 'callsbatch') {
    const [ns, nfr, seed0, reps] = [+(process.argv[3] || 64), +(process.argv[4] || 1000), +(process.argv[5] || 1000), +(process.argv[6] || 3)];
    const lefts = [];
    for (let i = 0; i < ns; i++) lefts.push(gen.sine(1152 * nfr, 1, seed0 + i)[0]);
    const run = () => {
        const encs = [];
        for (let i = 0; i < ns; i++) encs.push(new lamejs.Mp3Encoder(1, 44100, 128));
        const hs = encs.map(() => crypto.createHash('md5')), nb = new Array(ns).fill(0);
        const t0 = now();
        for (let c = 0; c < nfr; c++) {
            const outs = lamejs.encodeBatch(encs, lefts.map((a) => a.subarray(1152 * c, 1152 * c + 1152)));
            for (let i = 0; i < ns; i++) if (outs[i].length) { hs[i].update(Buffer.from(outs[i].buffer, outs[i].byteOffset, outs[i].length)); nb[i] += outs[i].length; }
        }
        const dt = now() - t0;
        lamejs.flushBatch(encs);
        return { dt, md5s: hs.map((h) => h.digest('hex')), nb };
    };
    run();
    const times = []; let last = null;
    for (let r = 0; r < reps; r++) { last = run(); times.push(last.dt); }
    const dt = median(times);
    console.log(JSON.stringify({ what: 'the 1152-sample call pattern over ' + ns + ' mono 128 kbps streams: ONE encodeBatch() call per 1152 samples of every stream, ' + nfr + ' calls (node ' + process.version + ', N-API addon); the clock includes hashing the returned arrays',
        streams: ns, calls: nfr, frames_per_stream: nfr - 1, seconds: +dt.toFixed(5), samples_s: times.map((t) => +t.toFixed(5)), frames_per_s: +(ns * (nfr - 1) / dt).toFixed(1), ms_per_call: +(1000 * dt / nfr).toFixed(4),
        seeds: lefts.map((_, i) => seed0 + i), md5_encode_buffer: last.md5s, bytes_encode_buffer: last.nb }));
} else if (process.argv[2] == 'batch') {
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
