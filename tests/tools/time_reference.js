/*
 * MEASUREMENT TOOL (only usable where /root/reference is mounted -- NOT on the GPU box).
 *
 * Times the UNMODIFIED reference (require('/root/reference/src/js/index.js').Mp3Encoder) on the same
 * synthetic corpus bench.py uses (tests/tools/pcm_gen.js `sine`, SURVEY.md 8d configs 2/3), one thread,
 * 1152-sample encodeBuffer calls, a warm-up run discarded, process.hrtime around the encode loop.
 * Output: one JSON line; committed under profiles/ next to the GPU bench lines as the
 * "reference's single-threaded Node.js path" figure.  The host is THIS container's CPU (stated in the
 * line), not the GPU box's: bench.py's cpu_baseline (the C port of the same algorithm) is what is timed
 * on the GPU box itself.
 *
 *   node tests/tools/time_reference.js <channels> <kbps> <frames>
 */
'use strict';
const os = require('os');
const path = require('path');
const REF = process.env.LAMEJS_REF || '/root/reference';
const lamejs = require(path.join(REF, 'src', 'js', 'index.js'));
const gen = require('./pcm_gen.js');

const ch = parseInt(process.argv[2] || '1'), kbps = parseInt(process.argv[3] || '128'), frames = parseInt(process.argv[4] || '5000');
const n = 1152 * frames;
const [L, R] = gen.sine(n, ch, 12345);

function run() {
    const enc = new lamejs.Mp3Encoder(ch, 44100, kbps);
    let bytes = 0;
    for (let i = 0; i < n; i += 1152) {
        const l = L.subarray(i, i + 1152);
        const out = ch == 2 ? enc.encodeBuffer(l, R.subarray(i, i + 1152)) : enc.encodeBuffer(l);
        bytes += out.length;
    }
    bytes += enc.flush().length;
    return bytes;
}
const warmFrames = Math.min(frames, 500);
{   /* warm-up (JIT): a short run, discarded */
    const enc = new lamejs.Mp3Encoder(ch, 44100, kbps);
    for (let i = 0; i < warmFrames * 1152; i += 1152) ch == 2 ? enc.encodeBuffer(L.subarray(i, i + 1152), R.subarray(i, i + 1152)) : enc.encodeBuffer(L.subarray(i, i + 1152));
}
const t0 = process.hrtime.bigint();
const bytes = run();
const dt = Number(process.hrtime.bigint() - t0) / 1e9;
console.log(JSON.stringify({
    what: 'unmodified lamejs reference, Node.js, 1 thread', node: process.version, v8: process.versions.v8,
    channels: ch, samplerate: 44100, kbps: kbps, frames: frames, seconds: +dt.toFixed(3),
    frames_per_s: +(frames / dt).toFixed(1), bytes: bytes,
    host: { cpu: os.cpus()[0].model, logical_cores: os.cpus().length, note: 'build container, not the GPU box' },
}));
