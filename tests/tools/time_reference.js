/*
 * MEASUREMENT TOOL (needs the reference: /root/reference/src/js, or its own single-file build oracle/_ref/lame.all.js -- the form
 * that travels to the GPU box, `make -C oracle ref_js`).
 *
 * Times the UNMODIFIED reference (require('/root/reference/src/js/index.js').Mp3Encoder) on the same
 * synthetic corpus bench.py uses (tests/tools/pcm_gen.js `sine`, SURVEY.md 8d configs 2/3), one thread,
 * 1152-sample encodeBuffer calls, a warm-up run discarded, process.hrtime around the encode loop.
 * Output: one JSON line (bench.py's cpu_baseline.reference_node runs this on the box it benchmarks: same_box = true).
 *
 *   node tests/tools/time_reference.js <channels> <kbps> <frames> [start_at_unix_time] [sine|fixture]
 *       fixture: the reference's own test material (tests/golden/{left,right}44100_full.s16 = testdata/*.wav as raw PCM), repeated until <frames> frames
 *       have been encoded (a fresh encoder per pass) -- what the `dropin_node_1152` lines of bench.py encode
 */
'use strict';
const os = require('os');
const path = require('path');
const REF = process.env.LAMEJS_REF || '/root/reference';
/* LAMEJS_REF_BUNDLE=1, or no /root/reference: the reference's own single-file build (oracle/_ref/lame.all.js, see ref_bundle.js) */
const useBundle = process.env.LAMEJS_USE_BUNDLE == '1' || !require('fs').existsSync(path.join(REF, 'src', 'js', 'index.js'));
const lamejs = useBundle ? require('./ref_bundle.js').load() : require(path.join(REF, 'src', 'js', 'index.js'));
const gen = require('./pcm_gen.js');

const ch = parseInt(process.argv[2] || '1'), kbps = parseInt(process.argv[3] || '128'), frames = parseInt(process.argv[4] || '5000');
const startAt = parseFloat(process.argv[5] || '0');      /* unix time to start the timed run at (the all-cores aggregate starts its workers together) */
const material = process.argv[6] || 'sine';
let n = 1152 * frames, L, R, passes = 1;
if (material == 'fixture') {
    const fs = require('fs');
    const rd = (f) => { const b = fs.readFileSync(path.join(__dirname, '..', 'golden', f)); return new Int16Array(b.buffer, b.byteOffset, b.length >> 1); };
    L = rd('left44100_full.s16'); R = ch == 2 ? rd('right44100_full.s16') : null;
    n = L.length; passes = Math.max(1, Math.round(frames / Math.ceil(n / 1152)));
} else [L, R] = gen.sine(n, ch, 12345);

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
const warmFrames = Math.min(Math.floor(n / 1152), 500);
{   /* warm-up (JIT): a short run, discarded */
    const enc = new lamejs.Mp3Encoder(ch, 44100, kbps);
    for (let i = 0; i < warmFrames * 1152; i += 1152) ch == 2 ? enc.encodeBuffer(L.subarray(i, i + 1152), R.subarray(i, i + 1152)) : enc.encodeBuffer(L.subarray(i, i + 1152));
}
while (startAt > 0 && Date.now() / 1000 < startAt) { /* spin: the workers' clocks start together */ }
const t0 = process.hrtime.bigint();
let bytes = 0;
for (let p = 0; p < passes; p++) bytes = run();
const dt = Number(process.hrtime.bigint() - t0) / 1e9;
const framesDone = passes * Math.ceil(n / 1152);
console.log(JSON.stringify({
    what: 'unmodified lamejs reference, Node.js, 1 thread', node: process.version, v8: process.versions.v8,
    channels: ch, samplerate: 44100, kbps: kbps, frames: framesDone, material: material, seconds: +dt.toFixed(3),
    frames_per_s: +(framesDone / dt).toFixed(1), bytes: bytes, source: useBundle ? 'lame.all.js (the reference\'s own single-file build)' : 'src/js/index.js',
    host: { cpu: os.cpus()[0].model, logical_cores: os.cpus().length },
}));
