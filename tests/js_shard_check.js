/* TEST TOOL: frame-range sharding of ONE stream through the JavaScript surface (Mp3Encoder.seek / getState / setState, extension):
 * the stream is cut at the given frame numbers, every piece goes to its own encoder (seek + H warm-up frames, state verified
 * against the state the previous piece ended in, transplanted on a miss) and the concatenation is compared with one encoder
 * doing the whole stream.  Also touches setDevices (the library deals encoders over the allowed GPUs).
 * usage: node tests/js_shard_check.js <corpus> <channels> <kbps> <nframes> <H> <cut> [<cut> ...] */
'use strict';
const path = require('path'), crypto = require('crypto');
const lamejs = require(path.join(__dirname, '..', 'lamejs_amd', 'js'));
const gen = require('./tools/pcm_gen.js');
const [corpus, chS, kbS, nfS, hS, ...cutS] = process.argv.slice(2);
const ch = +chS, kbps = +kbS, n = +nfS * 1152, H = +hS, fs = 1152;
const [L, R] = gen[corpus](n, ch);
const buf = (b) => Buffer.from(b.buffer, b.byteOffset, b.length);
const enc1 = (e, a, b) => buf(ch == 2 ? e.encodeBuffer(L.subarray(a, b), R.subarray(a, b)) : e.encodeBuffer(L.subarray(a, b)));
const same = (x, y) => x.length === y.length && Buffer.compare(Buffer.from(x.buffer, x.byteOffset, x.length), Buffer.from(y.buffer, y.byteOffset, y.length)) === 0;
const allowed = lamejs.setDevices(0);                         // 0 = every device
const whole = new lamejs.Mp3Encoder(ch, 44100, kbps);
const ref = Buffer.concat([enc1(whole, 0, n), buf(whole.flush())]);
const bounds = [0].concat(cutS.map((c) => +c * fs), [n]);
const parts = [];
let prev = null, missed = 0;
for (let r = 0; r + 1 < bounds.length; r++) {
    const a = bounds[r], b = bounds[r + 1], e = new lamejs.Mp3Encoder(ch, 44100, kbps);
    if (r > 0) {
        const p0 = a - H * fs, nt = e.seekTailSamples();
        e.seek(p0, L.subarray(p0 - nt, p0), ch == 2 ? R.subarray(p0 - nt, p0) : null);
        enc1(e, p0, a);                                        // warm-up frames, bytes thrown away
        if (!same(e.getState(), prev)) { missed++; e.setState(prev); if (!same(e.getState(), prev)) throw new Error('setState / getState round trip'); }
    }
    parts.push(enc1(e, a, b));
    prev = e.getState();
    if (!(prev instanceof Uint8Array)) throw new Error('getState must return a Uint8Array');
    if (r + 2 == bounds.length) parts.push(buf(e.flush()));
}
let threw = false;
try { whole.seek(4 * fs, L.subarray(0, whole.seekTailSamples()), ch == 2 ? R.subarray(0, whole.seekTailSamples()) : null); } catch (err) { threw = true; }
if (!threw) throw new Error('seek on a used encoder must throw');
const md5 = (x) => crypto.createHash('md5').update(x).digest('hex');
console.log(JSON.stringify({ whole: md5(ref), pieces: md5(Buffer.concat(parts)), bytes: ref.length, missed: missed, devices_allowed: allowed }));
