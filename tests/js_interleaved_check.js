/*
 * TEST: several LIVE encoders of different configurations in one process, called alternately -- the reference's usage once a process serves more than one
 * stream (index.js:117-135, worker-example/worker.js:41-64).  Two Mp3Encoder objects (mono 128 k, stereo 128 k) take 1152 samples per call in turn, an
 * encodeBatch over a third group (three mono 64 k streams) runs in between, a joint-stereo + reservoir encoder and a { pendingFrames } encoder (which also goes
 * through encodeBatch once: what it holds back must come out in front) run beside them.  Every stream must give the bytes it gives when it runs alone.
 * usage: node js_interleaved_check.js <frames>    -> one JSON line { alone: {name: md5}, interleaved: {name: md5}, bytes: {name: n} }
 */
'use strict';
const path = require('path'), crypto = require('crypto');
const gen = require('./tools/pcm_gen.js');
const lamejs = require(path.join(__dirname, '..', 'lamejs_amd', 'js', 'index.js'));
const nfr = +(process.argv[2] || 40);
const spec = {
    mono: { ch: 1, kbps: 128, pcm: gen.sine(1152 * nfr + 100, 1, 9001) },
    stereo: { ch: 2, kbps: 128, pcm: gen.bursts(1152 * nfr + 200, 2, 9002) },
    b0: { ch: 1, kbps: 64, pcm: gen.sine(1152 * nfr, 1, 9003) }, b1: { ch: 1, kbps: 64, pcm: gen.bursts(1152 * nfr - 300, 1, 9004) }, b2: { ch: 1, kbps: 64, pcm: gen.sine(1152 * nfr - 900, 1, 9005) },
    jr: { ch: 2, kbps: 192, opts: { jointStereo: true, reservoir: true }, pcm: gen.bursts(1152 * nfr + 300, 2, 9006) },
    pend: { ch: 2, kbps: 128, opts: { pendingFrames: 5 }, pcm: gen.sine(1152 * nfr + 50, 2, 9007) },
};
const mk = (s, plain) => (s.opts && !(plain && s.opts.pendingFrames) ? new lamejs.Mp3Encoder(s.ch, 44100, s.kbps, s.opts) : new lamejs.Mp3Encoder(s.ch, 44100, s.kbps));
const buf = (b) => { if (!(b instanceof Int8Array)) throw new Error('Int8Array expected'); return Buffer.from(b.buffer, b.byteOffset, b.length); };
const cut = (a, i, n) => a.subarray(Math.min(i, a.length), Math.min(i + n, a.length));
const feed = (e, s, i, n) => buf(s.ch == 2 ? e.encodeBuffer(cut(s.pcm[0], i, n), cut(s.pcm[1], i, n)) : e.encodeBuffer(cut(s.pcm[0], i, n)));
const md5 = (parts) => crypto.createHash('md5').update(Buffer.concat(parts)).digest('hex');
/* every stream alone (the { pendingFrames } stream without the option: the byte STREAM must not depend on it) */
const alone = {}, bytes = {};
for (const k of Object.keys(spec)) {
    const s = spec[k], e = mk(s, true), parts = [];
    for (let i = 0; i < s.pcm[0].length; i += 1152) parts.push(feed(e, s, i, 1152));
    parts.push(buf(e.flush()));
    alone[k] = md5(parts); bytes[k] = Buffer.concat(parts).length;
}
/* all of them live, called in turn */
const enc = {}, parts = {};
for (const k of Object.keys(spec)) { enc[k] = mk(spec[k], false); parts[k] = []; }
const group = ['b0', 'b1', 'b2'];
const maxN = Math.max(...Object.keys(spec).map((k) => spec[k].pcm[0].length));
for (let i = 0, call = 0; i < maxN; i += 1152, call++) {
    parts.mono.push(feed(enc.mono, spec.mono, i, 1152));
    parts.stereo.push(feed(enc.stereo, spec.stereo, i, 1152));
    lamejs.encodeBatch(group.map((k) => enc[k]), group.map((k) => cut(spec[k].pcm[0], i, 1152))).forEach((b, j) => parts[group[j]].push(buf(b)));
    parts.jr.push(feed(enc.jr, spec.jr, i, 1152));
    parts.mono.push(feed(enc.mono, spec.mono, 0, 0));                          /* a call without samples in between */
    if (call == 7) {                                                           /* the pending encoder through encodeBatch once, with input held back */
        if (enc.pend._lhip.pending() == 0) throw new Error('test assumption: input is pending at call 7');
        let threw = false; try { enc.pend.getState(); } catch (e) { threw = true; }
        if (!threw) throw new Error('getState must refuse while input is held back');
        const s = spec.pend;
        parts.pend.push(buf(lamejs.encodeBatch([enc.pend], [cut(s.pcm[0], i, 1152)], [cut(s.pcm[1], i, 1152)])[0]));
    } else parts.pend.push(feed(enc.pend, spec.pend, i, 1152));
}
parts.mono.push(buf(enc.mono.flush())); parts.jr.push(buf(enc.jr.flush()));
lamejs.flushBatch(group.map((k) => enc[k])).forEach((b, j) => parts[group[j]].push(buf(b)));
parts.stereo.push(buf(enc.stereo.flush()));
parts.pend.push(buf(lamejs.flushBatch([enc.pend])[0]));                        /* what is still held back comes out in front of the flush */
let bad = false; try { enc.pend.encodeBuffer(new Int16Array(10), new Int16Array(9)); } catch (e) { bad = e instanceof TypeError; }
if (!bad) throw new Error('a right channel of another length must be refused with a TypeError in { pendingFrames } mode too');
const inter = {};
for (const k of Object.keys(spec)) inter[k] = md5(parts[k]);
console.log(JSON.stringify({ alone: alone, interleaved: inter, bytes: bytes }));
