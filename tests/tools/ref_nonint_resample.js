/*
 * EVIDENCE TOOL (build container only: needs /root/reference): what the unmodified reference does in the configurations where it
 * resamples by a NON-INTEGER ratio -- the 49 (channels, sample rate, kbps) triples lamejs_amd refuses at construction.
 * For each of them the reference encodes the same 40 frames of a sine twice: through 1152-sample calls (the call pattern of its own
 * Tests.js) and through one large call; after every call the samples waiting in its encoder buffer (gfc.mfbuf) are scanned for NaN.
 * Prints one line per configuration: output rate, when the first NaN sample enters the encoder under either call pattern, how many NaN
 * samples the buffer holds at the end, and whether the two call patterns give the same bytes (a stream encoder's basic contract).
 * usage: node tests/tools/ref_nonint_resample.js > profiles/r03_reference_noninteger_resample.txt
 */
'use strict';
const crypto = require('crypto');
const { refEncoder } = require('./ref_harness.js');
const tables = require('../../lamejs_amd/js/tables.js');
const RATES = [8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000];
const KBPS = [8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, 192, 224, 256, 320];
const NFR = 40;
function sine(n, f, sr) { const a = new Int16Array(n); for (let i = 0; i < n; i++) a[i] = Math.round(12000 * Math.sin(2 * Math.PI * f * i / sr)); return a; }
function nanCount(e) { let c = 0; const mf = e.gfc.mfbuf[0]; for (let i = 0; i < e.gfc.mf_size; i++) if (mf[i] !== mf[i]) c++; return c; }
function run(ch, sr, kb, chunk) {
    const e = refEncoder(ch, sr, kb);
    const L = sine(1152 * NFR, 997, sr), R = ch == 2 ? sine(1152 * NFR, 1499, sr) : undefined;
    const h = crypto.createHash('md5');
    let first = -1, bytes = 0, calls = 0;
    for (let p = 0; p < L.length; p += chunk) {
        const o = e.encodeBuffer(L.subarray(p, p + chunk), R ? R.subarray(p, p + chunk) : undefined);
        h.update(Buffer.from(o.buffer, o.byteOffset, o.length)); bytes += o.length; calls++;
        if (first < 0 && nanCount(e) > 0) first = calls;
    }
    const atEnd = nanCount(e);
    const f = e.flush();
    h.update(Buffer.from(f.buffer, f.byteOffset, f.length)); bytes += f.length;
    return { first, atEnd, bytes, md5: h.digest('hex'), out: e.gfp.out_samplerate };
}
let n = 0, nanSmall = 0, nanLarge = 0, differ = 0;
console.log('# ch  in_rate kbps -> out_rate | 1152-sample calls: first NaN at call, NaN samples buffered at the end | one call: first NaN, NaN at the end | same bytes?');
for (const ch of [1, 2]) for (const sr of RATES) for (const kb of KBPS) {
    let why = null;
    try { tables.buildBlob(ch, sr, kb); } catch (e) { why = String(e.message); }
    if (!why || !/non-integer|integer ratio/i.test(why)) continue;
    let a, b;
    try { a = run(ch, sr, kb, 1152); b = run(ch, sr, kb, 1152 * NFR); } catch (e) { console.log(`${ch} ${sr} ${kb}: reference threw: ${e.message}`); continue; }
    n++; if (a.first > 0 || a.atEnd > 0) nanSmall++; if (b.first > 0 || b.atEnd > 0) nanLarge++; if (a.md5 !== b.md5) differ++;
    console.log(`${ch} ${String(sr).padStart(5)} ${String(kb).padStart(3)} -> ${a.out} | ${a.first < 0 ? 'never' : 'call ' + a.first}, ${a.atEnd} | ${b.first < 0 ? 'never' : 'call ' + b.first}, ${b.atEnd} | ${a.md5 === b.md5 ? 'same' : 'DIFFERENT (' + a.bytes + ' vs ' + b.bytes + ' bytes)'}`);
}
console.log(`# ${n} configurations; NaN samples reach the encoder with 1152-sample calls in ${nanSmall}, with one large call in ${nanLarge}; the two call patterns give different bytes in ${differ}`);
