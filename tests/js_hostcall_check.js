/* TEST TOOL: one long Mp3Encoder.encodeBuffer(left, right) call on PCM read from raw s16le files (random material written by the
 * pytest), a second ordinary call on the same encoder, flush(); prints md5 + length.  The long call takes the chunked, overlapped host
 * path of lhip_encode, and the returned Int8Array must be a fresh, exact-size array the library wrote into (lhip_napi.c).
 * usage: node tests/js_hostcall_check.js <left.s16> <right.s16|-> <kbps> <cut samples> [joint] */
'use strict';
const path = require('path'), crypto = require('crypto'), fs = require('fs');
const lamejs = require(path.join(__dirname, '..', 'lamejs_amd', 'js'));
const [lf, rf, kbS, cutS] = process.argv.slice(2), joint = process.argv.slice(6).includes('joint');
const rd = (f) => { const b = fs.readFileSync(f); return new Int16Array(b.buffer, b.byteOffset, b.length >> 1); };
const L = rd(lf), R = rf == '-' ? null : rd(rf), cut = +cutS, ch = R ? 2 : 1;
const enc = joint ? new lamejs.Mp3Encoder(ch, 44100, +kbS, { jointStereo: true }) : new lamejs.Mp3Encoder(ch, 44100, +kbS);
const a = R ? enc.encodeBuffer(L.subarray(0, cut), R.subarray(0, cut)) : enc.encodeBuffer(L.subarray(0, cut));
const b = R ? enc.encodeBuffer(L.subarray(cut), R.subarray(cut)) : enc.encodeBuffer(L.subarray(cut));
const c = enc.flush();
for (const x of [a, b, c]) if (!(x instanceof Int8Array)) throw new Error('results must be Int8Arrays');
if (a.byteOffset != 0 || a.buffer.byteLength != a.length) throw new Error('encodeBuffer must return a fresh exact-size array');
const h = crypto.createHash('md5');
for (const x of [a, b, c]) h.update(Buffer.from(x.buffer, x.byteOffset, x.length));
console.log(JSON.stringify({ bytes: a.length + b.length + c.length, md5: h.digest('hex'), first_call_bytes: a.length }));
