/*
 * Drop-in replacement for the encode path of zhuker/lamejs:
 *     const { Mp3Encoder, WavHeader } = require('lamejs_amd/js');
 *     const enc = new Mp3Encoder(channels, sampleRate, kbps);
 *     const bytes = enc.encodeBuffer(left[, right]);   // Int8Array, possibly empty
 *     const tail  = enc.flush();
 * Same constructor arguments, same return types and the same byte stream as the reference
 * (src/js/index.js:66-136, 138-196).  Parameter resolution and every lookup table are computed here,
 * in JavaScript (tables.js), with the engine's own Math.*; the per-frame hot path -- psychoacoustic
 * model, polyphase + MDCT, CBR iteration loop, bitstream formatting -- runs as hand-written HIP
 * kernels on an MI355X behind the C ABI of include/lamejs_hip.h, reached through a thin N-API addon.
 * GPU efficiency comes from batching: pass many frames per encodeBuffer() call (the reference API
 * already allows any length).  There is no CPU fallback.
 * Extension: new Mp3Encoder(2, sampleRate, kbps, { jointStereo: true }) encodes in the reference core's joint-stereo mode (per frame
 * mid/side or left/right), which the reference's own wrapper never selects (index.js:105 hard-codes MPEGMode.STEREO); { reservoir: true }
 * uses the bit reservoir (index.js:108 switches it off).  With the reservoir the frames of a stream are a serial chain -- a launch
 * encodes one frame per stream -- so that mode only uses the GPU well through encodeBatch() over many streams.
 * Extension for callers that feed 1152 samples per call (every usage the reference documents): { pendingFrames: N } lets the encoder
 * hold back up to N frames' worth of input and encode them in ONE launch -- a frame alone is one wavefront's serial search (0.2 - 0.8 ms),
 * sixty-four together take hardly longer.  The BYTE STREAM is the reference's (any chunking of the samples gives the same bytes); only
 * WHICH call returns which bytes changes: calls return empty arrays until N frames are pending, then all their frames at once, and
 * flush() returns the rest.  Off by default: without it every call returns exactly the bytes the reference's call returns.
 */
'use strict';
const path = require('path');
const tables = require('./tables.js');

let addon = null;
let defaultDevice = -1;                         // -1: the HIP runtime's current device
function loadAddon() {
    if (!addon) addon = require(path.join(__dirname, 'addon', 'lhip_napi.node'));
    return addon;
}

/* The table blob of a configuration is a pure function of (channels, samplerate, kbps, options): built once per process and shared by
 * every encoder of that configuration (1024 streams of BASELINE config 5 = one build, not 1024; the library shares the uploaded copy
 * between streams with identical blobs as well).  Configurations outside the envelope throw in buildBlob and are not cached. */
const blobCache = new Map();
function tablesBlob(channels, samplerate, kbps, opts) {
    const key = [channels, samplerate, kbps, opts && opts.jointStereo ? 1 : 0, opts && opts.reservoir ? 1 : 0].join('|');     /* (pendingFrames is host-side only) */
    let blob = blobCache.get(key);
    if (!blob) { blob = tables.buildBlob(channels, samplerate, kbps, opts).blob; blobCache.set(key, blob); }
    return blob;
}

function Mp3Encoder(channels, samplerate, kbps, opts) {
    if (arguments.length != 3 && !(arguments.length == 4 && opts !== null && typeof opts == 'object')) {
        opts = undefined;
        console.error('WARN: Mp3Encoder(channels, samplerate, kbps) not specified');
        channels = 1; samplerate = 44100; kbps = 128;
    }
    const native = loadAddon();
    const blob = tablesBlob(channels, samplerate, kbps, opts);
    const handle = native.create(blob, channels, samplerate, kbps, defaultDevice);
    const hooks = { handle: handle, channels: channels, drain: null, pending: () => 0 };
    Object.defineProperty(this, '_lhip', { value: hooks, enumerable: false });

    /* { pendingFrames: N }: input held back until N frames' worth has accumulated (see the header comment) */
    const pendMax = opts && opts.pendingFrames > 1 ? 1152 * (opts.pendingFrames | 0) : 0;
    let pendL = pendMax ? new Int16Array(pendMax + 1152) : null, pendR = pendMax && channels == 2 ? new Int16Array(pendMax + 1152) : null, pendN = 0;
    const EMPTY = () => new Int8Array(0);
    function drain() {
        if (pendN == 0) return EMPTY();
        const out = native.encode(handle, pendL.subarray(0, pendN), pendR ? pendR.subarray(0, pendN) : null);
        pendN = 0;
        return out;
    }
    /* what is held back belongs in front of anything that reaches the native handle by another way (encodeBatch / flushBatch drain it first and
     * return its bytes in front of their own; getState / setState / seek refuse while input is pending: the state would not contain it) */
    if (pendMax) { hooks.drain = drain; hooks.pending = () => pendN; }
    const noPending = (what) => { if (pendN > 0) throw new Error(what + ': ' + pendN + ' samples are held back by { pendingFrames }; call flush() first'); };
    this.encodeBuffer = function (left, right) {
        if (channels == 1) right = null;
        if (!(left instanceof Int16Array)) left = Int16Array.from(left);
        if (right && !(right instanceof Int16Array)) right = Int16Array.from(right);
        if (!pendMax) return native.encode(handle, left, right || null);
        if (right && right.length != left.length) throw new TypeError('right must be an Int16Array of the same length');       /* (the native path's own check and message) */
        if (left.length > pendL.length - pendN) {                /* does not fit beside what is pending: encode that first, then this */
            const a = drain(), b = left.length >= pendMax ? native.encode(handle, left, right || null) : null;
            if (b) { const r = new Int8Array(a.length + b.length); r.set(a, 0); r.set(b, a.length); return r; }
            pendL.set(left, 0); if (pendR) pendR.set(right || left, 0); pendN = left.length;
            return a;
        }
        pendL.set(left, pendN); if (pendR) pendR.set(right || left, pendN); pendN += left.length;
        return pendN >= pendMax ? drain() : EMPTY();
    };
    this.flush = function () {
        if (!pendMax || pendN == 0) return native.flush(handle);
        const a = drain(), b = native.flush(handle);
        const r = new Int8Array(a.length + b.length); r.set(a, 0); r.set(b, a.length);
        return r;
    };
    /* Extension -- frame-range sharding of ONE stream (include/lamejs_hip.h, lhip_seek ...; DESIGN.md 7): a fresh encoder is put
     * at input sample `samplePos` (a whole number >= 2 of frames) with the seekTailSamples() samples in front of it, encodes a few
     * warm-up frames whose bytes are thrown away, and its getState() at the cut is compared with the getState() of the encoder that
     * came from the left; equal states = equal futures, otherwise setState() transplants the true one. */
    this.seekTailSamples = function () { return native.seekTailSamples(handle); };
    this.seek = function (samplePos, tailLeft, tailRight) { noPending('seek'); native.seek(handle, samplePos, tailLeft, channels == 1 ? null : (tailRight || null)); };
    this.getState = function () { noPending('getState'); return native.stateGet(handle); };
    this.setState = function (state) { noPending('setState'); native.stateSet(handle, state); };
}

/* RIFF/WAVE header reader with the reference's field names (index.js:138-193) */
function WavHeader() { this.dataOffset = 0; this.dataLen = 0; this.channels = 0; this.sampleRate = 0; }
WavHeader.readHeader = function (dataView) {
    const tag = (o) => String.fromCharCode(dataView.getUint8(o), dataView.getUint8(o + 1), dataView.getUint8(o + 2), dataView.getUint8(o + 3));
    const w = new WavHeader();
    if (tag(0) != 'RIFF' || tag(8) != 'WAVE' || tag(12) != 'fmt ') return;
    const fmtLen = dataView.getUint32(16, true);
    if (fmtLen != 16 && fmtLen != 18) throw 'extended fmt chunk not implemented';
    w.channels = dataView.getUint16(22, true);
    w.sampleRate = dataView.getUint32(24, true);
    let pos = 20 + fmtLen, len = 0;
    for (;;) {
        const id = tag(pos);
        len = dataView.getUint32(pos + 4, true);
        if (id == 'data') break;
        pos += len + 8;
    }
    w.dataLen = len;
    w.dataOffset = pos + 8;
    return w;
};

module.exports.Mp3Encoder = Mp3Encoder;
module.exports.WavHeader = WavHeader;
module.exports.deviceCount = function () { return loadAddon().deviceCount(); };

/*
 * Extensions (not part of the reference API) for callers with many independent streams (BASELINE config 5):
 *   setDevice(d)                          encoders constructed afterwards live on HIP device d (deal streams round-robin
 *                                         over deviceCount() GPUs; streams never exchange data)
 *   encodeBatch(encoders, lefts[, rights]) one launch for all the encoders' new samples -> Int8Array per encoder, the same
 *                                         bytes each encoder's own encodeBuffer() would have returned
 *   flushBatch(encoders)                  likewise for flush()
 *   setDevices(mask)                      let the library deal new encoders round-robin over the GPUs named by the bit mask
 * The encoders of one call must share (channels, samplerate, kbps) and the device.
 */
module.exports.setDevice = function (d) { defaultDevice = d | 0; };
/* setDevices(mask): bit d = HIP device d may be used; encoders constructed with the default device (-1) are then dealt round-robin
 * over the allowed GPUs by the library (lhip_set_devices); returns how many devices are allowed.  mask 0 restores the default. */
module.exports.setDevices = function (mask) { defaultDevice = -1; return loadAddon().setDevices(mask); };
/* encoders constructed with { pendingFrames }: what they hold back is encoded first and its bytes are returned in front of the batch's own */
function drainPending(encoders) { return encoders.map((e) => (e._lhip.pending() > 0 ? e._lhip.drain() : null)); }
function prepend(heads, outs) {
    return outs.map((b, i) => { const a = heads[i]; if (!a || a.length == 0) return b; const r = new Int8Array(a.length + b.length); r.set(a, 0); r.set(b, a.length); return r; });
}
module.exports.encodeBatch = function (encoders, lefts, rights) {
    const hs = encoders.map((e) => e._lhip.handle);
    const L = lefts.map((a) => (a instanceof Int16Array ? a : Int16Array.from(a)));
    const stereo = encoders.length > 0 && encoders[0]._lhip.channels == 2;
    const R = stereo && rights ? rights.map((a) => (a instanceof Int16Array ? a : Int16Array.from(a))) : null;
    const heads = drainPending(encoders);
    return prepend(heads, loadAddon().encodeBatch(hs, L, R));
};
module.exports.flushBatch = function (encoders) { const heads = drainPending(encoders); return prepend(heads, loadAddon().flushBatch(encoders.map((e) => e._lhip.handle))); };
