/*
 * TEST / MEASUREMENT INFRASTRUCTURE: the UNMODIFIED reference as its own single-file build artefact (`lame.all.js`, produced by the
 * reference's makeall.sh from src/js), so that the reference itself can run where /root/reference does not exist -- the GPU box.
 * `make -C oracle ref_js` copies the file to oracle/_ref/lame.all.js (untracked, shipped with the lease like liblame_oracle.so).
 * The product never loads it (tests/test_abi.py::test_product_does_not_reference_oracle).
 *
 * The bundle is a plain script (`function lamejs() { ... lamejs.Mp3Encoder = Mp3Encoder; ... } lamejs();`): it is evaluated in a
 * fresh function scope and the `lamejs` function object -- carrying Mp3Encoder / WavHeader -- is returned.  For the two settings the
 * public wrapper hard-codes (index.js:105 STEREO, index.js:108 disable_reservoir) the module constructors the bundle keeps in its
 * closure are additionally handed out: one line is appended INSIDE the closure of the in-memory copy (the file is not touched) and
 * refEncoder() repeats the wiring of index.js:73-111 with them -- the same thing tests/tools/ref_harness.js does with src/js.
 */
'use strict';
const fs = require('fs');
const path = require('path');

function bundlePath() {
    const cands = [process.env.LAMEJS_REF_BUNDLE, path.join(__dirname, '..', '..', 'oracle', '_ref', 'lame.all.js'), '/root/reference/lame.all.js'];
    for (const c of cands) if (c && fs.existsSync(c)) return c;
    return null;
}

let cached = null;
function load() {
    if (cached) return cached;
    const p = bundlePath();
    if (!p) throw new Error('reference bundle not found (make -C oracle ref_js copies /root/reference/lame.all.js to oracle/_ref/)');
    let src = fs.readFileSync(p, 'utf8');
    const anchor = 'lamejs.Mp3Encoder = Mp3Encoder;';
    const at = src.lastIndexOf(anchor);
    if (at < 0) throw new Error('reference bundle: unexpected layout');
    const hook = 'lamejs.__modules = { Lame: Lame, Presets: Presets, GainAnalysis: GainAnalysis, QuantizePVT: QuantizePVT, Quantize: Quantize, ' +
                 'Takehiro: Takehiro, Reservoir: Reservoir, MPEGMode: MPEGMode, BitStream: BitStream, Version: Version, VBRTag: VBRTag };\n';
    src = src.slice(0, at) + hook + src.slice(at);
    cached = (new Function(src + '\nreturn lamejs;'))();
    cached.__path = p;
    return cached;
}

/* the reference's modules wired as index.js:73-111 wires them, with the two settings its wrapper does not offer */
function refEncoder(channels, samplerate, kbps, opts) {
    const M = load().__modules;
    function Stub() { this.setModules = function () {}; }
    const lame = new M.Lame(), gaud = new Stub(), ga = new M.GainAnalysis(), bs = new M.BitStream();
    const p = new M.Presets(), qupvt = new M.QuantizePVT(), qu = new M.Quantize(), vbr = new M.VBRTag();
    const ver = new M.Version(), id3 = new Stub(), rv = new M.Reservoir(), tak = new M.Takehiro();
    const parse = new Stub(), mpg = {};
    lame.setModules(ga, bs, p, qupvt, qu, vbr, ver, id3, mpg);
    bs.setModules(ga, mpg, ver, vbr);
    id3.setModules(bs, ver);
    p.setModules(lame);
    qu.setModules(bs, rv, qupvt, tak);
    qupvt.setModules(tak, rv, lame.enc.psy);
    rv.setModules(bs);
    tak.setModules(qupvt);
    vbr.setModules(lame, bs, ver);
    gaud.setModules(parse, mpg);
    parse.setModules(ver, id3, p);
    const gfp = lame.lame_init();
    gfp.num_channels = channels;
    gfp.in_samplerate = samplerate;
    gfp.brate = kbps;
    gfp.mode = (opts && opts.jointStereo && channels == 2) ? M.MPEGMode.JOINT_STEREO : M.MPEGMode.STEREO;
    gfp.quality = 3;
    gfp.bWriteVbrTag = false;
    gfp.disable_reservoir = !(opts && opts.reservoir);
    gfp.write_id3tag_automatic = false;
    const rc = lame.lame_init_params(gfp);
    if (rc != 0) throw new Error('lame_init_params rc=' + rc);
    let maxSamples = 1152, mp3buf_size = 0 | (1.25 * maxSamples + 7200), mp3buf = new Int8Array(mp3buf_size);
    return {
        encodeBuffer(left, right) {
            if (channels == 1) right = left;
            if (left.length > maxSamples) { maxSamples = left.length; mp3buf_size = 0 | (1.25 * maxSamples + 7200); mp3buf = new Int8Array(mp3buf_size); }
            const n = lame.lame_encode_buffer(gfp, left, right, left.length, mp3buf, 0, mp3buf_size);
            return new Int8Array(mp3buf.subarray(0, n));
        },
        flush() {
            const n = lame.lame_encode_flush(gfp, mp3buf, 0, mp3buf_size);
            return new Int8Array(mp3buf.subarray(0, n));
        }
    };
}

module.exports = { load, refEncoder, bundlePath };
