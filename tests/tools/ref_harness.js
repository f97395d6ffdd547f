/*
 * TEST INFRASTRUCTURE (needs the reference: /root/reference, or -- where that does not exist, i.e. on the GPU box -- the reference's
 * own single-file build oracle/_ref/lame.all.js, tests/tools/ref_bundle.js).
 *
 * Builds an *unmodified* reference encoder with its internals exposed, by repeating
 * the module wiring of the reference's index.js:73-111 (the public Mp3Encoder hides
 * these in a closure).  Used to (a) generate golden fixtures and (b) run differential
 * stage-level checks against the oracle / the new host code.
 */
'use strict';
const path = require('path');
const REF = process.env.LAMEJS_REF || '/root/reference';
const S = path.join(REF, 'src', 'js');

/* opts.jointStereo: gfp.mode = JOINT_STEREO instead of the STEREO that index.js:105 hard-codes -- still the unmodified reference
 * code, only driven with the one setting its public wrapper does not offer (SURVEY.md 8f #3) */
const HAVE_SRC = require('fs').existsSync(path.join(S, 'index.js'));
function refEncoder(channels, samplerate, kbps, opts) {
    if (!HAVE_SRC) return require('./ref_bundle.js').refEncoder(channels, samplerate, kbps, opts);
    const Lame = require(path.join(S, 'Lame.js'));
    const Presets = require(path.join(S, 'Presets.js'));
    const GainAnalysis = require(path.join(S, 'GainAnalysis.js'));
    const QuantizePVT = require(path.join(S, 'QuantizePVT.js'));
    const Quantize = require(path.join(S, 'Quantize.js'));
    const Takehiro = require(path.join(S, 'Takehiro.js'));
    const Reservoir = require(path.join(S, 'Reservoir.js'));
    const MPEGMode = require(path.join(S, 'MPEGMode.js'));
    const BitStream = require(path.join(S, 'BitStream.js'));
    const Version = require(path.join(S, 'Version.js'));
    const VBRTag = require(path.join(S, 'VBRTag.js'));
    function Stub() { this.setModules = function () {}; }

    const lame = new Lame(), gaud = new Stub(), ga = new GainAnalysis(), bs = new BitStream();
    const p = new Presets(), qupvt = new QuantizePVT(), qu = new Quantize(), vbr = new VBRTag();
    const ver = new Version(), id3 = new Stub(), rv = new Reservoir(), tak = new Takehiro();
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
    gfp.mode = (opts && opts.jointStereo && channels == 2) ? MPEGMode.JOINT_STEREO : MPEGMode.STEREO;
    gfp.quality = 3;
    gfp.bWriteVbrTag = false;
    gfp.disable_reservoir = !(opts && opts.reservoir);      /* opts.reservoir: the bit reservoir index.js:108 switches off (SURVEY.md 8f #4) */
    gfp.write_id3tag_automatic = false;
    const rc = lame.lame_init_params(gfp);
    if (rc != 0) throw new Error('lame_init_params rc=' + rc);

    let maxSamples = 1152, mp3buf_size = 0 | (1.25 * maxSamples + 7200), mp3buf = new Int8Array(mp3buf_size);
    return {
        lame, bs, qupvt, qu, tak, rv, gfp, gfc: gfp.internal_flags, psy: lame.enc.psy,
        encodeBuffer(left, right) {
            if (channels == 1) right = left;
            if (left.length > maxSamples) {
                maxSamples = left.length; mp3buf_size = 0 | (1.25 * maxSamples + 7200); mp3buf = new Int8Array(mp3buf_size);
            }
            const n = lame.lame_encode_buffer(gfp, left, right, left.length, mp3buf, 0, mp3buf_size);
            return new Int8Array(mp3buf.subarray(0, n));
        },
        flush() {
            const n = lame.lame_encode_flush(gfp, mp3buf, 0, mp3buf_size);
            return new Int8Array(mp3buf.subarray(0, n));
        }
    };
}

/* the public, unmodified reference encoder */
function refPublic() { return HAVE_SRC ? require(path.join(S, 'index.js')) : require('./ref_bundle.js').load(); }

module.exports = { refEncoder, refPublic, REF };
