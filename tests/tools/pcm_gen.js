/*
 * TEST TOOL: deterministic PCM corpora shared by the golden generator, the parity tests and bench.py
 * (bench.py re-implements the same generators in numpy; tests/test_pcm_gen.py pins both to fixtures).
 *
 *  sine   : SURVEY.md 8d config 2/3 -- L = round(8000 sin(2pi 440 i/44100) + 2000(2u-1)),
 *           R = round(6000 sin(2pi 660 i/44100) + 2000(2u'-1)); one LCG stepped twice per sample (u for L, u' for R).
 *           Mono uses only L but still steps the LCG once per sample.
 *  bursts : amplitude-30 noise with 2000-sample bursts of amplitude 20000 every 22050 samples
 *           (exercises ATH auto-adjust, attack detection, short blocks).
 */
'use strict';

function lcg(seed) {
    let s = seed >>> 0;
    return function () {
        /* s = (s*1103515245 + 12345) & 0x7fffffff, exact in 53-bit doubles via split multiply */
        const lo = (s & 0xffff) * 1103515245, hi = ((s >>> 16) * 1103515245) % 32768;
        s = ((hi * 65536) + lo + 12345) % 2147483648;
        return s / 0x7fffffff;
    };
}

function sine(nsamples, channels, seed) {
    const u = lcg(seed === undefined ? 12345 : seed);
    const L = new Int16Array(nsamples), R = channels == 2 ? new Int16Array(nsamples) : null;
    for (let i = 0; i < nsamples; i++) {
        L[i] = Math.round(8000 * Math.sin(2 * Math.PI * 440 * i / 44100) + 2000 * (2 * u() - 1));
        if (R) R[i] = Math.round(6000 * Math.sin(2 * Math.PI * 660 * i / 44100) + 2000 * (2 * u() - 1));
    }
    return [L, R];
}

function bursts(nsamples, channels, seed) {
    const u = lcg(seed === undefined ? 777 : seed);
    const L = new Int16Array(nsamples), R = channels == 2 ? new Int16Array(nsamples) : null;
    for (let i = 0; i < nsamples; i++) {
        const inBurst = (i % 22050) >= 11000 && (i % 22050) < 13000;
        const amp = inBurst ? 20000 : 30;
        L[i] = Math.round(amp * (2 * u() - 1));
        if (R) {
            const inBurstR = ((i + 5000) % 22050) >= 11000 && ((i + 5000) % 22050) < 13000;
            R[i] = Math.round((inBurstR ? 20000 : 30) * (2 * u() - 1));
        }
    }
    return [L, R];
}

/* minimal RIFF/WAVE PCM16 reader (for the reference's testdata fixtures) */
function readWav(buf) {
    let pos = 12, fmt = null;
    while (pos + 8 <= buf.length) {
        const id = buf.toString('ascii', pos, pos + 4), len = buf.readUInt32LE(pos + 4);
        if (id == 'fmt ') fmt = { channels: buf.readUInt16LE(pos + 10), rate: buf.readUInt32LE(pos + 12) };
        if (id == 'data') {
            const n = Math.min(len, buf.length - pos - 8) >> 1;
            const s = new Int16Array(n);
            for (let i = 0; i < n; i++) s[i] = buf.readInt16LE(pos + 8 + 2 * i);
            return { fmt, samples: s };
        }
        pos += 8 + len + (len & 1);
    }
    throw new Error('no data chunk');
}

module.exports = { lcg, sine, bursts, readWav };
