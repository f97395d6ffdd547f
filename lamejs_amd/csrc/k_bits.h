// Bitstream formatter kernel: one wave per frame builds the complete, byte-aligned MP3 frame
// (header, side info, scalefactors, Huffman data, ancillary stuffing) in LDS and copies it out.
// Follows reference BitStream.js: encodeSideInfo2 (259-426), writeMainData (600-689),
// Huffmancode (487-552), huffman_coder_count1 (428-482), drain_into_ancillary (175-213).
// With the reservoir disabled every frame is self-contained (main_data_begin == 0), so the frame
// is: sideinfo_len bytes | main data of gr0ch0, gr0ch1, gr1ch0, gr1ch1 | stuffing to the frame size
// (MPEG-2/2.5 frames carry one granule).
// Variable-length codes are placed with an integer prefix sum over code lengths (order-free).
#pragma once
#include "lhip_defs.h"
#include "lhip_wave.h"
#include "lhip_layout.h"
#include "k_quant.h"   // frame_bits_of / frame_padding

namespace lhip {

// largest legal frame: 320 kbps at 32 kHz (or 160 kbps at 8 kHz) = 1440 bytes (+1 padding byte at 44.1 kHz rates) =
// 361 words, and kb_bits also touches word nwords (a straddling put_bits); lhip_create checks the configuration's
// frame size against BITS_LDS_WORDS.
enum { BITS_LDS_WORDS = 368 };
// w: the frame being assembled; side / q: the frame's side records and quantized spectra, fetched once with many loads in flight
// (read field by field from global memory, every dependent look-up of the packer used to wait out a full memory latency)
struct BitsLds { uint32_t w[BITS_LDS_WORDS]; GrSide side[4]; uint32_t q[288]; };      // q: one granule-channel at a time (4.4 KB: LDS does not limit the occupancy)

// n 32-bit words global -> LDS, eight loads in flight per lane (the trip count is not a compile-time constant, so a plain
// lane-strided loop would issue one load per trip and wait for it)
LHIP_DEV void stage_words(uint32_t* dst, const uint32_t* src, int n, int lane) {
    enum { K = 8 };
    for (int i0 = 0; i0 < n; i0 += LHIP_NL * K) {
        uint32_t v[K];
#pragma unroll
        for (int k = 0; k < K; k++) { const int i = i0 + lane + LHIP_NL * k; v[k] = src[i < n ? i : n - 1]; }   // clamped, not predicated: a predicated load is waited for inside its branch
#pragma unroll
        for (int k = 0; k < K; k++) { const int i = i0 + lane + LHIP_NL * k; if (i < n) dst[i] = v[k]; }
    }
}

// write the low n bits of val at bit position pos (MSB-first stream); n <= 32
LHIP_DEV void put_bits(uint32_t* w, int pos, uint32_t val, int n) {
    if (n <= 0) return;
    if (n < 32) val &= (1u << n) - 1u;
    const int wi = pos >> 5, off = pos & 31;
    if (off + n <= 32) lds_or(&w[wi], val << (32 - off - n));
    else {
        const int n2 = off + n - 32;
        lds_or(&w[wi], val >> n2);
        lds_or(&w[wi + 1], val << (32 - n2));
    }
}

// Huffman-code the pairs [start, end) with table `t` starting at bit `pos`; returns bits written
LHIP_DEV int huff_region(const Tables& T, uint32_t* w, int pos, int t, int start, int end, const int16_t* q, int lane) {
    if (t == 0 || start >= end) return 0;
    const int32_t* hl = T.ht_hlen + T.ht_off[t];
    const int32_t* hc = T.ht_code + T.ht_off[t];
    const int linbits = T.ht_xlen[t];
    int total_all = 0;
    for (int base = start; base < end; base += 2 * LHIP_NL) {
        const int i = base + 2 * lane;
        int cbits = 0, xbits = 0;
        uint32_t code = 0, ext = 0;
        if (i < end) {
            int v1 = q[i], v2 = q[i + 1];
            int x1 = v1 < 0 ? -v1 : v1, x2 = v2 < 0 ? -v2 : v2;
            int xlen = linbits;
            if (x1 != 0) { if (v1 < 0) ext++; cbits--; }
            if (t > 15) {
                if (x1 > 14) { ext |= (uint32_t)(x1 - 15) << 1; xbits = linbits; x1 = 15; }
                if (x2 > 14) { ext <<= linbits; ext |= (uint32_t)(x2 - 15); xbits += linbits; x2 = 15; }
                xlen = 16;
            }
            if (x2 != 0) { ext <<= 1; if (v2 < 0) ext++; cbits--; }
            x1 = x1 * xlen + x2;
            xbits -= cbits;
            cbits += hl[x1];
            code = (uint32_t)hc[x1];
        }
        int tot;
        const int off = wave_excl_scan(cbits + xbits, lane, &tot);
        if (i < end) {
            put_bits(w, pos + off, code, cbits);
            put_bits(w, pos + off + cbits, ext, xbits);
        }
        pos += tot;
        total_all += tot;
    }
    return total_all;
}

LHIP_DEV int count1_region(const Tables& T, uint32_t* w, int pos, const GrSide& gi, const int16_t* q, int lane) {
    const int t = gi.count1table_select + 32;
    const int32_t* hl = T.ht_hlen + T.ht_off[t];
    const int32_t* hc = T.ht_code + T.ht_off[t];
    const int nquads = (gi.count1 - gi.big_values) / 4;
    int total_all = 0;
    for (int base = 0; base < nquads; base += LHIP_NL) {
        const int k = base + lane;
        int n = 0;
        uint32_t val = 0;
        if (k < nquads) {
            const int ix = gi.big_values + 4 * k;
            int huffbits = 0, p = 0;
            int v;
            v = q[ix + 0]; if (v != 0) { p += 8; if (v < 0) huffbits++; }
            v = q[ix + 1]; if (v != 0) { p += 4; huffbits *= 2; if (v < 0) huffbits++; }
            v = q[ix + 2]; if (v != 0) { p += 2; huffbits *= 2; if (v < 0) huffbits++; }
            v = q[ix + 3]; if (v != 0) { p++; huffbits *= 2; if (v < 0) huffbits++; }
            val = (uint32_t)(huffbits + hc[p]);
            n = hl[p];
        }
        int tot;
        const int off = wave_excl_scan(n, lane, &tot);
        if (k < nquads) put_bits(w, pos + off, val, n);
        pos += tot;
        total_all += tot;
    }
    return total_all;
}

// field f (0..14) of a granule-channel's side info: value and width (width 0: the field does not exist for this block type)
LHIP_DEV void side_field(const GrSide& gi, int f, int GR, uint32_t* v, int* n) {
    const bool win = gi.block_type != NORM_TYPE;
    int ts0 = gi.table_select[0], ts1 = gi.table_select[1], ts2 = gi.table_select[2];
    if (ts0 == 14) ts0 = 16;
    if (ts1 == 14) ts1 = 16;
    if (ts2 == 14) ts2 = 16;
    int val = 0, w = 0;
    switch (f) {
        case 0: val = gi.part2_3_length + gi.part2_length; w = 12; break;
        case 1: val = gi.big_values / 2; w = 9; break;
        case 2: val = gi.global_gain; w = 8; break;
        case 3: val = gi.scalefac_compress; w = GR == 2 ? 4 : 9; break;
        case 4: val = win ? 1 : 0; w = 1; break;                                    // window_switching_flag
        case 5: val = win ? gi.block_type : ts0; w = win ? 2 : 5; break;
        case 6: val = win ? 0 : ts1; w = win ? 1 : 5; break;                        // mixed_block_flag | table_select[1]
        case 7: val = win ? ts0 : ts2; w = 5; break;
        case 8: val = win ? ts1 : gi.region0_count; w = win ? 5 : 4; break;
        case 9: val = win ? gi.subblock_gain[0] : gi.region1_count; w = 3; break;
        case 10: val = gi.subblock_gain[1]; w = win ? 3 : 0; break;
        case 11: val = gi.subblock_gain[2]; w = win ? 3 : 0; break;
        case 12: val = gi.preflag; w = GR == 2 ? 1 : 0; break;
        case 13: val = gi.scalefac_scale; w = 1; break;
        case 14: val = gi.count1table_select; w = 1; break;
        default: break;
    }
    *v = (uint32_t)val; *n = w;
}
// the header part is written by lane 0 only; everybody continues from where it stopped
LHIP_DEV int uni_bits_pos(int pos) { return wave_bcast(pos, 0); }

// ---- bit reservoir (extension): the continuous stream (BitStream.js:100-215, 710-780, 836-900) ----
// ancillary stuffing at bit position pos (drain_into_ancillary): "LAME", the coerced version characters, then single bits that
// alternate while the reservoir is in use (gfc.ancillary_flag); one lane
LHIP_DEV int put_ancillary(const Tables& T, uint32_t* w, int pos, int remaining, int* flag) {
    const uint32_t lame[4] = {0x4c, 0x41, 0x4d, 0x45};
    for (int i = 0; i < 4; i++) if (remaining >= 8) { put_bits(w, pos, lame[i], 8); pos += 8; remaining -= 8; }
    if (remaining >= 32)
        for (int i = 0; i < T.n_version_bytes && remaining >= 8; ++i) { remaining -= 8; put_bits(w, pos, (uint32_t)T.version_bytes[i], 8); pos += 8; }
    for (; remaining >= 1; remaining -= 1) { put_bits(w, pos, (uint32_t)*flag, 1); pos += 1; *flag ^= (!T.disable_reservoir ? 1 : 0); }
    return pos;
}
// append `nbytes` bytes of the LDS image (from byte `from`) to the stream: wherever the stream position reaches the start of a
// frame whose header waits in the queue, that header + side info goes in first (putbits2's check before every byte).  The wave
// copies the pieces between such points in parallel; the verdict (stream position, queue read pointer) is stored by lane 0 after
// everybody has read the old one.  Returns the bytes written to `out` so far.
LHIP_DEV int resv_emit(const uint32_t* w, int from, int nbytes, int sideinfo_len, ResvState& rv, uint8_t* out, int n, int lane) {
    int64_t totbit = rv.totbit;
    int wp = rv.w_ptr;
    wave_sync_global();
    for (int i = 0; i < nbytes;) {
        const int64_t gap = (rv.timing[wp] - totbit) >> 3;                // bytes until the next waiting header is due
        if (gap == 0) {
            for (int b = lane; b < sideinfo_len; b += LHIP_NL) out[n + b] = rv.header[wp][b];
            n += sideinfo_len; totbit += 8 * sideinfo_len;
            wp = (wp + 1) & (RESV_HQ - 1);
            continue;
        }
        const int seg = (gap > 0 && gap < nbytes - i) ? (int)gap : nbytes - i;
        for (int b = lane; b < seg; b += LHIP_NL) { const int bi = from + i + b; out[n + b] = (uint8_t)(w[bi >> 2] >> (24 - 8 * (bi & 3))); }
        n += seg; i += seg; totbit += 8 * (int64_t)seg;
    }
    wave_sync_global();
    if (lane == 0) { rv.totbit = totbit; rv.w_ptr = wp; }
    return n;
}
// flush_bitstream (BitStream.js:710-780): pad the stream with ancillary data up to the end of the last frame; one wave per stream
// rv: the stream's reservoir record (global memory or the per-stream kernel's LDS copy); *nout: bytes this launch has appended to the
// stream's output so far
LHIP_DEV void kb_resv_flush(const Tables& T, const Workspace& W, int st, int lane, BitsLds& L, ResvState& rv, int32_t* nout) {
    for (int i = lane; i < BITS_LDS_WORDS; i += LHIP_NL) L.w[i] = 0;
    wave_sync_global();
    const int last_ptr = (rv.h_ptr - 1) & (RESV_HQ - 1), first_ptr = rv.w_ptr;
    int64_t flushbits = rv.timing[last_ptr] - rv.totbit;
    if (flushbits >= 0) {
        const int remaining_headers = ((last_ptr - first_ptr) & (RESV_HQ - 1)) + 1;
        flushbits -= (int64_t)remaining_headers * 8 * T.sideinfo_len;
    }
    flushbits += rv.last_frame_bits;                             // getframebits: with the last frame's padding
    if (flushbits < 0) return;
    int flag = rv.ancillary_flag;
    if (lane == 0) put_ancillary(T, L.w, 0, (int)flushbits, &flag);
    wave_sync_global();
    const int n = resv_emit(L.w, 0, (int)(flushbits >> 3), T.sideinfo_len, rv, W.io[st].out, *nout, lane);
    wave_sync_global();
    if (lane == 0) { rv.ancillary_flag = flag; *nout = n; rv.ResvSize = 0; rv.main_data_begin = 0; }
}

// ---- the pieces of a frame (kb_bits: one wave does them in order; kb_bits_mw_*: the granule-channels side by side on several waves) ----
struct BitsFrame { StreamDesc sd; int k, fidx, frame_bits; };
LHIP_DEV bool bits_frame(const Tables& T, const Workspace& W, const StreamDesc* SD, int fslot, BitsFrame& F) {
    const int st = W.fslot_stream[fslot];
    F.sd = SD[st];
    F.k = fslot - F.sd.fslot0 - 1;
    if (F.k < 0) return false;
    F.fidx = F.sd.out_slot0 + F.k;
    F.frame_bits = frame_bits_of(T, frame_padding(T, F.sd, F.k));
    return true;
}
// header + side info (BitStream.js:259-426) at bit 0 of w; the main data starts at 8 * T.sideinfo_len
LHIP_DEV void bits_header(const Tables& T, const Workspace& W, const BitsFrame& F, const GrSide* side, uint32_t* w, int lane) {
    const int C = T.channels_out, GR = T.mode_gr;
    const bool resv = !T.disable_reservoir;
    const int padding = frame_padding(T, F.sd, F.k);
    int pos = 0;
    if (lane == 0) {
#define PUT(v, n) { put_bits(w, pos, (uint32_t)(v), (n)); pos += (n); }
        PUT(T.out_samplerate < 16000 ? 0xffe : 0xfff, 12) PUT(T.version, 1)       // BitStream.js:262-267
        PUT(4 - 3, 2) PUT(!T.error_protection ? 1 : 0, 1)
        PUT(T.bitrate_index, 4) PUT(T.samplerate_index, 2) PUT(padding, 1) PUT(T.extension, 1)
        PUT(T.mode, 2) PUT(T.mode == 1 ? side[0].mode_ext : 0, 2)                  // mode_ext: the frame's M/S decision in joint stereo (BitStream.js:279)
        PUT(T.copyright, 1) PUT(T.original, 1) PUT(T.emphasis, 2)
        const int mdb = resv ? js_toint32(W.fr[F.fidx].main_data_begin) : 0;       // writeheader shifts the number: ToInt32 of a possibly fractional value
        if (GR == 2) {
            PUT(mdb, 9)
            PUT(0, C == 2 ? 3 : 5)
            for (int ch = 0; ch < C; ch++) {
                const int sc = side[(1 * C) + ch].scfsi;
                for (int band = 0; band < 4; band++) PUT((sc >> band) & 1, 1)
            }
        } else {                                             // MPEG-2/2.5 (BitStream.js:352-405)
            PUT(mdb, 8)
            PUT(0, C)
        }
#undef PUT
    }
    // per-granule-channel side info (BitStream.js:296-350 / 367-405): lane = (granule-channel, field); the widths are
    // turned into bit positions by an exclusive scan (lane order == the reference's gr, ch, field order)
    pos = uni_bits_pos(pos);
    for (int base = 0; base < 16 * GR * C; base += LHIP_NL) {
        const int it = base + lane, gc = it >> 4, f = it & 15;
        int n = 0; uint32_t v = 0;
        if (gc < GR * C) side_field(side[gc], f, GR, &v, &n);
        int tot;
        const int off = wave_excl_scan(n, lane, &tot);
        put_bits(w, pos + off, v, n);
        pos += tot;
    }
}
// main data of one granule-channel (scalefactors + Huffman data, BitStream.js:600-689) from bit `pos` of w; qbuf: 288 words of this wave's LDS.  Returns the
// position behind it (pos + part2_length + part2_3_length of the record: what the side info announces)
// (fetch = false: the caller has staged the spectrum in qbuf already)
LHIP_DEV int bits_main_gc(const Tables& T, const Workspace& W, const BitsFrame& F, const GrSide& gi, int gr, int ch, uint32_t* w, uint32_t* qbuf, int pos, int lane, bool fetch = true) {
    const int C = T.channels_out, GR = T.mode_gr;
    wave_sync();                                                 // the previous granule-channel's spectrum is no longer read
    if (fetch) stage_words(qbuf, (const uint32_t*)(W.l3 + (((int64_t)F.fidx * 2 + gr) * C + ch) * 576), 288, lane);
    wave_sync();
    const int16_t* q = (const int16_t*)qbuf;
    if (GR == 2) {
        // scalefactors: at most 36 short fields
        const int slen1 = T.slen1_tab[gi.scalefac_compress], slen2 = T.slen2_tab[gi.scalefac_compress];
        // lane = band: field widths turned into positions by an exclusive scan (bands shared through scfsi are -1: no field)
        for (int base = 0; base < gi.sfbmax; base += LHIP_NL) {
            const int sfb = base + lane;
            int n = 0, v = 0;
            if (sfb < gi.sfbmax) { v = gi.scalefac[sfb]; if (v != -1) n = sfb < gi.sfbdivide ? slen1 : slen2; }
            int tot;
            const int off = wave_excl_scan(n, lane, &tot);
            put_bits(w, pos + off, (uint32_t)v, n);
            pos += tot;
        }
    } else {
        // MPEG-2/2.5: four partitions with their own field widths (BitStream.js:645-686).  The widths and the
        // partition sizes are what a decoder derives from scalefac_compress (scale_bitcount_lsf packed them):
        // table 0 (no preflag) = slen1*80 + slen2*16 + slen3*4 + slen4 over {6,5,5,5} / {9,9,9,9} entries,
        // table 2 (preflag)    = 500 + slen1*3 + slen2 over {11,10,0,0} / {18,18,0,0} entries.
        const bool pre = gi.preflag != 0, sh = gi.block_type == SHORT_TYPE;
        const int sc = gi.scalefac_compress - (pre ? 500 : 0);
        const int sl0 = pre ? sc / 3 : (sc >> 4) / 5, sl1 = pre ? sc % 3 : (sc >> 4) % 5;
        const int sl2 = pre ? 0 : (sc >> 2) & 3, sl3 = pre ? 0 : sc & 3;
        const int n0 = pre ? (sh ? 18 : 11) : (sh ? 9 : 6), n1 = pre ? (sh ? 18 : 10) : (sh ? 9 : 5);
        const int n2 = pre ? 0 : (sh ? 9 : 5);
        const int b1 = n0, b2 = n0 + n1, b3 = b2 + n2, end = b3 + n2;
        for (int base = 0; base < end; base += LHIP_NL) {
            const int i = base + lane;
            int n = 0, v = 0;
            if (i < end) { n = i < b1 ? sl0 : i < b2 ? sl1 : i < b3 ? sl2 : sl3; v = gi.scalefac[i]; }
            int tot;
            const int off = wave_excl_scan(n, lane, &tot);
            put_bits(w, pos + off, (uint32_t)(v > 0 ? v : 0), n);
            pos += tot;
        }
    }
    int ts0 = gi.table_select[0], ts1 = gi.table_select[1], ts2 = gi.table_select[2];
    if (ts0 == 14) ts0 = 16;
    if (ts1 == 14) ts1 = 16;
    if (ts2 == 14) ts2 = 16;
    if (gi.block_type == SHORT_TYPE) {
        int r1 = 3 * T.sfb_s[3];
        if (r1 > gi.big_values) r1 = gi.big_values;
        pos += huff_region(T, w, pos, ts0, 0, r1, q, lane);
        pos += huff_region(T, w, pos, ts1, r1, gi.big_values, q, lane);
    } else {
        const int bigv = gi.big_values;
        int i = gi.region0_count + 1;
        int r1 = T.sfb_l[i];
        i += gi.region1_count + 1;
        int r2 = T.sfb_l[i];
        if (r1 > bigv) r1 = bigv;
        if (r2 > bigv) r2 = bigv;
        pos += huff_region(T, w, pos, ts0, 0, r1, q, lane);
        pos += huff_region(T, w, pos, ts1, r1, r2, q, lane);
        pos += huff_region(T, w, pos, ts2, r2, bigv, q, lane);
    }
    pos += count1_region(T, w, pos, gi, q, lane);
    return pos;
}
// ancillary stuffing of a frame without the reservoir (drain_into_ancillary): "LAME", coerced version chars, then zero bits; one lane
LHIP_DEV void bits_stuffing(const Tables& T, uint32_t* w, int pos, int frame_bits) {
    int remaining = frame_bits - pos;
    const uint32_t lame[4] = {0x4c, 0x41, 0x4d, 0x45};
    int p2 = pos;
    for (int i = 0; i < 4; i++) if (remaining >= 8) { put_bits(w, p2, lame[i], 8); p2 += 8; remaining -= 8; }
    if (remaining >= 32)
        for (int i = 0; i < T.n_version_bytes && remaining >= 8; ++i) { remaining -= 8; put_bits(w, p2, (uint32_t)T.version_bytes[i], 8); p2 += 8; }
}
// the finished frame's bytes to the stream's output (thread tid of nthr); frames of a stream are consecutive
LHIP_DEV void bits_copy_out(const Tables& T, const Workspace& W, const BitsFrame& F, const uint32_t* w, int tid, int nthr) {
    const int nbytes = F.frame_bits >> 3, k = F.k;
    // frame k starts after the k frames before it (sizes differ by the padding slot): k * base + (number of padded frames among them), the
    // paddings counted in closed form: the lag before frame j is (lag0 - j*frac) mod sr, a padding happens when that value < frac
    const int base = frame_bits_of(T, 0) >> 3;
    int64_t npad = 0;
    if (T.frac_SpF != 0) {
        const int64_t sr = T.out_samplerate;
        int64_t m0 = (int64_t)F.sd.slot_lag % sr; if (m0 < 0) m0 += sr;
        // after k decrements the unwrapped value is m0 - k*frac; each wrap adds sr; wraps = ceil((k*frac - m0)/sr) clipped at 0
        const int64_t need = (int64_t)k * T.frac_SpF - m0;
        npad = need > 0 ? (need + sr - 1) / sr : 0;
    }
    uint8_t* out = W.out + F.sd.out_off + (int64_t)k * base + npad;
    for (int i = tid; i < nbytes; i += nthr) out[i] = (uint8_t)(w[i >> 2] >> (24 - 8 * (i & 3)));
    if (tid == 0) W.frame_bytes[F.fidx] = nbytes;
}

LHIP_DEV void kb_bits(const Tables& T, const Workspace& W, const StreamDesc* SD, int fslot, int lane, BitsLds& L, ResvState* rvp = nullptr, int32_t* nout = nullptr) {
    const int C = T.channels_out;
    BitsFrame F;
    if (!bits_frame(T, W, SD, fslot, F)) return;
    const int st = W.fslot_stream[fslot];
    const int fidx = F.fidx, frame_bits = F.frame_bits;
    const bool resv = !T.disable_reservoir;
    const int nwords = resv ? BITS_LDS_WORDS - 1 : (frame_bits + 31) >> 5;      // reservoir: a frame's data may exceed its nominal size
    for (int i = lane; i < nwords + 1; i += LHIP_NL) L.w[i] = 0;
    const int GR = T.mode_gr;
    static_assert(sizeof(GrSide) % 4 == 0, "side records are staged as words");
    stage_words((uint32_t*)L.side, (const uint32_t*)(W.side + (int64_t)fidx * 2 * C), GR * C * (int)(sizeof(GrSide) / 4), lane);
    wave_sync();
    const GrSide* side = L.side;
    bits_header(T, W, F, side, L.w, lane);
    int pos = 8 * T.sideinfo_len;
    wave_sync();
    int anc_flag = 0;
    if (resv) {                                                  // drain_into_ancillary(resvDrain_pre) precedes the frame's main data
        anc_flag = rvp->ancillary_flag;
        if (lane == 0) put_ancillary(T, L.w, pos, W.fr[fidx].drain_pre, &anc_flag);
        anc_flag = wave_bcast(anc_flag, 0);
        pos += W.fr[fidx].drain_pre;
        wave_sync();
    }
    for (int gr = 0; gr < GR; gr++)
        for (int ch = 0; ch < C; ch++) pos = bits_main_gc(T, W, F, side[gr * C + ch], gr, ch, L.w, L.q, pos, lane);
    if (resv) {
        // bit reservoir: [drain_pre | main data | drain_post] joins the continuous stream, this frame's header + side info joins the
        // queue of headers waiting for the stream to reach their frame start; the reservoir state is committed (format_bitstream,
        // BitStream.js:836-900).  Sequential, one lane: a frame is ~0.4-1.4 KB.
        const FrameResv fr = W.fr[fidx];
        const int main_end = pos;
        if (lane == 0) put_ancillary(T, L.w, pos, fr.drain_post, &anc_flag);
        wave_sync();
        {
            ResvState& rv = *rvp;
            const int sl = T.sideinfo_len;
            const int old = rv.h_ptr;
            wave_sync_global();
            for (int b = lane; b < sl; b += LHIP_NL) rv.header[old][b] = (uint8_t)(L.w[b >> 2] >> (24 - 8 * (b & 3)));
            if (lane == 0) { rv.h_ptr = (old + 1) & (RESV_HQ - 1); rv.timing[(old + 1) & (RESV_HQ - 1)] = rv.timing[old] + frame_bits; }
            wave_sync_global();
            const int chunk_bits = main_end + fr.drain_post - 8 * sl;            // a whole number of bytes (ResvFrameEnd's stuffing)
            const int n0 = *nout;                                                // bytes of this launch's earlier frames
            wave_sync_global();
            const int n1 = resv_emit(L.w, sl, chunk_bits >> 3, sl, rv, W.io[st].out, n0, lane);
            if (lane == 0) {
                const int bits = main_end - fr.drain_pre + fr.drain_post;      // header + side info + main data + drain_post
                rv.main_data_begin = fr.main_data_begin + (double)(frame_bits - bits) / 8;
                rv.ResvSize = fr.ResvSize; rv.ResvMax = fr.ResvMax; rv.ancillary_flag = anc_flag; rv.last_frame_bits = frame_bits;
                for (int i = 0; i < 18; i++) rv.pefirbuf[i] = rv.pefirbuf[i + 1];
                rv.pefirbuf[18] = fr.pefir_new;
                *nout = n1;
                W.frame_bytes[fidx] = n1 - n0;
            }
        }
        return;
    }
    if (lane == 0) bits_stuffing(T, L.w, pos, frame_bits);
    wave_sync();
    bits_copy_out(T, W, F, L.w, lane, LHIP_NL);
}

// ---- one frame, no reservoir, the granule-channels side by side (one-frame launches: a wave alone waits out every table look-up of the packer,
// 5 us per granule-channel; g_frame has idle waves).  Wave `part` of `nparts` = GR * C packs granule-channel `part` at the bit position the side
// records announce (8 sideinfo_len + the lengths of the granule-channels before it) into the ONE frame image wsh (wave 0's; put_bits is an atomic OR);
// wave 0 also writes header, side info and the stuffing.  The packing waves meet twice (image cleared | packed | copied out) on the counters meet[0], meet[1]
// (LDS, zero on entry) -- not at the workgroup's barrier: the other waves of the workgroup are busy saving the state and must not hold the packers up.
LHIP_DEV void kb_bits_mw(const Tables& T, const Workspace& W, const StreamDesc* SD, int fslot, int lane, BitsLds& L, uint32_t* wsh, int part, int nparts, int* meet) {
    BitsFrame F;
    bool ok = bits_frame(T, W, SD, fslot, F);
    const int C = T.channels_out, NGC = T.mode_gr * C;
    if (ok) {
        // everything this wave will read from memory is fetched before the first barrier: the side records and its own granule-channel's spectrum
        const int nwords = (F.frame_bits + 31) >> 5;
        for (int i = part * LHIP_NL + lane; i < nwords + 1; i += nparts * LHIP_NL) wsh[i] = 0;
        stage_words((uint32_t*)L.side, (const uint32_t*)(W.side + (int64_t)F.fidx * 2 * C), NGC * (int)(sizeof(GrSide) / 4), lane);
        if (part < NGC) stage_words(L.q, (const uint32_t*)(W.l3 + (((int64_t)F.fidx * 2 + part / C) * C + part % C) * 576), 288, lane);
    }
    wg_meet(meet, nparts, lane);
    if (ok) {
        const GrSide* side = L.side;
        int pos = 8 * T.sideinfo_len, end = pos;
        for (int gc = 0; gc < NGC; gc++) { const int n = side[gc].part2_3_length + side[gc].part2_length; if (gc < part) pos += n; end += n; }
        if (part == 0) bits_header(T, W, F, side, wsh, lane);
        for (int gc = part; gc < NGC; gc += nparts) {
            (void)bits_main_gc(T, W, F, side[gc], gc / C, gc % C, wsh, L.q, pos, lane, gc != part);
            for (int g2 = gc; g2 < gc + nparts && g2 < NGC; g2++) pos += side[g2].part2_3_length + side[g2].part2_length;
        }
        if (part == 0 && lane == 0) bits_stuffing(T, wsh, end, F.frame_bits);
    }
    wg_meet(meet + 1, nparts, lane);
    if (ok) bits_copy_out(T, W, F, wsh, part * LHIP_NL + lane, nparts * LHIP_NL);
}

}  // namespace lhip
