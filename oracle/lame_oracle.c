/*
 * TEST INFRASTRUCTURE -- CPU oracle for the lamejs per-frame encode path (unity build).
 *
 * Stream driver + frame driver + bitstream formatter.  Restates reference
 *   Lame.js   lame_encode_buffer / _sample (1490-1667), lame_encode_flush (1381-1488)
 *   Encoder.js lame_encode_mp3_frame (388-659), lame_encode_frame_init (287-326)
 *   BitStream.js encodeSideInfo2 (259-426), Huffmancode (487-552), huffman_coder_count1
 *             (428-482), writeMainData (600-689), drain_into_ancillary (175-213)
 * for the fixed Mp3Encoder configuration (CBR, reservoir disabled, MPEG-1).
 *
 * PINNING: validated byte-for-byte against the unmodified reference running under Node on
 * the reference's own fixtures (testdata/*.wav -> MD5s of SURVEY.md 8c) and on synthetic
 * corpora; see tests/test_oracle_golden.py and tests/tools/gen_golden.js.
 */
#include "lo_common.h"
#include "lo_mdct.c"
#include "lo_psy.c"
#include "lo_quant.c"
#include "lame_oracle.h"

/* ------------------------------------------------------------------ */
/* blob loader                                                         */
/* ------------------------------------------------------------------ */

typedef struct { char name[32]; uint32_t dtype, count, offset, pad; } lhtb_entry;

static const lhtb_entry* lo_find(const uint8_t* blob, const char* name) {
    uint32_t n;
    const lhtb_entry* e = (const lhtb_entry*)(blob + 16);
    memcpy(&n, blob + 8, 4);
    for (uint32_t i = 0; i < n; i++)
        if (strncmp(e[i].name, name, 32) == 0) return &e[i];
    return NULL;
}

static const void* lo_arr(const uint8_t* blob, const char* name, uint32_t dtype, int* count) {
    const lhtb_entry* e = lo_find(blob, name);
    if (!e || e->dtype != dtype) { fprintf(stderr, "lame_oracle: blob entry '%s' missing or wrong type\n", name); abort(); }
    if (count) *count = (int)e->count;
    return blob + e->offset;
}

static int lo_named(const uint8_t* blob, const char* names_key, const char* key) {
    int n;
    const int32_t* chars = (const int32_t*)lo_arr(blob, names_key, 1, &n);
    int idx = 0, i = 0;
    size_t kl = strlen(key);
    while (i < n && chars[i]) {
        int j = i;
        while (j < n && chars[j] && chars[j] != ',') j++;
        if ((size_t)(j - i) == kl) {
            size_t t = 0;
            while (t < kl && chars[i + t] == key[t]) t++;
            if (t == kl) return idx;
        }
        idx++;
        i = (chars[j] == ',') ? j + 1 : j;
    }
    fprintf(stderr, "lame_oracle: config key '%s' missing\n", key);
    abort();
}

static int lo_load_cfg(lo_cfg* c, const void* blob_in, size_t nbytes) {
    uint32_t magic, total;
    if (nbytes < 16) return -1;
    memcpy(&magic, blob_in, 4);
    memcpy(&total, (const uint8_t*)blob_in + 12, 4);
    if (magic != 0x4254484cu || total > nbytes) return -1;
    c->blob_copy = malloc(nbytes);
    memcpy(c->blob_copy, blob_in, nbytes);
    const uint8_t* b = (const uint8_t*)c->blob_copy;
    const int32_t* ci = (const int32_t*)lo_arr(b, "cfg_i", 1, NULL);
    const double* cd = (const double*)lo_arr(b, "cfg_d", 3, NULL);
#define CI(f) c->f = ci[lo_named(b, "cfg_i_names", #f)]
#define CD(f) c->f = cd[lo_named(b, "cfg_d_names", #f)]
    CI(channels_out); CI(mode); CI(mode_gr); CI(version); CI(samplerate_index); CI(bitrate_index); CI(brate);
    CI(out_samplerate); CI(sideinfo_len); CI(frac_SpF); CI(noise_shaping); CI(noise_shaping_amp);
    CI(noise_shaping_stop); CI(subblock_gain); CI(use_best_huffman); CI(full_outer_loop); CI(substep_shaping);
    CI(sfb21_extra); CI(quant_comp); CI(quant_comp_short); CI(short_blocks_coupled); CI(useTemporal);
    CI(ATH_useAdjust); CI(athaa_loudapprox); CI(copyright); CI(original); CI(emphasis); CI(extension);
    CI(error_protection); CI(npart_l); CI(npart_s); CI(in_samplerate); CI(rs_filter_l); CI(rs_bpc);
    CI(disable_reservoir);
    CD(resample_ratio);
    CD(scale); CD(attackthre); CD(attackthre_s); CD(interChRatio); CD(masking_lower_long); CD(masking_lower_short);
    CD(ATH_aaSensitivityP); CD(ATH_floor); CD(decay); CD(ma_max_i1); CD(ma_max_i2); CD(ma_max_m); CD(VO_SCALE);
    CD(msfix); CD(ATHlower);
#define AF(f) c->f = (const float*)lo_arr(b, #f, 2, NULL)
#define AI(f) c->f = (const int32_t*)lo_arr(b, #f, 1, NULL)
#define AD(f) c->f = (const double*)lo_arr(b, #f, 3, NULL)
    AF(rs_blackfilt);
    AF(amp_filter); AF(ATH_l); AF(ATH_s); AF(ATH_psfb21); AF(ATH_psfb12); AF(ATH_cb_l); AF(ATH_cb_s); AF(eql_w);
    AF(pow43); AF(adj43); AF(ipow20); AF(pow20); AF(longfact); AF(shortfact); AF(rnumlines_l); AF(bo_l_weight);
    AF(bo_s_weight); AF(s3_ll); AF(s3_ss); AF(window); AF(window_s); AF(mld_l); AF(mld_s);
    AI(sfb_l); AI(sfb_s); AI(psfb21); AI(psfb12); AI(bv_scf); AI(numlines_l); AI(numlines_s); AI(bo_l); AI(bm_l);
    AI(bo_s); AI(bm_s); AI(s3ind); AI(s3ind_s); AI(fft_rv_tbl); AI(mdct_order); AI(pretab); AI(scfsi_band);
    AI(slen1_n); AI(slen2_n); AI(slen1_tab); AI(slen2_tab); AI(scale_short); AI(scale_long); AI(huf_tbl_noESC);
    AI(ht_xlen); AI(ht_linmax); AI(ht_off); AI(ht_code); AI(ht_hlen); AI(largetbl); AI(table23); AI(table56);
    AI(t32l); AI(t33l);
    c->version_bytes = (const int32_t*)lo_arr(b, "version_bytes", 1, &c->n_version_bytes);
    AD(fht_twiddle); AD(fht_costab); AD(enwindow); AD(mdct_win); AD(ma_tab); AD(ma_table1); AD(ma_table2);
    AD(ma_table3); AD(hpf_fircoef);
    if (c->quant_comp != 9 || c->quant_comp_short != 9 || (c->version != 1 && c->version != 0) || c->mode_gr != (c->version == 1 ? 2 : 1) || c->error_protection) {
        fprintf(stderr, "lame_oracle: configuration outside the supported envelope\n");
        return -1;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* bitstream formatter: one self-contained, byte-aligned frame          */
/* ------------------------------------------------------------------ */

typedef struct { uint8_t* p; int bitpos; } lo_bw;

static void lo_put(lo_bw* w, uint32_t val, int nbits) {
    while (nbits > 0) {
        int byte = w->bitpos >> 3, free = 8 - (w->bitpos & 7);
        int k = nbits < free ? nbits : free;
        nbits -= k;
        w->p[byte] |= (uint8_t)(((val >> nbits) & ((1u << k) - 1u)) << (free - k));
        w->bitpos += k;
    }
}

static int lo_huffmancode(const lo_cfg* c, lo_bw* w, int tableindex, int start, int end, const lo_gr* gi) {
    int bits = 0, i;
    if (0 == tableindex) return 0;
    const int32_t* hl = c->ht_hlen + c->ht_off[tableindex];
    const int32_t* hc = c->ht_code + c->ht_off[tableindex];
    for (i = start; i < end; i += 2) {
        int cbits = 0, xbits = 0;
        int linbits = c->ht_xlen[tableindex], xlen = c->ht_xlen[tableindex];
        uint32_t ext = 0;
        int x1 = gi->l3_enc[i], x2 = gi->l3_enc[i + 1];
        if (x1 != 0) { if (D(gi->xr[i]) < 0) ext++; cbits--; }
        if (tableindex > 15) {
            if (x1 > 14) { ext |= (uint32_t)(x1 - 15) << 1; xbits = linbits; x1 = 15; }
            if (x2 > 14) { ext <<= linbits; ext |= (uint32_t)(x2 - 15); xbits += linbits; x2 = 15; }
            xlen = 16;
        }
        if (x2 != 0) { ext <<= 1; if (D(gi->xr[i + 1]) < 0) ext++; cbits--; }
        x1 = x1 * xlen + x2;
        xbits -= cbits;
        cbits += hl[x1];
        lo_put(w, (uint32_t)hc[x1], cbits);
        lo_put(w, ext, xbits);
        bits += cbits + xbits;
    }
    return bits;
}

static int lo_count1_code(const lo_cfg* c, lo_bw* w, const lo_gr* gi) {
    const int t = gi->count1table_select + 32;
    const int32_t* hl = c->ht_hlen + c->ht_off[t];
    const int32_t* hc = c->ht_code + c->ht_off[t];
    int i, bits = 0, ix = gi->big_values;
    for (i = (gi->count1 - gi->big_values) / 4; i > 0; --i) {
        int huffbits = 0, p = 0, v;
        v = gi->l3_enc[ix + 0]; if (v != 0) { p += 8; if (D(gi->xr[ix + 0]) < 0) huffbits++; }
        v = gi->l3_enc[ix + 1]; if (v != 0) { p += 4; huffbits *= 2; if (D(gi->xr[ix + 1]) < 0) huffbits++; }
        v = gi->l3_enc[ix + 2]; if (v != 0) { p += 2; huffbits *= 2; if (D(gi->xr[ix + 2]) < 0) huffbits++; }
        v = gi->l3_enc[ix + 3]; if (v != 0) { p++; huffbits *= 2; if (D(gi->xr[ix + 3]) < 0) huffbits++; }
        ix += 4;
        lo_put(w, (uint32_t)(huffbits + hc[p]), hl[p]);
        bits += hl[p];
    }
    return bits;
}

static void lo_drain(const lo_cfg* c, lo_bw* w, int remaining) {
    static const uint32_t lame[4] = {0x4c, 0x41, 0x4d, 0x45};
    int i;
    for (i = 0; i < 4; i++) if (remaining >= 8) { lo_put(w, lame[i], 8); remaining -= 8; }
    if (remaining >= 32)
        for (i = 0; i < c->n_version_bytes && remaining >= 8; ++i) { remaining -= 8; lo_put(w, (uint32_t)c->version_bytes[i], 8); }
    for (; remaining >= 1; remaining -= 1) lo_put(w, 0, 1);   /* ancillary_flag stays 0 with the reservoir disabled */
}

/* returns frame size in bytes.  main_bits != NULL (bit reservoir): header + side info (with main_data_begin) and the main data are
 * written back to back into `out` (zeroed by the caller, large enough for a frame plus the reservoir), nothing is drained, and
 * *main_bits receives the length of the main data: the caller feeds both parts to the stream writer */
static int lo_format_frame(lo_enc* e, uint8_t* out, int* main_bits) {
    const lo_cfg* c = &e->c;
    const int frame_bits = lo_frame_bits(e);
    const uint32_t mdb = main_bits ? (uint32_t)js_toint32(e->main_data_begin) : 0;       /* writeheader shifts the number: ToInt32 */
    lo_bw w;
    int gr, ch, sfb, band;
    if (!main_bits) memset(out, 0, (size_t)frame_bits / 8);
    w.p = out; w.bitpos = 0;
    /* header (BitStream.js:267-270: MPEG-2.5 rates carry the 0xffe sync) */
    lo_put(&w, c->out_samplerate < 16000 ? 0xffe : 0xfff, 12);
    lo_put(&w, (uint32_t)c->version, 1);
    lo_put(&w, 4 - 3, 2);
    lo_put(&w, (!c->error_protection ? 1 : 0), 1);
    lo_put(&w, (uint32_t)c->bitrate_index, 4);
    lo_put(&w, (uint32_t)c->samplerate_index, 2);
    lo_put(&w, (uint32_t)e->padding, 1);
    lo_put(&w, (uint32_t)c->extension, 1);
    lo_put(&w, (uint32_t)c->mode, 2);
    lo_put(&w, (uint32_t)e->mode_ext, 2);               /* BitStream.js:279; 0 unless joint stereo chose M/S for this frame */
    lo_put(&w, (uint32_t)c->copyright, 1);
    lo_put(&w, (uint32_t)c->original, 1);
    lo_put(&w, (uint32_t)c->emphasis, 2);
    if (c->version == 1) {
    /* side info (MPEG-1) */
        lo_put(&w, mdb, 9);                                 /* main_data_begin */
        lo_put(&w, 0, c->channels_out == 2 ? 3 : 5);        /* private bits */
        for (ch = 0; ch < c->channels_out; ch++)
            for (band = 0; band < 4; band++) lo_put(&w, (uint32_t)e->scfsi[ch][band], 1);
        for (gr = 0; gr < 2; gr++)
            for (ch = 0; ch < c->channels_out; ch++) {
                lo_gr* gi = &e->tt[gr][ch];
                lo_put(&w, (uint32_t)(gi->part2_3_length + gi->part2_length), 12);
                lo_put(&w, (uint32_t)(gi->big_values / 2), 9);
                lo_put(&w, (uint32_t)gi->global_gain, 8);
                lo_put(&w, (uint32_t)gi->scalefac_compress, 4);
                if (gi->block_type != NORM_TYPE) {
                    lo_put(&w, 1, 1);
                    lo_put(&w, (uint32_t)gi->block_type, 2);
                    lo_put(&w, (uint32_t)gi->mixed_block_flag, 1);
                    if (gi->table_select[0] == 14) gi->table_select[0] = 16;
                    lo_put(&w, (uint32_t)gi->table_select[0], 5);
                    if (gi->table_select[1] == 14) gi->table_select[1] = 16;
                    lo_put(&w, (uint32_t)gi->table_select[1], 5);
                    lo_put(&w, (uint32_t)gi->subblock_gain[0], 3);
                    lo_put(&w, (uint32_t)gi->subblock_gain[1], 3);
                    lo_put(&w, (uint32_t)gi->subblock_gain[2], 3);
                } else {
                    lo_put(&w, 0, 1);
                    if (gi->table_select[0] == 14) gi->table_select[0] = 16;
                    lo_put(&w, (uint32_t)gi->table_select[0], 5);
                    if (gi->table_select[1] == 14) gi->table_select[1] = 16;
                    lo_put(&w, (uint32_t)gi->table_select[1], 5);
                    if (gi->table_select[2] == 14) gi->table_select[2] = 16;
                    lo_put(&w, (uint32_t)gi->table_select[2], 5);
                    lo_put(&w, (uint32_t)gi->region0_count, 4);
                    lo_put(&w, (uint32_t)gi->region1_count, 3);
                }
                lo_put(&w, (uint32_t)gi->preflag, 1);
                lo_put(&w, (uint32_t)gi->scalefac_scale, 1);
                lo_put(&w, (uint32_t)gi->count1table_select, 1);
            }
        /* main data */
        for (gr = 0; gr < 2; gr++)
            for (ch = 0; ch < c->channels_out; ch++) {
                const lo_gr* gi = &e->tt[gr][ch];
                const int slen1 = c->slen1_tab[gi->scalefac_compress], slen2 = c->slen2_tab[gi->scalefac_compress];
                for (sfb = 0; sfb < gi->sfbdivide; sfb++) {
                    if (gi->scalefac[sfb] == -1) continue;
                    lo_put(&w, (uint32_t)gi->scalefac[sfb], slen1);
                }
                for (; sfb < gi->sfbmax; sfb++) {
                    if (gi->scalefac[sfb] == -1) continue;
                    lo_put(&w, (uint32_t)gi->scalefac[sfb], slen2);
                }
                if (gi->block_type == SHORT_TYPE) {
                    int r1 = 3 * c->sfb_s[3];
                    if (r1 > gi->big_values) r1 = gi->big_values;
                    lo_huffmancode(c, &w, gi->table_select[0], 0, r1, gi);
                    lo_huffmancode(c, &w, gi->table_select[1], r1, gi->big_values, gi);
                } else {
                    int bigv = gi->big_values, i = gi->region0_count + 1, r1, r2;
                    r1 = c->sfb_l[i];
                    i += gi->region1_count + 1;
                    r2 = c->sfb_l[i];
                    if (r1 > bigv) r1 = bigv;
                    if (r2 > bigv) r2 = bigv;
                    lo_huffmancode(c, &w, gi->table_select[0], 0, r1, gi);
                    lo_huffmancode(c, &w, gi->table_select[1], r1, r2, gi);
                    lo_huffmancode(c, &w, gi->table_select[2], r2, bigv, gi);
                }
                lo_count1_code(c, &w, gi);
            }
    } else {
        /* MPEG-2 / 2.5 LSF: one granule, 8-bit main_data_begin, 9-bit scalefac_compress, no scfsi/preflag
         * (BitStream.js:352-405), scalefactors by partition with slen[] (BitStream.js:645-686) */
        extern const int lo_nr_of_sfb_block[6][3][4];
        lo_put(&w, mdb, 8);                                 /* main_data_begin */
        lo_put(&w, 0, c->channels_out);                     /* private bits */
        for (ch = 0; ch < c->channels_out; ch++) {
            lo_gr* gi = &e->tt[0][ch];
            lo_put(&w, (uint32_t)(gi->part2_3_length + gi->part2_length), 12);
            lo_put(&w, (uint32_t)(gi->big_values / 2), 9);
            lo_put(&w, (uint32_t)gi->global_gain, 8);
            lo_put(&w, (uint32_t)gi->scalefac_compress, 9);
            if (gi->block_type != NORM_TYPE) {
                lo_put(&w, 1, 1);
                lo_put(&w, (uint32_t)gi->block_type, 2);
                lo_put(&w, (uint32_t)gi->mixed_block_flag, 1);
                if (gi->table_select[0] == 14) gi->table_select[0] = 16;
                lo_put(&w, (uint32_t)gi->table_select[0], 5);
                if (gi->table_select[1] == 14) gi->table_select[1] = 16;
                lo_put(&w, (uint32_t)gi->table_select[1], 5);
                lo_put(&w, (uint32_t)gi->subblock_gain[0], 3);
                lo_put(&w, (uint32_t)gi->subblock_gain[1], 3);
                lo_put(&w, (uint32_t)gi->subblock_gain[2], 3);
            } else {
                lo_put(&w, 0, 1);
                if (gi->table_select[0] == 14) gi->table_select[0] = 16;
                lo_put(&w, (uint32_t)gi->table_select[0], 5);
                if (gi->table_select[1] == 14) gi->table_select[1] = 16;
                lo_put(&w, (uint32_t)gi->table_select[1], 5);
                if (gi->table_select[2] == 14) gi->table_select[2] = 16;
                lo_put(&w, (uint32_t)gi->table_select[2], 5);
                lo_put(&w, (uint32_t)gi->region0_count, 4);
                lo_put(&w, (uint32_t)gi->region1_count, 3);
            }
            lo_put(&w, (uint32_t)gi->scalefac_scale, 1);
            lo_put(&w, (uint32_t)gi->count1table_select, 1);
        }
        for (ch = 0; ch < c->channels_out; ch++) {
            const lo_gr* gi = &e->tt[0][ch];
            const int* pt = lo_nr_of_sfb_block[gi->sfb_part_tab][gi->sfb_part_row];
            int part, i;
            const int dbg_p0 = w.bitpos;
            sfb = 0;
            if (gi->block_type == SHORT_TYPE) {
                for (part = 0; part < 4; part++) {
                    const int sfbs = pt[part] / 3, slen = gi->slen[part];
                    for (i = 0; i < sfbs; i++, sfb++) {
                        lo_put(&w, (uint32_t)(gi->scalefac[sfb * 3 + 0] > 0 ? gi->scalefac[sfb * 3 + 0] : 0), slen);
                        lo_put(&w, (uint32_t)(gi->scalefac[sfb * 3 + 1] > 0 ? gi->scalefac[sfb * 3 + 1] : 0), slen);
                        lo_put(&w, (uint32_t)(gi->scalefac[sfb * 3 + 2] > 0 ? gi->scalefac[sfb * 3 + 2] : 0), slen);
                    }
                }
                {
                    int r1 = 3 * c->sfb_s[3];
                    if (r1 > gi->big_values) r1 = gi->big_values;
                    lo_huffmancode(c, &w, gi->table_select[0], 0, r1, gi);
                    lo_huffmancode(c, &w, gi->table_select[1], r1, gi->big_values, gi);
                }
            } else {
                for (part = 0; part < 4; part++) {
                    const int sfbs = pt[part], slen = gi->slen[part];
                    for (i = 0; i < sfbs; i++, sfb++) lo_put(&w, (uint32_t)(gi->scalefac[sfb] > 0 ? gi->scalefac[sfb] : 0), slen);
                }
                {
                    int bigv = gi->big_values, i2 = gi->region0_count + 1, r1, r2;
                    r1 = c->sfb_l[i2];
                    i2 += gi->region1_count + 1;
                    r2 = c->sfb_l[i2];
                    if (r1 > bigv) r1 = bigv;
                    if (r2 > bigv) r2 = bigv;
                    lo_huffmancode(c, &w, gi->table_select[0], 0, r1, gi);
                    lo_huffmancode(c, &w, gi->table_select[1], r1, r2, gi);
                    lo_huffmancode(c, &w, gi->table_select[2], r2, bigv, gi);
                }
            }
            lo_count1_code(c, &w, gi);
            if (getenv("LO_DEBUG") && w.bitpos - dbg_p0 != gi->part2_3_length + gi->part2_length)
                fprintf(stderr, "frame %ld ch %d: wrote %d, part2 %d part2_3 %d bt %d slen %d %d %d %d tab %d row %d sfc %d\n", e->frame_num, ch, w.bitpos - dbg_p0, gi->part2_length, gi->part2_3_length, gi->block_type, gi->slen[0], gi->slen[1], gi->slen[2], gi->slen[3], gi->sfb_part_tab, gi->sfb_part_row, gi->scalefac_compress);
        }
    }
    if (main_bits) { *main_bits = w.bitpos - 8 * c->sideinfo_len; return 0; }
    lo_drain(c, &w, e->resvDrain_post);
    if (w.bitpos != frame_bits) {
        fprintf(stderr, "lame_oracle: frame %ld wrote %d bits, expected %d\n", e->frame_num, w.bitpos, frame_bits);
        abort();
    }
    return frame_bits / 8;
}

/* ------------------------------------------------------------------ */
/* bit reservoir: the continuous stream writer (BitStream.js:100-215)   */
/* ------------------------------------------------------------------ */
/* Main data is written where the previous frame's ended; a frame's header + side info is inserted when the stream reaches the
 * frame's nominal start (write_timing), which may be in the middle of -- or after -- its own main data. */
#define LO_MAX_HEADER_BUF 256
typedef struct lo_stream {
    uint8_t buf[16384];                         /* bytes since the last copy-out */
    long idx, totbit; int bit_idx;              /* bufByteIdx (-1: empty), totbit, bufBitIdx */
    struct { long write_timing; uint8_t b[40]; } header[LO_MAX_HEADER_BUF];
    int h_ptr, w_ptr;
} lo_stream;

static void lo_putbits2(lo_enc* e, uint32_t val, int j) {
    lo_stream* s = e->bs;
    while (j > 0) {
        int k;
        if (s->bit_idx == 0) {
            s->bit_idx = 8;
            s->idx++;
            if (s->header[s->w_ptr].write_timing == s->totbit) {
                memcpy(s->buf + s->idx, s->header[s->w_ptr].b, (size_t)e->c.sideinfo_len);
                s->idx += e->c.sideinfo_len;
                s->totbit += e->c.sideinfo_len * 8;
                s->w_ptr = (s->w_ptr + 1) & (LO_MAX_HEADER_BUF - 1);
            }
            if ((size_t)s->idx >= sizeof s->buf) { fprintf(stderr, "lame_oracle: stream buffer overflow\n"); abort(); }
            s->buf[s->idx] = 0;
        }
        k = j < s->bit_idx ? j : s->bit_idx;
        j -= k;
        s->bit_idx -= k;
        s->buf[s->idx] |= (uint8_t)((val >> j) << s->bit_idx);
        s->totbit += k;
    }
}
static void lo_drain_into_ancillary(lo_enc* e, int remaining) {       /* BitStream.js:175-211 */
    static const uint32_t lame[4] = {0x4c, 0x41, 0x4d, 0x45};
    const lo_cfg* c = &e->c;
    int i;
    for (i = 0; i < 4; i++) if (remaining >= 8) { lo_putbits2(e, lame[i], 8); remaining -= 8; }
    if (remaining >= 32)
        for (i = 0; i < c->n_version_bytes && remaining >= 8; ++i) { remaining -= 8; lo_putbits2(e, (uint32_t)c->version_bytes[i], 8); }
    for (; remaining >= 1; remaining -= 1) {
        lo_putbits2(e, (uint32_t)e->ancillary_flag, 1);
        e->ancillary_flag ^= (!c->disable_reservoir ? 1 : 0);
    }
}
static long lo_copy_out(lo_enc* e, uint8_t* out, size_t cap) {         /* BitStream.copy_buffer: everything written so far */
    lo_stream* s = e->bs;
    const long n = s->idx + 1;
    if (n <= 0) return 0;
    if ((size_t)n > cap) return -1;
    memcpy(out, s->buf, (size_t)n);
    s->idx = -1; s->bit_idx = 0;
    return n;
}
/* format_bitstream (BitStream.js:836-900) */
static void lo_format_bitstream_resv(lo_enc* e) {
    const lo_cfg* c = &e->c;
    lo_stream* s = e->bs;
    static uint8_t tmp[4096];
    const int bitsPerFrame = lo_frame_bits(e);
    int main_bits = 0, i, old, bits;
    lo_drain_into_ancillary(e, e->resvDrain_pre);
    memset(tmp, 0, sizeof tmp);
    lo_format_frame(e, tmp, &main_bits);
    /* encodeSideInfo2: the header joins the queue, the next one is due one frame later */
    old = s->h_ptr;
    memcpy(s->header[old].b, tmp, (size_t)c->sideinfo_len);
    s->h_ptr = (old + 1) & (LO_MAX_HEADER_BUF - 1);
    s->header[s->h_ptr].write_timing = s->header[old].write_timing + bitsPerFrame;
    /* writeMainData */
    for (i = 0; i + 8 <= main_bits; i += 8) lo_putbits2(e, tmp[c->sideinfo_len + (i >> 3)], 8);
    if (i < main_bits) lo_putbits2(e, (uint32_t)(tmp[c->sideinfo_len + (i >> 3)] >> (8 - (main_bits - i))), main_bits - i);
    lo_drain_into_ancillary(e, e->resvDrain_post);
    bits = 8 * c->sideinfo_len + main_bits + e->resvDrain_post;
    e->main_data_begin += D(bitsPerFrame - bits) / 8;
    if (e->main_data_begin * 8 != D(e->ResvSize)) {
        fprintf(stderr, "lame_oracle: bit reservoir error (main_data_begin %g, ResvSize %d) -- the reference prints its 'fatal error' text here\n", e->main_data_begin, e->ResvSize);
        abort();
    }
    if (s->totbit % 8 != 0) { fprintf(stderr, "lame_oracle: stream not byte aligned after a frame\n"); abort(); }
}
/* flush_bitstream (BitStream.js:710-780): pad the stream with ancillary data up to the end of the last frame */
static void lo_flush_bitstream(lo_enc* e) {
    lo_stream* s = e->bs;
    int last_ptr = s->h_ptr - 1, first_ptr = s->w_ptr, remaining_headers;
    long flushbits;
    if (last_ptr == -1) last_ptr = LO_MAX_HEADER_BUF - 1;
    flushbits = s->header[last_ptr].write_timing - s->totbit;
    if (flushbits >= 0) {
        remaining_headers = 1 + last_ptr - first_ptr;
        if (last_ptr < first_ptr) remaining_headers = 1 + last_ptr - first_ptr + LO_MAX_HEADER_BUF;
        flushbits -= (long)remaining_headers * 8 * e->c.sideinfo_len;
    }
    flushbits += lo_frame_bits(e);
    if (flushbits < 0) return;
    lo_drain_into_ancillary(e, (int)flushbits);
    e->ResvSize = 0;
    e->main_data_begin = 0;
}

/* ------------------------------------------------------------------ */
/* frame + stream drivers                                              */
/* ------------------------------------------------------------------ */

static int lo_encode_frame(lo_enc* e, uint8_t* out) {
    const lo_cfg* c = &e->c;
    lo_ratio masking[2][2], masking_MS[2][2];
    double pe[2][2] = {{0, 0}, {0, 0}}, pe_MS[2][2] = {{0, 0}, {0, 0}}, ms_ener_ratio[2] = {.5, .5};
    float tot_ener[2][4];
    int gr, ch, n;
    const float* inbuf[2] = {e->mfbuf[0], e->mfbuf[1]};

    if (!e->frame_init_done) {
        /* prime the filterbank with [1152 zeros | first samples] and SHORT block types */
        static float prime[2][286 + 1152 + 576];
        int i, j;
        e->frame_init_done = 1;
        for (i = 0, j = 0; i < 286 + 576 * (1 + c->mode_gr); ++i) {
            if (i < 576 * c->mode_gr) { prime[0][i] = 0; prime[1][i] = 0; }
            else { prime[0][i] = inbuf[0][j]; if (c->channels_out == 2) prime[1][i] = inbuf[1][j]; ++j; }
        }
        for (gr = 0; gr < c->mode_gr; gr++)
            for (ch = 0; ch < c->channels_out; ch++) e->tt[gr][ch].block_type = SHORT_TYPE;
        lo_mdct_sub48(e, prime[0], prime[1]);
    }

    e->padding = 0;
    if ((e->slot_lag -= c->frac_SpF) < 0) { e->slot_lag += c->out_samplerate; e->padding = 1; }

    for (gr = 0; gr < c->mode_gr; gr++) {
        int blocktype[2] = {0, 0};
        const float* bufp[2];
        for (ch = 0; ch < c->channels_out; ch++) bufp[ch] = inbuf[ch] + 576 + gr * 576 - 272;
        if (c->channels_out == 1) bufp[1] = bufp[0];
        lo_psycho_anal(e, bufp, gr, masking, masking_MS, pe[gr], pe_MS[gr], tot_ener[gr], blocktype);
        if (c->mode == 1) {                             /* Encoder.js:482-486 */
            ms_ener_ratio[gr] = D(tot_ener[gr][2]) + D(tot_ener[gr][3]);
            if (ms_ener_ratio[gr] > 0) ms_ener_ratio[gr] = D(tot_ener[gr][3]) / ms_ener_ratio[gr];
        }
        for (ch = 0; ch < c->channels_out; ch++) {
            e->tt[gr][ch].block_type = blocktype[ch];
            e->tt[gr][ch].mixed_block_flag = 0;
        }
    }
    lo_adjust_ATH(e);
    lo_mdct_sub48(e, inbuf[0], inbuf[1]);
    /* M/S or L/R for this frame (Encoder.js:520-561): M/S if its perceptual entropy is not larger and the two channels
     * have the same block type in the first and in the last granule */
    e->mode_ext = 0;
    if (c->mode == 1) {
        double sum_pe_MS = 0., sum_pe_LR = 0.;
        for (gr = 0; gr < c->mode_gr; gr++)
            for (ch = 0; ch < c->channels_out; ch++) { sum_pe_MS += pe_MS[gr][ch]; sum_pe_LR += pe[gr][ch]; }
        if (sum_pe_MS <= 1.00 * sum_pe_LR) {
            const lo_gr *gi0 = e->tt[0], *gi1 = e->tt[c->mode_gr - 1];
            if (gi0[0].block_type == gi0[1].block_type && gi1[0].block_type == gi1[1].block_type) e->mode_ext = 2;
        }
    }
    if (e->tap) {
        e->tap->ath_adjust = e->ATH_adjust;
        for (gr = 0; gr < c->mode_gr; gr++)
            for (ch = 0; ch < c->channels_out; ch++) {
                memcpy(e->tap->xr[gr][ch], e->tt[gr][ch].xr, sizeof e->tt[gr][ch].xr);
                e->tap->block_type[gr][ch] = e->tt[gr][ch].block_type;
                e->tap->ratio[gr][ch] = (e->mode_ext == 2) ? masking_MS[gr][ch] : masking[gr][ch];
                e->tap->pe[gr][ch] = pe[gr][ch]; e->tap->pe_MS[gr][ch] = pe_MS[gr][ch];
            }
        e->tap->mode_ext = e->mode_ext;
        e->tap->ms_ener_ratio[0] = ms_ener_ratio[0]; e->tap->ms_ener_ratio[1] = ms_ener_ratio[1];
    }
    if (!c->disable_reservoir) {
        /* Encoder.js:600-626: the frame's perceptual entropies are scaled by a 19-frame FIR of their sums (dead with the reservoir off) */
        static const double fircoef[9] = {-0.0207887 * 5, -0.0378413 * 5, -0.0432472 * 5, -0.031183 * 5, 7.79609e-18 * 5, 0.0467745 * 5,
                                          0.10091 * 5, 0.151365 * 5, 0.187098 * 5};
        double (*pe_use)[2] = (e->mode_ext == 2) ? pe_MS : pe;
        double f = 0.0;
        int i;
        for (i = 0; i < 18; i++) e->pefirbuf[i] = e->pefirbuf[i + 1];
        for (gr = 0; gr < c->mode_gr; gr++)
            for (ch = 0; ch < c->channels_out; ch++) f += pe_use[gr][ch];
        e->pefirbuf[18] = (float)f;
        f = e->pefirbuf[9];
        for (i = 0; i < 9; i++) f += (D(e->pefirbuf[i]) + D(e->pefirbuf[18 - i])) * fircoef[i];
        f = (670 * 5 * c->mode_gr * c->channels_out) / f;
        for (gr = 0; gr < c->mode_gr; gr++)
            for (ch = 0; ch < c->channels_out; ch++) pe_use[gr][ch] *= f;
        lo_iteration_loop_resv(e, pe_use, (e->mode_ext == 2) ? masking_MS : masking, ms_ener_ratio);
        lo_format_bitstream_resv(e);
        n = (int)lo_copy_out(e, out, 1 << 20);
        e->frame_num++;
        return n;
    }
    lo_iteration_loop(e, (e->mode_ext == 2) ? masking_MS : masking, ms_ener_ratio);
    n = lo_format_frame(e, out, NULL);
    e->frame_num++;
    return n;
}

lo_enc* lo_create(const void* blob, size_t nbytes) {
    lo_enc* e = (lo_enc*)calloc(1, sizeof(lo_enc));
    int i, j, sb;
    if (!e) return NULL;
    if (lo_load_cfg(&e->c, blob, nbytes) != 0) { free(e); return NULL; }
    e->mf_size = 576 - 48;                 /* ENCDELAY - MDCTDELAY zeros in front */
    e->mf_samples_to_encode = 576 + 1152;  /* ENCDELAY + POSTDELAY */
    e->masking_lower = 1;                  /* Lame.js:175 */
    e->bs = (lo_stream*)calloc(1, sizeof(lo_stream));
    e->bs->idx = -1;
    for (i = 0; i < 19; i++) e->pefirbuf[i] = (float)(700 * e->c.mode_gr * e->c.channels_out);     /* Lame.js:1120 */
    e->OldValue[0] = e->OldValue[1] = 180;
    e->CurrentStep[0] = e->CurrentStep[1] = 4;
    e->slot_lag = e->c.frac_SpF;
    e->ATH_adjust = 0.01;
    e->ATH_adjustLimit = 1.0;
    e->blocktype_old[0] = e->blocktype_old[1] = NORM_TYPE;
    for (i = 0; i < 4; ++i) {
        for (j = 0; j < CBANDS; ++j) e->nb_s1[i][j] = e->nb_s2[i][j] = 1.0f;
        for (sb = 0; sb < SBMAX_l; sb++) { e->en[i].l[sb] = 1e20f; e->thm[i].l[sb] = 1e20f; }
        for (j = 0; j < 3; ++j)
            for (sb = 0; sb < SBMAX_s; sb++) { e->en[i].s[sb][j] = 1e20f; e->thm[i].s[sb][j] = 1e20f; }
        e->lastAttacks[i] = 0;
        for (j = 0; j < CBANDS; ++j) e->nb_1[i][j] = e->nb_2[i][j] = 1e20f;
        for (j = 0; j < 9; j++) e->last_en_subshort[i][j] = 10.f;
    }
    return e;
}

void lo_destroy(lo_enc* e) {
    if (!e) return;
    free(e->c.blob_copy);
    free(e->tap);
    free(e->bs);
    free(e);
}

int lo_frame_bytes_max(const lo_enc* e) {
    const lo_cfg* c = &e->c;
    return js_toint32(D((c->version + 1) * 72000 * c->brate) / c->out_samplerate + 1);
}

/* fill_buffer_resample (Lame.js:1719-1843) for one channel.  `len` and the buffer position stay doubles as in the
 * reference; with the integer ratios of the envelope they only ever hold integers (filter_l = 32, filter_l / 2 = 16). */
static int lo_fill_buffer_resample(lo_enc* e, float* outbuf, int desired_len, const float* inbuf, double len, double* num_used, int ch) {
    const lo_cfg* c = &e->c;
    const int bpc = c->rs_bpc, filter_l = c->rs_filter_l, BLACKSIZE = filter_l + 1;
    const double half = D(filter_l) / 2;
    float* inbuf_old = e->inbuf_old[ch];
    int i, j = 0, k;
    for (k = 0; k < desired_len; k++) {
        double time0 = k * c->resample_ratio, offset, xvalue = 0.;
        int joff;
        j = js_toint32(floor(time0 - e->itime[ch]));
        if ((filter_l + j - half) >= len) break;
        offset = (time0 - e->itime[ch] - (j + .5 * (filter_l % 2)));
        joff = js_toint32(floor((offset * 2 * bpc) + bpc + .5));
        for (i = 0; i <= filter_l; ++i) {
            const int j2 = js_toint32(i + j - half);
            const float y = (j2 < 0) ? inbuf_old[BLACKSIZE + j2] : inbuf[j2];
            xvalue += D(y) * D(c->rs_blackfilt[joff * BLACKSIZE + i]);
        }
        outbuf[k] = (float)xvalue;
    }
    *num_used = (len < filter_l + j - half) ? len : filter_l + j - half;
    e->itime[ch] += *num_used - k * c->resample_ratio;
    if (*num_used != floor(*num_used)) { fprintf(stderr, "lame_oracle: fractional resampler position (outside the envelope)\n"); abort(); }
    {
        const int nu = (int)*num_used;
        if (nu >= BLACKSIZE) {
            for (i = 0; i < BLACKSIZE; i++) inbuf_old[i] = inbuf[nu + i - BLACKSIZE];
        } else {
            const int n_shift = BLACKSIZE - nu;
            for (i = 0; i < n_shift; ++i) inbuf_old[i] = inbuf_old[i + nu];
            for (j = 0; i < BLACKSIZE; ++i, ++j) inbuf_old[i] = inbuf[j];
        }
    }
    return k;
}

static long lo_feed(lo_enc* e, const float* l, const float* r, size_t nsamples_in, uint8_t* out, size_t cap) {
    const lo_cfg* c = &e->c;
    const int framesize = 576 * c->mode_gr;
    const int mf_needed = 1024 + framesize - 272;   /* calcNeeded: max(BLKSIZE + framesize - FFTOFFSET, 512 + framesize - 32) */
    const int resample = (c->resample_ratio < .9999) || (c->resample_ratio > 1.0001);      /* Lame.js:1849 */
    long written = 0;
    size_t pos = 0;
    double nsamples = (double)nsamples_in;
    int ch, i;
    while (nsamples > 0) {
        int n_out;
        double n_in;
        if (resample) {
            n_out = 0; n_in = 0;
            for (ch = 0; ch < c->channels_out; ch++)
                n_out = lo_fill_buffer_resample(e, e->mfbuf[ch] + e->mf_size, framesize, (ch == 0 ? l : r) + pos, nsamples, &n_in, ch);
        } else {
            n_out = nsamples < framesize ? (int)nsamples : framesize;                       /* fill_buffer: at most one frame per pass */
            n_in = n_out;
            for (i = 0; i < n_out; i++) {
                e->mfbuf[0][e->mf_size + i] = l[pos + i];
                if (c->channels_out == 2) e->mfbuf[1][e->mf_size + i] = r[pos + i];
            }
        }
        nsamples -= n_in; pos += (size_t)n_in;
        e->mf_size += n_out;
        if (e->mf_samples_to_encode < 1) e->mf_samples_to_encode = 576 + 1152;
        e->mf_samples_to_encode += n_out;
        if (e->mf_size >= mf_needed) {
            if ((size_t)written + (size_t)lo_frame_bytes_max(e) > cap) return -1;
            written += lo_encode_frame(e, out + written);
            e->mf_size -= framesize;
            e->mf_samples_to_encode -= framesize;
            for (ch = 0; ch < c->channels_out; ch++)
                memmove(e->mfbuf[ch], e->mfbuf[ch] + framesize, (size_t)e->mf_size * sizeof(float));
        }
    }
    return written;
}

long lo_encode(lo_enc* e, const int16_t* left, const int16_t* right, size_t nsamples, uint8_t* out, size_t cap) {
    const lo_cfg* c = &e->c;
    if (!e) return -3;
    if (nsamples == 0) return 0;
    if (c->channels_out == 1 || !right) right = left;
    /* Int16 -> Float32 copy, then in-place scaling by gfp.scale (Lame.js:1506-1560) */
    float* bl = (float*)malloc(sizeof(float) * nsamples * 2);
    float* br = bl + nsamples;
    const int do_scale = !(c->scale == 0) && !(c->scale == 1.0);
    for (size_t i = 0; i < nsamples; i++) {
        bl[i] = (float)left[i];
        br[i] = (float)right[i];
        if (do_scale) {
            bl[i] = (float)(D(bl[i]) * c->scale);
            if (c->channels_out == 2) br[i] = (float)(D(br[i]) * c->scale);
        }
    }
    long n = lo_feed(e, bl, br, nsamples, out, cap);
    free(bl);
    return n;
}

long lo_flush(lo_enc* e, uint8_t* out, size_t cap) {                 /* Lame.js:1381-1443; doubles where the reference's numbers can be fractional */
    static const float zeros[1152];
    long written = 0;
    double samples_to_encode, end_padding, frames_left;
    if (e->mf_samples_to_encode < 1) return 0;
    const int framesize = 576 * e->c.mode_gr, mf_needed = 1024 + framesize - 272;
    samples_to_encode = e->mf_samples_to_encode - 1152;     /* POSTDELAY */
    if (e->c.in_samplerate != e->c.out_samplerate) samples_to_encode += 16. * e->c.out_samplerate / e->c.in_samplerate;
    end_padding = framesize - fmod(samples_to_encode, framesize);
    if (end_padding < 576) end_padding += framesize;
    frames_left = (samples_to_encode + end_padding) / framesize;
    while (frames_left > 0) {
        double bunch = mf_needed - e->mf_size;
        long fn = e->frame_num, n;
        bunch *= e->c.in_samplerate;
        bunch /= e->c.out_samplerate;
        if (bunch > 1152) bunch = 1152;
        if (bunch < 1) bunch = 1;
        if (bunch != floor(bunch)) { fprintf(stderr, "lame_oracle: fractional flush bunch (outside the envelope)\n"); abort(); }
        n = lo_feed(e, zeros, zeros, (size_t)bunch, out + written, cap - (size_t)written);
        if (n < 0) return n;
        written += n;
        frames_left -= (fn != e->frame_num) ? 1 : 0;
    }
    e->mf_samples_to_encode = 0;
    if (!e->c.disable_reservoir) {                           /* Lame.js:1445-1456: flush_bitstream + copy_buffer */
        long n;
        lo_flush_bitstream(e);
        n = lo_copy_out(e, out + written, cap - (size_t)written);
        if (n < 0) return n;
        written += n;
    }
    return written;
}

void lo_enable_tap(lo_enc* e) { if (!e->tap) e->tap = (lo_tap*)calloc(1, sizeof(lo_tap)); }
const void* lo_get_tap(const lo_enc* e) { return e->tap; }
size_t lo_tap_size(void) { return sizeof(lo_tap); }

/* test hook mirroring lhip_debug_math: the oracle's V8-exact math on n doubles */
void lo_math(int op, const double* in, double* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        const double x = in[i];
        switch (op) {
            case 0: out[i] = v8_log10(x); break;
            case 1: out[i] = v8_pow(10.0, x); break;
            case 2: out[i] = sqrt(x); break;
            case 3: out[i] = 1.0 / x; break;
            case 4: out[i] = (double)(float)x; break;
            case 5: out[i] = (double)js_toint32(x); break;
            case 7: out[i] = v8_log10(x); break;      /* the device's branch-free variant must equal plain log10 on its domain */
            default: out[i] = x / 3.0 + x * 0.1; break;
        }
    }
}
