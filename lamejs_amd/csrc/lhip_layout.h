// Workspace layout in HBM for one batch launch.
//
// A batch encodes, for each of S streams, F_s whole frames.  Streams are independent; inside a
// stream, state that the reference carries from granule to granule is kept in *slot* arrays:
// slot 0 of a stream holds the state carried in from the previous call, slot 1 + q belongs to
// the q-th granule (psy call / MDCT granule) of this batch.  Kernels always read "slot - 1",
// so the first granule of a batch and every later granule follow the same code path
// (parallel phase -> tiny scan -> parallel phase).
#pragma once
#include "lhip_defs.h"

namespace lhip {

enum {
    PK_STRIDE = 12,                 // per (slot, ch): 9 sub-block peaks (+ pad)
    EBL_STRIDE = 64, EBS_STRIDE = 3 * 64,
    E_STRIDE = 2 * (SBMAX_l + 3 * SBMAX_s),   // en.l en.s thm.l thm.s = 122 floats
    E_EN_L = 0, E_EN_S = SBMAX_l, E_THM_L = SBMAX_l + 3 * SBMAX_s, E_THM_S = 2 * SBMAX_l + 3 * SBMAX_s,
    SB_STRIDE = 18 * 32,
    FHT_STRIDE = 1024 + 3 * 256
};

// Per-stream descriptor for one launch (device resident array, one entry per stream).
struct StreamDesc {
    int32_t nframes;        // frames encoded in this batch (may be 0)
    int32_t fslot0;         // first frame slot of this stream (slot fslot0 = carry, frames at fslot0+1+k)
    int32_t gslot0;         // first granule slot (slot gslot0 = carry, granules at gslot0+1+q), q < 2*nframes
    int32_t out_slot0;      // first output-frame index (frames of all streams are numbered consecutively)
    int64_t pcm_off;        // float offset of this stream's sample segment inside the pcm plane
    int64_t out_off;        // byte offset of this stream's first frame in the output buffer
    int32_t seg_len;        // valid samples in the segment (tail + new)
    int32_t first_call;     // 1 if the stream has not produced a frame yet (filterbank priming rules)
    int32_t slot_lag;       // padding accumulator before the first frame of this batch
    int32_t flush;          // bit reservoir: the stream ends with this call -- pad its bitstream to the end of the last frame (flush_bitstream)
    int64_t frame_num0;     // absolute index of the first frame of this batch
};

enum { RS_TAPS = 33 };

// Bit reservoir (extension: Tables::disable_reservoir == 0).  What the reference carries from frame to frame once the reservoir is
// in use (Reservoir.js, BitStream.js:100-215, Encoder.js:600-626, PsyModel.js:1036-1038, 1300-1318).  A frame's bit budget, and through
// `pcfact` even its masking thresholds, depend on the bits all earlier frames spent: a stream is then a serial chain of frames, so
// a launch encodes ONE frame per stream and reads / updates this record directly (parallelism is across streams only).
enum { RESV_HQ = 16, RESV_HDR = 40 };            // pending headers: at most ceil(ResvMax / smallest frame) + 1 wait at any time -- 7 for the
                                                 // smallest frames of the envelope (48 bytes: 12 kHz, 8 kbps); the reference's ring holds 256
struct ResvState {
    int32_t ResvSize, ResvMax;                   // after the last frame's ResvFrameEnd / ResvFrameBegin
    int32_t ancillary_flag, h_ptr, w_ptr, last_frame_bits;      // last_frame_bits: getframebits of the last frame (its padding), for the flush
    double main_data_begin;                      // a double: the reference's arithmetic leaves fractions in it (Reservoir.js:281-286)
    int64_t totbit;                              // stream position
    int64_t timing[RESV_HQ];                     // stream position at which header[i] is due (the nominal start of its frame)
    uint8_t header[RESV_HQ][RESV_HDR];
    float pefirbuf[19];                          // NsPsy.js:30
    int32_t pad2_;
};
// what the quantization of a frame decides about the reservoir; committed to ResvState by the bit packer (which runs once per frame,
// whereas a frame may be quantized again by the seed-chain repair)
struct FrameResv { int32_t ResvSize, ResvMax, drain_pre, drain_post; double main_data_begin; float pefir_new; int32_t pad_; };                       // BLACKSIZE of the reference for an integer ratio (filter_l = 32)

struct StreamState {                // psy arrays hold 4 channels: L, R and -- joint stereo only -- mid, side
    float pcm_tail[2][MF_NEEDED];
    float sb[2][SB_STRIDE];
    float E[4][E_STRIDE];
    float ecb_s[4][EBS_STRIDE];
    float peaks[4][PK_STRIDE];
    float loud[2];
    float tot_ener[4];              // joint stereo: total FFT energy of the last psy call (LameInternalFlags.js:271)
    int32_t last_attack[4], tent[2];
    int32_t last_bt[2];             // block type of the last granule encoded (-1: none yet); joint stereo: selects the masking_lower the next
                                    // frame's perceptual entropy is computed with (gfc.masking_lower is left by the previous frame's last channel)
    double ath_adjust, ath_limit;
    int32_t seed[2][2];
    float rs_old[2][RS_TAPS - 1];   // resampling streams: the last 32 (scaled) input samples
    float nb1[4][EBL_STRIDE], nb2[4][EBL_STRIDE];   // long-block spreading results of the last two psy calls (pre-echo control; live with the reservoir)
    ResvState rv;
};

struct StreamIO {          // per stream, per launch (device array parallel to StreamDesc)
    StreamState* state;
    const int16_t* src[2]; // new samples (device addresses)
    uint8_t* out;          // where this stream's frames go (device address)
    int32_t n_new, mf_size;    // samples appended to the encoder's buffer by this call / already buffered
    int32_t n_in, rs_p0;       // resampling streams: input samples of this call; input position (relative to this call's
                               // first sample, >= -32) of tap 0 of the first new output sample
};

// Where the samples of a stream's segment (carried tail ++ this call's new samples) come from.  Without resampling nothing is
// materialised: a sample below `mf` is the carried tail (scaled f32, StreamState), the others are converted from the caller's
// Int16 on the fly -- Lame.js:1554-1560 is one multiply, `(float)((double)(float)s16 * scale)` -- so the psychoacoustics and the
// filterbank read 2 bytes per sample instead of a 4-byte copy that another kernel had to write first.  Resampling
// configurations (integer-ratio FIR in front, g_prep) read the f32 plane that kernel fills.
struct PcmSrc {
    const float* plane;      // != nullptr: materialised segment (resampling configurations)
    const float* tail;       // carried samples [0, mf)
    const int16_t* src;      // new samples [mf, ...)
    int mf, do_scale;
    double scale;
};
LHIP_DEV float pcm_at(const PcmSrc& P, int s) {
    if (P.plane) return P.plane[s];
    if (s < P.mf) return P.tail[s];
    float v = (float)P.src[s - P.mf];
    if (P.do_scale) v = (float)((double)v * P.scale);
    return v;
}

// All slot arrays of a launch (device pointers).  C = channels_out.
struct Workspace {
    int nstreams, nframes_total, nfslots, ngslots;   // slots include one carry slot per stream
    int64_t pcm_plane;          // floats per channel plane
    float* pcm;                 // [C][pcm_plane] scaled f32 samples, per stream segment = tail + new
    const int32_t* fslot_stream;   // [nfslots]  frame slot -> stream index
    const int32_t* gslot_stream;   // [ngslots]  granule slot -> stream index
    // psy phase A outputs, per (gslot, psy channel); Cp = Tables::psy_channels (C, or 4 in joint stereo: L, R, mid, side)
    float* peaks;               // [ngslots][Cp][PK_STRIDE]
    float* loud;                // [ngslots][C]   loudness computed by that psy call (L / R only)
    float* eb_l;                // [ngslots][Cp][64]
    int32_t* mask_idx;          // [ngslots][Cp][64]
    float* eb_s;                // [ngslots][Cp][3][64]
    float* ecb_s;               // [ngslots][Cp][3][64]   (carry slot: sblock 1 = nb_s2, sblock 2 = nb_s1)
    // joint stereo only: what the mid / side analysis needs of the L / R one (PsyModel.js:258-273, 1113-1121)
    float* fht;                 // [ngslots][2][FHT_STRIDE]  FHT outputs: 1024 long + 3 x 256 short
    float* hpf;                 // [ngslots][2][576]         high-passed samples
    float* tot_ener;            // [ngslots][4]              total FFT energy of that psy call
    // scan outputs
    int32_t* att_raw;           // [ngslots][Cp] bit j = raw ns_attacks[j]
    int32_t* att_clean;         // [ngslots][Cp] bit j = ns_attacks[j] after the clean-up (PsyModel.js:1183-1196): short-block pre-echo control
    float* nb1;                 // [ngslots][Cp][64] long-block spreading result (ecb) of that psy call, and
    float* nb2;                 // [ngslots][Cp][64] of the call before it (nb_1 / nb_2 of PsyModel.js:1300-1318)
    int32_t* uselong;           // [ngslots][C]  coupled uselongblock flag of that call
    int32_t* ul_tmp;            // [ngslots][Cp] scratch: lastAttacks before publication
    int32_t* last_attack;       // [ngslots][Cp] lastAttacks after that call
    int32_t* tent;              // [ngslots][C]  blocktype_old after that call (tentative type)
    int32_t* prev_short;        // [ngslots][C]  blocktype_old == SHORT as seen by that call's thresholds
    int32_t* blocktype;         // [ngslots][C]  final block type of the MDCT granule in that slot
    double* ath_adjust;         // [nfslots]     ATH.adjust after the frame in that slot (slot 0: carried in)
    double* ath_limit;          // [nfslots]
    // psy phase B output
    float* E;                   // [ngslots][Cp][E_STRIDE] thresholds computed by that psy call
    // filterbank
    float* sb;                  // [ngslots][C][18][32]
    float* xr;                  // [ngslots][C][576]
    // quantizer
    GrSide* side;               // [nframes_total][2][C]
    int16_t* l3;                // [nframes_total][2][C][576]  signed quantized spectrum
    int32_t* seed;              // [nfslots][C][2]  OldValue, CurrentStep after the frame in that slot
    uint32_t* vdig;             // [VD_WORDS][vdig_n] validation digest per granule-channel (index = (frame * 2 + gr) * C + ch), word-major
    int64_t vdig_n;             // nframes_total * 2 * C
    int32_t* seed_flag;         // [nframes_total] 1 = frame must be (re)quantized with the chain-implied seed
    int32_t* reval;             // [nframes_total] i > 0: repair pass i - 1 re-quantized the frame's predecessor -- validation pass i re-checks it
    int32_t* nflagged;          // [0] frames to re-quantize, [1] frames the memo-only validation could not decide
    int32_t* slow_list;         // [nfslots] frame slots of the latter
    int32_t* work_ctr;          // [8] frame-slot dispensers of the persistent quantization kernels (zeroed per launch)
    uint8_t* out;               // output MP3 bytes
    int32_t* frame_bytes;       // [nframes_total]
    FrameResv* fr;              // [nframes_total] bit reservoir: the frame's reservoir decisions (quant -> bits)
    int32_t* out_bytes;         // [nstreams] bit reservoir: bytes this launch appended to the stream's output
    const StreamIO* io;         // [nstreams] (device copy): the reservoir kernels read and update StreamState::rv through it
    unsigned long long* prof;   // [32] phase-profiling accumulators (profiling builds only)
    int32_t mode_gr;            // granules per frame (2: MPEG-1, 1: MPEG-2/2.5 LSF)
    int32_t spec_start, spec_step;   // seed assumed by the speculative pass (Quantize.js reset values 180 / 4)
};

// Validation digest of a granule-channel (quantization -> g_validate_fast): what the seed-chain check reads, compact and word-major so that
// one thread per frame reads it coalesced.  VD_HEAD: gain | seed start << 8 | seed step << 16 | active << 24; VD_TARG: target bits | memo
// entries << 24; VD_STATE: the conditionally assigned fields at the end of the bin search (GrSide::bs_state); then the first VD_ENT memo
// entries as (gain << 24 | bits, assignments) pairs.
#ifndef LHIP_VD_ENT
#define LHIP_VD_ENT 6      /* memo entries kept in the digest; 10 and 12 were measured in round 4 (profiles/r04_pass7_*): no effect on the validation's time */
#endif
enum { VD_HEAD = 0, VD_TARG = 1, VD_STATE = 2, VD_TAB = 3, VD_ENT = LHIP_VD_ENT, VD_WORDS = VD_TAB + 2 * VD_ENT };
LHIP_DEV uint32_t vd_head(int active, int start, int step, int gain) { return (uint32_t)(gain & 255) | ((uint32_t)(start & 255) << 8) | ((uint32_t)(step & 255) << 16) | ((uint32_t)(active != 0) << 24); }
LHIP_DEV int vd_active(uint32_t h) { return (int)((h >> 24) & 1u); }
LHIP_DEV int vd_gain(uint32_t h) { return (int)(h & 255u); }
LHIP_DEV int vd_start(uint32_t h) { return (int)((h >> 8) & 255u); }
LHIP_DEV int vd_step(uint32_t h) { return (int)((h >> 16) & 255u); }

LHIP_DEV PcmSrc pcm_source(const Tables& T, const Workspace& W, const StreamDesc& sd, const StreamIO& io, int ch) {
    PcmSrc P;
    P.plane = T.rs_ratio != 1 ? W.pcm + (int64_t)ch * W.pcm_plane + sd.pcm_off : nullptr;
    P.tail = io.state->pcm_tail[ch]; P.src = ch ? io.src[1] : io.src[0]; P.mf = io.mf_size;       // (not io.src[ch]: a dynamic index into a by-value copy of the record makes the copy a private array)
    P.do_scale = !(T.scale == 0.0) && !(T.scale == 1.0); P.scale = T.scale;
    return P;
}

}  // namespace lhip
