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
    SB_STRIDE = 18 * 32
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
    int32_t pad_;
    int64_t frame_num0;     // absolute index of the first frame of this batch
};

// All slot arrays of a launch (device pointers).  C = channels_out.
struct Workspace {
    int nstreams, nframes_total, nfslots, ngslots;   // slots include one carry slot per stream
    int64_t pcm_plane;          // floats per channel plane
    float* pcm;                 // [C][pcm_plane] scaled f32 samples, per stream segment = tail + new
    const int32_t* fslot_stream;   // [nfslots]  frame slot -> stream index
    const int32_t* gslot_stream;   // [ngslots]  granule slot -> stream index
    // psy phase A outputs, per (gslot, ch)
    float* peaks;               // [ngslots][C][PK_STRIDE]
    float* loud;                // [ngslots][C]   loudness computed by that psy call
    float* eb_l;                // [ngslots][C][64]
    int32_t* mask_idx;          // [ngslots][C][64]
    float* eb_s;                // [ngslots][C][3][64]
    float* ecb_s;               // [ngslots][C][3][64]   (carry slot: sblock 1 = nb_s2, sblock 2 = nb_s1)
    // scan outputs
    int32_t* att_raw;           // [ngslots][C]  bit j = raw ns_attacks[j]
    int32_t* uselong;           // [ngslots][C]  coupled uselongblock flag of that call
    int32_t* ul_tmp;            // [ngslots][C]  scratch: lastAttacks before publication
    int32_t* last_attack;       // [ngslots][C]  lastAttacks after that call
    int32_t* tent;              // [ngslots][C]  blocktype_old after that call (tentative type)
    int32_t* prev_short;        // [ngslots][C]  blocktype_old == SHORT as seen by that call's thresholds
    int32_t* blocktype;         // [ngslots][C]  final block type of the MDCT granule in that slot
    double* ath_adjust;         // [nfslots]     ATH.adjust after the frame in that slot (slot 0: carried in)
    double* ath_limit;          // [nfslots]
    // psy phase B output
    float* E;                   // [ngslots][C][E_STRIDE]  thresholds computed by that psy call
    // filterbank
    float* sb;                  // [ngslots][C][18][32]
    float* xr;                  // [ngslots][C][576]
    // quantizer
    GrSide* side;               // [nframes_total][2][C]
    int16_t* l3;                // [nframes_total][2][C][576]  signed quantized spectrum
    int32_t* seed;              // [nfslots][C][2]  OldValue, CurrentStep after the frame in that slot
    int32_t* seed_flag;         // [nframes_total] 1 = frame must be (re)quantized with the chain-implied seed
    int32_t* nflagged;          // [0] frames to re-quantize, [1] frames the memo-only validation could not decide
    int32_t* slow_list;         // [nfslots] frame slots of the latter
    int32_t* work_ctr;          // [8] frame-slot dispensers of the persistent quantization kernels (zeroed per launch)
    uint8_t* out;               // output MP3 bytes
    int32_t* frame_bytes;       // [nframes_total]
    unsigned long long* prof;   // [32] phase-profiling accumulators (profiling builds only)
    int32_t mode_gr;            // granules per frame (2: MPEG-1, 1: MPEG-2/2.5 LSF)
    int32_t spec_start, spec_step;   // seed assumed by the speculative pass (Quantize.js reset values 180 / 4)
};

}  // namespace lhip
