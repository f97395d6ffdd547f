/*
 * lamejs_hip.h -- C ABI of the MI355X-native MP3 frame-encode path (liblamejs_hip.so).
 *
 * This is the drop-in boundary for the hot path of zhuker/lamejs: everything that
 * `Mp3Encoder.encodeBuffer()/flush()` does after parameter resolution.  The entry points are
 * what the reference's JavaScript front end would bind through a thin N-API addon
 * (lamejs_amd/js/addon/lhip_napi.c); plain pointers and sizes only.
 *
 * Reference interfaces replaced (file:line in /root/reference):
 *   lhip_create   <- new Mp3Encoder(channels, samplerate, kbps)      src/js/index.js:66-111
 *                    (lame_init + lame_init_params; the resolved tables arrive as one blob built by
 *                     the host-side JavaScript lamejs_amd/js/tables.js with the host's own Math.*)
 *   lhip_encode   <- Mp3Encoder.encodeBuffer(left, right)            src/js/index.js:117-130
 *                    -> Lame.lame_encode_buffer                      src/js/Lame.js:1490-1514
 *                    -> lame_encode_buffer_sample                    src/js/Lame.js:1527-1667
 *                    -> Encoder.lame_encode_mp3_frame (hot path)     src/js/Encoder.js:388-659
 *   lhip_flush    <- Mp3Encoder.flush() -> Lame.lame_encode_flush    src/js/index.js:132-135, Lame.js:1381-1488
 *   lhip_destroy  <- (garbage collection of the encoder object)
 *
 * Channel mode.  The reference's Mp3Encoder hard-codes MPEGMode.STEREO for two channels (index.js:105).  As an extension the
 * blob may carry mode = 1 (tables.js buildBlob(..., { jointStereo: true })): the stream is then encoded in the reference core's
 * joint-stereo mode -- per frame mid/side or left/right (Encoder.js:520-561) -- byte for byte what the reference's own modules
 * produce when asked for MPEGMode.JOINT_STEREO.  Nothing in the signatures below changes.
 * Bit reservoir.  Likewise disable_reservoir = 0 in the blob ({ reservoir: true }; index.js:108 hard-codes it off): the frames of a
 * stream then depend on each other (budget and masking), so the library encodes one frame per stream per launch -- batches of many
 * streams are what uses the GPU -- the byte count of a call is data-dependent, and every call synchronises (sync = 0 is ignored).
 *
 * Semantics preserved: any chunking of the same sample stream yields the same bytes; a call
 * returns the bytes of all whole frames completed by that call (possibly 0); errors are negative
 * return codes mirroring the reference (-1 output buffer too small, -3 bad handle, -4 internal/device
 * error).  A stream handle is not thread-safe (same as the reference); distinct handles are independent.
 * The library never retains caller pointers past the call.
 *
 * There is NO CPU fallback: if no HIP device is usable every entry point fails with -4 and
 * lhip_last_error() explains why.
 */
#ifndef LAMEJS_HIP_H
#define LAMEJS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lhip_stream lhip_stream;

typedef struct lhip_config {
    int32_t channels;     /* 1 or 2 (as passed to Mp3Encoder) */
    int32_t samplerate;   /* Hz */
    int32_t kbps;         /* CBR bitrate */
    int32_t device;       /* HIP device ordinal; -1 = current device */
} lhip_config;

#define LHIP_ERR_BUFFER_TOO_SMALL (-1)
#define LHIP_ERR_BAD_HANDLE       (-3)
#define LHIP_ERR_INTERNAL         (-4)

/* number of usable HIP devices (0 if none / runtime unavailable) */
int lhip_device_count(void);

/* Restrict the devices the library may place streams on: bit d of `mask` = HIP device d.  A later
 * lhip_create with cfg.device == -1 then deals streams round-robin over the allowed devices (independent streams are the
 * path's multi-GPU axis); an explicit cfg.device outside the mask is refused.  mask == 0 restores the default (every
 * device; device -1 = the calling thread's current HIP device).  Returns the number of allowed devices or <0. */
int lhip_set_devices(uint64_t mask);
/* the HIP device a stream was placed on (>= 0), or LHIP_ERR_BAD_HANDLE */
int lhip_stream_device(const lhip_stream* s);
/* What tells two GPUs apart: PCI bus id ("0000:a7:00.0") and UUID (32 hex digits) of HIP device `device` (-1 = current), as NUL-terminated strings of at most
 * `cap` bytes each (40 is enough).  bench.py prints them per rank, so that a multi-GPU line shows N ranks on N distinct devices.  Returns 0 or <0. */
int lhip_device_identity(int device, char* pci_bus_id, char* uuid_hex, size_t cap);

/* Frame-range sharding of ONE stream (extension; SURVEY.md 8e, second mode).  A long stream can be cut at frame boundaries and the
 * pieces encoded side by side (other GPUs, other processes): the state at a cut is SPECULATED -- lhip_seek puts a fresh stream at an
 * input position with the samples in front of it, a few warm-up frames (output discarded) let the masking history, the
 * filterbank overlap, the attack / block-type chains, the ATH adjustment and the bin-search seeds converge -- and then VERIFIED:
 * lhip_state_get of that stream at the cut must equal, byte for byte, lhip_state_get of the stream that encoded up to the cut; if it
 * does, everything after the cut is what one stream would have produced (64 warm-up frames verified at every cut of every stream tried,
 * material with long silences included; 8 are enough on steady material).  On a miss the true state is transplanted (lhip_state_set)
 * and the piece is encoded again.  bench.py --config shard3 does exactly that over torch.distributed
 * (one range per rank; tests/test_shard_gloo.py runs it with two and three ranks, tests/test_hostsim_parity.py the API itself).
 *   lhip_seek(s, sample_pos, tail_l, tail_r): s fresh; sample_pos a whole number (>= 2) of frames; tail_*: the lhip_seek_tail_samples(s)
 *   input samples in front of sample_pos (host memory).  Not for resampling or bit-reservoir streams. */
/* lhip_state_get returns the blob in canonical form (fields no later launch can read are zeroed), so two streams that stand at the
 * same point compare equal whatever call sizes took them there.  lhip_state_set may be applied to a fresh or to a used stream of the
 * same configuration (it replaces everything the stream carries); like every call on a handle it must not run concurrently with
 * another call on the same handle.  A two-channel lhip_seek needs both tails. */
size_t lhip_state_bytes(const lhip_stream* s);
int lhip_state_get(lhip_stream* s, void* buf, size_t cap);
int lhip_state_set(lhip_stream* s, const void* buf, size_t n);
size_t lhip_seek_tail_samples(const lhip_stream* s);
int lhip_seek(lhip_stream* s, int64_t sample_pos, const int16_t* tail_left, const int16_t* tail_right);

/* Create an encoder stream.  `tables` is the LHTB blob produced by lamejs_amd/js/tables.js for
 * (channels, samplerate, kbps); it is validated against cfg, uploaded to HBM (shared between
 * streams with identical blobs) and may be freed by the caller on return.  Returns 0 or <0. */
int lhip_create(const lhip_config* cfg, const void* tables, size_t tables_bytes, lhip_stream** out);

/* Append nsamples Int16 samples per channel (right may be NULL for mono) and write the bytes of
 * every MP3 frame completed by them to out.  Returns bytes written (>= 0) or a negative code. */
int64_t lhip_encode(lhip_stream* s, const int16_t* left, const int16_t* right, size_t nsamples,
                    uint8_t* out, size_t out_cap);

/* Pad with zeros until all buffered samples are emitted (reference flush rules); a second call
 * returns 0. */
int64_t lhip_flush(lhip_stream* s, uint8_t* out, size_t out_cap);

void lhip_destroy(lhip_stream* s);

/* Upper bound of the bytes lhip_encode can return for nsamples more samples on this stream. */
size_t lhip_max_output_bytes(const lhip_stream* s, size_t nsamples);

/* Exactly the bytes the next lhip_encode(s, ..., nsamples, ...) will return: under CBR without the bit reservoir the frame sizes follow
 * from the sample count and the padding accumulator alone, so a binding can allocate the returned array at its final size and let the
 * library write into it -- no second buffer, no copy (the reference allocates a fresh exact-size Int8Array per call, index.js:129).
 * With the bit reservoir (extension) the count is data-dependent and this returns lhip_max_output_bytes().  < 0: bad handle. */
int64_t lhip_encode_output_bytes(const lhip_stream* s, size_t nsamples);
/* 1: lhip_encode_output_bytes(s, n) is the exact byte count of the next call (CBR without the bit reservoir); 0: it is only an upper bound (bit-reservoir
 * extension: the count is data-dependent) -- a binding then encodes into scratch memory and hands out an exact copy; < 0: bad handle. */
int lhip_output_bytes_is_exact(const lhip_stream* s);

/* Batch extension (BASELINE config 5: many independent streams, one launch): stream i receives
 * nsamples[i] samples from left[i]/right[i] and its frames are written to out[i] (capacity
 * out_cap[i]); written[i] receives the byte count or a negative code.  All streams must share one
 * configuration (same tables blob) and device.  Returns 0 or the first negative code. */
int lhip_encode_batch(lhip_stream* const* streams, size_t nstreams, const int16_t* const* left,
                      const int16_t* const* right, const size_t* nsamples, uint8_t* const* out,
                      const size_t* out_cap, int64_t* written);
int lhip_flush_batch(lhip_stream* const* streams, size_t nstreams, uint8_t* const* out,
                     const size_t* out_cap, int64_t* written);

/* Device-resident variants: the pointers are HBM addresses on the stream's device (e.g. a
 * torch tensor's data_ptr()); nothing crosses PCIe except a few hundred bytes of descriptors.
 * Work is enqueued on the HIP stream set by lhip_set_hip_stream (default: the null stream) and the
 * call returns after enqueueing unless `sync` is non-zero: nothing in the pipeline waits for the host (the bin-search seed chain
 * is validated and repaired by a persistent kernel on the device), so with sync == 0 the output bytes and lhip_last_batch_stats
 * are only valid after the stream has been synchronised (lhip_last_batch_stats does that itself).  The input buffers are read by
 * kernels up to the end of the batch (the samples are converted where they are consumed, there is no staging copy), so with
 * sync == 0 they must stay valid and unchanged until then as well. */
int lhip_encode_batch_device(lhip_stream* const* streams, size_t nstreams, const int16_t* const* d_left,
                             const int16_t* const* d_right, const size_t* nsamples, uint8_t* const* d_out,
                             const size_t* out_cap, int64_t* written, int sync);

/* Use this hipStream_t (passed as void*) for all work of streams on `device` (-1 = current). */
int lhip_set_hip_stream(int device, void* hip_stream);

/* Statistics of the most recent batch on the calling thread: frames encoded, frames that needed
 * the bin-search seed repair pass, repair iterations. */
void lhip_last_batch_stats(int64_t* frames, int64_t* repaired_frames, int64_t* repair_iterations);

/* Debug/test taps (tests only): copy intermediate results of the most recent batch to the host.
 * what: 0 xr [granule][ch][576] f32, 1 blocktype [granule][ch] i32, 2 E [granule][psy ch][122] f32 (psy ch = ch, or L R mid side in joint stereo; thresholds
 * handed to the quantizer for that granule), 3 ath_adjust [frame] f64, 4 side records (struct GrSide).
 * Returns bytes copied or <0. */
int64_t lhip_debug_read(int what, void* dst, size_t cap);

/* Per-kernel timing with HIP events recorded on the launch stream (used by bench.py for the roofline
 * line).  lhip_kernel_timing(1) resets and enables, returns the number of kernels; lhip_kernel_times(i)
 * reports name / accumulated milliseconds / launches of kernel i. */
int lhip_kernel_timing(int enable);
int lhip_kernel_times(int idx, const char** name, double* total_ms, int64_t* launches);

/* Test hook: evaluate the device math used by the path on n doubles.  op: 0 log10, 1 pow(10,x), 2 sqrt,
 * 3 1/x, 4 (double)(float)x, 5 ToInt32, 6 x/3 + x*0.1 (must not fuse), 7 log10 by the branch-free variant for positive
 * normal operands / +inf / NaN, 8 the quantizer's two truncations on records of 21 doubles [istep, xa[5], xb[5], adj_a[5], adj_b[5]]
 * (f32 values) -> [0, floor(x istep) x 10, floor(x istep + adj) x 10], 9 calc_noise's logarithm-free band class: noise_class(x)
 * + 1000 * class from the f64 log10 + 1e6 * class from its Float32 copy (the first must equal both others unless it is -1), 10 calc_noise's
 * division by a Float32 through its reciprocal on records of 2 doubles [a, b] -> [div_by_f32(a, (float)b), a / (float)b] (must be equal bit for bit),
 * 11 mask_add's table index for x >= 1: ma_index16(x) (-1 = take the logarithm) + 1000 * ToInt32(log10(x) * 16). */
int lhip_debug_math(int op, const double* in, double* out, size_t n);

/* Test hook: the seed the speculative quantization pass assumes for the reference's bin-search chain
 * (gfc.OldValue / gfc.CurrentStep, Quantize.js:324-326); default 180 / 4.  A poor seed (e.g. 255 / 1) makes the
 * validation flag frames, which exercises the repair passes; the output must not change. */
int lhip_debug_set_spec_seed(int start, int step);

/* Test hook: with LHIP_ALIAS_DEVICES=n in the environment (2 <= n <= 8) the library presents n devices that are n SEPARATE contexts -- own mutex, own HIP
 * stream, own workspaces, own table uploads -- on physical device 0, so that the multi-device paths (lhip_set_devices' round-robin, host threads batching on
 * two contexts at once) run against real HIP on a box with one GPU.  lhip_debug_release_context(d) gives back the stream the library created for such a
 * context (and the side stream of its ATH scan) once no stream lives on it; returns 0 or <0. */
int lhip_debug_release_context(int device);

const char* lhip_last_error(void);
const char* lhip_version(void);

#ifdef __cplusplus
}
#endif
#endif
