/*
 * TEST INFRASTRUCTURE -- public interface of the CPU oracle (see lame_oracle.c).
 * Mirrors the reference's stream API: create(config blob) / encode(Int16 PCM) / flush.
 */
#ifndef LAME_ORACLE_H
#define LAME_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct lo_enc lo_enc;
/* blob: LHTB table blob produced by lamejs_amd/js/tables.js (copied; caller may free) */
lo_enc* lo_create(const void* blob, size_t nbytes);
void lo_destroy(lo_enc* e);
/* appends whole MP3 frames completed by these samples; returns bytes written, -1 if out too small */
long lo_encode(lo_enc* e, const int16_t* left, const int16_t* right, size_t nsamples, uint8_t* out, size_t cap);
long lo_flush(lo_enc* e, uint8_t* out, size_t cap);
int lo_frame_bytes_max(const lo_enc* e);
/* stage-level taps of the most recent frame (struct lo_tap in lo_common.h) */
void lo_enable_tap(lo_enc* e);
const void* lo_get_tap(const lo_enc* e);
size_t lo_tap_size(void);
#ifdef __cplusplus
}
#endif
#endif
