/*
 * TEST TOOL: two devices inside one process through the C ABI of include/lamejs_hip.h -- what the GPU tier's
 * test_gpu_two_devices_round_robin_and_concurrent_batches does on a box with two GPUs, runnable in the CPU tier against the host
 * simulation with LHIP_HOSTSIM_DEVICES=2 (and built with -fsanitize=thread / address: the per-device contexts, their mutexes and
 * the library's process-wide state under two host threads).
 *   1. lhip_set_devices(0b11): streams created with the default device alternate between the two devices (lhip_stream_device);
 *      an explicit device outside a mask is refused
 *   2. two host threads, one per device, each encode their own material three times -- every second repetition through the chunked
 *      host path (LAMEJS_HIP_HOST_CHUNK_FRAMES small) -- while the other thread does the same on the other context
 *   3. every repetition must equal the single-threaded result of the same material; the bytes are written out for the caller to
 *      compare with the oracle
 * usage: two_devices tables.bin pcm0.s16 pcm1.s16 out0.mp3 out1.mp3     (two-channel interleaved s16le, 44.1 kHz, 128 kbps)
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/lamejs_hip.h"

static void* slurp(const char* p, size_t* n) {
    FILE* f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
    fseek(f, 0, SEEK_END); *n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    void* b = malloc(*n ? *n : 1); if (fread(b, 1, *n, f) != *n) { perror("read"); exit(2); } fclose(f); return b;
}
typedef struct { int dev; const void* blob; size_t nblob; int16_t *l, *r; size_t ns; uint8_t* out; size_t nout; int reps; int fail; } Work;

static size_t encode_once(Work* w, uint8_t* out, size_t cap, size_t call) {
    lhip_config cfg = {2, 44100, 128, w->dev};
    lhip_stream* s = NULL;
    if (lhip_create(&cfg, w->blob, w->nblob, &s) != 0) { fprintf(stderr, "create(dev %d): %s\n", w->dev, lhip_last_error()); w->fail = 1; return 0; }
    if (lhip_stream_device(s) != w->dev) { fprintf(stderr, "stream asked for device %d sits on %d\n", w->dev, lhip_stream_device(s)); w->fail = 1; }
    size_t off = 0;
    for (size_t p = 0; p < w->ns; p += call) {
        const size_t n = w->ns - p < call ? w->ns - p : call;
        const int64_t want = lhip_encode_output_bytes(s, n);
        const int64_t k = lhip_encode(s, w->l + p, w->r + p, n, out + off, cap - off);
        if (k < 0 || k != want) { fprintf(stderr, "encode(dev %d): %lld (promised %lld) %s\n", w->dev, (long long)k, (long long)want, lhip_last_error()); w->fail = 1; break; }
        off += (size_t)k;
    }
    const int64_t k = lhip_flush(s, out + off, cap - off);
    if (k < 0) { fprintf(stderr, "flush(dev %d): %s\n", w->dev, lhip_last_error()); w->fail = 1; } else off += (size_t)k;
    lhip_destroy(s);
    return off;
}
static void* thread_main(void* p) {
    Work* w = (Work*)p;
    const size_t cap = (w->ns / 1152 + 8) * 1500 + 16384;
    uint8_t* tmp = (uint8_t*)malloc(cap);
    for (int rep = 0; rep < w->reps && !w->fail; rep++) {
        /* odd repetitions: one call for everything (the chunked, overlapped host path); even ones: calls of 20 000 samples */
        const size_t n = encode_once(w, tmp, cap, (rep & 1) ? w->ns : 20000);
        if (n != w->nout || memcmp(tmp, w->out, n) != 0) { fprintf(stderr, "device %d, repetition %d: bytes differ from the single-threaded run\n", w->dev, rep); w->fail = 1; }
    }
    free(tmp);
    return NULL;
}
int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s tables.bin pcm0.s16 pcm1.s16 out0.mp3 out1.mp3\n", argv[0]); return 2; }
    size_t nb; void* blob = slurp(argv[1], &nb);
    if (lhip_device_count() < 2) { fprintf(stderr, "needs two devices (host simulation: LHIP_HOSTSIM_DEVICES=2)\n"); return 3; }
    /* 1. placement */
    if (lhip_set_devices(3) != 2) { fprintf(stderr, "lhip_set_devices(3): %s\n", lhip_last_error()); return 1; }
    int devs[6];
    for (int i = 0; i < 6; i++) {
        lhip_config cfg = {2, 44100, 128, -1}; lhip_stream* s = NULL;
        if (lhip_create(&cfg, blob, nb, &s) != 0) { fprintf(stderr, "create: %s\n", lhip_last_error()); return 1; }
        devs[i] = lhip_stream_device(s); lhip_destroy(s);
    }
    for (int i = 0; i < 6; i++) if (devs[i] != ((devs[0] + i) & 1)) { fprintf(stderr, "placement not round-robin: %d %d %d %d %d %d\n", devs[0], devs[1], devs[2], devs[3], devs[4], devs[5]); return 1; }
    if (lhip_set_devices(2) != 1) return 1;
    { lhip_config cfg = {2, 44100, 128, 0}; lhip_stream* s = NULL; if (lhip_create(&cfg, blob, nb, &s) == 0) { fprintf(stderr, "a device outside the mask was accepted\n"); return 1; } }
    { lhip_config cfg = {2, 44100, 128, -1}; lhip_stream* s = NULL; if (lhip_create(&cfg, blob, nb, &s) != 0 || lhip_stream_device(s) != 1) { fprintf(stderr, "mask 0b10 did not place on device 1\n"); return 1; } lhip_destroy(s); }
    if (lhip_set_devices(0) != 2) return 1;
    /* 2. single-threaded results, then the two threads */
    Work w[2];
    for (int i = 0; i < 2; i++) {
        size_t np; int16_t* pcm = (int16_t*)slurp(argv[2 + i], &np);
        memset(&w[i], 0, sizeof w[i]);
        w[i].dev = i; w[i].blob = blob; w[i].nblob = nb; w[i].ns = np / 4; w[i].reps = 4;
        w[i].l = (int16_t*)malloc(w[i].ns * 2 + 2); w[i].r = (int16_t*)malloc(w[i].ns * 2 + 2);
        for (size_t k = 0; k < w[i].ns; k++) { w[i].l[k] = pcm[2 * k]; w[i].r[k] = pcm[2 * k + 1]; }
        free(pcm);
        const size_t cap = (w[i].ns / 1152 + 8) * 1500 + 16384;
        w[i].out = (uint8_t*)malloc(cap);
        w[i].nout = encode_once(&w[i], w[i].out, cap, 7777);
        if (w[i].fail) return 1;
        FILE* f = fopen(argv[4 + i], "wb"); fwrite(w[i].out, 1, w[i].nout, f); fclose(f);
    }
    pthread_t th[2];
    for (int i = 0; i < 2; i++) pthread_create(&th[i], NULL, thread_main, &w[i]);
    for (int i = 0; i < 2; i++) pthread_join(th[i], NULL);
    if (w[0].fail || w[1].fail) return 1;
    printf("OK two devices: placement round-robin, %d concurrent repetitions per device identical to the single-threaded bytes (%zu, %zu)\n", w[0].reps, w[0].nout, w[1].nout);
    return 0;
}
