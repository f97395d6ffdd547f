/*
 * TEST TOOL: drives the C ABI of include/lamejs_hip.h (links against liblamejs_hip.so or the
 * host-simulation build).  usage: abi_cli tables.bin in.pcm out.mp3 channels samplerate kbps [chunk]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../../include/lamejs_hip.h"
static void* slurp(const char* p, size_t* n) {
    FILE* f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
    fseek(f, 0, SEEK_END); *n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    void* b = malloc(*n ? *n : 1); if (fread(b, 1, *n, f) != *n) { perror("read"); exit(2); } fclose(f); return b;
}
int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: %s tables.bin in.pcm out.mp3 channels samplerate kbps [chunk]\n", argv[0]); return 2; }
    size_t nb, np; void* blob = slurp(argv[1], &nb); int16_t* pcm = (int16_t*)slurp(argv[2], &np);
    int ch = atoi(argv[4]); size_t chunk = argc > 7 ? (size_t)atol(argv[7]) : 0;
    size_t ns = np / 2 / (size_t)ch; if (!chunk) chunk = ns;
    int16_t *l = malloc(ns * 2 + 2), *r = malloc(ns * 2 + 2);
    for (size_t i = 0; i < ns; i++) { l[i] = pcm[i * ch]; r[i] = pcm[i * ch + (ch - 1)]; }
    lhip_config cfg = {ch, atoi(argv[5]), atoi(argv[6]), -1};
    lhip_stream* s = NULL;
    if (lhip_create(&cfg, blob, nb, &s) != 0) { fprintf(stderr, "lhip_create: %s\n", lhip_last_error()); return 1; }
    size_t cap = (ns / 1152 + 8) * 1500 + 16384, off = 0; uint8_t* out = malloc(cap);
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    long long rep = 0;
    for (size_t p = 0; p < ns; p += chunk) {
        size_t n = ns - p < chunk ? ns - p : chunk;
        int64_t w = lhip_encode(s, l + p, ch == 2 ? r + p : NULL, n, out + off, cap - off);
        if (w < 0) { fprintf(stderr, "encode error %lld: %s\n", (long long)w, lhip_last_error()); return 1; }
        off += (size_t)w;
        int64_t fr, rp, it; lhip_last_batch_stats(&fr, &rp, &it); rep += rp;
    }
    int64_t w = lhip_flush(s, out + off, cap - off);
    if (w < 0) { fprintf(stderr, "flush error %lld: %s\n", (long long)w, lhip_last_error()); return 1; }
    off += (size_t)w;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    FILE* f = fopen(argv[3], "wb"); fwrite(out, 1, off, f); fclose(f);
    fprintf(stderr, "%s: %zu samples/ch -> %zu bytes in %.3f s (%.0f frames/s), seed-repaired frames: %lld\n", lhip_version(), ns, off, sec, (double)ns / 1152.0 / sec, rep);
    lhip_destroy(s);
    return 0;
}
