/*
 * TEST INFRASTRUCTURE: command-line driver for the CPU oracle.
 * usage: lo_cli <tables.bin> <in.pcm (s16le, interleaved if 2ch)> <out.mp3> <channels> [chunk_samples]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "lame_oracle.h"

static void* slurp(const char* p, size_t* n) {
    FILE* f = fopen(p, "rb");
    if (!f) { perror(p); exit(2); }
    fseek(f, 0, SEEK_END); *n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    void* b = malloc(*n ? *n : 1);
    if (fread(b, 1, *n, f) != *n) { perror("read"); exit(2); }
    fclose(f);
    return b;
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s tables.bin in.pcm out.mp3 channels [chunk]\n", argv[0]); return 2; }
    size_t nb, np;
    void* blob = slurp(argv[1], &nb);
    int16_t* pcm = (int16_t*)slurp(argv[2], &np);
    int ch = atoi(argv[4]);
    size_t chunk = argc > 5 ? (size_t)atol(argv[5]) : 1152;
    size_t ns = np / 2 / (size_t)ch;
    int16_t *l = (int16_t*)malloc(ns * 2 + 2), *r = (int16_t*)malloc(ns * 2 + 2);
    for (size_t i = 0; i < ns; i++) { l[i] = pcm[i * ch]; r[i] = pcm[i * ch + (ch - 1)]; }
    lo_enc* e = lo_create(blob, nb);
    if (!e) { fprintf(stderr, "lo_create failed\n"); return 1; }
    size_t cap = (ns / 1152 + 8) * 1500 + 16384, off = 0;
    uint8_t* out = (uint8_t*)malloc(cap);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (size_t p = 0; p < ns; p += chunk) {
        size_t n = ns - p < chunk ? ns - p : chunk;
        long w = lo_encode(e, l + p, r + p, n, out + off, cap - off);
        if (w < 0) { fprintf(stderr, "encode error %ld\n", w); return 1; }
        off += (size_t)w;
    }
    long w = lo_flush(e, out + off, cap - off);
    if (w < 0) { fprintf(stderr, "flush error %ld\n", w); return 1; }
    off += (size_t)w;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    FILE* f = fopen(argv[3], "wb");
    fwrite(out, 1, off, f);
    fclose(f);
    fprintf(stderr, "oracle: %zu samples/ch -> %zu bytes in %.3f s (%.0f frames/s)\n", ns, off, sec, (double)ns / 1152.0 / sec);
    lo_destroy(e);
    return 0;
}
