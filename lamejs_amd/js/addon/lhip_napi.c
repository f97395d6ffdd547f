/*
 * Thin N-API shim between the JavaScript host (lamejs_amd/js/index.js) and the C ABI of
 * include/lamejs_hip.h.  It only marshals typed arrays: no encoding logic lives here.
 * The HIP library is located at run time (dlopen of ../../lib/liblamejs_hip.so relative to this
 * addon, or $LAMEJS_HIP_LIB), so the addon itself builds with plain gcc + node headers:
 *     gcc -O2 -fPIC -shared -I/usr/include/node -I../../../include lhip_napi.c -o lhip_napi.node -ldl
 *
 * JS surface:  create(tablesBlob: Buffer, channels, samplerate, kbps[, device]) -> handle
 *              encode(handle, left: Int16Array, right: Int16Array|null) -> Int8Array
 *              flush(handle) -> Int8Array ;  deviceCount() -> number
 *              encodeBatch(handles[], lefts: Int16Array[], rights: Int16Array[]|null) -> Int8Array[]   (lhip_encode_batch:
 *              flushBatch(handles[]) -> Int8Array[]                                 many independent streams, one launch)
 *              seekTailSamples(handle) -> number ; seek(handle, samplePos, tailLeft, tailRight|null)       (frame-range sharding of
 *              stateGet(handle) -> Uint8Array ; stateSet(handle, Uint8Array)                                one stream: lhip_seek ...)
 */
#define _GNU_SOURCE
#define NAPI_VERSION 6
#include <node_api.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lamejs_hip.h"

static void* g_lib;
static int (*p_device_count)(void);
static int (*p_create)(const lhip_config*, const void*, size_t, lhip_stream**);
static int64_t (*p_encode)(lhip_stream*, const int16_t*, const int16_t*, size_t, uint8_t*, size_t);
static int64_t (*p_flush)(lhip_stream*, uint8_t*, size_t);
static void (*p_destroy)(lhip_stream*);
static size_t (*p_max_out)(const lhip_stream*, size_t);
static int (*p_is_exact)(const lhip_stream*);
static int64_t (*p_out_bytes)(const lhip_stream*, size_t);
static int (*p_encode_batch)(lhip_stream* const*, size_t, const int16_t* const*, const int16_t* const*, const size_t*, uint8_t* const*, const size_t*, int64_t*);
static int (*p_flush_batch)(lhip_stream* const*, size_t, uint8_t* const*, const size_t*, int64_t*);
static const char* (*p_last_error)(void);
static int (*p_set_devices)(uint64_t);
static size_t (*p_state_bytes)(const lhip_stream*);
static int (*p_state_get)(lhip_stream*, void*, size_t);
static int (*p_state_set)(lhip_stream*, const void*, size_t);
static size_t (*p_seek_tail)(const lhip_stream*);
static int (*p_seek)(lhip_stream*, int64_t, const int16_t*, const int16_t*);

static int load_lib(napi_env env) {
    if (g_lib) return 1;
    const char* path = getenv("LAMEJS_HIP_LIB");
    char buf[4096];
    if (!path) {
        Dl_info info;
        if (dladdr((void*)&load_lib, &info) && info.dli_fname) {
            snprintf(buf, sizeof buf, "%s", info.dli_fname);
            char* slash = strrchr(buf, '/');
            if (slash) { *slash = 0; strncat(buf, "/../../lib/liblamejs_hip.so", sizeof buf - strlen(buf) - 1); path = buf; }
        }
    }
    g_lib = path ? dlopen(path, RTLD_NOW | RTLD_LOCAL) : NULL;
    if (!g_lib) { napi_throw_error(env, NULL, "lamejs_amd: cannot load liblamejs_hip.so (no CPU fallback exists)"); return 0; }
#define SYM(v, n) *(void**)(&v) = dlsym(g_lib, n); if (!v) { napi_throw_error(env, NULL, "lamejs_amd: missing symbol " n); return 0; }
    SYM(p_device_count, "lhip_device_count") SYM(p_create, "lhip_create") SYM(p_encode, "lhip_encode") SYM(p_flush, "lhip_flush")
    SYM(p_destroy, "lhip_destroy") SYM(p_max_out, "lhip_max_output_bytes") SYM(p_is_exact, "lhip_output_bytes_is_exact") SYM(p_last_error, "lhip_last_error")
    SYM(p_encode_batch, "lhip_encode_batch") SYM(p_flush_batch, "lhip_flush_batch") SYM(p_set_devices, "lhip_set_devices")
    SYM(p_state_bytes, "lhip_state_bytes") SYM(p_state_get, "lhip_state_get") SYM(p_state_set, "lhip_state_set")
    SYM(p_seek_tail, "lhip_seek_tail_samples") SYM(p_seek, "lhip_seek") SYM(p_out_bytes, "lhip_encode_output_bytes")
#undef SYM
    return 1;
}

static void finalize_stream(napi_env env, void* data, void* hint) { (void)env; (void)hint; if (data && p_destroy) p_destroy((lhip_stream*)data); }

static napi_value js_device_count(napi_env env, napi_callback_info info) {
    (void)info;
    napi_value r;
    if (!load_lib(env)) return NULL;
    napi_create_int32(env, p_device_count(), &r);
    return r;
}

/* setDevices(mask: number) -> number of allowed devices: restricts where later encoders are placed (lhip_set_devices); encoders
 * created without an explicit device are then dealt round-robin over the allowed GPUs */
static napi_value js_set_devices(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    if (!load_lib(env)) return NULL;
    double m = 0;
    if (argc < 1 || napi_get_value_double(env, argv[0], &m) != napi_ok || m < 0 || m >= 18446744073709551616.0) { napi_throw_type_error(env, NULL, "mask must be a non-negative number"); return NULL; }
    const int n = p_set_devices((uint64_t)m);
    if (n < 0) { napi_throw_error(env, NULL, p_last_error()); return NULL; }
    napi_value r; napi_create_int32(env, n, &r);
    return r;
}

static napi_value js_create(napi_env env, napi_callback_info info) {
    size_t argc = 5; napi_value argv[5];
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    if (!load_lib(env)) return NULL;
    void* blob; size_t nblob;
    if (napi_get_buffer_info(env, argv[0], &blob, &nblob) != napi_ok) { napi_throw_type_error(env, NULL, "tables must be a Buffer"); return NULL; }
    lhip_config cfg; cfg.device = -1;
    if (argc < 4 || napi_get_value_int32(env, argv[1], &cfg.channels) != napi_ok || napi_get_value_int32(env, argv[2], &cfg.samplerate) != napi_ok ||
        napi_get_value_int32(env, argv[3], &cfg.kbps) != napi_ok) { napi_throw_type_error(env, NULL, "channels, samplerate and kbps must be numbers"); return NULL; }
    if (argc > 4) { napi_valuetype vt; napi_typeof(env, argv[4], &vt); if (vt == napi_number) napi_get_value_int32(env, argv[4], &cfg.device); }
    lhip_stream* s = NULL;
    if (p_create(&cfg, blob, nblob, &s) != 0) { napi_throw_error(env, NULL, p_last_error()); return NULL; }
    napi_value ext;
    napi_create_external(env, s, finalize_stream, NULL, &ext);
    return ext;
}

static napi_value make_i8(napi_env env, const uint8_t* src, size_t n) {
    napi_value ab, ta; void* data = NULL;
    if (napi_create_arraybuffer(env, n, &data, &ab) != napi_ok || (n && !data)) { napi_throw_error(env, NULL, "could not allocate the output buffer"); return NULL; }
    if (n) memcpy(data, src, n);
    if (napi_create_typedarray(env, napi_int8_array, n, ab, 0, &ta) != napi_ok) { napi_throw_error(env, NULL, "could not create the output Int8Array"); return NULL; }
    return ta;
}

/* The returned Int8Array is allocated FIRST, at its final size (lhip_encode_output_bytes: exact under CBR without the reservoir), and
 * the library writes the frames straight into it: one allocation per call, no intermediate buffer, no copy -- and the caller still owns
 * a fresh array per call, which is the one semantic of the reference's `new Int8Array(mp3buf.subarray(0, _sz))` (index.js:129) to keep.
 * Only when the count is not known beforehand (bit-reservoir extension) or the call fails is the result copied into an exact array. */
static napi_value encode_into_new_array(napi_env env, lhip_stream* s, const int16_t* dl, const int16_t* dr, size_t nl) {
    static uint8_t none[16];
    const int64_t want = p_out_bytes(s, nl);
    const size_t cap = want > 0 ? (size_t)want : 0;
    if (cap > 0 && p_is_exact(s) != 1) {
        /* the count is an upper bound, not the count (bit-reservoir extension): encode into scratch memory and hand out an exact copy -- a
         * zero-filled ArrayBuffer of the bound per call would be allocated only to be thrown away */
        uint8_t* tmp = (uint8_t*)malloc(cap);
        if (!tmp) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
        const int64_t m = p_encode(s, dl, dr, nl, tmp, cap);
        napi_value r = make_i8(env, tmp, m > 0 ? (size_t)m : 0);
        free(tmp);
        return r;
    }
    napi_value ab, ta; void* data = NULL;
    if (napi_create_arraybuffer(env, cap, &data, &ab) != napi_ok || (cap && !data)) { napi_throw_error(env, NULL, "could not allocate the output buffer"); return NULL; }
    const int64_t n = p_encode(s, dl, dr, nl, cap ? (uint8_t*)data : none, cap);
    /* the reference swallows negative codes and returns an empty array (index.js:128-129) */
    if (n != (int64_t)cap) return make_i8(env, (const uint8_t*)data, n > 0 ? (size_t)n : 0);
    if (napi_create_typedarray(env, napi_int8_array, cap, ab, 0, &ta) != napi_ok) { napi_throw_error(env, NULL, "could not create the output Int8Array"); return NULL; }
    return ta;
}

static napi_value js_encode(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value argv[3];
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    lhip_stream* s = NULL;
    if (argc < 2 || napi_get_value_external(env, argv[0], (void**)&s) != napi_ok || !s) { napi_throw_type_error(env, NULL, "first argument must be a stream handle"); return NULL; }
    napi_typedarray_type tt; size_t nl = 0, nr = 0; void *dl = NULL, *dr = NULL;
    if (napi_get_typedarray_info(env, argv[1], &tt, &nl, &dl, NULL, NULL) != napi_ok || tt != napi_int16_array) { napi_throw_type_error(env, NULL, "left must be an Int16Array"); return NULL; }
    napi_valuetype vt = napi_undefined;
    if (argc > 2) napi_typeof(env, argv[2], &vt);
    if (vt != napi_null && vt != napi_undefined) {
        if (napi_get_typedarray_info(env, argv[2], &tt, &nr, &dr, NULL, NULL) != napi_ok || tt != napi_int16_array || nr != nl) { napi_throw_type_error(env, NULL, "right must be an Int16Array of the same length"); return NULL; }
    }
    return encode_into_new_array(env, s, (const int16_t*)dl, (const int16_t*)dr, nl);
}

static napi_value js_flush(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    lhip_stream* s = NULL;
    if (argc < 1 || napi_get_value_external(env, argv[0], (void**)&s) != napi_ok || !s) { napi_throw_type_error(env, NULL, "first argument must be a stream handle"); return NULL; }
    size_t cap = p_max_out(s, 4 * 1152);
    uint8_t* out = (uint8_t*)malloc(cap ? cap : 1);
    if (!out) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    int64_t n = p_flush(s, out, cap);
    napi_value r = make_i8(env, out, n > 0 ? (size_t)n : 0);
    free(out);
    return r;
}

/* encodeBatch(handles[], lefts[], rights[]|null) / flushBatch(handles[]): the batch extension of the C ABI.  All streams of a
 * call must share one configuration and device (the library checks); a failed call throws (there is no reference behaviour to
 * mirror for it), a stream without completed frames gets an empty array. */
static napi_value batch_common(napi_env env, napi_callback_info info, int is_flush) {
    size_t argc = 3; napi_value argv[3];
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    uint32_t n = 0;
    if (argc < 1 || napi_get_array_length(env, argv[0], &n) != napi_ok) { napi_throw_type_error(env, NULL, "handles must be an array"); return NULL; }
    napi_value result; napi_create_array_with_length(env, n, &result);
    if (n == 0) return result;
    int have_right = 0;
    if (!is_flush) {
        uint32_t nl = 0;
        if (argc < 2 || napi_get_array_length(env, argv[1], &nl) != napi_ok || nl != n) { napi_throw_type_error(env, NULL, "lefts must be an array as long as handles"); return NULL; }
        if (argc > 2) { napi_valuetype vt; napi_typeof(env, argv[2], &vt); have_right = (vt != napi_null && vt != napi_undefined); }
    }
    lhip_stream** hs = (lhip_stream**)calloc(n, sizeof *hs);
    const int16_t** L = (const int16_t**)calloc(n, sizeof *L); const int16_t** R = (const int16_t**)calloc(n, sizeof *R);
    size_t* ns = (size_t*)calloc(n, sizeof *ns); size_t* caps = (size_t*)calloc(n, sizeof *caps);
    uint8_t** outs = (uint8_t**)calloc(n, sizeof *outs); int64_t* wr = (int64_t*)calloc(n, sizeof *wr);
    napi_value* abs_ = (napi_value*)calloc(n, sizeof *abs_);
    static uint8_t none[16];
    const char* err = NULL;
    for (uint32_t i = 0; i < n && !err; i++) {
        napi_value h; napi_get_element(env, argv[0], i, &h);
        if (napi_get_value_external(env, h, (void**)&hs[i]) != napi_ok) { err = "handles must come from create()"; break; }
        if (!is_flush) {
            napi_value a; napi_typedarray_type tt; void* d = NULL;
            napi_get_element(env, argv[1], i, &a);
            if (napi_get_typedarray_info(env, a, &tt, &ns[i], &d, NULL, NULL) != napi_ok || tt != napi_int16_array) { err = "lefts must hold Int16Arrays"; break; }
            L[i] = (const int16_t*)d;
            if (have_right) {
                size_t nr = 0; void* dr = NULL; napi_valuetype vt;
                napi_get_element(env, argv[2], i, &a); napi_typeof(env, a, &vt);
                if (vt != napi_null && vt != napi_undefined) {
                    if (napi_get_typedarray_info(env, a, &tt, &nr, &dr, NULL, NULL) != napi_ok || tt != napi_int16_array || nr != ns[i]) { err = "rights must hold Int16Arrays as long as their lefts"; break; }
                    R[i] = (const int16_t*)dr;
                }
            }
        }
        if (is_flush) { caps[i] = p_max_out(hs[i], 4 * 1152); outs[i] = (uint8_t*)malloc(caps[i] ? caps[i] : 1); }
        else {   /* the stream's result array at its final size (see encode_into_new_array): the launch writes into it */
            const int64_t want = p_out_bytes(hs[i], ns[i]);
            void* data = NULL;
            caps[i] = want > 0 ? (size_t)want : 0;
            if (caps[i] > 0 && p_is_exact(hs[i]) != 1) {      /* an upper bound only (bit reservoir): scratch memory, exact copy afterwards */
                outs[i] = (uint8_t*)malloc(caps[i]);
                if (!outs[i]) { err = "out of memory"; break; }
                continue;
            }
            if (napi_create_arraybuffer(env, caps[i], &data, &abs_[i]) != napi_ok || (caps[i] && !data)) { err = "could not allocate an output buffer"; break; }
            outs[i] = caps[i] ? (uint8_t*)data : none;
        }
    }
    if (!err) {
        const int rc = is_flush ? p_flush_batch(hs, n, outs, caps, wr) : p_encode_batch(hs, n, L, R, ns, outs, caps, wr);
        if (rc != 0) err = p_last_error();
    }
    if (!err) for (uint32_t i = 0; i < n; i++) {
        napi_value ta;
        if (!is_flush && abs_[i] && wr[i] == (int64_t)caps[i] && napi_create_typedarray(env, napi_int8_array, caps[i], abs_[i], 0, &ta) == napi_ok) napi_set_element(env, result, i, ta);
        else napi_set_element(env, result, i, make_i8(env, outs[i], wr[i] > 0 ? (size_t)wr[i] : 0));
    }
    for (uint32_t i = 0; i < n; i++) if (outs[i] && outs[i] != none && (is_flush || !abs_[i])) free(outs[i]);      /* scratch buffers (flush; bit-reservoir streams) */
    free(hs); free(L); free(R); free(ns); free(caps); free(outs); free(wr); free(abs_);
    if (err) { napi_throw_error(env, NULL, err); return NULL; }
    return result;
}
/* Frame-range sharding of one stream (include/lamejs_hip.h: lhip_seek / lhip_state_get / lhip_state_set).  Unlike encode(), these
 * throw on failure: there is no reference behaviour to mirror. */
static lhip_stream* handle_arg(napi_env env, napi_value v) {
    lhip_stream* s = NULL;
    if (napi_get_value_external(env, v, (void**)&s) != napi_ok || !s) { napi_throw_type_error(env, NULL, "first argument must be a stream handle"); return NULL; }
    return s;
}
static napi_value js_seek_tail(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1], r;
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    lhip_stream* s = argc >= 1 ? handle_arg(env, argv[0]) : NULL;
    if (!s) return NULL;
    napi_create_uint32(env, (uint32_t)p_seek_tail(s), &r);
    return r;
}
static napi_value js_seek(napi_env env, napi_callback_info info) {
    size_t argc = 4; napi_value argv[4];
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    lhip_stream* s = argc >= 3 ? handle_arg(env, argv[0]) : NULL;
    if (!s) return NULL;
    int64_t pos = 0;
    if (napi_get_value_int64(env, argv[1], &pos) != napi_ok) { napi_throw_type_error(env, NULL, "samplePos must be a number"); return NULL; }
    napi_typedarray_type tt; size_t nl = 0, nr = 0; void *dl = NULL, *dr = NULL;
    if (napi_get_typedarray_info(env, argv[2], &tt, &nl, &dl, NULL, NULL) != napi_ok || tt != napi_int16_array || nl != p_seek_tail(s)) { napi_throw_type_error(env, NULL, "tailLeft must be an Int16Array of seekTailSamples() samples"); return NULL; }
    napi_valuetype vt = napi_undefined;
    if (argc > 3) napi_typeof(env, argv[3], &vt);
    if (vt != napi_null && vt != napi_undefined) {
        if (napi_get_typedarray_info(env, argv[3], &tt, &nr, &dr, NULL, NULL) != napi_ok || tt != napi_int16_array || nr != nl) { napi_throw_type_error(env, NULL, "tailRight must be an Int16Array of the same length"); return NULL; }
    }
    if (p_seek(s, pos, (const int16_t*)dl, (const int16_t*)dr) != 0) { napi_throw_error(env, NULL, p_last_error()); return NULL; }
    return NULL;
}
static napi_value js_state_get(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1], ab, ta; void* data = NULL;
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    lhip_stream* s = argc >= 1 ? handle_arg(env, argv[0]) : NULL;
    if (!s) return NULL;
    const size_t n = p_state_bytes(s);
    if (n == 0 || napi_create_arraybuffer(env, n, &data, &ab) != napi_ok || !data) { napi_throw_error(env, NULL, "stateGet: could not allocate the state buffer"); return NULL; }
    if (p_state_get(s, data, n) != 0) { napi_throw_error(env, NULL, p_last_error()); return NULL; }
    if (napi_create_typedarray(env, napi_uint8_array, n, ab, 0, &ta) != napi_ok) { napi_throw_error(env, NULL, "stateGet: could not create the result array"); return NULL; }
    return ta;
}
static napi_value js_state_set(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    lhip_stream* s = argc >= 2 ? handle_arg(env, argv[0]) : NULL;
    if (!s) return NULL;
    napi_typedarray_type tt; size_t n = 0; void* d = NULL;
    if (napi_get_typedarray_info(env, argv[1], &tt, &n, &d, NULL, NULL) != napi_ok || tt != napi_uint8_array) { napi_throw_type_error(env, NULL, "state must be a Uint8Array"); return NULL; }
    if (p_state_set(s, d, n) != 0) { napi_throw_error(env, NULL, p_last_error()); return NULL; }
    return NULL;
}

static napi_value js_encode_batch(napi_env env, napi_callback_info info) { return batch_common(env, info, 0); }
static napi_value js_flush_batch(napi_env env, napi_callback_info info) { return batch_common(env, info, 1); }

static napi_value init(napi_env env, napi_value exports) {
    napi_property_descriptor d[] = {
        {"deviceCount", 0, js_device_count, 0, 0, 0, napi_default, 0}, {"create", 0, js_create, 0, 0, 0, napi_default, 0},
        {"encode", 0, js_encode, 0, 0, 0, napi_default, 0}, {"flush", 0, js_flush, 0, 0, 0, napi_default, 0},
        {"encodeBatch", 0, js_encode_batch, 0, 0, 0, napi_default, 0}, {"flushBatch", 0, js_flush_batch, 0, 0, 0, napi_default, 0},
        {"setDevices", 0, js_set_devices, 0, 0, 0, napi_default, 0},
        {"seekTailSamples", 0, js_seek_tail, 0, 0, 0, napi_default, 0}, {"seek", 0, js_seek, 0, 0, 0, napi_default, 0},
        {"stateGet", 0, js_state_get, 0, 0, 0, napi_default, 0}, {"stateSet", 0, js_state_set, 0, 0, 0, napi_default, 0}};
    napi_define_properties(env, exports, sizeof d / sizeof d[0], d);
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
