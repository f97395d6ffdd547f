/*
 * Thin N-API shim between the JavaScript host (lamejs_amd/js/index.js) and the C ABI of
 * include/lamejs_hip.h.  It only marshals typed arrays: no encoding logic lives here.
 * The HIP library is located at run time (dlopen of ../../lib/liblamejs_hip.so relative to this
 * addon, or $LAMEJS_HIP_LIB), so the addon itself builds with plain gcc + node headers:
 *     gcc -O2 -fPIC -shared -I/usr/include/node -I../../../include lhip_napi.c -o lhip_napi.node -ldl
 *
 * JS surface:  create(tablesBlob: Buffer, channels, samplerate, kbps) -> handle
 *              encode(handle, left: Int16Array, right: Int16Array|null) -> Int8Array
 *              flush(handle) -> Int8Array ;  destroy(handle) ;  deviceCount() -> number
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
static const char* (*p_last_error)(void);

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
    SYM(p_destroy, "lhip_destroy") SYM(p_max_out, "lhip_max_output_bytes") SYM(p_last_error, "lhip_last_error")
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

static napi_value js_create(napi_env env, napi_callback_info info) {
    size_t argc = 4; napi_value argv[4];
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    if (!load_lib(env)) return NULL;
    void* blob; size_t nblob;
    if (napi_get_buffer_info(env, argv[0], &blob, &nblob) != napi_ok) { napi_throw_type_error(env, NULL, "tables must be a Buffer"); return NULL; }
    lhip_config cfg; cfg.device = -1;
    napi_get_value_int32(env, argv[1], &cfg.channels);
    napi_get_value_int32(env, argv[2], &cfg.samplerate);
    napi_get_value_int32(env, argv[3], &cfg.kbps);
    lhip_stream* s = NULL;
    if (p_create(&cfg, blob, nblob, &s) != 0) { napi_throw_error(env, NULL, p_last_error()); return NULL; }
    napi_value ext;
    napi_create_external(env, s, finalize_stream, NULL, &ext);
    return ext;
}

static napi_value make_i8(napi_env env, const uint8_t* src, size_t n) {
    napi_value ab, ta; void* data;
    napi_create_arraybuffer(env, n, &data, &ab);
    if (n) memcpy(data, src, n);
    napi_create_typedarray(env, napi_int8_array, n, ab, 0, &ta);
    return ta;
}

static napi_value js_encode(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value argv[3];
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    lhip_stream* s; napi_get_value_external(env, argv[0], (void**)&s);
    napi_typedarray_type tt; size_t nl = 0, nr = 0; void *dl = NULL, *dr = NULL;
    if (napi_get_typedarray_info(env, argv[1], &tt, &nl, &dl, NULL, NULL) != napi_ok || tt != napi_int16_array) { napi_throw_type_error(env, NULL, "left must be an Int16Array"); return NULL; }
    napi_valuetype vt; napi_typeof(env, argv[2], &vt);
    if (vt != napi_null && vt != napi_undefined) {
        if (napi_get_typedarray_info(env, argv[2], &tt, &nr, &dr, NULL, NULL) != napi_ok || tt != napi_int16_array || nr != nl) { napi_throw_type_error(env, NULL, "right must be an Int16Array of the same length"); return NULL; }
    }
    size_t cap = p_max_out(s, nl);
    uint8_t* out = (uint8_t*)malloc(cap ? cap : 1);
    int64_t n = p_encode(s, (const int16_t*)dl, (const int16_t*)dr, nl, out, cap);
    /* the reference swallows negative codes and returns an empty array (index.js:128-129) */
    napi_value r = make_i8(env, out, n > 0 ? (size_t)n : 0);
    free(out);
    return r;
}

static napi_value js_flush(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    napi_get_cb_info(env, info, &argc, argv, NULL, NULL);
    lhip_stream* s; napi_get_value_external(env, argv[0], (void**)&s);
    size_t cap = p_max_out(s, 4 * 1152);
    uint8_t* out = (uint8_t*)malloc(cap);
    int64_t n = p_flush(s, out, cap);
    napi_value r = make_i8(env, out, n > 0 ? (size_t)n : 0);
    free(out);
    return r;
}

static napi_value init(napi_env env, napi_value exports) {
    napi_property_descriptor d[] = {
        {"deviceCount", 0, js_device_count, 0, 0, 0, napi_default, 0}, {"create", 0, js_create, 0, 0, 0, napi_default, 0},
        {"encode", 0, js_encode, 0, 0, 0, napi_default, 0}, {"flush", 0, js_flush, 0, 0, 0, napi_default, 0}};
    napi_define_properties(env, exports, 4, d);
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
