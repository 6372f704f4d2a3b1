/*
 * oracle/ref_backend_async.c -- TEST INFRASTRUCTURE.  Drives the module's optional backend entry points through the reference's PUBLIC
 * ggml-backend API (ggml/include/ggml-backend.h), the way the scheduler does for the activations that cross a layer split
 * (ggml-backend.cpp:414-433, 1473-1477):  two backends (streams) of a device of the HIP registry, ggml_backend_tensor_copy_async between
 * their buffers, ggml_backend_event_new / record / wait / synchronize.  Prints "OK" and returns 0 when every byte arrived.
 * usage: ref_backend_async /path/to/libggml-hip.so [n_floats]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ggml.h"
#include "ggml-backend.h"

int main(int argc, char ** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s libggml-hip.so [n]\n", argv[0]); return 2; }
    const int64_t n = argc > 2 ? atoll(argv[2]) : (1 << 20);
    ggml_backend_reg_t reg = ggml_backend_load(argv[1]);
    if (!reg || ggml_backend_reg_dev_count(reg) == 0) { fprintf(stderr, "module not loaded / no device\n"); return 3; }
    const size_t ndev = ggml_backend_reg_dev_count(reg);
    ggml_backend_dev_t d0 = ggml_backend_reg_dev_get(reg, 0), d1 = ggml_backend_reg_dev_get(reg, ndev > 1 ? 1 : 0);
    struct ggml_backend_dev_props props; ggml_backend_dev_get_props(d0, &props);
    if (!props.caps.events || !props.caps.async) { fprintf(stderr, "caps: events %d async %d\n", props.caps.events, props.caps.async); return 4; }
    ggml_backend_t a = ggml_backend_dev_init(d0, NULL), b = ggml_backend_dev_init(d1, NULL);
    if (!a || !b) { fprintf(stderr, "dev_init failed\n"); return 5; }

    struct ggml_init_params ip = { ggml_tensor_overhead() * 8, NULL, true };
    struct ggml_context * ca = ggml_init(ip), * cb = ggml_init(ip);
    struct ggml_tensor * src = ggml_new_tensor_1d(ca, GGML_TYPE_F32, n), * mid = ggml_new_tensor_1d(cb, GGML_TYPE_F32, n), * back = ggml_new_tensor_1d(ca, GGML_TYPE_F32, n);
    ggml_backend_buffer_t ba = ggml_backend_alloc_ctx_tensors(ca, a), bb = ggml_backend_alloc_ctx_tensors(cb, b);
    if (!ba || !bb) { fprintf(stderr, "alloc failed\n"); return 6; }

    float * h = (float *) malloc((size_t) n * 4), * g = (float *) malloc((size_t) n * 4);
    int bad = 0;
    for (int round = 0; round < 4 && !bad; round++) {
        for (int64_t i = 0; i < n; i++) h[i] = (float)(i % 9973) * 0.25f + (float) round;
        ggml_backend_tensor_set(src, h, 0, (size_t) n * 4);
        ggml_backend_tensor_copy_async(a, b, src, mid);            /* a -> b: queued behind a's work, b waits for it */
        ggml_backend_tensor_copy_async(b, a, mid, back);           /* b -> a: must see the first copy */
        ggml_backend_event_t ev = ggml_backend_event_new(d0);
        if (!ev) { fprintf(stderr, "event_new failed\n"); return 7; }
        ggml_backend_event_record(ev, a);
        ggml_backend_event_wait(b, ev);                            /* b orders itself behind a's event (no-op for the data, exercises the entry point) */
        ggml_backend_event_synchronize(ev);
        memset(g, 0, (size_t) n * 4);
        ggml_backend_tensor_get(back, g, 0, (size_t) n * 4);
        if (memcmp(g, h, (size_t) n * 4)) { bad = 1; fprintf(stderr, "round %d: data differs after a -> b -> a\n", round); }
        ggml_backend_synchronize(b);
        ggml_backend_tensor_get(mid, g, 0, (size_t) n * 4);
        if (memcmp(g, h, (size_t) n * 4)) { bad = 1; fprintf(stderr, "round %d: data differs on b\n", round); }
        ggml_backend_event_free(ev);
    }
    printf(bad ? "FAILED\n" : "OK devices=%zu bytes=%zu\n", ndev, (size_t) n * 4);
    ggml_backend_buffer_free(ba); ggml_backend_buffer_free(bb);
    ggml_free(ca); ggml_free(cb);
    ggml_backend_free(a); ggml_backend_free(b);
    free(h); free(g);
    return bad;
}
