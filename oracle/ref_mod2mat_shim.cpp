// TEST INFRASTRUCTURE — not part of the product.  A C-ABI shim over the REFERENCE's own GF(2) matrix code
// (/root/reference/lib/data/MNC/radford/mod2mat.cpp, compiled from where it lies by oracle/build_ref.sh into
// oracle/_ref/libmod2mat_ref.so) that encodes one message exactly as `s2t` does
// (/root/reference/lib/data/MNC/MNC_py.cpp:22-83: read G, set the K source bits, t = G*s, emit [s | t]).
// Used only to generate tests/golden/ldpc_datapath.npz and to pin the numpy restatement in oracle/fgnn_oracle.py.
#include <stdint.h>
#include <stdio.h>
#include "mod2mat.h"

extern "C" int ref_G_dims(const char* gfile, int* rows, int* cols) {
    FILE* fp = fopen(gfile, "rb");
    if (!fp) return -1;
    int code = 0;
    mod2mat* G = mod2mat_read(fp, &code);
    fclose(fp);
    if (!G) return -2;
    *rows = G->n_rows;
    *cols = G->n_cols;
    mod2mat_free(G);
    return 0;
}

// out[r * cols + c] = G(r, c)
extern "C" int ref_G_bits(const char* gfile, uint8_t* out) {
    FILE* fp = fopen(gfile, "rb");
    if (!fp) return -1;
    int code = 0;
    mod2mat* G = mod2mat_read(fp, &code);
    fclose(fp);
    if (!G) return -2;
    const int R = G->n_rows, C = G->n_cols;
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) out[r * C + c] = (uint8_t)mod2mat_get(G, r, c);
    mod2mat_free(G);
    return 0;
}

// One K-bit message -> [s (K bits) | t = G s (N bits)], the `smn=true` output of s2t.
extern "C" int ref_encode(const char* gfile, const uint8_t* src, int K, int N, uint8_t* out) {
    FILE* fp = fopen(gfile, "rb");
    if (!fp) return -1;
    int code = 0;
    mod2mat* G = mod2mat_read(fp, &code);
    fclose(fp);
    if (!G) return -2;
    mod2mat* s = mod2mat_allocate(K, 1);
    mod2mat* t = mod2mat_allocate(N, 1);
    for (int b = 0; b < K; ++b) mod2mat_set(s, b, 0, src[b]);
    mod2mat_multiply(G, s, t);
    for (int b = 0; b < K; ++b) out[b] = (uint8_t)mod2mat_get(s, b, 0);
    for (int b = 0; b < N; ++b) out[K + b] = (uint8_t)mod2mat_get(t, b, 0);
    mod2mat_free(s); mod2mat_free(t); mod2mat_free(G);
    return 0;
}
