// placeholder until the backward kernels land (next commit)
#include "fgnn_common.h"
extern "C" int fgnn_mpconv_backward(const fgnn_mpconv_desc*, const void*, const int64_t*, const void*,
                                    const float*, const void*, const void*, const uint8_t*, void*,
                                    float*, float*, float*, fgnn_stream_t) {
    FGNN_FAIL(FGNN_EUNSUPPORTED, "mpconv backward: not built yet");
}
