// bf16 MFMA attention kernels (throughput path).  Until a shape is covered here the dispatcher in
// attn.hip falls through to the row kernels (same results, fp32 math on the vector ALUs).
#include "common.h"

int b4c_attn_fwd_mfma(const void *, int, const uint8_t *, void *, int, float *, int, int, int, int, hipStream_t) {
    return B4C_EUNSUPPORTED;
}
int b4c_attn_bwd_mfma(const void *, int, const uint8_t *, const void *, int, const void *, int, const float *, float *,
                      void *, int, int, int, int, int, hipStream_t) {
    return B4C_EUNSUPPORTED;
}
