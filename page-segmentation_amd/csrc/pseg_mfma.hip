// pseg_mfma.hip -- bf16 throughput mode (placeholder until the MFMA kernels land).
#include "pseg_common.h"

namespace pseg {

int mfma_pack_op(Engine&, Op&) { return fail(PSEG_EUNSUPPORTED, "bf16 mode not built yet"); }
int mfma_launch_conv(Engine&, Op&, hipStream_t) { return fail(PSEG_EUNSUPPORTED, "bf16 mode not built yet"); }
int mfma_launch_deconv2(Engine&, Op&, hipStream_t) { return fail(PSEG_EUNSUPPORTED, "bf16 mode not built yet"); }
int mfma_launch_pool(Engine&, Op&, hipStream_t) { return fail(PSEG_EUNSUPPORTED, "bf16 mode not built yet"); }
int mfma_launch_logits(Engine&, Op&, float*, float*, int64_t*, uint8_t*, hipStream_t) {
    return fail(PSEG_EUNSUPPORTED, "bf16 mode not built yet");
}
int mfma_preprocess(Engine&, const uint8_t*, hipStream_t) { return fail(PSEG_EUNSUPPORTED, "bf16 mode not built yet"); }

}  // namespace pseg
