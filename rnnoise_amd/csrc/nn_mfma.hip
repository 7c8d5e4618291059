// nn_mfma.hip -- batched MFMA network path (placeholder until the vector path is validated on
// hardware; rnnoise_batch_set_nn_path(b, 1) is refused while this returns NotSupported).
#include <hip/hip_runtime.h>
#include "rn_dev.h"

extern "C" hipError_t rn_launch_nn_mfma(const RnGroupDev *, const RnModelDev *, const RnTablesDev *, hipStream_t) {
  return hipErrorNotSupported;
}
extern "C" int rn_nn_mfma_available(void) { return 0; }
