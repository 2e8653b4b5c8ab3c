// temporary: feature sets whose kernels are not written yet
#include "nfx_internal.h"
namespace nfx {
size_t color_scratch_bytes(const Job &) { return 256; }
hipError_t launch_color(nfx_ctx *, const Job &, int, void *) { return hipErrorNotSupported; }
hipError_t launch_glcm(nfx_ctx *, const Job &, int) { return hipErrorNotSupported; }
hipError_t launch_glrlm(nfx_ctx *, const Job &, int) { return hipErrorNotSupported; }
hipError_t launch_gabor(nfx_ctx *, const Job &, int) { return hipErrorNotSupported; }
}
