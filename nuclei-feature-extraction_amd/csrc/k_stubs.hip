// temporary: feature sets whose kernels are not written yet
#include "nfx_internal.h"
namespace nfx {

hipError_t launch_glrlm(nfx_ctx *, const Job &, int) { return hipErrorNotSupported; }
hipError_t launch_gabor(nfx_ctx *, const Job &, int) { return hipErrorNotSupported; }
}
