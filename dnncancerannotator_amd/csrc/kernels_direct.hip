// kernels_direct.hip -- tuned direct (VALU) kernels for the tiny-channel, high-resolution layers of configs/unet.yaml.
#include "fast.h"
#include "kernels.h"

namespace dnnca {

bool fast_conv_fwd(Model*, int, Op&, double, double) { return false; }
bool fast_conv_bwd(Model*, int, Op&, double, double, double) { return false; }
bool fast_pool_fwd(Model*, int, Op&, double) { return false; }
bool fast_pool_bwd(Model*, int, Op&, double) { return false; }
bool fast_tconv_fwd(Model*, int, Op&, double, double) { return false; }
bool fast_tconv_bwd(Model*, int, Op&, double, double, double) { return false; }

}  // namespace dnnca
