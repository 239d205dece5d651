// fast.h -- hooks through which model.hip routes an op to a tuned gfx950 kernel.  Each returns false when it has no
// specialisation for the op's shape, in which case the caller falls through to the generic kernel.
#pragma once
#include "model.h"

namespace dnnca {

bool fast_conv_fwd(Model* m, int B, Op& o, double bytes, double flops);
bool fast_conv_bwd(Model* m, int B, Op& o, double out_bytes, double in_bytes, double flops);
bool fast_pool_fwd(Model* m, int B, Op& o, double bytes);
bool fast_pool_bwd(Model* m, int B, Op& o, double bytes);
bool fast_tconv_fwd(Model* m, int B, Op& o, double bytes, double flops);
bool fast_tconv_bwd(Model* m, int B, Op& o, double out_bytes, double in_bytes, double flops);

}  // namespace dnnca
