// matrix-core engine, 11 window blocks: pad 57..72 (sigma 20: 131 taps, pad 65 -- the metric's kernel)
#include "mx_kernels.hpp"
BLUR_MX(11)
