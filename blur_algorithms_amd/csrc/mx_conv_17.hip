// matrix-core engine, 17 window blocks of 16 positions: pad 105..120 (2 pad + 1 taps)
#include "mx_kernels.hpp"
BLUR_MX(17)
