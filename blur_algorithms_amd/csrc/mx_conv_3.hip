// matrix-core engine, 3 window blocks of 16 positions: pad 0..8 (2 pad + 1 taps)
#include "mx_kernels.hpp"
BLUR_MX(3)
