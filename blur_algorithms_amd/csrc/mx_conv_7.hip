// matrix-core engine, 7 window blocks of 16 positions: pad 25..40 (2 pad + 1 taps)
#include "mx_kernels.hpp"
BLUR_MX(7)
