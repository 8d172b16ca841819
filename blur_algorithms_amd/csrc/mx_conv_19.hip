// matrix-core engine, 19 window blocks of 16 positions: pad 121..136 (2 pad + 1 taps)
#include "mx_kernels.hpp"
BLUR_MX(19)
