// matrix-core engine, 9 window blocks of 16 positions: pad 41..56 (2 pad + 1 taps)
#include "mx_kernels.hpp"
BLUR_MX(9)
