// matrix-core engine, 5 window blocks of 16 positions: pad 9..24 (2 pad + 1 taps)
#include "mx_kernels.hpp"
BLUR_MX(5)
