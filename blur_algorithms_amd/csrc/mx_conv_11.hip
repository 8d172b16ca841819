// matrix-core engine, 11 window blocks of 16 positions: pad 57..72 (2 pad + 1 taps)
#include "mx_kernels.hpp"
BLUR_MX(11)
