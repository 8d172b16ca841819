// matrix-core engine, 13 window blocks of 16 positions: pad 73..88 (2 pad + 1 taps)
#include "mx_kernels.hpp"
BLUR_MX(13)
