// matrix-core engine, 21 window blocks of 16 positions: pad 137..152 (2 pad + 1 taps)
#include "mx_kernels.hpp"
BLUR_MX(21)
