// fused matrix-core engine, 7 window blocks of 16 positions: pad <= 40 (2 pad + 1 taps)
#include "fx_kernels.hpp"
BLUR_FX(7)
