// fused matrix-core engine, 5 window blocks of 16 positions: pad <= 24 (2 pad + 1 taps)
#include "fx_kernels.hpp"
BLUR_FX(5)
