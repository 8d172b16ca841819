// fused matrix-core engine, 11 window blocks of 16 positions: pad 57..72 (2 pad + 1 taps)
#include "fx_kernels.hpp"
BLUR_FX(11)
