// matrix-core engine, 15 window blocks of 16 positions: pad 89..104 (2 pad + 1 taps)
#include "mx_kernels.hpp"
BLUR_MX(15)
