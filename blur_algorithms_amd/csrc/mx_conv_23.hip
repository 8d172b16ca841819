// matrix-core engine, 23 window blocks of 16 positions: pad 153..168 (2 pad + 1 taps)
#include "mx_kernels.hpp"
BLUR_MX(23)
