// fastboxblur on the integer matrix cores: the P sweeps of one direction in one kernel, no LDS, no running sums.
// bx_box.hip has the kernels; engine.hip calls these two and falls back to its accumulator kernels when `*ran` comes back false
// (box wider than the instantiated windows, tiny images, pitches that are not a multiple of 4 bytes, channel counts other than 1/3/4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace blur_amd {

// `passes` sweeps of a (2 r + 1)-row box down the columns of a [h][pitch] byte image (reflect-101, u8 rounding after every sweep:
// oracle/boxblur_oracle.c), in -> out, in != out.
hipError_t bx_vertical(hipStream_t st, const uint8_t* in, uint8_t* out, int h, int pitch, int r, int passes, int num_cus, bool* ran);

// `passes` sweeps of a (2 r + 1)-pixel box along the rows of a [h][w][C] byte image, in -> out, in != out.  `strips` is scratch of
// bx_horizontal_scratch(...) bytes (the mirrored margins of every row).
size_t bx_horizontal_scratch(int h, int w, int C, int r, int passes);
hipError_t bx_horizontal(hipStream_t st, const uint8_t* in, uint8_t* out, uint8_t* strips, int h, int w, int C, int r, int passes, int num_cus, bool* ran);

}  // namespace blur_amd
