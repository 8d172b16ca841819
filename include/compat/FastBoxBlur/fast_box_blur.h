// FastBoxBlur/fast_box_blur.h -- include-path shim for the un-vendored
// FastBoxBlur submodule (.gitmodules:1-3): the two symbols Source.cpp takes from it,
//   flip_block<T, C>(in, out, w, h)                       call sites Source.cpp:367,384,540,562
//   fastboxblur(in, w, h, channels, ksize, passes)        call site  Source.cpp:587
// provided by the MI355X engine (flip_block on the host, fastboxblur on the GPU through
// libblur_amd.so).  Use with  -I<repo>/include/compat -I<repo>/include.
//
// The companion shim pffft_pommier/pffft.h is a host implementation of pffft's per-line interface; the
// MI355X path replaces the whole pffft_() body instead (blur_amd.hpp: pffft_(Mat&, double)), see INTEGRATION.md.
#pragma once
#include "../../blur_amd.hpp"
using blur_amd::compat::fastboxblur;
using blur_amd::compat::flip_block;
