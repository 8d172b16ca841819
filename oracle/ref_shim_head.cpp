// ref_shim_head.cpp -- first part of the translation unit that becomes
// oracle/_ref/libref_utils.so.  TEST INFRASTRUCTURE ONLY.
//
// The Makefile builds that library from the REFERENCE'S OWN SOURCES where they
// lie under /root/reference (nothing is copied into this repository):
//     this file | Source.cpp:58-102 | Source.cpp:414-427 (both piped by sed) | ref_shim_tail.cpp
// with -I/root/reference so that "Utils.hpp" is the reference's header.
// Source.cpp as a whole cannot be compiled here (pffft, pocketfft, FastBoxBlur
// and OpenCV are absent), but gaussian_window / getGaussian (lines 60-102) are
// self-contained and only need the standard headers below, which the
// reference gets transitively from its own includes on the author's toolchain.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <numeric>
#include <type_traits>
#include <vector>
#include "Utils.hpp"
// ---- Source.cpp:58-102 (AlignedVector, gaussian_window, getGaussian) and Source.cpp:414-427
// ---- (pffft_sorted_optimized_convolution) follow
