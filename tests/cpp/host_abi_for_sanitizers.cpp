// host_abi_for_sanitizers.cpp -- TEST INFRASTRUCTURE for tests/test_sanitizers.py.
// The host-side entry points of include/blur_amd.h over the product's own host arithmetic (csrc/host_math.cpp), without HIP:
// the AddressSanitizer / UBSan build of tests/cpp/surface_check.cpp ("host" mode) and of tests/cpp/sanitize_vectors.cpp links
// this file instead of libblur_amd.so (GPU AddressSanitizer is not available on this pool; the reference's only hook of the
// kind is the commented LSan include, Utils.hpp:12).  Entry points that need the GPU report BLUR_ERR_HIP, as the library does
// on a box without one.
#include <cstring>

#include "blur_amd.h"
#include "host_math.hpp"

extern "C" {
int blur_gaussian_window(double sigma, int max_width) { return blur_amd::gaussian_window(sigma, max_width); }
int blur_get_gaussian(float* kernel, double sigma, int width, int fft_length)
{
    if (!kernel || !(sigma > 0) || width < 0 || fft_length < 0) return BLUR_ERR_INVALID;
    blur_amd::get_gaussian(kernel, sigma, width, fft_length);
    return BLUR_OK;
}
int blur_is_valid_size(int n) { return blur_amd::is_valid_size(n); }
int blur_nearest_transform_size(int n) { return blur_amd::nearest_transform_size(n); }
void blur_opts_default(blur_opts* o)
{
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->nyquist_quirk = 1;
}
const char* blur_last_error(const blur_ctx*) { return "no GPU in the sanitizer build"; }
int blur_ctx_create(blur_ctx**, int) { return BLUR_ERR_HIP; }
int blur_ctx_destroy(blur_ctx*) { return BLUR_OK; }
int blur_fastboxblur_u8_host(blur_ctx*, uint8_t*, int, int, int, int, int) { return BLUR_ERR_HIP; }
int blur_gaussian_u8c3_host(blur_ctx*, const uint8_t*, uint8_t*, int, int, double, const blur_opts*) { return BLUR_ERR_HIP; }
int blur_pocketfft2d_u8c3_host(blur_ctx*, const uint8_t*, uint8_t*, int, int, double, int) { return BLUR_ERR_HIP; }
}
