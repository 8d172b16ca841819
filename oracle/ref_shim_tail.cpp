// ---- end of Source.cpp:58-102, 414-427
// ref_shim_tail.cpp -- C entry points over the reference's own functions.
extern "C" {

int ref_gaussian_window(double sigma, int max_width) { return gaussian_window(sigma, max_width); }

// called the way pffft_() calls it (Source.cpp:465,471): a zero-filled
// AlignedVector<float> of FFT_length, then getGaussian(kernel, sigma, kSize, FFT_length)
void ref_get_gaussian(float* out, double sigma, int width, int fft_length)
{
    AlignedVector<float> k(fft_length ? fft_length : (width ? width : gaussian_window(sigma)));
    getGaussian(k, sigma, width, fft_length);
    std::copy(k.begin(), k.end(), out);
}

int ref_is_valid_size(int n) { return isValidSize(n); }
int ref_nearest_transform_size(int n) { return nearestTransformSize(n); }

void ref_deinterleave_bgr_u8_f32(const uint8_t* in, float* B, float* G, float* R, uint32_t total)
{
    float* planes[3] = { B, G, R };
    deinterleave_BGR(in, planes, total);
}

void ref_interleave_bgr_f32_u8(const float* B, const float* G, const float* R, uint8_t* out, uint32_t total)
{
    const float* planes[3] = { B, G, R };
    interleave_BGR(planes, out, total);
}

void ref_reflect_101_u8c3(const uint8_t* in, uint8_t* out, int pt, int pb, int pl, int pr, const int* size)
{ Reflect_101<uint8_t, 3>(in, out, pt, pb, pl, pr, size); }
void ref_reflect_101_u8c1(const uint8_t* in, uint8_t* out, int pt, int pb, int pl, int pr, const int* size)
{ Reflect_101<uint8_t, 1>(in, out, pt, pb, pl, pr, size); }
void ref_reflect_101_f32c1(const float* in, float* out, int pt, int pb, int pl, int pr, const int* size)
{ Reflect_101<float, 1>(in, out, pt, pb, pl, pr, size); }

// hybrid_loop (Utils.hpp:16-55): counts how often each index is visited
void ref_hybrid_loop_count(int end, int* hits)
{
    hybrid_loop(end, [&](auto i) {
#pragma omp atomic
        hits[i] += 1;
    });
}

// pffft_sorted_optimized_convolution (Source.cpp:414-427) on an ordered spectrum the way pffft_() calls it (Source.cpp:532,554):
// tile_dft and kernel_dft are AlignedVector<float> of the transform length, scaler = 1 / N
void ref_sorted_optimized_convolution(float* tile_dft, const float* kernel_dft, int size, float scaler)
{
    AlignedVector<float> t(tile_dft, tile_dft + size), k(kernel_dft, kernel_dft + size);
    pffft_sorted_optimized_convolution(t, k, scaler);
    std::copy(t.begin(), t.end(), tile_dft);
}

}  // extern "C"
