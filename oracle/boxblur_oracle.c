/*
 * boxblur_oracle.c -- CPU ORACLE for fastboxblur (BASELINE config 5).
 * TEST INFRASTRUCTURE ONLY (same rules as blur_oracle.c).
 *
 * PARITY UNPINNED.  The reference calls
 *     fastboxblur(in, cols, rows, channels, ksize, passes)      Source.cpp:587
 * but the implementation lives in the un-vendored FastBoxBlur submodule
 * (.gitmodules:1-3), which is absent from /root/reference, and the reference
 * holds no test or golden output for it.  What the reference does state:
 *   - "sliding accumulator and padding by reflection as default without
 *      increasing the memory usage"                            README.md:17-19
 *   - argument 5 is the box WIDTH and argument 6 the number of passes (the
 *     parallel OpenCV branch is cv::blur(Size(n*n, n*n)) twice, Source.cpp:597-600)
 *   - in place on the interleaved u8 buffer                    Source.cpp:586-587
 * The published algorithm restated here (FastBoxBlur / "fast gaussian blur"
 * family): per 1D sweep a running integer sum over 2r+1 taps, r = (ksize-1)/2
 * clamped to line_length-1, reflect-101 borders, result = (T)(acc * (1.f/(2r+1)) + 0.5f)
 * for integral T; `passes` horizontal sweeps, transpose, `passes` sweeps along the
 * other axis, transpose back.  Intermediates are u8 (rounded after every sweep).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int refl101(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

/* one sweep along lines of length n; element (line l, position x, channel c)
   lives at in[l*lstride + x*xstride + c] */
static void sweep_u8(const uint8_t* in, uint8_t* out, int nlines, int n, size_t lstride, size_t xstride, int C, int ksize)
{
    int r = (ksize - 1) / 2;
    if (r > n - 1) r = n - 1;
    if (r < 0) r = 0;
    const float iarr = 1.f / (float)(r + r + 1);
#pragma omp parallel for schedule(static)
    for (int l = 0; l < nlines; ++l)
        for (int c = 0; c < C; ++c) {
            const uint8_t* ip = in + (size_t)l * lstride + c;
            uint8_t* op = out + (size_t)l * lstride + c;
            int acc = 0;
            for (int d = -r; d <= r; ++d) acc += ip[(size_t)refl101(d, n) * xstride];
            op[0] = (uint8_t)(int)((float)acc * iarr + 0.5f);
            for (int x = 1; x < n; ++x) {
                acc += ip[(size_t)refl101(x + r, n) * xstride];
                acc -= ip[(size_t)refl101(x - r - 1, n) * xstride];
                op[(size_t)x * xstride] = (uint8_t)(int)((float)acc * iarr + 0.5f);
            }
        }
}

/* fastboxblur<uint8_t>(inout, w, h, channels, ksize, passes), in place */
int ora_fastboxblur_u8(uint8_t* inout, int w, int h, int channels, int ksize, int passes)
{
    if (w <= 0 || h <= 0 || channels <= 0 || ksize <= 0 || passes < 0) return -1;
    const size_t bytes = (size_t)w * h * channels;
    uint8_t* tmp = (uint8_t*)malloc(bytes);
    uint8_t *a = inout, *b = tmp;
    for (int p = 0; p < passes; ++p) {            /* horizontal sweeps */
        sweep_u8(a, b, h, w, (size_t)w * channels, (size_t)channels, channels, ksize);
        uint8_t* t = a; a = b; b = t;
    }
    for (int p = 0; p < passes; ++p) {            /* vertical sweeps (strided == transpose, sweep, transpose) */
        sweep_u8(a, b, w, h, (size_t)channels, (size_t)w * channels, channels, ksize);
        uint8_t* t = a; a = b; b = t;
    }
    if (a != inout) memcpy(inout, a, bytes);
    free(tmp);
    return 0;
}
