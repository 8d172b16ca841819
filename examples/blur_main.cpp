// examples/blur_main.cpp -- the reference's main() (Source.cpp:611-641) over the engine: argv[1] = algorithm flag,
// argv[2] = sigma (nsmooth), argv[3] = image file (binary PPM), result written next to it as <file>.out.ppm.
//   g++ -std=c++17 -O2 -I include examples/blur_main.cpp -L blur_algorithms_amd -lblur_amd \
//       -Wl,-rpath,$PWD/blur_algorithms_amd -Wl,-rpath,/opt/rocm/lib -o blur_main
//   ./blur_main 3 20 image.ppm
// Flags as in Test() (Source.cpp:574-608): 0 pocketfft_2D, 1 pocketfft_1D, 3 pffft_, 4 fastboxblur; the OpenCV
// comparison branch (flag 2) is out of scope.  Prints the wall time of the call like the reference does.
#define BLUR_AMD_GLOBAL_NAMES
#include "blur_amd.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <flag 0|1|3|4> <sigma> <image.ppm>\n", argv[0]);
        return 2;
    }
    const int flag = std::atoi(argv[1]);
    const double nsmooth = std::atof(argv[2]);
    Image image = imread(argv[3]);
    if (image.empty()) {
        std::fprintf(stderr, "cannot read %s (binary PPM, P6, maxval 255)\n", argv[3]);
        return 1;
    }
    try {
        const auto t0 = std::chrono::steady_clock::now();
        switch (flag) {
        case 0: pocketfft_2D(image, nsmooth); break;
        case 1: pocketfft_1D(image, nsmooth); break;
        case 3: pffft_(image, nsmooth); break;
        case 4: fastboxblur(image.data, image.size[1], image.size[0], image.channels(), static_cast<int>(nsmooth * nsmooth), 2); break;   // Source.cpp:587
        default: std::fprintf(stderr, "flag %d: not on the accelerated path\n", flag); return 2;
        }
        std::printf("flag %d: %f\n", flag, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return imwrite(std::string(argv[3]) + ".out.ppm", image) ? 0 : 1;
}
