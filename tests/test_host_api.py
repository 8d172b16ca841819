"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol the
header declares, the host-side sizing / kernel / plan arithmetic equals the oracle (and the
reference vectors), the device FFT arithmetic run on the CPU equals a float64 DFT, and the C++
header with the reference's names compiles and behaves.  No GPU compute is called here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import blur_algorithms_amd as B
from blur_algorithms_amd import _lib
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFV = np.load(os.path.join(ROOT, "tests", "golden", "ref_host_functions.npz"))
HIPCC = "/opt/rocm/bin/hipcc"


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "blur_amd.h")).read()
    declared = set(re.findall(r"\b(blur_[a-z0-9_]+)\s*\(", header)) - {"blur_opts", "blur_ctx"}
    assert len(declared) >= 25
    lib = ctypes.CDLL(B.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)     # the Python binding covers the whole header


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this process has a GPU")
    with pytest.raises(B.BlurError) as e:
        B.BlurContext(0)
    assert e.value.code == 3


def test_host_sizing_equals_reference_vectors():
    got = [B.gaussian_window(float(s), int(m)) for s, m in zip(REFV["gw_sigma"], REFV["gw_maxw"])]
    assert got == REFV["gw_width"].tolist()
    n = REFV["size_n"]
    assert [B.isValidSize(int(i)) for i in n] == REFV["size_valid"].tolist()
    assert [B.nearestTransformSize(int(i)) for i in n] == REFV["size_nearest"].tolist()
    for i, (s, w, f) in enumerate(REFV["gk_cases"]):
        assert np.array_equal(B.getGaussian(float(s), int(w), int(f)), REFV["gk_%d" % i])


def test_host_sizing_and_multipliers_equal_oracle():
    for rows, cols, s in [(512, 512, 5), (1080, 1920, 20), (2160, 3840, 50), (2160, 3840, 20), (4320, 7680, 20), (101, 203, 4)]:
        assert B.pffft_sizing(rows, cols, s) == O.pffft_sizing(rows, cols, s)
    for s, k, n in [(5, 31, 576), (20, 131, 4000), (20, 131, 2304), (50, 331, 4320), (3, 19, 128)]:
        assert np.array_equal(B.kernel_multipliers(s, k, n), O.kernel_multipliers(s, k, n))


def test_box_kernel_and_boxfft_sizing_equal_oracle():
    for klen, n in [(9, 64), (25, 320), (4, 32), (49, 160)]:
        assert np.array_equal(B.box_kernel(klen, n), O.box_kernel_1d(klen, n))
    for rows, cols, ns in [(200, 300, 3.0), (50, 60, 9.0), (1080, 1920, 5.5), (33, 47, 2.0)]:
        assert B.boxfft_sizing(rows, cols, ns) == O.boxfft_sizing(rows, cols, ns)


def test_plans_cover_every_valid_length():
    supported = {2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16, 18, 20, 25}
    for n in range(32, 13000, 32):
        if not B.isValidSize(n):
            continue
        r = B.fft_plan_radices(n)
        assert r and int(np.prod(r)) == n and set(r) <= supported, (n, r)
    assert B.fft_plan_radices(7 * 32) == []
    # the BASELINE lengths use the three/four-pass compile-time plans
    assert len(B.fft_plan_radices(4000)) <= 4 and len(B.fft_plan_radices(2304)) == 3


@pytest.fixture(scope="module")
def engine_host_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cpp") / "engine_host_check")
    subprocess.check_call([HIPCC, "--cuda-host-only", "-O2", "-std=c++17", "-DBLUR_ENGINE_ALL_RADICES",
                           os.path.join(ROOT, "tests", "cpp", "engine_host_check.hip"),
                           os.path.join(ROOT, "blur_algorithms_amd", "csrc", "host_math.cpp"), "-o", exe],
                          stderr=subprocess.DEVNULL)
    return exe


def test_device_fft_arithmetic_on_cpu(engine_host_check):
    """the __host__ __device__ butterflies / passes / permuted multipliers of fft_engine.hpp, driven
    thread by thread on the CPU, against a float64 O(N^2) circular convolution"""
    out = subprocess.run([engine_host_check, "32", "96", "160", "288", "480", "576", "800", "1280"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    # explicit radix lists: the compile-time plans' butterflies (25, 18, 15, 12, 20)
    out = subprocess.run([engine_host_check, "-r", "16,10,25", "-r", "16,18,15", "-r", "20,20,10", "-r", "12,12,16", "-r", "16,10,5,5"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_fastboxblur_include_path_shim_compiles(tmp_path):
    """a translation unit that includes "FastBoxBlur/fast_box_blur.h" the way Source.cpp does (line 7)"""
    src = tmp_path / "tu.cpp"
    src.write_text('#include "FastBoxBlur/fast_box_blur.h"\n'
                   'int main() { float a[6] = {0, 1, 2, 3, 4, 5}, b[6]; flip_block<float, 1>(a, b, 3, 2); '
                   'void (*f)(uint8_t*, int, int, int, int, int, blur_ctx*) = &fastboxblur; return (b[1] == 3.f && f) ? 0 : 1; }\n')
    exe = str(tmp_path / "tu")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include", "compat"), "-I" + os.path.join(ROOT, "include"),
                           str(src), "-L" + os.path.dirname(B.LIB_PATH), "-lblur_amd", "-Wl,-rpath," + os.path.dirname(B.LIB_PATH),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    assert subprocess.run([exe]).returncode == 0


def test_pffft_include_path_shim(tmp_path):
    """include/compat/pffft_pommier/pffft.h used as Source.cpp uses pffft (lines 477-566): ordered real transforms against a float64
    DFT for a dozen valid lengths, the unscaled round trip, refused sizes, and one reflect-padded tile through forward -> the
    pointwise rule of Source.cpp:414-427 -> backward against the direct convolution (tests/cpp/pffft_shim_check.cpp)"""
    exe = str(tmp_path / "pffft_shim_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include", "compat"),
                           os.path.join(ROOT, "tests", "cpp", "pffft_shim_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "pffft shim ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("mode", [["-fopenmp"], ["-DMYLOOP", "-pthread"], []])
def test_cpp_header_with_reference_names(mode):
    """include/blur_amd.hpp under the reference's three hybrid_loop build modes (Utils.hpp:24-54): OpenMP, -DMYLOOP
    (std::thread blocks: the published build, .vscode/tasks.json:21) and serial"""
    exe = os.path.join(ROOT, "tests", "cpp", "surface_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1"] + mode + ["-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "surface_check.cpp"),
                           "-L" + os.path.dirname(B.LIB_PATH), "-lblur_amd", "-Wl,-rpath," + os.path.dirname(B.LIB_PATH),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe, "host"], capture_output=True, text=True)
    assert out.returncode == 0 and "host ok" in out.stdout, out.stdout + out.stderr


def _reflect_tile(x, pad, n):
    """reflect-101 padded tile of Source.cpp:525-529, zero-extended to n"""
    L = len(x)
    t = np.zeros(n)
    t[:pad] = x[pad:0:-1]
    t[pad:pad + L] = x
    t[pad + L:pad + L + pad] = x[L - 2:L - 2 - pad:-1]
    return t


@pytest.mark.parametrize("L,sigma", [(2160, 20.0), (3840, 20.0), (700, 7.0), (300, 2.0)])
@pytest.mark.parametrize("quirk", [1, 0])
def test_wr_multipliers_reproduce_the_reference_length(L, sigma, quirk):
    """The wave-resident kernels transform at n = 256 * R0 instead of the reference's nearestTransformSize().  In the
    cropped region the result must not change: the convolution part is length-independent and the Nyquist-slot term
    of Source.cpp:420-425 (which does depend on the reference's length) is carried by the multiplier of bin n/2."""
    from blur_algorithms_amd import _lib
    lib = _lib.load()
    ks = lib.blur_gaussian_window(sigma, L)
    pad = (ks - 1) // 2
    n_ref = lib.blur_nearest_transform_size(L + 2 * pad) if not lib.blur_is_valid_size(L + 2 * pad) else L + 2 * pad
    x = np.random.default_rng(5).integers(0, 256, L).astype(np.float64)
    # the reference's own arithmetic at its own length: bins 0..n/2, slot 1 (Nyquist) scaled with bin 0's factor
    m = np.zeros(n_ref // 2 + 1, np.float32)
    assert lib.blur_kernel_multipliers(sigma, ks, n_ref, m.ctypes.data) == 0
    full = np.concatenate([m, m[-2:0:-1]]).astype(np.float64)
    if quirk:
        full[n_ref // 2] = m[0]
    want = np.fft.ifft(np.fft.fft(_reflect_tile(x, pad, n_ref)) * full).real[pad:pad + L] * n_ref
    for n in sorted({256 * r for r in (3, 4, 5, 9, 10, 16, 18, 20)}):
        if n < L + 2 * pad:
            continue
        mm = np.zeros(n, np.float32)
        assert lib.blur_wr_kernel_multipliers(sigma, ks, n, n_ref, quirk, mm.ctypes.data) == 0
        assert np.array_equal(mm[1:], mm[:0:-1])            # even: two real lines may share a complex line
        got = np.fft.ifft(np.fft.fft(_reflect_tile(x, pad, n)) * mm.astype(np.float64)).real[pad:pad + L] * n
        assert np.abs(got - want).max() < 3e-5, (n, np.abs(got - want).max())    # float32 rounding of the two tables


@pytest.mark.parametrize("sigma,n", [(20.0, 4000), (20.0, 2304), (5.0, 576), (50.0, 4320), (50.0, 2560), (2.0, 1280), (3.0, 96)])
def test_multiplier_model_against_a_float32_transform_of_the_kernel(sigma, n):
    """The engine computes the kernel spectrum as an exact cosine sum rounded to float once and scales it with the float
    1.f/N (host_math.cpp: kernel_multipliers); the reference takes pffft's FLOAT32 transform of the same taps and scales
    that (Source.cpp:485,423).  pffft is absent, so the assumption is bounded against the oracle's own float32
    real transform (ordered layout, Source.cpp:420-425: multiplier i = kerf[2i] * scaler): the two tables differ by at
    most 2 ULP of the DC multiplier (observed 1.0 .. 2.0), the imaginary parts the reference drops are below that too."""
    from oracle import oracle as O
    from blur_algorithms_amd import _lib
    lib = _lib.load()
    ks = lib.blur_gaussian_window(sigma, 1 << 20)
    kerf = O.RealFFT(n).transform_ordered(O.get_gaussian(sigma, ks, n))
    scaler = np.float32(1.0) / np.float32(n)
    port = (kerf[0::2] * scaler).astype(np.float32)            # bins 0 .. n/2 - 1 (slot 1 holds the Nyquist bin)
    m = np.zeros(n // 2 + 1, np.float32)
    assert lib.blur_kernel_multipliers(sigma, ks, n, m.ctypes.data) == 0
    ulp = float(np.spacing(np.float32(m[0])))
    assert np.abs(m[:n // 2].astype(np.float64) - port.astype(np.float64)).max() <= 3 * ulp
    assert np.abs(kerf[3::2].astype(np.float64)).max() * float(scaler) <= 3 * ulp
    # the wave-resident tables at the same length are the same numbers in natural order (quirk: bin n/2 <- bin 0's rule)
    if n % 256 == 0:
        mm = np.zeros(n, np.float32)
        assert lib.blur_wr_kernel_multipliers(sigma, ks, n, n, 0, mm.ctypes.data) == 0
        assert np.abs(mm[:n // 2 + 1].astype(np.float64) - m.astype(np.float64)).max() <= 2 * ulp


def test_python_image_files_round_trip(tmp_path):
    """blur_algorithms_amd.io: decode / encode either side of the blur (cv::imread / cv::imwrite in main(), Source.cpp:623,635)"""
    from blur_algorithms_amd import io
    img = np.random.default_rng(2).integers(0, 256, (37, 53, 3), dtype=np.uint8)
    for ext in ("png", "ppm", "bmp"):
        p = str(tmp_path / ("x." + ext))
        io.imwrite(p, img)
        assert np.array_equal(io.imread(p), img)


def test_example_main_compiles(tmp_path):
    """examples/blur_main.cpp: the reference's main() over the engine (compile and usage message only: no GPU here)"""
    exe = str(tmp_path / "blur_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "blur_main.cpp"),
                           "-L" + os.path.dirname(B.LIB_PATH), "-lblur_amd", "-Wl,-rpath," + os.path.dirname(B.LIB_PATH),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    assert subprocess.run([exe], capture_output=True).returncode == 2


def test_pocketfft2d_sizing_matches_the_restatement():
    """blur_pocketfft2d_sizing (Source.cpp:149-176) against oracle/pocketfft_paths._sizes_2d, including the float
    arithmetic of `border += new_pad / 2.f + 0.5f` for odd extra padding"""
    import itertools
    from blur_algorithms_amd.api import pocketfft2d_sizing
    from oracle import pocketfft_paths as P
    odd_extra = 0
    for r, c, s in itertools.product(range(20, 400, 37), range(25, 500, 41), (0.8, 2.0, 5.5, 13.0)):
        k, p, sz, b = P._sizes_2d(r, c, s)
        q = pocketfft2d_sizing(r, c, s)
        assert (q["kSize"], q["pad"], list(q["sizes"]), list(q["border"])) == (k, p, sz, b)
        assert sz[0] == r + b[0] + b[1] and sz[1] == c + b[2] + b[3]
        odd_extra += (b[1] != b[0]) + (b[3] != b[2])
    assert odd_extra > 10


def test_dft_image_restatement_is_self_consistent():
    """oracle/pocketfft_paths.dft_image_u8c3: DC at the fftshift centre, float32 and float64 transforms agree where the
    logarithm is well conditioned, and the directly read half of every row is the plain fftshift of Re F"""
    import scipy.fft as sfft
    from oracle import oracle as O
    from oracle import pocketfft_paths as P
    img = np.random.default_rng(2).integers(0, 256, (90, 120, 3), dtype=np.uint8)
    u8, l64, mags = P.dft_image_u8c3(img, 3.0, np.float64)
    _, l32, _ = P.dft_image_u8c3(img, 3.0, np.float32)
    ok = mags > 1e-5 * mags.max()
    assert ok.mean() > 0.9 and np.abs(l32 - l64)[ok].max() < 0.01
    _, _, sizes, border = P._sizes_2d(90, 120, 3.0)
    i, j = np.unravel_index(np.argmax(l64[1]), l64[1].shape)
    assert (i + border[0], j + border[2]) == (sizes[0] // 2, sizes[1] // 2)
    padded = O.reflect_101(img, *border)
    full = np.fft.fftshift(np.real(sfft.fft2(padded[..., 1].astype(np.float64))))
    want = 20 * np.log10(np.abs(full) + 1e-5)[border[0]:sizes[0] - border[1], border[2]:sizes[1] - border[3]]
    # fftshift puts frequency 0 at s1/2: displayed columns [s1/2, s1) hold frequencies 0..s1/2-1 (read directly),
    # displayed columns [0, s1/2) hold negative frequencies, which the reference reads mirrored in the SAME row
    direct = slice(sizes[1] // 2 - border[2], 120)
    assert np.abs(l64[1][:, direct] - want[:, direct])[ok[1][:, direct]].max() < 1e-3


def test_mx_fragments_are_the_toeplitz_band_in_two_halves():
    """blur_mx_fragments (host_math.cpp): binary16 conversion equals numpy's round-to-nearest-even, hi + lo carries the
    tap to 2^-22, and the fragment layout is Tz[16 kb + 8 (l >> 5) + j][l & 31] = taps[w - o - PADA + pad]"""
    from blur_algorithms_amd._lib import load
    lib = load()
    rng = np.random.default_rng(3)
    for pad, nkb in ((65, 11), (58, 11), (3, 3), (16, 4), (0, 2)):
        taps = (rng.random(2 * pad + 1) ** 4).astype(np.float32)
        taps[0] = 1e-7                                     # a subnormal binary16 after scaling
        out = np.zeros((2, nkb, 64, 8), np.uint16)
        assert lib.blur_mx_fragments(taps.ctypes.data, pad, nkb, out.ctypes.data) == 0
        fr = out.view(np.float16).astype(np.float64)
        pada = 8 * (nkb - 2)
        tz = np.zeros((16 * nkb, 32))
        for w in range(16 * nkb):
            for o in range(32):
                t = w - o - pada
                if -pad <= t <= pad:
                    tz[w, o] = taps[t + pad]
        for kb in range(nkb):
            for l in range(64):
                want = tz[16 * kb + 8 * (l >> 5):16 * kb + 8 * (l >> 5) + 8, l & 31] * 16384.0
                hi = want.astype(np.float32).astype(np.float16)
                assert np.array_equal(out[0, kb, l].view(np.float16), hi)
                lo = (want.astype(np.float32) - hi.astype(np.float32)).astype(np.float16)
                assert np.array_equal(out[1, kb, l].view(np.float16), lo)
                assert np.all(np.abs(fr[0, kb, l] + fr[1, kb, l] - want) <= np.maximum(np.abs(want) * 2.0 ** -21, 2.0 ** -24))
    assert lib.blur_mx_fragments(taps.ctypes.data, 65, 10, out.ctypes.data) != 0      # window too small for the taps
    assert lib.blur_mx_window_blocks(65) == 11


def test_bench_model_of_the_fused_launch():
    """bench.py prices the fused kernel's `roofline` on the matrix instructions it executes; its model of the launch (segments of any
    number of tiles, the first NT steps of a segment without the column products of tiles above it) must agree with the kernel:
    pinned to SQ_INSTS_MFMA of the committed counter profile of the metric's launch (profiles/r04fx_sq_counters.json)"""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    shape = bench.fused_launch_shape(2160, 3840, 65, 8)
    assert (shape["nkb"], shape["tasks"], shape["segments_per_strip"], shape["steps_per_strip"]) == (11, 240, 1, 73)
    counters = json.load(open(os.path.join(root, "profiles", "r04fx_sq_counters.json")))
    counters = counters.get("fx_blur_u8", counters)
    assert shape["mfma_instructions"] == int(counters["SQ_INSTS_MFMA"])
    assert shape["flops"] == shape["mfma_instructions"] * 32768
    # eight 1080p frames: 120 strips cut in two segments of 17 tiles (22 steps each), not 20 + 14
    c2 = bench.fused_launch_shape(1080, 1920, 65, 8)
    assert (c2["tasks"], c2["segments_per_strip"], c2["steps_per_strip"]) == (240, 2, 44)
    # one 4K frame: 30 strips in 8 segments of 9 / 5 tiles; the wide kernels: one channel per task
    one = bench.fused_launch_shape(2160, 3840, 65, 1)
    assert one["tasks"] == 240 and one["segments_per_strip"] == 8
    wide = bench.fused_launch_shape(2160, 3840, 165, 8)
    assert wide["nkb"] == 23 and wide["tasks"] == 720


def test_staging_addresses_of_the_fused_kernel_stay_inside_their_buffers():
    """tools/fx_staging_addresses.py mirrors fx_blur_u8's staging loads (strip / image resources, fx_strip_range, the clamp of the
    lanes past the window): over its list of shapes and window sizes no load may fall outside the frame or the written part of a
    strip (round 4: a clamp that pointed before the frame hung the GPU test of the first version)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fx_staging_addresses.py")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip().splitlines()[-1] == "bad 0", p.stdout[-500:] + p.stderr[-500:]
