"""CPU tests of the oracle itself: pinned against the reference's own code (committed golden
vectors generated from oracle/_ref, and oracle/_ref live when it is present), the known-answer
values of SURVEY.md 8(c) / README.md, and an independent numpy FFT."""
import os

import numpy as np
import pytest

from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REFV = np.load(os.path.join(G, "ref_host_functions.npz"))


# ---- known answers ---------------------------------------------------
def test_gaussian_window_known_answers():
    assert O.gaussian_window(5) == 31 and O.gaussian_window(20) == 131 and O.gaussian_window(50) == 331
    assert O.gaussian_window(1500 ** 0.5) == 257 and O.gaussian_window(11400 ** 0.5) == 709
    assert O.gaussian_window(200, 1024) == 1025           # clamped to max_width, then made odd


def test_nearest_transform_size_known_answers():
    want = {542: 576, 2050: 2304, 1210: 1280, 4170: 4320, 2490: 2560, 3970: 4000, 2290: 2304, 7810: 8000,
            4450: 4608, 1154: 1280, 8308: 8640, 12108: 12288, 1: 32}
    for n, v in want.items():
        assert O.nearest_transform_size(n) == v
    valid = [n for n in range(1, 2049) if O.is_valid_size(n)]
    assert valid == [32, 64, 96, 128, 160, 192, 256, 288, 320, 384, 480, 512, 576, 640, 768, 800, 864, 960, 1024, 1152,
                     1280, 1440, 1536, 1600, 1728, 1920, 2048]


def test_sizing_table_of_the_baseline_configs():
    # BASELINE.md section 3
    assert O.pffft_sizing(512, 512, 5) == dict(kSize=31, pad=15, N0=576, N1=576, tz0=34, tz1=34)
    assert O.pffft_sizing(1080, 1920, 20) == dict(kSize=131, pad=65, N0=1280, N1=2304, tz0=70, tz1=254)
    assert O.pffft_sizing(2160, 3840, 50) == dict(kSize=331, pad=165, N0=2560, N1=4320, tz0=70, tz1=150)
    assert O.pffft_sizing(2160, 3840, 20) == dict(kSize=131, pad=65, N0=2304, N1=4000, tz0=14, tz1=30)


def test_readme_reflect_example():
    # README.md:49-52: length 7, kernel 13, pad 6:  g f e d c b | A B C D E F G | f e d c b a
    x = np.arange(1, 8, dtype=np.float32)
    t = O.pad_tile(x, 6, 32)
    assert t[:19].tolist() == [7, 6, 5, 4, 3, 2, 1, 2, 3, 4, 5, 6, 7, 6, 5, 4, 3, 2, 1]
    assert not t[19:].any()


def test_readme_kernel_centering_example():
    # README.md:93-101: a 3-tap kernel in an FFT length of 8 -> k0 k1 0 0 0 0 0 k(-1)
    k = O.get_gaussian(1.0, 3, 8)
    raw = O.get_gaussian(1.0, 3, 0)
    assert k.tolist() == [raw[1], raw[2], 0, 0, 0, 0, 0, raw[0]]
    assert abs(float(raw.sum()) - 1) < 1e-6 and raw[0] == raw[2]


# ---- pinned by the reference's own code (committed vectors) -------------------------------
def test_gaussian_window_matches_reference_vectors():
    got = [O.gaussian_window(float(s), int(m)) for s, m in zip(REFV["gw_sigma"], REFV["gw_maxw"])]
    assert got == REFV["gw_width"].tolist()


def test_transform_sizes_match_reference_vectors():
    n = REFV["size_n"]
    assert [O.is_valid_size(int(i)) for i in n] == REFV["size_valid"].tolist()
    assert [O.nearest_transform_size(int(i)) for i in n] == REFV["size_nearest"].tolist()


def test_get_gaussian_bit_exact_against_reference_vectors():
    for i, (s, w, f) in enumerate(REFV["gk_cases"]):
        got = O.get_gaussian(float(s), int(w), int(f))
        assert np.array_equal(got, REFV["gk_%d" % i]), (s, w, f)


def test_de_interleave_match_reference_vectors():
    assert np.array_equal(O.deinterleave_bgr(REFV["di_img"]), REFV["di_planes"])
    assert np.array_equal(O.interleave_bgr(REFV["il_planes"]), REFV["il_out"])


def test_reflect_101_matches_reference_vectors():
    for i in range(int(REFV["rf_n"])):
        pads = REFV["rf_pads_%d" % i].tolist()
        got = O.reflect_101(REFV["rf_in_%d" % i], *pads)
        assert np.array_equal(got, REFV["rf_out_%d" % i]), i


def test_pointwise_rule_matches_the_reference_vectors():
    """pffft_sorted_optimized_convolution (Source.cpp:414-427) is the one place where the reference's TEXT, not mathematics,
    defines the result (slot 1 of the ordered layout, the Nyquist bin, is scaled with the DC gain kernel_dft[0]): the oracle's
    restatement against outputs of the reference's own function (compiled from where it lies, oracle/Makefile), bit for bit"""
    V = np.load(os.path.join(G, "ref_host_functions.npz"))
    for i in range(int(V["soc_n"])):
        tile, kern, want = V["soc_tile_%d" % i], V["soc_kernel_%d" % i], V["soc_out_%d" % i]
        n = tile.size
        got = O.sorted_optimized_convolution(tile, kern, 1.0 / n)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        m0 = np.float32(kern[0] * np.float32(1.0 / n))
        assert want[1] == np.float32(tile[1] * m0) and want[0] == np.float32(tile[0] * m0)       # the quirk: slot 1 gets slot 0's multiplier
        assert want[3] == np.float32(tile[3] * np.float32(kern[2] * np.float32(1.0 / n)))        # imaginary parts get the real part's


# ---- live against oracle/_ref (present in the build container and, prebuilt, on the GPU box)
needs_ref = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref/libref_utils.so not built")


@needs_ref
def test_live_reference_sizing_and_kernel():
    rng = np.random.default_rng(1)
    for s in rng.uniform(0.05, 400, 300):
        for mw in (0, int(rng.integers(1, 9000))):
            assert O.gaussian_window(float(s), mw) == O.ref().ref_gaussian_window(float(s), mw)
    for n in rng.integers(0, 40000, 2000):
        assert O.nearest_transform_size(int(n)) == O.ref().ref_nearest_transform_size(int(n))
        assert O.is_valid_size(int(n)) == O.ref().ref_is_valid_size(int(n))
    for s, w, f in [(rng.uniform(0.3, 60), 0, 0) for _ in range(20)] + [(20.0, 131, 2304), (2.2, 13, 64), (50.0, 331, 2560)]:
        assert np.array_equal(O.get_gaussian(s, w, f), O.ref_get_gaussian(s, w, f))


@needs_ref
def test_live_reference_pointwise_rule():
    rng = np.random.default_rng(5)
    for n in (32, 160, 1280, 4320):
        tile = (rng.standard_normal(n) * 50).astype(np.float32)
        kern = rng.uniform(-1, 1, n).astype(np.float32)
        a = O.sorted_optimized_convolution(tile, kern, 1.0 / n)
        b = O.ref_sorted_optimized_convolution(tile, kern, 1.0 / n)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@needs_ref
def test_live_reference_hybrid_loop_and_reflect():
    hits = np.zeros(1000, np.int32)
    O.ref().ref_hybrid_loop_count(1000, hits)      # hybrid_loop visits every index exactly once (Utils.hpp:16-55)
    assert (hits == 1).all()
    rng = np.random.default_rng(2)
    for _ in range(20):
        r, c = int(rng.integers(2, 40)), int(rng.integers(2, 40))
        img = rng.integers(0, 256, (r, c, 3), dtype=np.uint8)
        pads = [int(v) for v in rng.integers(0, 50, 4)]
        assert np.array_equal(O.reflect_101(img, *pads), O.ref_reflect_101(img, *pads))


# ---- the FFT part (parity unpinned): independent cross-checks --------------------------------
@pytest.mark.parametrize("n", [32, 96, 160, 576, 1280, 2304, 4000])
def test_fft64_against_numpy(n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    assert np.abs(O.fft64(x) - np.fft.fft(x)).max() < 1e-11 * n
    assert np.abs(O.fft64(x, True) - np.fft.ifft(x) * n).max() < 1e-11 * n


@pytest.mark.parametrize("n", [32, 64, 96, 160, 576, 2304, 4000])
def test_port_real_fft_has_pffft_ordered_layout(n):
    """[F0.re, F(N/2).re, F1.re, F1.im, ...] forward; unnormalised backward"""
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32)
    F = np.fft.rfft(x.astype(np.float64))
    f = O.RealFFT(n)
    o = f.transform_ordered(x)
    want = np.empty(n)
    want[0], want[1] = F[0].real, F[n // 2].real
    want[2::2], want[3::2] = F[1:n // 2].real, F[1:n // 2].imag
    assert np.abs(o - want).max() < 2e-6 * np.abs(want).max() * np.log2(n)
    back = f.transform_ordered(o, True)
    assert np.abs(back / n - x).max() < 2e-6


def test_pointwise_rule_scales_nyquist_slot_with_dc_gain():
    """Source.cpp:420-425: i = 0 multiplies slot 0 (DC) AND slot 1 (Nyquist) by kernel_dft[0]*scaler"""
    t = np.arange(1, 9, dtype=np.float32)
    k = np.array([2, 100, 3, 100, 5, 100, 7, 100], np.float32)   # only the even (real) slots are read
    out = O.sorted_optimized_convolution(t, k, 0.5)
    assert out.tolist() == [1 * 1.0, 2 * 1.0, 3 * 1.5, 4 * 1.5, 5 * 2.5, 6 * 2.5, 7 * 3.5, 8 * 3.5]


def _numpy_pipeline(img, sigma, quirk):
    rows, cols, _ = img.shape
    s = O.pffft_sizing(rows, cols, sigma)
    pad = s["pad"]

    def one_pass(pl, n, length):
        m = O.kernel_multipliers(sigma, s["kSize"], n).astype(np.float64)
        if quirk:
            m[n // 2] = m[0]
        t = np.pad(pl.astype(np.float64), ((0, 0), (pad, pad)), mode="reflect")
        t = np.pad(t, ((0, 0), (0, n - t.shape[1])))
        y = np.fft.irfft(np.fft.rfft(t, axis=1) * m[None, :], n=n, axis=1) * n
        return y[:, pad:pad + length].astype(np.float32)

    out = np.empty((3, rows, cols), np.float32)
    for c in range(3):
        a = one_pass(img[:, :, c].astype(np.float32), s["N1"], cols)
        out[c] = one_pass(np.ascontiguousarray(a.T), s["N0"], rows).T
    return out


@pytest.mark.parametrize("rows,cols,sigma", [(64, 96, 3.0), (100, 77, 5.0), (135, 240, 20.0)])
@pytest.mark.parametrize("quirk", [True, False])
def test_f64_oracle_against_independent_numpy_pipeline(rows, cols, sigma, quirk):
    img = np.random.default_rng(3).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    _, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk, want_planes=True)
    ref = _numpy_pipeline(img, sigma, quirk)
    assert np.abs(planes.astype(np.float64) - ref).max() < 2e-5


@pytest.mark.parametrize("rows,cols,sigma", [(90, 140, 6.0), (270, 480, 20.0), (101, 77, 4.5), (64, 64, 3.0)])
@pytest.mark.parametrize("path", ["pocketfft_1d_u8c3", "pocketfft_2d_u8c3"])
def test_oracle_without_quirk_equals_the_pocketfft_paths(rows, cols, sigma, path):
    """pocketfft_1D (Source.cpp:280-392) and pocketfft_2D (Source.cpp:143-277) restated on scipy.fft -- the pocketfft
    library the reference calls -- against this repo's own FFT restatement with the Nyquist quirk off: the three paths
    are one linear convolution.  float64 transforms agree to the float32 rounding of the returned planes; pocketfft in
    float32 (what the reference runs) sits within the engine's tolerance of both."""
    from oracle import pocketfft_paths as P
    img = np.random.default_rng(rows * 7 + cols).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, False, want_planes=True)
    u64, p64 = getattr(P, path)(img, sigma, np.float64, want_planes=True)
    assert np.abs(p64.astype(np.float64) - planes).max() <= 3.1e-5          # two float32 roundings of values < 256
    u32, p32 = getattr(P, path)(img, sigma, np.float32, want_planes=True)
    assert np.abs(p32.astype(np.float64) - planes).max() <= 1.5e-4
    for u in (u64, u32):
        d = u.astype(int) - want.astype(int)
        assert np.abs(d).max() <= 1 and (d != 0).mean() < 2e-3


def test_f32_port_agrees_with_f64_oracle():
    img = np.random.default_rng(4).integers(0, 256, (270, 480, 3), dtype=np.uint8)
    a = O.pffft_blur_u8c3_f32(img, 20.0)
    b = O.pffft_blur_u8c3_f64(img, 20.0, True)
    d = a.astype(int) - b.astype(int)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 1e-3


def test_constant_image_and_nyquist_term():
    """without the quirk a constant image is a fixed point; with it the reference adds a
    +-(sum of the alternating padded tile)/N checkerboard (Source.cpp:420-425)"""
    img = np.full((40, 56, 3), 200, np.uint8)
    assert (O.pffft_blur_u8c3_f64(img, 4.0, False) == 200).all()
    noisy = np.random.default_rng(5).integers(0, 256, (64, 64, 3), dtype=np.uint8)
    assert (O.pffft_blur_u8c3_f64(noisy, 3.0, True) != O.pffft_blur_u8c3_f64(noisy, 3.0, False)).mean() > 0.5


def test_pad_larger_than_image_is_refused():
    with pytest.raises(ValueError):
        O.pffft_blur_u8c3_f64(np.zeros((4, 64, 3), np.uint8), 20.0)


@pytest.mark.parametrize("name", ["colourgram", "collage_top", "baseline", "input7"])
def test_oracle_reproduces_committed_image_vectors(name):
    v = np.load(os.path.join(G, "img_%s.npz" % name))
    got, planes = O.pffft_blur_u8c3_f64(v["src"], float(v["sigma"]), True, want_planes=True)
    assert np.array_equal(got, v["oracle_u8"])
    assert np.abs(planes - v["oracle_planes"]).max() < 1e-5
    assert np.array_equal(O.pffft_blur_u8c3_f64(v["src"], float(v["sigma"]), False), v["oracle_u8_noquirk"])


def test_fastboxblur_oracle_properties():
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, (30, 44, 3), dtype=np.uint8)
    assert np.array_equal(O.fastboxblur_u8(img, 1, 3), img)                 # width-1 box is the identity
    flat = np.full((20, 20, 3), 77, np.uint8)
    assert (O.fastboxblur_u8(flat, 9, 2) == 77).all()
    # one horizontal sweep of width 3 against a direct reflect-101 mean
    row = rng.integers(0, 256, (1, 16, 1), dtype=np.uint8)
    got = O.fastboxblur_u8(np.repeat(row, 3, 0), 3, 1)                       # 3 identical rows: vertical sweep is a no-op
    p = np.pad(row[0, :, 0].astype(np.float32), 1, mode="reflect")
    want = ((p[:-2] + p[1:-1] + p[2:]) * np.float32(1 / 3) + np.float32(0.5)).astype(np.uint8)
    assert np.array_equal(got[1, :, 0], want)
