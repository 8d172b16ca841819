"""GPU tests of the wave-resident kernels (csrc/wr_kernels.hpp): pass 0 through LDS, 256-point sub-transforms on
16 lanes x 16 registers with DPP / permlane-swap transposes, columns first.  Against numpy's FFT for the line
engine and against the float64 oracle of pffft_() (Source.cpp:429-570) for whole images."""
import numpy as np
import pytest

from conftest import assert_u8_parity

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _rand_img(rows, cols, seed):
    return np.random.default_rng(seed).integers(0, 256, (rows, cols, 3), dtype=np.uint8)


@pytest.mark.parametrize("n", [768, 1024, 1280, 1536, 2048, 2304, 2560, 3072, 3840, 4096])
@pytest.mark.parametrize("nlines", [1, 7, 400])
def test_convolve_lines_against_numpy(ctx, n, nlines):
    """IDFT(m * DFT(x)) of complex lines: every butterfly, twiddle, transpose and the natural-order multiplier lookup"""
    torch = _torch()
    rng = np.random.default_rng(n + nlines)
    x = (rng.standard_normal((nlines, n)) + 1j * rng.standard_normal((nlines, n))).astype(np.complex64)
    m = rng.standard_normal(n).astype(np.float32)
    got = ctx.convolve_lines(torch.from_numpy(x).cuda(), m).cpu().numpy()
    want = np.fft.ifft(np.fft.fft(x.astype(np.complex128), axis=1) * m.astype(np.float64), axis=1) * n
    scale = np.abs(want).max()
    err = np.abs(got - want).max() / scale
    assert err < 2e-6, err
    # in place
    t = torch.from_numpy(x).cuda()
    ctx.convolve_lines(t, m, out=t)
    assert np.array_equal(t.cpu().numpy(), got)


def test_convolve_lines_impulse_positions(ctx):
    """a unit impulse at every position class: catches any index mix-up that random data would only show as 'wrong'"""
    torch = _torch()
    for n in (2304, 4096):
        pos = [0, 1, 15, 16, 17, 255, 256, 257, n // 2, n - 1, 1000]
        x = np.zeros((len(pos), n), np.complex64)
        for i, p in enumerate(pos):
            x[i, p] = 1 + 2j
        m = np.cos(2 * np.pi * 3 * np.arange(n) / n).astype(np.float32) + 2       # kernel 2 delta[0] + (delta[3] + delta[-3]) / 2
        got = ctx.convolve_lines(torch.from_numpy(x).cuda(), m).cpu().numpy() / n
        for i, p in enumerate(pos):
            want = np.zeros(n, np.complex128)
            want[p] += 2 * (1 + 2j)
            want[(p + 3) % n] += 0.5 * (1 + 2j)
            want[(p - 3) % n] += 0.5 * (1 + 2j)
            assert np.abs(got[i] - want).max() < 1e-5, (n, p)


# small and odd images forced through the wave-resident kernels (N = 2304 / 4096 whatever the image): ragged last strip,
# single last row, unaligned rows (byte path), pad parity both ways, fewer units than CUs
SMALL = [
    (64, 96, 3.0),
    (33, 47, 2.0),
    (100, 77, 5.0),
    (101, 203, 4.0),
    (270, 480, 20.0),
    (500, 748, 20.0),
    (16, 16, 1.0),
    (301, 8, 1.2),
    (700, 900, 90.0),     # pad 299 > 256: more than one edge round at each end (the general instantiation of both kernels)
    (640, 331, 60.0),
]


@pytest.mark.parametrize("rows,cols,sigma", SMALL)
@pytest.mark.parametrize("quirk", [True, False])
def test_small_images_forced_wave_resident(ctx, rows, cols, sigma, quirk):
    torch = _torch()
    from oracle import oracle as O
    img = _rand_img(rows, cols, 3)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk, want_planes=True)
    t = torch.from_numpy(img).cuda()
    got = ctx.pffft_(t, sigma, out=torch.empty_like(t), nyquist_quirk=quirk, wave_resident=True)
    assert_u8_parity(got.cpu().numpy(), want, planes)
    # in place, like the reference
    ctx.pffft_(t, sigma, nyquist_quirk=quirk, wave_resident=True)
    assert torch.equal(t, got)


# thin images whose long side is the metric's: the real transform lengths with the real pad, cheap for the oracle
THIN = [
    (2160, 70, 20.0),
    (2159, 67, 20.0),
    (70, 3840, 20.0),
    (71, 3839, 20.0),
    (2174, 66, 20.0),     # rows + 2 pad = 2304 exactly: no zero tail
    (66, 3966, 20.0),     # cols + 2 pad = 4096 exactly
]


@pytest.mark.parametrize("rows,cols,sigma", THIN)
def test_metric_lengths_forced_wave_resident(ctx, rows, cols, sigma):
    torch = _torch()
    from oracle import oracle as O
    img = _rand_img(rows, cols, 4)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, True, want_planes=True)
    t = torch.from_numpy(img).cuda()
    got = ctx.pffft_(t, sigma, out=torch.empty_like(t), wave_resident=True)
    assert_u8_parity(got.cpu().numpy(), want, planes)


def test_batch_equals_single_frames_wave_resident(ctx):
    torch = _torch()
    frames = np.stack([_rand_img(120, 200, 40 + i) for i in range(5)])
    t = torch.from_numpy(frames).cuda()
    got = ctx.pffft_(t, 6.0, out=torch.empty_like(t), wave_resident=True)
    for i in range(5):
        one = ctx.pffft_(t[i].contiguous(), 6.0, out=torch.empty_like(t[i]), wave_resident=True)
        assert torch.equal(got[i], one)
    # frames per launch is a scheduling choice
    again = ctx.pffft_(t, 6.0, out=torch.empty_like(t), wave_resident=True, frames_per_launch=2)
    assert torch.equal(got, again)


def test_wave_resident_is_the_fft_choice_on_the_metric_frame(ctx):
    """4K sigma 20: among the FFT kernels (engine="fft") the wave-resident family is chosen; switching it off gives the
    rows-first kernels.  The library's own default is the fused matrix-core kernel.  All of them obey the parity contract; they
    differ from each other only at rounding ties."""
    torch = _torch()
    from oracle import oracle as O
    img = _rand_img(2160, 3840, 99)
    want, planes = O.pffft_blur_u8c3_f64(img, 20.0, True, want_planes=True)
    t = torch.from_numpy(img).cuda()
    a = ctx.pffft_(t, 20.0, out=torch.empty_like(t), engine="fft")
    b = ctx.pffft_(t, 20.0, out=torch.empty_like(t), wave_resident=True)
    assert torch.equal(a, b)
    assert_u8_parity(a.cpu().numpy(), want, planes)
    c = ctx.pffft_(t, 20.0, out=torch.empty_like(t), wave_resident=False)
    assert_u8_parity(c.cpu().numpy(), want, planes)
    assert (a != c).float().mean().item() < 1e-4
    d = ctx.pffft_(t, 20.0, out=torch.empty_like(t))
    e = ctx.pffft_(t, 20.0, out=torch.empty_like(t), engine="fused")
    assert torch.equal(d, e)
    assert_u8_parity(d.cpu().numpy(), want, planes)
    assert (a != d).float().mean().item() < 1e-4
    m = ctx.pffft_(t, 20.0, out=torch.empty_like(t), engine="matrix")
    assert_u8_parity(m.cpu().numpy(), want, planes)
    assert (m != d).float().mean().item() < 1e-4


# (column role R0 = 12, 15, 16: the C = 2 kernels of round 4, strips of 4 columns)
@pytest.mark.parametrize("role,r0", [("col", r) for r in (3, 4, 5, 6, 8, 9, 10, 12, 15, 16)] + [("row", r) for r in (3, 4, 5, 6, 8, 9, 10, 12, 15, 16)])
@pytest.mark.parametrize("sigma", [3.0, 20.0])
def test_every_registered_length_in_its_role(ctx, role, r0, sigma):
    """a thin image whose long side lands on N = 256 * R0 in the column or the row role (the other side takes the smallest
    transform), odd sizes: every instantiated kernel runs with reflected borders, a zero tail and a ragged strip / last pair"""
    torch = _torch()
    import blur_algorithms_amd as B
    from oracle import oracle as O
    pad = B.pffft_sizing(4096, 4096, sigma)["pad"]
    long_side = 256 * r0 - 2 * pad - 5
    if long_side <= pad + 1:
        pytest.skip("the transform is shorter than the kernel")
    short = pad + 13
    rows, cols = (long_side, short) if role == "col" else (short, long_side)
    lib = B._lib.load()
    assert lib.blur_wr_length(long_side + 2 * pad, 1 if role == "col" else 0) == 256 * r0
    img = _rand_img(rows, cols, 100 + r0)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, True, want_planes=True)
    t = torch.from_numpy(img).cuda()
    got = ctx.pffft_(t, sigma, out=torch.empty_like(t), wave_resident=True)
    assert_u8_parity(got.cpu().numpy(), want, planes)


def test_config2_frame_1080p_wave_resident(ctx):
    """BASELINE config 2 whole (1920x1080, sigma 20): column 5 x 256, row 9 x 256"""
    torch = _torch()
    from oracle import oracle as O
    img = _rand_img(1080, 1920, 12)
    want, planes = O.pffft_blur_u8c3_f64(img, 20.0, True, want_planes=True)
    t = torch.from_numpy(img).cuda()
    got = ctx.pffft_(t, 20.0, out=torch.empty_like(t), wave_resident=True)
    assert_u8_parity(got.cpu().numpy(), want, planes)


@pytest.mark.parametrize("rows,cols,sigma", [(3300, 2200, 57.0), (3705, 1000, 30.0), (2900, 517, 44.0), (3600, 1283, 20.0), (3075, 2050, 55.45)])
def test_long_columns_whole_image(ctx, rows, cols, sigma):
    """columns of 2561 .. 4096 padded points: the C = 2 column kernels (strips of 4 columns) and a row kernel reading 4-column strips,
    the whole image in one transform per line; ragged widths, quirk on"""
    torch = _torch()
    from oracle import oracle as O
    img = _rand_img(rows, cols, rows + cols)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, True, want_planes=True)
    t = torch.from_numpy(img).cuda()
    got = ctx.pffft_(t, sigma, out=torch.empty_like(t), wave_resident=True)
    assert ctx.last_family() == 2
    assert_u8_parity(got.cpu().numpy(), want, planes)


@pytest.mark.parametrize("rows,cols,sigma,wr", [(2700, 518, 44.0, True), (900, 1270, 30.0, True), (2700, 515, 44.0, True)])
def test_ragged_last_strip_in_a_batch(ctx, rows, cols, sigma, wr):
    """widths that are no multiple of the column kernels' strip (4 or 8 columns), two frames per call: the last strip's pieces reach
    into the next row -- and, in its last row, into the next frame or past the batch -- so that row is fetched byte by byte and the
    bytes right of the image are masked (wr_kernels.hpp: strip_kind); each frame equals the frame blurred alone and the oracle"""
    torch = _torch()
    from oracle import oracle as O
    frames = np.stack([_rand_img(rows, cols, 5 * rows + cols + i) for i in range(2)])
    t = torch.from_numpy(frames).cuda()
    got = ctx.pffft_(t, sigma, out=torch.empty_like(t), wave_resident=wr).cpu().numpy()
    assert ctx.last_family() == 2
    for i in range(2):
        want, planes = O.pffft_blur_u8c3_f64(frames[i], sigma, True, want_planes=True)
        assert_u8_parity(got[i], want, planes)
