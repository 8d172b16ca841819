"""The matrix-core engine (mx_kernels.hpp: both passes as banded Toeplitz products on v_mfma_f32_32x32x16_f16) against the
same oracles and the same parity contract as the FFT kernels."""
import numpy as np
import pytest

from conftest import FLOAT_TOL, assert_u8_parity

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _rand_img(rows, cols, seed):
    return np.random.default_rng(seed).integers(0, 256, (rows, cols, 3), dtype=np.uint8)


# pad 57..72 (11 window blocks): sigma 17.6 .. 22
SHAPES = [(270, 480, 20.0), (200, 333, 20.0), (131, 150, 18.0), (540, 960, 21.5), (97, 641, 19.0), (1080, 1920, 20.0)]


@pytest.mark.parametrize("rows,cols,sigma", SHAPES)
@pytest.mark.parametrize("quirk", [False, True])
@pytest.mark.parametrize("engine", ["matrix", None])
def test_matrix_engine_matches_the_oracle(ctx, rows, cols, sigma, quirk, engine):
    from oracle import oracle as O
    torch = _torch()
    img = _rand_img(rows, cols, rows + 3 * cols)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=quirk, want_planes=True)
    got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma, nyquist_quirk=quirk, engine=engine).cpu().numpy()
    assert_u8_parity(got, want, planes)


def test_matrix_engine_batch_and_constant(ctx):
    torch = _torch()
    frames = torch.from_numpy(np.random.default_rng(9).integers(0, 256, (3, 150, 260, 3), dtype=np.uint8)).cuda()
    one = torch.stack([ctx.pffft_(f.clone(), 20.0, engine="matrix") for f in frames])
    assert torch.equal(ctx.pffft_(frames.clone(), 20.0, engine="matrix"), one)
    const = torch.full((200, 300, 3), 201, dtype=torch.uint8, device="cuda")
    assert int((ctx.pffft_(const.clone(), 20.0, engine="matrix") != 201).sum()) == 0


# one sigma per instantiated window size (NKB = 3, 5, ..., 23: pad 0..168), odd image sizes, both quirk settings;
# rows, cols > pad.  sigma 2.5 -> pad 7 (NKB 3) ... sigma 50 -> pad 166 (NKB 23); sigma 2.0 is the truncated Gaussian whose
# alternating sum is most negative (Nyquist gain 1.0037)
EVERY_WINDOW = [(2.0, 3), (2.5, 3), (7.5, 5), (12.5, 7), (17.0, 9), (22.0, 11), (26.5, 13), (31.5, 15), (36.5, 17), (41.0, 19), (46.0, 21), (50.0, 23)]


@pytest.mark.parametrize("sigma,nkb", EVERY_WINDOW)
def test_every_instantiated_window(ctx, sigma, nkb):
    import blur_algorithms_amd as B
    from oracle import oracle as O
    torch = _torch()
    lib = B._lib.load()
    rows, cols = 171 + 2 * nkb, 333
    pad = B.pffft_sizing(rows, cols, sigma)["pad"]
    assert lib.blur_mx_window_blocks(pad) == nkb
    img = _rand_img(rows, cols, nkb)
    for quirk in (True, False):
        want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=quirk, want_planes=True)
        got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma, nyquist_quirk=quirk, engine="matrix").cpu().numpy()
        assert_u8_parity(got, want, planes)


def test_the_librarys_choice_of_engine(ctx):
    """BLUR_ENGINE_AUTO (include/blur_amd.h): the fused matrix-core kernel (family 6) where the kernel's half width is at most 72
    (any width, any pointer alignment), and its wide-window form (half widths 73 .. 168) on frames of 1 MP and more;
    else the two-kernel matrix engine (family 4) for frames of 1 MP and more whose kernel it can hold; the FFT kernels for pad > 168,
    for kernels with negative taps and for small frames or the widest windows where the FFT engine has a compile-time family.  The choice never depends on the number of frames.  Asking for an engine
    explicitly where it cannot run is an error."""
    from blur_algorithms_amd.api import BlurError
    torch = _torch()
    fam = ctx.last_family
    batch = torch.zeros((8, 1080, 1920, 3), dtype=torch.uint8, device="cuda")
    big = torch.zeros((3000, 4100, 3), dtype=torch.uint8, device="cuda")
    small = torch.from_numpy(_rand_img(400, 420, 1)).cuda()
    odd = torch.zeros((2, 1080, 1921, 3), dtype=torch.uint8, device="cuda")
    ctx.pffft_(batch, 20.0)
    assert fam() == 6
    ctx.pffft_(batch[0], 20.0)                                    # one frame of the batch: the same engine as the batch
    assert fam() == 6
    ctx.pffft_(small.clone(), 20.0)                               # small frames too: one launch
    assert fam() == 6
    ctx.pffft_(odd, 20.0)                                         # width 1921: the same kernel (any width since round 4)
    assert fam() == 6
    ctx.pffft_(batch, 30.0)                                       # pad 98 on a 2 MP frame: the wide fused kernel (from 1 MP since round 4)
    assert fam() == 6
    mid = torch.from_numpy(_rand_img(700, 1200, 2)).cuda()
    ctx.pffft_(mid, 30.0)                                         # 0.84 MP: not the wide fused kernel
    code, note = ctx.last_engine()                                # blur_last_engine: what ran and why the faster engine was passed over
    assert code != 6 and "not taken" in note and "1 MP" in note
    ctx.pffft_(batch, 20.0)
    assert ctx.last_engine() == (6, "fused matrix-core kernel")
    ctx.pffft_(batch, 60.0)                                       # pad 195: beyond every matrix-core window
    assert fam() not in (4, 6)
    with pytest.raises(BlurError):
        ctx.pffft_(batch, 60.0, engine="matrix")
    ctx.pffft_(batch, 30.0, engine="fused")                       # asked for: the wide fused kernel
    assert fam() == 6
    with pytest.raises(BlurError):
        ctx.pffft_(batch, 60.0, engine="fused")
    small_odd = torch.from_numpy(_rand_img(400, 421, 1)).cuda()
    ctx.pffft_(small_odd, 20.0)                                   # 0.17 MP, width 421: the fused kernel as well
    assert fam() == 6
    uhd = torch.zeros((2, 2160, 3840, 3), dtype=torch.uint8, device="cuda")
    ctx.pffft_(uhd, 20.0)
    assert fam() == 6
    ctx.pffft_(uhd, 30.0)                                         # 4K, pad 98: the wide fused kernel
    assert fam() == 6
    ctx.pffft_(uhd, 44.0)                                         # pad 145, 21 window blocks: still
    assert fam() == 6
    ctx.pffft_(uhd, 50.0)                                         # 4K, 327 taps (23 blocks): the specialised FFT kernels (4320 / 2560)
    assert fam() == 1
    ctx.pffft_(uhd, 50.0, engine="matrix")
    assert fam() == 4
    ctx.pffft_(small.clone(), 20.0, engine="matrix")
    assert fam() == 4
    ctx.separable(big, np.array([0.25, 0.5, 0.25], np.float32))
    assert fam() == 6
    ctx.separable(big, np.array([-0.25, 1.5, -0.25], np.float32))  # symmetric, sum 1, negative side taps
    assert fam() not in (4, 6)


def test_the_table_cache_tells_pads_and_window_sizes_apart(ctx):
    """the same taps with pad 8 and then pad 10 (both 1024-point reference lengths for a 1000-pixel side) have different windows,
    and the two matrix-core engines may use different windows for one pad: neither may pick up the other's fragment table"""
    torch = _torch()
    img = torch.from_numpy(_rand_img(1000, 1000, 3)).cuda()
    taps = np.array([1, 4, 6, 4, 1], np.float32) / 16
    ref8 = ctx.separable(img.clone(), taps, pad=8, engine="fft").cpu().numpy()
    ref10 = ctx.separable(img.clone(), taps, pad=10, engine="fft").cpu().numpy()
    for eng in ("matrix", "fused", "matrix"):
        a = ctx.separable(img.clone(), taps, pad=8, engine=eng).cpu().numpy()
        b = ctx.separable(img.clone(), taps, pad=10, engine=eng).cpu().numpy()
        assert np.abs(a.astype(int) - ref8).max() <= 1 and (a != ref8).mean() < 1e-3
        assert np.abs(b.astype(int) - ref10).max() <= 1 and (b != ref10).mean() < 1e-3
    # a clamped window (sigma 20 on a 100 x 100 image: 101 taps, pad 50): 9 window blocks for the two kernels, 11 for the fused one
    sm = torch.from_numpy(_rand_img(100, 100, 4)).cuda()
    m1 = ctx.pffft_(sm.clone(), 20.0, engine="matrix")
    f1 = ctx.pffft_(sm.clone(), 20.0, engine="fused")
    m2 = ctx.pffft_(sm.clone(), 20.0, engine="matrix")
    assert torch.equal(m1, m2)
    assert int((m1.int() - f1.int()).abs().max()) <= 1


def test_matrix_engine_extreme_images(ctx):
    """the images that stretch the 24-bit intermediate: columns alternating 0 / 255 (the row pass's quirk term is
    +-255 on top of a value of 127.5) and rows alternating (the column term), all 255, all 0"""
    from oracle import oracle as O
    torch = _torch()
    rows, cols, sigma = 150, 262, 20.0
    x = np.arange(cols)[None, :, None]
    y = np.arange(rows)[:, None, None]
    for name, img in (("columns", np.broadcast_to(255 * (x & 1), (rows, cols, 3))), ("rows", np.broadcast_to(255 * (y & 1), (rows, cols, 3))),
                      ("checker", np.broadcast_to(255 * ((x + y) & 1), (rows, cols, 3))), ("white", np.full((rows, cols, 3), 255)), ("black", np.zeros((rows, cols, 3)))):
        img = np.ascontiguousarray(img).astype(np.uint8)
        want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=True, want_planes=True)
        got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma, engine="matrix").cpu().numpy()
        assert_u8_parity(got, want, planes), name


def test_matrix_engine_fuzz_time_boxed(ctx):
    """random shapes (odd widths: the per-byte load path; widths below one chunk; heights below one tile), random sigma over
    every window size, both quirk settings, uniform and 0 / 255 images, in place and out of place, against the float64
    oracle; with redzones around the image buffers.  About 30 s."""
    import time
    from oracle import oracle as O
    torch = _torch()
    rng = np.random.default_rng(20261004)
    t_end = time.time() + 30.0
    cases = 0
    while time.time() < t_end or cases < 15:
        sigma = float(rng.choice([0.7, 1.5, 3.0, 6.0, 9.5, 14.0, 20.0, 25.0, 30.0, 38.0, 44.0, 50.0]))
        pad = O.pffft_sizing(4096, 4096, sigma)["pad"]
        rows = int(rng.integers(pad + 1, pad + 260))
        cols = int(rng.integers(pad + 1, pad + 420))
        if rng.random() < 0.4:
            cols = (cols + 3) & ~3                                   # the aligned 12-byte group path
        if O.pffft_sizing(rows, cols, sigma)["pad"] > min(rows, cols) - 1:
            continue
        quirk = bool(rng.integers(0, 2))
        kind = rng.choice(["uniform", "binary"])
        img = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8) if kind == "uniform" else (rng.integers(0, 2, (rows, cols, 3)) * 255).astype(np.uint8)
        want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk, want_planes=True)
        guard = 4096
        n = rows * cols * 3
        buf = torch.full((2 * n + 3 * guard,), 0xA5, dtype=torch.uint8, device="cuda")
        src = buf[guard:guard + n].view(rows, cols, 3)
        dst = buf[2 * guard + n:2 * guard + 2 * n].view(rows, cols, 3)
        src.copy_(torch.from_numpy(img))
        inplace = rng.random() < 0.3
        got = ctx.pffft_(src, sigma, out=src if inplace else dst, nyquist_quirk=quirk, engine="matrix").cpu().numpy()
        try:
            assert_u8_parity(got, want, planes)
            red = torch.cat([buf[:guard], buf[guard + n:2 * guard + n], buf[2 * guard + 2 * n:]])
            assert int((red != 0xA5).sum()) == 0, "redzone overwritten"
            if not inplace:
                assert np.array_equal(src.cpu().numpy(), img), "source modified"
            assert ctx._lib.blur_debug_check_workspace_guards(ctx._h) == 0, "workspace guard overwritten"
        except AssertionError as e:
            raise AssertionError("rows=%d cols=%d sigma=%r quirk=%d kind=%s inplace=%d: %s" % (rows, cols, sigma, quirk, kind, inplace, e))
        cases += 1
