"""The fused matrix-core kernel (fx_kernels.hpp: row pass, register hand-off, column pass and byte emission in one launch, the
library's choice where it applies) against the float64 oracle under the same parity contract as every other engine."""
import numpy as np
import pytest

from conftest import FLOAT_TOL, assert_u8_parity

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _rand_img(rows, cols, seed):
    return np.random.default_rng(seed).integers(0, 256, (rows, cols, 3), dtype=np.uint8)


def _fam(ctx):
    return ctx.last_family()


# widths that are multiples of 4 (the others: RAGGED below); heights are anything; below one chunk of 128 columns, exactly one,
# several with a ragged last one; heights below one tile of 32 rows (pad + 1 at least), ragged last tiles
SHAPES = [(270, 480, 20.0), (200, 332, 20.0), (131, 152, 18.0), (540, 960, 21.5), (97, 644, 19.0), (1080, 1920, 20.0), (70, 68, 20.0), (100, 100, 20.0),
          (67, 256, 19.0), (300, 72, 20.0), (66, 132, 20.0)]


@pytest.mark.parametrize("rows,cols,sigma", SHAPES)
@pytest.mark.parametrize("quirk", [False, True])
def test_fused_engine_matches_the_oracle(ctx, rows, cols, sigma, quirk):
    from oracle import oracle as O
    torch = _torch()
    img = _rand_img(rows, cols, rows + 3 * cols)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=quirk, want_planes=True)
    got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma, nyquist_quirk=quirk, engine="fused").cpu().numpy()
    assert _fam(ctx) == 6
    assert_u8_parity(got, want, planes)
    # the library's own choice is this kernel, byte for byte
    auto = ctx.pffft_(torch.from_numpy(img).cuda(), sigma, nyquist_quirk=quirk).cpu().numpy()
    assert _fam(ctx) == 6
    assert np.array_equal(auto, got)


@pytest.mark.parametrize("rows,cols,sigma", [(270, 480, 20.0), (131, 152, 18.0), (97, 644, 19.0), (70, 68, 20.0), (66, 132, 20.0), (200, 332, 6.0), (150, 260, 2.0)])
@pytest.mark.parametrize("quirk", [True, False])
def test_fused_engine_row_pass_float_planes(ctx, rows, cols, sigma, quirk):
    """the float planes the fused kernel hands from its row pass to its column pass (a test instantiation of the same kernel
    writes them out through blur_rowpass_u8c3_dev) against the oracle's `resf` planes (Source.cpp:520-537), Nyquist term of the
    row transform included, at the tolerance the FFT kernels' planes are held to"""
    from oracle import oracle as O
    torch = _torch()
    img = _rand_img(rows, cols, 5 * rows + cols)
    planes = ctx.rowpass(torch.from_numpy(img).cuda(), sigma, nyquist_quirk=quirk, engine="fused").cpu().numpy()
    assert _fam(ctx) == 6
    for c in range(3):
        _, inter = O.pffft_plane_f64(img[:, :, c].astype(np.float32), sigma, quirk, want_inter=True)
        err = np.abs(planes[c].astype(np.float64) - inter).max()
        assert err <= FLOAT_TOL, "channel %d: max |err| %.3g" % (c, err)


# the wide fused kernels (fw_kernels.hpp: one channel per workgroup, 13 .. 23 window blocks, pad 73 .. 168): one sigma per window size
# on shapes with one chunk, several chunks with both kinds of edge strips (two chunks reach over the left edge once pad > 128), ragged
# heights, narrow images
WIDE = [(300, 400, 25.0, 13), (200, 332, 30.0, 15), (340, 132, 36.0, 17), (300, 644, 40.0, 19), (320, 520, 44.0, 21), (400, 520, 50.0, 23), (350, 256, 50.0, 23),
        (180, 644, 44.0, 21), (541, 1028, 33.0, 17), (170, 172, 50.0, 13)]


@pytest.mark.parametrize("rows,cols,sigma,nkb", WIDE)
@pytest.mark.parametrize("quirk", [False, True])
def test_wide_fused_kernels_match_the_oracle(ctx, rows, cols, sigma, nkb, quirk):
    import blur_algorithms_amd as B
    from oracle import oracle as O
    torch = _torch()
    pad = B.pffft_sizing(rows, cols, sigma)["pad"]
    assert 8 * (nkb - 4) < pad <= 8 * (nkb - 2)
    img = _rand_img(rows, cols, rows + 3 * cols)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=quirk, want_planes=True)
    got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma, nyquist_quirk=quirk, engine="fused").cpu().numpy()
    assert _fam(ctx) == 6
    assert_u8_parity(got, want, planes)


def test_wide_fused_batch_segments_in_place_and_extremes(ctx):
    """a batch equals its frames blurred alone (segments differ: one tall frame is cut, a batch is not), in place equals out of place,
    a constant image stays constant, and the 0 / 255 images that stretch the quirk's terms agree with the oracle"""
    from oracle import oracle as O
    torch = _torch()
    frames = torch.from_numpy(np.random.default_rng(19).integers(0, 256, (3, 700, 392, 3), dtype=np.uint8)).cuda()
    one = torch.stack([ctx.pffft_(f.clone(), 40.0, engine="fused") for f in frames])
    assert torch.equal(ctx.pffft_(frames.clone(), 40.0, engine="fused"), one)
    inplace = frames.clone()
    ctx.pffft_(inplace, 40.0, out=inplace, engine="fused")
    assert torch.equal(inplace, one)
    const = torch.full((300, 300, 3), 93, dtype=torch.uint8, device="cuda")
    assert int((ctx.pffft_(const.clone(), 45.0, engine="fused") != 93).sum()) == 0
    rows, cols, sigma = 300, 388, 40.0
    x = np.arange(cols)[None, :, None]
    y = np.arange(rows)[:, None, None]
    for name, img in (("columns", np.broadcast_to(255 * (x & 1), (rows, cols, 3))), ("rows", np.broadcast_to(255 * (y & 1), (rows, cols, 3))),
                      ("checker", np.broadcast_to(255 * ((x + y) & 1), (rows, cols, 3))), ("white", np.full((rows, cols, 3), 255))):
        img = np.ascontiguousarray(img).astype(np.uint8)
        want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=True, want_planes=True)
        got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma, engine="fused").cpu().numpy()
        assert_u8_parity(got, want, planes), name


def test_wide_fused_4k_frame_against_the_two_kernel_engine(ctx):
    """a whole 4K frame, sigma 36 (the library's own choice there): at most one grey level from the two-kernel engine's bytes, on fewer
    than 1e-4 of them (both are held to the oracle on smaller shapes; the float64 oracle takes minutes at this size)"""
    torch = _torch()
    g = torch.Generator(device="cuda").manual_seed(8)
    fr = torch.randint(0, 256, (2, 2160, 3840, 3), dtype=torch.uint8, device="cuda", generator=g)
    a = ctx.pffft_(fr, 36.0, out=torch.empty_like(fr))
    assert _fam(ctx) == 6
    b = ctx.pffft_(fr, 36.0, out=torch.empty_like(fr), engine="matrix")
    d = (a.int() - b.int()).abs()
    assert int(d.max()) <= 1 and float((d != 0).float().mean()) < 1e-4


# one sigma per instantiated window size (NKB = 3, 5, 7, 9, 11: pad <= 8, 24, 40, 56, 72); sigma 2.0 is the truncated Gaussian
# whose alternating sum is most negative (Nyquist gain 1.0037)
EVERY_WINDOW = [(2.0, 3), (2.5, 3), (7.5, 5), (12.5, 7), (17.0, 9), (22.0, 11)]


@pytest.mark.parametrize("sigma,nkb", EVERY_WINDOW)
def test_every_instantiated_window_of_the_fused_kernel(ctx, sigma, nkb):
    import blur_algorithms_amd as B
    from oracle import oracle as O
    torch = _torch()
    rows, cols = 171 + 2 * nkb, 332
    pad = B.pffft_sizing(rows, cols, sigma)["pad"]
    assert 8 * (nkb - 4) < pad <= 8 * (nkb - 2) or nkb == 3
    img = _rand_img(rows, cols, nkb)
    for quirk in (True, False):
        want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=quirk, want_planes=True)
        got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma, nyquist_quirk=quirk, engine="fused").cpu().numpy()
        assert _fam(ctx) == 6
        assert_u8_parity(got, want, planes)


def test_fused_engine_batch_segments_in_place_and_constant(ctx):
    """a frame blurred alone (its strips of columns are cut into segments of tiles to fill the chip) and inside a batch (whole
    strips) gives the same bytes; in place (the reference's calling convention, Source.cpp:429) equals out of place; a constant
    image stays constant"""
    torch = _torch()
    frames = torch.from_numpy(np.random.default_rng(9).integers(0, 256, (3, 470, 388, 3), dtype=np.uint8)).cuda()
    batch = ctx.pffft_(frames, 20.0, out=torch.empty_like(frames), engine="fused")
    for i in range(3):
        one = ctx.pffft_(frames[i], 20.0, out=torch.empty_like(frames[i]), engine="fused")
        assert torch.equal(one, batch[i])
    inplace = frames.clone()
    ctx.pffft_(inplace, 20.0, out=inplace, engine="fused")
    assert torch.equal(inplace, batch)
    const = torch.full((200, 300, 3), 201, dtype=torch.uint8, device="cuda")
    assert int((ctx.pffft_(const.clone(), 20.0, engine="fused") != 201).sum()) == 0


def test_fused_engine_extreme_images(ctx):
    """columns / rows alternating 0 and 255 (the quirk's row / column term is +-255 on top of 127.5: values past the byte range
    wrap like (uint8_t)(v + 0.5f) does on x86), checkerboard, all 255, all 0"""
    from oracle import oracle as O
    torch = _torch()
    rows, cols, sigma = 150, 264, 20.0
    x = np.arange(cols)[None, :, None]
    y = np.arange(rows)[:, None, None]
    for name, img in (("columns", np.broadcast_to(255 * (x & 1), (rows, cols, 3))), ("rows", np.broadcast_to(255 * (y & 1), (rows, cols, 3))),
                      ("checker", np.broadcast_to(255 * ((x + y) & 1), (rows, cols, 3))), ("white", np.full((rows, cols, 3), 255)), ("black", np.zeros((rows, cols, 3)))):
        img = np.ascontiguousarray(img).astype(np.uint8)
        want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=True, want_planes=True)
        got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma, engine="fused").cpu().numpy()
        assert_u8_parity(got, want, planes), name


def test_where_the_fused_kernel_does_not_apply(ctx):
    """a kernel wider than 337 taps: asking for the fused kernel is an error, the library's own choice takes another engine and gives
    the oracle's bytes (widths that are no multiples of 4 and unaligned frame pointers ARE the fused kernel's since round 4: below)"""
    from blur_algorithms_amd.api import BlurError
    from oracle import oracle as O
    torch = _torch()
    img = _rand_img(400, 400, 7)
    with pytest.raises(BlurError):
        ctx.pffft_(torch.from_numpy(img).cuda(), 55.0, engine="fused")          # pad 182 > 168
    want, planes = O.pffft_blur_u8c3_f64(img, 55.0, quirk=True, want_planes=True)
    got = ctx.pffft_(torch.from_numpy(img).cuda(), 55.0).cpu().numpy()
    assert _fam(ctx) != 6
    assert_u8_parity(got, want, planes)


# every residue of the width modulo 4 (the last quad of a row then holds 1 .. 3 pixels: pffft_() takes whatever cv::imread returns,
# Source.cpp:459-461,567; the reference's own test image crop Baseline.jpg is 333 x 251), narrow and several chunks wide, both edges
RAGGED = [(210, 333, 20.0), (251, 333, 20.0), (131, 154, 18.0), (97, 643, 19.0), (70, 69, 20.0), (66, 131, 20.0), (100, 257, 12.0), (150, 1026, 20.0), (80, 135, 5.0)]


@pytest.mark.parametrize("rows,cols,sigma", RAGGED)
@pytest.mark.parametrize("quirk", [False, True])
def test_fused_engine_any_width(ctx, rows, cols, sigma, quirk):
    from oracle import oracle as O
    torch = _torch()
    img = _rand_img(rows, cols, 7 * rows + cols)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=quirk, want_planes=True)
    src = torch.from_numpy(img).cuda()
    # guard bytes behind the output: the cut quad at the end of the last row must not write past the frame
    buf = torch.full((img.size + 64,), 0xA5, dtype=torch.uint8, device="cuda")
    out = buf[:img.size].view(rows, cols, 3)
    got = ctx.pffft_(src, sigma, out=out, nyquist_quirk=quirk).cpu().numpy()
    assert _fam(ctx) == 6
    assert_u8_parity(got, want, planes)
    assert bool((buf[img.size:] == 0xA5).all())


# Widths around every way a chunk of 128 columns can meet the image's edges (round 4: an edge chunk's strip holds only the staging
# loads that touch a mirrored pixel -- fx_strip_range -- and its other loads read the image): one chunk that is left and right edge at
# once, a right edge that cuts the last chunk after 1 .. 127 columns, two chunks whose windows reach past the right edge, windows
# that end exactly at the edge; a narrow window (sigma 6: pad 19) where the strip is a single staging load.
@pytest.mark.parametrize("cols,sigma", [(129, 20.0), (136, 20.0), (199, 20.0), (200, 20.0), (201, 20.0), (255, 20.0), (256, 20.0), (257, 20.0), (272, 20.0),
                                        (328, 20.0), (329, 20.0), (400, 20.0), (513, 20.0), (641, 20.0), (130, 6.0), (260, 6.0), (300, 12.0)])
def test_fused_engine_edge_strips(ctx, cols, sigma):
    from oracle import oracle as O
    torch = _torch()
    rows = 67
    img = _rand_img(rows, cols, 31 * cols)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=True, want_planes=True)
    got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma).cpu().numpy()
    assert _fam(ctx) == 6
    assert_u8_parity(got, want, planes)


@pytest.mark.parametrize("off_in,off_out", [(1, 0), (2, 3), (3, 1), (0, 2)])
@pytest.mark.parametrize("cols", [332, 333, 646])
def test_fused_engine_any_alignment(ctx, off_in, off_out, cols):
    """frame pointers at byte offsets 1, 2, 3 (a cv::Mat ROI, a tensor view): the same kernel, the same bytes as the aligned call"""
    from oracle import oracle as O
    torch = _torch()
    rows, sigma = 140, 20.0
    img = _rand_img(rows, cols, cols + off_in)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=True, want_planes=True)
    ibuf = torch.full((img.size + 80,), 0x5A, dtype=torch.uint8, device="cuda")
    obuf = torch.full((img.size + 80,), 0xA5, dtype=torch.uint8, device="cuda")
    src = ibuf[16 + off_in:16 + off_in + img.size].view(rows, cols, 3)
    dst = obuf[16 + off_out:16 + off_out + img.size].view(rows, cols, 3)
    src.copy_(torch.from_numpy(img))
    got = ctx.pffft_(src, sigma, out=dst).cpu().numpy()
    assert _fam(ctx) == 6
    assert_u8_parity(got, want, planes)
    assert bool((obuf[:16 + off_out] == 0xA5).all()) and bool((obuf[16 + off_out + img.size:] == 0xA5).all())
    aligned = ctx.pffft_(torch.from_numpy(img).cuda(), sigma).cpu().numpy()
    assert np.array_equal(aligned, got)


@pytest.mark.parametrize("rows,cols,sigma", [(300, 401, 25.0), (340, 131, 36.0), (400, 522, 50.0)])
def test_wide_fused_kernels_any_width(ctx, rows, cols, sigma):
    from oracle import oracle as O
    torch = _torch()
    img = _rand_img(rows, cols, rows + cols)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=True, want_planes=True)
    buf = torch.full((img.size + 67,), 0xA5, dtype=torch.uint8, device="cuda")
    out = buf[3:3 + img.size].view(rows, cols, 3)
    got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma, out=out, engine="fused").cpu().numpy()
    assert _fam(ctx) == 6
    assert_u8_parity(got, want, planes)
    assert bool((buf[:3] == 0xA5).all()) and bool((buf[3 + img.size:] == 0xA5).all())


def test_fused_engine_metric_frame(ctx):
    """the metric's own frame (3840 x 2160, sigma 20) whole against the float64 oracle, on the library's choice of engine"""
    from oracle import oracle as O
    torch = _torch()
    img = _rand_img(2160, 3840, 2160)
    want, planes = O.pffft_blur_u8c3_f64(img, 20.0, quirk=True, want_planes=True)
    got = ctx.pffft_(torch.from_numpy(img).cuda(), 20.0).cpu().numpy()
    assert _fam(ctx) == 6
    n = assert_u8_parity(got, want, planes)
    assert n < 2000


def _fused_fuzz(ctx, sigmas, seed, seconds):
    import time
    from oracle import oracle as O
    torch = _torch()
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    cases = 0
    while time.time() < t_end or cases < 15:
        sigma = float(rng.choice(sigmas))
        pad = O.pffft_sizing(4096, 4096, sigma)["pad"]
        rows = int(rng.integers(pad + 1, pad + 300))
        cols = (int(rng.integers(pad + 1, pad + 460)) + 3) & ~3
        if O.pffft_sizing(rows, cols, sigma)["pad"] > min(rows, cols) - 1:
            continue
        quirk = bool(rng.integers(0, 2))
        kind = rng.choice(["uniform", "binary"])
        nf = int(rng.choice([1, 1, 2]))
        img = rng.integers(0, 256, (nf, rows, cols, 3), dtype=np.uint8) if kind == "uniform" else (rng.integers(0, 2, (nf, rows, cols, 3)) * 255).astype(np.uint8)
        guard = 4096
        n = nf * rows * cols * 3
        buf = torch.full((2 * n + 3 * guard,), 0xA5, dtype=torch.uint8, device="cuda")
        src = buf[guard:guard + n].view(nf, rows, cols, 3)
        dst = buf[2 * guard + n:2 * guard + 2 * n].view(nf, rows, cols, 3)
        src.copy_(torch.from_numpy(img))
        inplace = rng.random() < 0.3
        got = ctx.pffft_(src, sigma, out=src if inplace else dst, nyquist_quirk=quirk, engine="fused").cpu().numpy()
        try:
            for i in range(nf):
                want, planes = O.pffft_blur_u8c3_f64(img[i], sigma, quirk, want_planes=True)
                assert_u8_parity(got[i], want, planes)
            red = torch.cat([buf[:guard], buf[guard + n:2 * guard + n], buf[2 * guard + 2 * n:]])
            assert int((red != 0xA5).sum()) == 0, "redzone overwritten"
            if not inplace:
                assert np.array_equal(src.cpu().numpy(), img), "source modified"
        except AssertionError as e:
            raise AssertionError("rows=%d cols=%d sigma=%r quirk=%d kind=%s inplace=%d frames=%d: %s" % (rows, cols, sigma, quirk, kind, inplace, nf, e))
        cases += 1


def test_fused_engine_fuzz_time_boxed(ctx):
    """random shapes (widths below one chunk of 128 columns, heights below one tile of 32 rows, ragged last chunks and tiles), random
    sigma over every window size, both quirk settings, uniform and 0 / 255 images, in place and out of place, single frames and
    small batches, against the float64 oracle; with redzones around the image buffers.  About 30 s."""
    _fused_fuzz(ctx, [0.7, 1.5, 3.0, 6.0, 9.5, 14.0, 17.5, 20.0, 22.0], 20261005, 30.0)


def test_wide_fused_kernels_fuzz_time_boxed(ctx):
    """the same for the wide kernels (sigma 23 .. 50: 13 .. 23 window blocks; one or two chunks reading mirrored pixels at the left
    edge, images narrower than the window).  About 30 s."""
    _fused_fuzz(ctx, [23.0, 25.5, 28.0, 31.0, 34.0, 37.5, 41.0, 44.5, 48.0, 50.0], 20261006, 30.0)
