"""GPU parity tests proper: the HIP path (through the C ABI) against the float64 CPU oracle
on the same seeded inputs.  Run on the MI355X box:  python -m pytest tests -m gpu -x -q"""
import os

import numpy as np
import pytest

from conftest import FLOAT_TOL, assert_u8_parity

pytestmark = pytest.mark.gpu


def _rand_img(rows, cols, seed):
    return np.random.default_rng(seed).integers(0, 256, (rows, cols, 3), dtype=np.uint8)


def _torch():
    import torch
    return torch


def test_library_is_the_hip_one(ctx):
    import blur_algorithms_amd as B
    import os
    assert os.path.exists(B.LIB_PATH)
    maps = open("/proc/self/maps").read()
    assert "libblur_amd.so" in maps


# sizes chosen so the FFT lengths cover radix 16/10/9/8/6/5/4/3/2 plans, odd dimensions
# (ragged last row pair / last column strip) and both N0 != N1 and N0 == N1
CASES = [
    (64, 96, 3.0),      # N 96 / 128
    (33, 47, 2.0),      # odd x odd, tiny
    (100, 77, 5.0),     # odd cols
    (101, 203, 4.0),
    (270, 480, 20.0),   # BASELINE C2 at quarter scale
    (512, 512, 5.0),    # BASELINE C1 geometry, N 576 / 576
    (500, 748, 20.0),   # test_images "Baseline.jpg" geometry
]


@pytest.mark.parametrize("rows,cols,sigma", CASES)
def test_rowpass_float_planes(ctx, rows, cols, sigma):
    """row pass (Source.cpp:520-537) against the oracle's `resf` planes"""
    torch = _torch()
    from oracle import oracle as O
    img = _rand_img(rows, cols, 11)
    planes = ctx.rowpass(torch.from_numpy(img).cuda(), sigma).cpu().numpy()
    for c in range(3):
        _, inter = O.pffft_plane_f64(img[:, :, c].astype(np.float32), sigma, True, want_inter=True)
        err = np.abs(planes[c].astype(np.float64) - inter).max()
        assert err <= FLOAT_TOL, "channel %d: max |err| %.3g" % (c, err)


@pytest.mark.parametrize("rows,cols,sigma", CASES)
@pytest.mark.parametrize("quirk", [True, False])
def test_pffft_u8c3(ctx, rows, cols, sigma, quirk):
    """whole pffft_() on u8 BGR against the float64 oracle, Nyquist quirk on and off"""
    torch = _torch()
    from oracle import oracle as O
    img = _rand_img(rows, cols, 7)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk, want_planes=True)
    t = torch.from_numpy(img).cuda()
    out = torch.empty_like(t)
    ctx.pffft_(t, sigma, out=out, nyquist_quirk=quirk)
    assert_u8_parity(out.cpu().numpy(), want, planes)
    # in place, like the reference
    ctx.pffft_(t, sigma, nyquist_quirk=quirk)
    assert torch.equal(t, out)


# Shapes whose FFT lengths have compile-time specialised kernels (fast_kernels.hpp): thin images
# keep the oracle cheap while the long side hits 4000 / 2304 / 1280 / 4320 / 2560 as row or
# column length; odd sizes exercise the single last row and the ragged last column strip.
FAST_CASES = [
    (70, 3840, 20.0),     # row 4000 (column generic)
    (71, 3839, 20.0),     # row 4000, odd x odd
    (2160, 70, 20.0),     # column 2304
    (2159, 67, 20.0),     # column 2304, odd x odd
    (66, 1920, 20.0),     # row 2304
    (1080, 66, 20.0),     # column 1280
    (170, 3840, 50.0),    # row 4320
    (2160, 170, 50.0),    # column 2560
    (1080, 1920, 20.0),   # BASELINE C2 whole: 2304 / 1280
    (1081, 1923, 20.0),   # both passes specialised, odd x odd: ragged last strip in the strip layout
    (2208, 15, 2.0),      # column 2304 on an image two strips wide: fewer workgroups than XCDs (found by tools/fuzz.py --targeted)
    (1200, 8, 2.0),       # column 1280, a single strip
    (40, 2270, 3.0),      # row 2304 with fewer row pairs than CUs
]


@pytest.mark.parametrize("rows,cols,sigma", FAST_CASES)
def test_specialised_lengths(ctx, rows, cols, sigma):
    torch = _torch()
    from oracle import oracle as O
    img = _rand_img(rows, cols, 21)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, True, want_planes=True)
    t = torch.from_numpy(img).cuda()
    got = ctx.pffft_(t, sigma, out=torch.empty_like(t))
    assert_u8_parity(got.cpu().numpy(), want, planes)
    gen = ctx.pffft_(t, sigma, out=torch.empty_like(t), force_generic=True)
    assert_u8_parity(gen.cpu().numpy(), want, planes)
    # the layout of the float intermediate (strips vs row-major planes) and the strip width are
    # data-movement choices: the bytes must not change
    assert torch.equal(got, ctx.pffft_(t, sigma, out=torch.empty_like(t), row_major_planes=True))
    assert torch.equal(got, ctx.pffft_(t, sigma, out=torch.empty_like(t), col_group=4))
    # float planes of the specialised row pass against the oracle's `resf`
    rp = ctx.rowpass(t, sigma).cpu().numpy()
    _, inter = O.pffft_plane_f64(img[:, :, 1].astype(np.float32), sigma, True, want_inter=True)
    assert np.abs(rp[1].astype(np.float64) - inter).max() <= FLOAT_TOL


def test_metric_frame_4k_sigma20(ctx):
    """the metric's own configuration, whole frame against the float64 oracle"""
    torch = _torch()
    from oracle import oracle as O
    img = _rand_img(2160, 3840, 99)
    want, planes = O.pffft_blur_u8c3_f64(img, 20.0, True, want_planes=True)
    t = torch.from_numpy(img).cuda()
    got = ctx.pffft_(t, 20.0, out=torch.empty_like(t)).cpu().numpy()
    n = assert_u8_parity(got, want, planes)
    print("4K sigma=20: %d of %d bytes differ (ties)" % (n, got.size))


@pytest.mark.parametrize("group", [2, 4, 8, 16])
def test_column_group_sizes_agree(ctx, group):
    """the column strip width is a tuning knob: results must not depend on it"""
    torch = _torch()
    img = torch.from_numpy(_rand_img(120, 90, 3)).cuda()
    ref = ctx.pffft_(img, 4.0, out=torch.empty_like(img), col_group=8)
    got = ctx.pffft_(img, 4.0, out=torch.empty_like(img), col_group=group)
    assert torch.equal(ref, got)


def test_pffft_plane_f32_config1(ctx):
    """BASELINE config 1: 512x512 single-channel float32, sigma 5"""
    torch = _torch()
    from oracle import oracle as O
    plane = np.random.default_rng(5).uniform(0, 255, (512, 512)).astype(np.float32)
    want = O.pffft_plane_f64(plane, 5.0, True)
    got = ctx.pffft_plane(torch.from_numpy(plane).cuda(), 5.0).cpu().numpy()
    err = np.abs(got.astype(np.float64) - want).max()
    assert err <= FLOAT_TOL, err
    # host-pointer entry point gives the same bits
    got_h = ctx.pffft_plane(plane, 5.0)
    assert np.array_equal(got_h, got)


def test_host_entry_point_matches_device(ctx):
    torch = _torch()
    img = _rand_img(90, 130, 2)
    dev = ctx.pffft_(torch.from_numpy(img).cuda(), 6.0).cpu().numpy()
    host = ctx.pffft_(img, 6.0)
    assert np.array_equal(dev, host)


@pytest.mark.parametrize("rows,cols,sigma", [(270, 480, 20.0), (101, 77, 4.5), (540, 960, 9.0)])
@pytest.mark.parametrize("path", ["pocketfft_1D", "pocketfft_2D"])
def test_pocketfft_paths(ctx, rows, cols, sigma, path):
    """the engine's pocketfft_1D / pocketfft_2D against scipy.fft (pocketfft) restatements of
    Source.cpp:280-392 and :143-277 -- float64 planes for the tie rule, and the float32 run (what the reference
    computes) must agree with the engine at least as well as it agrees with float64."""
    from oracle import pocketfft_paths as P
    fn = {"pocketfft_1D": P.pocketfft_1d_u8c3, "pocketfft_2D": P.pocketfft_2d_u8c3}[path]
    img = _rand_img(rows, cols, rows + cols)
    want, planes = fn(img, sigma, np.float64, want_planes=True)
    got = getattr(ctx, path)(_torch().from_numpy(img).cuda(), sigma).cpu().numpy()
    assert_u8_parity(got, want, planes)
    ref32 = fn(img, sigma, np.float32)
    assert (got != want).sum() <= max(8, 2 * (ref32 != want).sum())


@pytest.mark.parametrize("pinned", [False, True])
def test_host_batch_pipeline_equals_frame_by_frame(ctx, pinned):
    """7 host frames through the three-slot copy/compute pipeline (more frames than slots, so slots are reused)"""
    n, rows, cols = 7, 120, 200
    src = np.random.default_rng(77).integers(0, 256, (n, rows, cols, 3), dtype=np.uint8)
    if pinned:
        buf = ctx.pinned_empty(src.shape)
        buf[...] = src
        out = ctx.pffft_host_batch(buf, 5.0, out=ctx.pinned_empty(src.shape))
        inplace = ctx.pffft_host_batch(buf, 5.0, out=buf)
        assert inplace is buf and np.array_equal(buf, out)
    else:
        out = ctx.pffft_host_batch(src, 5.0)
    for i in range(n):
        assert np.array_equal(out[i], ctx.pffft_(src[i], 5.0)), i


def test_pitched_host_image_like_a_cv_mat_roi(ctx):
    """rows that are farther apart than cols*3 bytes (a region of interest of a larger image)"""
    big = _rand_img(140, 200, 6)
    roi = big[20:110, 30:170]                      # 90 x 140 view, row pitch 600 bytes
    assert not roi.flags["C_CONTIGUOUS"]
    got = ctx.pffft_(roi, 6.0)
    want = ctx.pffft_(np.ascontiguousarray(roi), 6.0)
    assert np.array_equal(got, want)


def test_batch_equals_single_frames(ctx):
    torch = _torch()
    frames = torch.from_numpy(np.stack([_rand_img(72, 100, s) for s in range(4)])).cuda()
    out = ctx.pffft_(frames, 5.0, out=torch.empty_like(frames))
    for i in range(4):
        one = ctx.pffft_(frames[i].contiguous(), 5.0, out=torch.empty_like(frames[i]))
        assert torch.equal(out[i], one)


def test_constant_image_is_preserved(ctx):
    """DC gain: a constant image stays constant (kernel sums to 1, reflect padding).
    Run without the Nyquist quirk, which adds +-255/N of checkerboard by design."""
    torch = _torch()
    img = torch.full((80, 112, 3), 200, dtype=torch.uint8, device="cuda")
    out = ctx.pffft_(img, 7.0, out=torch.empty_like(img), nyquist_quirk=False)
    assert int(out.min()) == 200 and int(out.max()) == 200


def test_linearity_of_float_path(ctx):
    """blur(a + b) == blur(a) + blur(b) up to round-off (size-independent property)"""
    torch = _torch()
    g = torch.Generator(device="cpu").manual_seed(3)
    a = (torch.rand((160, 224), generator=g) * 100).cuda()
    b = (torch.rand((160, 224), generator=g) * 100).cuda()
    lhs = ctx.pffft_plane(a + b, 9.0)
    rhs = ctx.pffft_plane(a, 9.0) + ctx.pffft_plane(b, 9.0)
    assert float((lhs - rhs).abs().max()) < 2 * FLOAT_TOL


def test_errors_are_loud(ctx):
    torch = _torch()
    import blur_algorithms_amd as B
    # the window is clamped to the LONGER side (Source.cpp:434), so a thin image gets
    # pad 32 > rows - 1 = 3: out-of-buffer reads in the reference (README.md:33-38), refused here
    thin = torch.zeros((4, 64, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(B.BlurError) as e:
        ctx.pffft_(thin, 20.0)
    assert e.value.code == 2
    img = torch.zeros((16, 16, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(B.BlurError):
        ctx.pffft_(img, -1.0)
    ctx.pffft_(img, 20.0)                # window clamped to 17, pad 8: fine


def test_pieces_flip_block_and_interleave(ctx):
    torch = _torch()
    from oracle import oracle as O
    w, h = 150, 70
    plane = torch.arange(w * h, dtype=torch.float32, device="cuda")
    got = ctx.flip_block(plane, w, h).cpu().numpy()
    assert np.array_equal(got, O.flip_block(plane.cpu().numpy(), w, h))
    img = _rand_img(37, 41, 9)
    planes = ctx.deinterleave_BGR(torch.from_numpy(img).cuda())
    assert np.array_equal(planes.cpu().numpy(), O.deinterleave_bgr(img))
    vals = np.random.default_rng(1).uniform(0, 255.49, (3, 1000)).astype(np.float32)
    got = ctx.interleave_BGR(torch.from_numpy(vals).cuda()).cpu().numpy()
    assert np.array_equal(got, O.interleave_bgr(vals))


def test_fastboxblur_config5_full_size(ctx):
    """BASELINE config 5: 8K RGB, 3 passes, box width 41 (nearest odd to sqrt(12 sigma^2 / 3 + 1), sigma 20)"""
    torch = _torch()
    from oracle import oracle as O
    img = np.random.default_rng(55).integers(0, 256, (4320, 7680, 3), dtype=np.uint8)
    want = O.fastboxblur_u8(img, 41, 3)
    got = ctx.fastboxblur(torch.from_numpy(img.copy()).cuda(), 41, 3).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("rows,cols,nsmooth", [(120, 160, 3.0), (201, 333, 5.0), (64, 80, 2.0), (50, 60, 9.0)])
def test_boxblur_mode_of_pffft(ctx, rows, cols, nsmooth):
    """pffft_() compiled with `#define boxblur` (FFT-domain tent kernel, passes = 2)"""
    torch = _torch()
    from oracle import oracle as O
    img = _rand_img(rows, cols, 17)
    want, planes = O.pffft_boxblur_u8c3_f64(img, nsmooth, True, want_planes=True)
    t = torch.from_numpy(img).cuda()
    got = ctx.pffft_boxblur(t, nsmooth, out=torch.empty_like(t)).cpu().numpy()
    assert_u8_parity(got, want, planes)


def test_user_supplied_separable_kernel(ctx):
    torch = _torch()
    from oracle import oracle as O
    taps = np.array([1, 4, 6, 4, 1], np.float32) / 16
    img = _rand_img(90, 120, 19)
    s = dict(N0=O.nearest_transform_size(90 + 4), N1=O.nearest_transform_size(120 + 4))
    def periodic(n):
        k = np.zeros(n, np.float32); k[0] = taps[2]; k[1] = taps[3]; k[2] = taps[4]; k[n - 1] = taps[1]; k[n - 2] = taps[0]; return k
    want, planes = O.pffft_blur_u8c3_f64_kernel(img, 2, periodic(s["N1"]), periodic(s["N0"]), True, want_planes=True)
    t = torch.from_numpy(img).cuda()
    got = ctx.separable(t, taps, out=torch.empty_like(t)).cpu().numpy()
    assert_u8_parity(got, want, planes)
    import blur_algorithms_amd as B
    with pytest.raises(B.BlurError):          # asymmetric kernels have a complex spectrum: refused
        ctx.separable(t, np.array([1, 2, 3], np.float32) / 6, out=torch.empty_like(t))


def test_seeded_fuzz_over_sizes_and_sigmas(ctx):
    """30 seeded random (rows, cols, sigma) draws: every run-time plan shape (radices 2..16 in all
    orders the planner emits), ragged sizes, windows clamped to the longer side"""
    torch = _torch()
    from oracle import oracle as O
    rng = np.random.default_rng(20241108)
    done = 0
    plans = set()
    while done < 30:
        rows, cols = int(rng.integers(8, 420)), int(rng.integers(8, 420))
        sigma = float(rng.choice([0.4, 0.8, 1.3, 2.0, 3.7, 6.0, 11.0, 25.0]))
        s = O.pffft_sizing(rows, cols, sigma)
        if s["pad"] > min(rows, cols) - 1:
            continue
        import blur_algorithms_amd as B
        plans.add(tuple(B.fft_plan_radices(s["N0"])))
        plans.add(tuple(B.fft_plan_radices(s["N1"])))
        img = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
        quirk = bool(done % 2)
        want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk, want_planes=True)
        got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma, nyquist_quirk=quirk).cpu().numpy()
        assert_u8_parity(got, want, planes)
        done += 1
    assert len(plans) >= 8


def test_tiny_sigma_is_the_identity_without_the_quirk(ctx):
    """sigma 0.3 -> a one-tap window (pad 0): the blur is the identity; with the quirk on, the
    reference adds its Nyquist checkerboard even here (Source.cpp:420-425)"""
    torch = _torch()
    from oracle import oracle as O
    assert O.gaussian_window(0.3) == 1
    img = _rand_img(40, 56, 5)
    t = torch.from_numpy(img).cuda()
    assert np.array_equal(ctx.pffft_(t, 0.3, out=torch.empty_like(t), nyquist_quirk=False).cpu().numpy(), img)
    want, planes = O.pffft_blur_u8c3_f64(img, 0.3, True, want_planes=True)
    assert_u8_parity(ctx.pffft_(t, 0.3, out=torch.empty_like(t)).cpu().numpy(), want, planes)


def test_two_contexts_and_streams_do_not_interfere(ctx):
    torch = _torch()
    import blur_algorithms_amd as B
    other = B.BlurContext(0)
    a = torch.from_numpy(_rand_img(200, 300, 1)).cuda()
    b = torch.from_numpy(_rand_img(200, 300, 2)).cuda()
    ra = ctx.pffft_(a, 5.0, out=torch.empty_like(a))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        rb = other.pffft_(b, 9.0, out=torch.empty_like(b))
    s.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(ra, ctx.pffft_(a, 5.0, out=torch.empty_like(a)))
    assert torch.equal(rb, ctx.pffft_(b, 9.0, out=torch.empty_like(b)))
    other.close()


def test_very_tall_image_uses_planar_column_fallback(ctx):
    """rows + 2 pad -> N0 = 12000: a complex line plus the u8 pixel stage exceed LDS, the engine
    falls back to float planes + interleave (the reference's largest benchmark image is this tall)"""
    torch = _torch()
    from oracle import oracle as O
    rows, cols, sigma = 10950, 352, 104.6
    assert O.pffft_sizing(rows, cols, sigma)["N0"] == 12000
    img = _rand_img(rows, cols, 77)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, True, want_planes=True)
    t = torch.from_numpy(img).cuda()
    assert_u8_parity(ctx.pffft_(t, sigma, out=torch.empty_like(t)).cpu().numpy(), want, planes)


def test_config3_frame_4k_sigma50(ctx):
    """BASELINE config 3 whole: 4K, sigma 50 (kSize 331, FFT lengths 4320 / 2560)"""
    torch = _torch()
    from oracle import oracle as O
    img = _rand_img(2160, 3840, 33)
    want, planes = O.pffft_blur_u8c3_f64(img, 50.0, True, want_planes=True)
    t = torch.from_numpy(img).cuda()
    assert_u8_parity(ctx.pffft_(t, 50.0, out=torch.empty_like(t)).cpu().numpy(), want, planes)


def test_batch_of_4k_frames_roundtrip_properties(ctx):
    """BASELINE config 4 shape on one GPU (8 frames of 4K): size-independent properties --
    every frame of the batch equals the same frame blurred alone, and blurring is idempotent
    on a constant frame (no oracle needed at this size)"""
    torch = _torch()
    g = torch.Generator(device="cuda").manual_seed(4)
    frames = torch.randint(0, 256, (8, 2160, 3840, 3), dtype=torch.uint8, device="cuda", generator=g)
    frames[3] = 117
    out = ctx.pffft_(frames, 20.0, out=torch.empty_like(frames), nyquist_quirk=False)
    assert int(out[3].min()) == 117 and int(out[3].max()) == 117
    for i in (0, 5, 7):
        one = ctx.pffft_(frames[i].contiguous(), 20.0, out=torch.empty_like(frames[i]), nyquist_quirk=False)
        assert torch.equal(one, out[i])
    # checksum of checksums is invariant to how the batch is chunked into launches
    a = ctx.pffft_(frames, 20.0, out=torch.empty_like(frames), frames_per_launch=1)
    b = ctx.pffft_(frames, 20.0, out=torch.empty_like(frames), frames_per_launch=3)
    assert torch.equal(a, b)


@pytest.mark.parametrize("w,h,ch,ksize,passes", [(97, 61, 3, 9, 2), (128, 64, 1, 5, 3), (50, 40, 3, 41, 3), (33, 20, 3, 81, 1),
                                                 (3, 9, 3, 5, 2), (1, 7, 1, 3, 1), (5, 5, 4, 9, 3), (130, 17, 4, 7, 2), (61, 33, 2, 11, 2),
                                                 (1030, 12, 3, 43, 3), (1027, 9, 1, 2001, 2), (259, 31, 3, 1, 2), (64, 64, 3, 3, 4)])
def test_fastboxblur(ctx, w, h, ch, ksize, passes):
    torch = _torch()
    from oracle import oracle as O
    img = np.random.default_rng(4).integers(0, 256, (h, w, ch), dtype=np.uint8)
    want = O.fastboxblur_u8(img, ksize, passes)
    got = ctx.fastboxblur(torch.from_numpy(img.copy()).cuda(), ksize, passes).cpu().numpy()
    assert np.array_equal(got, want)


# shapes that run on the integer matrix cores (csrc/bx_box.hip): every window size of the horizontal kernel (reach C r <= 24, 56,
# 88, 120 bytes) and of the vertical one (r <= 24, 56), 1 / 3 / 4 channels, more than three passes (two launches per direction),
# row pitches that are not a multiple of 16 bytes, heights that are not a multiple of 16, and shapes where one direction falls back
# to the accumulator kernels (box wider than the windows, too few rows for the pipeline, pitch not a multiple of 4); round 4: three
# channels run on the channel-plane kernel (r <= 24 and r <= 56; any width: rows that end in a cut quad of pixels, several segments
# per row, rows shorter than the kernel's reach past their ends)
BX_SHAPES = [(640, 480, 3, 41, 3), (256, 300, 1, 9, 2), (128, 200, 4, 15, 1), (600, 900, 3, 65, 2), (1000, 700, 3, 49, 3), (512, 400, 3, 3, 3),
             (512, 400, 3, 5, 4), (300, 333, 4, 11, 5), (644, 333, 3, 41, 3), (201, 257, 4, 21, 2), (700, 500, 3, 81, 3), (4096, 300, 1, 121, 2),
             (640, 60, 3, 41, 3), (333, 517, 3, 41, 3), (2048, 200, 1, 49, 3), (700, 420, 3, 115, 2), (64, 200, 3, 7, 3), (44, 300, 3, 7, 3),
             (335, 200, 3, 41, 3), (1001, 300, 3, 113, 2), (46, 40, 3, 9, 3), (641, 100, 3, 41, 1), (3847, 40, 3, 41, 3), (2050, 70, 3, 99, 3)]


@pytest.mark.parametrize("w,h,ch,ksize,passes", BX_SHAPES)
@pytest.mark.parametrize("kind", ["uniform", "binary"])
def test_fastboxblur_on_the_matrix_cores(ctx, w, h, ch, ksize, passes, kind):
    """bytes equal to oracle/boxblur_oracle.c; `binary` images (0 / 255) reach the largest sums the 24-bit multiply has to divide"""
    torch = _torch()
    from oracle import oracle as O
    rng = np.random.default_rng(w * 7 + h)
    img = rng.integers(0, 256, (h, w, ch), dtype=np.uint8) if kind == "uniform" else (rng.integers(0, 2, (h, w, ch)) * 255).astype(np.uint8)
    want = O.fastboxblur_u8(img, ksize, passes)
    got = ctx.fastboxblur(torch.from_numpy(img.copy()).cuda(), ksize, passes).cpu().numpy()
    assert np.array_equal(got, want)


def test_fastboxblur_every_box_width_and_extremes(ctx):
    """every odd width 3 .. 129 on one image (all four window sizes of both kernels, the widths past them), then the images with
    the largest and smallest sums: all 255, all 0, alternating columns, alternating rows"""
    torch = _torch()
    from oracle import oracle as O
    img = np.random.default_rng(77).integers(0, 256, (260, 384, 3), dtype=np.uint8)
    for k in range(3, 131, 2):
        want = O.fastboxblur_u8(img, k, 2)
        got = ctx.fastboxblur(torch.from_numpy(img.copy()).cuda(), k, 2).cpu().numpy()
        assert np.array_equal(got, want), "box width %d" % k
    x = np.arange(384)[None, :, None]
    y = np.arange(260)[:, None, None]
    for name, im in (("white", np.full((260, 384, 3), 255)), ("black", np.zeros((260, 384, 3))), ("columns", np.broadcast_to(255 * (x & 1), (260, 384, 3))),
                     ("rows", np.broadcast_to(255 * (y & 1), (260, 384, 3)))):
        im = np.ascontiguousarray(im).astype(np.uint8)
        for k in (3, 41, 49, 79):
            want = O.fastboxblur_u8(im, k, 3)
            got = ctx.fastboxblur(torch.from_numpy(im.copy()).cuda(), k, 3).cpu().numpy()
            assert np.array_equal(got, want), "%s, box width %d" % (name, k)


def test_fastboxblur_fuzz_time_boxed(ctx):
    """random widths, heights, channel counts (1, 3, 4), box widths 1 .. 131, passes 1 .. 5, uniform and 0 / 255 images: byte-equal to
    the oracle whichever kernels the library picks per direction (matrix-core pipelines, accumulator kernels).  About 25 s."""
    import time
    torch = _torch()
    from oracle import oracle as O
    rng = np.random.default_rng(20261007)
    t_end = time.time() + 25.0
    cases = 0
    while time.time() < t_end or cases < 20:
        ch = int(rng.choice([1, 3, 3, 4]))
        w = int(rng.integers(1, 900))
        if rng.random() < 0.6:
            w = (w + 3) & ~3
        h = int(rng.integers(1, 700))
        k = int(rng.integers(1, 132))
        p = int(rng.integers(1, 6))
        img = rng.integers(0, 256, (h, w, ch), dtype=np.uint8) if rng.random() < 0.7 else (rng.integers(0, 2, (h, w, ch)) * 255).astype(np.uint8)
        want = O.fastboxblur_u8(img, k, p)
        got = ctx.fastboxblur(torch.from_numpy(img.copy()).cuda(), k, p).cpu().numpy()
        assert np.array_equal(got, want), "w=%d h=%d channels=%d box=%d passes=%d: %d bytes differ" % (w, h, ch, k, p, int((got != want).sum()))
        cases += 1


def test_fastboxblur_writes_nothing_outside_the_image(ctx):
    """in place on a buffer with guard bytes either side, at a 4-byte-aligned and at an odd offset (the odd one runs the accumulator
    kernels: the matrix-core kernels want dword-aligned rows)"""
    torch = _torch()
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    for (w, h, ch, k, p, shift) in ((640, 300, 3, 41, 3, 0), (644, 301, 3, 41, 3, 0), (400, 300, 4, 15, 2, 0), (640, 300, 3, 41, 3, 1)):
        img = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        n, guard = img.size, 4096
        buf = torch.full((n + 2 * guard + 8,), 0xA5, dtype=torch.uint8, device="cuda")
        view = buf[guard + shift:guard + shift + n].view(h, w, ch)
        view.copy_(torch.from_numpy(img))
        ctx.fastboxblur(view, k, p)
        assert np.array_equal(view.cpu().numpy(), O.fastboxblur_u8(img, k, p))
        red = torch.cat([buf[:guard + shift], buf[guard + shift + n:]])
        assert int((red != 0xA5).sum()) == 0, "guard bytes overwritten"


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["colourgram", "collage_top", "baseline", "input7"])
def test_natural_images_against_committed_vectors(ctx, name):
    """decoded crops of the reference's test_images (committed raw, JPEG decoders differ) against
    the committed float64-oracle output -- runs on the GPU box without /root/reference"""
    torch = _torch()
    v = np.load(os.path.join(GOLDEN, "img_%s.npz" % name))
    src, sigma = v["src"], float(v["sigma"])
    got = ctx.pffft_(torch.from_numpy(src).cuda(), sigma).cpu().numpy()
    assert_u8_parity(got, v["oracle_u8"], v["oracle_planes"])
    got = ctx.pffft_(torch.from_numpy(src).cuda(), sigma, nyquist_quirk=False).cpu().numpy()
    d = got.astype(int) - v["oracle_u8_noquirk"].astype(int)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 2e-3


def test_cpp_surface_on_gpu(ctx, tmp_path):
    """a C++ caller using the reference's names (pffft_(Mat&, sigma), fastboxblur(...)) through
    include/blur_amd.hpp gets the same bytes as the Python binding"""
    import subprocess
    import blur_algorithms_amd as B
    torch = _torch()
    root = os.path.dirname(GOLDEN.rstrip("/")).rsplit("/tests", 1)[0]
    exe = str(tmp_path / "surface_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fopenmp", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "surface_check.cpp"),
                           "-L" + os.path.dirname(B.LIB_PATH), "-lblur_amd", "-Wl,-rpath," + os.path.dirname(B.LIB_PATH),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    img = _rand_img(96, 140, 13)
    (tmp_path / "in.raw").write_bytes(img.tobytes())
    subprocess.check_call([exe, "blur", str(tmp_path / "in.raw"), "96", "140", "6.0", str(tmp_path / "out.raw")])
    got = np.frombuffer((tmp_path / "out.raw").read_bytes(), np.uint8).reshape(img.shape)
    want = ctx.pffft_(torch.from_numpy(img).cuda(), 6.0).cpu().numpy()
    assert np.array_equal(got, want)
    want = ctx.pocketfft_1D(torch.from_numpy(img).cuda(), 6.0).cpu().numpy()
    for mode in ("pocket1d", "pocket2d"):
        subprocess.check_call([exe, mode, str(tmp_path / "in.raw"), "96", "140", "6.0", str(tmp_path / "out.raw")])
        assert np.array_equal(np.frombuffer((tmp_path / "out.raw").read_bytes(), np.uint8).reshape(img.shape), want)
    for mode, fn in (("whole2d", lambda t: ctx.pocketfft_2D(t, 6.0, whole_image=True)), ("dftimage", lambda t: ctx.DFT_image(t, 6.0))):
        subprocess.check_call([exe, mode, str(tmp_path / "in.raw"), "96", "140", "6.0", str(tmp_path / "out.raw")])
        assert np.array_equal(np.frombuffer((tmp_path / "out.raw").read_bytes(), np.uint8).reshape(img.shape), fn(torch.from_numpy(img).cuda()).cpu().numpy())
    subprocess.check_call([exe, "box", str(tmp_path / "in.raw"), "140", "96", "3", "9", "2", str(tmp_path / "box.raw")])
    got = np.frombuffer((tmp_path / "box.raw").read_bytes(), np.uint8).reshape(img.shape)
    from oracle import oracle as O
    assert np.array_equal(got, O.fastboxblur_u8(img, 9, 2))
