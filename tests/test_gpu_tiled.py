"""The tiled wave-resident path (engine.hip: run_wr_tiled; round 4): images whose lines are longer than the longest wave-resident
transform -- the reference's own benchmark, sigma = sqrt(side) on large images (Source.cpp:627-635) -- go through the same kernels in
bands of rows and tiles of columns, the Nyquist-slot quirk of pffft_() (Source.cpp:420-425) as rank-one terms from integer sums.
Against the float64 oracle under the parity contract of every other engine; `tile_points` forces small transforms so that small
images exercise every kind of band and tile (first / middle / last, odd and even offsets)."""
import numpy as np
import pytest

from conftest import assert_u8_parity

pytestmark = pytest.mark.gpu


def _rand_img(rows, cols, seed):
    return np.random.default_rng(seed).integers(0, 256, (rows, cols, 3), dtype=np.uint8)


# (rows, cols, sigma, tile_points): transforms of 768 / 1024 / 1280 points on images of several bands and tiles; widths that are no
# multiples of 16 or 8; a pad of odd and of even parity; one band but several tiles and the other way round
FORCED = [(900, 700, 20.0, 768), (901, 733, 20.0, 768), (1300, 520, 12.0, 768), (420, 1500, 12.5, 768), (700, 900, 30.0, 1024), (1000, 1010, 25.0, 1280),
          (640, 2100, 9.0, 768), (2000, 300, 21.0, 1024), (555, 777, 19.5, 1536),
          # several middle bands of one shape: they run as the "frames" of one launch per kernel
          (2600, 200, 12.0, 768), (3301, 136, 10.0, 768), (2900, 1400, 20.0, 1024)]


@pytest.mark.parametrize("rows,cols,sigma,points", FORCED)
@pytest.mark.parametrize("quirk", [True, False])
def test_tiled_path_matches_the_oracle(ctx, rows, cols, sigma, points, quirk):
    import torch
    from oracle import oracle as O
    img = _rand_img(rows, cols, rows * 7 + cols)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=quirk, want_planes=True)
    buf = torch.full((img.size + 96,), 0xA5, dtype=torch.uint8, device="cuda")
    out = buf[32:32 + img.size].view(rows, cols, 3)
    got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma, out=out, nyquist_quirk=quirk, tile_points=points).cpu().numpy()
    assert ctx.last_family() == 7
    assert_u8_parity(got, want, planes)
    assert bool((buf[:32] == 0xA5).all()) and bool((buf[32 + img.size:] == 0xA5).all())


def test_tiled_path_in_place_and_batch(ctx):
    import torch
    from oracle import oracle as O
    frames = np.stack([_rand_img(800, 650, 11 + i) for i in range(3)])
    t = torch.from_numpy(frames).cuda()
    ctx.pffft_(t, 18.0, tile_points=768)                     # in place, three frames
    assert ctx.last_family() == 7
    got = t.cpu().numpy()
    for i in range(3):
        want, planes = O.pffft_blur_u8c3_f64(frames[i], 18.0, quirk=True, want_planes=True)
        assert_u8_parity(got[i], want, planes)


@pytest.mark.parametrize("rows,cols,sigma", [(5400, 3100, 60.0), (3200, 5200, 58.0), (2800, 4200, 64.8)])
def test_the_librarys_choice_for_long_lines(ctx, rows, cols, sigma):
    """kernels wider than 337 taps with a padded line longer than one wave-resident transform (4096 points), on images of 16 MP and
    more -- or of 10 MP and more while the columns still fit one transform (the third shape: columns whole on a two-line column
    kernel, rows in two tiles): the library's own choice is the tiled path, and it gives the oracle's bytes"""
    import torch
    from oracle import oracle as O
    img = _rand_img(rows, cols, rows + cols)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=True, want_planes=True)
    got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma).cpu().numpy()
    code, note = ctx.last_engine()
    assert code == 7 and "tiled" in note
    assert_u8_parity(got, want, planes)
