"""The reference's own benchmark shape (Source.cpp:627-635: rows 1500 + 225 i, cols 1000 + 150 i, sigma = sqrt(rows)) against
the float64 oracle, for the sizes the oracle finishes in seconds; bench.py --preset reference-sweep times all 45."""
import math

import numpy as np
import pytest

from conftest import assert_u8_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("step", [0, 1, 2, 3, 5])
def test_reference_sweep_sizes_against_the_oracle(ctx, step):
    import torch
    from oracle import oracle as O
    rows, cols = 1500 + 225 * step, 1000 + 150 * step
    sigma = math.sqrt(rows)
    img = np.random.default_rng(step).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    want, planes = O.pffft_blur_u8c3_f64(img, sigma, True, want_planes=True)
    t = torch.from_numpy(img).cuda()
    got = ctx.pffft_(t, sigma, out=torch.empty_like(t))
    assert_u8_parity(got.cpu().numpy(), want, planes)


@pytest.mark.parametrize("step", [8, 12, 16, 20, 24, 28, 32, 36, 40, 44])
def test_reference_sweep_sizes_as_thin_images_against_the_oracle(ctx, step):
    """the larger sizes of the sweep (up to the last: 11400 x 7600, sigma = 106.8, 709 taps) as two thin images each -- full height x a
    few hundred columns and a few hundred rows x full width: the column pass, respectively the row pass, runs at the sweep size's own
    transform length and kernel, and the float64 oracle stays cheap"""
    import torch
    from oracle import oracle as O
    rows, cols = 1500 + 225 * step, 1000 + 150 * step
    sigma = math.sqrt(rows)
    pad = O.pffft_sizing(rows, cols, sigma)["pad"]
    thin = pad + 1 + 37 + 4 * step
    for r, c in ((rows, thin), (thin, cols)):
        assert O.pffft_sizing(r, c, sigma)["pad"] == pad
        img = np.random.default_rng(100 + step).integers(0, 256, (r, c, 3), dtype=np.uint8)
        want, planes = O.pffft_blur_u8c3_f64(img, sigma, True, want_planes=True)
        t = torch.from_numpy(img).cuda()
        got = ctx.pffft_(t, sigma, out=torch.empty_like(t))
        assert_u8_parity(got.cpu().numpy(), want, planes)
