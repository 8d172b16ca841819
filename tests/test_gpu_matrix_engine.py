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
