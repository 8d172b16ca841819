"""GPU tests aimed at where the bugs were: memory safety of the specialised kernels (clamped
unconditional loads, 16-byte staged rows, 8-byte stores) checked with guard bands around every buffer the kernels touch,
unaligned frame pointers, and a time-boxed random parity run that aims at the lengths with specialised kernels."""
import ctypes as C
import time

import numpy as np
import pytest

from conftest import assert_u8_parity

pytestmark = pytest.mark.gpu

GUARD = 4096


def _torch():
    import torch
    return torch


# (rows, cols, sigma, mode): every specialised (length, role) pair of both kernel families, ragged and odd shapes
REDZONE_CASES = [
    (70, 3840, 20.0, "rows_first"),      # row 4000
    (71, 3839, 20.0, "rows_first"),
    (2160, 70, 20.0, "rows_first"),      # column 2304
    (2159, 67, 20.0, "rows_first"),
    (66, 1920, 20.0, "rows_first"),      # row 2304
    (1080, 66, 20.0, "rows_first"),      # column 1280
    (170, 3840, 50.0, "rows_first"),     # row 4320
    (2160, 170, 50.0, "rows_first"),     # column 2560
    (1081, 1923, 20.0, "rows_first"),    # both specialised, ragged last strip
    (2208, 15, 2.0, "rows_first"),       # fewer workgroups than XCDs
    (2160, 72, 20.0, "wave_resident"),   # wave-resident: column 9 x 256 / row 16 x 256
    (2159, 67, 20.0, "wave_resident"),
    (70, 3840, 20.0, "wave_resident"),
    (71, 3839, 20.0, "wave_resident"),
    (301, 8, 1.2, "wave_resident"),
    (33, 47, 2.0, "wave_resident"),
    (100, 77, 5.0, "generic"),
    (500, 748, 20.0, "generic"),
]


@pytest.mark.parametrize("rows,cols,sigma,mode", REDZONE_CASES)
def test_guard_bands_stay_intact(rows, cols, sigma, mode):
    """src, dst and the float workspace each sit between 0xA5 guard bands; the frame pointers are tried 256-byte aligned,
    odd and dword-odd (the unaligned ones force the byte-load / byte-store paths of the specialised kernels)."""
    torch = _torch()
    import blur_algorithms_amd as B
    from blur_algorithms_amd import _lib
    lib = _lib.load()
    lib.blur_debug_check_workspace_guards.argtypes = [C.c_void_p]
    lib.blur_debug_check_workspace_guards.restype = C.c_int
    kw = {"rows_first": dict(wave_resident=False), "wave_resident": dict(wave_resident=True), "generic": dict(force_generic=True)}[mode]
    n = rows * cols * 3
    img = torch.from_numpy(np.random.default_rng(rows * 7 + cols).integers(0, 256, (rows, cols, 3), dtype=np.uint8)).cuda()
    ctx0 = B.BlurContext(0)
    want = ctx0.pffft_(img, sigma, out=torch.empty_like(img), **kw)
    ctx0.close()
    for off in (0, 1, 4, 7):
        ctx = B.BlurContext(0)                      # a fresh workspace of exactly this case's size between its own guards
        sbuf = torch.full((n + 2 * GUARD + 16,), 0xA5, dtype=torch.uint8, device="cuda")
        dbuf = torch.full((n + 2 * GUARD + 16,), 0xA5, dtype=torch.uint8, device="cuda")
        s = sbuf[GUARD + off:GUARD + off + n].view(rows, cols, 3)
        d = dbuf[GUARD + off:GUARD + off + n].view(rows, cols, 3)
        s.copy_(img)
        ctx.pffft_(s, sigma, out=d, **kw)
        torch.cuda.synchronize()
        assert torch.equal(d, want), "result changed with the pointer offset %d" % off
        assert torch.equal(s, img), "the source frame was written to"
        for name, buf in (("src", sbuf), ("dst", dbuf)):
            assert bool((buf[:GUARD + off] == 0xA5).all()) and bool((buf[GUARD + off + n:] == 0xA5).all()), "%s guard band overwritten (offset %d)" % (name, off)
        assert lib.blur_debug_check_workspace_guards(ctx._h) == 0, "workspace guard band overwritten"
        # in place as well (the reference's calling convention)
        ctx.pffft_(s, sigma, **kw)
        torch.cuda.synchronize()
        assert torch.equal(s, want)
        assert bool((sbuf[:GUARD + off] == 0xA5).all()) and bool((sbuf[GUARD + off + n:] == 0xA5).all())
        ctx.close()


def test_targeted_fuzz_time_boxed(ctx):
    """tools/fuzz.py --targeted inside the suite: one side lands on a length that has specialised kernels (rows-first
    family) or fits the wave-resident family, the other side stays small so that the float64 oracle is cheap; ~40 s."""
    torch = _torch()
    from oracle import oracle as O
    rng = np.random.default_rng(20260104)
    t_end = time.time() + 40.0
    cases = 0
    while time.time() < t_end or cases < 12:
        family = rng.choice(["rows_first", "wave_resident"])
        sigma = float(rng.choice([2.0, 5.0, 11.0, 20.0, 33.0, 50.0]))
        if family == "rows_first":
            n_target = int(rng.choice([4000, 2304, 4320, 1280, 2560]))
            long_side = None
            for _ in range(50):
                cand = int(rng.integers(max(8, n_target - 400), n_target))
                if O.pffft_sizing(cand, cand, sigma)["N0"] == n_target:
                    long_side = cand
                    break
            if long_side is None:
                continue
            kw = dict(wave_resident=False)
        else:
            pad = O.pffft_sizing(4096, 4096, sigma)["pad"]
            as_col = rng.random() < 0.5
            long_side = int(rng.integers(300, (2304 if as_col else 4096) - 2 * pad))
            kw = dict(wave_resident=True)
        short = int(rng.integers(8, 200))
        if family == "rows_first":
            rows, cols = (short, long_side) if rng.random() < 0.5 else (long_side, short)
        else:
            rows, cols = (long_side, short) if as_col else (short, long_side)
        s = O.pffft_sizing(rows, cols, sigma)
        if s["pad"] > min(rows, cols) - 1:
            continue
        quirk = bool(rng.integers(0, 2))
        kind = rng.choice(["uniform", "binary"])
        img = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8) if kind == "uniform" else (rng.integers(0, 2, (rows, cols, 3)) * 255).astype(np.uint8)
        want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk, want_planes=True)
        t = torch.from_numpy(img).cuda()
        got = ctx.pffft_(t, sigma, out=torch.empty_like(t), nyquist_quirk=quirk, **kw).cpu().numpy()
        try:
            assert_u8_parity(got, want, planes)
        except AssertionError as e:
            raise AssertionError("rows=%d cols=%d sigma=%r quirk=%d family=%s kind=%s: %s" % (rows, cols, sigma, quirk, family, kind, e))
        cases += 1
    assert cases >= 12


@pytest.mark.parametrize("shards", [[0, 0], [0, 0, 0]])
def test_multi_gpu_c_abi_with_logical_shards_on_one_device(ctx, shards):
    """blur_multi_*: frames shard over per-device contexts and streams from one host thread (the C++ caller's 8-GPU
    path).  Here every shard sits on device 0; the bytes must equal the unsharded call, for device-resident frames (shards
    on the frames' device work in place) and for host frames (every shard copies its own slice in and out)."""
    torch = _torch()
    import blur_algorithms_amd as B
    frames = np.random.default_rng(77).integers(0, 256, (5, 96, 130, 3), dtype=np.uint8)
    t = torch.from_numpy(frames).cuda()
    want = ctx.pffft_(t, 4.0, out=torch.empty_like(t))
    m = B.BlurMulti(shards)
    got = m.pffft_(t, 4.0, out=torch.empty_like(t))
    assert torch.equal(got, want)
    host = m.pffft_(frames, 4.0)
    assert np.array_equal(host, want.cpu().numpy())
    # fewer frames than shards, and in place
    one = t[:1].contiguous()
    m.pffft_(one, 4.0)
    assert torch.equal(one[0], want[0])
    m.close()


def test_image_file_in_blur_file_out(ctx, tmp_path):
    """decode -> GPU blur -> encode (main() of the reference, Source.cpp:611-641), in Python and through examples/blur_main.cpp"""
    import os
    import subprocess
    from blur_algorithms_amd import io
    import blur_algorithms_amd as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    img = np.random.default_rng(9).integers(0, 256, (90, 121, 3), dtype=np.uint8)
    want = ctx.pffft_(img, 4.0)
    src, dst = str(tmp_path / "in.png"), str(tmp_path / "out.png")
    io.imwrite(src, img)
    got = io.blur_file(src, dst, 4.0, ctx=ctx)
    assert np.array_equal(got, want) and np.array_equal(io.imread(dst), want)
    exe = str(tmp_path / "blur_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "blur_main.cpp"),
                           "-L" + os.path.dirname(B.LIB_PATH), "-lblur_amd", "-Wl,-rpath," + os.path.dirname(B.LIB_PATH),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    ppm = str(tmp_path / "in.ppm")
    io.imwrite(ppm, img)
    assert subprocess.run([exe, "3", "4.0", ppm]).returncode == 0
    assert np.array_equal(io.imread(ppm + ".out.ppm"), want)
