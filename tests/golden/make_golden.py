#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build container, where
/root/reference is mounted).  Two kinds, kept apart:

  ref_*.npz     PINNED vectors: outputs of the REFERENCE'S OWN CODE (oracle/_ref/libref_utils.so
                = /root/reference/Utils.hpp + Source.cpp:58-102 compiled where they lie) for the
                host-side functions of the hot path: gaussian_window, getGaussian, isValidSize,
                nearestTransformSize, deinterleave_BGR / interleave_BGR, Reflect_101.
  img_*.npz     decoded crops of the reference's test_images (inputs only -- the reference holds
                no output of pffft_() for them) plus the float64 ORACLE's blur of
                each (oracle-generated, hence "parity unpinned" for the FFT part; they freeze the
                oracle against regressions and let the GPU box test on natural images).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

REF = "/root/reference"


def ref_vectors():
    R = O.ref()
    rng = np.random.default_rng(20241108)
    sig = np.concatenate([np.array([0.1, 0.3, 0.5, 0.84, 1, 1.5, 2, 3, 3.3, 5, 7.7, 10, 20, 38.7298, 50, 106.77, 200.0]),
                          rng.uniform(0.05, 300, 200)])
    maxw = np.concatenate([np.zeros(len(sig) - 100, np.int64), rng.integers(1, 5000, 100)])
    gw = np.array([R.ref_gaussian_window(float(s), int(m)) for s, m in zip(sig, maxw)], np.int32)
    n = np.arange(0, 13001, dtype=np.int32)
    valid = np.array([R.ref_is_valid_size(int(i)) for i in n], np.int8)
    near = np.array([R.ref_nearest_transform_size(int(i)) for i in n], np.int32)
    out = dict(gw_sigma=sig, gw_maxw=maxw, gw_width=gw, size_n=n, size_valid=valid, size_nearest=near)
    # getGaussian the way pffft_() calls it: (sigma, kSize, FFT_length), plus unpadded forms
    cases = [(5.0, 31, 576), (20.0, 131, 4000), (20.0, 131, 2304), (20.0, 131, 1280), (50.0, 331, 4320), (50.0, 331, 2560),
             (3.0, 19, 128), (1.0, 7, 32), (0.5, 0, 0), (2.5, 0, 0), (7.0, 0, 64), (38.7298, 257, 1792), (11.0, 9, 96)]
    out["gk_cases"] = np.array(cases, np.float64)
    for i, (s, w, f) in enumerate(cases):
        out["gk_%d" % i] = O.ref_get_gaussian(s, int(w), int(f))
    # de/interleave, including the +0.5f / truncation rule on awkward values
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    pl = np.empty((3, 37 * 53), np.float32)
    R.ref_deinterleave_bgr_u8_f32(img.reshape(-1), pl[0], pl[1], pl[2], 37 * 53)
    vals = np.concatenate([rng.uniform(0, 255.49, 3 * 2000 - 12), [0.0, 0.49999997, 0.5, 0.50000006, 1.4999999, 1.5, 254.5, 254.50002,
                                                                   255.0, 255.49998, 127.5, 128.49999]]).astype(np.float32).reshape(3, 2000)
    inter = np.empty(3 * 2000, np.uint8)
    R.ref_interleave_bgr_f32_u8(vals[0], vals[1], vals[2], inter, 2000)
    out.update(di_img=img, di_planes=pl, il_planes=vals, il_out=inter)
    # Reflect_101, incl. the README example (length 7, pad 6) and clamped pads
    rcases = []
    a = np.arange(1, 8, dtype=np.uint8).reshape(1, 7)
    rcases.append((a, (0, 0, 6, 6)))
    rcases.append((rng.integers(0, 256, (9, 11, 3), dtype=np.uint8), (3, 2, 4, 5)))
    rcases.append((rng.integers(0, 256, (5, 4, 3), dtype=np.uint8), (10, 10, 10, 10)))     # clamped to dim-1
    rcases.append((rng.standard_normal((6, 8)).astype(np.float32), (2, 5, 7, 1)))
    rcases.append((rng.integers(0, 256, (8, 6), dtype=np.uint8), (1, 0, 0, 3)))
    out["rf_n"] = np.array(len(rcases))
    for i, (arr, pads) in enumerate(rcases):
        out["rf_in_%d" % i] = arr
        out["rf_pads_%d" % i] = np.array(pads, np.int32)
        out["rf_out_%d" % i] = O.ref_reflect_101(arr, *pads)
    # pffft_sorted_optimized_convolution (Source.cpp:414-427) on random ordered spectra: slot 0 (DC) and slot 1 (the Nyquist bin of
    # pffft's ordered layout) are both scaled with kernel_dft[0] -- the quirk -- and every other pair with its own real part
    for i, nn in enumerate((32, 96, 576, 2304, 4000)):
        tile = rng.standard_normal(nn).astype(np.float32) * 100
        kern = rng.uniform(0.0, 1.0, nn).astype(np.float32)
        out["soc_tile_%d" % i] = tile
        out["soc_kernel_%d" % i] = kern
        out["soc_out_%d" % i] = O.ref_sorted_optimized_convolution(tile, kern, 1.0 / nn)
    out["soc_n"] = np.array(5)
    np.savez_compressed(os.path.join(HERE, "ref_host_functions.npz"), **out)
    print("ref_host_functions.npz: %d gaussian_window cases, %d sizes, %d kernels, %d reflect cases" % (len(sig), len(n), len(cases), len(rcases)))


def image_vectors():
    from PIL import Image
    picks = [
        ("colourgram", "Test 2/Colourgrams", None, (0, 0, 256, 192), 5.0),
        ("collage_top", "spectrum_analysis/blur/collage/0.png", None, (300, 200, 300 + 320, 200 + 240), 20.0),
        ("baseline", "More Clean Up Comparisons/Baseline.jpg", None, (100, 50, 100 + 333, 50 + 251), 20.0),   # odd sizes
        ("input7", "More Clean Up Comparisons/input7.png", None, (0, 0, 200, 150), 3.0),
    ]
    for name, rel, _, box, sigma in picks:
        path = os.path.join(REF, "test_images", rel)
        if os.path.isdir(path):
            path = os.path.join(path, sorted(f for f in os.listdir(path) if f.lower().endswith((".jpg", ".png")))[0])
        im = Image.open(path).convert("RGB")
        x0, y0, x1, y1 = box
        x1, y1 = min(x1, im.size[0]), min(y1, im.size[1])
        rgb = np.asarray(im.crop((x0, y0, x1, y1)), np.uint8)
        bgr = np.ascontiguousarray(rgb[:, :, ::-1])          # cv::imread order (Source.cpp:623)
        want, planes = O.pffft_blur_u8c3_f64(bgr, sigma, True, want_planes=True)
        want_nq = O.pffft_blur_u8c3_f64(bgr, sigma, False)
        np.savez_compressed(os.path.join(HERE, "img_%s.npz" % name), src=bgr, sigma=np.float64(sigma), oracle_u8=want,
                            oracle_planes=planes, oracle_u8_noquirk=want_nq,
                            sha256=np.array(hashlib.sha256(bgr.tobytes()).hexdigest()),
                            source=np.array("%s crop %s" % (os.path.relpath(path, REF), (x0, y0, x1, y1))))
        print("img_%s.npz: %s %s sigma=%g" % (name, os.path.basename(path), bgr.shape, sigma))


if __name__ == "__main__":
    if not (os.path.exists(REF) and O.ref_available()):
        raise SystemExit("needs /root/reference and oracle/_ref (make -C oracle ref)")
    ref_vectors()
    image_vectors()
