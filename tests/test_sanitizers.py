"""CPU sanitizer job (the reference's only hook of this kind is the commented LSan include, Utils.hpp:12): the product's
host arithmetic (csrc/host_math.cpp), the C++ header with the reference's names (include/blur_amd.hpp, host half) and the oracle's
C restatement are built with -fsanitize=address,undefined and run over their vectors.  No GPU, no HIP library in these binaries
(GPU AddressSanitizer is not available on this pool)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
INC = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "blur_algorithms_amd", "csrc")]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="4")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None or shutil.which("gcc") is None, reason="no host compiler")


def _build(tmp_path, name, cxx_sources, c_sources=(), flags=()):
    objs = []
    for src in c_sources:
        o = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.check_call(["gcc", "-std=c11"] + SAN + ["-fopenmp", "-ffp-contract=off", "-c", src, "-o", o])
        objs.append(o)
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17"] + SAN + list(flags) + INC + list(cxx_sources) + objs + ["-fopenmp", "-lm", "-o", exe])
    return exe


def test_host_arithmetic_and_oracle_under_asan_ubsan(tmp_path):
    exe = _build(tmp_path, "sanitize_vectors",
                 [os.path.join(ROOT, "tests", "cpp", "sanitize_vectors.cpp"), os.path.join(ROOT, "blur_algorithms_amd", "csrc", "host_math.cpp")],
                 [os.path.join(ROOT, "oracle", "blur_oracle.c"), os.path.join(ROOT, "oracle", "boxblur_oracle.c")])
    out = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=600)
    assert out.returncode == 0 and "sanitize ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.parametrize("mode", [["-DOMP", "-fopenmp"], ["-DMYLOOP", "-pthread"], ["-DSINGLE"]])
def test_cpp_surface_host_half_under_asan_ubsan(tmp_path, mode):
    """tests/cpp/surface_check.cpp "host" (gaussian_window, getGaussian, nearestTransformSize, Reflect_101, de/interleave_BGR, flip_block,
    hybrid_loop, PFAlloc / AlignedVector in the reference's three threading modes) over the host entry points without HIP"""
    exe = _build(tmp_path, "surface_check_san",
                 [os.path.join(ROOT, "tests", "cpp", "surface_check.cpp"), os.path.join(ROOT, "tests", "cpp", "host_abi_for_sanitizers.cpp"),
                  os.path.join(ROOT, "blur_algorithms_amd", "csrc", "host_math.cpp")], flags=mode)
    out = subprocess.run([exe, "host"], capture_output=True, text=True, env=ENV, timeout=600)
    assert out.returncode == 0 and "host ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
