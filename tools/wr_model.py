#!/usr/bin/env python3
"""numpy model of the wave-resident FFT convolution (wr_kernels.hpp): checks the index conventions
(pass 0 of radix R0 through LDS, 256-point sub-FFTs on 16 lanes x 16 registers with a lane<->register
transpose built from xor-exchanges, natural-order multiplier lookup) against numpy's FFT.
Run: python3 tools/wr_model.py"""
import numpy as np

def dft(v, inv=False):
    n = len(v)
    k = np.arange(n)
    w = np.exp((2j if inv else -2j) * np.pi * np.outer(k, k) / n)
    return w @ v

def transpose16(v):
    """v[r][l] (16 regs x 16 lanes) -> v[l][r] with the four xor-exchange steps of the kernel"""
    v = v.copy()
    lanes = np.arange(16)
    for s in range(4):
        d = 1 << s
        for r in range(16):
            if r & d: continue
            a, b = v[r].copy(), v[r | d].copy()
            hi = (lanes & d) != 0
            v[r] = np.where(hi, b[lanes ^ d], a)          # a'[l] = (l&d) ? b[l^d] : a[l]
            v[r | d] = np.where(hi, b, a[lanes ^ d])      # b'[l] = (l&d) ? b[l] : a[l^d]
    return v

def wr_conv(x, M, R0):
    N = R0 * 256
    assert len(x) == N and len(M) == N
    wN = lambda e: np.exp(-2j * np.pi * (e % N) / N)
    w256 = lambda e: np.exp(-2j * np.pi * (e % 256) / 256)
    # pass 0
    X1 = np.zeros(N, complex)
    for j in range(256):
        V = dft(x[j::256])
        for q in range(R0): X1[q * 256 + j] = V[q] * wN(j * q)
    # middle, per sub-block
    for q in range(R0):
        d = X1[q * 256:(q + 1) * 256]
        v = np.zeros((16, 16), complex)               # v[reg][lane]
        for l in range(16):
            A = dft(d[l::16])                          # over k, element 16k + l
            for k in range(16): v[k][l] = A[k] * w256(l * k)
        v = transpose16(v)                             # lane k', reg l
        for lam in range(16):
            Y = dft(v[:, lam])                         # over l -> rho
            for rho in range(16): v[rho][lam] = Y[rho] * M[q + R0 * (lam + 16 * rho)]
        for lam in range(16):
            B = dft(v[:, lam], inv=True)               # over rho -> l
            for l in range(16): v[l][lam] = B[l] * np.conj(w256(l * lam))
        v = transpose16(v)                             # lane l, reg lam
        for l in range(16):
            o = dft(v[:, l], inv=True)                 # over lam -> k
            for k in range(16): d[16 * k + l] = o[k]
    # inverse pass 0
    y = np.zeros(N, complex)
    for j in range(256):
        V = np.array([X1[q * 256 + j] * np.conj(wN(j * q)) for q in range(R0)])
        y[j::256] = dft(V, inv=True)
    return y

if __name__ == "__main__":
    rng = np.random.default_rng(1)
    t = rng.standard_normal((16, 16))
    assert np.array_equal(transpose16(t), t.T), "transpose"
    for R0 in (5, 9, 16):
        N = R0 * 256
        x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        M = rng.standard_normal(N)
        ref = np.fft.ifft(np.fft.fft(x) * M) * N
        got = wr_conv(x, M, R0)
        print(R0, np.abs(got - ref).max() / np.abs(ref).max())
