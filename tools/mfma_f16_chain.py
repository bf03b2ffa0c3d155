"""tools/mfma_f16_chain.py gen <in.bin> | check <in.bin> <out.bin> -- the hi*hi part of the split-precision decimator
(cfg3 taps, a full-scale signal whose energy sits in the stop band) as MFMA operand tiles for tools/mfma_f16_chain, and
the comparison of the hardware's accumulated result with the exact sum of the same products."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "selenite-lite_amd"))
from selenite_rx import chain as ch

KS, NROW = 10, 64


def tiles():
    h = np.asarray(ch.baseline_spec("cfg3", 1).dec, np.float32)
    rng = np.random.default_rng(3)
    N = 64 * NROW + 320
    n = np.arange(N)
    x = sum(a * np.cos(2 * np.pi * rng.uniform(0.15, 0.5) * n + rng.uniform(0, 6)) for a in (0.4, 0.2, 0.1)) + rng.uniform(-0.05, 0.05, N)
    x = x.astype(np.float32)
    s = 14 - int(np.floor(np.log2(np.abs(x).max())))
    xh = np.ldexp(x, s).astype(np.float32).astype(np.float16).astype(np.float32)
    sc = 14 - int(np.floor(np.log2(np.abs(h).max())))
    hh = np.ldexp(h, sc).astype(np.float32).astype(np.float16).astype(np.float32)
    ncase = NROW // 16
    T = np.zeros((ncase, KS, 2, 512), np.float32)
    k = np.arange(32)
    for cs in range(ncase):
        for kk in range(KS):
            r = 16 * cs + np.arange(16)
            T[cs, kk, 0] = xh[64 * r[:, None] + 32 * kk + k[None, :]].ravel()          # A[i][k]: sample 64 (16 cs + i) + 32 kk + k
            ti = 32 * kk + k[:, None] - 4 * np.arange(16)[None, :]                    # B[k][n]: tap index K - 4 n, reversed
            T[cs, kk, 1] = np.where((ti >= 0) & (ti < 256), hh[255 - np.clip(ti, 0, 255)], 0.0).ravel()
    return T, x


if sys.argv[1] == "gen":
    T, _ = tiles()
    with open(sys.argv[2], "wb") as f:
        np.array([T.shape[0], KS], np.int32).tofile(f)
        T.tofile(f)
    print("wrote", T.shape)
else:
    T, x = tiles()
    D = np.fromfile(sys.argv[3], np.float32).reshape(T.shape[0], 16, 16).astype(np.float64)
    A = T[:, :, 0].reshape(-1, KS, 16, 32).astype(np.float64)
    B = T[:, :, 1].reshape(-1, KS, 32, 16).astype(np.float64)
    exact = np.einsum("csik,cskn->cin", A, B)
    absum = np.einsum("csik,cskn->cin", np.abs(A), np.abs(B))
    e = np.abs(D - exact)
    print("result magnitude / sum|products|: median %.3f" % np.median(np.abs(exact) / absum))
    print("hardware chain error / sum|products|: max 2^%.2f rms 2^%.2f" % (np.log2((e / absum).max()), np.log2(np.sqrt(((e / absum) ** 2).mean()))))
    print("hardware chain error / max|result|: max %.3g" % (e.max() / np.abs(exact).max()))
    # f32 sequential accumulation of the same products, for scale
    acc = np.zeros(exact.shape, np.float32)
    for s_ in range(KS):
        for k_ in range(32):
            acc = (acc + (A[:, s_, :, k_, None] * B[:, s_, None, k_, :]).astype(np.float32)).astype(np.float32)
    e32 = np.abs(acc - exact)
    print("f32 one-at-a-time accumulation of the same products: error / sum|products| max 2^%.2f" % np.log2((e32 / absum).max()))
