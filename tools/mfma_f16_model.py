"""tools/mfma_f16_model.py <cases.bin> -- compares the results of v_mfma_f32_16x16x32_f16 dumped by tools/mfma_f16_cases
with adder models.  Model parameters: group size, in-group window, final window."""
import sys
import numpy as np

NC = 256
raw = np.fromfile(sys.argv[1], np.float32)
A = raw[:NC * 512].reshape(NC, 16, 32).astype(np.float64)
B = raw[NC * 512:2 * NC * 512].reshape(NC, 32, 16).astype(np.float64)
C = raw[2 * NC * 512:2 * NC * 512 + NC * 256].reshape(NC, 16, 16).astype(np.float64)
D = raw[2 * NC * 512 + NC * 256:].reshape(NC, 16, 16).astype(np.float64)
P = A[:, :, :, None] * B[:, None, :, :]                 # [case, i, k, n] exact
P = np.moveaxis(P, 2, 3)                                 # [case, i, n, k]
exact = P.sum(axis=3) + C
ulp = np.ldexp(1.0, np.floor(np.log2(np.maximum(np.abs(D), 1e-300))).astype(int) - 23)
print("hardware vs exact: max |err| %.2f ulp(result), mean signed %.3f, rms %.2f" % (np.abs((D - exact) / ulp).max(), ((D - exact) / ulp).mean(), np.sqrt((((D - exact) / ulp) ** 2).mean())))
# error relative to the largest addend of the whole op
big = np.maximum(np.abs(P).max(axis=3), np.abs(C))
print("hardware |err| / largest addend: max 2^%.2f" % np.log2((np.abs(D - exact) / big).max()))


def trunc_to(v, q):
    return np.trunc(v / q) * q


def rnd_to(v, q):
    return np.rint(v / q) * q


def rn24(v):
    return v.astype(np.float32).astype(np.float64)


def model(group, wg, wf, ground=trunc_to, fround=trunc_to):
    G = P.reshape(NC, 16, 16, 32 // group, group)
    gmax = np.abs(G).max(axis=4)
    E = np.floor(np.log2(np.where(gmax > 0, gmax, 1.0))).astype(int)
    gs = ground(G, np.ldexp(1.0, E - wg)[..., None]).sum(axis=4)
    allv = np.concatenate([C[..., None], gs], axis=3)
    m = np.abs(allv).max(axis=3)
    Ef = np.floor(np.log2(np.where(m > 0, m, 1.0))).astype(int)
    return rn24(fround(allv, np.ldexp(1.0, Ef - wf)[..., None]).sum(axis=3))


for group in (32, 16, 8, 4):
    for wg in (23, 24, 25, 26, 27):
        for wf in (23, 24, 25, 26, 27):
            M = model(group, wg, wf)
            bad = (M != D).sum()
            if bad < 0.2 * D.size:
                print("group %2d in-group window 2^-%d final window 2^-%d: %6d of %d results differ" % (group, wg, wf, bad, D.size))

# --- diagnostics: granularity of the result relative to the largest addend
Emax = np.floor(np.log2(big)).astype(int)
for w in range(18, 30):
    q = np.ldexp(1.0, Emax - w)
    frac = np.abs(D / q - np.rint(D / q))
    print("results that are multiples of 2^(Emax-%d): %.4f" % (w, (frac == 0).mean()))
