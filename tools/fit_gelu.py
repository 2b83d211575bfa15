"""Exact-erf GELU without erff:  gelu(x) = max(x, 0) - |x| * Phi(-|x|),  Phi(-a) = 2^(-r(a)),  r a polynomial.

    python tools/fit_gelu.py            # error table of the coefficient sets the kernels use (tools/kgen/gemm_z192_gen.py GELU_FITS)
    python tools/fit_gelu.py --fit      # re-derive the coefficients (iteratively re-weighted least squares -> approx. minimax)

What is minimised is max_a |a 2^-r(a) - a Phi(-a)|, the absolute error of the GELU output itself.  The table also gives the error
AFTER the result is rounded to fp16 (what the hidden activation is stored as), next to the exact function rounded to fp16, and the
same for the polynomial evaluated in PACKED fp16 (v_pk_fma_f16 on the already-converted operand — the form the round-4 verdict asked
to be measured): degree 3 in f32 — what gemm_z192 ships since round 5, 6 VALU per element instead of 8 — adds 1.8 % to the rms error of
exact-then-rounded; degree 4 is indistinguishable from exact-then-rounded but its optimum has a negative leading coefficient and
overflows beyond |x| ~ 18 (unusable without a clamp, which costs the instruction it saves); packed fp16 adds 50 % and is not used.
Reference: nn.GELU() (exact erf) in the SAM fork's MLPBlock, reached through /root/reference/model.py:245-258."""
import sys

import numpy as np
from scipy.special import erf, log_ndtr


def fit(deg, free_const=True):
    from scipy.optimize import least_squares
    a = np.linspace(0, 12.0, 40001)
    r_true = -log_ndtr(-a) / np.log(2.0)          # -log2(Phi(-a))
    target = a * np.exp(log_ndtr(-a))             # a * Phi(-a)
    V = np.vander(a, deg + 1, increasing=True)
    if not free_const:
        V = V[:, 1:]
    base = 0.0 if free_const else 1.0
    sel = a < 5
    c = np.linalg.lstsq(V[sel], (r_true - base)[sel], rcond=None)[0]

    def res(c):
        return a * np.exp2(-np.clip(base + V @ c, -50, 200)) - target
    e = res(c)
    for it in range(60):
        w = np.ones_like(a) if it == 0 else (np.abs(e) / np.abs(e).max() + 0.02)
        c = least_squares(lambda cc: res(cc) * w, c, method="lm", xtol=1e-15, ftol=1e-15).x
        e = res(c)
    coef = list(c) if free_const else [1.0] + list(c)
    return coef[::-1], float(np.abs(e).max())      # highest power first


def gelu_f32(x, c):
    a = np.abs(x)
    r = np.float32(c[0])
    for k in c[1:]:
        r = (r * a + np.float32(k)).astype(np.float32)
    return (np.maximum(x, 0) - a * np.exp2(-r).astype(np.float32)).astype(np.float32)


def gelu_pk_f16(x, c):
    xh = x.astype(np.float16)
    a = np.abs(xh)
    r = np.float16(c[0])
    for k in c[1:]:
        r = (r.astype(np.float32) * a.astype(np.float32) + np.float32(np.float16(k))).astype(np.float16)
    e = np.exp2(-r.astype(np.float32)).astype(np.float16)
    return (np.maximum(xh, 0).astype(np.float32) - a.astype(np.float32) * e.astype(np.float32)).astype(np.float16)


def table(fits):
    x = np.linspace(-8, 8, 2_000_001).astype(np.float32)
    ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    core = np.abs(x) < 3
    rms = lambda d: float(np.sqrt(np.mean(d[core] ** 2)))
    ex = ref.astype(np.float16).astype(np.float64)
    print(f"exact erf, rounded to fp16:                      max {np.abs(ex - ref).max():.3e}   rms(|x|<3) {rms(ex - ref):.3e}")
    out = {}
    for deg, c in sorted(fits.items(), reverse=True):
        y = gelu_f32(x, c)
        y16 = y.astype(np.float16).astype(np.float64)
        h = gelu_pk_f16(x, c).astype(np.float64)
        out[deg] = (float(np.abs(y - ref).max()), rms(y16 - ref), rms(h - ref))
        print(f"degree {deg}: f32 evaluation max abs error {out[deg][0]:.3e};  after fp16 rounding rms(|x|<3) {out[deg][1]:.3e};"
              f"  packed-fp16 evaluation rms(|x|<3) {out[deg][2]:.3e}")
    return out


if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "kgen"))
    from gemm_z192_gen import GELU_FITS
    if "--fit" in sys.argv:
        for deg, free in ((5, False), (4, True), (3, True)):
            c, err = fit(deg, free)
            print(deg, f"max abs err {err:.3e}", c)
    table(GELU_FITS)
