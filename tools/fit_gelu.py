"""Minimax fit of the normal CDF used by the 16-bit GELU / GELU' epilogues (csrc/common.h: phi_cdf16):
Phi(x) ~ 1 / (1 + exp(-x (c0 + c1 x^2 + c2 x^4))), objective max(|dPhi|, |x dPhi|) over [-9, 9]. Prints the coefficients and their
-log2(e) multiples (the form the kernel evaluates with v_exp_f32 = exp2). CPU only: python tools/fit_gelu.py"""
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf

x = np.linspace(-9, 9, 72001)
Phi = 0.5 * (1 + erf(x / np.sqrt(2)))


def model(c, x):
    t = x * np.polyval(c[::-1], x * x)
    return 1 / (1 + np.exp(-np.clip(t, -80, 80)))


def obj(c):
    d = model(c, x) - Phi
    return max(np.abs(d).max(), np.abs(x * d).max())


c = np.array([1.5926, 0.0752, -8.3e-4])
for _ in range(6):
    c = minimize(obj, c, method="Nelder-Mead", options=dict(xatol=1e-12, fatol=1e-12, maxiter=20000, maxfev=40000)).x
d = model(c, x) - Phi
print("c =", repr(c), " max|dPhi| %.3e  max|x dPhi| %.3e" % (np.abs(d).max(), np.abs(x * d).max()))
print("-log2(e) c =", repr(-c * np.log2(np.e)))
