#!/usr/bin/env python3
"""Derivation and check of the GELU evaluation in hirest_amd/csrc/common.h:
   gelu(x) = max(x,0) - |x| * Phi(-|x|),  Phi(-a) = exp2(Q(a)),  Q = degree-8 fit of log2(0.5*erfcx(a/sqrt2)) - a^2/2*log2(e).
Prints the coefficients (highest first, as used by the Horner chain) and the fp32-evaluated error against x*ndtr(x)."""
import math
from math import comb
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erfcx, ndtr

AMAX, DEG = 9.4, 8
z = np.cos(np.pi * (np.arange(6000) + 0.5) / 6000)
a = (z + 1) / 2 * AMAX
c = C.chebfit(z, np.log2(erfcx(a / math.sqrt(2)) * 0.5), DEG)
pm, k, pa = C.cheb2poly(c), 2 / AMAX, np.zeros(DEG + 1)
for n, cn in enumerate(pm):
    for j in range(n + 1):
        pa[j] += cn * comb(n, j) * (k ** j) * ((-1) ** (n - j))
pa[2] -= 0.5 * math.log2(math.e)
print("coefficients, highest power first:", [float(np.float32(v)) for v in pa[::-1]])
f = np.float32
x = np.linspace(-12, 12, 2000001).astype(np.float32)
av = np.minimum(np.abs(x), f(AMAX))
q = np.full_like(av, f(pa[-1]))
for cc in pa[-2::-1]:
    q = (q * av + f(cc)).astype(np.float32)
g = (np.maximum(x, f(0)) - av * np.exp2(q.astype(np.float64)).astype(np.float32)).astype(np.float32)
ref = x.astype(np.float64) * ndtr(x.astype(np.float64))
err = np.abs(g - ref)
ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 1e-300))) - 7)
m = (np.abs(ref) > 1e-30) & (np.abs(x) <= AMAX)
print("max abs error", err.max(), " max error in bf16 ulps of the result, |x| <= 9.4:", (err / ulp)[m].max())
