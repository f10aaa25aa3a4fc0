"""The radix-2 split of the folded DFT stages (route (c)) checked in NumPy against the direct sums the kernels
evaluate today (proposal_device.h: U[y] = sum_{k <= h} P[k] cos(2 pi k y / n), V[y] = sum_{k < h} M[k] sin(2 pi k y / n), y in [0, h],
n = 2 h).  Split by the parity of k:  U[y] = Ue[y] + Uo[y],  U[h - y] = Ue[y] - Uo[y],  V[y] = Ve[y] + Vo[y],  V[h - y] = -Ve[y] + Vo[y]
with the half-sums evaluated for y in [0, h // 2] only.  Prints the largest deviation and the product counts of both forms.

    python scripts/radix2_dft_check.py"""
import numpy as np

rng = np.random.default_rng(5)
worst = 0.0
for n in range(8, 122, 2):                      # block sizes are even (MCMC.py:576-579: // 2 * 2)
    h = n // 2
    P = rng.normal(size=h + 1)                  # P[k], k = 0 .. h   (P[0] and P[h] are the self-conjugate terms)
    M = np.concatenate([rng.normal(size=h), [0.0]]); M[0] = 0.0      # M[k] = 0 for k in {0, h}
    k = np.arange(h + 1)
    y = np.arange(h + 1)
    U = (P[:, None] * np.cos(2 * np.pi * np.outer(k, y) / n)).sum(axis=0)
    V = (M[:, None] * np.sin(2 * np.pi * np.outer(k, y) / n)).sum(axis=0)
    yh = np.arange(h // 2 + 1)                  # the outputs that are computed; h - y mirrors them
    ke, ko = k[0::2], k[1::2]
    Ue = (P[ke, None] * np.cos(2 * np.pi * np.outer(ke, yh) / n)).sum(axis=0)
    Uo = (P[ko, None] * np.cos(2 * np.pi * np.outer(ko, yh) / n)).sum(axis=0)
    Ve = (M[ke, None] * np.sin(2 * np.pi * np.outer(ke, yh) / n)).sum(axis=0)
    Vo = (M[ko, None] * np.sin(2 * np.pi * np.outer(ko, yh) / n)).sum(axis=0)
    U2 = np.empty(h + 1); V2 = np.empty(h + 1)
    U2[yh] = Ue + Uo; U2[h - yh] = Ue - Uo      # y = h / 2 (h even) is its own mirror: Uo = 0 there, both writes agree
    V2[h - yh] = -Ve + Vo; V2[yh] = Ve + Vo
    scale = np.abs(U).max() + np.abs(V).max()
    worst = max(worst, np.abs(U2 - U).max() / scale, np.abs(V2 - V).max() / scale)
print(f"largest relative deviation over n = 8 .. 120: {worst:.2e}")

def r(x, m): return (x + m - 1) // m * m
tot_old = tot_new = 0
for bh in (50, 56, 64, 72, 80):
    nrow = bh // 2 + 1; hh = bh // 2
    old = (r(nrow, 16) // 16) * (r(nrow, 4) // 4)                               # y tiles x K steps per column tile and (re | im) unit
    ne, no = hh // 2 + 1, (hh + 1) // 2                                          # terms of the even / odd half-sum
    ny = hh // 2 + 1
    new = (r(ny, 16) // 16) * (r(ne, 4) // 4 + r(no, 4) // 4)
    tot_old += old; tot_new += new
    print(f"bh {bh}: stage-1 tile-K-steps per column tile and unit {old} -> {new}")
print(f"table of BASELINE configs[1]: {tot_old} -> {tot_new} ({100 * (1 - tot_new / tot_old):.0f} % fewer)")
