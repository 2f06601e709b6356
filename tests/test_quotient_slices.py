"""CPU check of the slice algebra behind PlonkKzgSnark's quotient round (jellyfish_amd/plonk.py quotient_slices /
unslice_matrix): a polynomial with fewer coefficients than the slices have points is rebuilt from its residues
mod X^(ml_j) - c_j.  The residues are what a coset iNTT of the slice evaluations returns (reference: the 8 n point
evaluation of plonk/src/proof_system/prover.rs:524-661, of which the slices are sub-cosets)."""
import random

import pytest

from jellyfish_amd import hostglue as H
from jellyfish_amd.plonk import quotient_slices, slice_offset, unslice_matrix


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("wires", [5, 6])
def test_unslice_rebuilds_polynomial(curve, wires):
    cp = H.CURVES[curve]
    r = cp.r
    n = 16
    m = 8 * n
    ncoeffs = wires * (n + 1) + 3
    slices = quotient_slices(n, m, ncoeffs)
    assert slices == ([(0, 2), (1, 4)] if wires == 5 else [(0, 2), (1, 4), (3, 8)])
    rnd = random.Random(7 + curve + wires)
    t = [rnd.randrange(r) for _ in range(ncoeffs)]
    s, inv = unslice_matrix(cp, m, slices)
    chunks = []
    for g, G in slices:
        ml = m // G
        cj = pow(slice_offset(cp, m, g), ml, r)
        res = [0] * ml
        for k, a in enumerate(t):                     # X^k = c^(k // ml) X^(k % ml)
            res[k % ml] = (res[k % ml] + a * pow(cj, k // ml, r)) % r
        chunks += [res[e * s:(e + 1) * s] for e in range(ml // s)]
    K = len(chunks)
    assert K * s >= ncoeffs
    back = []
    for k in range(K):
        back += [sum(inv[k][j] * chunks[j][x] for j in range(K)) % r for x in range(s)]
    assert back[:ncoeffs] == t and not any(back[ncoeffs:])


def test_slices_fall_back():
    assert quotient_slices(8, 64, 50) == [(0, 1)]          # tiny domain
    assert quotient_slices(64, 256, 300) == [(0, 1)]       # ratio other than 8
    assert quotient_slices(64, 512, 7 * 64 + 1) == [(0, 1)]  # more coefficients than 7 n points
