"""The algebra behind pld.hip's moment-form Gram (pld_moment_gram_kernel / moment_plan), restated in numpy and checked on
CPU: the Gram matrix of the order-o products of k components is determined by the moments of the canonical
(o smallest | o largest) splits, every canonical pair lies inside the staircase of 16 x 16 tiles the kernel computes, and
the tile count bench.py prices the kernel with is the one this construction gives.  (The kernel itself is compared with
the reference's PLDCorrector in tests/test_pld_gpu.py.)"""
import importlib.util
import itertools
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(k, o):
    """moment_plan of pld.hip: row order, column starts, wave tiles (first row, first column, 16-bit mask)."""
    cols = list(itertools.combinations_with_replacement(range(k), o))
    pc = len(cols)
    rperm = sorted(range(pc), key=lambda i: (cols[i][-1], i))          # rows by largest factor (stable)
    colstart = [sum(1 for c in cols if c[0] < m) for m in range(k + 2)]  # columns are sorted by smallest factor already
    wt = []
    for r0 in range(0, pc, 64):
        rt = [min((cols[x][-1] for x in rperm[r0 + 16 * i:r0 + 16 * i + 16]), default=None) for i in range(4)]
        mn = min(x for x in rt if x is not None)
        for cg in range((colstart[mn] // 16) * 16, pc, 64):
            mask = 0
            for i in range(4):
                for j in range(4):
                    if rt[i] is not None and cg + 16 * j < pc and cg + 16 * j + 16 > colstart[rt[i]]:
                        mask |= 1 << (4 * i + j)
            if mask:
                wt.append((r0, cg, mask))
    return cols, rperm, wt


@pytest.mark.parametrize("k,o", [(16, 3), (16, 2), (5, 3), (7, 4), (20, 2)])
def test_canonical_split_covers_the_gram(k, o):
    rng = np.random.default_rng(k * 10 + o)
    n = 257
    u = rng.normal(size=(n, k))
    cols, rperm, wt = _plan(k, o)
    pc = len(cols)
    a = np.stack([np.prod(u[:, list(c)], axis=1) for c in cols], axis=1)  # the materialised products (n x pc)
    gram = a.T @ a
    rowpos = {cols[i]: r for r, i in enumerate(rperm)}
    nat = {c: i for i, c in enumerate(cols)}
    # what the kernel computes: entries (row position, natural column) of the tiles in the masks
    computed = np.zeros((pc, pc), dtype=bool)
    mcan = np.full((pc, pc), np.nan)
    ar = a[:, rperm]
    for r0, c0, mask in wt:
        for i in range(4):
            for j in range(4):
                if mask >> (4 * i + j) & 1:
                    rs, cs = slice(r0 + 16 * i, min(pc, r0 + 16 * i + 16)), slice(c0 + 16 * j, min(pc, c0 + 16 * j + 16))
                    mcan[rs, cs] = ar[:, rs].T @ a[:, cs]
                    computed[rs, cs] = True
    # expansion (pld_moment_expand_kernel's index table): merged multiset -> (o smallest | o largest)
    out = np.empty((pc, pc))
    for i, ci in enumerate(cols):
        for j, cj in enumerate(cols):
            z = tuple(sorted(ci + cj))
            r, c = rowpos[z[:o]], nat[z[o:]]
            assert computed[r, c], (ci, cj)
            out[i, j] = mcan[r, c]
    assert np.allclose(out, gram, rtol=1e-12, atol=1e-12 * np.abs(gram).max())
    assert np.array_equal(out, out.T)  # symmetric by construction, not merely to rounding


def test_tile_count_matches_bench():
    spec = importlib.util.spec_from_file_location("bench_for_tiles", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for k, o in [(16, 3), (16, 2)]:
        _, _, wt = _plan(k, o)
        assert sum(bin(m).count("1") for _, _, m in wt) == bench._moment_tiles(k, o)
    # the numbers DESIGN.md quotes for configs[4]
    assert bench._moment_tiles(16, 3) == 278 and bench._moment_tiles(16, 2) == 28
