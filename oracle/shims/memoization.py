"""Shim: lightkurve only uses memoization.cached to cache neighbour downloads (correctors/metrics.py:13,279)."""


def cached(*a, **k):
    return a[0] if a and callable(a[0]) else (lambda f: f)
