"""Oracle-environment shim (TEST INFRASTRUCTURE ONLY, never imported by the product).

Lets the reference (lightkurve, /root/reference/src) import under the only astropy on this
machine: /opt/conda/bin/python3.9 (astropy 4.3.1, scipy 1.7.1) with a newer numpy (1.26.4).
Restores numpy aliases astropy 4.3.1 still references and patches Quantity.any/all, which
numpy>=1.22's nanmedian calls on Quantity-typed boolean arrays.  See SURVEY.md Appendix A.
"""
import numpy as np

for _name, _fn in dict(asscalar=lambda a: a.item(), alen=len, msort=lambda a: np.sort(a, axis=0),
                       sometrue=np.any, alltrue=np.all, product=np.prod, cumproduct=np.cumprod,
                       round_=np.round, float=float, int=int, bool=bool, object=object,
                       complex=complex, str=str).items():
    if not hasattr(np, _name):
        setattr(np, _name, _fn)


def _patch_quantity():
    from astropy.units import Quantity
    Quantity.any = lambda self, axis=None, out=None, **kw: self.value.any(axis=axis, out=out, **kw)
    Quantity.all = lambda self, axis=None, out=None, **kw: self.value.all(axis=axis, out=out, **kw)


import builtins as _b

_orig_import, _done = _b.__import__, [False]


def _imp(name, *a, **k):
    m = _orig_import(name, *a, **k)
    if not _done[0] and name.startswith("lightkurve"):
        _done[0] = True
        _patch_quantity()
    return m


_b.__import__ = _imp
