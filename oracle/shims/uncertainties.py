"""TEST INFRASTRUCTURE shim: `uncertainties` is not installed in the oracle interpreter; lightkurve.seismology builds a few
module-level constants with it at import (stellar_estimators.py) although nothing on the hot path uses them.  Inert
placeholders: a value/std_dev holder and a umath namespace that refuses to compute."""


class _UFloat(object):
    def __init__(self, nominal, std_dev=0.0):
        self.nominal_value, self.n = nominal, nominal
        self.std_dev, self.s = std_dev, std_dev


def ufloat(nominal, std_dev=0.0):
    return _UFloat(nominal, std_dev)


class _UMath(object):                              # pragma: no cover
    def __getattr__(self, name):
        raise NotImplementedError("uncertainties shim: not available in the oracle environment")


umath = _UMath()
