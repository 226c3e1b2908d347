"""Dense design-matrix containers (reference: src/lightkurve/correctors/designmatrix.py:28-130, 284-304, 387-444).

pandas-free: the values are a float64 (cadences x regressors) ndarray; column names, name, prior_mu and
prior_sigma carry the same meaning as in the reference (prior_sigma defaults to +inf = no prior)."""
import copy as _copy

import numpy as np

__all__ = ["DesignMatrix", "DesignMatrixCollection", "SparseDesignMatrix", "SparseDesignMatrixCollection",
           "create_sparse_spline_matrix"]


class DesignMatrix(object):
    def __init__(self, df, columns=None, name="unnamed_matrix", prior_mu=None, prior_sigma=None):
        if isinstance(df, dict):
            columns = list(df.keys()) if columns is None else columns
            values = np.column_stack([np.asarray(v, dtype=np.float64) for v in df.values()])
        elif hasattr(df, "toarray"):     # scipy.sparse matrix: densified (K <= ~500 columns on this path)
            values = np.asarray(df.toarray(), dtype=np.float64)
        else:
            values = np.asarray(getattr(df, "values", df), dtype=np.float64)
            if values.ndim == 1:
                values = values[:, None]
            if columns is None and hasattr(df, "columns"):
                columns = list(df.columns)
        if values.ndim != 2:
            raise ValueError("a design matrix must be 2-dimensional (cadences x regressors)")
        self.values = values
        self.columns = list(columns) if columns is not None else list(range(values.shape[1]))
        if len(self.columns) != values.shape[1]:
            raise ValueError("`columns` must name every column")
        self.name = name
        k = values.shape[1]
        self.prior_mu = np.atleast_1d(np.zeros(k) if prior_mu is None else np.asarray(prior_mu, dtype=np.float64))
        self.prior_sigma = np.atleast_1d(
            np.ones(k) * np.inf if prior_sigma is None else np.asarray(prior_sigma, dtype=np.float64))

    @property
    def X(self):
        return self.values

    @property
    def shape(self):
        return self.values.shape

    def copy(self):
        return _copy.deepcopy(self)

    def append_constant(self, prior_mu=0, prior_sigma=np.inf, inplace=False):
        """Column of ones named "offset" appended (reference :284-304)."""
        dm = self if inplace else self.copy()
        dm.values = np.hstack([self.values, np.ones((self.shape[0], 1))])
        dm.columns = list(self.columns) + ["offset"]
        dm.prior_mu = np.append(self.prior_mu, prior_mu)
        dm.prior_sigma = np.append(self.prior_sigma, prior_sigma)
        return dm

    def validate(self, rank=False):
        if self.values.shape[1] != len(self.prior_mu) or self.values.shape[1] != len(self.prior_sigma):
            raise ValueError("prior_mu and prior_sigma must have one entry per column")

    def __repr__(self):
        return "{} DesignMatrix {}".format(self.name, self.shape)


class DesignMatrixCollection(object):
    def __init__(self, matrices):
        self.matrices = list(matrices)
        self.X = np.hstack(tuple(m.X for m in self.matrices))
        self.validate()

    @property
    def values(self):
        return np.hstack(tuple(m.values for m in self.matrices))

    @property
    def prior_mu(self):
        return np.hstack([m.prior_mu for m in self.matrices])

    @property
    def prior_sigma(self):
        return np.hstack([m.prior_sigma for m in self.matrices])

    @property
    def shape(self):
        return self.X.shape

    def __iter__(self):
        return iter(self.matrices)

    def __getitem__(self, key):
        if isinstance(key, str):
            for m in self.matrices:
                if m.name == key:
                    return m
            raise KeyError(key)
        return self.matrices[key]

    def to_designmatrix(self, name=None):
        name = self.matrices[0].name if name is None else name
        return DesignMatrix(self.X, name=name, prior_mu=self.prior_mu, prior_sigma=self.prior_sigma)

    def validate(self):
        n = {m.shape[0] for m in self.matrices}
        if len(n) != 1:
            raise ValueError("all design matrices must have the same number of cadences")
        for m in self.matrices:
            m.validate()

    def __repr__(self):
        return "DesignMatrixCollection:\n" + "".join("\t{}\n".format(m.__repr__()) for m in self.matrices)


class SparseDesignMatrix(DesignMatrix):
    """Reference: src/lightkurve/correctors/sparsedesignmatrix.py.  The HIP regression path is dense (the Gram matrix is
    formed on the fp64 matrix cores; only the spline block of a PLD matrix is genuinely sparse, degree + 1 non-zeros per
    row), so a sparse matrix is accepted and densified at construction; names, priors and collection semantics are the
    reference's."""

    def __repr__(self):
        return "{} SparseDesignMatrix {}".format(self.name, self.shape)


class SparseDesignMatrixCollection(DesignMatrixCollection):
    def __repr__(self):
        return "SparseDesignMatrixCollection:\n" + "".join("\t{}\n".format(m.__repr__()) for m in self.matrices)


def _spline_basis_vector(x, degree, i, knots):
    """Cox-de Boor recursion exactly as the reference writes it (designmatrix.py:853-893): degree-0 pieces are CLOSED on
    both ends (a sample that sits on a knot belongs to both neighbouring pieces), zero-width denominators give 0."""
    if degree == 0:
        B = np.zeros(len(x))
        B[(x >= knots[i]) & (x <= knots[i + 1])] = 1
        return B
    da = knots[degree + i] - knots[i]
    db = knots[i + degree + 1] - knots[i + 1]
    alpha1 = (x - knots[i]) / da if da != 0 else np.zeros(len(x))
    alpha2 = (knots[i + degree + 1] - x) / db if db != 0 else np.zeros(len(x))
    return _spline_basis_vector(x, degree - 1, i, knots) * alpha1 + _spline_basis_vector(x, degree - 1, i + 1, knots) * alpha2


def create_sparse_spline_matrix(x, n_knots=20, knots=None, degree=3, name="spline"):
    """The spline block of ``PLDCorrector.create_design_matrix(sparse=True)`` (reference designmatrix.py:896-949): knots at
    the mid-points between the samples that end each of ``n_knots - degree`` equal chunks of the sorted ``x``, boundary
    knots repeated ``degree`` times, all-zero basis vectors dropped.  A different basis from the patsy splines of the dense
    path (designmatrix.py:952-997), so ``sparse=True`` and ``sparse=False`` differ in the reference too.  Host-side
    construction of an N x ~n_knots block, like the knot selection of the dense path."""
    x = np.asarray(x, np.float64)
    if not isinstance(n_knots, (int, np.integer)):
        raise ValueError("`n_knots` must be an integer.")
    if n_knots - degree <= 0:
        raise ValueError("n_knots must be greater than degree.")
    if knots is None:
        ends = np.asarray([s_[-1] for s_ in np.array_split(np.argsort(x), n_knots - degree)[:-1]])
        knots = [np.mean([x[k], x[k + 1]]) for k in ends]
    knots = np.append(np.append(x.min(), knots), x.max())
    knots = np.unique(knots)
    knots_wbounds = np.append(np.append([x.min()] * (degree - 1), knots), [x.max()] * degree)
    cols = [_spline_basis_vector(x, degree, idx, knots_wbounds) for idx in np.arange(-1, len(knots_wbounds) - degree - 1)]
    cols = [c for c in cols if c.sum() != 0]
    return SparseDesignMatrix(np.column_stack(cols), name=name)
