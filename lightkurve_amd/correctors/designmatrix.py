"""Dense design-matrix containers (reference: src/lightkurve/correctors/designmatrix.py:28-130, 167-349, 387-444) and the
spline builders (:896-997).

pandas-free: the values are a float64 (cadences x regressors) ndarray; column names, name, prior_mu and
prior_sigma carry the same meaning as in the reference (prior_sigma defaults to +inf = no prior).  ``pca``, ``standardize``
and ``create_spline_matrix`` run on the GPU (lk_pca_batch, lk_standardize_batch, lk_spline_basis_batch)."""
import copy as _copy
import warnings

import numpy as np

__all__ = ["DesignMatrix", "DesignMatrixCollection", "SparseDesignMatrix", "SparseDesignMatrixCollection",
           "create_sparse_spline_matrix", "create_spline_matrix", "LightkurveWarning"]


class LightkurveWarning(Warning):
    """Mirror of lightkurve.utils.LightkurveWarning (the class the reference's validate() warns with)."""


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

    def split(self, row_indices, inplace=False):
        """Every regressor split over len(row_indices) + 1 columns that are non-zero on consecutive row ranges only
        (reference :167-213): shape (n_rows, (len(row_indices) + 1) * n_columns), priors repeated per range, column
        "name" of range i renamed "name i+1"."""
        if isinstance(row_indices, (int, np.integer)):
            row_indices = [int(row_indices)]
        if row_indices is None or len(row_indices) == 0 or list(row_indices) == [0]:
            return self
        lower = np.append(0, row_indices).astype(int)
        upper = np.append(row_indices, self.shape[0]).astype(int)
        n, k = self.shape
        vals = np.zeros((n, k * len(lower)))
        cols = []
        for idx, (a, b) in enumerate(zip(lower, upper)):
            vals[a:b, idx * k:(idx + 1) * k] = self.values[a:b]
            cols += ["{} {}".format(c, idx + 1) for c in self.columns]
        dm = self if inplace else self.copy()
        dm.values = vals
        dm.columns = cols
        dm.prior_mu = np.hstack([self.prior_mu for _ in lower])
        dm.prior_sigma = np.hstack([self.prior_sigma for _ in lower])
        return dm

    def standardize(self, inplace=False, device=0):
        """Columns median-subtracted and sigma-divided, zeros treated as missing, constant columns unchanged (reference
        :215-250) — on the GPU (lk_standardize_batch: one radix-select median per column)."""
        from .. import _capi
        dm = self if inplace else self.copy()
        dm.values = _capi.standardize_batch(self.values, device=device)
        return dm

    def pca(self, nterms=6, n_iter=10, device=0):
        """A new DesignMatrix whose ``nterms`` columns are the leading left singular vectors of the column-centred matrix
        (reference :252-282, which calls the randomised ``fbpca.pca(values, nterms, n_iter)``; ``n_iter`` is accepted and
        ignored: the GPU path iterates its subspace to a 1e-10 residual, i.e. to the exact singular subspace).
        The kernel range is lk_pca_batch's (include/lkhip.h): ``nterms`` <= 48, <= 4096 columns, >= 2 rows — a ValueError
        beyond it (there is no CPU fallback in this class; the lightkurve seam keeps the original method for such calls)."""
        from .. import _capi
        if nterms > self.shape[1]:
            nterms = self.shape[1]
        return DesignMatrix(_capi.pca_batch(self.values, nterms, device=device), name=self.name)

    @property
    def rank(self):
        return np.linalg.matrix_rank(self.values)

    def _validate(self, rank=True):
        """Reference :306-337: LightkurveWarning for a matrix whose rank is below half its column count, ValueError for
        priors of the wrong length and for prior widths <= 0 (which would otherwise reach the kernel's 1 / sigma^2)."""
        if rank and self.shape[1] > 0:
            r = self.rank
            if r < 0.5 * self.shape[1]:
                warnings.warn("The design matrix has low rank ({}) compared to the number of columns ({}), which suggests "
                              "that the matrix contains duplicate or correlated columns. This may prevent the regression "
                              "from succeeding. Consider reducing the dimensionality by calling the `pca()` method."
                              "".format(r, self.shape[1]), LightkurveWarning)
        if self.prior_mu is not None and len(self.prior_mu) != self.shape[1]:
            raise ValueError("`prior_mu` must have shape {}".format(self.shape[1]))
        if self.prior_sigma is not None:
            if len(self.prior_sigma) != self.shape[1]:
                raise ValueError("`prior_sigma` must have shape {}".format(self.shape[1]))
            if np.any(np.asarray(self.prior_sigma) <= 0):
                raise ValueError("`prior_sigma` values cannot be smaller than or equal to zero")

    def validate(self, rank=True):
        self._validate(rank=rank)

    def __repr__(self):
        return "{} DesignMatrix {}".format(self.name, self.shape)


class DesignMatrixCollection(object):
    def __init__(self, matrices, validate_rank=True):
        """``validate_rank=False`` skips the reference constructor's per-matrix rank check (an SVD on the host, ~25 ms for
        4500 x 137): the batch entry points build thousands of collections and check shapes / priors only."""
        self.matrices = list(matrices)
        self.X = np.hstack(tuple(m.X for m in self.matrices))
        self.validate(rank=validate_rank)

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

    def validate(self, rank=True):
        n = {m.shape[0] for m in self.matrices}
        if len(n) != 1:
            raise ValueError("all design matrices must have the same number of cadences")
        for m in self.matrices:
            m.validate() if rank else m.validate(rank=False)

    def __repr__(self):
        return "DesignMatrixCollection:\n" + "".join("\t{}\n".format(m.__repr__()) for m in self.matrices)


class SparseDesignMatrix(DesignMatrix):
    """Reference: src/lightkurve/correctors/sparsedesignmatrix.py.  The HIP regression path is dense (the Gram matrix is
    formed on the fp64 matrix cores; only the spline block of a PLD matrix is genuinely sparse, degree + 1 non-zeros per
    row), so a sparse matrix is accepted and densified at construction; names, priors and collection semantics are the
    reference's."""

    def validate(self, rank=False):
        """Rank checks are off by default for sparse matrices (reference designmatrix.py:339-349)."""
        self._validate(rank=rank)

    def __repr__(self):
        return "{} SparseDesignMatrix {}".format(self.name, self.shape)


class SparseDesignMatrixCollection(DesignMatrixCollection):
    def __repr__(self):
        return "SparseDesignMatrixCollection:\n" + "".join("\t{}\n".format(m.__repr__()) for m in self.matrices)


def create_sparse_spline_matrix(x, n_knots=20, knots=None, degree=3, name="spline", device=0):
    """The spline block of ``PLDCorrector.create_design_matrix(sparse=True)`` (reference designmatrix.py:896-949): knots at
    the mid-points between the samples that end each of ``n_knots - degree`` equal chunks of the sorted ``x``, all-zero
    basis vectors dropped.  The reference evaluates the basis with a Python Cox-de Boor recursion (:853-893) whose result is
    the clamped B-spline basis on [min x, knots, max x] — the basis of the dense builder below with other knots (verified
    against the reference: 3e-16) — so it is evaluated by the same kernel (lk_spline_basis_batch).  ``sparse=True`` and
    ``sparse=False`` differ in the reference too: different interior knots."""
    from .. import _capi
    x = np.asarray(x, np.float64)
    if not isinstance(n_knots, (int, np.integer)):
        raise ValueError("`n_knots` must be an integer.")
    if n_knots - degree <= 0:
        raise ValueError("n_knots must be greater than degree.")
    if knots is None:
        ends = np.asarray([s_[-1] for s_ in np.array_split(np.argsort(x), n_knots - degree)[:-1]])
        knots = [np.mean([x[k], x[k + 1]]) for k in ends]
    knots = np.unique(np.append(np.append(x.min(), knots), x.max()))
    basis = _capi.spline_basis_batch(x, knots, degree=int(degree), device=device)
    return SparseDesignMatrix(basis[:, basis.sum(axis=0) != 0], name=name)


def create_spline_matrix(x, n_knots=20, knots=None, degree=3, name="spline", include_intercept=True, device=0):
    """DesignMatrix of B-splines as ``patsy.dmatrix("bs(x, df|knots, degree, include_intercept) - 1")`` builds it (reference
    designmatrix.py:952-997): with ``knots`` the given interior knots, otherwise ``n_knots`` = df columns whose interior
    knots are equally spaced quantiles of x (numpy's linear-interpolation percentiles, as patsy takes them); the boundary
    knots are min(x) and max(x).  The basis itself is evaluated on the GPU (lk_spline_basis_batch, de Boor)."""
    from .. import _capi
    x = np.asarray(x, dtype=np.float64)
    order = int(degree) + 1
    if knots is not None:
        inner = np.sort(np.asarray(knots, dtype=np.float64))
    else:
        n_inner = int(n_knots) - order + (0 if include_intercept else 1)
        if n_inner < 0:
            raise ValueError("df={} is too small for degree={}; must be >= {}".format(
                n_knots, degree, order - (0 if include_intercept else 1)))
        inner = np.percentile(x, np.linspace(0, 100, n_inner + 2)[1:-1]) if n_inner > 0 else np.zeros(0)
    full = np.concatenate([[np.min(x)], inner, [np.max(x)]])
    basis = _capi.spline_basis_batch(x, full, degree=degree, device=device)
    if not include_intercept:
        basis = basis[:, 1:]
    return DesignMatrix(basis, columns=["knot{}".format(i + 1) for i in range(basis.shape[1])], name=name)
