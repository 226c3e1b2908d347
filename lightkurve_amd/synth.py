"""Synthetic light curves / grids for the parity tests and bench.py.

Pure numpy, importable from either interpreter (py3.10 product env, conda py3.9 oracle env).
Every target has its own RNG: ``default_rng(20260925 + 1000003*config + target_index)``.
Nothing here is an oracle and nothing here touches the GPU: it only manufactures inputs.
"""
import numpy as np

SEED0 = 20260925


def _rng(config, index):
    return np.random.default_rng(SEED0 + 1000003 * int(config) + int(index))


def tess_like_times(rng, n, cadence_days=2.0 / 1440.0):
    """Irregular "TESS-like" sampling: regular base grid, 1-d mid-span gap, 3 % dropouts, 1e-5 d jitter."""
    nbase = int(np.ceil(1.1 * n))
    while True:
        t = np.arange(nbase) * cadence_days
        mid = 0.5 * t[-1]
        keep = np.abs(t - mid) > 0.5
        keep &= rng.random(nbase) >= 0.03
        t = t[keep]
        if len(t) >= n:
            break
        nbase = int(nbase * 1.1) + 8  # tiny n: the 1-d gap can eat everything, widen the base grid
    t = t[:n] + rng.normal(0.0, 1e-5, n)
    t.sort()
    return t


def ls_target(config, index, n, cadence_days=2.0 / 1440.0):
    """(time, flux, flux_err, truth) for one Lomb-Scargle target (sinusoid + white noise)."""
    rng = _rng(config, index)
    t = tess_like_times(rng, n, cadence_days)
    amp = rng.uniform(2e-4, 2e-3)
    period = np.exp(rng.uniform(np.log(0.2), np.log(12.0)))
    phi = rng.uniform(0, 2 * np.pi)
    sigma = 5e-4
    flux = 1.0 + amp * np.sin(2 * np.pi * t / period + phi) + rng.normal(0, sigma, n)
    return t, flux, np.full(n, sigma), dict(amp=amp, period=period, phi=phi)


def ls_frequency_grid(m, fmax=360.0):
    """Regular grid f_j = (j+1)*df, df = fmax/m [1/d] (top = 2-min-cadence Nyquist)."""
    df = fmax / m
    return (np.arange(m) + 1.0) * df


def bls_target(config, index, n, cadence_days=2.0 / 1440.0):
    """(time, flux, flux_err, truth) for one BLS target (box transits + white noise)."""
    rng = _rng(config, index)
    t = tess_like_times(rng, n, cadence_days)
    depth = rng.uniform(5e-4, 5e-3)
    period = rng.uniform(1.0, 12.0)
    duration = rng.uniform(0.05, 0.3)
    t0 = rng.uniform(0, period)
    sigma = 5e-4
    flux = 1.0 + rng.normal(0, sigma, n)
    in_transit = np.abs((t - t0 + 0.5 * period) % period - 0.5 * period) < 0.5 * duration
    flux[in_transit] -= depth
    return t, flux, np.full(n, sigma), dict(depth=depth, period=period, duration=duration, t0=t0)


def bls_grid(n_periods=50000, n_durations=200, pmin=0.6, pmax=13.0, dmin=0.02, dmax=0.5):
    """BASELINE config C4 grids: periods uniform in frequency, ascending; durations linear."""
    period = 1.0 / np.linspace(1.0 / pmax, 1.0 / pmin, n_periods)[::-1]
    duration = np.linspace(dmin, dmax, n_durations)
    return np.ascontiguousarray(period), duration


def pack_ragged(arrays):
    """List of 1-D arrays -> (concatenated float64 array, int64 prefix offsets[B+1])."""
    off = np.zeros(len(arrays) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(a) for a in arrays])
    flat = np.concatenate([np.asarray(a, dtype=np.float64) for a in arrays]) if arrays else np.zeros(0)
    return np.ascontiguousarray(flat), off


def ls_batch(config, b, n, first_index=0, cadence_days=2.0 / 1440.0):
    """B LS targets of n cadences each, packed ragged: (t, y, dy, n_off)."""
    ts, ys, es = [], [], []
    for i in range(b):
        t, y, e, _ = ls_target(config, first_index + i, n, cadence_days)
        ts.append(t), ys.append(y), es.append(e)
    t, off = pack_ragged(ts)
    return t, pack_ragged(ys)[0], pack_ragged(es)[0], off


def bls_batch(config, b, n, first_index=0, cadence_days=2.0 / 1440.0):
    ts, ys, es = [], [], []
    for i in range(b):
        t, y, e, _ = bls_target(config, first_index + i, n, cadence_days)
        ts.append(t), ys.append(y), es.append(e)
    t, off = pack_ragged(ts)
    return t, pack_ragged(ys)[0], pack_ragged(es)[0], off


def pld_cutout(config, index, n=3500, npix=11, cadence_days=0.0204):
    """Synthetic K2-like TPF cutout: (time[n], flux[n,npix,npix] float32, flux_err, truth).

    Gaussian PSF (sigma 1.2 px, 2e4 e-/s peak), background 50, read noise 5, roll-like centroid
    jitter 0.3 sin(2 pi t/0.25) + 0.05 N(0,1) px in x and 0.2 cos(...) in y.
    """
    rng = _rng(config, index)
    t = np.arange(n) * cadence_days
    cx = 0.5 * (npix - 1) + 0.3 * np.sin(2 * np.pi * t / 0.25) + 0.05 * rng.normal(0, 1, n)
    cy = 0.5 * (npix - 1) + 0.2 * np.cos(2 * np.pi * t / 0.25)
    yy, xx = np.mgrid[0:npix, 0:npix]
    amp = rng.uniform(5e-4, 2e-3)
    period = rng.uniform(1.0, 8.0)
    star = 1.0 + amp * np.sin(2 * np.pi * t / period)
    psf = np.exp(-0.5 * ((xx[None] - cx[:, None, None]) ** 2 + (yy[None] - cy[:, None, None]) ** 2) / 1.2 ** 2)
    flux = 2e4 * star[:, None, None] * psf + 50.0
    err = np.sqrt(np.abs(flux) + 25.0)
    flux = flux + rng.normal(0, 1, flux.shape) * err
    return t, flux.astype(np.float32), err.astype(np.float32), dict(amp=amp, period=period)
