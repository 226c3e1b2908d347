#!/usr/bin/env python
"""bench.py — headline benchmark: batched Lomb-Scargle on MI355X (BASELINE.json configs[1]).

    python bench.py [--gpus N --steps K --warmup W]        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = one pass of the hot path over one batch: B targets x M trial frequencies through liblkhip.so's
lk_ls_fast_batch_dev (headline: the reference's DEFAULT method ls_method="fast", extirpolation + FFT) and, with the
default --ls-method both, lk_ls_power_batch_dev (the exact direct sums, reported as `other_method`), each followed
by lk_argmax_batch_dev; inputs already resident in HBM.  Prints ONE JSON line on rank 0 (metric
frequencies*targets/sec, whole job over all ranks) carrying `roofline` (fast: algorithmic HBM bytes over the HIP-event
time against 8 TB/s, with the PMC traffic from profiles/traffic.json; exact: 16 flop per (cadence, frequency) pair
against the fp64 vector peak) and `cpu_baseline` (the reference's default algorithm run through astropy itself on this
box's usable cores, on a bounded sample; numpy port if astropy is absent).

Weak scaling: every rank owns B targets (targets are independent; no data-path collective is needed to compute).
With N>1 the per-target (max power, argmax) are all-gathered over RCCL each step (--gather summary, default), or the
full spectra chunk by chunk so the collective of chunk k overlaps the kernels of chunk k+1 (--gather spectra).

Other workloads, same protocol and JSON shape: --workload bls | pld | regress | flatten | fold | lschi2 | pgsmooth.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X fp64 vector == fp64 MFMA dense peak (256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--targets", type=int, default=1000, help="targets per GPU (B)")
    ap.add_argument("--cadences", type=int, default=20000, help="cadences per target (N)")
    ap.add_argument("--freqs", type=int, default=100000, help="trial frequencies (M)")
    ap.add_argument("--chunks", type=int, default=4, help="target chunks per step (comm/compute overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="ls", choices=["ls", "bls", "pld", "flatten", "lschi2", "pgsmooth", "fold", "regress"])
    ap.add_argument("--regressors", type=int, default=135, help="regress: design-matrix columns K")
    ap.add_argument("--nterms", type=int, default=2, help="lschi2: Fourier terms")
    ap.add_argument("--cutouts", type=int, default=500, help="PLD: cutouts per GPU")
    ap.add_argument("--pld-cadences", type=int, default=3500)
    ap.add_argument("--ls-method", default="both", choices=["both", "exact", "fast"],
                    help="fast: the reference's default method (extirpolation + FFT), the headline; exact: direct fp64 trig "
                         "sums; both (default): headline = fast, the exact kernel is measured too and reported beside it")
    ap.add_argument("--gather", default="summary", choices=["summary", "spectra", "none"],
                    help="N>1: what is all-gathered over RCCL each step: per-target (max power, argmax), the full spectra, or nothing")
    ap.add_argument("--periods", type=int, default=50000)
    ap.add_argument("--durations", type=int, default=200)
    return ap.parse_args()


def effective_cores(cap=64):
    """Host cores this process may actually use: min(affinity mask, cgroup CPU quota) — the GPU box advertises
    256 logical CPUs but its cgroup grants 16 (cpu.max = "1600000 100000")."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, min(n, cap))


CONDA = "/opt/conda/bin/python3.9"
SYS_STDCXX = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"


def _astropy_baseline(argv):
    """Run oracle/astropy_baseline.py under the conda interpreter that ships astropy (the reference's real
    numerical dependency).  Returns its JSON dict, or None if that interpreter / astropy is not there."""
    import subprocess
    if not os.path.exists(CONDA):
        return None
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "oracle", "shims") + os.pathsep + ROOT)
    if os.path.exists(SYS_STDCXX):
        env["LD_PRELOAD"] = SYS_STDCXX
    try:
        p = subprocess.run([CONDA, "-W", "ignore", os.path.join(ROOT, "oracle", "astropy_baseline.py")] +
                           [str(a) for a in argv], env=env, capture_output=True, timeout=900)
        for line in p.stdout.decode().splitlines():
            if line.startswith("BASELINE "):
                return json.loads(line[9:])
    except Exception:
        pass
    return None


def cpu_baseline_ls(args):
    """The reference's DEFAULT CPU path on this box's host cores, on a bounded sample of the same workload,
    before torch/HIP is initialised.  Preferred: astropy itself (LombScargle.power(method='fast') exactly as
    lightkurve calls it, periodogram.py:961-964) under the conda interpreter, one process per core (kind
    "reference").  Fallback: the numpy port of the same algorithm from oracle/ (kind "port")."""
    import multiprocessing as mp
    from oracle import cpu_baseline as cb
    cores = effective_cores()
    exact_pairs_per_s = cb.ls_exact_rate(args.cadences)
    extra = {"exact_port_1core": {"value": exact_pairs_per_s / args.cadences, "unit": "frequencies*targets/sec",
                                  "note": "C oracle (the exact direct-sum arithmetic the GPU kernel performs), 1 core"}}
    n_targets = cores * 16
    r = _astropy_baseline(["ls", n_targets, args.cadences, args.freqs, cores, "fast"])
    if r is not None:
        out = {"value": r["units_per_s"], "unit": "frequencies*targets/sec", "cores": cores, "kind": "reference",
               "sample": "astropy %s LombScargle.power(method='fast') as lightkurve calls it (periodogram.py:961-964), "
                         "%d targets x %d freqs, N=%d, %d processes, %.1f s" % (r["astropy"], n_targets, args.freqs,
                                                                              args.cadences, cores, r["seconds"])}
        out.update(extra)
        return out
    n_targets = cores * 2
    jobs = [(1, i, args.cadences, args.freqs) for i in range(n_targets)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        pool.map(cb.ls_fast_one, jobs[:cores])          # warm (imports, FFT plans)
        t0 = time.perf_counter()
        pool.map(cb.ls_fast_one, jobs)
        dt = time.perf_counter() - t0
    out = {"value": n_targets * args.freqs / dt, "unit": "frequencies*targets/sec", "cores": cores, "kind": "port",
           "sample": "%d targets x %d freqs, N=%d, numpy port of the reference default ls_method='fast' "
                     "(Press-Rybicki FFT), %d processes" % (n_targets, args.freqs, args.cadences, cores)}
    out.update(extra)
    return out


def cpu_baseline_bls(args):
    from oracle import cpu_baseline as cb
    cores = effective_cores()
    r = _astropy_baseline(["bls", cores * 4, args.cadences, 480, args.durations, cores])
    if r is not None:
        return {"value": r["units_per_s"], "unit": "periods*targets/sec", "cores": cores, "kind": "reference",
                "sample": "astropy %s BoxLeastSquares.power (compiled run_bls) as lightkurve calls it "
                          "(periodogram.py:1161-1169), %d targets x 480 periods x %d durations, N=%d, %d processes, "
                          "%.1f s" % (r["astropy"], cores * 4, args.durations, args.cadences, cores, r["seconds"])}
    rate = cb.bls_rate(args.cadences, args.durations)
    return {"value": rate, "unit": "periods*targets/sec", "cores": 1, "kind": "port",
            "sample": "C oracle (restated astropy run_bls), 1 target x 24 periods x %d durations, N=%d, 1 core"
                      % (args.durations, args.cadences)}


def cpu_baseline_pld(args):
    """numpy/LAPACK port of PLDCorrector.correct (oracle.np_oracle.pld_correct: exact SVD in place of fbpca), 1 core,
    2 cutouts of the bench shape."""
    from lightkurve_amd import synth
    from oracle import np_oracle as O
    allm = np.ones((11, 11), bool)
    t0 = time.perf_counter()
    n = 2
    for i in range(n):
        t, flux, err, _ = synth.pld_cutout(4, i, n=args.pld_cadences, npix=11)
        O.pld_correct(t, flux, err, allm, allm, allm, pld_order=3, pca_components=16, spline_degree=5)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "cutouts/sec", "cores": 1, "kind": "port",
            "sample": "%d cutouts 11x11 x %d cadences, order 3, 16 PCA comps, numpy port (LAPACK may thread)"
                      % (n, args.pld_cadences)}


def cpu_baseline_flatten(args):
    from lightkurve_amd import synth
    from oracle import np_oracle as O
    n = 8
    lcs = [synth.ls_target(6, i, args.cadences) for i in range(n)]
    t0 = time.perf_counter()
    for t, y, e, _ in lcs:
        O.flatten_trend(t, y, 401, 2, 5, 3, 3)
    dt = time.perf_counter() - t0
    return {"value": n * args.cadences / dt, "unit": "cadences/sec", "cores": 1, "kind": "port",
            "sample": "%d light curves x %d cadences, window 401, numpy port of LightCurve.flatten" % (n, args.cadences)}


def cpu_baseline_lschi2(args):
    from lightkurve_amd import synth
    from oracle import np_oracle as O
    t, y, e, _ = synth.ls_target(1, 0, 2000)
    f = 0.0036 * (1 + np.arange(400))
    t0 = time.perf_counter()
    O.ls_power_chi2(t - t[0], y, None, f, nterms=args.nterms, normalization="lk_amplitude")
    dt = time.perf_counter() - t0
    # cost is linear in cadences x frequencies: scale the sample's pair rate to a frequencies*targets rate at N cadences
    return {"value": len(f) * 2000.0 / args.cadences / dt, "unit": "frequencies*targets/sec", "cores": 1, "kind": "port",
            "sample": "numpy restatement of astropy lombscargle_chi2 (nterms=%d), 2000 cadences x 400 frequencies, "
                      "rate rescaled to %d cadences per target" % (args.nterms, args.cadences)}


def cpu_baseline_pgsmooth(args):
    from oracle import np_oracle as O
    rng = np.random.default_rng(0)
    M = args.freqs
    f = 0.0036 * (1 + np.arange(M))
    p = rng.chisquare(2, M)
    t0 = time.perf_counter()
    O.pg_smooth_logmedian(f, p, 0.01)
    dt = time.perf_counter() - t0
    return {"value": M / dt, "unit": "frequencies*targets/sec", "cores": 1, "kind": "port",
            "sample": "1 periodogram x %d frequencies, logmedian filter_width 0.01, numpy restatement of the reference loop" % M}


def cpu_baseline_regress(args):
    from oracle import np_oracle as O
    rng = np.random.default_rng(0)
    n, K, nb = args.pld_cadences, args.regressors, 4
    t0 = None
    Xs = [rng.standard_normal((n, K)) for _ in range(nb)]
    ys = [X @ rng.standard_normal(K) * 1e-3 + 1 + 1e-3 * rng.standard_normal(n) for X in Xs]
    t0 = time.perf_counter()
    for X, y in zip(Xs, ys):
        O.regression_correct(X, y, np.full(n, 1e-3), prior_mu=np.zeros(K), prior_sigma=np.full(K, 10.0))
    dt = time.perf_counter() - t0
    return {"value": nb / dt, "unit": "fits/sec", "cores": effective_cores(), "kind": "port",
            "sample": "%d fits, N=%d, K=%d, 5 sigma-clip iterations, numpy port of RegressionCorrector.correct "
                      "(BLAS/LAPACK may thread)" % (nb, n, K)}


def cpu_baseline_fold(args):
    from lightkurve_amd import synth
    from oracle import np_oracle as O
    lcs = [synth.ls_target(6, i, args.cadences) for i in range(32)]
    t0 = time.perf_counter()
    for t, y, e, _ in lcs:
        ph, order, _c = O.fold(t, 3.3, t[0])
        y[order], e[order]
    dt = time.perf_counter() - t0
    return {"value": 32 * args.cadences / dt, "unit": "cadences/sec", "cores": 1, "kind": "port",
            "sample": "32 light curves x %d cadences, numpy mod + stable argsort + two gathers" % args.cadences}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    cpu_base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_base = {"ls": cpu_baseline_ls, "bls": cpu_baseline_bls, "pld": cpu_baseline_pld,
                    "flatten": cpu_baseline_flatten, "lschi2": cpu_baseline_lschi2, "pgsmooth": cpu_baseline_pgsmooth,
                    "fold": cpu_baseline_fold, "regress": cpu_baseline_regress}[args.workload](args)

    import torch
    import torch.distributed as dist
    from lightkurve_amd import _capi, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # LK_BENCH_FORCE_DIST=1 runs the N>1 code paths (RCCL init, gathers, barrier, max over ranks) with a single rank —
    # the only way to exercise them on a 1-GPU box
    dist_on = world > 1 or os.environ.get("LK_BENCH_FORCE_DIST") == "1"
    if dist_on:
        # RCCL prints a version banner on STDOUT when the first communicator is created; this script's stdout is one
        # JSON line, so file descriptor 1 points at stderr until the communicator exists
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    handle = _capi.Handle.get(local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    B, N, M = args.targets, args.cadences, args.freqs
    out = {}
    if args.workload == "ls":
        # ---- synthetic inputs, rank r owns targets [r*B, (r+1)*B)
        t, y, dy, off = synth.ls_batch(1, B, N, first_index=rank * B)
        for b in range(B):
            t[off[b]:off[b + 1]] -= t[off[b]]
        df = 360.0 / M
        d_t, d_y = torch.from_numpy(t).to(dev), torch.from_numpy(y).to(dev)
        d_pow = torch.empty((B, M), dtype=torch.float64, device=dev)
        d_max = torch.empty(B, dtype=torch.float64, device=dev)
        d_arg = torch.empty(B, dtype=torch.int64, device=dev)
        gather_spec = dist_on and args.gather == "spectra"
        gather_sum = dist_on and args.gather == "summary"
        nch = max(1, min(args.chunks, B)) if gather_spec else 1
        bounds = np.linspace(0, B, nch + 1).astype(int)
        d_all = [torch.empty((world, bounds[c + 1] - bounds[c], M), dtype=torch.float64, device=dev)
                 for c in range(nch)] if gather_spec else None
        d_sum = torch.empty((B, 2), dtype=torch.float64, device=dev)
        d_sum_all = torch.empty((world, B, 2), dtype=torch.float64, device=dev) if gather_sum else None
        headline = "fast" if args.ls_method in ("both", "fast") else "exact"
        nev = 2 * (args.steps + args.warmup)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nev)]

        def ls_step(k, method):
            works = []
            e0, e1 = ev[k]
            for c in range(nch):
                b0, b1 = int(bounds[c]), int(bounds[c + 1])
                offc = off[b0:b1 + 1] - off[b0]
                if c == 0:
                    e0.record()
                if method == "fast":
                    _capi.ls_fast_batch_dev(handle, b1 - b0, offc, d_t.data_ptr() + 8 * int(off[b0]),
                                            d_y.data_ptr() + 8 * int(off[b0]), 0, df, df, M, True, True,
                                            "lk_amplitude", 0, 5, d_pow.data_ptr() + 8 * b0 * M, stream)
                else:
                    _capi.ls_power_batch_dev(handle, b1 - b0, offc, d_t.data_ptr() + 8 * int(off[b0]),
                                             d_y.data_ptr() + 8 * int(off[b0]), 0, 0, df, df, M, True, True,
                                             "lk_amplitude", 0, d_pow.data_ptr() + 8 * b0 * M, stream)
                if c == nch - 1:
                    e1.record()
                if gather_spec:
                    works.append(dist.all_gather_into_tensor(d_all[c], d_pow[b0:b1], async_op=True))
            _capi.argmax_batch_dev(handle, B, M, d_pow.data_ptr(), d_max.data_ptr(), d_arg.data_ptr(), stream)
            if gather_sum:   # every rank ends with every target's (max power, argmax): 16 B per target over xGMI
                d_sum[:, 0] = d_max
                d_sum[:, 1] = d_arg.to(torch.float64)
                dist.all_gather_into_tensor(d_sum_all.view(world * B, 2), d_sum)
            for w in works:
                w.wait()

        def step(k):
            ls_step(k, headline)

        units_per_step = B * M
        pairs_per_step = float(sum(int(off[b + 1] - off[b]) for b in range(B))) * M
        names = {"exact": "exact GLS, direct fp64 trig sums", "fast": "ls_method='fast' (reference default): extirpolation + FFT"}
        metric = "frequencies*targets/sec (Lomb-Scargle, %s)" % names[headline]
        unit = "frequencies*targets/sec"
        workload = ("configs[1]: %d TESS-like %d-cadence targets x %d freqs Lomb-Scargle per GPU, ls_method=%s"
                    % (B, N, M, headline))
    elif args.workload == "pld":
        Bc, Nc, npix = args.cutouts, args.pld_cadences, 11
        P = npix * npix
        cubes = [synth.pld_cutout(4, rank * Bc + i, n=Nc, npix=npix) for i in range(Bc)]
        tt = np.stack([c[0] for c in cubes])
        pix = np.stack([c[1].reshape(Nc, P) for c in cubes]).astype(np.float32)
        epx = np.stack([c[2].reshape(Nc, P) for c in cubes]).astype(np.float32)
        lcf = pix.sum(axis=2, dtype=np.float32)
        lce = np.sqrt((epx.astype(np.float64) ** 2).sum(axis=2))
        deg, nkn = 5, Nc // 50
        n_inner = nkn - deg - 1
        knots = np.stack([np.concatenate([[t.min()], np.percentile(t, np.linspace(0, 100, n_inner + 2)[1:-1]), [t.max()]])
                          for t in tt])
        K = _capi.pld_design_width(P, P, 3, 16, nkn)
        d_pix, d_lcf, d_t, d_kn = (torch.from_numpy(a).to(dev) for a in (pix, lcf, tt, knots))
        d_y, d_err = torch.from_numpy(lcf.astype(np.float64).ravel()).to(dev), torch.from_numpy(lce.ravel()).to(dev)
        d_X = torch.empty((Bc, Nc, K), dtype=torch.float64, device=dev)
        d_ps = torch.empty((Bc, K), dtype=torch.float64, device=dev)
        d_mu = torch.zeros((Bc, K), dtype=torch.float64, device=dev)
        d_w = torch.empty((Bc, K), dtype=torch.float64, device=dev)
        d_model = torch.empty(Bc * Nc, dtype=torch.float64, device=dev)
        d_out = torch.empty(Bc * Nc, dtype=torch.uint8, device=dev)
        offp = np.arange(Bc + 1, dtype=np.int64) * Nc
        lib = _capi.load_library()
        import ctypes
        vp = ctypes.c_void_p
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps + args.warmup)]

        def step(k):
            e0, e1 = ev[k]
            e0.record()
            _capi._check(lib.lk_pld_design_batch_dev(handle._h, Bc, Nc, P, P, vp(d_pix.data_ptr()), vp(d_pix.data_ptr()),
                                                     vp(d_lcf.data_ptr()), vp(d_t.data_ptr()), vp(d_kn.data_ptr()),
                                                     n_inner, 3, 16, nkn, deg, 1, K, vp(d_X.data_ptr()),
                                                     vp(d_ps.data_ptr()), vp(stream)))
            _capi._check(lib.lk_regress_batch_dev(handle._h, Bc, offp.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), K,
                                                  vp(d_X.data_ptr()), vp(d_y.data_ptr()), vp(d_err.data_ptr()), None,
                                                  vp(d_mu.data_ptr()), vp(d_ps.data_ptr()), 5.0, 5, vp(d_w.data_ptr()),
                                                  vp(d_model.data_ptr()), vp(d_out.data_ptr()), vp(stream)))
            e1.record()

        units_per_step = Bc
        gram_cols = [P, 136, 816, P]
        pairs_per_step = float(Bc) * (2.0 * Nc * sum(c * c for c in gram_cols) + 5 * 2.0 * Nc * (K + 1) ** 2)  # MFMA flop
        metric, unit = "PLD cutouts/sec (design matrix + regression)", "cutouts/sec"
        workload = ("configs[4]: %d K2-like 11x11-pixel cutouts x %d cadences, 3rd-order design matrix (K=%d), "
                    "MFMA Gram per GPU" % (Bc, Nc, K))
        B, N = Bc, Nc
    elif args.workload == "flatten":
        t, y, dy, off = synth.ls_batch(6, B, N, first_index=rank * B)
        d_t, d_y = torch.from_numpy(t).to(dev), torch.from_numpy(y).to(dev)
        d_tr = torch.empty_like(d_y)
        lib = _capi.load_library()
        import ctypes
        vp = ctypes.c_void_p
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps + args.warmup)]

        def step(k):
            e0, e1 = ev[k]
            e0.record()
            _capi._check(lib.lk_savgol_trend_batch_dev(handle._h, B, off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                                       vp(d_t.data_ptr()), vp(d_y.data_ptr()), None, 401, 2, 5.0, 3, 3.0,
                                                       vp(d_tr.data_ptr()), None, vp(stream)))
            e1.record()

        units_per_step = int(off[-1])
        pairs_per_step = float(off[-1])
        metric, unit = "flatten cadences/sec (window 401, niters 3)", "cadences/sec"
        workload = "flatten: %d light curves x %d cadences, window 401, polyorder 2, niters 3 per GPU" % (B, N)
    elif args.workload == "regress":
        Bc, Nc, K = args.cutouts, args.pld_cadences, args.regressors
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        d_X = torch.randn((Bc * Nc, K), dtype=torch.float64, device=dev, generator=g)
        d_coef = torch.randn((Bc, K), dtype=torch.float64, device=dev, generator=g) * 1e-3
        d_y = (d_X.view(Bc, Nc, K) @ d_coef.unsqueeze(-1)).reshape(-1) + 1.0 + \
            1e-3 * torch.randn(Bc * Nc, dtype=torch.float64, device=dev, generator=g)
        d_err = torch.full((Bc * Nc,), 1e-3, dtype=torch.float64, device=dev)
        d_mu = torch.zeros(K, dtype=torch.float64, device=dev)
        d_ps = torch.full((K,), 10.0, dtype=torch.float64, device=dev)
        d_w = torch.empty((Bc, K), dtype=torch.float64, device=dev)
        d_model = torch.empty(Bc * Nc, dtype=torch.float64, device=dev)
        d_out = torch.empty(Bc * Nc, dtype=torch.uint8, device=dev)
        offp = np.arange(Bc + 1, dtype=np.int64) * Nc
        lib = _capi.load_library()
        import ctypes
        vp = ctypes.c_void_p
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps + args.warmup)]

        def step(k):
            e0, e1 = ev[k]
            e0.record()
            _capi._check(lib.lk_regress_batch_dev(handle._h, Bc, offp.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), K,
                                                  vp(d_X.data_ptr()), vp(d_y.data_ptr()), vp(d_err.data_ptr()), None,
                                                  vp(d_mu.data_ptr()), vp(d_ps.data_ptr()), 5.0, 5, vp(d_w.data_ptr()),
                                                  vp(d_model.data_ptr()), vp(d_out.data_ptr()), vp(stream)))
            e1.record()

        off = offp
        units_per_step = Bc
        pairs_per_step = float(Bc) * 5 * 2.0 * Nc * (K + 1) ** 2   # Gram flop over the 5 clip iterations
        metric, unit = "RegressionCorrector fits/sec (N=%d, K=%d, 5 sigma-clip iterations)" % (Nc, K), "fits/sec"
        workload = "configs[4] regression stage: %d fits, N=%d cadences, K=%d regressors per GPU" % (Bc, Nc, K)
        B, N = Bc, Nc
    elif args.workload == "lschi2":
        t, y, dy, off = synth.ls_batch(1, B, N, first_index=rank * B)
        for b in range(B):
            t[off[b]:off[b + 1]] -= t[off[b]]
        d_t, d_y = torch.from_numpy(t).to(dev), torch.from_numpy(y).to(dev)
        d_p = torch.empty((B, M), dtype=torch.float64, device=dev)
        df = 360.0 / M
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps + args.warmup)]

        def step(k):
            e0, e1 = ev[k]
            e0.record()
            _capi.ls_power_batch_dev(handle, B, off, d_t.data_ptr(), d_y.data_ptr(), 0, 0, df, df, M, True, True,
                                     "lk_amplitude", 0, d_p.data_ptr(), stream, nterms=args.nterms)
            e1.record()

        units_per_step = B * M
        pairs_per_step = float(off[-1]) * M
        metric, unit = "frequencies*targets/sec (Lomb-Scargle, nterms=%d, exact chi2)" % args.nterms, "frequencies*targets/sec"
        workload = "%d targets x %d cadences x %d frequencies, nterms=%d multi-term Lomb-Scargle per GPU" % (B, N, M, args.nterms)
    elif args.workload == "pgsmooth":
        from lightkurve_amd.periodogram import _logmedian_windows
        f = (360.0 / M) * (1 + np.arange(M))
        tabs = [np.ascontiguousarray(a, dtype=np.int32) for a in _logmedian_windows(f, 0.01)]
        d_p = torch.rand((B, M), dtype=torch.float64, device=dev)
        d_o = torch.empty_like(d_p)
        lib = _capi.load_library()
        import ctypes
        vp, i32p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps + args.warmup)]

        def step(k):
            e0, e1 = ev[k]
            e0.record()
            _capi._check(lib.lk_pg_logmedian_batch_dev(handle._h, B, M, vp(d_p.data_ptr()), int(tabs[0].size),
                                                       tabs[0].ctypes.data_as(i32p), tabs[1].ctypes.data_as(i32p),
                                                       tabs[2].ctypes.data_as(i32p), tabs[3].ctypes.data_as(i32p),
                                                       (8.0 / 9.0) ** 3, vp(d_o.data_ptr()), vp(stream)))
            e1.record()

        off = np.array([0, B * M])
        units_per_step = B * M
        pairs_per_step = float(B) * float(np.sum(tabs[1] - tabs[0]))   # window memberships = values the medians look at
        metric, unit = "frequencies*targets/sec (Periodogram.smooth logmedian, filter_width 0.01)", "frequencies*targets/sec"
        workload = "%d periodograms x %d frequencies, logmedian smoothing (%d windows) per GPU" % (B, M, tabs[0].size)
    elif args.workload == "fold":
        t, y, dy, off = synth.ls_batch(6, B, N, first_index=rank * B)
        d_t, d_y, d_e = (torch.from_numpy(a).to(dev) for a in (t, y, dy))
        d_ph, d_fy, d_fe = torch.empty_like(d_t), torch.empty_like(d_t), torch.empty_like(d_t)
        d_ord = torch.empty(len(t), dtype=torch.int64, device=dev)
        period = np.linspace(0.7, 9.0, B)
        epoch = np.array([t[off[b]] for b in range(B)])
        wrap = period / 2
        lib = _capi.load_library()
        import ctypes
        vp, dp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)
        cin = (vp * 2)(d_y.data_ptr(), d_e.data_ptr())
        cout = (vp * 2)(d_fy.data_ptr(), d_fe.data_ptr())
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps + args.warmup)]

        def step(k):
            e0, e1 = ev[k]
            e0.record()
            _capi._check(lib.lk_fold_batch_dev(handle._h, B, off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                               vp(d_t.data_ptr()), period.ctypes.data_as(dp), epoch.ctypes.data_as(dp), 0.0,
                                               wrap.ctypes.data_as(dp), 0, 2, cin, cout, vp(d_ph.data_ptr()),
                                               vp(d_ord.data_ptr()), vp(stream)))
            e1.record()

        units_per_step = int(off[-1])
        pairs_per_step = float(off[-1])
        metric, unit = "fold cadences/sec (phase + stable sort + 2 gathered columns)", "cadences/sec"
        workload = "fold: %d light curves x %d cadences per GPU" % (B, N)
    else:
        t, y, dy, off = synth.bls_batch(3, B, N, first_index=rank * B)
        ivar = 1.0 / dy ** 2
        for b in range(B):
            s = slice(off[b], off[b + 1])
            t[s] -= t[s].min()
            y[s] -= np.median(y[s])
        period, duration = synth.bls_grid(args.periods, args.durations)
        d_t, d_y, d_w = (torch.from_numpy(a).to(dev) for a in (t, y, ivar))
        d_per = torch.from_numpy(period).to(dev)
        d_out = torch.empty((7, B, len(period)), dtype=torch.float64, device=dev)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(args.steps + args.warmup)]

        def step(k):
            e0, e1 = ev[k]
            e0.record()
            _capi.bls_batch_dev(handle, B, off, d_t.data_ptr(), d_y.data_ptr(), d_w.data_ptr(), period,
                                d_per.data_ptr(), duration, 10, True, d_out.data_ptr(), stream)
            e1.record()

        units_per_step = B * len(period)
        nb = np.ceil(period / (duration.min() / 10)) + 10
        durb = np.unique(np.round(duration / (duration.min() / 10)))
        cand = float(np.sum(np.maximum(nb[:, None] - durb[None, :] + 1, 0)))
        pairs_per_step = cand * B     # (start bin, duration) candidates
        metric, unit = "BLS periods*targets/sec", "periods*targets/sec"
        workload = ("configs[3]: %d targets x %d cadences, %d periods x %d durations BLS per GPU"
                    % (B, N, len(period), len(duration)))

    def sync():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync()
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        step(k)
    sync()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kern_ms = float(np.mean([ev[k][0].elapsed_time(ev[k][1]) for k in range(args.warmup, args.warmup + args.steps)]))
    second = None
    if args.workload == "ls" and args.ls_method == "both":
        # the other Lomb-Scargle method, measured the same way (W warmup + K timed steps, barrier + sync, max over ranks)
        other = "exact" if headline == "fast" else "fast"
        k0 = args.warmup + args.steps
        for k in range(k0, k0 + args.warmup):
            ls_step(k, other)
        sync()
        t1 = time.perf_counter()
        for k in range(k0 + args.warmup, k0 + args.warmup + args.steps):
            ls_step(k, other)
        sync()
        dt2 = time.perf_counter() - t1
        if dist_on:
            tt = torch.tensor([dt2], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt2 = float(tt.item())
        kms2 = float(np.mean([ev[k][0].elapsed_time(ev[k][1])
                              for k in range(k0 + args.warmup, k0 + args.warmup + args.steps)]))
        second = (other, dt2, kms2)

    if rank == 0:
        value = units_per_step * world * args.steps / dt
        out = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "targets_per_gpu": B, "cadences": N,
                       "parallelism": "targets sharded over %d rank(s), no data-path collective%s"
                                      % (world, ("; RCCL all-gather of %s each step" % (
                                          "the per-target (max power, argmax)" if args.gather == "summary" else
                                          "the full spectra, overlapped with compute"))
                                         if world > 1 and args.workload == "ls" and args.gather != "none" else "")},
        }
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))
                if args.workload != "ls":
                    traffic = traffic.get(args.workload)
            except Exception:
                traffic = None
        def ls_roofline(method, kms):
            if method == "fast":
                nfft = 1 << int(np.ceil(np.log2(5 * M)))
                m2 = int(np.log2(nfft)) // 2
                n2 = 1 << m2
                # rows of the Nfft = N1 x N2 grid that can hold samples (what the spreader writes and step 1 reads)
                used = 0.0
                for b in range(B):
                    span = (t[off[b + 1] - 1] - t[off[b]]) * nfft * df
                    used += 2 * min(nfft, (int((span + 4) / n2) + 1) * n2) + min(nfft, (int((2 * span + 4) / n2) + 1) * n2)
                algo = (B * (3 * nfft * 16.0 * 2 + 3 * M * 16.0 * 2 + 8.0 * M) + used * 16.0 * 2 + 16.0 * float(off[-1]))
                return {"bound": "hbm", "achieved": algo / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": algo / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": (traffic or {}).get("ls_fast")
                        if isinstance(traffic, dict) else None,
                        "kernel": "fft_cols_reg_kernel + fft_rows_reg_kernel (+ spreader, epilogue): whole step",
                        "kernel_ms_per_step": kms, "algorithmic_bytes_per_step": algo,
                        "note": "algorithmic bytes per target: 3 complex grids of Nfft=%d written by FFT step 1 and read by "
                                "step 2 (2 x 16 B x Nfft each), the sample-bearing rows written by the spreader and read by "
                                "step 1, 3 spectra of M written + read, 16 B/cadence in, 8 B/frequency out" % nfft}
            flops = 16.0 * pairs_per_step
            ach = flops / (kms * 1e-3) / 1e12
            algo_bytes = 16.0 * float(off[-1]) + 8.0 * B * M
            return {"bound": "valu", "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / FP64_VECTOR_PEAK_TFLOPS, "traffic": (traffic or {}).get("ls") if isinstance(traffic, dict)
                    else traffic, "kernel": "ls_grid_kernel<16> (+ls_prep_kernel)", "kernel_ms_per_step": kms,
                    "note": "fp64 direct trig sums: 16 flop (8 v_fma_f64) per (cadence, frequency) pair; arithmetic "
                            "intensity ~3e4 flop/B so the fp64 VECTOR pipe binds (peak == fp64 MFMA dense peak), not HBM",
                    "hbm": {"achieved": algo_bytes / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": algo_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "algorithmic_bytes_per_step": algo_bytes}}

        if args.workload == "ls":
            out["roofline"] = ls_roofline(headline, kern_ms)
            out["config"]["ls_method"] = headline
            if second is not None:
                other, dt2, kms2 = second
                out["other_method"] = {"ls_method": other, "value": units_per_step * world * args.steps / dt2,
                                       "unit": unit, "ms_per_step": 1e3 * dt2 / args.steps,
                                       "roofline": ls_roofline(other, kms2),
                                       "note": "same workload, same timing protocol; 'exact' = the direct-sum kernel "
                                               "(matches the reference's slow/cython/chi2 to 1e-9), 'fast' = the reference's "
                                               "default algorithm (matches lightkurve's default output to 1e-9)"}
        elif args.workload == "pld":
            ach = pairs_per_step / (kern_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / FP64_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                               "kernel": "gram_mfma_kernel (v_mfma_f64_16x16x4_f64)", "kernel_ms_per_step": kern_ms,
                               "note": "algorithmic 2*N*P^2 flop per PCA Gram (P = 121, 136, 816, 121) + 5 x 2*N*(K+1)^2 "
                                       "for the regression, over the WHOLE step time (eigen-solver, projections, LU, "
                                       "clipping included), against the fp64 MFMA dense peak"}
        elif args.workload == "flatten":
            algo = 24.0 * pairs_per_step
            out["roofline"] = {"bound": "hbm", "achieved": algo / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": algo / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                               "kernel": "flatten_kernel", "kernel_ms_per_step": kern_ms,
                               "note": "algorithmic 24 B per cadence (time, flux in; trend out)"}
        elif args.workload == "regress":
            ach = pairs_per_step / (kern_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / FP64_VECTOR_PEAK_TFLOPS, "traffic": None,
                               "kernel": "gram_mfma_kernel (v_mfma_f64_16x16x4_f64)", "kernel_ms_per_step": kern_ms,
                               "note": "algorithmic 2*N*(K+1)^2 flop per Gram build x 5 sigma-clip iterations (SURVEY.md "
                                       "8(d)); the step also runs 5 LU solves, 5 model products and 5 sigma-clips per fit; "
                                       "X (N*K*8 B per fit) is re-read each iteration"}
        elif args.workload == "lschi2":
            fl = 2.0 * (4 + 4 * (2 * args.nterms - 1) + 6 * args.nterms) * pairs_per_step
            out["roofline"] = {"bound": "valu", "achieved": fl / (kern_ms * 1e-3) / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": fl / (kern_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                               "traffic": None, "kernel": "ls_chi2_grid_kernel<%d>" % args.nterms,
                               "kernel_ms_per_step": kern_ms,
                               "note": "4 + 4(2n-1) + 6n FMAs per (cadence, frequency) pair (phasor recurrence, Chebyshev "
                                       "harmonics, 6n accumulates), n = nterms"}
        elif args.workload == "pgsmooth":
            algo = 16.0 * units_per_step
            out["roofline"] = {"bound": "hbm", "achieved": algo / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": algo / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "pg_window_median_kernel (+pg_window_average_kernel)", "kernel_ms_per_step": kern_ms,
                               "note": "algorithmic 16 B per frequency (power in, smoothed out); every value sits in ~4 "
                                       "overlapping windows and each window median is an 8-pass radix select over its "
                                       "members (%.3g member reads per step, served by L2)" % pairs_per_step}
        elif args.workload == "fold":
            algo = 72.0 * pairs_per_step
            out["roofline"] = {"bound": "hbm", "achieved": algo / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": algo / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "fold_kernel (+2 fold_gather_kernel)", "kernel_ms_per_step": kern_ms,
                               "note": "algorithmic 72 B per cadence (t in, phase + order out, two columns in and out + "
                                       "the order read per gather); the sort itself runs in LDS tiles / an L2-resident slab"}
        else:
            out["roofline"] = {
                "bound": "valu", "achieved": 12.0 * pairs_per_step / (kern_ms * 1e-3) / 1e12,
                "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": 12.0 * pairs_per_step / (kern_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                "traffic": traffic, "kernel": "bls_kernel", "kernel_ms_per_step": kern_ms,
                "note": "algorithmic 12 flop per (start bin, duration) candidate (SURVEY.md §8(d)); the bit-exact "
                        "kernel skips 80-95 % of the candidates with a rigorous growth bound, so this is an equivalent rate",
            }
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
            out["speedup_vs_cpu_baseline"] = value / cpu_base["value"]
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
