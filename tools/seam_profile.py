#!/usr/bin/env python
"""cProfile of single seam calls (run like tools/seams_latency.py, under the conda interpreter with the staged reference):
where the milliseconds of a B = 1 call go once the kernels take a fraction of one."""
import cProfile, pstats, io, sys, warnings, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import seams_latency as SL

def main():
    import lightkurve as lk
    from lightkurve_amd import seams
    warnings.simplefilter("ignore")
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    want = sys.argv[2:] or ["RegressionCorrector", "method='bls'"]
    cs = [c for c in SL.cases(lk, N) if any(w in c[0] for w in want)]
    seams.install()
    try:
        for name, fn, _ in cs:
            fn(); fn()
            pr = cProfile.Profile(); pr.enable(); fn(); pr.disable()
            s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
            print("## " + name); print("\n".join(s.getvalue().splitlines()[:48]))
    finally:
        seams.uninstall()

if __name__ == "__main__":
    main()
