# debugging aid: adds wall_clock64 phase timers to flatten.hip (LK_FLAT_DEBUG=1 prints them); never committed applied
import sys
p='lightkurve_amd/csrc/flatten.hip'; s=open(p).read()
open('/tmp/flatten_orig.hip','w').write(s)
s=s.replace("const int64_t *__restrict__ scratch_off, double *__restrict__ trend, uint8_t *__restrict__ final_mask) {","const int64_t *__restrict__ scratch_off, double *__restrict__ trend, uint8_t *__restrict__ final_mask, int debug) {\n    long long tk[12]; int ntk = 0;\n#define TICK() do { if (debug && ntk < 12) tk[ntk++] = wall_clock64(); } while (0)\n    TICK();")
s=s.replace("    for (int it = 0; it < niters; ++it) {\n        const int nm = block_compact(","    TICK();\n    for (int it = 0; it < niters; ++it) {\n        if (it == 1) TICK();\n        const int nm = block_compact(")
for key in ["        // ---- gap segmentation: cut where","        // ---- per segment: median for short ones","        // ---- clip: |flux - trend|","        // ---- linear interpolation / extrapolation","        // ---- mask[mask] &= mask1"]:
    s=s.replace(key,"        if (it == 1) TICK();\n"+key)
s=s.replace("    if (final_mask)\n        for (int i = tid; i < N; i += nt) final_mask[i] = mask[i];\n}","    if (final_mask)\n        for (int i = tid; i < N; i += nt) final_mask[i] = mask[i];\n    TICK();\n    if (debug && blockIdx.x == 0 && tid == 0) { for (int q = 1; q < ntk; ++q) printf(\"phase %d: %lld ticks\\n\", q, tk[q] - tk[q-1]); }\n}")
s=s.replace("break_tol, niters, sigma, d_c, d_e, d_s, d_soff, trend, final_mask);","break_tol, niters, sigma, d_c, d_e, d_s, d_soff, trend, final_mask, getenv(\"LK_FLAT_DEBUG\") ? 1 : 0);")
open(p,'w').write(s)
