#!/usr/bin/env python
"""List the global loads of a .hip file's kernels that sit behind a branch with a full wait right after them.

    python tools/isa_guarded_loads.py lightkurve_amd/csrc/pld.hip [--lines] [extra hipcc flags ...]

`cond ? load : 0`, `if (in) x = load`, a short-circuit `a[i] && b[i]` ... compile to
    s_cbranch_execz  ->  global_load  ->  s_waitcnt vmcnt(0)
— every such load is a memory round trip of its own, and loads written "unconditional on a clamped address, selected
afterwards" are turned back into this form by the compiler when the value is only used under the condition (DESIGN.md
section 5, "serialised loads").  The script compiles the file to gfx950 assembly (no GPU needed) and counts, per kernel, the
branches that are followed within a few instructions by a global load with `vmcnt(0)` behind it; with --lines the sites are
mapped back to source lines (-gline-tables-only).  A count is a place to look, not a verdict: a guarded load in a cold path
costs nothing.
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def main():
    args = sys.argv[1:]
    if not args:
        sys.exit(__doc__)
    src = args[0]
    lines_mode = "--lines" in args
    extra = [a for a in args[1:] if a != "--lines"]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
               "-I" + os.path.dirname(os.path.abspath(src)), src, "-o", asm] + extra
        if lines_mode:
            cmd.append("-gline-tables-only")
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read().split("\n")
    files = {}
    for l in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[m.group(1)] = os.path.basename(m.group(3) or m.group(2))
    funcs, cur, loc = {}, None, None
    for l in text:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        t = l.strip()
        if t.startswith(".Lfunc_end"):
            cur = None
        if cur is None:
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = (files.get(m.group(1), "?"), int(m.group(2)))
            continue
        if t and not t.startswith(";") and not t.startswith("."):
            funcs[cur].append((t, loc))
    for name, ins in funcs.items():
        sites = Counter()
        for i, (x, _) in enumerate(ins):
            if not x.startswith("s_cbranch"):
                continue
            window = ins[i + 1:i + 10]
            for j, (y, l2) in enumerate(window):
                if y.startswith("global_load") or y.startswith("buffer_load"):
                    if any("vmcnt(0)" in z for z, _ in window[j + 1:j + 4]):
                        sites[l2] += 1
                    break
        total = sum(sites.values())
        if total < 2:
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        nload = sum(1 for x, _ in ins if x.startswith("global_load") or x.startswith("buffer_load"))
        print("%-90s loads %4d  guarded-then-wait %3d" % (dem[:90], nload, total))
        if lines_mode:
            for (f, ln), c in sorted((k, v) for k, v in sites.items() if k):
                print("      %s:%d  x%d" % (f, ln, c))


if __name__ == "__main__":
    main()
