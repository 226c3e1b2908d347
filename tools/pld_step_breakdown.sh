#!/bin/bash
# Per-kernel time of ONE 500-cutout PLD step from a kernel trace: the averages of rocprofv3 --stats mix the bench's big launches with the
# 3-cutout accuracy launches (bandwidth-bound kernels take ~10 us there, latency-bound ones as long as in a big launch), so this sums each
# kernel's dispatches inside ONE timed step (the middle one of the 500-cutout steps) instead.  tools/pld_step_breakdown.sh <outdir> [VAR=value ...]
out=$1; shift; mkdir -p "$out"; R=$PWD; export TMPDIR=/tmp; cd /tmp
env "$@" rocprofv3 --kernel-trace -d "$R/$out/trace" -o pld -- python "$R/bench.py" --workload pld --no-cpu-baseline --no-api --steps 3 --warmup 1 > "$R/$out/bench.json" 2> "$R/$out/bench.err"
cd "$R"
db=$(ls $out/trace/*/*results.db $out/trace/*results.db 2>/dev/null | head -1)
python - "$db" <<'PY'
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = con.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
# one step = from a pld_ratio_kernel launch that follows a regression kernel (clip / model / solve of the previous step) or starts the
# trace, to the next such launch; the 500-cutout steps are the longest
starts = [i for i, r in enumerate(rows) if "pld_rowdiv_kernel" in r[0] or (i == 0)]
# rowdiv runs once per step, early; back up to the first kernel after the previous step's last clip / model kernel
def step_begin(i):
    j = i
    while j > 0 and not any(t in rows[j - 1][0] for t in ("clip_kernel", "model_kernel", "model_part_kernel", "solve_lds_kernel")):
        j -= 1
    return j
begins = sorted(set(step_begin(i) for i in starts))
spans = [(begins[j], begins[j + 1]) for j in range(len(begins) - 1)] + [(begins[-1], len(rows))]
dur = [(sum(r[2] - r[1] for r in rows[a:b]), a, b) for a, b in spans if b > a]
big = max(dur)[0]
cand = [(a, b) for d, a, b in dur if d > 0.7 * big]
a, b = cand[len(cand) // 2]  # a timed step (the first steps after an idle GPU run 5-20 % slower: clocks)
acc = collections.OrderedDict()
for name, st, en in rows[a:b]:
    short = name.split("(")[0].replace("void ", "").replace("lk::", "")
    acc.setdefault(short, [0, 0.0])
    acc[short][0] += 1
    acc[short][1] += (en - st) * 1e-3
# every launch of the three longest kernels, in order (spread between steps / clock block / accuracy launches)
for key in ("pld_moment_gram_kernelILi3ELb0", "pld_topk_eig_kernel", "pld_project_kernel"):
    ds = ["%.0f" % ((en - st) * 1e-3) for name, st, en in rows if key in name]
    print("# all launches of %s: %s" % (key, " ".join(ds)))
print("# the step's launches in order (us): " + " | ".join("%s %.0f" % (name.split("(")[0].replace("void ", "").replace("lk::", "").replace("_kernel", "")[:28], (en - st) * 1e-3) for name, st, en in rows[a:b] if "clock_probe" not in name))
tot = sum(v[1] for v in acc.values())
print("# one 500-cutout PLD step: span %.0f us, kernels %.0f us" % ((rows[b - 1][2] - rows[a][1]) * 1e-3, tot))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-50s %3d launches %9.1f us %5.1f %%" % (k[:50], v[0], v[1], 100 * v[1] / tot))
PY
find $out -name "*.db" -delete
