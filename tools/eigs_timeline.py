#!/usr/bin/env python3
"""Timeline of one phase-split eigen solve (pld_eigs_* launches) from a rocprofv3 result db: tools/eigs_timeline.py <db> [which]"""
import sqlite3, sys
db = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 6
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = con.execute(f"select s.kernel_name, d.start, d.end, d.queue_id, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if 'eigs_init' in r[0]]
def short(n):
    for k in ['init', 'prod', 'rr', 'orth', 'topk']:
        if k in n:
            return k
    return n[:20]
i0 = idx[which]
t0 = rows[i0][1]
j = i0
while 'topk_eig' not in rows[j][0]:
    j += 1
print('matrices', rows[i0][4] // rows[i0][5], 'span us', (rows[j][2] - t0) / 1e3)
tot = {}
for r in rows[i0:j + 1]:
    d = (r[2] - r[1]) / 1e3
    tot[short(r[0])] = tot.get(short(r[0]), 0) + d
    if d > 20:
        print(f"{short(r[0]):8s} q{r[3]} start {(r[1]-t0)/1e3:9.1f} dur {d:8.1f}")
print(tot)
