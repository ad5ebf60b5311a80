#!/usr/bin/env python3
"""Per-dispatch summary of the coarse-matrix inversion kernels (k_gj_*) in a rocprofv3 rocpd database:
median / min / max duration per kernel over the last inversion and its total span.  usage: gj_timeline.py results.db"""
import sqlite3, sys, statistics
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
scols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "kernel_name" if "kernel_name" in scols else scols[1]
rows = list(cur.execute("select s.%s, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                        "on d.kernel_id = s.id where s.%s like '%%k_gj%%' or s.%s like '%%k_mirror%%' order by d.start" % (name_col, name_col, name_col)))
last = max(i for i, r in enumerate(rows) if "k_gj_diag" in r[0])
rows = rows[last:]
by = {}
for nm, st, en in rows:
    k = nm.split("(")[0].split("::")[-1]
    by.setdefault(k, []).append((en - st) / 1e3)
for k, v in by.items():
    print("%-28s n=%3d median %6.1f us  min %6.1f  max %6.1f  sum %7.1f us" % (k, len(v), statistics.median(v), min(v), max(v), sum(v)))
print("inversion span: %.3f ms" % ((rows[-1][2] - rows[0][1]) / 1e6))
