#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max duration.
usage: python tools_rocpd_stats.py results.db [out.csv]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
scols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else scols[1])
q = ("select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
     "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, name_col))
rows = list(cur.execute(q))
tot = sum(r[2] for r in rows) or 1
lines = ["kernel,calls,total_ns,avg_ns,min_ns,max_ns,pct"]
for r in rows:
    nm = re.sub(r"\s+", " ", str(r[0])).replace(",", ";")
    lines.append("%s,%d,%d,%.1f,%d,%d,%.2f" % (nm, r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
txt = "\n".join(lines)
if len(sys.argv) > 2: open(sys.argv[2], "w").write(txt + "\n")
print(txt)
