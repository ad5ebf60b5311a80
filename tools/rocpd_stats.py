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
# the SELL kernels serve every level under one symbol: break the big launches out by grid size, so that e.g. the fine-level
# y = A x (k_sell<0,...>, grid = 4 slices per 256-thread block over the finest matrix) has its own row
q2 = ("select s.%s, d.grid_size_x, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
      "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id where s.%s like '%%k_sell%%' "
      "group by s.%s, d.grid_size_x having d.grid_size_x >= 200000 order by 4 desc" % (name_col, name_col, name_col))
lines.append("# per (kernel, grid) for SELL launches of >= 200000 threads: kernel @grid,calls,total_ns,avg_ns,min_ns,max_ns,pct")
for r in cur.execute(q2):
    nm = re.sub(r"\(.*", "", re.sub(r"\s+", " ", str(r[0]))).replace(",", ";")
    lines.append("%s @grid=%d,%d,%d,%.1f,%d,%d,%.2f" % (nm, r[1], r[2], r[3], r[4], r[5], r[6], 100.0 * r[3] / tot))
txt = "\n".join(lines)
if len(sys.argv) > 2: open(sys.argv[2], "w").write(txt + "\n")
print(txt)
