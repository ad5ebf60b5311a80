#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd sqlite db.  usage: rocpd_pmc.py results.db [out.csv]"""
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
def cols(t): return [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
pc, ic, kc, sc = cols("rocpd_pmc_event"), cols("rocpd_info_pmc"), cols("rocpd_kernel_dispatch"), cols("rocpd_info_kernel_symbol")
name_col = "display_name" if "display_name" in sc else "kernel_name"
q = ("select s.%s, p.name, count(*), avg(e.value), sum(e.value) from rocpd_pmc_event e "
     "join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id "
     "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.%s, p.name order by 1, 2" % (name_col, name_col))
try:
    rows = list(cur.execute(q))
except Exception as ex:
    print("schema:", pc, ic, [c for c in kc]); raise
lines = ["kernel,counter,dispatches,avg,sum"]
for r in rows:
    nm = re.sub(r"\s+", " ", str(r[0])).replace(",", ";")
    lines.append("%s,%s,%d,%.1f,%.1f" % (nm, r[1], r[2], r[3], r[4]))
txt = "\n".join(lines)
if len(sys.argv) > 2: open(sys.argv[2], "w").write(txt + "\n")
print(txt)
