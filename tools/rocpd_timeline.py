#!/usr/bin/env python3
"""Timeline of the LAST outer iteration in a rocprofv3 rocpd db: per kernel start offset, duration, gap to previous.
usage: rocpd_timeline.py results.db [n_last_kernels]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = list(cur.execute("select s.display_name, d.start, d.end, d.grid_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
# windows between consecutive k_ss_finalize_decide launches = outer iterations; of the last ten, show the one with the smallest span
# (the profiler stalls the queue now and then: an iteration with a 100 us hole in it says nothing about the kernels)
idx = [i for i, r in enumerate(rows) if "finalize_decide" in r[0]]
if len(idx) >= 3:
    cands = [(rows[idx[j + 1]][1] - rows[idx[j]][1], idx[j], idx[j + 1]) for j in range(max(0, len(idx) - 11), len(idx) - 1)]
    _, a, b = min(cands)
    win = rows[a:b]
else:
    win = rows[-N:]
t0 = win[0][1]; prev_end = None; tot_dur = 0; tot_gap = 0
for r in win:
    nm = re.sub(r"\(.*", "", r[0]).replace("void smg::", "").replace("smg::", "")
    gap = (r[1] - prev_end) if prev_end is not None else 0
    print("%8.2f us  dur %6.2f  gap %6.2f  grid %8d  %s" % ((r[1]-t0)/1e3, (r[2]-r[1])/1e3, gap/1e3, r[3], nm[:60]))
    prev_end = r[2]; tot_dur += r[2]-r[1]; tot_gap += max(gap, 0)
print("window: %d kernels, span %.1f us, sum(dur) %.1f us, sum(gap) %.1f us" % (len(win), (win[-1][2]-t0)/1e3, tot_dur/1e3, tot_gap/1e3))
