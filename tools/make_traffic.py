#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries of tools/profile_round.sh: HBM (fabric-side) bytes per launch of the fine-level
y = A x kernel, per workload.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 64 B per 128-B request,
MI355X_MICROARCH.md HBM section), cross-checked by TCC_MISS_sum * 128 B.
usage: make_traffic.py pmc_summary_C3.json pmc_summary_C5.json tag"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
# the kernel sources these passes ran on: bench.py reports a committed traffic figure only while this hash matches the tree it runs from
out = {"kernel_source_sha": bench.kernel_source_hash()}
tag = sys.argv[3]
for wl, path in (("C3", sys.argv[1]), ("C5", sys.argv[2])):
    d = json.load(open(path))
    # the whole-matrix SpMV of the fine level: SELL_AX (mode 0), one column, the largest grid
    cand = [(k, v) for k, v in d.items() if k.startswith("k_sell<0") or k.startswith("k_sell<(smg::SellMode)0")]
    if not cand:
        continue
    k, v = max(cand, key=lambda kv: int(kv[0].split("grid=")[1]))
    out[wl] = {"kernel": k, "hbm_bytes_per_launch": int(v.get("hbm_bytes_per_launch_fetchx2_plus_write", 0)),
               "fetch_size_kib": v.get("FETCH_SIZE"), "write_size_kib": v.get("WRITE_SIZE"), "tcc_miss_x128B": v.get("tcc_miss_x128B"),
               "avg_ns_under_pmc": v.get("avg_ns_under_pmc"),
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                         "(gfx950: FETCH_SIZE counts 64 B per 128 B request, MI355X_MICROARCH.md HBM section); cross-check TCC_MISS_sum*128 B",
               "source": "profiles/%s_pmc_summary_%s.json" % (tag, wl)}
print(json.dumps(out, indent=1))
