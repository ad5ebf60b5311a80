#!/usr/bin/env python3
"""A/B of the level-0 Gauss-Seidel sweep under an environment knob read once per process (SMG_GS_LDS, SMG_GS_WPB, ...):
    tools/gs_sweep_ab.py KNOB=value [KNOB=value ...]      runs the C3 sweep without and with the knobs (two child processes each way,
alternated), prints us per sweep, GB/s, and whether the iterate after 3 sweeps has the same bits."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import hashlib, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import bench as B
import surface_multigrid_code_amd as smg
from surface_multigrid_code_amd import mesh
wl = sys.argv[2]
mg, A, Mb, Vf, Ff, label, _ = B.build_workload(wl, smg, mesh)
mg.precompute(A)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); mg.set_stream(st.cuda_stream)
n = A.shape[0]
rng = np.random.default_rng(7)
b = torch.from_numpy(Mb @ rng.uniform(-1, 1, n)).to(dev)
u = torch.from_numpy(rng.uniform(-1, 1, n)).to(dev)
mg.raw_relax(0, b.data_ptr(), u.data_ptr(), 1, 3)
torch.cuda.synchronize()
hsh = hashlib.sha256(u.cpu().numpy().tobytes()).hexdigest()[:16]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    for _ in range(10): mg.raw_relax(0, b.data_ptr(), u.data_ptr(), 1, 1)
    torch.cuda.synchronize(); e0.record(st)
    for _ in range(200): mg.raw_relax(0, b.data_ptr(), u.data_ptr(), 1, 1)
    e1.record(st); torch.cuda.synchronize()
    best = min(best, 1e3 * e0.elapsed_time(e1) / 200)
byt = 12 * A.nnz + 4 * (n + 1) + 24 * n
cyc = mg.bench_vcycle(0, 1, 2, 2, 100)
print("RESULT %s sweep_us %.3f gbs %.0f frac %.3f vcycle_us %.2f" % (hsh, best, byt / best / 1e3, byt / best / 1e3 / 8000, cyc))
"""
knobs = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
wl = next((a for a in sys.argv[1:] if "=" not in a), "C3")
for rnd in range(2):
    for name, extra in (("base", {}), ("knob", knobs)):
        env = dict(os.environ); env.update(extra)
        r = subprocess.run([sys.executable, "-c", CHILD, ROOT, wl], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print(name, extra, line[0] if line else ("FAILED: " + r.stderr[-800:]))
