"""CPU: the line bench.py prints for the driver.  Round 5's line grew to 23.7 KB and the driver could not parse it; the line is now built by
`bench.compact_line` from the full record (which goes to bench_extra.json) and is capped at 8 000 characters, strict JSON."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench as B

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "multi_gpu_preflight")


def _canned():
    """the largest full record ever produced (round 5, 23.7 KB) as the canned input"""
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_c3_n1.json")))


def test_bench_line_is_compact_and_strict_json():
    full = _canned()
    line = B.compact_line(full)
    s = json.dumps(line, allow_nan=False)
    assert "\n" not in s and len(s) < 6000 < B.LINE_HARD_CAP, len(s)
    back = json.loads(s)
    for k in CONTRACT:
        assert k in back, k
    assert back["value"] == float("%.6g" % full["value"]) and back["steps"] == full["steps"] and back["n_gpus"] == 1
    assert set(("workload", "n_verts", "nnz", "levels", "cycle", "smoother")) <= set(back["config"])
    rf = back["roofline"]
    assert set(("kernel", "bound", "achieved", "peak", "unit", "frac", "bytes_per_launch", "us_per_launch", "traffic")) <= set(rf)
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5
    # achieved = algorithmic bytes / launch time, to the rounding of the line
    assert abs(rf["achieved"] - rf["bytes_per_launch"] / rf["us_per_launch"] * 1e-3) < 1e-3 * rf["achieved"]
    cb = back["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample", "ms_per_cycle")) <= set(cb) and cb["kind"] in ("port", "reference")
    assert len(back["extra"]) >= 10 and all(not isinstance(v, (dict, list)) for v in back["extra"].values())


def test_bench_line_survives_nan_errors_and_a_multi_gpu_record():
    full = _canned()
    full["value"] = float("nan")                              # strict JSON has no NaN: becomes null, the line still parses
    full["roofline"]["traffic"] = None
    full["roofline_c5"] = {"error": "RuntimeError('x' * 1000)" + "x" * 1000}
    full["multi_gpu_preflight"] = {"rccl_comm_ranks": 8, "distinct_devices": 8, "devices": ["node|device %d|%s" % (i, "u" * 60) for i in range(8)],
                                   "device_info": ["u" * 60] * 8, "visible_devices_per_rank": 8, "backend": "nccl", "allreduce": "x" * 200, "sharing_detectable": True}
    full["n_gpus"] = 8
    del full["cpu_baseline"]
    s = json.dumps(B.compact_line(full), allow_nan=False)
    back = json.loads(s)
    assert len(s) < B.LINE_HARD_CAP and back["value"] is None and back["roofline"]["traffic"] is None
    assert back["multi_gpu_preflight"]["rccl_comm_ranks"] == 8 and "devices" not in back["multi_gpu_preflight"]
    assert "roofline_c5" in back["leg_errors"] and len(back["leg_errors"]["roofline_c5"]) <= 120
    assert "cpu_baseline" not in back


def test_bench_line_sheds_optional_parts_before_it_breaks_the_cap(monkeypatch):
    full = _canned()
    full["config"]["workload"] = "w" * 7000                  # prose upstream is cut at 240 characters ...
    s = json.dumps(B.compact_line(full), allow_nan=False)
    assert len(s) < 6000 and len(json.loads(s)["config"]["workload"]) == 240
    monkeypatch.setattr(B, "LINE_HARD_CAP", 2000)            # ... and if the line still came out too long, the extras go, the contract keys stay
    s = json.dumps(B.compact_line(full), allow_nan=False)
    back = json.loads(s)
    assert len(s) <= 2000 and "extra" not in back and all(k in back for k in CONTRACT) and "cpu_baseline" in back


def test_full_record_is_written_as_strict_json(tmp_path, monkeypatch):
    import numpy as np
    monkeypatch.setattr(B, "ROOT", str(tmp_path))
    full = _canned()
    full["x"] = {"nan": float("nan"), "arr": np.arange(3), "np": np.float64(1.5), "inf": math.inf}
    paths = B.write_extra(full)
    assert len(paths) == 2
    back = json.load(open(paths[0]))
    assert back["x"] == {"nan": None, "arr": [0, 1, 2], "np": 1.5, "inf": None} and back["c3_decimated"] == full["c3_decimated"]
