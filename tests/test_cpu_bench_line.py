"""The driver parses the LAST stdout line of bench.py: it has to stay a small, flat record (the r05 line had grown to
20 KB and came back unparsed).  compact_line() is run on every full record kept under profiles/ and on a worst-case synthetic one."""
import glob
import json
import os

from khronos_amd import bench_line as bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIMIT = 6144
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _full_records():
    recs = []
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench*.json"))):
        try:
            d = json.load(open(p))
        except ValueError:
            continue
        if isinstance(d, dict) and "metric" in d and "config" in d and isinstance(d["config"], dict):
            recs.append((p, d))
    return recs


def test_compact_line_of_every_recorded_run_is_small_and_complete():
    recs = _full_records()
    assert recs, "no recorded bench lines under profiles/"
    for p, d in recs:
        line = bench.compact_line(d)
        s = json.dumps(line)
        assert len(s) < LIMIT, (p, len(s))
        for k in REQUIRED:
            if k in d:
                assert k in line, (p, k)
        assert line["value"] == d["value"] and line["ms_per_step"] == d["ms_per_step"]
        if "roofline" in d:
            for k in ("bound", "achieved", "peak", "unit", "frac"):
                assert line["roofline"][k] == d["roofline"][k]
            assert "traffic" in line["roofline"]
        if "cpu_baseline" in d:
            for k in ("value", "unit", "cores", "kind", "sample"):
                assert k in line["cpu_baseline"]
        json.loads(s)


def test_compact_line_is_bounded_for_a_bloated_record():
    recs = _full_records()
    d = json.loads(json.dumps(max(recs, key=lambda r: len(json.dumps(r[1])))[1]))
    d["config"]["workload"] = "x" * 5000
    d["config"]["parallelism"] = "y" * 5000
    d.setdefault("roofline", {})["traffic_source"] = "z" * 5000
    d.setdefault("cpu_baseline", {"value": 1.0, "unit": "frames/s", "cores": 1, "kind": "port"})["sample"] = "s" * 5000
    d["streams"] = {"stream_%d" % i: {"value": 1234.5678, "note": "n" * 4000} for i in range(12)}
    assert len(json.dumps(bench.compact_line(d))) < LIMIT
