"""The one stdout line of bench.py that the driver parses (no torch import here: the CPU test of the line's size must not pull a second
HIP runtime into the test process)."""


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(out):
    """The one stdout line the driver parses: headline + roofline + cpu_baseline + what the timed window contained."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                       "dtype", "data", "mvoxel_updates_per_s", "speedup_vs_cpu", "emulation"))
    cfg = out.get("config", {})
    line["config"] = {"workload": cfg.get("workload", "")[:700], "preset": cfg.get("preset"),
                      "parallelism": (cfg.get("parallelism") or "")[:160]}
    if "roofline" in out:
        r = _pick(out["roofline"], ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "launches",
                                    "algorithmic_bytes_per_launch", "traffic"))
        r["traffic_source"] = (out["roofline"].get("traffic_source") or "")[:120] or None
        line["roofline"] = r
    if "cpu_baseline" in out:
        c = _pick(out["cpu_baseline"], ("value", "unit", "cores", "kind"))
        c["sample"] = out["cpu_baseline"].get("sample", "")[:260]
        c["best"] = _pick(out["cpu_baseline"].get("best", {}), ("threads", "frames_per_s"))
        line["cpu_baseline"] = c
    tr = out.get("timed_region", {})
    line["timed_region"] = _pick(tr, ("steps_ms", "drain_and_join_ms", "frames_with_dynamic_clusters", "outputs", "objects_extracted"))
    if "seed_wait" in out:
        line["seed_wait"] = _pick(out["seed_wait"], ("waits", "late_over_500us", "max_us"))
    if out.get("latency_ms_per_frame"):
        line["latency_ms_per_frame_mean"] = out["latency_ms_per_frame"]["mean"]
    if "kernel_rooflines" in out:
        line["kernel_frac"] = {k["kernel"]: round(k["frac"], 3) for k in out["kernel_rooflines"]["kernels"]}
    if "streams" in out:
        line["streams_frames_per_s"] = {k: (round(v["value"], 1) if "value" in v else "error") for k, v in out["streams"].items()}
    if "rccl" in out:
        line["rccl"] = {"rccl_ranks": out["rccl"]["rccl_ranks"],
                        "ms_per_call": {k: round(v["ms_per_call"], 4) for k, v in out["rccl"].get("collectives", {}).items()}}
    line["detail"] = "bench_detail.json (next to bench.py) and stderr carry the full record"
    return line
