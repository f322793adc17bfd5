#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/<dir>/.../*.csv) into the small summaries committed under profiles/.

    python profiles/summarize.py r01 binary gpurun_out/r01_trace gpurun_out/r01_fetch gpurun_out/r01_write [gpurun_out/r01_sq]

* <tag>_<name>_kernel_stats.csv : the --kernel-trace --stats table (per-kernel calls / average ns)
* <tag>_<name>_pmc.json         : HBM traffic of the dominant kernel from SEPARATE --pmc passes
      FETCH_SIZE, WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
      (MI355X_MICROARCH.md, HBM), so read bytes = 2 * FETCH_SIZE * 1024 -- confirmed here on a known byte
      count: the decode kernel reads every channel value exactly once (N*F*4 bytes).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def one(pattern):
    g = sorted(glob.glob(pattern, recursive=True))
    return g[-1] if g else None


def _db(d):
    return one(os.path.join(d, "**", "*_results.db"))


def counters(d, match):
    """Average per launch of every counter of the kernels matching `match` (csv output, or rocprofv3's sqlite database)."""
    p = one(os.path.join(d, "**", "*counter_collection.csv"))
    agg = collections.defaultdict(list)
    kname = None
    if p:
        for r in csv.DictReader(open(p)):
            if match in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                kname = r["Kernel_Name"]
    elif _db(d):
        import sqlite3
        cur = sqlite3.connect(_db(d)).cursor()
        # one row per (dispatch, counter, hardware instance): sum the instances of a dispatch, average the dispatches
        per = collections.defaultdict(float)
        for kn, cn, disp, val in cur.execute("select kernel_name, counter_name, dispatch_id, value from counters_collection"):
            if match in kn:
                per[(cn, disp)] += float(val)
                kname = kn
        for (cn, _), v in per.items():
            agg[cn].append(v)
    return {k: sum(v) / len(v) for k, v in agg.items()}, kname


def kernel_stats(trace, dst):
    """Copy (csv) or rebuild (sqlite) the --kernel-trace --stats table; returns its rows."""
    st = one(os.path.join(trace, "**", "*kernel_stats.csv"))
    if st:
        shutil.copy(st, dst)
    else:
        import sqlite3
        cur = sqlite3.connect(_db(trace)).cursor()
        rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                                "group by name order by 3 desc"))
        tot = sum(r[2] for r in rows) or 1
        with open(dst, "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100.0 * r[2] / tot, 2), r[4], r[5]])
    return list(csv.DictReader(open(dst)))


def main():
    tag, name, trace, fetch, write = sys.argv[1:6]
    sq = sys.argv[6] if len(sys.argv) > 6 else None
    match = sys.argv[7] if len(sys.argv) > 7 else "k_qc"
    here = os.path.dirname(os.path.abspath(__file__))
    rows = [r for r in kernel_stats(trace, os.path.join(here, "%s_%s_kernel_stats.csv" % (tag, name))) if match in r["Name"]]
    dom = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    f, kn = counters(fetch, match)
    w, _ = counters(write, match)
    out = {"kernel_symbol": dom["Name"], "calls": int(dom["Calls"]), "avg_ms": float(dom["AverageNs"]) / 1e6,
           "FETCH_SIZE_KiB": f.get("FETCH_SIZE"), "WRITE_SIZE_KiB": w.get("WRITE_SIZE"),
           "read_bytes_per_launch": 2 * 1024 * f["FETCH_SIZE"] if "FETCH_SIZE" in f else None,
           "write_bytes_per_launch": 1024 * w["WRITE_SIZE"] if "WRITE_SIZE" in w else None}
    if out["read_bytes_per_launch"] and out["write_bytes_per_launch"]:
        out["hbm_bytes_per_launch"] = out["read_bytes_per_launch"] + out["write_bytes_per_launch"]
    if sq:
        out["sq"] = {}
        for d in sq.split(","):  # several SQ passes (the counters of one pass are limited)
            s, _ = counters(d, match)
            out["sq"].update(s)
    kfile = os.path.join(here, "..", "gpurun_out", "%s_%s_bench.json" % (tag, name))
    if os.path.exists(kfile):
        try:
            j = json.loads(open(kfile).read().strip().splitlines()[-1])
            out["kernel"] = j["roofline"].get("kernel") or j["config"].get("kernel")
            out["bench_kernel_ms"] = j["roofline"]["kernel_ms"]
        except Exception:
            pass
    json.dump(out, open(os.path.join(here, "%s_%s_pmc.json" % (tag, name)), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
