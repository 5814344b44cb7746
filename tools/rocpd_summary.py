"""Reduce rocprofv3's rocpd SQLite output (ROCm 7 default) to the small text artefacts kept under profiles/.

    python tools/rocpd_summary.py stats  <kernel-trace results.db> <out.csv>
        per-kernel Calls / TotalDurationNs / AverageNs / Percentage / MinNs / MaxNs / StdDev (what --stats prints)
    python tools/rocpd_summary.py traffic <FETCH_SIZE results.db> <WRITE_SIZE results.db> <out.json>
        per-launch HBM traffic per kernel for bench.py's roofline.traffic (see tools/pmc_traffic.py for the gfx950
        corrections: KiB units; reads bracketed [x1, x2], the upper end is used)
    python tools/rocpd_summary.py kernel_counters <out.json> <PMC results.db> ...
        per-kernel mean of every collected counter (tools/pmc_mfma.sh: MfmaUtil, LDS bank conflicts)
"""
import collections
import csv
import json
import math
import sqlite3
import sys


def stats(db, out):
    c = sqlite3.connect(db)
    agg = collections.defaultdict(list)
    for name, dur in c.execute("select name, duration from kernels"):
        agg[name].append(dur)
    total = sum(sum(v) for v in agg.values())
    rows = []
    for k, v in agg.items():
        n, s = len(v), sum(v)
        mean = s / n
        sd = math.sqrt(sum((x - mean) ** 2 for x in v) / n)
        rows.append((k, n, s, mean, 100.0 * s / total, min(v), max(v), sd))
    rows.sort(key=lambda r: -r[2])
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 3), round(r[4], 4), r[5], r[6], round(r[7], 3)])
    for r in rows[:8]:
        print(f"{r[0][:90]:90s} n={r[1]:5d} avg {r[3] / 1e3:9.1f} us  {r[4]:5.1f} %")


def counter(db, name):
    c = sqlite3.connect(db)
    agg = collections.defaultdict(list)
    for k, v in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (name,)):
        agg[k].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def traffic(fdb, wdb, out):
    fetch, write = counter(fdb, "FETCH_SIZE"), counter(wdb, "WRITE_SIZE")
    res = {}
    for k in fetch:
        if k not in write:
            continue
        rd, wr = fetch[k][0] * 1024.0, write[k][0] * 1024.0
        short = k.split("(")[0].replace("void mcedm::", "").replace("mcedm::", "").strip()
        res[short] = {"launches": fetch[k][1], "read_bytes_x1": rd, "read_bytes_x2": 2 * rd, "write_bytes": wr,
                      "traffic_bytes": 2 * rd + wr}
    # stamped with the digest of the kernel sources the passes were taken on: bench.py refuses a stale file
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump({"csrc_digest": bench.csrc_digest(), "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB units; "
               "reads bracketed [x1, x2] (gfx950 counts 128-B requests at 64 B for wide streams), traffic_bytes = 2 * read + write",
               "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["launches"])[:8]:
        print(f"{k[:70]:70s} n={v['launches']:5d} read {v['read_bytes_x1'] / 1e6:8.1f}..{v['read_bytes_x2'] / 1e6:8.1f} MB"
              f"  write {v['write_bytes'] / 1e6:8.1f} MB")


def counters(dbs):
    """print the per-launch average of every counter in the given PMC result databases, per kernel"""
    for db in dbs:
        c = sqlite3.connect(db)
        agg = collections.defaultdict(list)
        for k, name, v in c.execute("select kernel_name, counter_name, value from counters_collection"):
            agg[(k.split("(")[0][-60:], name)].append(v)
        for (k, name), v in sorted(agg.items()):
            if len(v) >= 3 and "conv" in k:
                print(f"{k:60s} {name:32s} n={len(v):4d} avg {sum(v) / len(v):16.1f}")


def kernel_counters(out, dbs):
    """Per-kernel mean of every counter in the given PMC results.db files -> JSON {kernel: {counter: mean, 'launches': n}}."""
    res = {}
    for db in dbs:
        c = sqlite3.connect(db)
        agg = collections.defaultdict(list)
        for k, name, v in c.execute("select kernel_name, counter_name, value from counters_collection"):
            agg[(k.split("(")[0].replace("void mcedm::", "").replace("mcedm::", "").strip(), name)].append(v)
        for (k, name), v in agg.items():
            e = res.setdefault(k, {})
            e[name] = sum(v) / len(v)
            e["launches"] = len(v)
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("launches", 0))[:12]:
        print(f"{k[:70]:70s} " + "  ".join(f"{n} {v:.4g}" for n, v in e.items()))


if __name__ == "__main__":
    if sys.argv[1] == "counters":
        counters(sys.argv[2:])
    elif sys.argv[1] == "kernel_counters":
        kernel_counters(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
