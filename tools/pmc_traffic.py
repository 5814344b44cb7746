"""Reduce rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs of `python bench.py ...`) to
per-launch HBM traffic of each kernel and write profiles/<round>_traffic.json, which bench.py reads for
`roofline.traffic`.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB-like units of 1024 B; FETCH_SIZE counts
128-B requests at 64 B for wide coalesced streams (16 B/lane), i.e. reads are under-counted by up to 2x.  The conv
kernels stage dword-per-lane loads (uncalibrated width), so the read side is reported as a [x1, x2] bracket and the
bracket's upper end is used as `traffic` (conservative: more traffic = worse).

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1_traffic.json
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    if k not in write:
        continue
    rd, wr = fetch[k][0] * 1024.0, write[k][0] * 1024.0
    short = k.split("(")[0].replace("void mcedm::", "").replace("mcedm::", "").strip()
    out[short] = {"launches": fetch[k][1], "read_bytes_x1": rd, "read_bytes_x2": 2 * rd, "write_bytes": wr,
                  "traffic_bytes": 2 * rd + wr}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["launches"])[:8]:
    print(f"{k[:70]:70s} n={v['launches']:5d} read {v['read_bytes_x1']/1e6:8.1f}..{v['read_bytes_x2']/1e6:8.1f} MB  write {v['write_bytes']/1e6:8.1f} MB")
