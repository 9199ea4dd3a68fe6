"""Aggregate FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, separate passes) per kernel class into a JSON for profiles/.
Reads the `counters_collection` view of the rocpd .db the profiler writes (or the older *counter_collection.csv files).
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE under-reports wide coalesced reads by 2x -> doubled.
Units: FETCH_SIZE/WRITE_SIZE are in KiB.      usage: pmc_traffic.py <fetch pass dir> <write pass dir>"""
import collections
import csv
import glob
import json
import sqlite3
import sys


def rows_of(d):
    dbs = glob.glob(d + "/**/*results.db", recursive=True)
    if dbs:
        for f in dbs:
            for name, cname, val in sqlite3.connect(f).execute("select kernel_name, counter_name, value from counters_collection"):
                yield name, cname, float(val)
        return
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            yield r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])


out = {}
parts = collections.defaultdict(dict)     # per GEMM kernel instantiation: where the class average comes from
for tag, d in (("fetch", sys.argv[1]), ("write", sys.argv[2])):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for kname, cname, val in rows_of(d):
        if cname not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        name = kname.split("(")[0].replace("void ", "")
        # GEMM classes as bench.py's roofline leg counts them: all bf16 instantiations incl. the persistent gemm_bf16_p44; all split-f16
        # ones incl. the ping-pong kernel and the fused MLP (one launch = both GEMMs of a ConvNeXt block)
        cls = "gemm_bf16_kernel" if "gemm_bf16" in name else ("gemm_h2_kernel" if ("gemm_h2" in name or "mlp_fused" in name) else name[:40])
        acc[cls][0] += val * 1024.0 * (2.0 if tag == "fetch" else 1.0)
        acc[cls][1] += 1
        if cls in ("gemm_bf16_kernel", "gemm_h2_kernel"):
            pk = parts[cls].setdefault(name[:72], {"fetch": [0.0, 0], "write": [0.0, 0]})
            pk[tag][0] += val * 1024.0 * (2.0 if tag == "fetch" else 1.0)
            pk[tag][1] += 1
    for k, (b, n) in acc.items():
        out.setdefault(k, {})[tag + "_bytes_per_launch"] = b / max(n, 1)
        out[k]["launches_" + tag] = n
for cls, pp in parts.items():
    out[cls]["parts"] = {k: {"launches": v["fetch"][1], "fetch_MB_per_launch": round(v["fetch"][0] / max(v["fetch"][1], 1) / 1e6, 1),
                             "write_MB_per_launch": round(v["write"][0] / max(v["write"][1], 1) / 1e6, 1)}
                         for k, v in sorted(pp.items(), key=lambda kv: -kv[1]["fetch"][0])}
print(json.dumps(out, indent=1))
