"""Aggregate FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, separate passes) per kernel class into a JSON for profiles/.
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE under-reports wide coalesced reads by 2x -> doubled.
Units: FETCH_SIZE/WRITE_SIZE are in KiB."""
import csv, glob, json, sys, collections
out = {}
for tag, d in (("fetch", sys.argv[1]), ("write", sys.argv[2])):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            cls = "gemm_bf16_kernel" if "gemm_bf16" in name else name[:40]   # incl. the persistent gemm_bf16_p44 / _pipe kernels
            acc[cls][0] += float(r["Counter_Value"]) * 1024.0 * (2.0 if tag == "fetch" else 1.0)
            acc[cls][1] += 1
    for k, (b, n) in acc.items():
        out.setdefault(k, {})[tag + "_bytes_per_launch"] = b / max(n, 1)
        out[k]["launches_" + tag] = n
print(json.dumps(out, indent=1))
