"""Per-kernel averages of every counter in a rocprofv3 --pmc pass (rocpd .db `counters_collection` view).
usage: pmc_kernel.py <pass dir> [kernel-name substring]"""
import collections, glob, sqlite3, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*results.db", recursive=True):
    for name, cname, val in sqlite3.connect(f).execute("select kernel_name, counter_name, value from counters_collection"):
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        a = acc[name.split("(")[0].replace("void ", "")[:60]][cname]
        a[0] += float(val); a[1] += 1
for k, d in acc.items():
    print(k)
    for c, (v, n) in sorted(d.items()):
        print("   %-28s %14.1f  (avg of %d)" % (c, v / n, n))
