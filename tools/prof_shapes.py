"""Aggregate a UNI_PROF_DUMP file (engine per-launch records) by GEMM shape: calls per step, avg us, TFLOP/s, share."""
import sys, collections
rows = collections.defaultdict(lambda: [0, 0.0, 0.0])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tot = 0.0
for ln in open(sys.argv[1]):
    c, M, N, K, conv, us, work = ln.split()
    key = ("gemm" if c == "0" else "cls" + c, int(M), int(N), int(K), int(conv))
    r = rows[key]; r[0] += 1; r[1] += float(us); r[2] += float(work); tot += float(us)
print("%-6s %8s %6s %6s %4s %6s %9s %9s %6s" % ("class", "M", "N", "K", "conv", "n/step", "avg_us", "TF|GB/s", "share"))
for k, r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    rate = r[2] / (r[1] * 1e-6) / (1e12 if k[0] == "gemm" else 1e9)
    print("%-6s %8d %6d %6d %4d %6.1f %9.1f %9.1f %5.1f%%" % (k[0], k[1], k[2], k[3], k[4], r[0] / steps, r[1] / r[0], rate, 100 * r[1] / tot))
