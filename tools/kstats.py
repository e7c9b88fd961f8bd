"""Summarise a rocprofv3 --kernel-trace --stats run: per-kernel calls/step, average, ms/step."""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open((glob.glob(d + "/*/*kernel_stats.csv") + glob.glob(d + "/*kernel_stats.csv"))[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / steps / 1e6:.2f} ms/step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    n = r["Name"]
    n = n[n.find("k_"):][:52] if "k_" in n else n[:52]
    print(f"{n:52s} n/step={int(r['Calls']) / steps:6.1f} avg_us={float(r['AverageNs']) / 1e3:8.1f} ms/step={float(r['TotalDurationNs']) / steps / 1e6:7.3f}")
