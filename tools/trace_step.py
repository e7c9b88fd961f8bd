"""Per-launch durations of the conv kernels in ONE step of a rocprofv3 --kernel-trace run (the last complete step).
usage: python3 tools/trace_step.py <dir with *_kernel_trace.csv> [name substrings...]"""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv"))[0]
subs = sys.argv[2:] or ["k_conv_bf16", "k_wgrad_bf16r"]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_ce_fwd" in r["Kernel_Name"]]
lo, hi = (marks[-2], marks[-1]) if len(marks) >= 2 else (0, len(rows))
tot = {}
for r in rows[lo:hi]:
    n = r["Kernel_Name"]
    if not any(s in n for s in subs):
        continue
    short = n[n.find("k_"):][:44]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot[short] = tot.get(short, 0.0) + d
    print(f"{short:44s} wgs={int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])):6d} {d:8.1f} us")
print({k: round(v / 1e3, 3) for k, v in tot.items()})
