import csv, collections, glob, sys
pat = sys.argv[2] if len(sys.argv) > 2 else "k_conv3x3_wino"
cc = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0])))
d = collections.OrderedDict()
for r in cc:
    if pat not in r["Kernel_Name"]:
        continue
    e = d.setdefault(r["Dispatch_Id"], {"grid": r["Grid_Size"], "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
seen = set()
for k, v in d.items():
    if v["grid"] in seen:
        continue
    seen.add(v["grid"])
    w = v["SQ_WAVE_CYCLES"]
    parts = " ".join(f"{n[3:].lower()}={v[n] / w:.3f}" for n in v if n.startswith("SQ_") and n != "SQ_WAVE_CYCLES")
    print(f"grid={v['grid']:>8s} t={v['t']:7.1f}us {parts}")
