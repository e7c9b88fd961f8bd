#!/bin/bash
# HBM traffic of the conv kernels from PMC counters (separate passes, as MI355X_MICROARCH.md §HBM prescribes).
set -euo pipefail
: "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
BENCH_ARGS=${BENCH_ARGS:-}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-dp-overhead $BENCH_ARGS > /dev/null 2> gpurun_out/pmc_$c.err
  ls gpurun_out/pmc_$c/*/*counter_collection.csv > /dev/null
done
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "t": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/*/*counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_" not in n: continue
        n = n[n.index("k_"):]
        n = n[:n.index("(")] if "(" in n else n
        e = res[n]
        e[c] += float(r["Counter_Value"])
        if c == "FETCH_SIZE":
            e["n"] += 1; e["t"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
out = {}
for n, e in sorted(res.items(), key=lambda kv: -kv[1]["t"]):
    if e["n"] == 0: continue
    # counters are in KiB; gfx950 FETCH_SIZE reports exactly half of a wide coalesced read stream -> doubled
    rd = 2.0 * e["FETCH_SIZE"] * 1024 / e["n"]; wr = e["WRITE_SIZE"] * 1024 / e["n"]
    out[n] = {"launches": e["n"], "avg_us": round(e["t"] / e["n"] * 1e6, 1), "read_MB_per_launch": round(rd / 1e6, 1),
              "write_MB_per_launch": round(wr / 1e6, 1), "hbm_GBps": round((rd + wr) / (e["t"] / e["n"]) / 1e9, 1)}
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
for n, v in list(out.items())[:14]: print(n[:44].ljust(44), v)
PY
