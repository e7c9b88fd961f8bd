#!/bin/bash
# PMC counters of ONE kernel of a small driver script (separate passes): HBM traffic, matrix-pipe / wait states, LDS, L2 hit rate.
# usage (on the GPU box): tools/pmc_kernel.sh <kernel-name-substring> <out.json> python3 tools/run_bf16p.py 128 128 360 480
set -euo pipefail
: "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"
KSUB=$1; OUTJ=$2; shift 2
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_kernel
rm -rf "$OUT"
mkdir -p "$OUT"
i=0
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES" \
         "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/p$i" -- "$@" > /dev/null 2> "$OUT/p$i.err"
  ls "$OUT/p$i"/*/*counter_collection.csv > /dev/null
done
KSUB="$KSUB" OUTJ="$OUTJ" python3 - <<'PY'
import csv, glob, collections, json, os
ksub = os.environ["KSUB"]
res = collections.defaultdict(float); n = 0; t = 0.0; seen = set(); npass = collections.Counter()
for f in sorted(glob.glob("gpurun_out/pmc_kernel/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if ksub not in r["Kernel_Name"]: continue
        res[r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if "/p1/" in f and key not in seen:
            seen.add(key); n += 1; t += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
k = max(n, 1)
out = {"kernel": ksub, "launches": n, "avg_us": round(t / k * 1e6, 1),
       "read_MB_per_launch": round(2.0 * res["FETCH_SIZE"] * 1024 / k / 1e6, 1), "write_MB_per_launch": round(res["WRITE_SIZE"] * 1024 / k / 1e6, 1)}
gui = res["GRBM_GUI_ACTIVE"]; wc = res["SQ_WAVE_CYCLES"] / 2 if res["SQ_WAVE_CYCLES"] else 0     # collected in two passes
if gui:
    out["mfma_busy_frac_of_simd_cycles"] = round(res["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024), 4)
    out["gui_active_cycles_per_launch_per_xcd"] = round(gui / 8 / k)
for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
    if wc: out[c.lower() + "_frac_of_wave_cycles"] = round(res[c] / wc, 4)
for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU"):
    out[c.lower() + "_per_launch"] = round(res[c] / k)
if res["TCC_HIT_sum"] + res["TCC_MISS_sum"]:
    out["l2_hit_rate"] = round(res["TCC_HIT_sum"] / (res["TCC_HIT_sum"] + res["TCC_MISS_sum"]), 4)
json.dump(out, open(os.environ["OUTJ"], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
