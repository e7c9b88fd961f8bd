#!/bin/bash
# MFMA / wait-state counters of the conv kernels (one SQ pass + GRBM), per-kernel averages -> gpurun_out/pmc_mfma.json
# BENCH_ARGS="--precision bf16 --height 720 --width 960 --batch 4" selects the configs[3] workload; a second pass collects the
# LDS bank-conflict counters -> gpurun_out/pmc_lds.json
set -euo pipefail
: "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
BENCH_ARGS=${BENCH_ARGS:-}
rm -rf gpurun_out/pmc_mfma gpurun_out/pmc_lds
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-dp-overhead $BENCH_ARGS > /dev/null 2> gpurun_out/pmc_mfma.err
python3 - <<'PY'
import csv, glob, collections, json
f = glob.glob("gpurun_out/pmc_mfma/*/*counter_collection.csv")[0]
res = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_" not in n: continue
    n = n[n.index("k_"):]; n = n[:n.index("(")] if "(" in n else n
    res[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen[n]:
        seen[n].add(r["Dispatch_Id"]); res[n]["_t"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
out = {}
for n, e in sorted(res.items(), key=lambda kv: -kv[1]["_t"])[:12]:
    k = len(seen[n]); gui = e["GRBM_GUI_ACTIVE"]
    out[n] = {"launches": k, "avg_us": round(e["_t"] / k * 1e6, 1),
              # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
              "gui_active_cycles_per_launch_per_xcd": round(gui / 8 / k), "clock_GHz": round(gui / 8 / e["_t"] / 1e9, 3) if e["_t"] else None,
              "mfma_busy_frac_of_simd_cycles": round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024), 4) if gui else None,
              "wait_any_frac": round(e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"], 3) if e["SQ_WAVE_CYCLES"] else None,
              "wait_inst_any_frac": round(e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"], 3) if e["SQ_WAVE_CYCLES"] else None,
              "active_inst_frac": round(e["SQ_ACTIVE_INST_ANY"] / e["SQ_WAVE_CYCLES"], 3) if e["SQ_WAVE_CYCLES"] else None,
              "wait_inst_lds_frac": round(e["SQ_WAIT_INST_LDS"] / e["SQ_WAVE_CYCLES"], 3) if e["SQ_WAVE_CYCLES"] else None}
json.dump(out, open("gpurun_out/pmc_mfma.json", "w"), indent=1)
for n, v in out.items(): print(n[:46].ljust(46), v)
PY

rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 \
  --kernel-trace --output-format csv -d gpurun_out/pmc_lds -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-dp-overhead $BENCH_ARGS > /dev/null 2> gpurun_out/pmc_lds.err
python3 - <<'PY'
import csv, glob, collections, json
fs = glob.glob("gpurun_out/pmc_lds/*/*counter_collection.csv")
if fs:
    res = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        if "k_" not in n: continue
        n = n[n.index("k_"):]; n = n[:n.index("(")] if "(" in n else n
        res[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[n]:
            seen[n].add(r["Dispatch_Id"]); res[n]["_t"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    out = {}
    for n, e in sorted(res.items(), key=lambda kv: -kv[1]["_t"])[:12]:
        k = len(seen[n])
        out[n] = {"launches": k, "avg_us": round(e["_t"] / k * 1e6, 1),
                  "lds_bank_conflict_frac_of_lds_cycles": round(e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"], 4) if e["SQ_LDS_IDX_ACTIVE"] else None,
                  "lds_active_frac_of_wave_cycles": round(e["SQ_LDS_IDX_ACTIVE"] / e["SQ_WAVE_CYCLES"], 4) if e["SQ_WAVE_CYCLES"] else None,
                  "valu_insts_per_launch": round(e["SQ_INSTS_VALU"] / k), "lds_insts_per_launch": round(e["SQ_INSTS_LDS"] / k)}
    json.dump(out, open("gpurun_out/pmc_lds.json", "w"), indent=1)
    for n, v in out.items(): print(n[:46].ljust(46), v)
PY
