#!/bin/bash
# MFMA / wait-state counters of the conv kernels (one SQ pass + GRBM), per-kernel averages -> gpurun_out/pmc_mfma.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile > /dev/null 2> gpurun_out/pmc_mfma.err
python - <<'PY'
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
