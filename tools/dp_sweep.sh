#!/bin/bash
# VERDICT r4 #4a: world-size-1 RCCL overhead of the DP step against NCCL_MAX_NCHANNELS (= CVK_DP_RESERVE_CUS, coupled by
# ddp.init_process_group), fp32 headline and bf16 configs[3].  One process per setting (RCCL reads its environment once).
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/dp_sweep
for ch in ${CHANNELS:-2 4 8 16}; do
  NCCL_MAX_NCHANNELS=$ch python3 bench.py --dp-overhead --no-extra-configs --no-cpu-baseline --steps 20 --warmup 5 \
      > gpurun_out/dp_sweep/fp32_ch$ch.json 2> gpurun_out/dp_sweep/fp32_ch$ch.err
  NCCL_MAX_NCHANNELS=$ch python3 bench.py --dp-overhead --no-extra-configs --no-cpu-baseline --steps 20 --warmup 5 \
      --precision bf16 --height 720 --width 960 --batch 4 > gpurun_out/dp_sweep/bf16_ch$ch.json 2> gpurun_out/dp_sweep/bf16_ch$ch.err
  echo "channels $ch done"
done
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/dp_sweep/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])["dp_overhead"]
        print(os.path.basename(f), {k: d.get(k) for k in ("plain_ms_per_step", "dp_world1_ms_per_step", "overhead_pct", "allreduce_exposed_ms", "persistent_workgroups", "graphed_dp_ms_per_step")})
    except Exception as e:
        print(os.path.basename(f), "failed", e)
PY
