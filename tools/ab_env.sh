#!/bin/bash
# Interleaved A/B of one environment knob on one box: tools/ab_env.sh VAR A B [bench args...]  (two runs per setting, alternating)
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
VAR=$1; A=$2; B=$3; shift 3
for rep in 1 2; do
  for v in "$A" "$B"; do
    env "$VAR=$v" python3 bench.py --no-extra-configs --no-cpu-baseline --no-dp-overhead "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python3 -c "
import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('$VAR=$v', d['value'], d['ms_per_step'], 'hbm passes', d['roofline']['hbm_bound_kernels_ms_per_step'], 'conv', d['roofline']['all_conv_kernels']['ms_per_step'])"
  done
done
