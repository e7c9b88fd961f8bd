#!/bin/bash
# Interleaved A/B of two builds of the library on one box: tools/ab_lib.sh LIB_A LIB_B [bench args...]   (CVK_LIB_PATH selects the build)
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for v in "$A" "$B"; do
    CVK_LIB_PATH="$GRAFT_REPO_ROOT/$v" python3 bench.py --no-extra-configs --no-cpu-baseline --no-dp-overhead "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python3 -c "
import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('$v', d['value'], d['ms_per_step'], 'conv', d['roofline']['all_conv_kernels']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['conv_kernels'].items() if 'bf16q' in k or 'bf16h' in k})"
  done
done
