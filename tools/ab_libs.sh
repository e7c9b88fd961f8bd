#!/bin/bash
# Interleaved A/B of several builds of the library on one box: tools/ab_libs.sh REPS LIB_A LIB_B ... -- [bench args]   (CVK_LIB_PATH selects the build)
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
REPS=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ $# -gt 0 ] && shift
for rep in $(seq 1 "$REPS"); do
  for v in "${LIBS[@]}"; do
    CVK_LIB_PATH="$GRAFT_REPO_ROOT/$v" python3 bench.py --no-extra-configs --no-cpu-baseline --no-dp-overhead "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python3 -c "
import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('$v', d['value'], d['ms_per_step'], 'conv', d['roofline']['all_conv_kernels']['ms_per_step'], 'passes', d['roofline']['hbm_bound_kernels_ms_per_step'], {k:v['ms_per_step'] for k,v in d['conv_kernels'].items() if 'wino4f' in k or 'wgradp' in k})"
  done
done
