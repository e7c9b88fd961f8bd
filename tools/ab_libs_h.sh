#!/bin/bash
# like ab_libs_k.sh, printing memory-bound passes: tools/ab_libs_h.sh REPS PATTERN LIB_A LIB_B ... -- [bench args]
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
REPS=$1; PAT=$2; shift 2
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ $# -gt 0 ] && shift
for rep in $(seq 1 "$REPS"); do
  for v in "${LIBS[@]}"; do
    CVK_LIB_PATH="$GRAFT_REPO_ROOT/$v" python3 bench.py --no-extra-configs --no-cpu-baseline --no-dp-overhead "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python3 -c "
import json,re;d=json.load(open('gpurun_out/ab_tmp.json'));print('$v', d['value'], d['ms_per_step'], 'passes', d['roofline']['hbm_bound_kernels_ms_per_step'], {k:(v['ms_per_step'],v['frac_of_8TBps']) for k,v in d['hbm_kernels'].items() if re.search('$PAT', k)})"
  done
done
