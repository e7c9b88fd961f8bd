#!/bin/bash
# Interleaved A/B/C on one box: the exact-fp32 default against the opt-in split-operand GEMMs (bench.py --w2d-split=3 | 2), headline workload
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for v in "" "--w2d-split=3" "--w2d-split=2"; do
    python3 bench.py --no-extra-configs --no-cpu-baseline --no-dp-overhead --steps 30 $v > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python3 -c "
import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('${v:-default}', d['value'], d['ms_per_step'], 'loss', d['config']['loss'], 'conv', d['roofline']['all_conv_kernels']['ms_per_step'], 'passes', d['roofline']['hbm_bound_kernels_ms_per_step'], {k:v['ms_per_step'] for k,v in d['conv_kernels'].items() if 'gemm' in k})"
  done
done
