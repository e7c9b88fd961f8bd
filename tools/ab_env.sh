#!/bin/bash
# Interleaved A/B of one environment knob on one box: tools/ab_env.sh VAR A B [bench args...]  (two runs per setting, alternating)
set -euo pipefail
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
VAR=$1; A=$2; B=$3; shift 3
# The product libcvk.so reads no environment (cvk_knob() is its default there): C-side knobs only exist in the experiments build.
# Python-side knobs (read by engine.py / ddp.py) work with either library.
PY_KNOBS=$(grep -oh 'environ\(\.get\)\?[(\[]"CVK_[A-Z0-9_]*' pytorch-camvid_amd/*.py | grep -o 'CVK_[A-Z0-9_]*' | sort -u)
if ! grep -qx "$VAR" <<< "$PY_KNOBS"; then
  if [ -z "${CVK_LIB_PATH:-}" ]; then
    make -C pytorch-camvid_amd/csrc -j8 experiments > /dev/null
    export CVK_LIB_PATH="$GRAFT_REPO_ROOT/pytorch-camvid_amd/lib/libcvk_exp.so"
  fi
  if ! strings "$CVK_LIB_PATH" | grep -qx "$VAR"; then
    echo "ab_env.sh: $VAR is read neither by the Python host side nor by $CVK_LIB_PATH: both arms would run identical code" >&2
    exit 2
  fi
  echo "ab_env.sh: C-side knob, library $CVK_LIB_PATH"
fi
for rep in 1 2; do
  for v in "$A" "$B"; do
    env "$VAR=$v" python3 bench.py --no-extra-configs --no-cpu-baseline --no-dp-overhead "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    python3 -c "
import json;d=json.load(open('gpurun_out/ab_tmp.json'));print('$VAR=$v', d['value'], d['ms_per_step'], 'hbm passes', d['roofline']['hbm_bound_kernels_ms_per_step'], 'conv', d['roofline']['all_conv_kernels']['ms_per_step'])"
  done
done
