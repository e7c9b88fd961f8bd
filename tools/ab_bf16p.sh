#!/bin/bash
# A/B and ablation runs of the ping-pong bf16 conv kernel (each setting in its own process: the knobs are read once)
cd "$GRAFT_REPO_ROOT"
CVK_BF16P=0 python3 tools/bench_bf16p.py old
python3 tools/bench_bf16p.py pp
CVK_BF16P_VAR=1 python3 tools/bench_bf16p.py pp_noprio
for d in ${DBGS:-}; do CVK_BF16P_DBG=$d python3 tools/bench_bf16p.py dbg$d; done
