#!/bin/bash
# A/B and ablation runs of the ping-pong bf16 conv kernel (each setting in its own process: the knobs are read once)
cd "$GRAFT_REPO_ROOT"
CVK_BF16P=0 python3 tools/bench_bf16p.py old
CVK_BF16P_MF=32 python3 tools/bench_bf16p.py pp32
CVK_BF16P_MF=16 python3 tools/bench_bf16p.py pp16
CVK_BF16P_MF=32 python3 tools/bench_bf16p.py pp32
CVK_BF16P_MF=16 python3 tools/bench_bf16p.py pp16
for d in ${DBGS:-}; do CVK_BF16P_MF=32 CVK_BF16P_DBG=$d python3 tools/bench_bf16p.py dbg$d; done
