#!/bin/bash
# timing-only ablations of the conv main loop (results are wrong by construction)
for a in 0 4 5 6; do
  if [ $a = 0 ]; then unset CVK_LIB_PATH; else export CVK_LIB_PATH=$PWD/pytorch-camvid_amd/lib/libcvk_ab$a.so; fi
  echo "== ablate $a (4=no barrier/loads/lds-store 5=no epilogue 6=MFMA only)"
  python tools/bench_conv.py fwd 2>&1 | grep -E "down2.1|ups3.conv|up1.0|down1.1|ups4|TOTAL"
done
