#!/bin/bash
# timing-only ablations of the Winograd forward main loop (results are wrong by construction)
for a in 0 11 12 13 14 15; do
  if [ $a = 0 ]; then unset CVK_LIB_PATH; else export CVK_LIB_PATH=$PWD/pytorch-camvid_amd/lib/libcvk_ab$a.so; fi
  echo "== ablate $a (11=no flush stores 12=no q/weight loads 13=no p loads 14=no LDS store 15=12+13+14)"
  python tools/bench_conv.py wino 2>&1 | grep -E "down2.1|ups3.conv|up1.0|down1.1|ups4|TOTAL"
done
