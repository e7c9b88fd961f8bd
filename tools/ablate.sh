#!/bin/bash
# timing-only ablations of the conv main loop (results are wrong by construction)
for a in 0 1 2 3 4; do
  if [ $a = 0 ]; then unset CVK_LIB_PATH; else export CVK_LIB_PATH=$PWD/pytorch-camvid_amd/lib/libcvk_ab$a.so; fi
  echo "== ablate $a (1=no barrier 2=no global loads 3=no lds store 4=all three)"
  python tools/bench_conv.py fwd 2>&1 | grep -E "down2.1|ups3.conv|up1.0|down1.1|TOTAL"
done
