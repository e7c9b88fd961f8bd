#!/bin/bash
# One box, one call: the bench lines and rocprofv3 summaries a round's profiles/ entry is made of.
#   usage (on the GPU box): tools/profile_round.sh r03_a      -> gpurun_out/<tag>_*
TAG=${1:-rXX}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
python3 bench.py > $O/${TAG}_bench_fp32.json 2> $O/${TAG}_bench_fp32.err
echo "bench done: $(python3 -c "import json;d=json.load(open('$O/${TAG}_bench_fp32.json'));print(d['value'], d['ms_per_step'])")"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $O/${TAG}_bench_fp32_under_rocprof.json 2> $O/${TAG}_stats.err
cp $(ls $O/${TAG}_stats/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_fp32.csv
python3 tools/kstats.py $O/${TAG}_stats 16 12
BENCH_ARGS="--no-extra-configs" tools/pmc_traffic.sh > $O/${TAG}_pmc_traffic.txt 2>&1 && cp $O/pmc_traffic.json $O/${TAG}_pmc_hbm_traffic.json
echo "traffic done"
BENCH_ARGS="--no-extra-configs" tools/pmc_mfma.sh > $O/${TAG}_pmc_mfma.txt 2>&1 && cp $O/pmc_mfma.json $O/${TAG}_pmc_mfma.json && cp $O/pmc_lds.json $O/${TAG}_pmc_lds.json
echo "mfma done"
python3 bench.py --model segnet --no-cpu-baseline > $O/${TAG}_bench_segnet.json 2> /dev/null
python3 bench.py --precision bf16 --height 720 --width 960 --batch 4 --no-cpu-baseline > $O/${TAG}_bench_bf16_720.json 2> /dev/null
echo "all done"
