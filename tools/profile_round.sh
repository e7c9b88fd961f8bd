#!/bin/bash
# One box, one call: the bench lines and rocprofv3 summaries a round's profiles/ entry is made of.
#   usage (on the GPU box): tools/profile_round.sh r04_a [fp32|bf16|all]     -> gpurun_out/<tag>_*
# Every pass writes into a directory that is removed first, every rocprofv3 exit status is checked, and a summary is only
# copied when the CSV it is made from exists (stale files must not pass for this round's profile).
set -euo pipefail
TAG=${1:?usage: profile_round.sh TAG [fp32|bf16|split2|all]}
LEG=${2:-all}
: "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run on the GPU box through gpurun)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p "$O"
BF16_ARGS="--precision bf16 --height 720 --width 960 --batch 4"

stats_pass() {   # $1 suffix, $2... bench args
    local sfx=$1; shift
    local dir="$O/${TAG}_stats_${sfx}"
    rm -rf "$dir"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$dir" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-dp-overhead "$@" \
        > "$O/${TAG}_bench_${sfx}_under_rocprof.json" 2> "$O/${TAG}_stats_${sfx}.err"
    local csv
    csv=$(ls "$dir"/*/*kernel_stats.csv | head -1)
    test -s "$csv"
    cp "$csv" "$O/${TAG}_kernel_stats_${sfx}.csv"
    python3 tools/kstats.py "$dir" 16 14
}

pmc_pass() {     # $1 suffix, $2 bench args (one string)
    local sfx=$1 args=$2
    rm -f "$O/pmc_traffic.json" "$O/pmc_mfma.json" "$O/pmc_lds.json"
    BENCH_ARGS="--no-extra-configs $args" tools/pmc_traffic.sh > "$O/${TAG}_pmc_traffic_${sfx}.txt" 2>&1
    test -s "$O/pmc_traffic.json" && cp "$O/pmc_traffic.json" "$O/${TAG}_pmc_hbm_traffic_${sfx}.json"
    echo "traffic done ($sfx)"
    BENCH_ARGS="--no-extra-configs $args" tools/pmc_mfma.sh > "$O/${TAG}_pmc_mfma_${sfx}.txt" 2>&1
    test -s "$O/pmc_mfma.json" && cp "$O/pmc_mfma.json" "$O/${TAG}_pmc_mfma_${sfx}.json"
    test -s "$O/pmc_lds.json" && cp "$O/pmc_lds.json" "$O/${TAG}_pmc_lds_${sfx}.json"
    echo "mfma done ($sfx)"
}

if [ "$LEG" = fp32 ] || [ "$LEG" = all ]; then
    python3 bench.py > "$O/${TAG}_bench_fp32.json" 2> "$O/${TAG}_bench_fp32.err"
    echo "bench done: $(python3 -c "import json;d=json.load(open('$O/${TAG}_bench_fp32.json'));print(d['value'], d['ms_per_step'])")"
    stats_pass fp32
    pmc_pass fp32 ""
    python3 bench.py --model segnet --no-cpu-baseline > "$O/${TAG}_bench_segnet.json" 2> "$O/${TAG}_bench_segnet.err"
fi
if [ "$LEG" = bf16 ] || [ "$LEG" = all ]; then
    python3 bench.py $BF16_ARGS --no-cpu-baseline > "$O/${TAG}_bench_bf16_720.json" 2> "$O/${TAG}_bench_bf16_720.err"
    echo "bf16 bench done: $(python3 -c "import json;d=json.load(open('$O/${TAG}_bench_bf16_720.json'));print(d['value'], d['ms_per_step'])")"
    stats_pass bf16 $BF16_ARGS
    pmc_pass bf16 "$BF16_ARGS"
fi
if [ "$LEG" = split2 ]; then      # the opt-in fp16 split-operand mode on the headline workload (never part of "all": it is not the product default)
    python3 bench.py --w2d-split=2 --no-cpu-baseline --no-extra-configs > "$O/${TAG}_bench_fp32_split2.json" 2> "$O/${TAG}_bench_fp32_split2.err"
    echo "split2 bench done: $(python3 -c "import json;d=json.load(open('$O/${TAG}_bench_fp32_split2.json'));print(d['value'], d['ms_per_step'])")"
    stats_pass fp32_split2 --w2d-split=2
    pmc_pass fp32_split2 "--w2d-split=2"
fi
echo "all done"
