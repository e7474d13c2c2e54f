#!/bin/bash
# One pass over everything profiles/ is refreshed from (run on the GPU box through gpurun; outputs under gpurun_out/<tag>/).
#   bash tools/round_profiles.sh r02_b
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 240 python $R/bench.py > $O/bench_line.json 2> $O/bench.err < /dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 30 --warmup 16 --no-cpu-baseline > /dev/null 2>&1 < /dev/null
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_bench.csv
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    n=$(echo $c | tr ' ' '_')
    timeout 120 rocprofv3 --pmc $c -d /tmp/pmc_$n -- python $R/tools/pmc_pillar.py > /dev/null 2>&1 < /dev/null
    db=$(find /tmp/pmc_$n -name "*.db" | head -1)
    [ -n "$db" ] && python $R/tools/pmc_report.py "$db" > $O/pmc_$n.txt 2>/dev/null
done
LAV_PILLAR_TRACE=1 timeout 90 python $R/tools/pillar_probe.py > $O/pillar_trace.txt 2>&1 < /dev/null
timeout 200 python $R/bench.py --mode train_full --steps 6 --warmup 3 2>/dev/null < /dev/null | tail -1 > $O/train_full.json
timeout 200 python $R/bench.py --mode train_full --steps 6 --warmup 3 --log-every 1 2>/dev/null < /dev/null | tail -1 > $O/train_full_log1.json
timeout 200 python $R/bench.py --mode train_bev --steps 10 --warmup 3 2>/dev/null < /dev/null | tail -1 > $O/train_bev.json
ls -la $O
cut -c1-400 $O/bench_line.json
