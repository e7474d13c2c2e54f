#!/bin/bash
# One pass over everything profiles/ is refreshed from (run on the GPU box through gpurun; outputs under gpurun_out/<tag>/).
#   bash tools/round_profiles.sh r03 [skip-train-trace]
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# the bench line (frame + roofline blocks + forced-N frames + the two training lines + cpu baselines)
timeout 900 python $R/bench.py > $O/bench_line.json 2> $O/bench.err < /dev/null
# per-kernel times of the same frame loop
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --steps 30 --warmup 16 --no-cpu-baseline --no-train --no-variants > /dev/null 2>&1 < /dev/null
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_bench.csv
# PMC passes, one counter set per run (never together with a trace)
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    n=$(echo $c | tr ' ' '_')
    timeout 120 rocprofv3 --pmc $c -d /tmp/pmc_$n -- python $R/tools/pmc_pillar.py > /dev/null 2>&1 < /dev/null
    db=$(find /tmp/pmc_$n -name "*.db" | head -1)
    [ -n "$db" ] && python $R/tools/pmc_report.py "$db" > $O/pmc_$n.txt 2>/dev/null
done
python $R/tools/pmc_pillar_json.py $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES_SQ_BUSY_CYCLES.txt > $O/pmc_pillar.json 2> $O/pmc_pillar_json.err
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/pmc_conv -- python $R/tools/pmc_conv.py > /dev/null 2>&1 < /dev/null
db=$(find /tmp/pmc_conv -name "*.db" | head -1); [ -n "$db" ] && python $R/tools/pmc_report.py "$db" > $O/pmc_conv.txt 2>/dev/null
LAV_PILLAR_TRACE=1 timeout 90 python $R/tools/pillar_probe.py > $O/pillar_trace.txt 2>&1 < /dev/null
if [ -z "$2" ]; then
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trainprof -- python $R/bench.py --mode train_full --steps 6 --warmup 3 --log-every 1 > $O/train_prof_run.log 2>&1 < /dev/null
    python $R/tools/trace_top.py /tmp/trainprof --window 0.6 --steps 4 --top 45 > $O/train_full_kernel_top.txt 2>&1
fi
ls -la $O
cut -c1-400 $O/bench_line.json
