#!/bin/bash
# train_full / train_bev steps of the working tree against _base/, interleaved on one box.   bash tools/ab_train.sh out.txt [mode] "ENV=..." ...
out=$1; mode=${2:-train_full}; shift; shift
: > $out
run() { echo "== $1 $2" >> $out; (cd $1 && env $2 timeout 400 python bench.py --mode $mode --steps ${STEPS:-8} --warmup 3 --log-every 1 $EXTRA 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])") >> $out; }
run _base "X=0"
run . "X=0"
for v in "$@"; do run . "$v"; done
run _base "X=0"
run . "X=0"
cat $out
