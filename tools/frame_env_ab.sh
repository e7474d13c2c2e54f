#!/bin/bash
# frame_ab.py under environment variants, one process each (the knobs are read at start-up), default first and last.
#   bash tools/frame_env_ab.sh out.txt "A=1" "B=2 C=3" ...
out=$1; shift
: > $out
run() { echo "== ${1:-default}" >> $out; env $1 timeout 300 python tools/frame_ab.py --graphs --variants all,chain --rounds 2 --steps 60 2>/dev/null | tail -2 >> $out; }
run "X=0"
for v in "$@"; do run "$v"; done
run "X=0"
cat $out
