#!/bin/bash
# frame_ab.py of the working tree under environment variants, against _base/, interleaved on one box.
#   bash tools/ab_env.sh out.txt "A=1" "B=2 C=3" ...
out=$1; shift
: > $out
run() { echo "== $2" >> $out; (cd $1 && env $2 timeout 300 python tools/frame_ab.py --graphs --variants all,chain --rounds 2 --steps 60 2>/dev/null | tail -2) >> $out; }
run _base "B=1"
run . "X=0"
for v in "$@"; do run . "$v"; done
run _base "B=1"
run . "X=0"
cat $out
