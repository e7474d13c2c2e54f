#!/bin/bash
# The working tree's frame against the build of HEAD kept in _base/ (git worktree + its own liblav_amd.so), interleaved on ONE box
# (boxes of the pool differ by up to 8 %).    bash tools/ab_base.sh out.txt [rounds=2] [extra env for the NEW tree]
out=$1; n=${2:-2}; extra=$3
: > $out
for r in $(seq $n); do
  echo "== base" >> $out; (cd _base && timeout 300 python tools/frame_ab.py --variants all,chain --rounds 2 --steps 60 2>/dev/null | tail -1) >> $out
  echo "== new $extra" >> $out; env $extra timeout 300 python tools/frame_ab.py --variants all,chain --rounds 2 --steps 60 2>/dev/null | tail -1 >> $out
done
cat $out
