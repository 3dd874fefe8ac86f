#!/bin/bash
# the dispatcher's windows as whole records (MSGPU_NO_WIRE_COPY=1) against the wire form + host unpack (default), lean region
out=${1:-gpurun_out/wire_copy_ab}
mkdir -p $out
for v in records wire records wire; do
  if [ $v = records ]; then export MSGPU_NO_WIRE_COPY=1; else unset MSGPU_NO_WIRE_COPY; fi
  python tools/experiments/lean_wall.py $v 0 2>>$out/err.txt | tee -a $out/summary.txt || exit 1
done
for w in 2 4 5 6; do python tools/experiments/lean_wall.py wire_w$w $w 2>>$out/err.txt | tee -a $out/summary.txt || exit 1; done
