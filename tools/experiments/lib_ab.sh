#!/bin/bash
# A/B of two builds of the library on one box: muchsalsa_amd/libmsgpu_base.so (built from the commit before) against the tree's
# libmsgpu.so; bench.py --kernels-only, alternating, three times each.   tools/experiments/lib_ab.sh <out dir>
out=${1:-gpurun_out/lib_ab}
mkdir -p $out
for v in base new base new base new; do
  if [ $v = new ]; then unset MSGPU_LIB; else export MSGPU_LIB=$PWD/muchsalsa_amd/libmsgpu_$v.so; fi
  python bench.py --kernels-only --steps 30 --warmup 3 > $out/b_$v.json 2> $out/b_$v.err || exit 1
  python -c "
import json; d=json.load(open('$out/b_$v.json')); print('$v', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['stage_ms'].items() if k!='note'})" | tee -a $out/summary.txt
done
