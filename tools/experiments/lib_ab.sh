#!/bin/bash
# A/B of the tree's library against muchsalsa_amd/libmsgpu_$1.so (an experiment build: make BUILD=build_x OUT=../libmsgpu_x.so EXTRA=-D...)
v=$1; out=${2:-gpurun_out/lib_ab_$v}; mkdir -p $out
for w in base $v base $v base $v; do
  if [ $w = base ]; then unset MSGPU_LIB; else export MSGPU_LIB=$PWD/muchsalsa_amd/libmsgpu_$w.so; fi
  python bench.py --kernels-only --steps 30 --warmup 3 > $out/b_$w.json 2> $out/b_$w.err || exit 1
  python -c "
import json; d=json.load(open('$out/b_$w.json')); print('$w', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['stage_ms'].items() if k!='note'})" | tee -a $out/summary.txt
done
