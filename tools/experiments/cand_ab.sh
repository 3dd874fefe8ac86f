#!/bin/bash
mkdir -p gpurun_out/r4_14
for v in base c128 c64 base c128 c64; do
  if [ $v = base ]; then unset MSGPU_LIB; else export MSGPU_LIB=$PWD/muchsalsa_amd/libmsgpu_$v.so; fi
  python bench.py --kernels-only --steps 30 --warmup 3 > gpurun_out/r4_14/b_$v.json 2> gpurun_out/r4_14/b_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/r4_14/b_$v.json')); print('$v', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['stage_ms'].items() if k!='note'})"
done
