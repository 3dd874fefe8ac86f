# Round 5: the chain stage's classes one after the other, a launch each (MSGPU_CHAIN_SERIAL=1), against the default (k_chain and
# k_chain_sub_all beside each other), same box, alternating.
for i in 1 2 3; do
  python bench.py --kernels-only --steps 30 --warmup 3 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print('beside %.4f chain kernels %.4f'%(d['ms_per_step'],d['stage_ms']['chain_kernel']))"
  MSGPU_CHAIN_SERIAL=1 python bench.py --kernels-only --steps 30 --warmup 3 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print('serial %.4f chain kernels %.4f'%(d['ms_per_step'],d['stage_ms']['chain_kernel']))"
done
