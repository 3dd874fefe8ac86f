# Round 5, not kept: A/B of k_chain_sub<32> on the candidate stage's side stream (MSGPU_CHAIN_SUB32_BESIDE=1, a switch that
# existed only in the experiment build: msgpu_api.hip at the commit of this file's parent).  Result in profiles/r5_05/README.md.
for i in 1 2 3; do
  python bench.py --kernels-only --steps 30 --warmup 3 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print('N1 default      %.4f chain %.4f'%(d['ms_per_step'],d['stage_ms']['chain_total']))"
  MSGPU_CHAIN_SUB32_BESIDE=1 python bench.py --kernels-only --steps 30 --warmup 3 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print('N1 sub32 beside %.4f chain %.4f'%(d['ms_per_step'],d['stage_ms']['chain_total']))"
done
python tools/shard_projection.py cfg3 | python -c "import json,sys;d=json.load(sys.stdin);print('shards default     ', {k:(v['ms'],v['chain']) for k,v in d.items()})"
MSGPU_CHAIN_SUB32_BESIDE=1 python tools/shard_projection.py cfg3 | python -c "import json,sys;d=json.load(sys.stdin);print('shards sub32 beside', {k:(v['ms'],v['chain']) for k,v in d.items()})"
