set -e
mkdir -p gpurun_out/r3_04
python -m pytest tests/test_gpu_parity.py -x -q -k "wire or bench or shards" > gpurun_out/r3_04/pytest_wire.log 2>&1 || { tail -40 gpurun_out/r3_04/pytest_wire.log; exit 1; }
tail -3 gpurun_out/r3_04/pytest_wire.log
for i in 1 2; do
for f in whole wire; do
python bench.py --force-dist --kernels-only --scaling weak --steps 20 --warmup 3 --exchange-format $f > gpurun_out/r3_04/weak_w1_${f}_$i.json 2> gpurun_out/r3_04/weak_w1_${f}_$i.err
python -c "
import json,sys
l=json.loads([x for x in open('gpurun_out/r3_04/weak_w1_${f}_$i.json') if x.startswith('{')][-1])
print('$f', $i, l['ms_per_step'], l['exchange']['slab_bytes'], l['config'].get('merged_edge_list_consistent'))
"
done
done
python bench.py --kernels-only --steps 20 --warmup 3 > gpurun_out/r3_04/single.json 2> gpurun_out/r3_04/single.err
python -c "
import json
l=json.loads([x for x in open('gpurun_out/r3_04/single.json') if x.startswith('{')][-1])
print('single', l['ms_per_step'])
"
