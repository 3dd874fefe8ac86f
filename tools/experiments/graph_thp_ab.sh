# Round 5: the host graph stage with and without the huge-page advice for its large tables (MSGPU_GRAPH_THP=0), one box,
# six repetitions each; the kernel's THP mode first.  Kept: profiles/r5_08/README.md.
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>&1
echo "--- THP advice on"
MSGPU_GRAPH_DEBUG=1 timeout -k 10 300 python tools/graph_only.py > gpurun_out/graph_only4.log 2>&1; grep -E "^create" gpurun_out/graph_only4.log
echo "--- THP advice off"
MSGPU_GRAPH_THP=0 MSGPU_GRAPH_DEBUG=1 timeout -k 10 300 python tools/graph_only.py > gpurun_out/graph_only5.log 2>&1; grep -E "^create" gpurun_out/graph_only5.log
