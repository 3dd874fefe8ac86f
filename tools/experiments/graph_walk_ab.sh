# the walk of getDirectedGraph with and without the prefetch of the neighbours' arc rows, same box, alternating
for k in 1 2 3; do
  MSGPU_GRAPH_DEBUG=1 python tools/graph_only.py 2>&1 | grep -E "c2 .*dg: walk|^create" | awk '{print "pf   ", $0}' | tail -4
  MSGPU_GRAPH_WALK_NOPF=1 MSGPU_GRAPH_DEBUG=1 python tools/graph_only.py 2>&1 | grep -E "c2 .*dg: walk|^create" | awk '{print "nopf ", $0}' | tail -4
done
