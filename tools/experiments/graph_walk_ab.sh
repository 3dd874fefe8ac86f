# Round 5, not kept: the walk of getDirectedGraph with and without a prefetch of the neighbours' arc rows, same box, alternating
# (MSGPU_GRAPH_WALK_NOPF existed only in the experiment build: graph_stage.cpp at the commit that added this file).  Result:
# profiles/r5_08/graph_walk_prefetch_ab.txt -- no difference.
for k in 1 2 3; do
  MSGPU_GRAPH_DEBUG=1 python tools/graph_only.py 2>&1 | grep -E "c2 .*dg: walk|^create" | awk '{print "pf   ", $0}' | tail -4
  MSGPU_GRAPH_WALK_NOPF=1 MSGPU_GRAPH_DEBUG=1 python tools/graph_only.py 2>&1 | grep -E "c2 .*dg: walk|^create" | awk '{print "nopf ", $0}' | tail -4
done
