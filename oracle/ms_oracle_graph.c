/*
 * ms_oracle_graph.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT) for the first graph clean-up step after the
 * overlap path: findContractionEdges (src/main.cpp:183-190, 416-463) and the kernel it calls, sanityCheck
 * (libms/src/kernel/sc.cpp:29-90), restated over the flat result tables of ms_oracle_overlap.
 *
 * PARITY UNPINNED (no reference fixture, reference unbuildable here -- see ms_oracle.h).  Deterministic in the
 * reference: the neighbour loop runs over a std::map (ascending id) and only ANDs, the order loop takes the first hit.
 */
#include <stdlib.h>
#include <string.h>

#include "ms_oracle.h"

typedef struct {
  const ms_edge  *edges;
  size_t          n_edges;
  const ms_order *orders;
} graph_view;

/* Graph::getEdge / hasEdge on the undirected graph: edges are sorted by (v1, v2), v1 < v2 */
static long find_edge(const graph_view *g, uint32_t a, uint32_t b) {
  uint32_t lo_v = a < b ? a : b, hi_v = a < b ? b : a;
  size_t   lo = 0, hi = g->n_edges;
  while (lo < hi) {
    size_t mid = lo + (hi - lo) / 2;
    const ms_edge *e = &g->edges[mid];
    if (e->v1 < lo_v || (e->v1 == lo_v && e->v2 < hi_v)) lo = mid + 1;
    else hi = mid;
  }
  if (lo < g->n_edges && g->edges[lo].v1 == lo_v && g->edges[lo].v2 == hi_v) return (long)lo;
  return -1;
}

#define DIR(o) (((o)->flags & MS_ORD_DIR) != 0)
#define CONTAINED(o) (((o)->flags & MS_ORD_CONTAINED) != 0)

/* sanityCheck(graph, subnode, node, target, order, wiggleRoom), sc.cpp:29-90 */
static int sanity_check(const graph_view *g, uint32_t subnode, uint32_t node, uint32_t target, const ms_order *order,
                        uint64_t wiggle) {
  long e_on = find_edge(g, node, target), e_for = find_edge(g, subnode, target);
  const ms_edge *on_e = &g->edges[e_on], *for_e = &g->edges[e_for];
  for (uint32_t i = 0; i < on_e->order_cnt; ++i) {
    const ms_order *on = &g->orders[on_e->order_off + i];
    for (uint32_t j = 0; j < for_e->order_cnt; ++j) {
      const ms_order *fr = &g->orders[for_e->order_off + j];
      int sane = ((DIR(order) == DIR(on)) ? 1 : 0) == (DIR(fr) ? 1 : 0); /* Toggle * Toggle is XNOR, :36 */
      if (CONTAINED(fr) && CONTAINED(on)) { /* :41-43 */
        sane &= (fr->start == target || fr->end == target) && on->start == target;
      } else if (CONTAINED(fr) && !CONTAINED(on)) { /* :44-70 */
        if (fr->end != target) {
          int l1 = 0, l2 = 0, l3 = 0;
          if ((!DIR(on) && ((node == on->base && on->end == target) || (node != on->base && on->end == target))) ||
              (DIR(on) && on->end == target)) {
            if (!DIR(order)) l2 = 1;
          } else {
            l1 = 1;
            l3 = 1;
            if (DIR(order)) l2 = 1;
          }
          if (!DIR(order) && order->base != order->end) l1 = !l1;
          if (!DIR(fr) && fr->base != fr->end) l2 = !l2;
          double d1 = l1 ? order->left_offset : order->right_offset;
          double d2 = l2 ? fr->left_offset : fr->right_offset;
          double d3 = l3 ? on->left_offset : on->right_offset;
          sane &= (d1 + d2 + d3) < (double)wiggle;
        }
      } else if (!CONTAINED(fr) && CONTAINED(on)) { /* :71-72 */
        sane &= on->start == target;
      } else { /* :73-82 */
        int d1 = fr->start == target, d2 = on->start == target;
        if (!DIR(fr) && fr->base == target) d1 = !d1;
        if (!DIR(on) && on->base == target) d2 = !d2;
        if (!DIR(order)) d1 = !d1;
        sane &= d1 == d2;
      }
      if (sane) return 1;
    }
  }
  return 0;
}

/* out[e] = index in the order table of the contraction order of edge e, or -1 (src/main.cpp:416-463) */
int ms_oracle_find_contraction_edges(const ms_tables *t, uint64_t wiggle, int64_t *out) {
  graph_view g = {t->edges, t->n_edges, t->orders};
  /* Graph::getNeighbors: adjacency in both directions */
  uint32_t  V = t->n_reads;
  uint64_t *off = calloc((size_t)V + 1, sizeof(uint64_t));
  uint32_t *adj = malloc((t->n_edges ? 2 * t->n_edges : 1) * sizeof(uint32_t));
  if (!off || !adj) {
    free(off);
    free(adj);
    return -4;
  }
  for (size_t e = 0; e < t->n_edges; ++e) {
    off[t->edges[e].v1 + 1]++;
    off[t->edges[e].v2 + 1]++;
  }
  for (uint32_t v = 0; v < V; ++v) off[v + 1] += off[v];
  uint64_t *cur = malloc(((size_t)V + 1) * sizeof(uint64_t));
  if (!cur) {
    free(off);
    free(adj);
    return -4;
  }
  memcpy(cur, off, ((size_t)V + 1) * sizeof(uint64_t));
  for (size_t e = 0; e < t->n_edges; ++e) {
    adj[cur[t->edges[e].v1]++] = (uint32_t)e;
    adj[cur[t->edges[e].v2]++] = (uint32_t)e;
  }
  for (size_t e = 0; e < t->n_edges; ++e) {
    const ms_edge *ed = &t->edges[e];
    out[e] = -1;
    for (uint32_t k = 0; k < ed->order_cnt && out[e] < 0; ++k) {
      const ms_order *order = &t->orders[ed->order_off + k];
      if (!(CONTAINED(order) && (order->flags & MS_ORD_PRIMARY))) continue;
      int sane = 1;
      /* the reference walks the neighbours in ascending id; the result is an AND, so any order gives the same answer */
      for (uint64_t a = off[order->start]; a < off[order->start + 1] && sane; ++a) {
        const ms_edge *sub = &t->edges[adj[a]];
        uint32_t target = sub->v1 == order->start ? sub->v2 : sub->v1;
        if (target == order->end || sub->shadow) continue;
        if (find_edge(&g, order->end, target) < 0) {
          sane = 0;
          break;
        }
        sane &= sanity_check(&g, order->start, order->end, target, order, wiggle);
      }
      if (sane) out[e] = (int64_t)(ed->order_off + k);
    }
  }
  free(off);
  free(adj);
  free(cur);
  return 0;
}
