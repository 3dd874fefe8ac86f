// consensus_base.cpp -- updateConsensusBase (libms/src/kernel/ap.cpp:205-229) on piece lists.
//
// In the reference the growing contig of visitOrdered (ap.cpp:231-349) is a std::string plus two borders; every
// update copies the whole string again (O(L^2) over a path).  Here the contig is a list of copy
// pieces: an update cuts the uncovered head or tail out of the new segment's pieces (a sub-range of a piece is another
// piece) and prepends / appends them.  No base is touched until the gather kernel runs once over the finished layout.
#include <algorithm>
#include <deque>
#include <new>
#include <vector>

#include "msgpu.h"

struct msgpu_consensus {
  std::deque<msgpu_copy> pieces; // dst_off unused while building; assigned by msgpu_consensus_pieces
  bool     has = false;
  int32_t  lo = 0, hi = 0;
  uint64_t length = 0;
};

namespace {

// pieces of the sub-string [a, b) of a segment given as pieces laid out back to back
void cut(const msgpu_copy *seg, uint32_t n, uint64_t a, uint64_t b, std::vector<msgpu_copy> &out) {
  uint64_t pos = 0;
  for (uint32_t i = 0; i < n && pos < b; ++i) {
    const uint64_t L = seg[i].len, s = std::max(a, pos), e = std::min(b, pos + L);
    if (s < e) {
      msgpu_copy p = seg[i];
      const uint64_t from = s - pos, to = e - pos; // range inside this (oriented) piece
      // a reverse-complemented piece reads its source backwards: oriented [from, to) = source [L - to, L - from)
      p.src_off = seg[i].src_off + ((p.flags & MSGPU_COPY_REVCOMP) ? L - to : from);
      p.len     = static_cast<uint32_t>(to - from);
      p.dst_off = 0;
      out.push_back(p);
    }
    pos += L;
  }
}

} // namespace

extern "C" {

msgpu_consensus *msgpu_consensus_new(void) { return new (std::nothrow) msgpu_consensus(); }
void             msgpu_consensus_free(msgpu_consensus *c) { delete c; }

msgpu_consensus *msgpu_consensus_clone(const msgpu_consensus *c) { return c ? new (std::nothrow) msgpu_consensus(*c) : nullptr; }

// updateConsensusBase(oldSequence, oldBorders, newSequence, newBorders)
int msgpu_consensus_update(msgpu_consensus *c, const msgpu_copy *seg, uint32_t n, int32_t new_lo, int32_t new_hi) {
  if (!c || (n && !seg)) return MSGPU_E_ARG;
  try {
    uint64_t size = 0;
    for (uint32_t i = 0; i < n; ++i) size += seg[i].len;
    std::vector<msgpu_copy> part;
    if (!c->has) { // :209-211
      cut(seg, n, 0, size, part);
      c->pieces.assign(part.begin(), part.end());
      c->has    = true;
      c->lo     = new_lo;
      c->hi     = new_hi;
      c->length = size;
      return MSGPU_OK;
    }
    if (new_lo < c->lo) { // :214-218: strSlice(newSequence, 0, oldBorders.first - newBorders.first) + old
      uint64_t       len   = 0;
      const uint64_t start = msgpu_str_slice(size, 0, c->lo - new_lo, &len);
      cut(seg, n, start, start + len, part);
      c->pieces.insert(c->pieces.begin(), part.begin(), part.end());
      c->length += len;
    } else if (new_hi > c->hi) { // :219-223: old + strSlice(newSequence, -(new.second - old.second), size)
      uint64_t       len   = 0;
      const uint64_t start = msgpu_str_slice(size, -(new_hi - c->hi), static_cast<int32_t>(size), &len);
      cut(seg, n, start, start + len, part);
      c->pieces.insert(c->pieces.end(), part.begin(), part.end());
      c->length += len;
    } // else :224-226 unchanged
    c->lo = std::min(c->lo, new_lo); // :228
    c->hi = std::max(c->hi, new_hi);
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
  return MSGPU_OK;
}

int msgpu_consensus_borders(const msgpu_consensus *c, int32_t *lo, int32_t *hi, uint64_t *length) {
  if (!c) return MSGPU_E_ARG;
  if (lo) *lo = c->lo;
  if (hi) *hi = c->hi;
  if (length) *length = c->length;
  return MSGPU_OK;
}

// the contig as pieces with dst_off = base, base + len0, ...; returns the number of pieces (call with cap 0 to size)
size_t msgpu_consensus_pieces(const msgpu_consensus *c, uint64_t base, msgpu_copy *out, size_t cap) {
  if (!c) return 0;
  if (out) {
    uint64_t off = base;
    size_t   i   = 0;
    for (const msgpu_copy &p : c->pieces) {
      if (i >= cap) break;
      out[i]         = p;
      out[i].dst_off = off;
      off += p.len;
      ++i;
    }
  }
  return c->pieces.size();
}

} // extern "C"
