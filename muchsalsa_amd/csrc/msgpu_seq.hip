// msgpu_seq.hip -- sequence store in HBM + batched slice / reverse-complement / stitch kernel.
//
// Device half of the reference's "consensus" stage: assemblePath (libms/src/kernel/ap.cpp)
// decides WHERE every piece of sequence goes (layout, host); the bytes themselves are produced by
//   SequenceAccessor::get{Nanopore,Illumina}Sequence      (SequenceAccessor.cpp:54-69,129-139: whole-record fetch)
//   strSlice / getReverseComplement / get*Sequence(l,r,d) (SequenceUtils.cpp:27-85)
//   updateConsensusBase                                   (ap.cpp:205-229: prepend / append the uncovered part)
// which here become one kernel over a list of copy pieces {source range, orientation, destination}.  Sequences stay
// resident in HBM one byte per base (exact: N, lower case and IUPAC codes survive; only A,C,G,T are complemented).
// HBM-bound byte work: 16 B per lane, aligned 16-B stores, unaligned source handled with v_alignbyte.
#include <atomic>
#include <hip/hip_runtime.h>

#include <memory>
#include <system_error>
#include <thread>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>

#include "asm_internal.h"
#include "host_pool.h"
#include "msgpu.h"
#include "msgpu_internal.h"

namespace msgpu {

constexpr uint32_t GCHUNK = 1024; // output bytes per wavefront: 64 lanes x 16 B (a workgroup takes 4 consecutive chunks)

struct __attribute__((packed, aligned(4))) U4A4 { // 16 bytes that are only dword-aligned
  uint32_t x, y, z, w;
};

// 0x80 in every byte of w that equals c
__device__ __forceinline__ uint32_t byte_eq(uint32_t w, uint32_t c) {
  const uint32_t z = w ^ (c * 0x01010101u);
  const uint32_t t = (z & 0x7f7f7f7fu) + 0x7f7f7f7fu;
  return ~(t | z | 0x7f7f7f7fu);
}
// getReverseComplement's switch (SequenceUtils.cpp:46-57) on four bytes at once: A<->T, C<->G, all else unchanged
__device__ __forceinline__ uint32_t complement4(uint32_t w) {
  const uint32_t at = byte_eq(w, 'A') | byte_eq(w, 'T');
  const uint32_t cg = byte_eq(w, 'C') | byte_eq(w, 'G');
  return w ^ (((at >> 7) * 0x15u) | ((cg >> 7) * 0x04u));
}

// the output of a gather is written once and not read again by the kernel: a non-temporal 16-byte store keeps it from
// displacing the sequence words in L2 (0.37 -> 0.27 ms on the cfg3 consensus leg)
__device__ __forceinline__ void store16_streaming(uint8_t *q, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(v4u{w0, w1, w2, w3}, reinterpret_cast<v4u *>(q));
}

// 16 bytes starting at an arbitrary byte address (the store is padded, so reading a few bytes around is safe)
__device__ __forceinline__ void load16_unaligned(const uint8_t *p, uint32_t out[4]) {
  const uintptr_t a  = reinterpret_cast<uintptr_t>(p);
  const uint32_t  sh = static_cast<uint32_t>(a & 3);
  const uint8_t  *b  = reinterpret_cast<const uint8_t *>(a & ~static_cast<uintptr_t>(3));
  const U4A4      v  = *reinterpret_cast<const U4A4 *>(b);
  const uint32_t  w4 = *reinterpret_cast<const uint32_t *>(b + 16);
  out[0]             = __builtin_amdgcn_alignbyte(v.y, v.x, sh);
  out[1]             = __builtin_amdgcn_alignbyte(v.z, v.y, sh);
  out[2]             = __builtin_amdgcn_alignbyte(v.w, v.z, sh);
  out[3]             = __builtin_amdgcn_alignbyte(w4, v.w, sh);
}

#ifndef MSGPU_GUNROLL
#define MSGPU_GUNROLL 2
#endif
constexpr int GUNROLL = MSGPU_GUNROLL; // chunks per wavefront: all chunks' loads are in flight before the first is consumed

__global__ __launch_bounds__(256) void k_gather(const msgpu_copy *pieces, const uint2 *chunk_map, uint64_t n_chunks,
                                                const uint8_t *base0, const uint8_t *base1, uint8_t *out) {
  // 1 KiB output chunks; chunk_map[chunk] = (piece, index of the chunk inside the piece), precomputed by the host when
  // the plan was made (a binary search here would cost ~18 dependent loads per KiB)
  const uint64_t c0   = (static_cast<uint64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * GUNROLL;
  const int      lane = threadIdx.x & 63;
  uint64_t       A[GUNROLL], qlo[GUNROLL], qhi[GUNROLL];
  uint32_t       r[GUNROLL][4];
  bool           rev[GUNROLL];
#pragma unroll
  for (int u = 0; u < GUNROLL; ++u) {
    const uint64_t chunk = c0 + u;
    qlo[u] = qhi[u] = A[u] = 0;
    rev[u]                 = false;
    if (chunk >= n_chunks) continue;
    const uint2      cm = chunk_map[chunk];
    const msgpu_copy pc = pieces[cm.x];
    const uint64_t   d0 = pc.dst_off, d1 = pc.dst_off + pc.len;
    A[u]   = (d0 & ~15ull) + static_cast<uint64_t>(cm.y) * GCHUNK + lane * 16ull; // 16-B aligned output address
    qlo[u] = A[u] > d0 ? A[u] : d0;
    qhi[u] = (A[u] + 16 < d1) ? A[u] + 16 : d1;
    if (qlo[u] >= qhi[u]) continue;
    const uint8_t *src = (pc.flags & MSGPU_COPY_ILLUMINA) ? base1 : base0;
    rev[u]             = (pc.flags & MSGPU_COPY_REVCOMP) != 0;
    // forward: out[q] = src[src_off + (q - d0)]   (A < d0 only for the head lane: reads into the padding)
    // reverse: out[q] = complement(src[src_off + len - 1 - (q - d0)]): the 16 source bytes ending at that address
    const uint8_t *p = rev[u] ? src + pc.src_off + pc.len - 1 - (A[u] - d0) - 15 : src + pc.src_off + (A[u] - d0);
    load16_unaligned(p, r[u]);
  }
#pragma unroll
  for (int u = 0; u < GUNROLL; ++u) {
    if (qlo[u] >= qhi[u]) continue;
    uint32_t w[4];
    if (!rev[u]) {
      w[0] = r[u][0];
      w[1] = r[u][1];
      w[2] = r[u][2];
      w[3] = r[u][3];
    } else {
      w[0] = complement4(__builtin_bswap32(r[u][3]));
      w[1] = complement4(__builtin_bswap32(r[u][2]));
      w[2] = complement4(__builtin_bswap32(r[u][1]));
      w[3] = complement4(__builtin_bswap32(r[u][0]));
    }
    if (qlo[u] == A[u] && qhi[u] == A[u] + 16) {
      store16_streaming(out + A[u], w[0], w[1], w[2], w[3]);
    } else { // head / tail of a piece
      for (uint64_t q = qlo[u]; q < qhi[u]; ++q) {
        const uint32_t k = static_cast<uint32_t>(q - A[u]);
        out[q]           = static_cast<uint8_t>(w[k >> 2] >> (8 * (k & 3)));
      }
    }
  }
}

// ---- 2-bit store ---------------------------------------------------------------------------------
// After msgpu_seq_pack a store holds 2 bits per base (A C G T = 0 1 2 3, base i in word i >> 4 at bits 2 (i & 15)) plus
// a sorted list of the positions whose byte is none of ACGT (N, lower case, IUPAC codes; they carry code 0 in the
// packed words and are written back verbatim by k_patch_exceptions).  The gather then reads a quarter of the bytes:
// 1.25 B of HBM traffic per base instead of 2.  Complement is code ^ 3; reversing a lane's 16 bases is a bit reverse.

// the 16 codes starting at base index b (may be negative by < 16 at the head of a piece: the words are padded)
__device__ __forceinline__ uint32_t codes16(const uint32_t *words, int64_t b) {
  const int64_t  w  = b >> 4; // floor
  const uint32_t sh = static_cast<uint32_t>(b & 15) * 2;
  return __builtin_amdgcn_alignbit(words[w + 1], words[w], sh);
}
// 4 codes (low 8 bits of c) -> 4 ASCII bytes
__device__ __forceinline__ uint32_t decode4(uint32_t c) {
  const uint32_t sel = (c | (c << 6) | (c << 12) | (c << 18)) & 0x03030303u;
  return __builtin_amdgcn_perm(0u, 0x54474341u /* 'A','C','G','T' */, sel);
}

// One wavefront per 4 KiB of one piece's output (SCHUNK): the piece record and everything derived from it are wave-uniform
// (scalar registers), a lane's share is four 16-byte stores 1 KiB apart, all four packed loads are issued before the
// first decode, and the per-lane arithmetic is 32-bit (offsets relative to the piece).
#ifndef MSGPU_SSUB
#define MSGPU_SSUB 4
#endif
constexpr int      SSUB   = MSGPU_SSUB; // 1-KiB sub-chunks per wavefront
constexpr uint32_t SCHUNK = 1024u * SSUB;

__global__ __launch_bounds__(256) void k_gather_packed(const msgpu_copy *pieces, const uint2 *super_map,
                                                       uint64_t n_super, const uint32_t *words0,
                                                       const uint32_t *words1, uint8_t *out) {
  const uint64_t sc = static_cast<uint64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (sc >= n_super) return;
  const int      lane = threadIdx.x & 63;
  const uint2    cm   = super_map[sc];
  const uint32_t pi   = __builtin_amdgcn_readfirstlane(cm.x);
  const uint32_t sub  = __builtin_amdgcn_readfirstlane(cm.y);
  const msgpu_copy *pp = pieces + pi; // uniform address: scalar loads
  const uint64_t src_off = pp->src_off, d0 = pp->dst_off;
  const uint32_t len = pp->len, flags = pp->flags;
  const bool     rev = (flags & MSGPU_COPY_REVCOMP) != 0;
  const uint32_t *words = (flags & MSGPU_COPY_ILLUMINA) ? words1 : words0;
  // output window of this wave: aligned start of the piece + sub * 4 KiB; rel = offset of a byte from d0
  uint8_t *const wout   = out + (d0 & ~15ull) + static_cast<uint64_t>(sub) * SCHUNK;
  const int32_t  rel0   = static_cast<int32_t>(sub * SCHUNK) - static_cast<int32_t>(d0 & 15) + lane * 16;
  const uint64_t s_word = src_off >> 4; // base index = 16 * s_word + s_lo + (offset inside the piece)
  const int32_t  s_lo   = static_cast<int32_t>(src_off & 15);
  uint32_t       code[SSUB];
#pragma unroll
  for (int k = 0; k < SSUB; ++k) {
    const int32_t rel = rel0 + k * 1024;
    code[k]           = 0;
    if (rel + 16 <= 0 || rel >= static_cast<int32_t>(len)) continue; // (len < 2^31 is checked when the plan is made)
    // first of the 16 bases this lane needs, as an offset from base 16 * s_word
    const int32_t t  = s_lo + (rev ? static_cast<int32_t>(len) - 1 - rel - 15 : rel);
    const uint32_t *w = words + (static_cast<int64_t>(s_word) + (t >> 4));
    code[k]           = __builtin_amdgcn_alignbit(w[1], w[0], static_cast<uint32_t>(t & 15) * 2);
  }
#pragma unroll
  for (int k = 0; k < SSUB; ++k) {
    const int32_t rel = rel0 + k * 1024;
    if (rel + 16 <= 0 || rel >= static_cast<int32_t>(len)) continue;
    uint32_t c = code[k];
    if (rev) { // reverse the 16 two-bit fields and complement them
      c = __builtin_bitreverse32(c);
      c = ((c >> 1) & 0x55555555u) | ((c & 0x55555555u) << 1);
      c = ~c;
    }
    const uint32_t w0 = decode4(c & 0xffu), w1 = decode4((c >> 8) & 0xffu), w2 = decode4((c >> 16) & 0xffu),
                   w3 = decode4(c >> 24);
    uint8_t *const q = wout + (lane * 16 + k * 1024);
    if (rel >= 0 && rel + 16 <= static_cast<int32_t>(len)) {
      store16_streaming(q, w0, w1, w2, w3);
    } else { // head / tail of the piece
      const uint32_t w[4] = {w0, w1, w2, w3};
      const int      lo = rel < 0 ? -rel : 0, hi = static_cast<int32_t>(len) - rel < 16 ? static_cast<int32_t>(len) - rel : 16;
      for (int j = lo; j < hi; ++j) q[j] = static_cast<uint8_t>(w[j >> 2] >> (8 * (j & 3)));
    }
  }
}

// 16 bases -> one word; counts (pass 0) or records (pass 1) the bytes that are none of ACGT
__global__ __launch_bounds__(256) void k_pack_bases(const uint8_t *bases, uint64_t n_bases, uint32_t *words,
                                                    unsigned long long *n_exc, uint64_t *exc_pos, uint8_t *exc_byte,
                                                    int record) {
  const uint64_t w = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (w * 16 >= n_bases) return;
  uint32_t word = 0;
  for (int k = 0; k < 16; ++k) {
    const uint64_t i = w * 16 + k;
    if (i >= n_bases) break;
    const uint8_t b = bases[i];
    uint32_t      c = 0;
    if (b == 'C') c = 1;
    else if (b == 'G') c = 2;
    else if (b == 'T') c = 3;
    else if (b != 'A') {
      const unsigned long long slot = atomicAdd(n_exc, 1ull);
      if (record) {
        exc_pos[slot]  = i;
        exc_byte[slot] = b;
      }
    }
    word |= c << (2 * k);
  }
  if (!record) words[w] = word;
}

// bytes that are not ACGT: written verbatim at their (possibly mirrored) place, SequenceUtils.cpp:46-57 leaves them alone
__global__ __launch_bounds__(256) void k_patch_exceptions(const msgpu_copy *pieces, uint32_t n_pieces, const uint64_t *pos0,
                                                          const uint8_t *byte0, uint64_t n0, const uint64_t *pos1,
                                                          const uint8_t *byte1, uint64_t n1, uint8_t *out) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pieces) return;
  const msgpu_copy pc  = pieces[p];
  const bool       il  = (pc.flags & MSGPU_COPY_ILLUMINA) != 0;
  const uint64_t  *pos = il ? pos1 : pos0;
  const uint8_t   *byt = il ? byte1 : byte0;
  const uint64_t   n   = il ? n1 : n0;
  uint64_t         lo = 0, hi = n; // first exception at or after src_off
  while (lo < hi) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (pos[mid] < pc.src_off) lo = mid + 1;
    else hi = mid;
  }
  for (; lo < n && pos[lo] < pc.src_off + pc.len; ++lo) {
    const uint64_t r = pos[lo] - pc.src_off;
    out[pc.dst_off + ((pc.flags & MSGPU_COPY_REVCOMP) ? pc.len - 1 - r : r)] = byt[lo];
  }
}

// FASTA wrapping (limitLength, ap.cpp:61-76): text record = header, the bases in lines of 60 separated by '\n', '\n'.
// One wavefront per 1 KiB of text, 16 output bytes per lane, one aligned 16-B store; chunk_map as in k_gather.
constexpr uint32_t FLINE = 60;

__global__ __launch_bounds__(256) void k_fasta_format(const msgpu_fasta_record *recs, const uint2 *chunk_map,
                                                      uint64_t n_chunks, const uint8_t *raw, const uint8_t *headers,
                                                      uint8_t *text) {
  const uint64_t chunk = static_cast<uint64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (chunk >= n_chunks) return;
  const int                lane = threadIdx.x & 63;
  const uint2              cm   = chunk_map[chunk];
  const msgpu_fasta_record rc   = recs[cm.x];
  const uint64_t body = rc.len + (rc.len ? (rc.len - 1) / FLINE : 0); // bases + inner newlines
  const uint64_t t0 = rc.text_off, t1 = t0 + rc.header_len + body + 1;
  const uint64_t A  = (t0 & ~15ull) + static_cast<uint64_t>(cm.y) * GCHUNK + lane * 16ull;
  const uint64_t qlo = A > t0 ? A : t0, qhi = (A + 16 < t1) ? A + 16 : t1;
  if (qlo >= qhi) return;
  uint32_t w[4] = {0, 0, 0, 0};
  uint64_t j    = qlo - t0; // position inside the record's text
  // position inside the body -> (line, column); advanced incrementally (at most one line break per 16 bytes)
  uint64_t b = j > rc.header_len ? j - rc.header_len : 0;
  uint64_t line = b / (FLINE + 1);
  uint32_t col  = static_cast<uint32_t>(b - line * (FLINE + 1));
  const uint8_t *src = raw + rc.raw_off;
  for (uint64_t q = qlo; q < qhi; ++q, ++j) {
    uint8_t c;
    if (j < rc.header_len) {
      c = headers[rc.header_off + j];
    } else {
      if (j - rc.header_len == body || col == FLINE) {
        c = '\n';
        col = 0;
        ++line;
      } else {
        c = src[line * FLINE + col];
        ++col;
      }
    }
    const uint32_t k = static_cast<uint32_t>(q - A);
    w[k >> 2] |= static_cast<uint32_t>(c) << (8 * (k & 3));
  }
  if (qlo == A && qhi == A + 16) {
    *reinterpret_cast<uint4 *>(text + A) = make_uint4(w[0], w[1], w[2], w[3]);
  } else {
    for (uint64_t q = qlo; q < qhi; ++q) {
      const uint32_t k = static_cast<uint32_t>(q - A);
      text[q]          = static_cast<uint8_t>(w[k >> 2] >> (8 * (k & 3)));
    }
  }
}

} // namespace msgpu

using namespace msgpu;

// ---- host side -----------------------------------------------------------------------------------------------------

namespace {
constexpr size_t SEQ_PAD  = 64; // bytes of zero padding in front of and behind the bases of a store
constexpr size_t PACK_PAD = 16; // words of zero padding in front of and behind the packed words

struct SeqStore {
  void                 *d_buf = nullptr; // SEQ_PAD + bases + SEQ_PAD (freed by msgpu_seq_pack)
  uint64_t              n_bases = 0;
  hipStream_t           stream = nullptr; // uploads and the 2-bit conversion of THIS store (the two stores may be filled from two host threads)
  void                 *ring = nullptr;   // page-locked slots of msgpu_seq_parse_upload (kept for the next file)
  size_t                ring_bytes = 0;
  // 2-bit form (msgpu_seq_pack): 16 words of zero padding, (n_bases + 15) / 16 words, 16 words of padding
  void    *d_words = nullptr, *d_exc_pos = nullptr, *d_exc_byte = nullptr;
  uint64_t n_exc = 0;
  bool     packed = false;
  void     drop_packed() {
    if (d_words) (void)hipFree(d_words);
    if (d_exc_pos) (void)hipFree(d_exc_pos);
    if (d_exc_byte) (void)hipFree(d_exc_byte);
    d_words = d_exc_pos = d_exc_byte = nullptr;
    n_exc                            = 0;
    packed                           = false;
  }
  std::vector<uint64_t> off;   // by id: offset of the sequence inside the bases (~0 = no such id)
  std::vector<uint64_t> len;
};
} // namespace

struct msgpu_seqctx {
  int         device = 0;
  hipStream_t stream = nullptr;
  SeqStore    st[2];
  char        err[256] = {0};
  // device scratch that only grows (msgpu_assembly_finish / msgpu_fasta_format would otherwise pay several
  // hipMalloc + hipFree round trips, each a device-wide synchronisation, per call)
  struct Scratch {
    void  *p   = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
      if (bytes <= cap) return hipSuccess;
      if (p) (void)hipFree(p);
      p   = nullptr;
      cap = 0;
      const size_t want = bytes + bytes / 4 + 4096;
      hipError_t   e    = hipMalloc(&p, want);
      if (e == hipSuccess) cap = want;
      return e;
    }
  } scr_text, scr_recs, scr_map, scr_hdr, scr_pairs, scr_dist; // (the last two: msgpu_edit_distance's pair table and result)
};

struct msgpu_gather_plan {
  void    *d_pieces = nullptr, *d_chunk_map = nullptr; // chunk_map: uint2 {piece, chunk inside the piece} per 1 KiB chunk
  void    *d_super_map = nullptr;                      // the same per 4 KiB (k_gather_packed)
  uint32_t n = 0;
  uint64_t n_chunks = 0, n_super = 0, out_bytes = 0, bases = 0;
};

namespace {
int sfail(msgpu_seqctx *c, int code, const char *what, hipError_t e) {
  snprintf(c->err, sizeof(c->err), "%s: %s", what, hipGetErrorString(e));
  return code;
}
#define SHIP(c, expr)                                                                                                  \
  do {                                                                                                                 \
    hipError_t _e = (expr);                                                                                            \
    if (_e != hipSuccess) return sfail((c), _e == hipErrorOutOfMemory ? MSGPU_E_NOMEM : MSGPU_E_HIP, #expr, _e);        \
  } while (0)
} // namespace

extern "C" {

int msgpu_seq_create(int device, msgpu_seqctx **out) {
  if (!out) return MSGPU_E_ARG;
  *out = nullptr;
  if (device == -1) { // layout-only context: slice arithmetic and segment composers work, nothing can be gathered
    auto *c = new (std::nothrow) msgpu_seqctx();
    if (!c) return MSGPU_E_NOMEM;
    c->device = -1;
    *out      = c;
    return MSGPU_OK;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MSGPU_E_NODEVICE;
  if (device < 0 || device >= ndev) return MSGPU_E_ARG;
  auto *c = new (std::nothrow) msgpu_seqctx();
  if (!c) return MSGPU_E_NOMEM;
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&c->st[0].stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&c->st[1].stream, hipStreamNonBlocking) != hipSuccess) {
    for (auto &st : c->st)
      if (st.stream) (void)hipStreamDestroy(st.stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return MSGPU_E_HIP;
  }
  *out = c;
  return MSGPU_OK;
}

void msgpu_seq_destroy(msgpu_seqctx *c) {
  if (!c) return;
  if (c->device < 0) {
    delete c;
    return;
  }
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto &s : c->st) {
    if (s.stream) {
      (void)hipStreamSynchronize(s.stream);
      (void)hipStreamDestroy(s.stream);
    }
    if (s.d_buf) (void)hipFree(s.d_buf);
    if (s.ring) msgpu::pinned_block_free(s.ring);
    s.drop_packed();
  }
  for (msgpu_seqctx::Scratch *x : {&c->scr_text, &c->scr_recs, &c->scr_map, &c->scr_hdr, &c->scr_pairs, &c->scr_dist})
    if (x->p) (void)hipFree(x->p);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *msgpu_seq_last_error(const msgpu_seqctx *c) { return c ? c->err : "null context"; }

// The two halves of msgpu_seq_upload.  The bytes need nothing but the parsed file, the id table needs the Registry ids of
// its records (which exist once the PAF is read): a caller that parses the sequence files beside the PAF sends the bytes
// as soon as a file is parsed (from the parsing thread: the two stores have their own streams) and sets the ids later.
int msgpu_seq_upload_bases(msgpu_seqctx *c, int kind, const msgpu_seqfile *f) {
  if (!c || !f || kind < 0 || kind > 1) return MSGPU_E_ARG;
  SeqStore &s = c->st[kind];
  uint64_t  total = 0; // the records' bytes go up as they lie in the loader's buffer (with what lies between them)
  const char *first = msgpu_seq_buffer(f, &total);
  if (total && !first && c->device >= 0) return MSGPU_E_ARG; // (a file of msgpu_seq_parse_upload: its bytes are in a store already)
  s.off.clear();
  s.len.clear();
  s.n_bases = total;
  if (c->device < 0) return MSGPU_OK; // layout-only context: offsets and lengths are all it needs
  SHIP(c, hipSetDevice(c->device));
  s.drop_packed();
  if (s.d_buf) {
    SHIP(c, hipFree(s.d_buf));
    s.d_buf = nullptr;
  }
  SHIP(c, hipMalloc(&s.d_buf, total + 2 * SEQ_PAD));
  SHIP(c, hipMemsetAsync(s.d_buf, 0, SEQ_PAD, s.stream));
  SHIP(c, hipMemsetAsync(static_cast<uint8_t *>(s.d_buf) + SEQ_PAD + total, 0, SEQ_PAD, s.stream));
  if (total)
    SHIP(c, hipMemcpyAsync(static_cast<uint8_t *>(s.d_buf) + SEQ_PAD, first, total, hipMemcpyHostToDevice, s.stream));
  SHIP(c, hipStreamSynchronize(s.stream));
  return MSGPU_OK;
}

namespace {
// msgpu_seq_parse_upload: every parser thread strips its stretch of the file into page-locked slots of a few MiB, and a
// full slot goes to its place in HBM (the position its bytes have in the file-sized layout of seq_loader.cpp) while the
// thread fills the next one.  The host never holds the file's bases: no gigabyte of first-touch page faults, nothing to
// give back afterwards, and the copy engine works while the file is still being read.
constexpr size_t RING_SLOT  = size_t(4) << 20;
constexpr int    RING_SLOTS = 2; // per stretch
struct DeviceDestination final : msgpu::SeqDestination {
  struct Sink final : msgpu::ByteSink {
    DeviceDestination *d = nullptr;
    char              *slot[RING_SLOTS] = {nullptr};
    hipEvent_t         ev[RING_SLOTS]   = {nullptr};
    bool               busy[RING_SLOTS] = {false};
    int                cur = 0;
    size_t             fill = 0;
    uint64_t           at0 = 0; // position of slot[cur][0]
    void flush() {
      if (!fill) return;
      hipError_t e = hipMemcpyAsync(d->d_bases + at0, slot[cur], fill, hipMemcpyHostToDevice, d->stream);
      if (e == hipSuccess) e = hipEventRecord(ev[cur], d->stream);
      if (e != hipSuccess) d->fail(e);
      busy[cur] = true;
      at0 += fill;
      fill = 0;
      cur  = (cur + 1) % RING_SLOTS;
      if (busy[cur]) { // the slot we are about to fill again must have left
        e = hipEventSynchronize(ev[cur]);
        if (e != hipSuccess) d->fail(e);
        busy[cur] = false;
      }
    }
    void put(uint64_t at, const char *p, size_t n) override {
      if (at != at0 + fill) { // (a stretch writes back to back; a jump can only be the start of the stretch)
        flush();
        at0 = at;
      }
      while (n) {
        const size_t k = std::min(n, RING_SLOT - fill);
        memcpy(slot[cur] + fill, p, k);
        fill += k;
        p += k;
        n -= k;
        if (fill == RING_SLOT) flush();
      }
    }
    void done() override { flush(); }
  };
  msgpu_seqctx     *c;
  SeqStore         &s;
  hipStream_t       stream;
  uint8_t          *d_bases = nullptr;
  std::vector<Sink> sinks;
  std::atomic<int>  err{hipSuccess};
  DeviceDestination(msgpu_seqctx *ctx, SeqStore &st) : c(ctx), s(st), stream(st.stream) {}
  ~DeviceDestination() override {
    for (Sink &k : sinks)
      for (hipEvent_t e : k.ev)
        if (e) (void)hipEventDestroy(e);
  }
  void fail(hipError_t e) {
    int none = hipSuccess;
    err.compare_exchange_strong(none, static_cast<int>(e));
  }
  void prepare(uint64_t file_bytes, unsigned n) override {
    if (hipSetDevice(c->device) != hipSuccess) throw std::bad_alloc();
    s.off.clear();
    s.len.clear();
    s.n_bases = 0;
    s.drop_packed();
    if (s.d_buf) {
      (void)hipFree(s.d_buf);
      s.d_buf = nullptr;
    }
    if (hipMalloc(&s.d_buf, file_bytes + 2 * SEQ_PAD) != hipSuccess) throw std::bad_alloc();
    d_bases = static_cast<uint8_t *>(s.d_buf) + SEQ_PAD;
    // what no record uses (description lines, line ends) must read as bases: the 2-bit form lists every other byte
    hipError_t e = hipMemsetAsync(s.d_buf, 0, SEQ_PAD, stream);
    if (e == hipSuccess && file_bytes) e = hipMemsetAsync(d_bases, 'A', file_bytes, stream);
    if (e != hipSuccess) fail(e);
    const size_t want = size_t(n) * RING_SLOTS * RING_SLOT;
    if (s.ring_bytes < want) {
      if (s.ring) msgpu::pinned_block_free(s.ring);
      s.ring_bytes = 0;
      s.ring       = msgpu::pinned_block_alloc(want);
      if (!s.ring) throw std::bad_alloc();
      s.ring_bytes = want;
    }
    sinks.resize(n);
    for (unsigned k = 0; k < n; ++k) {
      sinks[k].d = this;
      for (int q = 0; q < RING_SLOTS; ++q) {
        sinks[k].slot[q] = static_cast<char *>(s.ring) + (size_t(k) * RING_SLOTS + q) * RING_SLOT;
        if (hipEventCreateWithFlags(&sinks[k].ev[q], hipEventDisableTiming) != hipSuccess) throw std::bad_alloc();
      }
    }
  }
  msgpu::ByteSink *stretch(unsigned k) override {
    (void)hipSetDevice(c->device); // (called on the thread that is about to fill the stretch)
    return &sinks[k];
  }
  void restart() override {
    for (Sink &k : sinks) { // (every stretch has called done(): nothing is half filled)
      k.at0  = 0;
      k.fill = 0;
    }
    // what the stretches sent is overwritten by the one pass that follows, in stream order
  }
  void fill(uint64_t, uint64_t) override {} // (the whole buffer was filled with a base before the first byte arrived)
  int  finish(uint64_t extent) override {
    hipError_t e = hipMemsetAsync(d_bases + extent, 0, SEQ_PAD, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) fail(e);
    if (err.load() != hipSuccess) return sfail(c, MSGPU_E_HIP, "msgpu_seq_parse_upload", static_cast<hipError_t>(err.load()));
    s.n_bases = extent;
    return MSGPU_OK;
  }
};
} // namespace

int msgpu_seq_parse_upload(msgpu_seqctx *c, int kind, const char *path, int is_fastq, msgpu_seqfile **out) {
  if (!c || !path || !out || kind < 0 || kind > 1) return MSGPU_E_ARG;
  if (c->device < 0) return MSGPU_E_NODEVICE;
  try {
    DeviceDestination dst(c, c->st[kind]);
    return msgpu::seq_parse_into(path, is_fastq, &dst, out);
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
}

int msgpu_seq_set_ids(msgpu_seqctx *c, int kind, const msgpu_seqfile *f, const uint32_t *ids, uint32_t n_ids) {
  if (!c || !f || kind < 0 || kind > 1) return MSGPU_E_ARG;
  SeqStore      &s = c->st[kind];
  const uint32_t n = msgpu_seq_count(f);
  uint32_t       space = ids ? n_ids : n;
  uint64_t       total = 0;
  (void)msgpu_seq_buffer(f, &total);
  if (total != s.n_bases) { // not the file msgpu_seq_upload_bases / msgpu_seq_parse_upload sent
    snprintf(c->err, sizeof(c->err), "msgpu_seq_set_ids: the store holds %llu bytes, the file %llu",
             static_cast<unsigned long long>(s.n_bases), static_cast<unsigned long long>(total));
    return MSGPU_E_STATE;
  }
  for (uint32_t i = 0; ids && i < n; ++i)
    if (ids[i] != 0xffffffffu && ids[i] >= space) return MSGPU_E_ARG;
  try {
    s.off.assign(space, ~0ull);
    s.len.assign(space, 0);
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t id = ids ? ids[i] : i;
    if (id != 0xffffffffu && s.off[id] == ~0ull) { // emplace: the first record of an id wins
      s.off[id] = msgpu_seq_offset(f, i);
      s.len[id] = msgpu_seq_length(f, i);
    }
  }
  return MSGPU_OK;
}

int msgpu_seq_upload(msgpu_seqctx *c, int kind, const msgpu_seqfile *f, const uint32_t *ids, uint32_t n_ids) {
  if (!c || !f || kind < 0 || kind > 1) return MSGPU_E_ARG;
  const uint32_t n = msgpu_seq_count(f);
  for (uint32_t i = 0; ids && i < n; ++i) // (before anything is touched)
    if (ids[i] != 0xffffffffu && ids[i] >= n_ids) return MSGPU_E_ARG;
  const int rc = msgpu_seq_upload_bases(c, kind, f);
  return rc != MSGPU_OK ? rc : msgpu_seq_set_ids(c, kind, f, ids, n_ids);
}

int msgpu_seq_upload_device(msgpu_seqctx *c, int kind, const void *d_bases, uint64_t n_bases, const uint64_t *off,
                            const uint64_t *len, uint32_t n_ids) {
  if (!c || kind < 0 || kind > 1 || (n_bases && !d_bases) || (n_ids && (!off || !len))) return MSGPU_E_ARG;
  if (c->device < 0) return MSGPU_E_NODEVICE;
  SHIP(c, hipSetDevice(c->device));
  SeqStore &s = c->st[kind];
  for (uint32_t i = 0; i < n_ids; ++i)
    if (off[i] != ~0ull && off[i] + len[i] > n_bases) return MSGPU_E_ARG;
  s.off.assign(off, off + n_ids);
  s.len.assign(len, len + n_ids);
  s.drop_packed();
  if (s.d_buf) {
    SHIP(c, hipFree(s.d_buf));
    s.d_buf = nullptr;
  }
  s.n_bases = n_bases;
  SHIP(c, hipMalloc(&s.d_buf, n_bases + 2 * SEQ_PAD));
  SHIP(c, hipMemsetAsync(s.d_buf, 0, SEQ_PAD, c->stream));
  SHIP(c, hipMemsetAsync(static_cast<uint8_t *>(s.d_buf) + SEQ_PAD + n_bases, 0, SEQ_PAD, c->stream));
  if (n_bases)
    SHIP(c, hipMemcpyAsync(static_cast<uint8_t *>(s.d_buf) + SEQ_PAD, d_bases, n_bases, hipMemcpyDeviceToDevice, c->stream));
  SHIP(c, hipStreamSynchronize(c->stream));
  return MSGPU_OK;
}

int msgpu_seq_resolve(msgpu_seqctx *c, int kind, uint32_t seq_id, int32_t left, int32_t right, int direction,
                      msgpu_copy *out) {
  if (!c || !out || kind < 0 || kind > 1) return MSGPU_E_ARG;
  const SeqStore &s = c->st[kind];
  if (seq_id >= s.off.size() || s.off[seq_id] == ~0ull) return MSGPU_E_ARG; // m_idx.at(id) would throw
  uint64_t       len   = 0;
  const uint64_t start = msgpu_str_slice(s.len[seq_id], left, right + 1, &len); // SequenceUtils.cpp:66,78
  if (len > 0xffffffffull) return MSGPU_E_ARG;
  out->src_off = s.off[seq_id] + start;
  out->len     = static_cast<uint32_t>(len);
  out->flags   = (kind ? MSGPU_COPY_ILLUMINA : 0u) | (direction ? 0u : MSGPU_COPY_REVCOMP);
  return MSGPU_OK;
}

int msgpu_gather_plan_create(msgpu_seqctx *c, const msgpu_copy *pieces, size_t n, msgpu_gather_plan **out) {
  if (!c || !out || (n && !pieces) || n >= 0xfffffff0ull) return MSGPU_E_ARG;
  *out = nullptr;
  if (c->device < 0) return MSGPU_E_NODEVICE;
  SHIP(c, hipSetDevice(c->device));
  // the work partition: every piece is cut into 1-KiB chunks (cmap) and 4-KiB super-chunks (smap) of its output, 2 words
  // each.  A deep assembly has millions of pieces: counted and filled in stretches on host threads (a stretch's first
  // chunk number is the prefix over the stretches before it).
  std::unique_ptr<uint32_t[]> cmap, smap;
  uint64_t                    chunks = 0, supers = 0, out_bytes = 0, bases = 0;
  {
    unsigned nt = std::thread::hardware_concurrency();
    nt          = nt == 0 ? 1 : (nt > 16 ? 16 : nt);
    if (n < 200000) nt = 1;
    struct Part {
      uint64_t chunks = 0, supers = 0, out_bytes = 0, bases = 0;
      int      rc = MSGPU_OK;
    };
    std::vector<Part> part(nt);
    auto chunks_of = [](const msgpu_copy &p) { return ((p.dst_off + p.len) - (p.dst_off & ~15ull) + GCHUNK - 1) / GCHUNK; };
    auto on_threads = [&](auto &&body) { // body(t) for t < nt on the library's parked threads
      msgpu::HostPool::get().run(nt, nt, [&](size_t t) { body(static_cast<unsigned>(t)); });
    };
    try {
      on_threads([&](unsigned t) {
        Part &pt = part[t];
        for (size_t i = n * t / nt; i < n * (t + 1) / nt; ++i) {
          const msgpu_copy &p = pieces[i];
          const SeqStore   &s = c->st[(p.flags & MSGPU_COPY_ILLUMINA) ? 1 : 0];
          if (p.src_off + p.len > s.n_bases || p.len >= 0x7fffffffu) { // never read outside the store
            pt.rc = MSGPU_E_ARG;
            return;
          }
          if (p.len) {
            const uint64_t k = chunks_of(p);
            pt.chunks += k;
            pt.supers += (k + SSUB - 1) / SSUB;
          }
          if (p.dst_off + p.len > pt.out_bytes) pt.out_bytes = p.dst_off + p.len;
          pt.bases += p.len;
        }
      });
      std::vector<uint64_t> cbase(nt + 1, 0), sbase(nt + 1, 0);
      for (unsigned t = 0; t < nt; ++t) {
        if (part[t].rc != MSGPU_OK) return part[t].rc;
        cbase[t + 1] = cbase[t] + part[t].chunks;
        sbase[t + 1] = sbase[t] + part[t].supers;
        if (part[t].out_bytes > out_bytes) out_bytes = part[t].out_bytes;
        bases += part[t].bases;
      }
      chunks = cbase[nt];
      supers = sbase[nt];
      if (chunks >= 0x1fffffffcull) return MSGPU_E_ARG; // grid.x = chunks / 4 must fit 31 bits
      cmap.reset(new uint32_t[2 * (chunks ? chunks : 1)]);
      smap.reset(new uint32_t[2 * (supers ? supers : 1)]);
      on_threads([&](unsigned t) {
        uint64_t cq = 2 * cbase[t], sq = 2 * sbase[t];
        for (size_t i = n * t / nt; i < n * (t + 1) / nt; ++i) {
          const msgpu_copy &p = pieces[i];
          if (!p.len) continue;
          const uint64_t k = chunks_of(p);
          for (uint64_t q = 0; q < k; ++q) {
            cmap[cq++] = static_cast<uint32_t>(i);
            cmap[cq++] = static_cast<uint32_t>(q);
          }
          for (uint64_t q = 0; q < (k + SSUB - 1) / SSUB; ++q) {
            smap[sq++] = static_cast<uint32_t>(i);
            smap[sq++] = static_cast<uint32_t>(q);
          }
        }
      });
    } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; } catch (std::system_error const &) {
      return MSGPU_E_NOMEM;
    }
  }
  auto *pl = new (std::nothrow) msgpu_gather_plan();
  if (!pl) return MSGPU_E_NOMEM;
  pl->n         = static_cast<uint32_t>(n);
  pl->n_chunks  = chunks;
  pl->n_super   = supers;
  pl->out_bytes = out_bytes;
  pl->bases     = bases;
  hipError_t e  = hipMalloc(&pl->d_pieces, (n ? n : 1) * sizeof(msgpu_copy));
  if (e == hipSuccess) e = hipMalloc(&pl->d_chunk_map, (chunks ? chunks : 1) * 8);
  if (e == hipSuccess) e = hipMalloc(&pl->d_super_map, (supers ? supers : 1) * 8);
  if (e == hipSuccess && n)
    e = hipMemcpyAsync(pl->d_pieces, pieces, n * sizeof(msgpu_copy), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && chunks)
    e = hipMemcpyAsync(pl->d_chunk_map, cmap.get(), chunks * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && supers)
    e = hipMemcpyAsync(pl->d_super_map, smap.get(), supers * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) {
    if (pl->d_pieces) (void)hipFree(pl->d_pieces);
    if (pl->d_chunk_map) (void)hipFree(pl->d_chunk_map);
    if (pl->d_super_map) (void)hipFree(pl->d_super_map);
    delete pl;
    return sfail(c, e == hipErrorOutOfMemory ? MSGPU_E_NOMEM : MSGPU_E_HIP, "gather plan", e);
  }
  *out = pl;
  return MSGPU_OK;
}

void msgpu_gather_plan_free(msgpu_gather_plan *pl) {
  if (!pl) return;
  if (pl->d_pieces) (void)hipFree(pl->d_pieces);
  if (pl->d_chunk_map) (void)hipFree(pl->d_chunk_map);
  if (pl->d_super_map) (void)hipFree(pl->d_super_map);
  delete pl;
}

uint64_t msgpu_gather_plan_out_bytes(const msgpu_gather_plan *pl) { return pl ? pl->out_bytes : 0; }
uint64_t msgpu_gather_plan_bases(const msgpu_gather_plan *pl) { return pl ? pl->bases : 0; }

int msgpu_gather_run(msgpu_seqctx *c, const msgpu_gather_plan *pl, void *d_out, uint64_t out_capacity, void *hip_stream) {
  if (!c || !pl || (!d_out && pl->out_bytes)) return MSGPU_E_ARG;
  if (out_capacity < pl->out_bytes) return MSGPU_E_ARG;
  if (c->device < 0) return MSGPU_E_NODEVICE;
  SHIP(c, hipSetDevice(c->device));
  if (!pl->n_chunks) return MSGPU_OK;
  hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
  if ((c->st[0].packed && c->st[1].d_buf) || (c->st[1].packed && c->st[0].d_buf)) {
    snprintf(c->err, sizeof(c->err), "one store is packed and the other is not: call msgpu_seq_pack after the last upload");
    return MSGPU_E_STATE;
  }
  if (c->st[0].packed || c->st[1].packed) { // msgpu_seq_pack converts both stores
    const uint32_t *w0 = c->st[0].d_words ? static_cast<const uint32_t *>(c->st[0].d_words) + PACK_PAD : nullptr;
    const uint32_t *w1 = c->st[1].d_words ? static_cast<const uint32_t *>(c->st[1].d_words) + PACK_PAD : nullptr;
    hipLaunchKernelGGL(k_gather_packed, dim3(static_cast<uint32_t>((pl->n_super + 3) / 4)), dim3(256), 0, st,
                       static_cast<const msgpu_copy *>(pl->d_pieces), static_cast<const uint2 *>(pl->d_super_map),
                       pl->n_super, w0, w1, static_cast<uint8_t *>(d_out));
    if (c->st[0].n_exc || c->st[1].n_exc)
      hipLaunchKernelGGL(k_patch_exceptions, dim3((pl->n + 255) / 256), dim3(256), 0, st,
                         static_cast<const msgpu_copy *>(pl->d_pieces), pl->n,
                         static_cast<const uint64_t *>(c->st[0].d_exc_pos), static_cast<const uint8_t *>(c->st[0].d_exc_byte),
                         c->st[0].n_exc, static_cast<const uint64_t *>(c->st[1].d_exc_pos),
                         static_cast<const uint8_t *>(c->st[1].d_exc_byte), c->st[1].n_exc, static_cast<uint8_t *>(d_out));
    SHIP(c, hipGetLastError());
    return MSGPU_OK;
  }
  const uint8_t *b0 = c->st[0].d_buf ? static_cast<const uint8_t *>(c->st[0].d_buf) + SEQ_PAD : nullptr;
  const uint8_t *b1 = c->st[1].d_buf ? static_cast<const uint8_t *>(c->st[1].d_buf) + SEQ_PAD : nullptr;
  hipLaunchKernelGGL(k_gather, dim3(static_cast<uint32_t>((pl->n_chunks + 4 * GUNROLL - 1) / (4 * GUNROLL))), dim3(256), 0, st,
                     static_cast<const msgpu_copy *>(pl->d_pieces), static_cast<const uint2 *>(pl->d_chunk_map),
                     pl->n_chunks, b0, b1, static_cast<uint8_t *>(d_out));
  SHIP(c, hipGetLastError());
  return MSGPU_OK;
}

// Convert one resident store to the 2-bit form (+ exception list) and free its byte-per-base buffer (on the store's own
// stream: the two stores may be converted from two host threads).
int msgpu_seq_pack_store(msgpu_seqctx *c, int kind) {
  if (!c || kind < 0 || kind > 1) return MSGPU_E_ARG;
  if (c->device < 0) return MSGPU_E_NODEVICE;
  SHIP(c, hipSetDevice(c->device));
  {
    SeqStore &s = c->st[kind];
    if (s.packed || !s.d_buf) return MSGPU_OK;
    hipStream_t st = s.stream;
    const uint64_t n_words = (s.n_bases + 15) / 16;
    const uint8_t *bases   = static_cast<const uint8_t *>(s.d_buf) + SEQ_PAD;
    void          *d_cnt   = nullptr;
    SHIP(c, hipMalloc(&s.d_words, (n_words + 2 * PACK_PAD) * 4));
    hipError_t e = hipMalloc(&d_cnt, 8);
    if (e == hipSuccess) e = hipMemsetAsync(s.d_words, 0, (n_words + 2 * PACK_PAD) * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, 8, st);
    unsigned long long n_exc = 0;
    if (e == hipSuccess && n_words) {
      hipLaunchKernelGGL(k_pack_bases, dim3(static_cast<uint32_t>((n_words + 255) / 256)), dim3(256), 0, st, bases,
                         s.n_bases, static_cast<uint32_t *>(s.d_words) + PACK_PAD, static_cast<unsigned long long *>(d_cnt),
                         static_cast<uint64_t *>(nullptr), static_cast<uint8_t *>(nullptr), 0);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&n_exc, d_cnt, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && n_exc) { // second pass records them; sorted by position on the host (they are few)
      e = hipMalloc(&s.d_exc_pos, n_exc * 8);
      if (e == hipSuccess) e = hipMalloc(&s.d_exc_byte, n_exc);
      if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, 8, st);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_pack_bases, dim3(static_cast<uint32_t>((n_words + 255) / 256)), dim3(256), 0, st, bases,
                           s.n_bases, static_cast<uint32_t *>(nullptr), static_cast<unsigned long long *>(d_cnt),
                           static_cast<uint64_t *>(s.d_exc_pos), static_cast<uint8_t *>(s.d_exc_byte), 1);
        e = hipGetLastError();
      }
      std::vector<uint64_t> pos(n_exc);
      std::vector<uint8_t>  byt(n_exc);
      if (e == hipSuccess) e = hipMemcpyAsync(pos.data(), s.d_exc_pos, n_exc * 8, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipMemcpyAsync(byt.data(), s.d_exc_byte, n_exc, hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e == hipSuccess) {
        std::vector<uint64_t> idx(n_exc);
        for (uint64_t i = 0; i < n_exc; ++i) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](uint64_t x, uint64_t y) { return pos[x] < pos[y]; });
        std::vector<uint64_t> spos(n_exc);
        std::vector<uint8_t>  sbyt(n_exc);
        for (uint64_t i = 0; i < n_exc; ++i) {
          spos[i] = pos[idx[i]];
          sbyt[i] = byt[idx[i]];
        }
        e = hipMemcpyAsync(s.d_exc_pos, spos.data(), n_exc * 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(s.d_exc_byte, sbyt.data(), n_exc, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
      }
    }
    if (d_cnt) (void)hipFree(d_cnt);
    if (e != hipSuccess) {
      s.drop_packed();
      return sfail(c, e == hipErrorOutOfMemory ? MSGPU_E_NOMEM : MSGPU_E_HIP, "msgpu_seq_pack", e);
    }
    s.n_exc  = n_exc;
    s.packed = true;
    (void)hipFree(s.d_buf); // the byte-per-base copy is no longer needed
    s.d_buf = nullptr;
  }
  return MSGPU_OK;
}

// Convert both resident stores.
int msgpu_seq_pack(msgpu_seqctx *c) {
  if (!c) return MSGPU_E_ARG;
  for (int kind = 0; kind < 2; ++kind) {
    const int rc = msgpu_seq_pack_store(c, kind);
    if (rc != MSGPU_OK) return rc;
  }
  return MSGPU_OK;
}

uint64_t msgpu_fasta_text_bytes(uint32_t header_len, uint64_t len) {
  return header_len + len + (len ? (len - 1) / FLINE : 0) + 1;
}

int msgpu_fasta_format(msgpu_seqctx *c, const void *d_raw, const msgpu_fasta_record *records, size_t n,
                       const char *headers, size_t headers_bytes, void *d_text, uint64_t text_capacity,
                       void *hip_stream) {
  if (!c || (n && (!records || !d_text)) || (headers_bytes && !headers) || n >= 0xfffffff0ull) return MSGPU_E_ARG;
  if (c->device < 0) return MSGPU_E_NODEVICE;
  SHIP(c, hipSetDevice(c->device));
  if (!n) return MSGPU_OK;
  std::vector<uint32_t> cmap;
  uint64_t              chunks = 0;
  try {
    for (size_t i = 0; i < n; ++i) {
      const msgpu_fasta_record &r = records[i];
      if (static_cast<uint64_t>(r.header_off) + r.header_len > headers_bytes) return MSGPU_E_ARG;
      if (r.len && !d_raw) return MSGPU_E_ARG;
      const uint64_t t1 = r.text_off + msgpu_fasta_text_bytes(r.header_len, r.len);
      if (t1 > text_capacity) return MSGPU_E_ARG; // never write outside the text buffer
      const uint64_t k = (t1 - (r.text_off & ~15ull) + GCHUNK - 1) / GCHUNK;
      if (chunks + k >= 0x1fffffffcull) return MSGPU_E_ARG;
      for (uint64_t q = 0; q < k; ++q) {
        cmap.push_back(static_cast<uint32_t>(i));
        cmap.push_back(static_cast<uint32_t>(q));
      }
      chunks += k;
    }
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
  hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
  hipError_t  e  = c->scr_recs.ensure(n * sizeof(msgpu_fasta_record));
  if (e == hipSuccess) e = c->scr_map.ensure(chunks * 8);
  if (e == hipSuccess) e = c->scr_hdr.ensure(headers_bytes ? headers_bytes : 1);
  void *d_recs = c->scr_recs.p, *d_map = c->scr_map.p, *d_hdr = c->scr_hdr.p;
  if (e == hipSuccess) e = hipMemcpyAsync(d_recs, records, n * sizeof(msgpu_fasta_record), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_map, cmap.data(), chunks * 8, hipMemcpyHostToDevice, st);
  if (e == hipSuccess && headers_bytes) e = hipMemcpyAsync(d_hdr, headers, headers_bytes, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_fasta_format, dim3(static_cast<uint32_t>((chunks + 3) / 4)), dim3(256), 0, st,
                       static_cast<const msgpu_fasta_record *>(d_recs), static_cast<const uint2 *>(d_map), chunks,
                       static_cast<const uint8_t *>(d_raw), static_cast<const uint8_t *>(d_hdr),
                       static_cast<uint8_t *>(d_text));
    e = hipGetLastError();
  }
  // the staging copies come from pageable host memory (`cmap` is local): wait before they go out of scope
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return sfail(c, e == hipErrorOutOfMemory ? MSGPU_E_NOMEM : MSGPU_E_HIP, "fasta format", e);
  return MSGPU_OK;
}

} // extern "C"

namespace {
// Process-wide pools for the two buffers an assembly keeps after msgpu_assembly_finish (the gathered bases on the device,
// the texts in pinned host memory).  Pinning 20 MB costs > 1 ms and a hipMalloc / hipFree pair synchronises the device,
// so a freed assembly parks its buffers here (at most two of each kind) for the next one; nothing here refers to a
// context, so an assembly may outlive the sequence context it was laid out over.
struct ParkedBuffers {
  struct Buf {
    void  *p;
    size_t cap;
    int    device; // -1: pinned host memory
  };
  std::mutex       m;
  std::vector<Buf> free_list;
  void *take(size_t bytes, int device) {
    std::lock_guard<std::mutex> g(m);
    for (size_t i = 0; i < free_list.size(); ++i)
      if (free_list[i].device == device && free_list[i].cap >= bytes && free_list[i].cap <= 2 * bytes + (1u << 20)) {
        void *p = free_list[i].p;
        caps[p] = free_list[i].cap;
        free_list.erase(free_list.begin() + static_cast<long>(i));
        return p;
      }
    return nullptr;
  }
  void give(void *p, size_t cap, int device) {
    std::unique_lock<std::mutex> g(m);
    caps.erase(p);
    size_t same = 0;
    for (const Buf &b : free_list) same += b.device == device;
    if (same >= 2) { // keep the pool small: release instead
      g.unlock();
      if (device < 0) msgpu::pinned_block_free(p);
      else (void)hipFree(p);
      return;
    }
    free_list.push_back(Buf{p, cap, device});
  }
  std::unordered_map<void *, size_t> caps; // capacity of buffers currently lent out
  size_t cap_of(void *p) {
    std::lock_guard<std::mutex> g(m);
    auto it = caps.find(p);
    return it == caps.end() ? 0 : it->second;
  }
  void lend(void *p, size_t cap) {
    std::lock_guard<std::mutex> g(m);
    caps[p] = cap;
  }
};
ParkedBuffers &parked() {
  static ParkedBuffers *pool = new ParkedBuffers(); // never destroyed: no HIP calls during static destruction
  return *pool;
}
} // namespace

extern "C" {

// gather every piece of every path once, wrap targets and queries into FASTA text on the device, copy the texts back
int msgpu_assembly_finish(msgpu_assembly *a, void *hip_stream) {
  if (!a) return MSGPU_E_ARG;
  if (a->finished) return MSGPU_OK;
  msgpu_seqctx *c = a->ctx;
  if (c->device < 0) return MSGPU_E_NODEVICE;
  SHIP(c, hipSetDevice(c->device));
  hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
  // one text buffer: all target records, then (16-byte aligned) all query records; one wrapping launch, one copy back
  std::vector<msgpu_fasta_record> recs;
  std::string                     hdr;
  uint64_t                        t_bytes = 0, q_bytes = 0, q_off = 0;
  try {
    for (const msgpu_path_info &p : a->paths) {
      if (p.target_len > 0xffffffffull) return MSGPU_E_ARG;
      const std::string h = msgpu::target_header(p.asm_idx);
      recs.push_back(msgpu_fasta_record{p.target_raw_off, t_bytes, static_cast<uint32_t>(p.target_len),
                                        static_cast<uint32_t>(hdr.size()), static_cast<uint32_t>(h.size()), 0});
      hdr += h;
      t_bytes += msgpu_fasta_text_bytes(static_cast<uint32_t>(h.size()), p.target_len);
    }
    q_off = (t_bytes + 15) & ~15ull;
    // the query records (path order = ascending record number): header texts and text offsets of a deep assembly -- 643 k
    // records on the tiled workload -- are built in stretches on host threads, then shifted by what precedes each stretch
    const size_t nq = a->queries.size();
    unsigned     nt = std::thread::hardware_concurrency();
    nt              = nt == 0 ? 1 : (nt > 16 ? 16 : nt);
    if (nq < 50000) nt = 1;
    struct Part {
      std::string                     hdr;
      std::vector<msgpu_fasta_record> recs;
      uint64_t                        text = 0;
      bool                            oom  = false;
    };
    std::vector<Part> parts(nt);
    auto              build = [&](unsigned t) {
      Part &pt = parts[t];
      try {
        const size_t b = nq * t / nt, e = nq * (t + 1) / nt;
        pt.recs.reserve(e - b);
        for (size_t q = b; q < e; ++q) {
          const msgpu_query_info &qi = a->queries[q];
          const msgpu_path_info  &p  = a->paths[qi.path];
          const std::string       qh = msgpu::query_header(qi.kind, p.asm_idx, static_cast<uint32_t>(q) - p.query_begin);
          pt.recs.push_back(msgpu_fasta_record{qi.raw_off, pt.text, static_cast<uint32_t>(qi.len),
                                               static_cast<uint32_t>(pt.hdr.size()), static_cast<uint32_t>(qh.size()), 0});
          pt.hdr += qh;
          pt.text += msgpu_fasta_text_bytes(static_cast<uint32_t>(qh.size()), qi.len);
        }
      } catch (std::bad_alloc const &) { pt.oom = true; }
    };
    msgpu::HostPool::get().run(nt, nt, [&](size_t t) { build(static_cast<unsigned>(t)); });
    for (const Part &pt : parts)
      if (pt.oom) return MSGPU_E_NOMEM;
    size_t hdr_total = hdr.size();
    for (const Part &pt : parts) hdr_total += pt.hdr.size();
    if (hdr_total > 0xffffffffull) return MSGPU_E_ARG;
    recs.reserve(recs.size() + nq);
    hdr.reserve(hdr_total);
    for (Part &pt : parts) {
      const uint32_t hbase = static_cast<uint32_t>(hdr.size());
      for (msgpu_fasta_record r : pt.recs) {
        r.text_off += q_off + q_bytes;
        r.header_off += hbase;
        recs.push_back(r);
      }
      hdr += pt.hdr;
      q_bytes += pt.text;
    }
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
  const uint64_t     text_bytes = q_off + q_bytes;
  const bool         dbg        = std::getenv("MSGPU_SEQ_DEBUG") != nullptr;
  auto               t_prev     = std::chrono::steady_clock::now();
  auto               tick       = [&](const char *what) {
    if (!dbg) return;
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[finish] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t_prev).count());
    t_prev = n;
  };
  msgpu_gather_plan *plan       = nullptr;
  int                rc         = msgpu_gather_plan_create(c, a->pieces.data(), a->pieces.size(), &plan);
  if (rc != MSGPU_OK) return rc;
  tick("plan");
  hipError_t e      = hipSuccess;
  void      *d_raw  = parked().take(a->raw_bytes + 64, c->device);
  char      *h_text = static_cast<char *>(parked().take(text_bytes + 16, -1));
  if (!d_raw) {
    e = hipMalloc(&d_raw, a->raw_bytes + 64);
    if (e == hipSuccess) parked().lend(d_raw, a->raw_bytes + 64);
  }
  if (e == hipSuccess) e = c->scr_text.ensure(text_bytes + 16);
  if (e == hipSuccess && !h_text) {
    h_text = static_cast<char *>(msgpu::pinned_block_alloc(text_bytes + 16));
    if (!h_text) e = hipErrorOutOfMemory;
    if (e == hipSuccess) parked().lend(h_text, text_bytes + 16);
  }
  if (e == hipSuccess && std::getenv("MSGPU_POISON")) { // see DevBuf::ensure in msgpu_api.hip
    e = hipMemset(d_raw, 0xA5, a->raw_bytes + 64);
    if (e == hipSuccess) e = hipMemset(c->scr_text.p, 0xA5, text_bytes + 16);
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  tick("allocations");
  if (e == hipSuccess) {
    rc = msgpu_gather_run(c, plan, d_raw, a->raw_bytes + 64, st);
    if (rc == MSGPU_OK)
      rc = msgpu_fasta_format(c, d_raw, recs.data(), recs.size(), hdr.data(), hdr.size(), c->scr_text.p, text_bytes, st);
    tick("gather + format");
    if (rc == MSGPU_OK && text_bytes) e = hipMemcpyAsync(h_text, c->scr_text.p, text_bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    tick("copy back");
  }
  msgpu_gather_plan_free(plan);
  tick("plan free");
  if (e != hipSuccess || rc != MSGPU_OK) {
    if (d_raw) parked().give(d_raw, parked().cap_of(d_raw), c->device);
    if (h_text) parked().give(h_text, parked().cap_of(h_text), -1);
    if (e != hipSuccess) return sfail(c, e == hipErrorOutOfMemory ? MSGPU_E_NOMEM : MSGPU_E_HIP, "assembly finish", e);
    return rc;
  }
  a->text          = h_text;
  a->target_fa_len = t_bytes;
  a->query_fa_off  = q_off;
  a->query_fa_len  = q_bytes;
  a->d_raw         = d_raw; // kept for msgpu_assembly_validate
  a->raw_device    = c->device;
  a->release       = [](msgpu_assembly *x) { // must not touch x->ctx: the sequence context may already be gone
    if (x->d_raw) parked().give(x->d_raw, parked().cap_of(x->d_raw), x->raw_device);
    if (x->text) parked().give(x->text, parked().cap_of(x->text), -1);
    x->d_raw = nullptr;
    x->text  = nullptr;
  };
  a->finished = true;
  return MSGPU_OK;
}

// Banded edit distance of every query record against the stretch of its contig that its PAF line names (columns 8-9):
// the self-check of an assembly, and the first real workload of the anti-diagonal DP kernel.
int msgpu_assembly_validate(msgpu_assembly *a, uint32_t band, uint32_t *distance, uint64_t *dp_cells) {
  if (!a || (!a->queries.empty() && !distance)) return MSGPU_E_ARG;
  if (!a->finished || !a->d_raw) return MSGPU_E_STATE;
  std::vector<msgpu_align_pair> pairs(a->queries.size());
  uint64_t                      cells = 0;
  for (size_t i = 0; i < pairs.size(); ++i) {
    const msgpu_query_info &q = a->queries[i];
    const msgpu_path_info  &p = a->paths[q.path];
    // window [lb, rb] on the contig, clipped to it (records placed outside the contig get an empty window)
    int64_t lo = q.lb < 0 ? 0 : q.lb, hi = q.rb >= static_cast<int64_t>(p.target_len) ? static_cast<int64_t>(p.target_len) - 1 : q.rb;
    if (hi < lo) {
      lo = 0;
      hi = -1;
    }
    if (q.len > 0xffffffffull) return MSGPU_E_ARG;
    pairs[i].a_off = q.raw_off;
    pairs[i].a_len = static_cast<uint32_t>(q.len);
    pairs[i].b_off = p.target_raw_off + static_cast<uint64_t>(lo);
    pairs[i].b_len = static_cast<uint32_t>(hi - lo + 1);
    const uint64_t shorter = pairs[i].a_len < pairs[i].b_len ? pairs[i].a_len : pairs[i].b_len;
    cells += shorter * (2ull * band + 1);
  }
  if (dp_cells) *dp_cells = cells;
  return msgpu_edit_distance(a->ctx, a->d_raw, a->d_raw, pairs.data(), pairs.size(), band, distance);
}

int msgpu_seq_synchronize(msgpu_seqctx *c) {
  if (!c) return MSGPU_E_ARG;
  if (c->device < 0) return MSGPU_OK;
  SHIP(c, hipStreamSynchronize(c->stream));
  return MSGPU_OK;
}

} // extern "C"

// =====================================================================================================================
// banded edit distance
// =====================================================================================================================
//
// The reference contains no sequence-level alignment; north_star asks for consensus sequences "within a
// stated edit-distance tolerance" of the reference's, so the tolerance needs a meter: this kernel measures the
// Levenshtein distance of pairs of sequences (e.g. our contig vs. another build's contig, or a query vs. its window of
// the target given by align.paf).  Unit costs, global alignment, band |j - i| <= W: exact whenever the true distance is
// <= W, otherwise reported as W + 1.
//
// One wavefront per pair.  The DP runs along anti-diagonals t = i + j: cell (i, j) on diagonal k = j - i needs
//   up   (i-1, j)   = diagonal k+1, anti-diagonal t-1
//   left (i, j-1)   = diagonal k-1, anti-diagonal t-1
//   diag (i-1, j-1) = diagonal k,   anti-diagonal t-2
// so one value per diagonal is all the state there is, and only diagonals with k = t (mod 2) are active in step t.
// Each lane owns four consecutive diagonals (two active per step), exchanges its edge values with the neighbour lanes
// by wavefront shuffle once per step, and reads the bases from LDS where both sequences were staged.

namespace msgpu {

constexpr int      ED_DPL   = 4;                 // diagonals per lane
constexpr int      ED_NDIAG = 64 * ED_DPL;       // 256 diagonals: k in [-128, 127]
constexpr uint32_t ED_MAXW  = 127;
#ifndef MSGPU_ED_LDS
#define MSGPU_ED_LDS (12 * 1024)
#endif
constexpr uint32_t ED_LDS   = MSGPU_ED_LDS; // bytes of LDS per sequence (12 KiB: 3 workgroups per CU; 24 KiB left one and
                                            // ran 2.5x slower on 7 kb queries); longer sequences stay in global memory
constexpr uint32_t ED_INF   = 0x3fffffffu;

template <bool IN_LDS>
__device__ __forceinline__ uint32_t edit_distance_wave(const uint8_t *a, uint32_t n, const uint8_t *b, uint32_t m,
                                                       uint32_t W, const uint8_t *la, const uint8_t *lb) {
  const int lane = threadIdx.x & 63;
  const int k0   = lane * ED_DPL - ED_NDIAG / 2; // first diagonal of this lane
  uint32_t  val[ED_DPL];
#pragma unroll
  for (int c = 0; c < ED_DPL; ++c) {
    const int k = k0 + c;
    val[c]      = (static_cast<uint32_t>(k < 0 ? -k : k) <= W) ? static_cast<uint32_t>(k < 0 ? -k : k) : ED_INF;
  }
  const int total = static_cast<int>(n + m);
  for (int t = 2; t <= total; ++t) {
    // values of the neighbouring lanes' edge diagonals (computed in step t-1, or boundary / INF)
    const uint32_t from_left  = __shfl_up(val[ED_DPL - 1], 1);  // diagonal k0 - 1
    const uint32_t from_right = __shfl_down(val[0], 1);         // diagonal k0 + ED_DPL
    const uint32_t nb_lo      = lane == 0 ? ED_INF : from_left;
    const uint32_t nb_hi      = lane == 63 ? ED_INF : from_right;
    const int      par        = (t - k0) & 1; // active diagonals: c = par, par + 2
#pragma unroll
    for (int cc = 0; cc < ED_DPL; cc += 2) {
      // written so that c is a compile-time pair {cc, cc+1} selected by `par` without dynamic register indexing
      const int      k_even = k0 + cc, k_odd = k0 + cc + 1;
      const int      k      = par ? k_odd : k_even;
      const uint32_t ak     = static_cast<uint32_t>(k < 0 ? -k : k);
      const int      i = (t - k) >> 1, j = (t + k) >> 1;
      const bool     ok = ak <= W && i >= 1 && j >= 1 && i <= static_cast<int>(n) && j <= static_cast<int>(m);
      // neighbours k-1 / k+1
      uint32_t left, up, self;
      if (!par) { // c = cc: left = (cc == 0 ? nb_lo : val[cc-1]), up = val[cc+1]
        left = cc == 0 ? nb_lo : val[cc - 1];
        up   = val[cc + 1];
        self = val[cc];
      } else { // c = cc+1: left = val[cc], up = (cc+2 == ED_DPL ? nb_hi : val[cc+2])
        left = val[cc];
        up   = cc + 2 == ED_DPL ? nb_hi : val[cc + 2];
        self = val[cc + 1];
      }
      uint32_t nv = self;
      if (ok) {
        const uint8_t ca = IN_LDS ? la[i - 1] : a[i - 1];
        const uint8_t cb = IN_LDS ? lb[j - 1] : b[j - 1];
        nv               = min(min(up, left) + 1u, self + (ca != cb ? 1u : 0u));
      }
      if (!par)
        val[cc] = nv;
      else
        val[cc + 1] = nv;
    }
  }
  // D[n][m] lives on diagonal k* = m - n
  const int ks = static_cast<int>(m) - static_cast<int>(n);
  if (static_cast<uint32_t>(ks < 0 ? -ks : ks) > W) return W + 1;
  const int idx = ks + ED_NDIAG / 2, src_lane = idx / ED_DPL, c = idx % ED_DPL;
  uint32_t  v   = c == 0 ? val[0] : c == 1 ? val[1] : c == 2 ? val[2] : val[3];
  v             = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), src_lane));
  return v > W ? W + 1 : v;
}

__global__ __launch_bounds__(128) void k_edit_distance_dp(const uint8_t *base_a, const uint8_t *base_b,
                                                       const msgpu_align_pair *pairs, uint32_t n_pairs, uint32_t W,
                                                       uint32_t *out) {
  __shared__ uint8_t s_a[2][ED_LDS], s_b[2][ED_LDS];
  const int      wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t p    = blockIdx.x * 2 + wave;
  if (p >= n_pairs) return;
  const msgpu_align_pair pr = pairs[p];
  const uint8_t *a = base_a + pr.a_off, *b = base_b + pr.b_off;
  uint32_t       d;
  if (pr.a_len <= ED_LDS && pr.b_len <= ED_LDS) {
    for (uint32_t i = lane; i < pr.a_len; i += 64) s_a[wave][i] = a[i];
    for (uint32_t i = lane; i < pr.b_len; i += 64) s_b[wave][i] = b[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    d = edit_distance_wave<true>(a, pr.a_len, b, pr.b_len, W, s_a[wave], s_b[wave]);
  } else {
    d = edit_distance_wave<false>(a, pr.a_len, b, pr.b_len, W, nullptr, nullptr);
  }
  if (lane == 0) out[p] = d;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same distance by furthest-reaching points (Ukkonen 1985, Myers 1986): the DP over (edits e, diagonal k) instead of
// over (i, j).  F_e[k] = the largest i such that a[0, i) and b[0, i + k) are within e edits:
//     F_e[k] = slide(max(F_{e-1}[k] + 1, F_{e-1}[k-1], F_{e-1}[k+1] + 1)),   slide(i) = i + |common prefix of a[i..), b[i+k..)|
// and the distance is the first e with F_e[m - n] = n.  What k_edit_distance_dp spends on (n + m) anti-diagonal steps of
// 2W + 1 cells each, this form spends on at most W + 1 steps of 2e + 1 diagonals plus the slides -- for the pairs the
// meter exists for (a query against its window of the contig: a handful of edits in 7 kb) one slide over the sequence.
// The result is the same number for every input: the banded DP gives min(d, W + 1) because a path of d <= W edits never
// leaves the band, and so does this (tests: both kernels against the full-DP oracle and against each other).
//
// One wavefront per pair.  Lane l owns the diagonals k = l + 64 c - 128, c = 0..3 (neighbouring diagonals in neighbouring
// lanes: one shuffle per side and step).  A slide starts lane-local, 8 bytes per comparison; a lane that is still
// matching after 16 bytes hands its diagonal to the whole wavefront, which compares 512 bytes per round (8 per lane,
// first mismatch by ballot) -- the long slides of near-identical pairs run at the wavefront's width, not one lane's.
// No LDS: every base is read about once, straight from HBM/L2 (staging would be a second pass); 8 waves per SIMD.
// ---------------------------------------------------------------------------------------------------------------------

constexpr int FR_NONE = -(1 << 30);

// length of the common prefix of pa[0, valid) and pb[0, valid), valid <= 8 (nothing behind `valid` is read)
__device__ __forceinline__ int match_run8(const uint8_t *pa, const uint8_t *pb, int valid) {
  if (valid >= 8) {
    uint64_t x, y;
    __builtin_memcpy(&x, pa, 8);
    __builtin_memcpy(&y, pb, 8);
    const uint64_t d = x ^ y;
    return d ? (__builtin_ctzll(d) >> 3) : 8;
  }
  int r = 0;
  while (r < valid && pa[r] == pb[r]) ++r;
  return r;
}

__device__ __forceinline__ uint32_t edit_distance_fr_wave(const uint8_t *a, int n, const uint8_t *b, int m, int W) {
  const int lane = threadIdx.x & 63;
  const int ks   = m - n;
  if ((ks < 0 ? -ks : ks) > W) return static_cast<uint32_t>(W + 1);
  int fr[4] = {FR_NONE, FR_NONE, FR_NONE, FR_NONE};
  for (int e = 0; e <= W; ++e) {
    int nf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = lane + 64 * c - 128;
      int       x;
      if (e == 0) {
        x = k == 0 ? 0 : FR_NONE;
      } else {
        int lo = __shfl_up(fr[c], 1), hi = __shfl_down(fr[c], 1); // diagonals k - 1, k + 1
        const int lo_edge = c > 0 ? __shfl(fr[c > 0 ? c - 1 : 0], 63) : FR_NONE;
        const int hi_edge = c < 3 ? __shfl(fr[c < 3 ? c + 1 : 3], 0) : FR_NONE;
        if (lane == 0) lo = lo_edge;
        if (lane == 63) hi = hi_edge;
        x = max(max(fr[c] + 1, lo), hi + 1);
        const int ak = k < 0 ? -k : k;
        if (ak > W || ak > e) x = FR_NONE;
        x = min(x, min(n, m - k));
        if (x < 0 || x + k < 0) x = FR_NONE;
      }
      nf[c] = x;
    }
    const int R = e < W ? e : W; // diagonals |k| <= R can hold a point
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (!(64 * c - 128 <= R && 64 * c - 65 >= -R)) continue; // (uniform: none of this c's diagonals is in reach)
      const int k = lane + 64 * c - 128;
      int       x = nf[c];
      bool      more = false;
      if (x >= 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int valid = min(8, min(n - x, m - (x + k)));
          const int run   = match_run8(a + x, b + (x + k), valid);
          x += run;
          more = run == 8;
          if (!more) break;
        }
      }
      unsigned long long mask = __ballot(more);
      while (mask) { // the diagonals that keep matching, one after the other at the wavefront's width
        const int L = __builtin_ctzll(mask);
        mask &= mask - 1;
        int       xs = __shfl(x, L);
        const int kk = L + 64 * c - 128;
        for (;;) {
          const int px    = xs + 8 * lane;
          const int valid = min(8, min(n - px, m - (px + kk)));
          const int run   = valid > 0 ? match_run8(a + px, b + (px + kk), valid) : 0;
          const unsigned long long stop = __ballot(run < 8);
          if (stop) {
            const int F = __builtin_ctzll(stop);
            xs += 8 * F + __shfl(run, F);
            break;
          }
          xs += 512;
        }
        if (lane == L) x = xs;
      }
      nf[c] = x;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) fr[c] = nf[c];
    const int slot = ks + 128, cs = slot >> 6;
    const int v    = __shfl(cs == 0 ? fr[0] : cs == 1 ? fr[1] : cs == 2 ? fr[2] : fr[3], slot & 63);
    if (v >= n) return static_cast<uint32_t>(e);
  }
  return static_cast<uint32_t>(W + 1);
}

__global__ __launch_bounds__(256) void k_edit_distance(const uint8_t *base_a, const uint8_t *base_b,
                                                       const msgpu_align_pair *pairs, uint32_t n_pairs, uint32_t W,
                                                       uint32_t *out) {
  const uint32_t p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_pairs) return; // (whole wavefronts leave)
  const msgpu_align_pair pr = pairs[p];
  const uint32_t d = edit_distance_fr_wave(base_a + pr.a_off, static_cast<int>(pr.a_len), base_b + pr.b_off,
                                           static_cast<int>(pr.b_len), static_cast<int>(W));
  if ((threadIdx.x & 63) == 0) out[p] = d;
}

} // namespace msgpu

extern "C" {

int msgpu_edit_distance(msgpu_seqctx *c, const void *d_a, const void *d_b, const msgpu_align_pair *pairs, size_t n,
                        uint32_t band, uint32_t *out) {
  if (!c || (n && (!pairs || !out || !d_a || !d_b)) || band > ED_MAXW || n >= 0x7fffffffull) return MSGPU_E_ARG;
  if (c->device < 0) return MSGPU_E_NODEVICE;
  if (!n) return MSGPU_OK;
  for (size_t i = 0; i < n; ++i)
    if (pairs[i].a_len >= (1u << 30) || pairs[i].b_len >= (1u << 30)) return MSGPU_E_ARG; // (positions are ints on the device)
  SHIP(c, hipSetDevice(c->device));
  SHIP(c, c->scr_pairs.ensure(n * sizeof(msgpu_align_pair)));
  SHIP(c, c->scr_dist.ensure(n * sizeof(uint32_t)));
  void *d_pairs = c->scr_pairs.p, *d_out = c->scr_dist.p;
  hipError_t e = hipMemcpyAsync(d_pairs, pairs, n * sizeof(msgpu_align_pair), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    // MSGPU_ED_DP=1: the anti-diagonal banded DP instead of the furthest-reaching form (same numbers; A/B and cross-check)
    const char *ed_dp  = std::getenv("MSGPU_ED_DP");
    const bool  use_dp = ed_dp && ed_dp[0] == '1';
    if (use_dp)
      hipLaunchKernelGGL(k_edit_distance_dp, dim3(static_cast<uint32_t>((n + 1) / 2)), dim3(128), 0, c->stream,
                         static_cast<const uint8_t *>(d_a), static_cast<const uint8_t *>(d_b),
                         static_cast<const msgpu_align_pair *>(d_pairs), static_cast<uint32_t>(n), band,
                         static_cast<uint32_t *>(d_out));
    else
      hipLaunchKernelGGL(k_edit_distance, dim3(static_cast<uint32_t>((n + 3) / 4)), dim3(256), 0, c->stream,
                         static_cast<const uint8_t *>(d_a), static_cast<const uint8_t *>(d_b),
                         static_cast<const msgpu_align_pair *>(d_pairs), static_cast<uint32_t>(n), band,
                         static_cast<uint32_t *>(d_out));
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return sfail(c, MSGPU_E_HIP, "msgpu_edit_distance", e);
  return MSGPU_OK;
}

} // extern "C"
