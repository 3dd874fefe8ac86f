// asm_internal.h -- the assembly object shared by assemble_path.cpp (host layout) and msgpu_seq.hip (device finish).
#pragma once
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <new>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "msgpu.h"

struct msgpu_assembly {
  msgpu_seqctx                 *ctx = nullptr;
  // msgpu_assembly_set_rows: the MatchMap's VertexMatches.  rows = a copy in input order; row_recs = one record per row
  // grouped by read and, inside a read, ordered by (anchor id, line); the records of read r are
  // row_recs[row_start[r] .. row_start[r + 1])
  struct RowRec {
    uint64_t key; // read id << 32 | anchor id
    uint32_t line, idx; // idx = position in rows
  };
  template <class T> struct RawBuf { // storage that is NOT zero-filled on allocation (hundreds of MB here)
    std::unique_ptr<T[]> p;
    size_t               n = 0;
    void   resize(size_t k) {
      p.reset(k ? new T[k] : nullptr); // default-initialisation: no fill for these plain records
      n = k;
    }
    T       *data() { return p.get(); }
    const T *data() const { return p.get(); }
    const T *begin() const { return p.get(); }
    size_t   size() const { return n; }
    T       &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
  };
  RawBuf<msgpu_row>             rows;      // msgpu_assembly_set_rows: the copy
  const msgpu_row              *rows_view = nullptr; // the installed table: `rows`, or the caller's (msgpu_assembly_borrow_rows)
  RawBuf<RowRec>                row_recs;
  std::vector<uint64_t>         row_start;
  // rows that arrive in ascending anchor-id order (a PAF is grouped by its query) need no per-read table:
  // rows[anchor_start[a] .. anchor_start[a + 1]) are anchor a's, and row_recs / row_start stay empty
  std::vector<uint64_t>         anchor_start;
  bool                          anchor_sorted_by_read = false; // ... and inside every anchor the read ids never decrease: look-ups are binary searches
  std::vector<msgpu_copy>       pieces; // dst_off = position in the raw buffer (records start 16-B aligned)
  uint64_t                      raw_bytes = 0;
  std::vector<msgpu_path_info>  paths;
  std::vector<msgpu_query_info> queries;
  std::string                   paf;                 // temp_1.align.paf
  // filled by msgpu_assembly_finish: both texts live in one pinned host buffer (target text, then query text)
  char    *text = nullptr;
  uint64_t target_fa_len = 0, query_fa_off = 0, query_fa_len = 0;
  bool                          finished = false;
  int                           raw_device = 0;
  void                         *d_raw    = nullptr; // the gathered bases, kept after finish for msgpu_assembly_validate
  void (*release)(msgpu_assembly *) = nullptr;      // frees d_raw (set by msgpu_seq.hip, which owns the HIP calls)
  char                          err[256] = {0};
};

namespace msgpu {
// Host memory for tables that travel to HBM (msgpu_api.hip): page-locked when a HIP device is present, so that the copy
// runs at link speed; plain memory otherwise (parsing without a GPU).  Never null: throws std::bad_alloc.
void *host_table_alloc(size_t bytes);
void  host_table_free(void *p) noexcept;
// ... page-locked once it is filled: the loader's threads touch the block's pages when they write the rows (first touches
// spread over the threads), and locking pages that are there already takes a fraction of a millisecond per 100 MB
void host_table_pin(void *p) noexcept;
// Page-locked host memory for everything else (result tables, text, rings): an anonymous mapping on 2 MiB pages where the
// kernel grants them, first touched on several threads, then registered with the HIP runtime -- a few milliseconds per
// 200 MB where hipHostMalloc takes 30 (it touches every 4 KiB page on the calling thread).  nullptr when out of memory.
void *pinned_block_alloc(size_t bytes) noexcept;
void  pinned_block_free(void *p) noexcept;
template <class T> struct HostTableAlloc { // std::allocator for such tables
  using value_type = T;
  HostTableAlloc() = default;
  template <class U> HostTableAlloc(const HostTableAlloc<U> &) noexcept {}
  T   *allocate(size_t n) { return static_cast<T *>(host_table_alloc(n * sizeof(T))); }
  void deallocate(T *p, size_t) noexcept { host_table_free(p); }
  // resize() leaves new rows of a plain type as they are (the loader's threads fill them): no pass of zeroes over the table
  template <class U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
  template <class U> bool operator==(const HostTableAlloc<U> &) const noexcept { return true; }
  template <class U> bool operator!=(const HostTableAlloc<U> &) const noexcept { return false; }
};
// Where the whitespace-free bytes of a sequence file go while it is parsed (seq_loader.cpp): the loader's own host buffer
// (msgpu_seq_parse) or, through a ring of page-locked slots, straight to a store in HBM (msgpu_seq_parse_upload).  The
// file is parsed in stretches on several threads; stretch k writes ascending positions of [start of the stretch in the
// file, start of the next one).
struct ByteSink {
  virtual void put(uint64_t at, const char *p, size_t n) = 0;
  virtual void done() {} // the stretch is over: nothing more comes
  virtual ~ByteSink() = default;
};
struct SeqDestination {
  virtual void      prepare(uint64_t file_bytes, unsigned n_stretches) = 0; // before the first byte (may throw std::bad_alloc)
  virtual ByteSink *stretch(unsigned k) = 0;            // owned by the destination; used by one thread at a time
  virtual void      restart() = 0;                      // the cuts did not verify: what came so far is void, stretch(0) takes the whole file
  virtual void      fill(uint64_t from, uint64_t to) = 0; // bytes no record uses, between two stretches: they must read as bases
  virtual int       finish(uint64_t extent) = 0;        // everything is in (extent = end of the last record) -> MSGPU_OK or the error met on the way
  virtual ~SeqDestination() = default;
};
// msgpu_seq_parse with the bytes going to dst (nullptr: a host buffer inside the msgpu_seqfile)
int seq_parse_into(const char *path, int is_fastq, SeqDestination *dst, msgpu_seqfile **out);
// std::allocator for host tables that may grow to gigabytes (the loaders' per-chunk tables): large blocks come from
// mappings on 2 MiB pages (first touches and, above all, giving the pages back cost a fraction of what 4 KiB pages do),
// smaller ones from malloc (a small table on a 2 MiB page pays for zeroing the whole page, and malloc hands a block of
// up to 32 MiB that was freed straight to the next caller).  New elements of a plain type
// are left as they are.
template <class T> struct BigTableAlloc {
  using value_type = T;
  static constexpr size_t BIG = size_t(40) << 20; // (above the 32 MiB up to which malloc keeps freed blocks for the next one)
  BigTableAlloc() = default;
  template <class U> BigTableAlloc(const BigTableAlloc<U> &) noexcept {}
  T *allocate(size_t n) {
    if (n * sizeof(T) >= BIG) return static_cast<T *>(host_table_alloc(n * sizeof(T)));
    void *p = std::malloc(n * sizeof(T) ? n * sizeof(T) : 1);
    if (!p) throw std::bad_alloc();
    return static_cast<T *>(p);
  }
  void deallocate(T *p, size_t n) noexcept {
    if (n * sizeof(T) >= BIG) host_table_free(p);
    else std::free(p);
  }
  template <class U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
  template <class U> bool operator==(const BigTableAlloc<U> &) const noexcept { return true; }
  template <class U> bool operator!=(const BigTableAlloc<U> &) const noexcept { return false; }
};
// Toggle::operator* (include/ms/types/Toggle.h:127-153): the product of two toggles is their XNOR
inline bool toggle_mul(bool a, bool b) { return a == b; }
// header lines of the FASTA records, reference spelling (ap.cpp:1035-1040, 1059-1066, 1118-1125, 1175-1182, 1309-1318)
std::string target_header(int32_t asm_idx);
std::string query_header(uint32_t kind, int32_t asm_idx, uint32_t query_idx);
} // namespace msgpu
