// paf_loader.cpp -- host-side PAF loader of libmsgpu (A1 of SURVEY.md section 8).
//
// Replaces BlastFileAccessor::_buildIndex (libms/src/BlastFileAccessor.cpp:77-91) and
// BlastFileReader::read/parseLine (libms/src/BlastFileReader.cpp:72-130) with one pass over an mmap'ed file:
// no per-line job, no file mutex, no istringstream.  Behaviour kept:
//   * every '\n'-terminated line plus a non-empty unterminated tail is a line (libms/src/IO.cpp:54-97);
//   * the LAST line is never parsed (BlastFileReader.cpp:76);
//   * a line is kept iff col9 >= MINIMUM_MATCHES and col3-col2 >= MINIMUM_MATCHES (:106-107);
//   * Registry ids in first-seen order, nanopore id registered before the illumina id (:110-111);
//   * ranges are inclusive (col3-1, col8-1), isPrimary = span >= TH_LENGTH && matches >= TH_MATCHES (:121-122).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <string_view>
#include <system_error>
#include <thread>
#include <vector>

#include "asm_internal.h"
#include "msgpu.h"

namespace {

inline uint64_t name_hash(const char *s, size_t n) { // eight bytes per step (internal: only equal names need equal values)
  uint64_t h = 0x9e3779b97f4a7c15ull ^ (static_cast<uint64_t>(n) * 0xff51afd7ed558ccdull);
  for (; n >= 8; s += 8, n -= 8) {
    uint64_t x;
    memcpy(&x, s, 8);
    h = (h ^ x) * 0x9fb21c651e98df25ull;
    h ^= h >> 29;
  }
  if (n) {
    uint64_t x = 0;
    for (size_t i = 0; i < n; ++i) x |= static_cast<uint64_t>(static_cast<unsigned char>(s[i])) << (8 * i);
    h = (h ^ x) * 0x9fb21c651e98df25ull;
  }
  h ^= h >> 32;
  h *= 0xd6e8feb86659fd93ull;
  return h ^ (h >> 29);
}

// Registry (libms/src/Registry.cpp:36-45): name -> dense id in first-seen order, plus the reverse table.
// Open addressing over precomputed hashes; names are copied into individually allocated strings (stable addresses).
class NameRegistry {
public:
  NameRegistry() : m_slots(1024) {}
  uint32_t get(const char *s, size_t n, uint64_t h) {
    if (m_stale) index_names();
    if ((static_cast<size_t>(size()) + 1) * 2 > m_slots.size()) grow();
    size_t i = h & (m_slots.size() - 1);
    while (m_slots[i].id_plus_1) {
      const Slot &sl = m_slots[i];
      if (sl.hash == h && length(sl.id_plus_1 - 1) == n && memcmp(name(sl.id_plus_1 - 1), s, n) == 0) return sl.id_plus_1 - 1;
      i = (i + 1) & (m_slots.size() - 1);
    }
    m_later.push_back(std::make_unique<std::string>(s, n));
    m_slots[i] = Slot{h, size()};
    return size() - 1;
  }
  uint32_t operator[](std::string_view name) { return get(name.data(), name.size(), name_hash(name.data(), name.size())); }
  uint32_t size() const { return static_cast<uint32_t>(m_bulk + m_later.size()); }
  void     clear() { // Registry::clear (Registry.cpp:47-52): numbering starts again at 0
    m_later.clear();
    std::vector<char>().swap(m_block);
    std::vector<uint64_t>().swap(m_at);
    m_bulk = 0;
    m_slots.assign(1024, Slot{});
    m_stale = false;
  }
  // a look-up touches a slot, the name's place and its characters -- three cache lines that a table of millions of names
  // keeps in none of the caches: callers with many names to look up ask for them a block ahead (stage 0, 1, 2 in turn)
  void want(uint64_t h, int stage) {
    if (m_stale) index_names();
    const Slot &sl = m_slots[h & (m_slots.size() - 1)];
    if (stage == 0) {
      __builtin_prefetch(&sl);
      return;
    }
    if (!sl.id_plus_1 || sl.hash != h) return;
    const uint32_t id = sl.id_plus_1 - 1;
    if (id >= m_bulk) return;
    if (stage == 1) __builtin_prefetch(&m_at[id]);
    else __builtin_prefetch(m_block.data() + m_at[id]);
  }
  // NUL-terminated; stays where it is as long as the registry lives (later names are allocated one by one)
  const char *name(uint32_t id) const {
    return id < m_bulk ? m_block.data() + m_at[id] : id < size() ? m_later[id - m_bulk]->c_str() : nullptr;
  }
  size_t length(uint32_t id) const { return id < m_bulk ? static_cast<size_t>(m_at[id + 1] - m_at[id] - 1) : m_later[id - m_bulk]->size(); }

  // The loader's way in: it knows the id of every name already (first-seen order over the whole file, worked out on its
  // threads).  The names of a file lie in ONE block, id after id, each with its NUL: the loader says how many there are
  // and how many characters, its threads put stretches of consecutive ids in place (put: the id's place in the block is
  // where the id before it ends), and the hash index follows when the first look-up asks for it.  No allocation per name,
  // nothing to free per name.
  void bulk_begin(uint32_t n, uint64_t characters) {
    clear();
    m_bulk = n;
    m_block.resize(characters + n);
    m_at.resize(static_cast<size_t>(n) + 1);
    m_at[n] = characters + n;
    m_stale = true;
  }
  // the name of `id` goes to `at` (= characters + NULs of the ids before it) -> where the next id's name goes
  uint64_t put(uint32_t id, uint64_t at, const char *s, size_t n) {
    m_at[id] = at;
    memcpy(m_block.data() + at, s, n);
    m_block[at + n] = 0;
    return at + n + 1;
  }

private:
  struct Slot {
    uint64_t hash      = 0;
    uint32_t id_plus_1 = 0;
  };
  void place(std::vector<Slot> &into, const Slot &sl) {
    size_t i = sl.hash & (into.size() - 1);
    while (into[i].id_plus_1) i = (i + 1) & (into.size() - 1);
    into[i] = sl;
  }
  void grow() {
    std::vector<Slot> ns(m_slots.size() * 2);
    for (const Slot &sl : m_slots)
      if (sl.id_plus_1) place(ns, sl);
    m_slots.swap(ns);
  }
  void index_names() {
    size_t cap = 1024;
    while (cap < 2 * (static_cast<size_t>(size()) + 1)) cap <<= 1;
    m_slots.assign(cap, Slot{});
    for (uint32_t id = 0; id < size(); ++id) place(m_slots, Slot{name_hash(name(id), length(id)), id + 1});
    m_stale = false;
  }
  std::vector<char>                         m_block; // the bulk-loaded names, id after id, NUL-terminated
  std::vector<uint64_t>                     m_at;    // m_bulk + 1 offsets into m_block
  uint32_t                                  m_bulk = 0;
  std::vector<std::unique_ptr<std::string>> m_later; // ids m_bulk, m_bulk + 1, ...: registered one by one
  std::vector<Slot>                         m_slots;
  bool                                      m_stale = false; // names were bulk-loaded: m_slots is rebuilt by the next get()
};

// std::stoi: optional whitespace, optional sign, at least one digit, value must fit int.
bool parse_int(const char *s, const char *e, int &out) {
  while (s < e && (*s == ' ' || (*s >= '\t' && *s <= '\r'))) ++s;
  bool neg = false;
  if (s < e && (*s == '+' || *s == '-')) neg = *s++ == '-';
  if (s >= e || *s < '0' || *s > '9') return false;
  long long v = 0;
  for (; s < e && *s >= '0' && *s <= '9'; ++s) {
    v = v * 10 + (*s - '0');
    if (v > static_cast<long long>(INT_MAX) + 1) return false;
  }
  if (neg) v = -v;
  if (v > INT_MAX || v < INT_MIN) return false;
  out = static_cast<int>(v);
  return true;
}

} // namespace

struct msgpu_paf {
  std::vector<msgpu_row, msgpu::HostTableAlloc<msgpu_row>> rows; // page-locked when a GPU is present: msgpu_load_rows / msgpu_overlap_batched copy it at link speed
  size_t                 n_lines = 0;
  NameRegistry           reads, anchors;
};

namespace {

// The names of one chunk of the file (one list per registry), in the order the chunk meets them: what the Registry would
// hold had the file begun with this chunk.  The ids of the whole file follow from these lists (assign_ids below).
struct NameEntry {
  const char      *s; // in the chunk's own name block (an offset into it until ChunkNames::bind)
  uint32_t         n;
  uint32_t         id;    // Registry id (set by assign_ids)
  uint64_t         hash;
  const NameEntry *first; // the entry of an earlier chunk with the same name (nullptr: this is the name's first appearance)
};
class ChunkNames {
public:
  void set_shares(unsigned n) { share.assign(n, {}); }
  // position of the name in the chunk's list.  A look-up touches one slot (hash, length, place of the characters) and the
  // characters: two cache lines, wherever the tables have grown to.
  uint32_t get(const char *s, uint32_t n, uint64_t h) {
    if ((list.size() + 1) * 2 > m_slots.size()) grow();
    size_t i = h & (m_slots.size() - 1);
    for (; m_slots[i].at_plus_1; i = (i + 1) & (m_slots.size() - 1)) {
      const Slot &sl = m_slots[i];
      if (sl.hash == h && sl.n == n && memcmp(m_block.data() + sl.off, s, n) == 0) return sl.at_plus_1 - 1;
    }
    if (m_block.size() + n > 0xffffffffull) throw std::bad_alloc(); // (4 GB of distinct names in one chunk)
    const uint32_t off = static_cast<uint32_t>(m_block.size());
    list.push_back(NameEntry{reinterpret_cast<const char *>(static_cast<uintptr_t>(off)), n, 0, h, nullptr});
    m_block.append(s, n); // (the file image is given back as soon as the chunk is parsed)
    m_slots[i] = Slot{h, static_cast<uint32_t>(list.size()), off, n};
    share[share_of(h, static_cast<unsigned>(share.size()))].push_back(static_cast<uint32_t>(list.size() - 1));
    return static_cast<uint32_t>(list.size() - 1);
  }
  // A look-up costs two cache lines (slot, characters), and with hundreds of thousands of names per chunk neither is in
  // a cache: the tokeniser works in blocks of lines and asks for the lines of a whole block ahead of resolving them.
  void want_slot(uint64_t h) const {
    if (!m_slots.empty()) __builtin_prefetch(&m_slots[h & (m_slots.size() - 1)]);
  }
  void want_chars(uint64_t h) const {
    if (m_slots.empty()) return;
    const Slot &sl = m_slots[h & (m_slots.size() - 1)];
    if (sl.at_plus_1 && sl.hash == h) __builtin_prefetch(m_block.data() + sl.off);
  }
  void bind() { // the chunk is parsed: offsets -> pointers, the look-up table goes
    for (NameEntry &e : list) e.s = m_block.data() + reinterpret_cast<uintptr_t>(e.s);
    SlotTable().swap(m_slots);
  }
  static unsigned share_of(uint64_t h, unsigned n) { return static_cast<unsigned>((h >> 40) % n); }
  std::vector<NameEntry, msgpu::BigTableAlloc<NameEntry>>  list;
  std::vector<std::vector<uint32_t>>                        share; // share[t]: the positions of the names thread t looks after (by hash), ascending

private:
  struct Slot {
    uint64_t hash;
    uint32_t at_plus_1, off, n;
  };
  void grow() {
    SlotTable ns(m_slots.empty() ? 4096 : m_slots.size() * 2, Slot{0, 0, 0, 0});
    for (const Slot &v : m_slots)
      if (v.at_plus_1) {
        size_t i = v.hash & (ns.size() - 1);
        while (ns[i].at_plus_1) i = (i + 1) & (ns.size() - 1);
        ns[i] = v;
      }
    m_slots.swap(ns);
  }
  using SlotTable = std::vector<Slot, msgpu::BigTableAlloc<Slot>>;
  SlotTable   m_slots;
  std::string m_block;
};

struct Chunk {
  const char            *begin = nullptr, *end = nullptr;
  size_t                 n_lines = 0, first_line = 0;
  // accepted lines; read_id / anchor_id = positions in the chunk's own name lists.  (Tables of a chunk live in mappings on
  // 2 MiB pages where the kernel grants them: at sizes of gigabytes, first touches and -- above all -- giving 4 KiB pages
  // back one by one cost as much as the tokeniser itself.)
  std::vector<msgpu_row, msgpu::BigTableAlloc<msgpu_row>> rows;
  ChunkNames             reads, anchors;
  int                    err = MSGPU_OK;
  size_t                 err_line = 0;
};

// parse the lines of one chunk (BlastFileReader::parseLine, BlastFileReader.cpp:86-130), except the file's last line
void parse_chunk(Chunk &ch, size_t last_line, const msgpu_params &p) {
  const char *q = ch.begin;
  size_t      li = ch.first_line;
  const char *last_q = nullptr; // the query name of the last accepted line and its place in the chunk's list
  uint32_t    last_qn = 0, last_id = 0;
  ch.rows.reserve(ch.n_lines);
  // Accepted lines wait in a block until their names are resolved: first the block's lines are tokenised and the slots of
  // their names asked for, then the characters behind the slots, then the look-ups run -- on warm lines -- in line order.
  struct Pending {
    msgpu_row   r;
    const char *tname, *qname;
    uint32_t    tn, qn;
    uint64_t    th, qh; // qh = 0: the same query name as the pending line before (no look-up of its own)
  };
  constexpr int BLOCK = 16;
  Pending       pend[BLOCK];
  int           n_pend = 0;
  auto          resolve = [&] {
    for (int k = 0; k < n_pend; ++k) {
      ch.reads.want_chars(pend[k].th);
      if (pend[k].qh) ch.anchors.want_chars(pend[k].qh);
    }
    for (int k = 0; k < n_pend; ++k) {
      Pending &x = pend[k];
      // Registry::operator[] (BlastFileReader.cpp:110-111), chunk-local for now
      x.r.read_id = ch.reads.get(x.tname, x.tn, x.th);
      if (!(last_q && x.qn == last_qn && memcmp(last_q, x.qname, x.qn) == 0)) { // (a PAF is grouped by its query: mostly the line before's)
        last_id = ch.anchors.get(x.qname, x.qn, x.qh ? x.qh : name_hash(x.qname, x.qn));
        last_q  = x.qname;
        last_qn = x.qn;
      }
      x.r.anchor_id = last_id;
      ch.rows.push_back(x.r);
    }
    n_pend = 0;
  };
  for (; q < ch.end; ++li) {
    const void *nlp = memchr(q, '\n', static_cast<size_t>(ch.end - q));
    const char *le  = nlp ? static_cast<const char *>(nlp) : ch.end;
    const char *ls  = q;
    q               = nlp ? le + 1 : ch.end;
    if (li == last_line) break; // BlastFileReader.cpp:76: the last line is never parsed
    // columns 0..9; std::getline drops an empty token after the final delimiter
    const char *tb[10], *te[10];
    int         nt = 0;
    const char *s  = ls;
    for (const char *c = ls;; ++c) {
      if (c == le || *c == '\t') {
        if (c == le && s == le) break;
        tb[nt] = s;
        te[nt] = c;
        if (++nt == 10) break; // (the columns behind the tenth are not read)
        s = c + 1;
        if (c == le) break;
      }
    }
    // the reference checks for >= 9 tokens and then reads tokens[9]; we ask for the 10 it needs
    if (nt < 10) {
      ch.err      = MSGPU_E_FORMAT;
      ch.err_line = li;
      return;
    }
    int irs, ire, nom, nle;
    if (!parse_int(tb[2], te[2], irs) || !parse_int(tb[3], te[3], ire) || !parse_int(tb[9], te[9], nom) ||
        !parse_int(tb[6], te[6], nle) || nom < 0) {
      ch.err      = MSGPU_E_NUMBER;
      ch.err_line = li;
      return;
    }
    const int span = (ire - 1) - irs + 1;
    if (!(static_cast<uint32_t>(nom) >= p.min_matches && span >= static_cast<int>(p.min_matches))) continue;
    int nrs, nre;
    if (!parse_int(tb[7], te[7], nrs) || !parse_int(tb[8], te[8], nre)) {
      ch.err      = MSGPU_E_NUMBER;
      ch.err_line = li;
      return;
    }
    const bool dir  = (te[4] - tb[4] == 1) && *tb[4] == '+';
    const bool prim = span >= static_cast<int>(p.th_length) && static_cast<uint32_t>(nom) >= p.th_matches;
    Pending   &x    = pend[n_pend];
    x.tname         = tb[5];
    x.tn            = static_cast<uint32_t>(te[5] - tb[5]);
    x.th            = name_hash(x.tname, x.tn);
    x.qname         = tb[0];
    x.qn            = static_cast<uint32_t>(te[0] - tb[0]);
    const char *before   = n_pend ? pend[n_pend - 1].qname : last_q;
    const uint32_t before_n = n_pend ? pend[n_pend - 1].qn : last_qn;
    x.qh            = before && before_n == x.qn && memcmp(before, x.qname, x.qn) == 0 ? 0 : name_hash(x.qname, x.qn);
    ch.reads.want_slot(x.th);
    if (x.qh) ch.anchors.want_slot(x.qh);
    x.r.read_len = nle;
    x.r.i_lo     = irs;
    x.r.i_hi     = ire - 1;
    x.r.n_lo     = nrs;
    x.r.n_hi     = nre - 1;
    x.r.score    = static_cast<uint32_t>(nom);
    x.r.line     = static_cast<uint32_t>(li);
    x.r.flags    = (dir ? MSGPU_ROW_DIR : 0u) | (prim ? MSGPU_ROW_PRIMARY : 0u);
    if (++n_pend == BLOCK) resolve();
  }
  resolve();
}

// The line structure of a text (BlastFileAccessor::_buildIndex over readline, BlastFileAccessor.cpp:77-91, IO.cpp:54-97):
// every '\n' ends a line and stays part of it; a non-empty tail without '\n' is a line too.  starts (optional) receives
// the offset of every line relative to `base`.
size_t count_lines(const char *b, const char *e, const char *base = nullptr, uint64_t *starts = nullptr, size_t cap = 0) {
  size_t n = 0;
  for (const char *q = b; q < e;) {
    if (starts && n < cap) starts[n] = static_cast<uint64_t>(q - base);
    ++n;
    const void *nl = memchr(q, '\n', static_cast<size_t>(e - q));
    q              = nl ? static_cast<const char *>(nl) + 1 : e;
  }
  return n;
}

} // namespace

extern "C" {

int msgpu_parse_paf(const char *path, const msgpu_params *params, msgpu_paf **out) {
  if (!path || !out) return MSGPU_E_ARG;
  *out = nullptr;
  msgpu_params p;
  if (params)
    p = *params;
  else
    msgpu_default_params(&p);

  int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return MSGPU_E_IO;
  struct stat st;
  if (fstat(fd, &st) != 0) {
    close(fd);
    return MSGPU_E_IO;
  }
  size_t      len  = static_cast<size_t>(st.st_size);
  const char *data = nullptr;
  if (len) {
    void *m = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
      close(fd);
      return MSGPU_E_IO;
    }
    data = static_cast<const char *>(m);
  }
  close(fd);

  msgpu_paf *paf = nullptr;
  int        rc  = MSGPU_OK;
  try { // no exception may cross the C-ABI
    paf = new msgpu_paf();
    // The reference parses one line per pool job; here the file is cut into one chunk per host thread at line
    // boundaries.  Lines are independent except for the Registry (first-seen ids), which is applied afterwards in
    // line order -- the ids are those of the single-thread reference.
    unsigned nthr = std::thread::hardware_concurrency();
    if (nthr == 0) nthr = 1;
    if (nthr > 16) nthr = 16;
    if (const char *e = getenv("MSGPU_PARSE_THREADS")) // (tests: any number of chunks at any size)
      nthr = static_cast<unsigned>(std::min<size_t>(std::max(1, atoi(e)), std::max<size_t>(1, len)));
    else
      while (nthr > 1 && len / nthr < (1u << 20)) --nthr; // at least 1 MiB per thread
    std::vector<Chunk> chunks(nthr);
    const char        *end = data + len, *cur = data;
    for (unsigned t = 0; t < nthr; ++t) {
      chunks[t].begin = cur;
      const char *cut = (t + 1 == nthr) ? end : data + (len / nthr) * (t + 1);
      if (cut < cur) cut = cur;
      if (cut < end) {
        const void *nl = memchr(cut, '\n', static_cast<size_t>(end - cut));
        cut            = nl ? static_cast<const char *>(nl) + 1 : end;
      }
      chunks[t].end = cut;
      cur           = cut;
    }
    auto run = [&](auto &&fn) {
      std::vector<std::thread> th;
      struct JoinAll { // (a thread that could not be started must not leave the started ones unjoined behind the exception)
        std::vector<std::thread> &t;
        ~JoinAll() {
          for (auto &x : t)
            if (x.joinable()) x.join();
        }
      } join_all{th};
      for (unsigned t = 1; t < nthr; ++t) th.emplace_back(fn, t);
      fn(0u);
      for (auto &x : th) x.join();
    };
    const bool dbg = getenv("MSGPU_PARSE_DEBUG") != nullptr;
    auto       t0  = std::chrono::steady_clock::now();
    auto       lap = [&](const char *what) {
      if (!dbg) return;
      auto t1 = std::chrono::steady_clock::now();
      fprintf(stderr, "msgpu_parse_paf: %-10s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
      t0 = t1;
    };
    run([&](unsigned t) { chunks[t].n_lines = count_lines(chunks[t].begin, chunks[t].end); });
    lap("count");
    size_t n_lines = 0;
    for (auto &c : chunks) {
      c.first_line = n_lines;
      n_lines += c.n_lines;
    }
    paf->n_lines          = n_lines;
    const size_t last     = n_lines ? n_lines - 1 : 0;
    std::vector<int> oom(nthr, 0);
    const size_t page = static_cast<size_t>(sysconf(_SC_PAGESIZE));
    run([&](unsigned t) {
      try {
        Chunk &c = chunks[t];
        c.reads.set_shares(nthr);
        c.anchors.set_shares(nthr);
        if (n_lines) parse_chunk(c, last, p);
        c.reads.bind();
        c.anchors.bind();
        // this chunk's pages leave the mapping here, on this thread (the final munmap would do all of them on one)
        const size_t lo = (static_cast<size_t>(c.begin - data) + page - 1) / page * page, hi = static_cast<size_t>(c.end - data) / page * page;
        if (hi > lo) madvise(const_cast<char *>(data) + lo, hi - lo, MADV_DONTNEED);
      } catch (std::bad_alloc const &) { oom[t] = 1; }
    });
    lap("tokenise");
    for (unsigned t = 0; t < nthr; ++t)
      if (oom[t]) throw std::bad_alloc();
    // the reference stops at the first bad line in line order
    for (auto &c : chunks)
      if (c.err != MSGPU_OK) {
        rc = c.err;
        break;
      }
    if (rc == MSGPU_OK) {
      // Registry ids = first-seen order over the whole file, line by line (:110-111), from the chunks' own lists: a name's
      // id is its rank among the names' first appearances, and the first appearances of the file are, chunk after chunk,
      // the names of that chunk's list no earlier chunk holds.
      std::vector<int> bad(nthr, 0);
      auto             guarded = [&](auto &&fn) {
        run([&](unsigned t) {
          try {
            fn(t);
          } catch (std::bad_alloc const &) { bad[t] = 1; }
        });
        for (int o : bad)
          if (o) throw std::bad_alloc();
      };
      for (int which = 0; which < 2; ++which) {
        auto          names_of = [&](Chunk &c) -> ChunkNames          &{ return which == 0 ? c.reads : c.anchors; };
        NameRegistry &reg = which == 0 ? paf->reads : paf->anchors;
        size_t        n_entries = 0;
        for (auto &c : chunks) n_entries += names_of(c).list.size();
        // (1) which entries are first appearances: thread t owns the names whose hash falls into its share and walks the
        // lists in file order with an open-addressing table of its own
        guarded([&](unsigned t) {
          size_t cap = 1024;
          while (cap < 4 * (n_entries / nthr + 1)) cap <<= 1;
          std::vector<NameEntry *> slot(cap, nullptr);
          size_t                   used = 0;
          for (unsigned k = 0; k < nthr; ++k)
            for (uint32_t at : names_of(chunks[k]).share[t]) {
              NameEntry &e = names_of(chunks[k]).list[at];
              if ((used + 1) * 2 > cap) { // (only a very uneven hash gets here)
                std::vector<NameEntry *> ns(cap * 2, nullptr);
                for (NameEntry *x : slot)
                  if (x) {
                    size_t i = x->hash & (cap * 2 - 1);
                    while (ns[i]) i = (i + 1) & (cap * 2 - 1);
                    ns[i] = x;
                  }
                slot.swap(ns);
                cap *= 2;
              }
              size_t i = e.hash & (cap - 1);
              for (; slot[i]; i = (i + 1) & (cap - 1)) {
                const NameEntry *x = slot[i];
                if (x->hash == e.hash && x->n == e.n && memcmp(x->s, e.s, e.n) == 0) break;
              }
              if (slot[i]) {
                e.first = slot[i];
              } else {
                slot[i] = &e;
                ++used;
              }
            }
        });
        // (2) ids of the first appearances, chunk after chunk; (3) every other entry takes the id of its first appearance
        std::vector<uint32_t> base(nthr + 1, 0);
        std::vector<uint64_t> at(nthr + 1, 0); // where a chunk's first appearances start in the registry's block of names
        for (unsigned k = 0; k < nthr; ++k) {
          uint32_t firsts = 0;
          uint64_t chars  = 0;
          for (const NameEntry &e : names_of(chunks[k]).list)
            if (e.first == nullptr) {
              ++firsts;
              chars += e.n + 1;
            }
          base[k + 1] = base[k] + firsts;
          at[k + 1]   = at[k] + chars;
        }
        reg.bulk_begin(base[nthr], at[nthr] - base[nthr]);
        guarded([&](unsigned k) {
          uint32_t id = base[k];
          uint64_t to = at[k];
          for (NameEntry &e : names_of(chunks[k]).list)
            if (!e.first) {
              e.id = id++;
              to   = reg.put(e.id, to, e.s, e.n);
            }
        });
        guarded([&](unsigned k) {
          for (NameEntry &e : names_of(chunks[k]).list)
            if (e.first) e.id = e.first->id;
        });
      }
      lap("registry");
      // the rows of the file, chunk after chunk, with the file's ids
      std::vector<size_t> row_base(nthr + 1, 0);
      for (unsigned k = 0; k < nthr; ++k) row_base[k + 1] = row_base[k] + chunks[k].rows.size();
      paf->rows.resize(row_base[nthr]);
      guarded([&](unsigned k) {
        const NameEntry *rd = chunks[k].reads.list.data(), *an = chunks[k].anchors.list.data();
        msgpu_row       *o  = paf->rows.data() + row_base[k];
        for (const msgpu_row &r : chunks[k].rows) {
          *o           = r;
          o->read_id   = rd[r.read_id].id;
          o->anchor_id = an[r.anchor_id].id;
          ++o;
        }
      });
      // (the table's pages were first touched by the threads above, each its own stretch: locking them for the copy to
      // HBM -- a GPU present -- is a fraction of a millisecond now)
      if (!paf->rows.empty()) msgpu::host_table_pin(paf->rows.data());
      lap("rows");
      // (what the chunks held goes back on the threads too: at sizes of gigabytes, handing their row and name tables back one
      // after the other at the end of this function took longer than counting the file's lines)
      run([&](unsigned t) {
        Chunk &c = chunks[t];
        std::vector<msgpu_row, msgpu::BigTableAlloc<msgpu_row>>().swap(c.rows);
        c.reads   = ChunkNames();
        c.anchors = ChunkNames();
      });
      lap("release");
    }
  } catch (std::bad_alloc const &) { rc = MSGPU_E_NOMEM; } catch (std::system_error const &) {
    rc = MSGPU_E_NOMEM; // could not start a thread
  }
  if (data) munmap(const_cast<char *>(data), len);
  if (rc != MSGPU_OK) {
    delete paf;
    return rc;
  }
  *out = paf;
  return MSGPU_OK;
}

void msgpu_paf_free(msgpu_paf *paf) { delete paf; }

// Registry::operator[] for every record of a sequence file, on the PAF's own registries (SequenceAccessor.cpp:171,215
// call the Registry objects BlastFileReader filled, src/main.cpp:149-163)
int msgpu_paf_register_sequences(msgpu_paf *paf, int kind, const msgpu_seqfile *f, uint32_t *ids, uint32_t *id_space) {
  if (!paf || !f || (kind != 0 && kind != 1)) return MSGPU_E_ARG;
  const uint32_t n = msgpu_seq_count(f);
  if (n && !ids) return MSGPU_E_ARG;
  try {
    NameRegistry &reg = kind == 0 ? paf->reads : paf->anchors;
    constexpr uint32_t BLOCK = 16; // (the look-ups of a block of records are asked for before they are made)
    for (uint32_t b = 0; b < n; b += BLOCK) {
      const uint32_t   e = std::min(n, b + BLOCK);
      std::string_view name[BLOCK];
      uint64_t         h[BLOCK];
      for (uint32_t i = b; i < e; ++i) {
        name[i - b] = std::string_view(msgpu_seq_name(f, i));
        h[i - b]    = name_hash(name[i - b].data(), name[i - b].size());
        reg.want(h[i - b], 0);
      }
      for (int stage = 1; stage <= 2; ++stage)
        for (uint32_t i = b; i < e; ++i) reg.want(h[i - b], stage);
      for (uint32_t i = b; i < e; ++i) ids[i] = reg.get(name[i - b].data(), name[i - b].size(), h[i - b]);
    }
    if (id_space) *id_space = reg.size();
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
  return MSGPU_OK;
}

const msgpu_row *msgpu_paf_rows(const msgpu_paf *paf, size_t *n_rows) {
  if (n_rows) *n_rows = paf ? paf->rows.size() : 0;
  return paf && !paf->rows.empty() ? paf->rows.data() : nullptr;
}
size_t      msgpu_paf_line_count(const msgpu_paf *paf) { return paf ? paf->n_lines : 0; }
uint32_t    msgpu_paf_read_count(const msgpu_paf *paf) { return paf ? paf->reads.size() : 0; }
uint32_t    msgpu_paf_anchor_count(const msgpu_paf *paf) { return paf ? paf->anchors.size() : 0; }
const char *msgpu_paf_read_name(const msgpu_paf *paf, uint32_t id) { return paf ? paf->reads.name(id) : nullptr; }
const char *msgpu_paf_anchor_name(const msgpu_paf *paf, uint32_t id) { return paf ? paf->anchors.name(id) : nullptr; }


/* ---- host utilities of libms the path relies on, at the boundary (vectors of the reference's unit tests) ------------ */

/* BlastFileAccessor::_buildIndex (BlastFileAccessor.cpp:77-91) over readline (IO.cpp:54-97): the line starts of a file,
 * by the same routine msgpu_parse_paf counts its lines with. */
int msgpu_index_lines(const char *path, uint64_t *offsets, size_t capacity, size_t *n_lines) {
  if (!path || !n_lines || (capacity && !offsets)) return MSGPU_E_ARG;
  int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return MSGPU_E_IO;
  struct stat st;
  if (fstat(fd, &st) != 0) {
    close(fd);
    return MSGPU_E_IO;
  }
  const size_t len  = static_cast<size_t>(st.st_size);
  const char  *data = nullptr;
  if (len) {
    void *m = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
      close(fd);
      return MSGPU_E_IO;
    }
    data = static_cast<const char *>(m);
  }
  close(fd);
  const size_t n = count_lines(data, data + len, data, offsets, capacity);
  if (offsets && n < capacity) offsets[n] = len; // end of the last line
  *n_lines = n;
  if (data) munmap(const_cast<char *>(data), len);
  return MSGPU_OK;
}

struct msgpu_registry {
  NameRegistry r;
};

msgpu_registry *msgpu_registry_new(void) {
  try {
    return new msgpu_registry();
  } catch (std::bad_alloc const &) { return nullptr; }
}
void msgpu_registry_free(msgpu_registry *r) { delete r; }
/* Registry::operator[] (Registry.cpp:36-45): the id of `name`, the next free id when it is new; 0xffffffff on error */
uint32_t msgpu_registry_id(msgpu_registry *r, const char *name) {
  if (!r || !name) return 0xffffffffu;
  try {
    return r->r[std::string_view(name)];
  } catch (std::bad_alloc const &) { return 0xffffffffu; }
}
uint32_t msgpu_registry_size(const msgpu_registry *r) { return r ? r->r.size() : 0; }
void     msgpu_registry_clear(msgpu_registry *r) {
  if (r) r->r.clear();
}

/* Toggle::operator* / operator*= (include/ms/types/Toggle.h:127-153): XNOR */
int msgpu_toggle_mul(int a, int b) { return msgpu::toggle_mul(a != 0, b != 0) ? 1 : 0; }

} // extern "C"
