// Device-side table builders: plain per-item arrays in HBM (in the caller's row order) -> the tables the grid
// kernels read, derived columns, sorts and the category partition included (include/nsm_hip.h, "Builders").
//
// The reference does the equivalent per-item preparation in Python, once per PAIR (set(...) / join_sorted in
// compare/score_functions.py:10-11,16-17,24-25); round 1 of this build did it once per item in numpy
// (napkon_string_matching_amd/tables.py) -- 37 s for configs[4]'s 1.5 M items against 1 s of grid kernels,
// and a host in another language had to restate the hashing.  Here it is a handful of streaming kernels
// (HBM-bound: every input byte is read once or twice) plus one stable radix sort of (key, row) pairs and one
// scan (rocPRIM -- a library sort is not the hot path).
//
// The builders SORT, so a caller cannot hand the grids an unsorted table (round-1 verdict: the grids take
// lane 0 of a wavefront as its largest row and were silently wrong otherwise).  They synchronise `stream`
// before they return: the row count of a partitioned table and the validation verdict live on the device.
#include <cstring>
#include <type_traits>

#include <rocprim/rocprim.hpp>

#include "nsm_common.hpp"

namespace nsm {
namespace {

constexpr int kEmptyCategoryBit = 63;  // stands for "no category at all" when empty-vs-empty counts as a match
constexpr int kThreads = 256;

enum BuildError : int {
  kErrNone = 0,
  kErrRowTooWide = 1,    // more valid ids than the table's width
  kErrDuplicateId = 2,   // the same id twice in one row
  kErrBadCode = 3,       // code unit above the alphabet / length outside [0, stride]
  kErrBit63 = 4,         // category bit 63 in use: the empty items cannot become a category of their own
  kErrBadLevels = 5,     // nlev outside [1, max_levels] / plen not non-decreasing or above cnt
  kErrBadVocab = 6,      // an id >= the table's vocab (global inverted index)
};

struct Status {
  int err;
  int rows;
};

struct Scratch {  // stream-ordered allocations of one builder call
  hipStream_t stream;
  void* ptr[48];
  int n = 0;
  bool failed = false;
  explicit Scratch(hipStream_t s) : stream(s) {}
  template <typename T>
  T* get(size_t count) {
    void* p = nullptr;
    if (n >= 48 || hipMallocAsync(&p, (count ? count : 1) * sizeof(T), stream) != hipSuccess) {
      failed = true;
      return nullptr;
    }
    ptr[n++] = p;
    return static_cast<T*>(p);
  }
  ~Scratch() {
    for (int k = 0; k < n; ++k) (void)hipFreeAsync(ptr[k], stream);
  }
};

inline dim3 blocks_for(long long n) { return dim3(static_cast<unsigned>((n + kThreads - 1) / kThreads > 0 ? (n + kThreads - 1) / kThreads : 1)); }

__global__ void iota_kernel(int32_t* v, int n) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i < n) v[i] = i;
}

// perm = row indices ordered by key ascending, ties in row order (stable)
int sort_rows(const uint32_t* keys, int n, int key_bits, int32_t* perm, Scratch& sc) {
  if (n <= 0) return 0;
  uint32_t* keys_out = sc.get<uint32_t>(n);
  int32_t* iota = sc.get<int32_t>(n);
  if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
  hipLaunchKernelGGL(iota_kernel, blocks_for(n), dim3(kThreads), 0, sc.stream, iota, n);
  size_t bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, keys, keys_out, iota, perm, static_cast<size_t>(n), 0u,
                                           static_cast<unsigned>(key_bits), sc.stream);
  if (e != hipSuccess) return hip_status(e, "radix_sort_pairs (size)");
  char* temp = sc.get<char>(bytes);
  if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
  e = rocprim::radix_sort_pairs(temp, bytes, keys, keys_out, iota, perm, static_cast<size_t>(n), 0u,
                                static_cast<unsigned>(key_bits), sc.stream);
  return hip_status(e, "radix_sort_pairs");
}

int exclusive_scan_i32(const int32_t* in, int32_t* out, int n, Scratch& sc) {
  if (n <= 0) return 0;
  size_t bytes = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, bytes, in, out, 0, static_cast<size_t>(n), rocprim::plus<int32_t>(), sc.stream);
  if (e != hipSuccess) return hip_status(e, "exclusive_scan (size)");
  char* temp = sc.get<char>(bytes);
  if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
  e = rocprim::exclusive_scan(temp, bytes, in, out, 0, static_cast<size_t>(n), rocprim::plus<int32_t>(), sc.stream);
  return hip_status(e, "exclusive_scan");
}

// ---------------------------------------------------------------------------------------------------- categories
// cat' = the mask the partition works with: "both empty also matches" turns the empty items into one more
// category (bit 63), which then must be free.  rows[i] = rows item i expands to.
__global__ void category_rows_kernel(const uint64_t* __restrict__ cat_in, int n, int both_empty, int partition,
                                     uint64_t* __restrict__ cat_out, int32_t* __restrict__ rows, Status* st) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  uint64_t c = cat_in ? cat_in[i] : 0ull;
  if (partition && both_empty) {
    if (c >> kEmptyCategoryBit) atomicMax(&st->err, static_cast<int>(kErrBit63));
    if (c == 0ull) c = 1ull << kEmptyCategoryBit;
  }
  if (cat_out) cat_out[i] = c;
  rows[i] = partition ? __popcll(c) : 1;
}

// expanded row -> (item, category): item i owns rows [off[i], off[i] + popcount(cat'[i])), categories ascending
__global__ void expand_kernel(const uint64_t* __restrict__ cat, const int32_t* __restrict__ off, int n, int partition,
                              int32_t* __restrict__ item_of, int32_t* __restrict__ seg_of) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  int at = off[i];
  if (!partition) {
    item_of[at] = i;
    seg_of[at] = 0;
    return;
  }
  for (uint64_t c = cat[i]; c; c &= c - 1) {
    item_of[at] = i;
    seg_of[at] = __builtin_ctzll(c);
    ++at;
  }
}

__global__ void make_keys_kernel(const int32_t* __restrict__ item_of, const int32_t* __restrict__ seg_of,
                                 const int32_t* __restrict__ small_first, int rows, int radix, uint32_t* __restrict__ keys) {
  const int r = blockIdx.x * kThreads + threadIdx.x;
  if (r >= rows) return;
  keys[r] = static_cast<uint32_t>(seg_of[r] * radix + small_first[item_of[r]]);
}

// start[c] = number of rows whose class is < c, for c in [0, n_classes]: in a table sorted by class that is the
// first row of class c.  Block-private LDS histogram, one global atomic per (block, class), then a tiny scan.
__global__ void class_hist_kernel(const uint32_t* __restrict__ cls, int rows, int n_classes, int32_t* __restrict__ counts) {
  __shared__ int s_h[520];
  for (int c = threadIdx.x; c < n_classes; c += kThreads) s_h[c] = 0;
  __syncthreads();
  const int r = blockIdx.x * kThreads + threadIdx.x;
  if (r < rows) atomicAdd(&s_h[min(static_cast<int>(cls[r]), n_classes - 1)], 1);
  __syncthreads();
  for (int c = threadIdx.x; c < n_classes; c += kThreads)
    if (s_h[c]) atomicAdd(&counts[c], s_h[c]);
}

__global__ void class_scan_kernel(const int32_t* __restrict__ counts, int n_classes, int32_t* __restrict__ start) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int at = 0;
    for (int c = 0; c < n_classes; ++c) {
      start[c] = at;
      at += counts[c];
    }
    start[n_classes] = at;
  }
}

__global__ void widen_kernel(const int32_t* __restrict__ in, int n, uint32_t* __restrict__ out) {
  const int r = blockIdx.x * kThreads + threadIdx.x;
  if (r < n) out[r] = static_cast<uint32_t>(in[r]);
}

// ---------------------------------------------------------------------------------------------------- set tables
__device__ __forceinline__ uint64_t signature_word(const int32_t* ids, int cnt, uint32_t multiplier) {
  uint64_t word = 0;
  for (int k = 0; k < cnt; ++k) {
    const uint32_t h16 = ((static_cast<uint32_t>(ids[k]) * multiplier) >> 16) & 0xffffu;
    word |= 1ull << ((static_cast<uint64_t>(h16) * 58ull) >> 16);
  }
  const int extra = cnt - __popcll(word);  // ids that collided inside the row
  if (extra > 6) return ~0ull;
  return word | (((1ull << extra) - 1ull) << 58);
}

// per input item: packed ids (valid first, input order kept), count, validation; the sort's size class
__global__ void set_rows_kernel(const int32_t* __restrict__ ids_in, int n, int width_in, int width, int validate,
                                const int32_t* __restrict__ nlev_in, const uint8_t* __restrict__ plen_in, int max_levels,
                                int32_t* __restrict__ ids_tmp, int32_t* __restrict__ cnt_tmp, int32_t* __restrict__ small_first,
                                Status* st) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const int32_t* src = ids_in + static_cast<size_t>(i) * width_in;
  int32_t* dst = ids_tmp + static_cast<size_t>(i) * width;
  // The operand is a SET (the reference builds set(...) per pair, compare/score_functions.py:10-11): an id that
  // occurs twice counts once.  RAW rows drop the repeats here; a levels row cannot (its prefix lengths count the
  // caller's slots), so there a repeat is a data error -- always checked: a table with a repeated id would
  // over-count |A n B| in the position matrix and in the inverted index (round-2 advice; `validate` is kept in the
  // signature for ABI v2 callers and no longer changes anything).
  (void)validate;
  int cnt = 0;
  bool too_wide = false;
  for (int k = 0; k < width_in; ++k) {
    const int32_t v = src[k];
    if (v < 0) continue;
    bool seen = false;
    for (int b = 0; b < min(cnt, width); ++b) seen = seen || dst[b] == v;
    if (seen) {
      if (nlev_in) atomicMax(&st->err, static_cast<int>(kErrDuplicateId));
      else continue;
    }
    if (cnt < width) dst[cnt] = v;
    else too_wide = true;
    ++cnt;
  }
  if (too_wide) {
    atomicMax(&st->err, static_cast<int>(kErrRowTooWide));
    cnt = width;
  }
  if (!nlev_in) {
    // RAW rows: ids ascending (include/nsm_hip.h) -- ascending id is the global token order of the inverted index, and the
    // prefix of a row in that order is its first few slots.  Insertion sort: rows hold <= 64 ids.
    for (int a = 1; a < cnt; ++a) {
      const int32_t v = dst[a];
      int b = a - 1;
      while (b >= 0 && dst[b] > v) {
        dst[b + 1] = dst[b];
        --b;
      }
      dst[b + 1] = v;
    }
  }
  if (nlev_in) {
    const int L = nlev_in[i];
    if (L < 1 || L > max_levels) atomicMax(&st->err, static_cast<int>(kErrBadLevels));
    else {
      const uint8_t* pl = plen_in + static_cast<size_t>(i) * max_levels;
      for (int l = 0; l < L; ++l)
        if (pl[l] > cnt || (l > 0 && pl[l] < pl[l - 1])) atomicMax(&st->err, static_cast<int>(kErrBadLevels));
    }
  }
  cnt_tmp[i] = cnt;
  small_first[i] = width - cnt;  // larger sets first
}

// per output row: every column of the table
__global__ void set_gather_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ item_of,
                                  const int32_t* __restrict__ seg_of, int rows, int width, int pad,
                                  const int32_t* __restrict__ ids_tmp, const int32_t* __restrict__ cnt_tmp,
                                  const int32_t* __restrict__ orig_in, const int32_t* __restrict__ nlev_in,
                                  const uint8_t* __restrict__ plen_in, int max_levels, const uint64_t* __restrict__ cat,
                                  int partition, int32_t* __restrict__ ids, int32_t* __restrict__ cnt_out,
                                  uint64_t* __restrict__ sig, uint64_t* __restrict__ sig2, int32_t* __restrict__ orig,
                                  int32_t* __restrict__ nlev, uint8_t* __restrict__ plen, uint64_t* __restrict__ cat_out,
                                  uint32_t* __restrict__ filt, int32_t* __restrict__ seg, uint32_t* __restrict__ size_class) {
  const int r = blockIdx.x * kThreads + threadIdx.x;
  if (r >= rows) return;
  const int e = perm[r];
  const int i = item_of[e];
  const int32_t* src = ids_tmp + static_cast<size_t>(i) * width;
  int32_t* dst = ids + static_cast<size_t>(r) * width;
  const int cnt = cnt_tmp[i];
  for (int k = 0; k < width; ++k) dst[k] = k < cnt ? src[k] : pad;
  cnt_out[r] = cnt;
  const uint64_t s1 = signature_word(src, cnt, 0x9E3779B1u);
  sig[r] = s1;
  if (sig2) sig2[r] = signature_word(src, cnt, 0xC2B2AE35u);
  orig[r] = orig_in ? orig_in[i] : i;
  size_class[r] = static_cast<uint32_t>(width - cnt);
  if (partition && seg) seg[r] = seg_of[e];
  if (nlev_in) {
    const int L = nlev_in[i];
    nlev[r] = L;
    const uint8_t* pl = plen_in + static_cast<size_t>(i) * max_levels;
    uint8_t* po = plen + static_cast<size_t>(r) * max_levels;
    for (int l = 0; l < max_levels; ++l) po[l] = pl[min(l, max(L, 1) - 1)];  // levels past the last repeat it
    const uint64_t c = cat ? cat[i] : 0ull;
    if (cat_out) cat_out[r] = c;
    const int plen1 = pl[min(min(1, max_levels - 1), max(L, 1) - 1)];  // the set every step of compare_terms contains
    const uint64_t sl1 = signature_word(src, min(plen1, cnt), 0x9E3779B1u);
    uint32_t* f = filt + static_cast<size_t>(r) * 8;
    f[0] = static_cast<uint32_t>(s1);
    f[1] = static_cast<uint32_t>(s1 >> 32);
    f[2] = static_cast<uint32_t>(c);
    f[3] = static_cast<uint32_t>(c >> 32);
    f[4] = static_cast<uint32_t>(plen1) | (static_cast<uint32_t>(cnt) << 8) | (static_cast<uint32_t>(L) << 16);
    f[5] = static_cast<uint32_t>(sl1);
    f[6] = static_cast<uint32_t>(sl1 >> 32);
    f[7] = 0u;
  }
}

// ---------------------------------------------------------------------------------------------------- global inverted index
// (RAW tables; include/nsm_hip.h: post / post_start).  One (key, entry) pair per table slot, key = id << 8 | position
// (unused slots: all ones, sorted last), one radix sort, then the offsets: 5 boundaries per id (positions < 1, 2, 4, 8, any).
__device__ __forceinline__ int post_class(int pos) { return pos < 1 ? 0 : pos < 2 ? 1 : pos < 4 ? 2 : pos < 8 ? 3 : 4; }

// FORMAT 0: entry = row | position << 32 | cnt << 40 (64 bits); 1: the compact entry row | position << row_bits |
// (cnt - 1) << (row_bits + log2 width) (32 bits); 2: the compact entry with the 27-bit fold of the row's signature word above
// it (64 bits; nsm_hip.h: post)
__device__ __forceinline__ uint32_t sig_fold27(uint64_t sig) {
  constexpr uint32_t kM = (1u << 27) - 1u;
  return (static_cast<uint32_t>(sig) & kM) | (static_cast<uint32_t>(sig >> 27) & kM) | (static_cast<uint32_t>(sig >> 54) & 0xFu);
}

template <typename VAL, int FORMAT>
__global__ void post_keys_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ cnt, const int32_t* __restrict__ seg,
                                 const uint64_t* __restrict__ sig, int rows, int width, int vocab, int row_bits, int width_log,
                                 unsigned long long* __restrict__ keys, VAL* __restrict__ vals, Status* st) {
  const long long s = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (s >= static_cast<long long>(rows) * width) return;
  const int r = static_cast<int>(s / width), k = static_cast<int>(s % width);
  const int c = cnt[r];
  unsigned long long key = ~0ull;
  VAL val = 0;
  if (k < c) {
    const int32_t id = ids[s];
    if (id >= vocab) atomicMax(&st->err, static_cast<int>(kErrBadVocab));
    else {
      // (a partitioned levels table: one key space per category segment, so a probe only meets rows of its own category)
      const unsigned long long tok = (seg ? static_cast<unsigned long long>(seg[r]) * static_cast<unsigned long long>(vocab) : 0ull) +
                                     static_cast<unsigned long long>(static_cast<uint32_t>(id));
      key = (tok << 8) | static_cast<unsigned long long>(k);
      const uint32_t compact = static_cast<uint32_t>(r) | (static_cast<uint32_t>(k) << row_bits) |
                               (static_cast<uint32_t>(c - 1) << (row_bits + width_log));
      if constexpr (FORMAT == 0)
        val = static_cast<unsigned long long>(static_cast<uint32_t>(r)) | (static_cast<unsigned long long>(k) << 32) |
              (static_cast<unsigned long long>(c) << 40);
      else if constexpr (FORMAT == 1) val = compact;
      else val = static_cast<unsigned long long>(compact) | (static_cast<unsigned long long>(sig_fold27(sig[r])) << 32);
    }
  }
  keys[s] = key;
  vals[s] = val;
}

// post_start[q] = index of the first sorted entry whose (id * 5 + class) is >= q, for q in [0, 5 vocab] = the number of
// entries whose class key is below q.  The last entry e of every run of equal class keys k writes e + 1 to slot k + 1 of the
// zeroed array; a forward MAX-scan then fills the slots of the empty (id, class) lists -- most of the key space of a
// partitioned levels table, where one thread walking a gap took 41 ms per table (profiles/pmc_c5.json of round 4).
template <typename VAL>
__global__ void post_bounds_kernel(const unsigned long long* __restrict__ keys, long long n, int vocab, unsigned long long key_mask,
                                   int32_t* __restrict__ post_start, VAL* __restrict__ vals) {
  const long long e = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (e >= n) return;
  const long long sentinel = 5ll * vocab;
  auto ckey = [&](long long at) -> long long {
    if (at >= n) return sentinel;
    const unsigned long long k = keys[at];
    if ((k & key_mask) == key_mask) return sentinel;  // an unused slot (all ones in the sorted bits)
    return static_cast<long long>(k >> 8) * 5 + post_class(static_cast<int>(k & 0xffu));
  };
  const long long cur = ckey(e);
  if (cur == sentinel) {
    vals[e] = 0;  // (the tail stays zero whatever the sort left there)
    return;
  }
  if (ckey(e + 1) != cur) post_start[cur + 1] = static_cast<int32_t>(e + 1);
}

__global__ void post_stats_kernel(const int32_t* __restrict__ post_start, int vocab, unsigned long long* __restrict__ sq) {
  __shared__ unsigned long long s_sq[5];
  if (threadIdx.x < 5) s_sq[threadIdx.x] = 0ull;
  __syncthreads();
  const int t = blockIdx.x * kThreads + threadIdx.x;
  if (t < vocab) {
    const long long base = post_start[5ll * t];
    for (int c = 0; c < 5; ++c) {
      const unsigned long long len = static_cast<unsigned long long>(post_start[5ll * t + c + 1] - base);
      if (len) atomicAdd(&s_sq[c], len * len);
    }
  }
  __syncthreads();
  if (threadIdx.x < 5 && s_sq[threadIdx.x]) atomicAdd(&sq[threadIdx.x], s_sq[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------- string tables
__global__ void str_keys_kernel(const int32_t* __restrict__ len_in, int n, int stride, uint32_t* __restrict__ keys, Status* st) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  int len = len_in[i];
  if (len < 0 || len > stride) {
    atomicMax(&st->err, static_cast<int>(kErrBadCode));
    len = max(0, min(len, stride));
  }
  keys[i] = static_cast<uint32_t>(stride - len);  // longer strings first
}

// per output row: codes (padding rewritten), length, caller id, 32-bucket histogram (counts saturate at 255)
__global__ void str_gather_kernel(const int32_t* __restrict__ perm, int n, int stride, int alphabet,
                                  const uint8_t* __restrict__ codes_in, const int32_t* __restrict__ len_in,
                                  const int32_t* __restrict__ orig_in, uint8_t* __restrict__ codes, int32_t* __restrict__ len,
                                  int32_t* __restrict__ orig, uint8_t* __restrict__ hist, uint8_t* __restrict__ hist16,
                                  uint32_t* __restrict__ len_class, Status* st) {
  __shared__ uint8_t s_hist[kThreads][36];  // 36-byte rows: the threads of a wavefront spread over the banks
  const int r = blockIdx.x * kThreads + threadIdx.x;
  if (r >= n) return;
  const int i = perm ? perm[r] : r;
  const int L = max(0, min(len_in[i], stride));
  uint8_t* h = s_hist[threadIdx.x];
  for (int b = 0; b < 32; ++b) h[b] = 0;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(codes_in + static_cast<size_t>(i) * stride);
  uint32_t* dst = reinterpret_cast<uint32_t*>(codes + static_cast<size_t>(r) * stride);
  const uint32_t padw = static_cast<uint32_t>(alphabet) * 0x01010101u;
  bool bad = false;
  for (int w = 0; w < stride / 4; ++w) {
    uint32_t v = src[w], out = padw;
    for (int b = 0; b < 4; ++b) {
      const int pos = 4 * w + b;
      if (pos < L) {
        const uint32_t c = (v >> (8 * b)) & 0xffu;
        bad = bad || c >= static_cast<uint32_t>(alphabet);
        out = (out & ~(0xffu << (8 * b))) | (c << (8 * b));
        uint8_t& slot = h[c & 31u];
        if (slot != 255) ++slot;
      }
    }
    dst[w] = out;
  }
  if (bad) atomicMax(&st->err, static_cast<int>(kErrBadCode));
  len[r] = L;
  orig[r] = orig_in ? orig_in[i] : i;
  if (hist) {
    uint32_t* ho = reinterpret_cast<uint32_t*>(hist + static_cast<size_t>(r) * 32);
    for (int q = 0; q < 8; ++q)
      ho[q] = static_cast<uint32_t>(h[4 * q]) | (static_cast<uint32_t>(h[4 * q + 1]) << 8) |
              (static_cast<uint32_t>(h[4 * q + 2]) << 16) | (static_cast<uint32_t>(h[4 * q + 3]) << 24);
  }
  if (hist16) {  // 16 buckets: bucket b and bucket b + 16 together, saturating like the 32-bucket counts
    uint32_t* ho = reinterpret_cast<uint32_t*>(hist16 + static_cast<size_t>(r) * 16);
    auto both = [&](int b) -> uint32_t { return static_cast<uint32_t>(min(255, static_cast<int>(h[b]) + static_cast<int>(h[b + 16]))); };
    for (int q = 0; q < 4; ++q) ho[q] = both(4 * q) | (both(4 * q + 1) << 8) | (both(4 * q + 2) << 16) | (both(4 * q + 3) << 24);
  }
  if (len_class) len_class[r] = static_cast<uint32_t>(stride - L);
}

// ---------------------------------------------------------------------------------------------------- level items
__global__ void level_keys_kernel(const int32_t* __restrict__ nlev_in, int n, int32_t* __restrict__ small_first, Status* st) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const int L = nlev_in[i];
  if (L < 0 || L > 64) atomicMax(&st->err, static_cast<int>(kErrBadLevels));
  small_first[i] = 64 - max(0, min(L, 64));  // deeper items first
}

__global__ void level_gather_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ item_of,
                                    const int32_t* __restrict__ seg_of, int rows, const int32_t* __restrict__ first_in,
                                    const int32_t* __restrict__ nlev_in, const int32_t* __restrict__ orig_in,
                                    const uint64_t* __restrict__ cat, int partition, int32_t* __restrict__ first,
                                    int32_t* __restrict__ nlev, int32_t* __restrict__ orig, uint64_t* __restrict__ cat_out,
                                    int32_t* __restrict__ seg, uint32_t* __restrict__ seg_class) {
  const int r = blockIdx.x * kThreads + threadIdx.x;
  if (r >= rows) return;
  const int e = perm[r];
  const int i = item_of[e];
  first[r] = first_in[i];
  nlev[r] = nlev_in[i];
  orig[r] = orig_in ? orig_in[i] : i;
  if (cat_out) cat_out[r] = cat ? cat[i] : 0ull;
  if (partition) {
    seg[r] = seg_of[e];
    seg_class[r] = static_cast<uint32_t>(seg_of[e]);
  }
}

// start[0 .. n_classes] of a class column (n_classes <= 513)
int class_starts(const uint32_t* cls, int rows, int n_classes, int32_t* start, Scratch& sc) {
  int32_t* counts = sc.get<int32_t>(n_classes);
  if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
  (void)hipMemsetAsync(counts, 0, sizeof(int32_t) * n_classes, sc.stream);
  if (rows > 0) hipLaunchKernelGGL(class_hist_kernel, blocks_for(rows), dim3(kThreads), 0, sc.stream, cls, rows, n_classes, counts);
  hipLaunchKernelGGL(class_scan_kernel, dim3(1), dim3(64), 0, sc.stream, counts, n_classes, start);
  return 0;
}

const char* error_text(int err) {
  switch (err) {
    case kErrRowTooWide: return "a row holds more ids than the table's width";
    case kErrDuplicateId: return "duplicate id inside a row (sets must be de-duplicated)";
    case kErrBadCode: return "code unit outside the alphabet, or a length outside [0, stride]";
    case kErrBit63: return "category bit 63 is in use: the empty items cannot become a category of their own "
                           "(encode both sides without a partition)";
    case kErrBadLevels: return "nlev outside its range, or plen not a non-decreasing prefix-length row";
    case kErrBadVocab: return "an id is >= the table's vocab (the global inverted index addresses its offsets by id)";
    default: return "unknown";
  }
}

// items -> expanded, sorted rows.  Fills item_of / seg_of / perm (scratch) and *rows_out (synchronises).
struct RowPlan {
  int32_t* item_of = nullptr;
  int32_t* seg_of = nullptr;
  int32_t* perm = nullptr;
  uint64_t* cat = nullptr;  // cat' per ITEM (nullptr when the table has no category column)
  int rows = 0;
};

int plan_rows(const uint64_t* cat_in, const int32_t* small_first, int n, int radix, int category_mode, int partition,
              int capacity, Status* d_status, Scratch& sc, RowPlan* plan, const char* who) {
  hipStream_t stream = sc.stream;
  const int both_empty = category_mode == NSM_CAT_INTERSECT_OR_BOTH_EMPTY;
  int32_t* rows_per = sc.get<int32_t>(n);
  int32_t* off = sc.get<int32_t>(n);
  plan->cat = cat_in ? sc.get<uint64_t>(n) : nullptr;
  if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
  hipLaunchKernelGGL(category_rows_kernel, blocks_for(n), dim3(kThreads), 0, stream, cat_in, n, both_empty, partition,
                     plan->cat, rows_per, d_status);
  int rows = n;
  if (partition) {
    if (int rc = exclusive_scan_i32(rows_per, off, n, sc)) return rc;
    int last[2] = {0, 0};
    if (n > 0) {
      if (hipMemcpyAsync(&last[0], off + n - 1, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
          hipMemcpyAsync(&last[1], rows_per + n - 1, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
          hipStreamSynchronize(stream) != hipSuccess)
        return hip_status(hipGetLastError(), "builder row count");
    }
    rows = last[0] + last[1];
  } else {
    hipLaunchKernelGGL(iota_kernel, blocks_for(n), dim3(kThreads), 0, stream, off, n);
  }
  if (rows > capacity) {
    set_error("%s: the table needs %d rows, the output columns hold %d", who, rows, capacity);
    return NSM_E_BADARG;
  }
  plan->rows = rows;
  plan->item_of = sc.get<int32_t>(rows);
  plan->seg_of = sc.get<int32_t>(rows);
  plan->perm = sc.get<int32_t>(rows);
  uint32_t* keys = sc.get<uint32_t>(rows);
  if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(expand_kernel, blocks_for(n), dim3(kThreads), 0, stream, plan->cat, off, n, partition, plan->item_of,
                     plan->seg_of);
  hipLaunchKernelGGL(make_keys_kernel, blocks_for(rows), dim3(kThreads), 0, stream, plan->item_of, plan->seg_of, small_first,
                     rows, radix, keys);
  int bits = 1;
  while ((1 << bits) < 64 * radix) ++bits;
  return sort_rows(keys, rows, bits, plan->perm, sc);
}

int finish(Status* d_status, hipStream_t stream, const char* who, int* rows_out) {
  Status h{0, 0};
  if (hipMemcpyAsync(&h, d_status, sizeof(Status), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess)
    return hip_status(hipGetLastError(), who);
  if (h.err != kErrNone) {
    set_error("%s: %s", who, error_text(h.err));
    return NSM_E_BADARG;
  }
  (void)rows_out;
  return hip_status(hipGetLastError(), who);
}

}  // namespace
}  // namespace nsm

using namespace nsm;

extern "C" int nsm_build_set_table(const int32_t* ids_in, int32_t n, int32_t width_in, int32_t side,
                                   const int32_t* nlev_in, const uint8_t* plen_in, const uint64_t* cat_in,
                                   const int32_t* orig_in, int32_t category_mode, uint32_t flags, nsm_set_table* out,
                                   void* stream_) {
  const char* who = "nsm_build_set_table";
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!out || n < 0 || width_in < 1 || (n > 0 && !ids_in) || (side != 0 && side != 1)) {
    set_error("%s: null / negative argument", who);
    return NSM_E_BADARG;
  }
  const int width = out->width;
  if (width != 16 && width != 32 && width != 64) {
    set_error("%s: width %d unsupported (16, 32 or 64)", who, width);
    return NSM_E_UNSUPPORTED;
  }
  const bool levels = nlev_in != nullptr;
  if (levels && (!plen_in || out->max_levels < 1 || out->max_levels > 64 || !out->nlev || !out->plen || !out->filt)) {
    set_error("%s: a levels table needs plen, max_levels in [1, 64] and the nlev / plen / filt columns", who);
    return NSM_E_BADARG;
  }
  if (!out->ids || !out->cnt || !out->sig || !out->orig || !out->size_start) {
    set_error("%s: output column missing (ids, cnt, sig, orig, size_start are required)", who);
    return NSM_E_BADARG;
  }
  const int mode = (levels && cat_in) ? category_mode : NSM_CAT_NONE;
  const int partition = (mode != NSM_CAT_NONE && (flags & NSM_BUILD_PARTITION)) ? 1 : 0;
  if (partition && (!out->seg || !out->seg_start || !out->cat)) {
    set_error("%s: a category partition needs the cat / seg / seg_start columns", who);
    return NSM_E_BADARG;
  }
  const int capacity = out->n;
  Scratch sc(stream);
  Status* d_status = sc.get<Status>(1);
  int32_t* ids_tmp = sc.get<int32_t>(static_cast<size_t>(n) * width);
  int32_t* cnt_tmp = sc.get<int32_t>(n);
  int32_t* small_first = sc.get<int32_t>(n);
  if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
  (void)hipMemsetAsync(d_status, 0, sizeof(Status), stream);
  hipLaunchKernelGGL(set_rows_kernel, blocks_for(n), dim3(kThreads), 0, stream, ids_in, n, width_in, width,
                     (flags & NSM_BUILD_VALIDATE) ? 1 : 0, nlev_in, plen_in, out->max_levels, ids_tmp, cnt_tmp, small_first,
                     d_status);
  RowPlan plan;
  if (int rc = plan_rows(cat_in, small_first, n, width + 1, mode, partition, capacity, d_status, sc, &plan, who)) return rc;
  const int rows = plan.rows;
  uint32_t* size_class = sc.get<uint32_t>(rows + 1);
  uint32_t* seg_class = sc.get<uint32_t>(rows + 1);
  if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
  if (rows > 0)
    hipLaunchKernelGGL(set_gather_kernel, blocks_for(rows), dim3(kThreads), 0, stream, plan.perm, plan.item_of, plan.seg_of, rows,
                       width, side == 0 ? -1 : -2, ids_tmp, cnt_tmp, orig_in, nlev_in, plen_in, out->max_levels, plan.cat,
                       partition, const_cast<int32_t*>(out->ids), const_cast<int32_t*>(out->cnt),
                       const_cast<uint64_t*>(out->sig), const_cast<uint64_t*>(out->sig2), const_cast<int32_t*>(out->orig),
                       const_cast<int32_t*>(out->nlev), const_cast<uint8_t*>(out->plen), const_cast<uint64_t*>(out->cat),
                       const_cast<uint32_t*>(out->filt), const_cast<int32_t*>(out->seg), size_class);
  // size_start[c] = rows of a size class < c (class = width - cnt): in the unpartitioned table, rows of size
  // (width - c) are [size_start[c], size_start[c + 1]) (what the RAW grid reads); seg_start likewise per category
  if (int rc = class_starts(size_class, rows, width + 1, const_cast<int32_t*>(out->size_start), sc)) return rc;
  if (partition) {
    if (rows > 0) hipLaunchKernelGGL(widen_kernel, blocks_for(rows), dim3(kThreads), 0, stream, out->seg, rows, seg_class);
    if (int rc = class_starts(seg_class, rows, 64, const_cast<int32_t*>(out->seg_start), sc)) return rc;
  }
  // global inverted index (RAW tables whose caller provides the two columns)
  unsigned long long* d_sq = nullptr;
  for (int c = 0; c < 5; ++c) out->post_sq[c] = 0;
  if (out->post || out->post_start) {
    // key space: ids, times 64 category segments for a partitioned levels table
    const long long n_keys = static_cast<long long>(out->vocab) * (partition ? 64 : 1);
    if (!out->post || !out->post_start || out->vocab < 1 || n_keys > (1ll << 25)) {
      set_error("%s: the global inverted index needs both the post and post_start columns and 1 <= vocab (x 64 with a category "
                "partition) <= 2^25", who);
      return NSM_E_BADARG;
    }
    const long long slots = static_cast<long long>(rows) * width;
    const int row_bits = out->post_row_bits, format = out->post_format;
    int width_log = 0;
    while ((1 << width_log) < width) ++width_log;
    if (format < 0 || format > 2 || (format == 0) != (row_bits == 0) || row_bits < 0 ||
        (row_bits > 0 && (row_bits + 2 * width_log > 32 || (static_cast<long long>(rows) > (1ll << row_bits))))) {
      set_error("%s: post_format %d with post_row_bits %d cannot hold %d rows of width %d", who, format, row_bits, rows, width);
      return NSM_E_BADARG;
    }
    unsigned long long* keys = sc.get<unsigned long long>(slots);
    unsigned long long* keys_sorted = sc.get<unsigned long long>(slots);
    d_sq = sc.get<unsigned long long>(5);
    if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
    (void)hipMemsetAsync(d_sq, 0, 5 * sizeof(unsigned long long), stream);
    int bits = 1;
    while ((1ll << bits) < n_keys) ++bits;
    bits += 8;
    const unsigned long long key_mask = (1ull << bits) - 1ull;
    int32_t* post_start = const_cast<int32_t*>(out->post_start);
    const size_t n_bounds = static_cast<size_t>(5 * n_keys + 1);
    (void)hipMemsetAsync(post_start, 0, n_bounds * sizeof(int32_t), stream);
    // (the entry type only changes the sort's value type and two stores)
    auto fill = [&](auto* post, auto format_tag) -> int {
      using VAL = std::remove_pointer_t<decltype(post)>;
      constexpr int kFormat = decltype(format_tag)::value;
      if (slots <= 0) return 0;
      VAL* vals = sc.get<VAL>(slots);
      if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
      hipLaunchKernelGGL((post_keys_kernel<VAL, kFormat>), blocks_for(slots), dim3(kThreads), 0, stream, out->ids, out->cnt,
                         partition ? out->seg : static_cast<const int32_t*>(nullptr), out->sig, rows, width, out->vocab, row_bits,
                         width_log, keys, vals, d_status);
      size_t bytes = 0;
      hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, keys, keys_sorted, vals, post, static_cast<size_t>(slots), 0u,
                                               static_cast<unsigned>(bits), stream);
      if (e != hipSuccess) return hip_status(e, "radix_sort_pairs (postings, size)");
      char* temp = sc.get<char>(bytes);
      if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
      e = rocprim::radix_sort_pairs(temp, bytes, keys, keys_sorted, vals, post, static_cast<size_t>(slots), 0u,
                                    static_cast<unsigned>(bits), stream);
      if (e != hipSuccess) return hip_status(e, "radix_sort_pairs (postings)");
      hipLaunchKernelGGL(post_bounds_kernel<VAL>, blocks_for(slots), dim3(kThreads), 0, stream, keys_sorted, slots,
                         static_cast<int>(n_keys), key_mask, post_start, post);
      return 0;
    };
    void* post_col = const_cast<void*>(static_cast<const void*>(out->post));
    if (int rc = format == 1   ? fill(static_cast<uint32_t*>(post_col), std::integral_constant<int, 1>{})
                 : format == 2 ? fill(static_cast<unsigned long long*>(post_col), std::integral_constant<int, 2>{})
                               : fill(static_cast<unsigned long long*>(post_col), std::integral_constant<int, 0>{}))
      return rc;
    {
      size_t bytes = 0;
      hipError_t e = rocprim::inclusive_scan(nullptr, bytes, post_start, post_start, n_bounds, rocprim::maximum<int32_t>(), stream);
      if (e != hipSuccess) return hip_status(e, "inclusive_scan (posting bounds, size)");
      char* temp = sc.get<char>(bytes);
      if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
      e = rocprim::inclusive_scan(temp, bytes, post_start, post_start, n_bounds, rocprim::maximum<int32_t>(), stream);
      if (e != hipSuccess) return hip_status(e, "inclusive_scan (posting bounds)");
    }
    hipLaunchKernelGGL(post_stats_kernel, blocks_for(n_keys), dim3(kThreads), 0, stream, out->post_start, static_cast<int>(n_keys),
                       d_sq);
  }
  out->n = rows;
  if (d_sq) {
    unsigned long long h_sq[5] = {0, 0, 0, 0, 0};
    if (hipMemcpyAsync(h_sq, d_sq, sizeof(h_sq), hipMemcpyDeviceToHost, stream) != hipSuccess)
      return hip_status(hipGetLastError(), who);
    const int rc = finish(d_status, stream, who, nullptr);  // (synchronises: h_sq has arrived)
    for (int c = 0; c < 5; ++c) out->post_sq[c] = h_sq[c];
    return rc;
  }
  return finish(d_status, stream, who, nullptr);
}

extern "C" int nsm_build_str_table(const uint8_t* codes_in, const int32_t* len_in, const int32_t* orig_in, int32_t n,
                                   uint32_t flags, nsm_str_table* out, void* stream_) {
  const char* who = "nsm_build_str_table";
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!out || n < 0 || (n > 0 && (!codes_in || !len_in)) || !out->codes || !out->len || !out->orig) {
    set_error("%s: null / negative argument (codes, len, orig columns are required)", who);
    return NSM_E_BADARG;
  }
  const int stride = out->stride;
  if (stride != 64 && stride != 128 && stride != 256 && stride != 512) {
    set_error("%s: stride %d unsupported (64, 128, 256 or 512)", who, stride);
    return NSM_E_UNSUPPORTED;
  }
  if (out->alphabet < 1 || out->alphabet > 255 || out->n < n) {
    set_error("%s: alphabet outside [1, 255] or output columns shorter than %d rows", who, n);
    return NSM_E_BADARG;
  }
  const bool sort = (flags & NSM_BUILD_SORT) != 0;
  if (sort && !out->len_start) {
    set_error("%s: a sorted table needs the len_start column", who);
    return NSM_E_BADARG;
  }
  Scratch sc(stream);
  Status* d_status = sc.get<Status>(1);
  uint32_t* keys = sc.get<uint32_t>(n);
  int32_t* perm = sort ? sc.get<int32_t>(n) : nullptr;
  uint32_t* len_class = sort ? sc.get<uint32_t>(n + 1) : nullptr;
  if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
  (void)hipMemsetAsync(d_status, 0, sizeof(Status), stream);
  if (n > 0) {
    hipLaunchKernelGGL(str_keys_kernel, blocks_for(n), dim3(kThreads), 0, stream, len_in, n, stride, keys, d_status);
    if (sort) {
      if (int rc = sort_rows(keys, n, 10, perm, sc)) return rc;
    }
    hipLaunchKernelGGL(str_gather_kernel, blocks_for(n), dim3(kThreads), 0, stream, perm, n, stride, out->alphabet, codes_in,
                       len_in, orig_in, const_cast<uint8_t*>(out->codes), const_cast<int32_t*>(out->len),
                       const_cast<int32_t*>(out->orig), const_cast<uint8_t*>(out->hist), const_cast<uint8_t*>(out->hist16), len_class,
                       d_status);
  }
  if (sort) {
    if (int rc = class_starts(len_class, n, stride + 1, const_cast<int32_t*>(out->len_start), sc)) return rc;
  }
  out->n = n;
  return finish(d_status, stream, who, nullptr);
}

extern "C" int nsm_build_level_items(const int32_t* first_in, const int32_t* nlev_in, const uint64_t* cat_in,
                                     const int32_t* orig_in, int32_t n, int32_t category_mode, uint32_t flags,
                                     nsm_level_items* out, void* stream_) {
  const char* who = "nsm_build_level_items";
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!out || n < 0 || (n > 0 && (!first_in || !nlev_in)) || !out->first || !out->nlev || !out->orig) {
    set_error("%s: null / negative argument (first, nlev, orig columns are required)", who);
    return NSM_E_BADARG;
  }
  const int mode = cat_in ? category_mode : NSM_CAT_NONE;
  const int partition = (mode != NSM_CAT_NONE && (flags & NSM_BUILD_PARTITION)) ? 1 : 0;
  if ((cat_in && !out->cat) || (partition && (!out->seg || !out->seg_start))) {
    set_error("%s: the cat column (and seg / seg_start for a partition) is missing", who);
    return NSM_E_BADARG;
  }
  Scratch sc(stream);
  Status* d_status = sc.get<Status>(1);
  int32_t* small_first = sc.get<int32_t>(n);
  if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
  (void)hipMemsetAsync(d_status, 0, sizeof(Status), stream);
  if (n > 0) hipLaunchKernelGGL(level_keys_kernel, blocks_for(n), dim3(kThreads), 0, stream, nlev_in, n, small_first, d_status);
  RowPlan plan;
  if (int rc = plan_rows(cat_in, small_first, n, 65, mode, partition, out->n, d_status, sc, &plan, who)) return rc;
  const int rows = plan.rows;
  uint32_t* seg_class = sc.get<uint32_t>(rows + 1);
  if (sc.failed) return hip_status(hipErrorOutOfMemory, "builder scratch");
  if (rows > 0)
    hipLaunchKernelGGL(level_gather_kernel, blocks_for(rows), dim3(kThreads), 0, stream, plan.perm, plan.item_of, plan.seg_of,
                       rows, first_in, nlev_in, orig_in, plan.cat, partition, const_cast<int32_t*>(out->first),
                       const_cast<int32_t*>(out->nlev), const_cast<int32_t*>(out->orig), const_cast<uint64_t*>(out->cat),
                       const_cast<int32_t*>(out->seg), seg_class);
  if (partition) {
    if (int rc = class_starts(seg_class, rows, 64, const_cast<int32_t*>(out->seg_start), sc)) return rc;
  }
  out->n = rows;
  return finish(d_status, stream, who, nullptr);
}
