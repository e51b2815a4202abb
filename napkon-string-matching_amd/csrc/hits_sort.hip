// Canonical ordering of a hit list: score descending, then i, then j ascending.
// Stands in for Comparable.sort_by_score (reference: napkon_string_matching/types/comparable.py:69-70,
// whose quicksort leaves the order of equal scores unspecified; (i, j) is unique per grid, so this
// order is total and deterministic).
//
// The number of hits lives in device memory (the grids append with an atomic counter), so the
// launch geometry comes from `capacity` and every kernel reads the count itself:
//   * up to 8192 live records (the common case): ONE workgroup sorts 128-bit integer keys
//     (~score bits, i, j) with a bitonic network in LDS (16 bytes per record, 128 KB at most) and
//     writes the records back in place -- a single launch;
//   * more: ONE radix sort (rocPRIM, a library sort) over 128-bit keys built in place from the records -- inverted
//     score bits, then i, then j, packed so that only the bits that can differ are sorted (the caller's id_limit bounds
//     i and j: 94 key bits = 12 digit passes for a 20 000-item cohort instead of 16).  The record IS its key, so the
//     sort moves 16 bytes per record and pass and nothing else.  Its temporary storage is stream-ordered
//     (hipMallocAsync / hipFreeAsync inside the call).  10.5 M hits (the reference's default configuration at the cache
//     threshold): 6 ms of bitonic passes before, see DESIGN.md for the measured radix time;
//   * on a stream that is being captured into a graph (no allocation allowed): bitonic network, "flip" form (all
//     comparators point the same way), which needs no padding to a power of two.  Every comparator pass whose partner distance is below 2048 stays
//     inside an aligned 2048-record tile, so those passes run fused in LDS (one launch sorts all
//     tiles, one launch finishes each larger merge); only the passes with distance >= 2048 are
//     separate launches over global memory.  Passes beyond the live count exit immediately.
#include "nsm_common.hpp"

#include <atomic>
#include <cstring>

#include <rocprim/rocprim.hpp>

namespace nsm {

__device__ __forceinline__ bool hit_before(const nsm_hit& a, const nsm_hit& b) {
  if (a.score != b.score) return a.score > b.score;
  if (a.i != b.i) return a.i < b.i;
  return a.j < b.j;
}

__device__ __forceinline__ unsigned long long live_count(const unsigned long long* count,
                                                         unsigned long long capacity) {
  const unsigned long long c = *count;
  return c < capacity ? c : capacity;
}

constexpr int kSmallSortMax = 8192;  // one workgroup, 128-bit keys in LDS: 128 KB of the CU's 160 KB at the maximum
constexpr unsigned long long kRankSortMax = 8192;
constexpr int kSmallSortThreads = 1024;

// Order-preserving integer image of a double, inverted: larger score -> smaller key.
__device__ __forceinline__ unsigned long long score_key(double score) {
  const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(score));
  const unsigned long long asc = bits ^ ((bits >> 63) ? ~0ull : (1ull << 63));
  return ~asc;
}

__global__ __launch_bounds__(kSmallSortThreads) void small_sort_kernel(
    nsm_hit* __restrict__ hits, unsigned long long capacity, const unsigned long long* __restrict__ count, int lds_records) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];  // [p] high words | [p] low words, p <= lds_records
  const unsigned long long n64 = live_count(count, capacity);
  if (n64 < 2 || n64 > static_cast<unsigned long long>(lds_records)) return;
  const int n = static_cast<int>(n64);
  int p = 2;
  while (p < n) p <<= 1;
  unsigned long long* key_hi = s_keys;
  unsigned long long* key_lo = s_keys + p;
  for (int t = threadIdx.x; t < p; t += kSmallSortThreads) {
    if (t < n) {
      const nsm_hit h = hits[t];
      key_hi[t] = score_key(h.score);
      key_lo[t] = (static_cast<unsigned long long>(static_cast<uint32_t>(h.i) ^ 0x80000000u) << 32) |
                  (static_cast<uint32_t>(h.j) ^ 0x80000000u);
    } else {
      key_hi[t] = ~0ull;  // padding sorts last
      key_lo[t] = ~0ull;
    }
  }
  __syncthreads();
  for (int k = 2; k <= p; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (p >> 1); t += kSmallSortThreads) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // element with bit j clear
        const int hi = lo | j;
        const bool up = (lo & k) == 0;
        const unsigned long long ah = key_hi[lo], al = key_lo[lo], bh = key_hi[hi], bl = key_lo[hi];
        const bool a_gt_b = ah > bh || (ah == bh && al > bl);
        if (a_gt_b == up) {
          key_hi[lo] = bh; key_lo[lo] = bl;
          key_hi[hi] = ah; key_lo[hi] = al;
        }
      }
      __syncthreads();
    }
  }
  for (int t = threadIdx.x; t < n; t += kSmallSortThreads) {
    const unsigned long long asc = ~key_hi[t];
    const unsigned long long bits = asc ^ ((asc >> 63) ? (1ull << 63) : ~0ull);
    nsm_hit h;
    h.score = __longlong_as_double(static_cast<long long>(bits));
    h.i = static_cast<int32_t>(static_cast<uint32_t>(key_lo[t] >> 32) ^ 0x80000000u);
    h.j = static_cast<int32_t>(static_cast<uint32_t>(key_lo[t]) ^ 0x80000000u);
    hits[t] = h;
  }
}

// One comparator pass of the flip-form bitonic network: k = merge size, j = partner distance
// (j == 0 marks the first, "flip" step of a merge: partner = idx ^ (k - 1)).
__global__ __launch_bounds__(kBlock) void bitonic_pass_kernel(nsm_hit* __restrict__ hits,
                                                              unsigned long long capacity,
                                                              const unsigned long long* __restrict__ count,
                                                              unsigned long long k, unsigned long long j) {
  const unsigned long long n = live_count(count, capacity);
  if (n <= kRankSortMax) return;  // the LDS / rank sorts handled it
  if ((k >> 1) >= n) return;      // merges larger than the (virtually padded) list do nothing
  for (unsigned long long idx = static_cast<unsigned long long>(blockIdx.x) * kBlock + threadIdx.x; idx < n;
       idx += static_cast<unsigned long long>(gridDim.x) * kBlock) {
    const unsigned long long partner = j == 0 ? (idx ^ (k - 1)) : (idx ^ j);
    if (partner > idx && partner < n) {
      const nsm_hit a = hits[idx];
      const nsm_hit b = hits[partner];
      if (hit_before(b, a)) {
        hits[idx] = b;
        hits[partner] = a;
      }
    }
  }
}

constexpr int kTile = 2048;

__device__ __forceinline__ void tile_pass(nsm_hit* tile, unsigned long long base, unsigned long long n, int k, int j) {
  // comparators of one pass inside the tile: kTile / 2 of them, 4 per thread
  for (int t = threadIdx.x; t < kTile / 2; t += kBlock) {
    int lo, hi;
    if (j == 0) {  // flip: partner = idx ^ (k - 1) inside the aligned k-block
      const int half = k >> 1;
      const int blk = t / half, off = t % half;
      lo = blk * k + off;
      hi = blk * k + (k - 1 - off);
    } else {
      lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
      hi = lo | j;
    }
    if (base + hi < n) {
      const nsm_hit a = tile[lo];
      const nsm_hit b = tile[hi];
      if (hit_before(b, a)) {
        tile[lo] = b;
        tile[hi] = a;
      }
    }
  }
  __syncthreads();
}

// merge == 0: sort every tile (all merges up to kTile).  merge > kTile: the passes j = kTile/2 .. 1
// that finish the merge of size `merge`.
__global__ __launch_bounds__(kBlock) void bitonic_tile_kernel(nsm_hit* __restrict__ hits,
                                                              unsigned long long capacity,
                                                              const unsigned long long* __restrict__ count,
                                                              unsigned long long merge) {
  __shared__ nsm_hit tile[kTile];
  const unsigned long long n = live_count(count, capacity);
  if (n <= kRankSortMax) return;
  if (merge && (merge >> 1) >= n) return;
  const unsigned long long base = static_cast<unsigned long long>(blockIdx.x) * kTile;
  if (base >= n) return;
  const int m = static_cast<int>(n - base < kTile ? n - base : kTile);
  for (int t = threadIdx.x; t < m; t += kBlock) tile[t] = hits[base + t];
  __syncthreads();
  if (merge == 0) {
    for (int k = 2; k <= kTile; k <<= 1) {
      tile_pass(tile, base, n, k, 0);
      for (int j = k >> 2; j > 0; j >>= 1) tile_pass(tile, base, n, k, j);
    }
  } else {
    for (int j = kTile >> 1; j > 0; j >>= 1) tile_pass(tile, base, n, 0, j);
  }
  for (int t = threadIdx.x; t < m; t += kBlock) hits[base + t] = tile[t];
}

// ---- radix path: record <-> 128-bit key, in place
struct Key128 {
  unsigned long long lo, hi;  // (value = hi : lo; the decomposer hands rocPRIM hi first = most significant)
};
static_assert(sizeof(Key128) == sizeof(nsm_hit), "the record is its own key");
struct Key128Decomposer {
  __host__ __device__ ::rocprim::tuple<unsigned long long&, unsigned long long&> operator()(Key128& k) const {
    return ::rocprim::tuple<unsigned long long&, unsigned long long&>(k.hi, k.lo);
  }
};

// key = score_key : i' : j' with i', j' in `bits` bits each (bits == 32: the ids' sign bit flipped, any int32 sorts right)
__global__ __launch_bounds__(kBlock) void hits_to_keys_kernel(nsm_hit* __restrict__ hits, unsigned long long capacity,
                                                              const unsigned long long* __restrict__ count,
                                                              unsigned long long bound, int bits) {
  const unsigned long long n = live_count(count, capacity);
  const unsigned long long t = static_cast<unsigned long long>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= bound) return;
  Key128* keys = reinterpret_cast<Key128*>(hits);
  Key128 k;
  if (t < n) {
    const nsm_hit h = hits[t];
    const unsigned long long sk = score_key(h.score);
    const uint32_t flip = bits == 32 ? 0x80000000u : 0u;
    const unsigned long long ids = (static_cast<unsigned long long>(static_cast<uint32_t>(h.i) ^ flip) << bits) |
                                   static_cast<unsigned long long>(static_cast<uint32_t>(h.j) ^ flip);
    const int idb = 2 * bits;  // 2 .. 64
    k.lo = idb == 64 ? ids : (ids | (sk << idb));
    k.hi = idb == 64 ? sk : (sk >> (64 - idb));
  } else {
    k.lo = ~0ull;  // padding up to the host's bound sorts last
    k.hi = ~0ull;
  }
  keys[t] = k;
}

__global__ __launch_bounds__(kBlock) void keys_to_hits_kernel(const Key128* __restrict__ keys, nsm_hit* __restrict__ hits,
                                                              unsigned long long capacity,
                                                              const unsigned long long* __restrict__ count,
                                                              unsigned long long bound, int bits) {
  unsigned long long n = live_count(count, capacity);
  if (n > bound) n = bound;
  const unsigned long long t = static_cast<unsigned long long>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= n) return;
  const Key128 k = keys[t];
  const int idb = 2 * bits;
  const unsigned long long sk = idb == 64 ? k.hi : ((k.hi << (64 - idb)) | (k.lo >> idb));
  const unsigned long long ids = idb == 64 ? k.lo : (k.lo & ((1ull << idb) - 1ull));
  const uint32_t flip = bits == 32 ? 0x80000000u : 0u;
  const unsigned long long asc = ~sk;
  const unsigned long long fbits = asc ^ ((asc >> 63) ? (1ull << 63) : ~0ull);
  nsm_hit h;
  h.score = __longlong_as_double(static_cast<long long>(fbits));
  h.i = static_cast<int32_t>(static_cast<uint32_t>(ids >> bits) ^ flip);
  h.j = static_cast<int32_t>(static_cast<uint32_t>(bits == 32 ? ids : (ids & ((1ull << bits) - 1ull))) ^ flip);
  hits[t] = h;
}

static int radix_sort_hits(nsm_hit* hits, nsm_hit* scratch, unsigned long long capacity, const unsigned long long* hit_count,
                           unsigned long long bound, uint32_t id_limit, hipStream_t s) {
  int bits = 32;
  if (id_limit > 0 && id_limit <= 0x80000000u) {
    bits = 1;
    while ((1ull << bits) < id_limit) ++bits;
  }
  const unsigned blocks = static_cast<unsigned>((bound + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(hits_to_keys_kernel, dim3(blocks), dim3(kBlock), 0, s, hits, capacity, hit_count, bound, bits);
  ::rocprim::double_buffer<Key128> keys(reinterpret_cast<Key128*>(hits), reinterpret_cast<Key128*>(scratch));
  const unsigned end_bit = static_cast<unsigned>(64 + 2 * bits);
  size_t bytes = 0;
  hipError_t e = ::rocprim::radix_sort_keys(nullptr, bytes, keys, static_cast<size_t>(bound), Key128Decomposer{}, 0u, end_bit, s);
  if (e != hipSuccess) return hip_status(e, "radix_sort_keys (size)");
  void* temp = nullptr;
  e = hipMallocAsync(&temp, bytes ? bytes : 1, s);
  if (e != hipSuccess) return hip_status(e, "nsm_sort_hits: stream-ordered scratch");
  e = ::rocprim::radix_sort_keys(temp, bytes, keys, static_cast<size_t>(bound), Key128Decomposer{}, 0u, end_bit, s);
  (void)hipFreeAsync(temp, s);
  if (e != hipSuccess) return hip_status(e, "radix_sort_keys");
  // (keys.current() is wherever the last digit pass left the result; the records go back to `hits` either way -- in place
  // when that is `hits` itself: one thread reads and rewrites its own 16 bytes)
  hipLaunchKernelGGL(keys_to_hits_kernel, dim3(blocks), dim3(kBlock), 0, s, keys.current(), hits, capacity, hit_count, bound, bits);
  return hip_status(hipGetLastError(), "nsm_sort_hits (radix)");
}

}  // namespace nsm

extern "C" int nsm_sort_hits(nsm_hit* hits, nsm_hit* scratch, uint64_t capacity, const unsigned long long* hit_count,
                             uint64_t n_hint, uint32_t id_limit, void* stream) {
  using namespace nsm;
  if (!hits || !hit_count) {
    set_error("nsm_sort_hits: null argument");
    return NSM_E_BADARG;
  }
  if (capacity == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // The count lives on the device, so the launch geometry comes from what the host knows: `capacity`, or the caller's promise
  // n_hint >= min(*hit_count, capacity) (0 = no promise).  A host that has read the counter -- run_grid does, to size its
  // copy -- pays one launch for ten hits whatever the buffer's capacity.
  const unsigned long long bound = (n_hint != 0 && n_hint < capacity) ? n_hint : capacity;
  if (bound < 2) return 0;
  // up to 8192 live records: ONE launch of one workgroup (bitonic network on 128-bit keys in LDS).  Round 2 sorted
  // 2049..8192 records with a rank-sort launch plus a copy launch; both were launched for every capacity above 2048
  // (the count lives on the device) and two empty launches cost the headline step more than they ever saved.
  const int lds_records = static_cast<int>(bound < kSmallSortMax ? bound : kSmallSortMax);
  int p2 = 2;
  while (p2 < lds_records) p2 <<= 1;
  {
    static std::atomic<unsigned long long> attr_devs{0};  // one bit per device ordinal: the attribute is per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
    if (!bit || !(attr_devs.load(std::memory_order_acquire) & bit)) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&small_sort_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kSmallSortMax * 16);
      if (e != hipSuccess) return hip_status(e, "hipFuncSetAttribute(small_sort_kernel)");
      attr_devs.fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(small_sort_kernel, dim3(1), dim3(kSmallSortThreads), static_cast<size_t>(p2) * 16, s, hits, capacity,
                     hit_count, p2);
  if (bound <= kSmallSortMax) return hip_status(hipGetLastError(), "nsm_sort_hits (small)");
  if (!scratch) {
    set_error("nsm_sort_hits: scratch buffer required");
    return NSM_E_BADARG;
  }
  if (bound > 0x7fffffffull * 32ull) {
    set_error("nsm_sort_hits: capacity too large");
    return NSM_E_UNSUPPORTED;
  }
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &capture) != hipSuccess) capture = hipStreamCaptureStatusNone;
#ifndef NSM_SORT_BITONIC
  // (the single-workgroup kernel above has already returned without touching anything when more than 8192 records are live;
  // when fewer are, it has sorted them and the radix sort below re-sorts a sorted list: correct, and only paid by callers
  // that pass no n_hint although few records are live)
  if (capture == hipStreamCaptureStatusNone) return radix_sort_hits(hits, scratch, capacity, hit_count, bound, id_limit, s);
#endif
  unsigned long long blocks64 = (bound + kBlock - 1) / kBlock;
  const unsigned blocks = static_cast<unsigned>(blocks64 < 8192 ? blocks64 : 8192);
  unsigned long long tiles64 = (bound + kTile - 1) / kTile;
  if (tiles64 > 0x7fffffffull) {
    set_error("nsm_sort_hits: capacity too large");
    return NSM_E_UNSUPPORTED;
  }
  const unsigned tiles = static_cast<unsigned>(tiles64);
  hipLaunchKernelGGL(bitonic_tile_kernel, dim3(tiles), dim3(kBlock), 0, s, hits, capacity, hit_count, 0ull);
  for (unsigned long long k = 2ull * kTile; (k >> 1) < bound; k <<= 1) {
    hipLaunchKernelGGL(bitonic_pass_kernel, dim3(blocks), dim3(kBlock), 0, s, hits, capacity, hit_count, k, 0ull);
    for (unsigned long long j = k >> 2; j >= static_cast<unsigned long long>(kTile); j >>= 1)
      hipLaunchKernelGGL(bitonic_pass_kernel, dim3(blocks), dim3(kBlock), 0, s, hits, capacity, hit_count, k, j);
    hipLaunchKernelGGL(bitonic_tile_kernel, dim3(tiles), dim3(kBlock), 0, s, hits, capacity, hit_count, k);
  }
  return hip_status(hipGetLastError(), "nsm_sort_hits (bitonic)");
}
