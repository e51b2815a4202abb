// RAW Indel-ratio grid, pruning kernel with a TWO-STAGE histogram filter (included by indel_raw.hip).
//
// The one-stage kernel (indel_raw_kernel<true>) spends its time in eight half-rate v_sad_u8 per pair: the L1 distance of two
// 32-bucket symbol histograms.  Merging bucket b with bucket b + 16 gives a 16-bucket histogram whose L1 distance can only be
// SMALLER (|a1 + a2 - b1 - b2| <= |a1 - b1| + |a2 - b2|), so "coarse L1 <= limit" is a necessary condition too and costs four
// v_sad_u8.  On configs[2] it passes 1 % of the pairs -- far too many for a wave-wide second look (some lane of a row's 64
// passes in 20 % of the rows), so the survivors are handled PER PAIR:
//
//   scan    per batch of R = 8 left rows (their 128 bytes of coarse histograms arrive with two s_load_dwordx16) and T = 2
//           right tiles: 4 v_sad_u8 per pair, the sign of the seeded sum shifted into a per-lane bit mask (v_alignbit);
//   stack   lanes whose mask is not empty push (batch, lane, mask) on the wave's LDS stack -- one ballot and one ds_write per
//           BATCH, nothing per pair;
//   drain   whenever the stack holds more than 128 entries, 64 at a time (64 of 64 lanes busy): lane = one entry, ONE of
//           its pairs (an entry with more pairs goes back on the stack) -- the 32-bucket L1 of the pair (the left histogram
//           gathered from global memory, the right one read from the tile's copy in LDS, 8 v_sad_u8: 1 % of the pairs), then,
//           for the 1e-5 that remain, the bit-parallel LCS of the pair on the SCALAR unit (raw_lcs_pair).  Entries carry
//           their row and its length, so the stack outlives the length classes and is emptied once, at the end.
//
// Every test that drops a pair is an upper bound of the LCS: the hits are the one-stage kernel's, the exhaustive kernel's and
// the oracle's.  Where the time goes on configs[2] (200k x 200k, threshold 0.8, same box, variant builds NSM_C3C_X_*):
// scan 3.11 ms (VALU-bound: 64 v_sad_u8 + 16 v_alignbit + one push per batch), pop / re-push 0.11, the 32-bucket test of
// 4e8 pairs 0.39, 4.1e5 LCS 0.16 -- 3.76 ms against the one-stage kernel's 5.11 (two tiles per wave; 6.54 with one tile and
// the wave-wide LCS).
#pragma once

namespace nsm {

#ifndef NSM_C3C_TILES
#define NSM_C3C_TILES 2
#endif
#ifndef NSM_C3C_ROWS
#define NSM_C3C_ROWS 8
#endif
constexpr int kC3cStack = 192;  // entries per wave; drained when fewer than 64 slots are left
// a stack entry: low word = the batch's pass mask (R rows x T tiles <= 32 bits), high word = (first row - chunk start) << 13 |
// la << 6 | lane  (a chunk has at most 2^15 rows: nsm_indel_raw_grid)

// dynamic LDS: [wave][kC3cStack] u64 stack | [wave][T][8][64] u32 right histograms | [wave][T][64] u8 right lengths
//              | lcsmin bytes
static inline size_t c3c_lds_bytes(int tiles) {
  return static_cast<size_t>(kWavesPerBlock) * (kC3cStack * 8 + static_cast<size_t>(tiles) * (8 * kWave * 4 + kWave)) + 136;
}

#ifndef NSM_C3C_OCC
#define NSM_C3C_OCC
#endif
template <int T, int R>
__global__ __launch_bounds__(kBlock) NSM_C3C_OCC void indel_raw_coarse_kernel(
    const uint8_t* __restrict__ lcodes, const int32_t* __restrict__ llen, const int32_t* __restrict__ lstart,
    const int32_t* __restrict__ lorig, const uint32_t* __restrict__ lhist, const uint32_t* __restrict__ lh16,
    const uint8_t* __restrict__ rcodes, const int32_t* __restrict__ rlen, const int32_t* __restrict__ rorig,
    const uint32_t* __restrict__ rhist, const uint32_t* __restrict__ rh16, nsm_hit* __restrict__ hits,
    unsigned long long* __restrict__ count, const IndelRawParams p) {
  static_assert((T == 1 || T == 2 || T == 4) && (R == 4 || R == 8) && R * T <= 32, "R rows x T right tiles per batch: one mask bit each");
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_mem[];
  unsigned long long* s_stack = s_mem;
  uint32_t* s_rh = reinterpret_cast<uint32_t*>(s_stack + kWavesPerBlock * kC3cStack);
  uint8_t* s_rl = reinterpret_cast<uint8_t*>(s_rh + kWavesPerBlock * T * 8 * kWave);
  uint8_t* s_lcsmin = s_rl + kWavesPerBlock * T * kWave;
  for (int t = threadIdx.x; t < 132; t += kBlock) s_lcsmin[t] = p.lcsmin[t];
  __syncthreads();

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int tile0 = (blockIdx.x * kWavesPerBlock + wave) * T;
  if (tile0 * kWave >= p.n_right) return;
  unsigned long long* stack = s_stack + wave * kC3cStack;
  uint32_t* rh_lds = s_rh + wave * T * 8 * kWave;  // [t][q][lane]
  uint8_t* rl_lds = s_rl + wave * T * kWave;

  bool valid[T];
  int jc[T], lbj[T];
  uint32_t hc[T][4];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int j = (tile0 + t) * kWave + lane;
    valid[t] = j < p.n_right;
    jc[t] = valid[t] ? j : p.n_right - 1;
    lbj[t] = valid[t] ? rlen[jc[t]] : 0;
    const uint4 h = reinterpret_cast<const uint4*>(rh16)[jc[t]];
    hc[t][0] = h.x; hc[t][1] = h.y; hc[t][2] = h.z; hc[t][3] = h.w;
    const uint4* fp = reinterpret_cast<const uint4*>(rhist + static_cast<size_t>(jc[t]) * 8);
    const uint4 f0 = fp[0], f1 = fp[1];
    uint32_t* dst = rh_lds + t * 8 * kWave + lane;
    dst[0 * kWave] = f0.x; dst[1 * kWave] = f0.y; dst[2 * kWave] = f0.z; dst[3 * kWave] = f0.w;
    dst[4 * kWave] = f1.x; dst[5 * kWave] = f1.y; dst[6 * kWave] = f1.z; dst[7 * kWave] = f1.w;
    rl_lds[t * kWave + lane] = static_cast<uint8_t>(lbj[t]);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int i0 = blockIdx.y * p.rows_per_chunk;
  const int i1 = min(p.n_left, i0 + p.rows_per_chunk);
  int q_cnt = 0;  // wave-uniform: entries on the stack

  auto need_of = [&](int la, int lb) -> int { return (la == 0 || lb == 0) ? p.zero_need : static_cast<int>(s_lcsmin[la + lb]); };

  // pop up to 64 entries: lane = one entry, its first pair; entries with more pairs go back on the stack
  auto drain_pass = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n = min(q_cnt, kWave);
    const int base = q_cnt - n;
    const bool active = lane < n;
#ifdef NSM_C3C_X_NODRAIN  // (timing experiments: the entries are dropped)
    q_cnt = base;
    return;
#endif
    const unsigned long long e = stack[base + (active ? lane : 0)];
    uint32_t bits = active ? static_cast<uint32_t>(e) : 0u;
    const uint32_t ehi = static_cast<uint32_t>(e >> 32);
    const int tl = static_cast<int>(ehi & 63u);
    const int la = static_cast<int>((ehi >> 6) & 127u);
    const int ib = i0 + static_cast<int>(ehi >> 13);
    const int pos = active ? 31 - __clz(bits) : 0;  // (an entry on the stack has a bit set)
    bits &= ~(1u << pos);
    const int k = R * T - 1 - pos;  // pair k of the batch: row k / T, tile k % T
    const int r = k / T, t = k - r * T;
    const int row = active ? ib + r : i0;
    // the 32-bucket filter of the pair
    const uint4* lp = reinterpret_cast<const uint4*>(lhist + static_cast<size_t>(row) * 8);
    const uint4 l0 = lp[0], l1v = lp[1];
    const int lb = rl_lds[t * kWave + tl];
    uint32_t rq[8];
    {
      const uint32_t* rp = rh_lds + t * 8 * kWave + tl;
#pragma unroll
      for (int q = 0; q < 8; ++q) rq[q] = rp[q * kWave];
    }
    uint32_t l1 = 0;
    l1 = __builtin_amdgcn_sad_u8(l0.x, rq[0], l1);
    l1 = __builtin_amdgcn_sad_u8(l0.y, rq[1], l1);
    l1 = __builtin_amdgcn_sad_u8(l0.z, rq[2], l1);
    l1 = __builtin_amdgcn_sad_u8(l0.w, rq[3], l1);
    l1 = __builtin_amdgcn_sad_u8(l1v.x, rq[4], l1);
    l1 = __builtin_amdgcn_sad_u8(l1v.y, rq[5], l1);
    l1 = __builtin_amdgcn_sad_u8(l1v.z, rq[6], l1);
    l1 = __builtin_amdgcn_sad_u8(l1v.w, rq[7], l1);
    const int nd = need_of(la, lb);
    bool pass = active && min(la, lb) >= nd && static_cast<int>(l1) <= la + lb - 2 * nd;
    // entries with pairs left go back (every lane has read its entry: the slots [base, base + n) are free)
    __builtin_amdgcn_wave_barrier();
    const bool more = bits != 0u;
    const unsigned long long mm = __ballot(more);
    if (more) {
      const int slot = base + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mm), 0u));
      stack[slot] = (static_cast<unsigned long long>(ehi) << 32) | bits;
    }
    q_cnt = base + __popcll(mm);
    // the pairs that remain (1e-5 of all on configs[2]), one after the other on the scalar unit
#ifdef NSM_C3C_X_NOLCS  // (timing experiments)
    pass = false;
#endif
#ifdef NSM_C3C_X_FINEONLY  // (timing experiments: the 32-bucket test runs, nobody passes -- not known at compile time)
    pass = pass && p.n_left < 0;
#endif
    const int jp = (tile0 + t) * kWave + tl;
    for (unsigned long long todo = __ballot(pass); todo; todo &= todo - 1ull) {
      const int leader = __builtin_ctzll(todo);
      const int row_s = __builtin_amdgcn_readlane(row, leader);
      const int j_s = __builtin_amdgcn_readlane(jp, leader);
      const int la_s = __builtin_amdgcn_readlane(la, leader);
      const int lb_s = __builtin_amdgcn_readlane(lb, leader);
      const int lcs = raw_lcs_pair(lcodes, row_s, la_s, rcodes, j_s, lb_s, lane);
#ifdef NSM_C3C_X_COUNTLCS  // (experiments: one record per LCS call)
      if (lane == 0)
#else
      if (lcs >= need_of(la_s, lb_s) && lane == 0)
#endif
        emit_hit(hits, p.cap, count, indel_score(la_s, lb_s, lcs), lorig[row_s], rorig[j_s]);
    }
  };

  // rows are sorted by length (descending): rows of length 64 - c are [lstart[c], lstart[c + 1])
  const int c_first = 64 - llen[i0];
  const int c_last = 64 - llen[i1 - 1];
  for (int c = c_first; c <= c_last; ++c) {
    const int a = max(i0, lstart[c]);
    const int b = min(i1, lstart[c + 1]);
    if (a >= b) continue;
    const int la = 64 - c;
    uint32_t seed[T];
    bool some = false;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int need = valid[t] ? need_of(la, lbj[t]) : static_cast<int>(kNever);
      const bool fits = min(la, lbj[t]) >= need;  // exact length filter: LCS <= min(la, lb)
      some = some || fits;
      // the pair can only hit if L1 <= la + lb - 2 need; the SAD chain is seeded with -(limit + 1): negative <=> passes
      const int limit = fits ? la + lbj[t] - 2 * need : -1;
      seed[t] = static_cast<uint32_t>(-(limit + 1));
    }
    if (!__any(some)) continue;

    const uint32_t* __restrict__ hp = lh16 + static_cast<size_t>(a) * 4;
    const uint32_t lane_field = (static_cast<uint32_t>(la) << 6) | static_cast<uint32_t>(lane);
    // the 4 R histogram dwords of a batch of `nrows` rows at hp_ (rows past the class re-read its last row; their bits are
    // dropped below)
    auto load_batch = [&](uint32_t (&h)[4 * R], const uint32_t* __restrict__ hp_, int nrows) {
      if (nrows >= R) {
#pragma unroll
        for (int q = 0; q < 4 * R; ++q) h[q] = hp_[q];
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t* __restrict__ hr_ = hp_ + 4 * min(r, nrows - 1);
#pragma unroll
          for (int q = 0; q < 4; ++q) h[4 * r + q] = hr_[q];
        }
      }
    };
    for (int i = a; i < b;) {
      for (; i < b && q_cnt <= kC3cStack - kWave; i += R, hp += 4 * R) {
        const int nrows = min(R, b - i);
        uint32_t h[4 * R];
        load_batch(h, hp, nrows);
        // acc: bit (R T - 1 - k) set = pair k = (row i + r, tile t), k = r T + t, passed the coarse test
        uint32_t acc = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
          for (int t = 0; t < T; ++t) {
            uint32_t l1 = seed[t];
#pragma unroll
            for (int q = 0; q < 4; ++q) l1 = __builtin_amdgcn_sad_u8(h[4 * r + q], hc[t][q], l1);
            acc = __builtin_amdgcn_alignbit(acc, l1, 31);  // acc = acc << 1 | sign(l1)
          }
        }
        if (nrows < R) acc &= ~((1u << ((R - nrows) * T)) - 1u);
        const bool nz = acc != 0u;
        const unsigned long long m = __ballot(nz);
        if (m != 0ull) {  // one ds_write per batch: (first row, lane, mask)
          if (nz) {
            const int slot = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), static_cast<uint32_t>(q_cnt)));
            stack[slot] = (static_cast<unsigned long long>((static_cast<uint32_t>(i - i0) << 13) | lane_field) << 32) | acc;
          }
          q_cnt += __popcll(m);
        }
      }
      while (q_cnt >= kWave) drain_pass();  // full passes only: the rest waits for more
    }
  }
  while (q_cnt > 0) drain_pass();
}

}  // namespace nsm
