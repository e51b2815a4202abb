// Levels-mode Indel-ratio grid, split path (one-word level strings): the FINISH kernel.
//
// Reference: types/comparable_data.py:248-265 (compare_terms) x compare/score_functions.py:20-27 (fuzzy_match); the
// steps 2..S of the pairs that the scan kernel (indel_levels_park_kernel<1, SPLIT>) found alive after step 1.
//
// The fused kernel finishes its survivors in "dense passes" between block barriers: per pass the parked pairs of ONE
// batch of 8 left rows (they share the 8 match-mask tables), ~18 of 64 lanes filled at configs[4]'s survival rate,
// and the pass's registers on top of the scan's are what pushes that kernel into scratch (260 B per lane, 234 GB
// of HBM traffic per launch).  Here the survivors come from a global queue, 64 per wavefront whatever rows they belong to:
//
//   lane = one pair.  Each lane owns a column of a match-mask table in LDS -- [code][plane][lane] 32-bit words, plane 0 =
//   pattern positions 0..31, plane 1 = 32..63; a lane's accesses fall into its own bank, whatever codes the lanes read --
//   builds the masks of its OWN pattern with ds_or (the pattern is the shorter of the pair's two level strings: the LCS is
//   symmetric, and 32-bit words suffice unless both strings are longer than 32), and runs Hyyro's recurrence over its own
//   text.  No wave-uniform operand, no table shared between lanes, hence no grouping of the queue by left row.
//
// (Tried: ONE table plane per wave -- 10 KB instead of 20 KB of LDS, 16 waves per CU -- with two sweeps over the text and
// one carry bit per text position when both strings are longer than 32: configs[4]'s fuzzy grids 289 -> 295 ms; the
// levels of steps 2 and 3 are mostly longer than 32 code units there, and two table builds cost more than the 64-bit
// recurrence.)
//
// Arithmetic and tests are those of the fused kernel's dense steps (same double operations in the same order; every
// test that drops a pair is an upper bound), so the hits are identical.
#pragma once

namespace nsm {

struct FinishParams {
  int32_t pm_stride;  // table entries: alphabet + 1 (the pad code) rounded up to 8
  int32_t pad_code;
  int32_t use_hist;
  double threshold;
  unsigned long long cap;
  unsigned long long qcap;
};

constexpr int kFinishBlocks = 2048;  // one wave per block, 8 blocks per CU (20 KB of LDS each at 37 symbols)

__global__ __launch_bounds__(kWave) void indel_levels_finish_kernel(
    const int32_t* __restrict__ lfirst, const int32_t* __restrict__ lnlev, const int32_t* __restrict__ lorig,
    const uint8_t* __restrict__ lcodes, const int32_t* __restrict__ llen, const uint8_t* __restrict__ lhist,
    const int32_t* __restrict__ rfirst, const int32_t* __restrict__ rnlev, const int32_t* __restrict__ rorig,
    const uint8_t* __restrict__ rcodes, const int32_t* __restrict__ rlen, const uint8_t* __restrict__ rhist,
    nsm_hit* __restrict__ hits, unsigned long long* __restrict__ count, const unsigned long long* __restrict__ queue,
    const unsigned long long* __restrict__ qcount, const int* __restrict__ qflag, const FinishParams p) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_tab[];  // [pm_stride][2][64]
  if (*qflag != 0) return;  // the queue overflowed: entries are missing, the fused kernel redoes the whole grid
  const int lane = threadIdx.x;
  unsigned long long n = *qcount;
  if (n > p.qcap) n = p.qcap;
  const bool use_hist = p.use_hist != 0;

  auto load_row = [](const uint8_t* __restrict__ codes, int row, uint32_t (&w)[16]) {
    const uint4* tp = reinterpret_cast<const uint4*>(codes + static_cast<size_t>(row) * 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = tp[q];
      w[4 * q + 0] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
  };
  auto step_ub = [&](int lrow, int rrow, int la_, int lb_) -> float {
    if (!use_hist) return 1.0f;
    uint32_t a[8], b[8];
    load_hist<8>(lhist, lrow, a);
    load_hist<8>(rhist, rrow, b);
    return hist_ratio_ub(hist_l1<8>(a, b), la_, lb_);
  };

  // LCS of the lane's two level strings (lanes with want = false take part with empty strings)
  auto lane_lcs = [&](bool want, int lrow, int la, int rrow, int lb) -> int {
    uint32_t pat[16], txt[16];
    load_row(lcodes, lrow, pat);
    load_row(rcodes, rrow, txt);
    int np = want ? la : 0, nt = want ? lb : 0;
    if (np > nt) {  // the shorter string is the pattern
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const uint32_t t = pat[q];
        pat[q] = txt[q];
        txt[q] = t;
      }
      const int t = np;
      np = nt;
      nt = t;
    }
    const int np_max = wave_max_i32(np), nt_max = wave_max_i32(nt);
    const bool wide = np_max > 32;
    uint32_t* col = s_tab + lane;
    for (int c = 0; c < p.pm_stride; ++c) {
      col[(2 * c) * kWave] = 0u;
      if (wide) col[(2 * c + 1) * kWave] = 0u;
    }
    // masks: positions past the pattern's end hold the pad code, whose entry is cleared again below
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if (g * 4 < np_max) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t c = (pat[g] >> (8 * k)) & 0xffu;
          atomicOr(&col[(2 * c + (g >> 3)) * kWave], 1u << ((4 * g + k) & 31));
        }
      }
    }
    col[(2 * p.pad_code) * kWave] = 0u;
    col[(2 * p.pad_code + 1) * kWave] = 0u;
    int lcs;
    if (!wide) {
      uint32_t v = ~0u;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        if (g * 4 < nt_max) {
          uint32_t m[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) m[k] = col[(2 * ((txt[g] >> (8 * k)) & 0xffu)) * kWave];
#pragma unroll
          for (int k = 0; k < 4; ++k) v = lcs_step32(v, m[k]);
        }
      }
      lcs = 32 - __popc(v);
    } else {
      unsigned long long v = ~0ull;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        if (g * 4 < nt_max) {
          unsigned long long m[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t c = (txt[g] >> (8 * k)) & 0xffu;
            m[k] = static_cast<unsigned long long>(col[(2 * c) * kWave]) |
                   (static_cast<unsigned long long>(col[(2 * c + 1) * kWave]) << 32);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) v = lcs_step64(v, m[k]);
        }
      }
      lcs = 64 - __popcll(v);
    }
    return lcs;
  };

  // Chunks of 256 entries (four passes) are dealt to the blocks round-robin: a scan wave flushes ~200 entries at a time, all
  // from its 64 right items and a few neighbouring left rows, so the passes of one chunk gather the same level strings and
  // histograms (64-entry passes dealt round-robin spread them over the 8 XCDs' L2s; one contiguous range per block keeps
  // them together but balances worse: 255 vs 251 ms).
  constexpr unsigned long long kChunk = 4 * kWave;
  for (unsigned long long base0 = static_cast<unsigned long long>(blockIdx.x) * kChunk; base0 < n;
       base0 += static_cast<unsigned long long>(gridDim.x) * kChunk)
  for (unsigned long long base = base0; base < n && base < base0 + kChunk; base += kWave) {
    const bool active = base + lane < n;
    const unsigned long long e = queue[active ? base + lane : base];
    const int i = static_cast<int>(e >> 31);
    const int jr = static_cast<int>((e >> 7) & ((1u << kQueueRowBits) - 1u));
    const int lcs1 = static_cast<int>(e & 127u);
    const int ll = lnlev[i], lf = lfirst[i];
    const int lrj = rnlev[jr], rr0 = rfirst[jr];
    // the score of step 1 is 2^-1 * ratio
    double score = indel_score_dev(llen[lf + max(0, min(1, ll - 1))], rlen[rr0 + max(0, min(1, lrj - 1))], lcs1) * 0.5;
    const int S = max(ll, lrj);
    const int s_hi = wave_max_i32(active ? S : 0);
    int prev_a = -1, prev_b = -1;
    double ratio = 0.0;
    bool alive = active;  // still a candidate: running, or finished with its final score
    double factor = 0.5;
    for (int s = 2; s <= s_hi; ++s) {
      factor *= 0.5;
      const bool run = alive && s <= S;
      if (!__any(run)) break;
      const int a = max(0, min(s, ll - 1)), b = max(0, min(s, lrj - 1));
      const bool fresh = run && (a != prev_a || b != prev_b);
      if (__any(fresh)) {
        const int lrow = lf + a, rrow = rr0 + b;
        const int la = llen[lrow], lbj = rlen[rrow];
        const int lcs = lane_lcs(fresh, lrow, la, rrow, lbj);
        if (fresh) {
          ratio = indel_score_dev(la, lbj, lcs);
          prev_a = a;
          prev_b = b;
        }
      }
      if (run) {
        score += ratio * factor;
        // steps still to come: histogram bound of the next level pair (exact upper bound; 1e-6 covers the float
        // arithmetic of the bound and the rounding of the double sum)
        float rest = 0.0f;
        if (s < S) {
          const int lrow_n = lf + max(0, min(s + 1, ll - 1)), rrow_n = rr0 + max(0, min(s + 1, lrj - 1));
          rest = rest_bound(s, S, step_ub(lrow_n, rrow_n, llen[lrow_n], rlen[rrow_n]));
        }
        alive = score + static_cast<double>(rest) + 1e-6 >= p.threshold;
      }
    }
    emit_hits_wave(hits, p.cap, count, active && alive && score >= p.threshold, score, lorig[i], rorig[jr]);
  }
}

// first and last launch of the split path: remember the hit counter / put it back when the queue overflowed (the fused
// kernel that follows then appends every hit of the grid again)
__global__ void split_begin_kernel(unsigned long long* __restrict__ ctl, int words, const unsigned long long* __restrict__ count) {
  for (int t = threadIdx.x; t < words; t += blockDim.x) ctl[t] = t == 0 ? *count : 0ull;
}
__global__ void split_end_kernel(const unsigned long long* __restrict__ ctl, unsigned long long* __restrict__ count) {
  if (threadIdx.x == 0 && *reinterpret_cast<const int*>(ctl + 1) != 0) *count = ctl[0];
}

}  // namespace nsm
