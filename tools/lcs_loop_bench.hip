// Micro-benchmark of the shared-tile kernel's inner loops (napkon-string-matching_amd/csrc/indel_tile_lcs.hpp) in isolation:
// SIMD-cycles per 4-code-unit iteration of the two-row / one-row LCS pass at 1..4 waves per SIMD, against the loop's
// pure issue cost.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I napkon-string-matching_amd/csrc -I include tools/lcs_loop_bench.hip -o /tmp/lcs_loop_bench && /tmp/lcs_loop_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "indel_tile_lcs.hpp"

namespace nsm {
void set_error(const char*, ...) {}
int hip_status(hipError_t e, const char*) { return static_cast<int>(e); }
}  // namespace nsm
using namespace nsm;

constexpr int K = 4;
constexpr int kChars = 256;
constexpr int kEntries = 32;

// MODE 0: two rows per pass (tile_lcs2<K, L>); 1: one row (tile_lcs1<K, L>); 2: two rows, masks NOT from LDS (one
// register pair per code unit: the recurrence alone); 3: one row per lane = the dense pass (per-lane table of 4)
template <int L, int MODE>
__global__ __launch_bounds__(1024) void bench(int* out, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_mem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  uint32_t* img = reinterpret_cast<uint32_t*>(s_mem);                    // [64 dwords][64 lanes]
  unsigned long long* tables = s_mem + 64 * 64 / 2;                        // per wave: 4 tables
  const int tbl_entries = kEntries * kTileWords<K>;
  unsigned long long* pm = tables + static_cast<size_t>(wave) * 4 * tbl_entries;
  for (int c = lane; c < 4 * tbl_entries; c += 64) pm[c] = 0x9E3779B97F4A7C15ull * (c + 1 + wave);
  if (wave == 0)
    for (int w = 0; w < 64; ++w) {
      uint32_t x = (lane * 2654435761u + w * 40503u) ^ (w << 7);
      img[w * 64 + lane] = (x & 0x0f0f0f0fu) | ((x >> 9) & 0x10101010u & 0u);  // symbols 0..15 (bank-conflict-free for b128)
    }
  __syncthreads();
  int acc = 0;
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) {
      int a, b;
      tile_lcs2<K, L, false>(pm, tbl_entries, img + lane, kChars, a, b);
      acc += a + b;
    } else if (MODE == 1) {
      acc += tile_lcs1<K, L>(pm, img + lane, 64, kChars);
    } else if (MODE == 3) {
      acc += tile_lcs1<K, L>(pm + (lane & 3) * tbl_entries, img + ((lane * 7) & 63), 64, kChars);
    } else {
      uint32_t va[L], vb[L], m[L];
      for (int k = 0; k < L; ++k) { va[k] = vb[k] = ~0u; m[k] = lane * 0x01010101u + k + r; }
      for (int w = 0; w < kChars; ++w) {
        limb_step<L>(va, m);
        limb_step<L>(vb, m);
        asm volatile("" : "+v"(m[0]));
      }
      acc += limb_zeros<L>(va) + limb_zeros<L>(vb);
    }
    asm volatile("" : "+v"(acc));
  }
  if (acc == 0x7fffffff) out[0] = acc;
}

template <int L, int MODE>
void run(const char* what, double issue_cycles_per_iter) {
  int* out;
  hipMalloc(&out, 4);
  const int reps = 200;
  for (int waves : {4, 8, 12, 16}) {
    const size_t lds = 64 * 64 * 4 + static_cast<size_t>(waves) * 4 * kEntries * kTileWords<K> * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&bench<L, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((bench<L, MODE>), dim3(256), dim3(waves * 64), lds, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((bench<L, MODE>), dim3(256), dim3(waves * 64), lds, 0, out, reps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double iters = 256.0 * waves * reps * (kChars / 4);  // 4-code-unit iterations (MODE 2: the same count of code units)
    const double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;
    printf("%-34s L=%d waves/CU=%2d  %8.3f ms  %7.1f SIMD-cycles per iteration (issue cost ~%.0f)  err=%d\n", what, L, waves, ms,
           simd_cycles / iters, issue_cycles_per_iter, static_cast<int>(hipGetLastError()));
  }
  hipFree(out);
}

int main() {
  // issue cost model: e32 and / xor / or 2.3 cycles, v_add_co / v_addc_co 4.6, address ops ~11 per code unit
  run<2, 0>("two rows, LDS masks", 4 * (2 * 2 * (3 * 2.3 + 4.6) + 11));
  run<3, 0>("two rows, LDS masks", 4 * (2 * 3 * (3 * 2.3 + 4.6) + 11));
  run<4, 0>("two rows, LDS masks", 4 * (2 * 4 * (3 * 2.3 + 4.6) + 11));
  run<4, 2>("two rows, recurrence only", 4 * (2 * 4 * (3 * 2.3 + 4.6)));
  run<2, 2>("two rows, recurrence only", 4 * (2 * 2 * (3 * 2.3 + 4.6)));
  run<4, 1>("one row, LDS masks", 4 * (4 * (3 * 2.3 + 4.6) + 11));
  run<4, 3>("one row, per-lane table (dense)", 4 * (4 * (3 * 2.3 + 4.6) + 11));
  run<6, 3>("one row, per-lane table (dense)", 4 * (6 * (3 * 2.3 + 4.6) + 11));
  return 0;
}
