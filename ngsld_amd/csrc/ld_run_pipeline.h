// ld_run_pipeline.h -- what the kernels that work through RUNS of items share (pair_ld_run_kernel, pair_ld_group_kernel, the
// a/b kernels of ld_pair_ab.hip, the genotype-combination kernel of ld_pair_hard.hip): the asynchronous global -> LDS copy of a
// site, the ring of results, the run's claim list.
#pragma once

#include "ld_em.h"

namespace ngsld {

// ---------------------------------------------------------------------------------------------
// One wavefront per pair (n_ind <= 640): the four wavefronts of a workgroup work on ONE row s1, whose vector sits in LDS.
// Each wavefront claims the next s2 from an LDS counter (dynamic balance of the 3..100-iteration spread), and as soon as
// it has turned the current buffer into P it starts the asynchronous copy (global_load_lds, 16 B per lane, no VGPR round
// trip) of the site it will work on NEXT -- the copy flies during the whole EM loop, so the ~2.5 us HBM/Infinity-Cache
// latency that a direct load pays at every pair start is off the critical path (measured: 1.62e8 against 1.42e8 pairs/s).
// ---------------------------------------------------------------------------------------------
// Asynchronous copy of one site's planes (SLOTS*1536 B, contiguous) into LDS, 1 KiB per wave-instruction
// (lane l moves 16 B to lds_dst + k*1024 + l*16).  With `stride` > 1 only chunks k % stride == first are
// issued (several wavefronts sharing one copy).  A trailing half chunk (odd SLOTS) is issued by lanes 0..31.
template <int SLOTS>
__device__ __forceinline__ void dma_site_to_lds(const double *site, char *lds_dst, int lane, int first, int stride) {
  constexpr int kBytes = SLOTS * 64 * 3 * 8;
  constexpr int kChunks = (kBytes + 1023) / 1024;
  const char *g = reinterpret_cast<const char *>(site) + lane * 16;
  // The instruction's immediate offset applies to the global AND the LDS address, and the copy is contiguous on both
  // sides: four chunks share one address pair (offsets 0 .. 3072 fit the 12-bit field) instead of one 64-bit add and
  // one M0 write per chunk.
#pragma unroll
  for (int k0 = 0; k0 < kChunks; k0 += 4) {
    glb_void_t *gb = (glb_void_t *)(g + k0 * 1024);
    lds_void_t *lb = (lds_void_t *)(lds_dst + k0 * 1024);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = k0 + kk;
      if (k >= kChunks) break;
      if (stride != 1 && (k % stride) != first) continue;
      if ((k + 1) * 1024 <= kBytes || lane * 16 < kBytes - k * 1024) {
        switch (kk) {
          case 0: __builtin_amdgcn_global_load_lds(gb, lb, 16, 0, 0); break;
          case 1: __builtin_amdgcn_global_load_lds(gb, lb, 16, 1024, 0); break;
          case 2: __builtin_amdgcn_global_load_lds(gb, lb, 16, 2048, 0); break;
          default: __builtin_amdgcn_global_load_lds(gb, lb, 16, 3072, 0); break;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Run kernel (n_ind <= 640; eight slots = the headline shape): the pair pipeline above without per-item costs.
// A workgroup works through a RUN of up to kRunItems consecutive items of one row (512 candidate sites) instead of
// one item: the row vector is brought into LDS once per run, the four wavefronts claim candidates from one LDS
// counter for the whole run, and NOTHING inside the run synchronises them -- no barrier at item boundaries, no
// workgroup turnover every 64 pairs (with one item per workgroup the SIMDs held 1.8 of 2 wavefronts on average:
// launch, the barrier at the item's end and the wait for its slowest wavefront).
//   * item headers of the run sit in LDS (claims need mask / count / first_record; a global load per claim would be
//     a ~2 us round trip on the critical path);
//   * a site's scalars {maf, mean_e, rsx} travel with its planes: one more 32-byte global->LDS copy behind the site
//     copy, so the pair loop has no ordinary global load to wait for at all;
//   * results collect in a wave-private LDS ring and are turned into records 32 at a time, one LANE per pair
//     (write_pair is ~100 wavefront-uniform f64 instructions: issued per pair they would cost 4 % of the kernel).
//   LDS: [row vector][4 x (site buffer + 32 B scalars)][4 x ring of 32 results][item headers][claim counter]
// ---------------------------------------------------------------------------------------------
struct RunResult {
  double f[4], sxy, rsx2;
  uint32_t x, n_iter;
  uint64_t rec;
};

// The computed pairs of a run, as a list: cand[j] = candidate index (64 * item + offset = s2 - s2 of the run's first
// candidate) of the run's j-th computed pair, in increasing s2 -- so its record is simply the run's first record + j.
// Built once per run from the items' masks by the whole workgroup; a claim is then one LDS atomic and one 2-byte read
// whatever the masks look like.  (Claiming candidate by candidate and skipping the masked-out ones cost a dependent LDS
// round trip per dropped candidate: -11 % at --rnd_sample 0.1, -49 % at 0.02.)
struct RunList {
  uint16_t cand[kRunItems * 64];
  uint32_t base[kRunItems + 1];  // computed pairs before each item; base[n_items] = all of the run's
  uint32_t claim;
  uint32_t pad[2];
  Item items[kRunItems];         // the run's item headers
};

// Called by all 256 threads; ends with a barrier (which also completes whatever global->LDS copies the callers issued
// before it: __syncthreads waits for the wavefront's own memory operations first).
__device__ __forceinline__ void build_run_list(RunList *L, const Item *g_items, uint32_t n_items) {
  if (threadIdx.x < n_items * 2)  // item headers, 16 bytes per thread
    reinterpret_cast<uint4 *>(L->items)[threadIdx.x] = reinterpret_cast<const uint4 *>(g_items)[threadIdx.x];
  if (threadIdx.x == 0) L->claim = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t k = 0; k < n_items; ++k) {
      L->base[k] = acc;
      acc += (uint32_t)__popcll(L->items[k].mask);  // bits at or beyond `count` are never set (items_kernel)
    }
    L->base[n_items] = acc;
  }
  __syncthreads();
  for (uint32_t idx = threadIdx.x; idx < n_items * 64; idx += 256) {
    const uint32_t k = idx >> 6, c = idx & 63u;
    const unsigned long long m = L->items[k].mask;
    if ((m >> c) & 1ull) L->cand[L->base[k] + (uint32_t)__popcll(m & ((1ull << c) - 1ull))] = (uint16_t)idx;
  }
  __syncthreads();
}

}  // namespace ngsld
