// ld_kernel_stream.h -- the streaming kernels for cohorts beyond 5,120 individuals: pair_ld_stream_kernel (both vectors re-read
// every iteration) and pair_ld_bres_kernel (the candidate's vector resident in registers); instantiated in ld_pair_stream.hip.
#pragma once

#include "ld_em.h"

namespace ngsld {

// ---------------------------------------------------------------------------------------------
// Streaming kernel (n_ind > 10,240; 5,121 .. 10,240 go to pair_ld_bres_kernel below): no limit on the number of individuals.  One 256-thread workgroup per pair at a
// time; wavefront w takes the 64-individual blocks w, w+4, w+8, ...  P does not fit in registers any more, so
// every EM iteration re-reads both site vectors (from L2: a pair's two vectors are 48*n_ind bytes) and forms
//   s = sum_g1 a[g1] * (sum_g2 W[g1][g2] b[g2]),   R[g1][g2] += (r a[g1]) * b[g2]
// on the fly: 24 f64 VALU + rcp + 6 loads per individual and iteration instead of 21 + rcp from registers.
// Same reduction order rules as the other kernels (fixed, deterministic).
// (Round 3 tried ONE wavefront per pair instead -- no exchange, no barrier, the four wavefronts of a workgroup walking the same
// row vector so that a neighbour's lines in the CU's L1 would serve the a-loads: -15..-17 % at 5,121..10,000 individuals,
// profiles/r03/sweep_stream.txt.  A wavefront streaming a whole pair alone has a quarter of the loads in flight per pair, and
// the L1 sharing did not happen.)
// ---------------------------------------------------------------------------------------------
template <bool MASKED>
__global__ __launch_bounds__(256, 2) void pair_ld_stream_kernel(PairArgs A) {
  __shared__ double xch[2][4][4];
  __shared__ double xch0[4][2];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const Item it = A.items[blockIdx.x];
  const uint32_t s1 = it.s1;
  const double m1 = A.maf[s1];
  const double mean1 = A.mean_e[s1];
  const double rsx1 = A.rsx[s1];
  const uint64_t rec0 = it.first_record - A.out_base;
  const double *pa = A.planes + (uint64_t)s1 * A.site_stride;
  const uint32_t np = A.np;
  const uint32_t n_blocks = np / 64;

  for (uint32_t c = 0; c < it.count; ++c) {
    if (!((it.mask >> c) & 1ull)) continue;  // ngsLD.cpp:270-282
    const uint32_t s2 = it.s2_begin + c;
    const double *pb = A.planes + (uint64_t)s2 * A.site_stride;
    const double m2 = A.maf[s2], mean2 = A.mean_e[s2], rsx2 = A.rsx[s2];

    // ---- pass 0: individuals with data, Pearson cross moment ----
    uint32_t x = 0;
    double sxy = 0.0;
    for (uint32_t b = (uint32_t)wave; b < n_blocks; b += 4) {
      const uint32_t i = b * 64 + (uint32_t)lane;
      const double a0 = pa[i], a1 = pa[np + i], a2 = pa[2 * np + i];
      const double b0 = pb[i], b1 = pb[np + i], b2 = pb[2 * np + i];
      const bool inb = i < A.n_ind;
      bool ok = inb;
      if (MASKED) ok = inb && !miss_data(a0, a1, a2) && !miss_data(b0, b1, b2);
      x += (uint32_t)__popcll(__ballot(ok));
      const double c1 = inb ? fma(2.0, a2, a1) - mean1 : 0.0;
      const double c2 = inb ? fma(2.0, b2, b1) - mean2 : 0.0;
      sxy = fma(c1, c2, sxy);
    }
    sxy = wave_sum1(sxy);
    if (lane == 0) {
      xch0[wave][0] = sxy;
      xch0[wave][1] = (double)x;
    }
    __syncthreads();
    sxy = ((xch0[0][0] + xch0[1][0]) + xch0[2][0]) + xch0[3][0];
    x = (uint32_t)(((xch0[0][1] + xch0[1][1]) + xch0[2][1]) + xch0[3][1]);
    __syncthreads();

    // ---- haplo_freq (gen_func.cpp:1027-1059) ----
    double f0 = (1 - m1) * (1 - m2), f1 = (1 - m1) * m2, f2 = m1 * (1 - m2), f3 = m1 * m2;
    if (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1) {
      if (threadIdx.x == 0) atomicExch(A.status, (int)NGSLD_ERR_MAF_RANGE);
      f0 = f1 = f2 = f3 = __builtin_nan("");
    }
    const double inv_x = 1.0 / (double)x;
    bool bad = false, tie = false;
    uint32_t n_iter = 0;
    for (; n_iter < (uint32_t)kIterMax; ++n_iter) {
      const double p00 = f0 * f0, p01 = f0 * f1, p02 = f0 * f2, p03 = f0 * f3, p11 = f1 * f1;
      const double p12 = f1 * f2, p13 = f1 * f3, p22 = f2 * f2, p23 = f2 * f3, p33 = f3 * f3;
      const double w1 = p01 + p01, w3 = p02 + p02, w4 = 2.0 * (p03 + p12), w5 = p13 + p13, w7 = p23 + p23;
      double R0 = 0, R1 = 0, R2 = 0, R3 = 0, R4 = 0, R5 = 0, R6 = 0, R7 = 0, R8 = 0;
      // Four blocks of 64 individuals per trip: their 24 loads are in flight together, and an individual that does not count
      // (padding, no data) takes part with r = 0 instead of being branched around -- one block per trip waited an L2 round
      // trip for every 64 individuals, and its branch kept the loads of the next block behind the arithmetic of this one.
      // (Same additions in the same order: adding +0 changes nothing.)
      for (uint32_t bq = (uint32_t)wave; bq < n_blocks; bq += 16) {
        double av[4][3], bv[4][3];
        bool okv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t b = bq + 4u * (uint32_t)u;
          const bool in = b < n_blocks;
          const uint32_t i = (in ? b : bq) * 64 + (uint32_t)lane;
          av[u][0] = pa[i]; av[u][1] = pa[np + i]; av[u][2] = pa[2 * np + i];
          bv[u][0] = pb[i]; bv[u][1] = pb[np + i]; bv[u][2] = pb[2 * np + i];
          okv[u] = in && i < A.n_ind;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double a0 = av[u][0], a1 = av[u][1], a2 = av[u][2], b0 = bv[u][0], b1 = bv[u][1], b2 = bv[u][2];
          bool ok = okv[u];
          if (MASKED) ok = ok && !miss_data(a0, a1, a2) && !miss_data(b0, b1, b2);
          const double v0 = fma(p11, b2, fma(w1, b1, p00 * b0));  // sum_g2 W[0][g2] b[g2]
          const double v1 = fma(w5, b2, fma(w4, b1, w3 * b0));
          const double v2 = fma(p33, b2, fma(w7, b1, p22 * b0));
          const double s = fma(a2, v2, fma(a1, v1, a0 * v0));
          const double r = ok ? rcp_refined(s) : 0.0;
          const double r0 = r * a0, r1 = r * a1, r2 = r * a2;
          R0 = fma(r0, b0, R0); R1 = fma(r0, b1, R1); R2 = fma(r0, b2, R2);
          R3 = fma(r1, b0, R3); R4 = fma(r1, b1, R4); R5 = fma(r1, b2, R5);
          R6 = fma(r2, b0, R6); R7 = fma(r2, b1, R7); R8 = fma(r2, b2, R8);
        }
      }
      double t0 = fma(p03, R4, fma(p02, R3, fma(p01, R1, p00 * R0)));
      double t1 = fma(p13, R5, fma(p12, R4, fma(p11, R2, p01 * R1)));
      double t2 = fma(p23, R7, fma(p22, R6, fma(p12, R4, p02 * R3)));
      double t3 = fma(p33, R8, fma(p23, R7, fma(p13, R5, p03 * R4)));
      wave_sum4(t0, t1, t2, t3);
      const int par = (int)(n_iter & 1u);
      if (lane == 0) {
        xch[par][wave][0] = t0; xch[par][wave][1] = t1; xch[par][wave][2] = t2; xch[par][wave][3] = t3;
      }
      __syncthreads();
      t0 = ((xch[par][0][0] + xch[par][1][0]) + xch[par][2][0]) + xch[par][3][0];
      t1 = ((xch[par][0][1] + xch[par][1][1]) + xch[par][2][1]) + xch[par][3][1];
      t2 = ((xch[par][0][2] + xch[par][1][2]) + xch[par][2][2]) + xch[par][3][2];
      t3 = ((xch[par][0][3] + xch[par][1][3]) + xch[par][2][3]) + xch[par][3][3];
      const double n0 = t0 * inv_x, n1 = t1 * inv_x, n2 = t2 * inv_x, n3 = t3 * inv_x;
      const double sn = (n0 + n1) + (n2 + n3);
      if (__builtin_amdgcn_readfirstlane((int)!(sn < 2.0))) {  // the reference's all-NaN step (see em_pair)
        bad = true;
        break;
      }
      const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
      f0 = n0; f1 = n1; f2 = n2; f3 = n3;
      if (fabs(eps - kEpsilon) < kTieMargin) tie = true;
      if (__builtin_amdgcn_readfirstlane((int)(eps < kEpsilon))) break;
    }
    if (bad) f0 = f1 = f2 = f3 = __builtin_nan("");
    if (threadIdx.x == 0)
      write_pair(A, rec0 + (uint64_t)__popcll(it.mask & ((1ull << c) - 1ull)), f0, f1, f2, f3, sxy, rsx1, rsx2, x,
                 n_iter | (tie ? kTieBit : 0u));
  }
}

// ---------------------------------------------------------------------------------------------
// Streaming kernel with the CANDIDATE's vector resident (5,121 .. 10,240 individuals; ld_pair_stream.hip).  Half of what
// the streaming kernel reads in every EM iteration never changes during a pair: the candidate site's vector b.  Eight
// wavefronts share a pair here -- wavefront w takes the 64-individual blocks w, w + 8, ... -- and with at most 20 blocks
// per wavefront b fits in registers (6 VGPRs per slot), loaded once in the pass that counts the individuals and forms the
// Pearson moment.  An iteration then reads the row vector a only: 24 bytes per individual from L2 instead of 48, half the
// load instructions, and the step itself is the streaming kernel's (same four-value form, one reciprocal per individual).
// One workgroup per item of 16 candidates, as there.
// ---------------------------------------------------------------------------------------------
template <int SLOTS, bool MASKED, bool TAIL = false>
__global__ __launch_bounds__(512, 2) void pair_ld_bres_kernel(PairArgs A) {
  constexpr int kWaves = 8;
  constexpr int kChunk = SLOTS <= 14 ? 4 : (SLOTS <= 17 ? 3 : 2), kChunks = (SLOTS + kChunk - 1) / kChunk;  // slots whose row values travel together
  __shared__ double xch[2][kWaves][4];
  __shared__ double xch0[kWaves][2];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const Item it = A.items[blockIdx.x];
  const uint32_t s1 = it.s1;
  const double m1 = A.maf[s1];
  const double mean1 = A.mean_e[s1];
  const double rsx1 = A.rsx[s1];
  const uint64_t rec0 = it.first_record - A.out_base;
  typedef const __attribute__((address_space(1))) double gdouble_t;  // global_load with an SGPR base + one 32-bit lane offset
  const uint32_t np = A.np;
  // the three plane bases of a site as wavefront-uniform pointers (see stage_pair, A_GLOBAL: left to itself the compiler keeps a
  // 64-bit VGPR address per load of the unrolled loops)
  const double *pa = A.planes + (uint64_t)s1 * A.site_stride;
  gdouble_t *pa0 = (gdouble_t *)uniform_ptr(pa), *pa1 = (gdouble_t *)uniform_ptr(pa + np), *pa2 = (gdouble_t *)uniform_ptr(pa + 2 * np);
  const uint32_t n_blocks = np / 64;  // (> 8 * (SLOTS - 1): the launcher picked SLOTS = ceil(n_blocks / 8); TAIL: > 8 * SLOTS)
  // slot j of this wavefront is block j * 8 + wave; only the last slot can lie beyond the planes (then it re-reads slot 0's
  // block and counts for nothing)
  auto index_of = [&](int j) -> uint32_t {
    const uint32_t blk = (uint32_t)(j * kWaves + wave);
    return ((TAIL || j < SLOTS - 1 || blk < n_blocks) ? blk : (uint32_t)wave) * 64u + (uint32_t)lane;
  };
  // TAIL (more than 10,240 individuals): the blocks beyond the 8 * SLOTS resident ones are streamed as in the plain kernel --
  // both vectors from memory in every iteration, two blocks per trip, after the resident slots (a fixed order of additions)
  constexpr uint32_t kTail0 = (uint32_t)(SLOTS * kWaves);
  // Every load below is SGPR base + 32-bit byte offset of the slot (one VGPR per slot, shared by the six planes).  The offsets
  // never change, and that is what has to be hidden from the compiler: loop-invariant code motion otherwise forms each load's
  // 64-bit address once, in front of the loops, where nothing folds it into the addressing mode any more -- 6 * SLOTS
  // registers of addresses, spilled and reloaded one by one in front of their loads.
  typedef const __attribute__((address_space(1))) char gchar_t;
  uint32_t off[SLOTS];
#pragma unroll
  for (int j = 0; j < SLOTS; ++j) off[j] = index_of(j) * 8u;
  auto hide_offsets = [&]() {
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) asm volatile("" : "+v"(off[j]));
  };
  auto ld = [&](gdouble_t *base, int j) -> double { return *(gdouble_t *)((gchar_t *)base + off[j]); };

  auto a_of = [&](int j, int g) -> double { return ld(g == 0 ? pa0 : (g == 1 ? pa1 : pa2), j); };

  for (uint32_t c = 0; c < it.count; ++c) {
    if (!((it.mask >> c) & 1ull)) continue;  // ngsLD.cpp:270-282
    const uint32_t s2 = it.s2_begin + c;
    const double *pb = A.planes + (uint64_t)s2 * A.site_stride;
    gdouble_t *pb0 = (gdouble_t *)uniform_ptr(pb), *pb1 = (gdouble_t *)uniform_ptr(pb + np), *pb2 = (gdouble_t *)uniform_ptr(pb + 2 * np);
    const double m2 = A.maf[s2], mean2 = A.mean_e[s2], rsx2 = A.rsx[s2];

    hide_offsets();
    // ---- pass 0: b into registers; individuals with data, Pearson cross moment (ngsLD.cpp:290: over ALL individuals) ----
    // (kChunk slots at a time, here and in the iterations: with every load of the unrolled loop hoisted to its top the row
    // vector's values alone would take 6 * SLOTS registers beside b's 6 * SLOTS)
    double bv[SLOTS][3];
    uint32_t vbits = 0, x = 0;
    double sxy = 0.0;
#pragma unroll
    for (int j0 = 0; j0 < SLOTS; j0 += kChunk) {
#pragma unroll
      for (int j = j0; j < j0 + kChunk && j < SLOTS; ++j) {
        const double a0 = a_of(j, 0), a1 = a_of(j, 1), a2 = a_of(j, 2);
        bv[j][0] = ld(pb0, j); bv[j][1] = ld(pb1, j); bv[j][2] = ld(pb2, j);
        const bool inb = TAIL || ((j < SLOTS - 1 || (uint32_t)(j * kWaves + wave) < n_blocks) && index_of(j) < A.n_ind);
        bool ok = inb;
        if (MASKED) ok = inb && !miss_data(a0, a1, a2) && !miss_data(bv[j][0], bv[j][1], bv[j][2]);
        vbits |= (ok ? 1u : 0u) << j;
        x += (uint32_t)__popcll(__ballot(ok));
        const double c1 = inb ? fma(2.0, a2, a1) - mean1 : 0.0;
        const double c2 = inb ? fma(2.0, bv[j][2], bv[j][1]) - mean2 : 0.0;
        sxy = fma(c1, c2, sxy);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (TAIL) {
      for (uint32_t blk = kTail0 + (uint32_t)wave; blk < n_blocks; blk += (uint32_t)kWaves) {
        const uint32_t i = blk * 64u + (uint32_t)lane;
        const double a0 = pa0[i], a1 = pa1[i], a2 = pa2[i], b0 = pb0[i], b1 = pb1[i], b2 = pb2[i];
        const bool inb = i < A.n_ind;
        bool ok = inb;
        if (MASKED) ok = inb && !miss_data(a0, a1, a2) && !miss_data(b0, b1, b2);
        x += (uint32_t)__popcll(__ballot(ok));
        const double c1 = inb ? fma(2.0, a2, a1) - mean1 : 0.0;
        const double c2 = inb ? fma(2.0, b2, b1) - mean2 : 0.0;
        sxy = fma(c1, c2, sxy);
      }
    }
    sxy = wave_sum1(sxy);
    if (lane == 0) {
      xch0[wave][0] = sxy;
      xch0[wave][1] = (double)x;
    }
    __syncthreads();
    sxy = 0.0;
    double xs = 0.0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      sxy += xch0[w][0];
      xs += xch0[w][1];
    }
    x = (uint32_t)xs;
    __syncthreads();

    // ---- haplo_freq (gen_func.cpp:1027-1059) ----
    double f0 = (1 - m1) * (1 - m2), f1 = (1 - m1) * m2, f2 = m1 * (1 - m2), f3 = m1 * m2;
    if (m1 < 0 || m1 > 1 || m2 < 0 || m2 > 1) {
      if (threadIdx.x == 0) atomicExch(A.status, (int)NGSLD_ERR_MAF_RANGE);
      f0 = f1 = f2 = f3 = __builtin_nan("");
    }
    const double inv_x = 1.0 / (double)x;
    bool bad = false, tie = false;
    uint32_t n_iter = 0;
    // The row vector's values arrive one chunk of slots ahead of the arithmetic: chunk k + 1 is requested before chunk k is
    // worked on -- and chunk 0 of the NEXT iteration (the same values: a does not change) before this iteration's sums meet,
    // so that its round trip to L2 runs beside the reduction and the exchange instead of in front of the next step.
    double av[2][kChunk][3];
    auto fetch = [&](int k) {
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        const int j = k * kChunk + u;
        if (j < SLOTS) {
          av[k & 1][u][0] = a_of(j, 0); av[k & 1][u][1] = a_of(j, 1); av[k & 1][u][2] = a_of(j, 2);
        }
      }
    };
    fetch(0);
    for (; n_iter < (uint32_t)kIterMax; ++n_iter) {
      const double p00 = f0 * f0, p01 = f0 * f1, p02 = f0 * f2, p03 = f0 * f3, p11 = f1 * f1;
      const double p12 = f1 * f2, p13 = f1 * f3, p22 = f2 * f2, p23 = f2 * f3, p33 = f3 * f3;
      const double w1 = p01 + p01, w3 = p02 + p02, w4 = 2.0 * (p03 + p12), w5 = p13 + p13, w7 = p23 + p23;
      double R0 = 0, R1 = 0, R2 = 0, R3 = 0, R4 = 0, R5 = 0, R6 = 0, R7 = 0, R8 = 0;
      // An individual that does not count -- padding, no data -- takes part with r = 0 instead of being branched around, as
      // in the streaming kernel.
      hide_offsets();
#pragma unroll
      for (int k = 0; k < kChunks; ++k) {
        if (k + 1 < kChunks) fetch(k + 1);
#pragma unroll
        for (int u = 0; u < kChunk; ++u) {
          const int j = k * kChunk + u;
          if (j < SLOTS) {
            const double a0 = av[k & 1][u][0], a1 = av[k & 1][u][1], a2 = av[k & 1][u][2];
            const double b0 = bv[j][0], b1 = bv[j][1], b2 = bv[j][2];
            const double v0 = fma(p11, b2, fma(w1, b1, p00 * b0));  // sum_g2 W[0][g2] b[g2]
            const double v1 = fma(w5, b2, fma(w4, b1, w3 * b0));
            const double v2 = fma(p33, b2, fma(w7, b1, p22 * b0));
            const double s = fma(a2, v2, fma(a1, v1, a0 * v0));
            // (every individual counts: the cohort ends inside the LAST slot -- block (n_ind - 1) / 64 is slot SLOTS - 1 of its
            // wavefront, the planes being padded to the next 64 -- or, TAIL, beyond the resident slots: no select before it)
            const double r = ((!MASKED && (TAIL || j < SLOTS - 1)) || ((vbits >> j) & 1u)) ? rcp_refined(s) : 0.0;
            const double r0 = r * a0, r1 = r * a1, r2 = r * a2;
            R0 = fma(r0, b0, R0); R1 = fma(r0, b1, R1); R2 = fma(r0, b2, R2);
            R3 = fma(r1, b0, R3); R4 = fma(r1, b1, R4); R5 = fma(r1, b2, R5);
            R6 = fma(r2, b0, R6); R7 = fma(r2, b1, R7); R8 = fma(r2, b2, R8);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (TAIL) {
        for (uint32_t bq = kTail0 + (uint32_t)wave; bq < n_blocks; bq += 2u * (uint32_t)kWaves) {
          double ta[2][3], tb[2][3];
          bool okv[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const uint32_t blk = bq + (uint32_t)(u * kWaves);
            const bool in = blk < n_blocks;
            const uint32_t i = (in ? blk : bq) * 64u + (uint32_t)lane;
            ta[u][0] = pa0[i]; ta[u][1] = pa1[i]; ta[u][2] = pa2[i];
            tb[u][0] = pb0[i]; tb[u][1] = pb1[i]; tb[u][2] = pb2[i];
            okv[u] = in && i < A.n_ind;
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const double a0 = ta[u][0], a1 = ta[u][1], a2 = ta[u][2], b0 = tb[u][0], b1 = tb[u][1], b2 = tb[u][2];
            bool ok = okv[u];
            if (MASKED) ok = ok && !miss_data(a0, a1, a2) && !miss_data(b0, b1, b2);
            const double v0 = fma(p11, b2, fma(w1, b1, p00 * b0));
            const double v1 = fma(w5, b2, fma(w4, b1, w3 * b0));
            const double v2 = fma(p33, b2, fma(w7, b1, p22 * b0));
            const double r = ok ? rcp_refined(fma(a2, v2, fma(a1, v1, a0 * v0))) : 0.0;
            const double r0 = r * a0, r1 = r * a1, r2 = r * a2;
            R0 = fma(r0, b0, R0); R1 = fma(r0, b1, R1); R2 = fma(r0, b2, R2);
            R3 = fma(r1, b0, R3); R4 = fma(r1, b1, R4); R5 = fma(r1, b2, R5);
            R6 = fma(r2, b0, R6); R7 = fma(r2, b1, R7); R8 = fma(r2, b2, R8);
          }
        }
      }
      fetch(0);  // for the next iteration (dropped if this one converges)
      __builtin_amdgcn_sched_barrier(0);
      double t0 = fma(p03, R4, fma(p02, R3, fma(p01, R1, p00 * R0)));
      double t1 = fma(p13, R5, fma(p12, R4, fma(p11, R2, p01 * R1)));
      double t2 = fma(p23, R7, fma(p22, R6, fma(p12, R4, p02 * R3)));
      double t3 = fma(p33, R8, fma(p23, R7, fma(p13, R5, p03 * R4)));
      wave_sum4(t0, t1, t2, t3);
      const int par = (int)(n_iter & 1u);
      if (lane == 0) {
        xch[par][wave][0] = t0; xch[par][wave][1] = t1; xch[par][wave][2] = t2; xch[par][wave][3] = t3;
      }
      lds_barrier();  // (LDS traffic only: the prefetch stays in flight)
      t0 = t1 = t2 = t3 = 0.0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {  // the same order in every wavefront: they leave the loop together
        t0 += xch[par][w][0]; t1 += xch[par][w][1]; t2 += xch[par][w][2]; t3 += xch[par][w][3];
      }
      const double n0 = t0 * inv_x, n1 = t1 * inv_x, n2 = t2 * inv_x, n3 = t3 * inv_x;
      const double sn = (n0 + n1) + (n2 + n3);
      if (__builtin_amdgcn_readfirstlane((int)!(sn < 2.0))) {  // the reference's all-NaN step (see em_pair)
        bad = true;
        break;
      }
      const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
      f0 = n0; f1 = n1; f2 = n2; f3 = n3;
      if (fabs(eps - kEpsilon) < kTieMargin) tie = true;
      if (__builtin_amdgcn_readfirstlane((int)(eps < kEpsilon))) break;
    }
    if (bad) f0 = f1 = f2 = f3 = __builtin_nan("");
    if (threadIdx.x == 0)
      write_pair(A, rec0 + (uint64_t)__popcll(it.mask & ((1ull << c) - 1ull)), f0, f1, f2, f3, sxy, rsx1, rsx2, x,
                 n_iter | (tie ? kTieBit : 0u));
  }
}

}  // namespace ngsld
