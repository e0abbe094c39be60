// EXPERIMENT (round 6, verdict item 4) -- not part of the library, not on any product path.
// A lane per pair for configs[1]'s shape (all pairs, n_ind = 100), the STREAMED form of tools/r06_c1_lane_model.py:
//   * one workgroup of 256 lanes (four wavefronts, one per SIMD) per CU, persistent;
//   * 32 row sites resident in LDS, the partner sites passing through a ring of 32 LDS slots (2,408 B a slot: an odd number of
//     doubles, so that the 32 slots start in 32 different bank pairs and a ds_read_b64 gather over any mix of slots is
//     conflict-free; equal slots broadcast);
//   * a lane takes the next pair of the stream (an LDS counter) when its own converges; wavefront 0 reloads a ring slot when
//     the 32 pairs of the partner it held are done;
//   * the EM step of ld_em.h's em_step in its four-value form (s = a . (W b), one refined reciprocal per individual,
//     R += a (b r)), serial over the individuals of the lane's pair: no cross-lane reduction, no lockstep tail.
// RESULT (README.md, profiles/r06/c1_lanes_experiment.txt): every record equal to the library's, 31.0 ms a pass against the lockstep
// kernel's 20.2 ms; the mean workgroup 21.7 ms.  Not adopted.
// Out: hap[4], n_iter and (D, D', r2) per pair, at the pair's all-pairs record index.  Flags, r2_ExpG, missing data: not here
// (the comparison with the lockstep kernel is generous to this form by that much).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int kRows = 32, kRing = 32, kMaxSeg = 64;
#ifndef C1_THREADS
#define C1_THREADS 256
#endif
constexpr int kThreads = C1_THREADS;  // lanes per workgroup: 256 = one wavefront per SIMD, 512 = two
constexpr int kInd = 100;             // the experiment is compiled for configs[1]'s cohort
constexpr int kMaxInd = kInd;
constexpr int kStride = 3 * kInd + 1;  // doubles per LDS slot: odd
constexpr double kEps = 1e-5;
constexpr uint32_t kIterMax = 100;

struct Segment {
  uint32_t row0, k0, k1, pad;  // partners (sites) [k0, k1) against the rows row0 .. row0 + 31
};

struct Args {
  const double *gl;   // [site][3][n_ind], normal space, normalised
  const double *maf;  // [site]
  uint32_t n_sites, n_ind;
  const Segment *segs;     // [workgroup][kMaxSeg]
  const uint32_t *n_segs;  // [workgroup]
  double *out_f;           // [pair][4]
  double *out_ld;          // [pair][3]
  uint32_t *out_iter;      // [pair]
  unsigned long long *dbg; // [workgroup][wave][4]: cycles in the individuals' loop, cycles in all, steps, lane-steps (or null)
};

__device__ __forceinline__ double rcp_refined(double s) {
  const double r0 = __builtin_amdgcn_rcp(s);
  const double e = fma(-s, r0, 1.0);
  const double t = fma(e, e, e);
  return fma(r0, t, r0);
}

__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p; }

// A single ds_read_b64: 64-bank mode, two LDS cycles a wavefront, conflict-free over the slots (the compiler's own choice for two
// neighbouring doubles, ds_read2_b64, runs in 32-bank mode over groups of 16 lanes: 32 slots on 16 bank pairs, 34.9 ms a pass).
template <int OFF>
__device__ __forceinline__ double lds_rd(uint32_t addr) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// individual J of a block of four: v[6 J + g] = the row site's, v[6 J + 3 + g] = the partner's
template <int J>
__device__ __forceinline__ void load_ind(double (&v)[24], uint32_t aa, uint32_t bb) {
  v[6 * J + 0] = lds_rd<0 * kInd * 8 + J * 8>(aa);
  v[6 * J + 1] = lds_rd<1 * kInd * 8 + J * 8>(aa);
  v[6 * J + 2] = lds_rd<2 * kInd * 8 + J * 8>(aa);
  v[6 * J + 3] = lds_rd<0 * kInd * 8 + J * 8>(bb);
  v[6 * J + 4] = lds_rd<1 * kInd * 8 + J * 8>(bb);
  v[6 * J + 5] = lds_rd<2 * kInd * 8 + J * 8>(bb);
}
__device__ __forceinline__ void load4(double (&v)[24], uint32_t aa, uint32_t bb) {
  load_ind<0>(v, aa, bb); load_ind<1>(v, aa, bb); load_ind<2>(v, aa, bb); load_ind<3>(v, aa, bb);
}
__device__ __forceinline__ void wait4(double (&v)[24]) {  // the reads above have landed; nothing that uses v moves before this
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int k = 0; k < 24; ++k) asm volatile("" : "+v"(v[k]));
}

enum : uint32_t { IDLE = 0, PENDING = 1, RUN = 2, DRY = 3 };

__global__ __launch_bounds__(kThreads) void c1_lanes_kernel(Args A) {
  __shared__ double S[(kRows + kRing) * kStride];
  __shared__ uint32_t ctl[64];  // 0: next pair of the stream, 1: partners ready, 2..33: pairs done per ring slot
  volatile uint32_t *vctl = ctl;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  constexpr uint32_t n_ind = kInd, site_len = 3 * n_ind;
  const double inv_x = 1.0 / (double)n_ind;
  const uint32_t n_seg = A.n_segs[blockIdx.x];
  for (uint32_t sj = 0; sj < n_seg; ++sj) {
    const Segment sg = A.segs[blockIdx.x * kMaxSeg + sj];
    const uint32_t nk = sg.k1 - sg.k0, q_total = nk * kRows;
    __syncthreads();  // the segment before is drained: every wavefront left its loop
    for (uint32_t idx = tid; idx < kRows * site_len; idx += kThreads) {
      const uint32_t r = idx / site_len, o = idx - r * site_len, site = sg.row0 + r;
      S[r * kStride + o] = site < A.n_sites ? A.gl[(uint64_t)site * site_len + o] : 0.0;
    }
    if (tid < (uint32_t)kRows) S[tid * kStride + site_len] = sg.row0 + tid < A.n_sites ? A.maf[sg.row0 + tid] : 0.0;  // (the slot's odd double)
    const uint32_t pre = nk < (uint32_t)kRing ? nk : (uint32_t)kRing;
    for (uint32_t idx = tid; idx < pre * site_len; idx += kThreads) {
      const uint32_t p = idx / site_len, o = idx - p * site_len;
      S[(kRows + p) * kStride + o] = A.gl[(uint64_t)(sg.k0 + p) * site_len + o];
    }
    if (tid < pre) S[(kRows + tid) * kStride + site_len] = A.maf[sg.k0 + tid];
    if (tid < 64) ctl[tid] = tid == 1 ? pre : 0u;
    __syncthreads();
    uint32_t k_loaded = pre;  // (wavefront 0's)
    unsigned long long t_loop = 0, n_steps = 0, n_lane_steps = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    uint32_t state = IDLE, q = 0, iter = 0, a_off = 0, b_off = 0, slot = 0;
    uint64_t rec = 0;
    double f0 = 0, f1 = 0, f2 = 0, f3 = 0;
    for (;;) {
      // ---- wavefront 0: a ring slot whose partner is done takes the next partner of the stream ----
      if (wave == 0) {
        while (k_loaded < nk) {
          const uint32_t sl = k_loaded & (kRing - 1);
          if (vctl[2 + sl] != (uint32_t)kRows) break;
          const double *src = A.gl + (uint64_t)(sg.k0 + k_loaded) * site_len;
          for (uint32_t o = lane; o < site_len; o += 64) S[(kRows + sl) * kStride + o] = src[o];
          if (lane == 0) S[(kRows + sl) * kStride + site_len] = A.maf[sg.k0 + k_loaded];
          __builtin_amdgcn_s_waitcnt(0);  // vmcnt / lgkmcnt: the slot is written
          ++k_loaded;
          if (lane == 0) {
            vctl[2 + sl] = 0;
            vctl[1] = k_loaded;
          }
          __builtin_amdgcn_s_waitcnt(0);
        }
      }
      // ---- lanes without a pair take the next of the stream ----
      for (;;) {
        if (state == IDLE) {
          q = atomicAdd(&ctl[0], 1u);
          state = q >= q_total ? DRY : PENDING;
        }
        const uint32_t ready = vctl[1];
        if (state == PENDING && (q >> 5) < ready) {
          const uint32_t k = q >> 5, r = q & 31u;
          const uint32_t s1 = sg.row0 + r, s2 = sg.k0 + k;
          slot = k & (kRing - 1);
          if (s1 < s2 && s1 < A.n_sites) {
            const double m1 = S[r * kStride + site_len], m2 = S[(kRows + slot) * kStride + site_len];
            f0 = (1 - m1) * (1 - m2); f1 = (1 - m1) * m2; f2 = m1 * (1 - m2); f3 = m1 * m2;
            iter = 0;
            a_off = r * kStride;
            b_off = (kRows + slot) * kStride;
            rec = (uint64_t)s1 * (2ull * A.n_sites - s1 - 1) / 2 + (s2 - s1 - 1);
            state = RUN;
          } else {
            atomicAdd(&ctl[2 + slot], 1u);
            state = IDLE;
          }
        }
        if (!__any(state == IDLE)) break;
      }
      if (!__any(state != DRY)) {
        if (wave != 0 || k_loaded >= nk) break;
        __builtin_amdgcn_s_sleep(8);
        continue;
      }
      if (!__any(state == RUN)) {
        __builtin_amdgcn_s_sleep(4);
        continue;
      }
      // ---- one EM step for the lanes that hold a pair ----
      if (state == RUN) {
        const double p00 = f0 * f0, p01 = f0 * f1, p02 = f0 * f2, p03 = f0 * f3, p11 = f1 * f1;
        const double p12 = f1 * f2, p13 = f1 * f3, p22 = f2 * f2, p23 = f2 * f3, p33 = f3 * f3;
        const double w1 = p01 + p01, w3 = p02 + p02, w4 = 2.0 * (p03 + p12), w5 = p13 + p13, w7 = p23 + p23;
        double R0 = 0, R1 = 0, R2 = 0, R3 = 0, R4 = 0, R5 = 0, R6 = 0, R7 = 0, R8 = 0;
        auto compute4 = [&](const double (&v)[24]) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const double a0 = v[6 * j + 0], a1 = v[6 * j + 1], a2 = v[6 * j + 2];
            const double b0 = v[6 * j + 3], b1 = v[6 * j + 4], b2 = v[6 * j + 5];
            const double c0 = fma(p11, b2, fma(w1, b1, p00 * b0));
            const double c1 = fma(w5, b2, fma(w4, b1, w3 * b0));
            const double c2 = fma(p33, b2, fma(w7, b1, p22 * b0));
            const double s = fma(a2, c2, fma(a1, c1, a0 * c0));
            const double r = rcp_refined(s);
            const double u0 = b0 * r, u1 = b1 * r, u2 = b2 * r;
            R0 = fma(a0, u0, R0); R1 = fma(a0, u1, R1); R2 = fma(a0, u2, R2);
            R3 = fma(a1, u0, R3); R4 = fma(a1, u1, R4); R5 = fma(a1, u2, R5);
            R6 = fma(a2, u0, R6); R7 = fma(a2, u1, R7); R8 = fma(a2, u2, R8);
          }
        };
        const unsigned long long t_in = __builtin_amdgcn_s_memtime();
        static_assert(kInd % 8 == 4, "25 blocks of four individuals: twelve trips of two blocks and one more");
        uint32_t aa = lds_addr(S + a_off), bb = lds_addr(S + b_off);
        double u[24], w[24];
        load4(u, aa, bb);
        for (int trip = 0; trip < kInd / 8; ++trip) {
          wait4(u);
          load4(w, aa + 32, bb + 32);
          __builtin_amdgcn_sched_barrier(0);  // (the block in hand is worked on while the next one's reads are in flight --
          compute4(u);                        //  left alone, the compiler merges the two blocks and waits right behind the reads)
          __builtin_amdgcn_sched_barrier(0);
          wait4(w);
          aa += 64; bb += 64;
          load4(u, aa, bb);
          __builtin_amdgcn_sched_barrier(0);
          compute4(w);
          __builtin_amdgcn_sched_barrier(0);
        }
        wait4(u);
        compute4(u);
        t_loop += __builtin_amdgcn_s_memtime() - t_in;
        n_steps += 1;
        n_lane_steps += (unsigned long long)__popcll(__ballot(true));
        const double t0 = fma(p03, R4, fma(p02, R3, fma(p01, R1, p00 * R0)));
        const double t1 = fma(p13, R5, fma(p12, R4, fma(p11, R2, p01 * R1)));
        const double t2 = fma(p23, R7, fma(p22, R6, fma(p12, R4, p02 * R3)));
        const double t3 = fma(p33, R8, fma(p23, R7, fma(p13, R5, p03 * R4)));
        const double n0 = t0 * inv_x, n1 = t1 * inv_x, n2 = t2 * inv_x, n3 = t3 * inv_x;
        bool finished;
        if (!(n1 < 2.0)) {
          f0 = f1 = f2 = f3 = __builtin_nan("");
          finished = true;
        } else {
          const double eps = fmax(fmax(fabs(n0 - f0), fabs(n1 - f1)), fmax(fabs(n2 - f2), fabs(n3 - f3)));
          f0 = n0; f1 = n1; f2 = n2; f3 = n3;
          finished = eps < kEps;
          if (!finished && ++iter == kIterMax) finished = true;
        }
        if (finished) {
          const double hm0 = 1 - (f0 + f1), hm1 = 1 - (f0 + f2);
          const double D = f0 * f3 - f1 * f2;
          const double den = D < 0 ? -fmin(hm0 * hm1, (1 - hm0) * (1 - hm1)) : fmin(hm0 * (1 - hm1), (1 - hm0) * hm1);
          const double rr = D / __dsqrt_rn(hm0 * hm1 * (1 - hm0) * (1 - hm1));
          double *of = A.out_f + rec * 4;
          of[0] = f0; of[1] = f1; of[2] = f2; of[3] = f3;
          double *ol = A.out_ld + rec * 3;
          ol[0] = D; ol[1] = D / den; ol[2] = rr * rr;
          A.out_iter[rec] = iter;
          atomicAdd(&ctl[2 + slot], 1u);
          state = IDLE;
        }
      }
    }
    if (A.dbg != nullptr && lane == 0) {
      unsigned long long *d = A.dbg + ((uint64_t)blockIdx.x * (kThreads / 64) + wave) * 4;
      d[0] += t_loop; d[1] += __builtin_amdgcn_s_memtime() - t_begin; d[2] += n_steps; d[3] += n_lane_steps;
    }
  }
}

}  // namespace

extern "C" int c1_lanes_max_seg() { return kMaxSeg; }

extern "C" int c1_lanes_run(const double *gl, const double *maf, uint32_t n_sites, uint32_t n_ind, const void *segs, const uint32_t *n_segs,
                            uint32_t n_workgroups, double *out_f, double *out_ld, uint32_t *out_iter, void *stream, unsigned long long *dbg) {
  if (n_ind != (uint32_t)kInd) return -1;
  Args a{gl, maf, n_sites, n_ind, (const Segment *)segs, n_segs, out_f, out_ld, out_iter, dbg};
  hipLaunchKernelGGL(c1_lanes_kernel, dim3(n_workgroups), dim3(kThreads), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
