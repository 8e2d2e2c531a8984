// Neighbour hand-off between the workgroups of ONE persistent launch (csrc/resblock_chain.hip, csrc/resblock_plane.hip): tagged
// 16-byte granules {v01, tag, v23, tag} -- the data is the flag (cdna_hip_programming.md section 6, Guideline 16, form R2) --
// published with plain stores (ring P: readers on the writer's XCD, whose L2 is their coherence point) or write-through stores
// (ring S: readers on another XCD) and swept with L1-bypassing loads until every tag equals the expected epoch.
#pragma once
#include "common.h"
#include <type_traits>

typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2c __attribute__((ext_vector_type(2)));

namespace {
constexpr unsigned RC_OOB = 0x80000000u;
constexpr int RC_SC1 = 16;                      // buffer aux: sc1 (agent scope: write-through store / L1-bypassing load)
}  // namespace

template <int I, int N, typename F>
__device__ __forceinline__ void rc_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    rc_static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ u32x2c rc_pack4(const float (&v)[4]) {
  u32x2c o;
  o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
  o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
  return o;
}
__device__ __forceinline__ void rc_unpack4(const u32x2c& a, float (&f)[4]) {
  f[0] = __uint_as_float(a.x << 16);
  f[1] = __uint_as_float(a.x & 0xffff0000u);
  f[2] = __uint_as_float(a.y << 16);
  f[3] = __uint_as_float(a.y & 0xffff0000u);
}

// Sweep of NI 16-byte granule pairs per lane until every tag equals `tag` (wave-uniform result), then the 8 payload bytes of each
// go to LDS.  goff: byte offset inside a ring slot (RC_OOB: position outside the image, nothing to wait for, the LDS position
// keeps the zeros of the first staging), lpos: LDS byte address of the 8 payload bytes.
// SM (sweep mode): 3 = two polls in flight; 0 / 1 / 2 = one poll at a time, the first one 0 / 128 / 256 cycles after the publish
template <int NI, int SM = 0>
__device__ __forceinline__ bool rc_sweep(const __amdgpu_buffer_rsrc_t& rsG, const unsigned (&goff)[NI], const int (&lpos)[NI],
                                         unsigned char* xs, unsigned soff, unsigned tag, unsigned limit,
                                         unsigned long long* stat = nullptr) {
  // One poll at a time (round trip ~560 cycles; a neighbour's granules become visible ~300 after its stores).  Two polls in
  // flight half a round trip apart (SM = 3) and a delayed first poll (SM = 1, 2) were measured: 3.39 / 3.32 / 3.31 against 3.30 us
  // per block (profiles/r06t_trace_chain.txt) -- what a sweep waits for is the LAST of eight neighbours, not the poll phase.
  // While it waits a wave runs at priority 0 and backs off after 32 round trips: co-resident work of other kernels -- whose progress is what frees a
  // compute unit for a workgroup of THIS launch that is not resident yet -- is not starved by the pollers (session F: a variant
  // small enough to stack five workgroups per compute unit beside a GEMM gave up for exactly that reason).
  u32x4c ga[NI], gb[NI];
  limit = __builtin_amdgcn_readfirstlane(limit);
#ifdef TG_RC_TRACE
  const unsigned long long t0 = clock64();
#endif
  auto issue = [&](u32x4c (&g)[NI]) {
    asm volatile("" ::: "memory");                 // (a poll is re-issued: the loads may not be hoisted or merged)
#pragma unroll
    for (int k = 0; k < NI; ++k) g[k] = __builtin_amdgcn_raw_buffer_load_b128(rsG, (int)goff[k], (int)soff, RC_SC1);
  };
  auto complete = [&](const u32x4c (&g)[NI]) {
    unsigned bad = 0;
#pragma unroll
    for (int k = 0; k < NI; ++k) bad |= (goff[k] != RC_OOB ? 0xffffffffu : 0u) & ((g[k].y ^ tag) | (g[k].w ^ tag));
    return !__any(bad != 0);                       // wave-uniform
  };
  __builtin_amdgcn_s_setprio(0);
  bool ok = false, useb = false;
  unsigned spins = 0;
  if constexpr (SM == 3) {
    issue(ga);
    __builtin_amdgcn_s_sleep(4);
    for (; spins <= limit; ++spins) {
      issue(gb);
      if (complete(ga)) { ok = true; break; }
      issue(ga);
      if (complete(gb)) { ok = true; useb = true; break; }
      if (spins > 32) __builtin_amdgcn_s_sleep(32);
    }
  } else {
    if constexpr (SM == 1) __builtin_amdgcn_s_sleep(2);
    if constexpr (SM == 2) __builtin_amdgcn_s_sleep(4);
    for (; spins <= limit; ++spins) {
      issue(ga);
      if (complete(ga)) { ok = true; break; }
      if (spins > 32) __builtin_amdgcn_s_sleep(32);
      else __builtin_amdgcn_s_sleep(1);
    }
  }
  if (!ok) return false;
  auto deliver = [&](const u32x4c (&g)[NI]) {
#pragma unroll
    for (int k = 0; k < NI; ++k)
      if (goff[k] != RC_OOB) *reinterpret_cast<u32x2c*>(xs + lpos[k]) = u32x2c{g[k].x, g[k].z};
  };
  if (useb) deliver(gb);                           // (two code paths: a select between the register sets became a scratch array)
  else deliver(ga);
#ifdef TG_RC_TRACE
  if (stat) *stat = ((unsigned long long)(clock64() - t0) << 32) | (2 * spins + 1 + (useb ? 1 : 0));      // cycles | polls checked
#endif
  (void)stat;
  return true;
}

