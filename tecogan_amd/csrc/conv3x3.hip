// 3x3 stride-1 SAME convolution, halo-tile / weights-stationary variant (gfx950 MFMA).
//
// This is the workhorse of the path: every generator_F res-block conv, FNet, VGG-19 and the D input
// conv of the reference (lib/frvsr.py:7-80, lib/ops.py:319-327, lib/Teco.py:48 via lib/ops.py:47-56)
// and -- with the taps flipped -- all of their input gradients.
//
// Structure (one 256-thread workgroup = 4 waves, one workgroup per CU, persistent over tiles):
//   * output tile = TH x 16 pixels x BN channels; the (TH+2) x 18 pixel input halo tile of one 128-byte
//     channel chunk is staged ONCE in LDS and all 9 taps read their A fragments from it: HBM/L2 input
//     traffic is (TH+2)*18/(TH*16) = 1.27x (TH=16) instead of 9x for a per-tap gather;
//   * the 9 x BN weight panel of the chunk is staged in LDS; when Cin fits one chunk it is loaded once per
//     workgroup and stays resident while the workgroup walks its tiles (weights-stationary);
//   * the next stage's global loads are issued into registers before the MFMA block of the current stage
//     (one wave per SIMD, ~250 VGPRs), so HBM latency overlaps the matrix work without a second LDS buffer;
//   * rows of 16 consecutive pixels with a 144-byte pitch make every ds_read_b128 fragment read
//     conflict-free; the epilogue (bias, activation, residual, act-grad mask) is the engine's.
// fp32 uses v_mfma_f32_16x16x4_f32 (exact, K-permuted as in conv_igemm.hip), bf16 v_mfma_f32_16x16x32_bf16.
#include "common.h"
#include <mutex>
#include <type_traits>
#include <stdlib.h>

int tg_conv3x3_ws_try(const tg_conv_desc* d, const void* in, const void* weight, const float* bias, const void* res,
                      const void* aux, void* out, hipStream_t st);        // conv3x3_ws.hip
int tg_conv3x3_dma_try(const tg_conv_desc* d, const void* in, const void* weight, const float* bias, const void* res,
                       const void* aux, void* out, hipStream_t st);       // conv3x3_dma.hip

struct Conv3P {
  const void* in;
  const void* w;      // [9][Cout][Cin]
  const float* bias;
  const void* res;
  const void* aux;
  void* out;
  int N, H, W, Cin, Cout;
  int flip;           // 1: taps mirrored (input-gradient form)
  int act;
  float act_alpha;
  float nslope;       // none: 1, ReLU: 0, LeakyReLU: alpha  -> act(v) = max(v, v*nslope)
  float mslope;       // act-grad mask: aux > 0 ? 1 : mslope (ReLU 0, LeakyReLU alpha, none 1)
  int tiles_y, tiles_x, ntiles;
  int direct_epi;     // per-lane stores instead of the LDS-staged rows
  int prio;           // s_setprio 3 for the small-tile (latency-bound chain) instantiations
  unsigned in_bytes, w_bytes;   // extents of `in` / `w` for the bounds-checked buffer loads (< 2^31)
};
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ABL: profiling-only ablation bits (1 = skip epilogue stores, 2 = skip the MFMA block, 4 = skip reloads of later
// stages, 8 = skip LDS staging writes of later stages); the product always launches ABL = 0.
// PACK > 1 (images no wider than 16/PACK pixels: VGG conv5 at 8x8, FNet's 8x8 / 4x4 levels): the 16 tile columns hold
// PACK images side by side -- LDS columns [0 | image 0 | 0 | image 1 | 0 ...], the zero columns are each image's left /
// right padding -- instead of one image and 8 (12) idle columns; the fragment address only gains a per-lane column
// offset (frow / WP).  conv5 ran at 135 TFLOP/s with half of every MFMA multiplying padding.
template <typename TIn, typename TOut, int TH, int BN, int ABL = 0, int PACK = 1>
__global__ __launch_bounds__(256, 1) void conv3x3_tile_kernel(Conv3P p) {
  constexpr int EPV = 16 / (int)sizeof(TIn);
  constexpr int BK = 8 * EPV;                       // channels per 128-byte chunk
  constexpr int ROWB = 144;
  constexpr int WP = 16 / PACK;                      // columns per packed image
  constexpr int HW = PACK > 1 ? 16 + PACK + 1 : 18;  // LDS halo columns
  constexpr int HALO_PIX = (TH + 2) * HW;
  constexpr int A_ITEMS = HALO_PIX * 8, A_LOADS = (A_ITEMS + 255) / 256;
  constexpr int B_ITEMS = 9 * BN * 8, B_LOADS = (B_ITEMS + 255) / 256;
  constexpr int WAVES_M = TH >= 4 ? 4 : TH, WAVES_N = 4 / WAVES_M;
  constexpr int TM = TH / WAVES_M, TN = (BN / 16) / WAVES_N;
  constexpr bool F32 = sizeof(TIn) == 4;
  static_assert(TM >= 1 && TN >= 1, "tile too small for 4 waves");
  // Small (latency-regime) bf16 tiles: MFMA operands swapped (A = weights, B = pixels), so a lane's four accumulator
  // rows are four CONSECUTIVE OUTPUT CHANNELS of one pixel and the epilogue is register-only with one 8-byte store per
  // accumulator -- no LDS staging, no epilogue barriers on the recurrent chain's critical path (conv3x3_ws.hip does
  // the same in the throughput regime).
  constexpr bool SWAP = !F32 && sizeof(TOut) == 2 && TH <= 4;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;                          // [HALO_PIX][144]
  unsigned char* Bs = smem + HALO_PIX * ROWB;        // [9*BN][144]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int frow = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.y * BN;
  const int nchunk = (p.Cin + BK - 1) / BK;
  // The recurrent chain's launches are latency bound; when throughput kernels of another stream share the CU
  // (engine.py overlap schedule) the chain's waves should win the issue arbitration.
  if constexpr (TH <= 4) {
    if (p.prio) __builtin_amdgcn_s_setprio(3);
  }
  const bool b_stationary = nchunk == 1;

  TOut* __restrict__ gout = static_cast<TOut*>(p.out);
  const TOut* __restrict__ gres = static_cast<const TOut*>(p.res);
  const TOut* __restrict__ gaux = static_cast<const TOut*>(p.aux);

  uint4 ra[A_LOADS], rb[B_LOADS];

  // ---- per-thread load descriptors, computed ONCE (tile-independent): the per-stage address is then
  //      scalar tile base + one v_add.  All offsets are 32-bit (tensors are < 2^31 elements).
  int relA[A_LOADS], dydx[A_LOADS];
#pragma unroll
  for (int k = 0; k < A_LOADS; ++k) {
    const int item = min(tid + k * 256, A_ITEMS - 1);
    const int pix = item >> 3, ch = item & 7;
    const int dy = pix / HW, dx = pix % HW;
    if constexpr (PACK == 1) {
      relA[k] = ((dy * p.W + dx) * p.Cin + ch * EPV) * (int)sizeof(TIn);      // bytes
      dydx[k] = dy | (dx << 8) | ((ch * EPV) << 16);
    } else {                  // dx field: local column + 1 (0 = separator / padding column); bits 28..: image within the group
      const int g = dx / (WP + 1), lc1 = dx % (WP + 1);
      relA[k] = (((g * p.H + dy) * p.W + lc1 - 1) * p.Cin + ch * EPV) * (int)sizeof(TIn);
      dydx[k] = dy | (lc1 << 8) | ((ch * EPV) << 16) | (g << 28);
    }
  }
  int relB[B_LOADS];          // < 0: row beyond Cout (stays zero)
#pragma unroll
  for (int k = 0; k < B_LOADS; ++k) {
    const int item = min(tid + k * 256, B_ITEMS - 1);
    const int row = item >> 3, ch = item & 7;
    const int tap = row / BN, co = n0 + row % BN;
    const int wtap = p.flip ? 8 - tap : tap;
    relB[k] = co < p.Cout ? ((wtap * p.Cout + co) * p.Cin + ch * EPV) * (int)sizeof(TIn) : -1;   // bytes
  }
  // Zero padding comes from the buffer descriptor's bounds check: a lane outside the image (or past Cin / Cout) is
  // given an out-of-range offset and the load returns 0.  The earlier form -- clamped address, then `if (!ok) v = 0`
  // -- put a select behind every load IN the issuing basic block, so each prefetch waited for its own data
  // (s_waitcnt vmcnt(17..0) right after the burst) and never overlapped the MFMA block it was issued ahead of.
  constexpr unsigned OOB = 0x80000000u;
  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  auto load_stage = [&](int tile, int chunk, bool with_b) {
    const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
    const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
    const int y0 = ty * TH - 1, x0 = PACK > 1 ? 0 : tx * 16 - 1;
    const int c0 = chunk * BK;
    const int base = (((n * PACK * p.H + y0) * p.W + x0) * p.Cin + c0) * (int)sizeof(TIn);   // wave-uniform, bytes
#pragma unroll
    for (int k = 0; k < A_LOADS; ++k) {
      // Loads are UNCONDITIONAL (a branch around each load makes hipcc wait vmcnt(0) per load).
      const int y = y0 + (dydx[k] & 255), c = c0 + ((dydx[k] >> 16) & 0xfff);
      bool ok = (unsigned)y < (unsigned)p.H && c < p.Cin;
      if constexpr (PACK == 1) {
        const int x = x0 + ((dydx[k] >> 8) & 255);
        ok = ok && (unsigned)x < (unsigned)p.W;
      } else {
        const int lc1 = (dydx[k] >> 8) & 255;
        ok = ok && lc1 != 0 && lc1 <= p.W && n * PACK + (dydx[k] >> 28) < p.N;
      }
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, (int)(ok ? (unsigned)(base + relA[k]) : OOB), 0, 0);
      ra[k] = make_uint4(v.x, v.y, v.z, v.w);
    }
    if (with_b) {
#pragma unroll
      for (int k = 0; k < B_LOADS; ++k) {
        const int c = c0 + (tid & 7) * EPV;
        const bool ok = relB[k] >= 0 && c < p.Cin;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
            rsrcB, (int)(ok ? (unsigned)(relB[k] + c0 * (int)sizeof(TIn)) : OOB), 0, 0);
        rb[k] = make_uint4(v.x, v.y, v.z, v.w);
      }
    }
  };
  auto store_stage = [&](bool with_b) {
#pragma unroll
    for (int k = 0; k < A_LOADS; ++k) {
      const int item = tid + k * 256;
      if (item < A_ITEMS) *reinterpret_cast<uint4*>(As + (item >> 3) * ROWB + (item & 7) * 16) = ra[k];
    }
    if (with_b) {
#pragma unroll
      for (int k = 0; k < B_LOADS; ++k) {
        const int item = tid + k * 256;
        if (item < B_ITEMS) *reinterpret_cast<uint4*>(Bs + (item >> 3) * ROWB + (item & 7) * 16) = rb[k];
      }
    }
  };

  // fragment bases: lane part + wave part once; every (tap, kk, tile) offset below is a compile-time constant
  const unsigned char* Afrag = As + (wm * TM * HW + frow + (PACK > 1 ? frow / WP : 0)) * ROWB + fg * 16;
  const unsigned char* Bfrag = Bs + (wn * TN * 16 + frow) * ROWB + fg * 16;
  const int col0 = n0 + wn * TN * 16 + frow;
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bv[j] = (p.bias && col0 + j * 16 < p.Cout) ? p.bias[col0 + j * 16] : 0.f;
  float bvs[SWAP ? TN : 1][4];                       // swapped layout: bias of channels chan0 + 16 j + 4 fg + r
  if constexpr (SWAP) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = n0 + wn * TN * 16 + j * 16 + fg * 4 + r;
        bvs[j][r] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
      }
  }

  f32x4 acc[TM][TN];

  int tile = blockIdx.x;
  if (tile >= p.ntiles) return;
  int chunk = 0;
  bool first = true;
  load_stage(tile, 0, true);
  while (tile < p.ntiles) {
    const bool with_b = first || !b_stationary;
    __syncthreads();                       // previous stage's fragment reads are done: LDS may be rewritten
    if (!(ABL & 8) || first) store_stage(with_b);
    __syncthreads();
    // prefetch the next stage into registers (overlaps the MFMA block below)
    int ntile = tile, nchunk_i = chunk + 1;
    if (nchunk_i == nchunk) {
      nchunk_i = 0;
      ntile = tile + gridDim.x;
    }
    if (ntile < p.ntiles && !(ABL & 4)) load_stage(ntile, nchunk_i, !b_stationary);

    if (chunk == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (!(ABL & 2))
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap % 3;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        uint4 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = *reinterpret_cast<const uint4*>(Afrag + ((i + kh) * HW + kw) * ROWB + kk * 64);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bfr[j] = *reinterpret_cast<const uint4*>(Bfrag + (tap * BN + j * 16) * ROWB + kk * 64);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (F32) {
              const float* a = reinterpret_cast<const float*>(&af[i]);
              const float* b = reinterpret_cast<const float*>(&bfr[j]);
#pragma unroll
              for (int e = 0; e < 4; ++e)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc[i][j], 0, 0, 0);
            } else if constexpr (SWAP) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&bfr[j]),
                                                                  *reinterpret_cast<bf16x8*>(&af[i]), acc[i][j],
                                                                  0, 0, 0);
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&af[i]),
                                                                  *reinterpret_cast<bf16x8*>(&bfr[j]), acc[i][j],
                                                                  0, 0, 0);
            }
          }
      }
    }

    if (chunk == nchunk - 1) {             // ---- fused epilogue for this tile ----
      const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
      const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
      const bool has_res = gres != nullptr, has_aux = gaux != nullptr;
      constexpr bool LDS_EPI = sizeof(TOut) == 2;      // bf16 outputs: stage through LDS, 16-byte global rows
      if constexpr (SWAP) {
        // accumulator r of lane (frow, fg) = pixel column frow, output channel chan0 + 16 j + 4 fg + r
        const int x = PACK > 1 ? frow % WP : tx * 16 + frow;
        const int nimg = PACK > 1 ? n * PACK + frow / WP : n;
        const bool vec4 = (p.Cout & 3) == 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int y = ty * TH + wm * TM + i;
          if (y >= p.H || x >= p.W || nimg >= p.N) continue;
          const int pix = ((nimg * p.H + y) * p.W + x) * p.Cout;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int co = n0 + wn * TN * 16 + j * 16 + fg * 4;
            if (co >= p.Cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = acc[i][j][r] + bvs[j][r];
              if (p.act >= TG_ACT_TANH) v[r] = act_fwd(v[r], p.act, p.act_alpha);
              else v[r] = fmaxf(v[r], v[r] * p.nslope);
            }
            if (vec4) {
              if (has_res) {
                const uint2 rr = *reinterpret_cast<const uint2*>(gres + pix + co);
                v[0] += __uint_as_float(rr.x << 16);
                v[1] += __uint_as_float(rr.x & 0xffff0000u);
                v[2] += __uint_as_float(rr.y << 16);
                v[3] += __uint_as_float(rr.y & 0xffff0000u);
              }
              if (has_aux) {
                const uint2 aa = *reinterpret_cast<const uint2*>(gaux + pix + co);
                v[0] *= __uint_as_float(aa.x << 16) > 0.f ? 1.f : p.mslope;
                v[1] *= __uint_as_float(aa.x & 0xffff0000u) > 0.f ? 1.f : p.mslope;
                v[2] *= __uint_as_float(aa.y << 16) > 0.f ? 1.f : p.mslope;
                v[3] *= __uint_as_float(aa.y & 0xffff0000u) > 0.f ? 1.f : p.mslope;
              }
              uint2 o;
              o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
              o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
              if (ABL & 1) { asm volatile("" ::"v"(o.x), "v"(o.y)); } else *reinterpret_cast<uint2*>(gout + pix + co) = o;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                if (co + r >= p.Cout) continue;
                float w = v[r];
                if (has_res) w += Elem<TOut>::ld(gres + pix + co + r);
                if (has_aux) w *= Elem<TOut>::ld(gaux + pix + co + r) > 0.f ? 1.f : p.mslope;
                Elem<TOut>::st(gout + pix + co + r, w);
              }
            }
          }
        }
      } else if (LDS_EPI && (p.Cout & 7) == 0 && !p.direct_epi) {
        // Phase 1: bias + activation in fp32 registers, bf16 tile into the (now idle) halo region of LDS.
        // Phase 2: every thread moves 16-byte rows: residual / mask operands arrive as vector loads and the
        // result leaves as one dwordx4 store per 8 channels (the per-lane 2-byte stores of the direct epilogue
        // cost 10 of 27 us at [1,270,480,64->64]: store-issue bound).
        __syncthreads();                   // all waves are done reading the halo tile
        u16* stage = reinterpret_cast<u16*>(As);
        constexpr int SP = 72;             // u16 per staged pixel row: 64 channels + 8 pad (144 B)
        // two uniformly selected copies: a per-element `slow ? tanh/sigmoid : max` left 191 scalar branches (and unpaired
        // bf16 conversions) in the 64-element fast path
        auto stage_tile = [&](auto slow_tag) {
          constexpr bool SLOW = decltype(slow_tag)::value;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int pl = ((wm * TM + i) * 16 + fg * 4 + r) * SP + wn * TN * 16 + frow;
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                float v = acc[i][j][r] + bv[j];
                if constexpr (SLOW) v = act_fwd(v, p.act, p.act_alpha);
                else v = fmaxf(v, v * p.nslope);
                stage[pl + j * 16] = f2bf(v);
              }
            }
        };
        if (p.act >= TG_ACT_TANH) stage_tile(std::true_type{});
        else stage_tile(std::false_type{});
        __syncthreads();
        constexpr int VPP = BN / 8;        // 16-byte vectors per pixel
        constexpr int NV = TH * 16 * VPP;
        const int y0 = ty * TH, x0 = tx * 16;
        for (int it = tid; it < NV; it += 256) {
          const int pl = it / VPP, cv = it % VPP;
          const int col = pl % 16;
          const int y = y0 + pl / 16, x = PACK > 1 ? col % WP : x0 + col, c = n0 + cv * 8;
          const int nimg = PACK > 1 ? n * PACK + col / WP : n;
          if (ABL & 1) continue;
          if (y >= p.H || x >= p.W || c >= p.Cout || nimg >= p.N) continue;
          uint4 o = *reinterpret_cast<const uint4*>(stage + pl * SP + cv * 8);
          const int idx = ((nimg * p.H + y) * p.W + x) * p.Cout + c;
          if (has_res || has_aux) {
            uint4 rr = make_uint4(0, 0, 0, 0), aa = rr;
            if (has_res) rr = *reinterpret_cast<const uint4*>(gres + idx);
            if (has_aux) aa = *reinterpret_cast<const uint4*>(gaux + idx);
            u16* ov = reinterpret_cast<u16*>(&o);
            const u16* rv = reinterpret_cast<const u16*>(&rr);
            const u16* av = reinterpret_cast<const u16*>(&aa);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float v = bf2f(ov[e]);
              if (has_res) v += bf2f(rv[e]);
              if (has_aux) v *= bf2f(av[e]) > 0.f ? 1.f : p.mslope;
              ov[e] = f2bf(v);
            }
          }
          *reinterpret_cast<uint4*>(gout + idx) = o;
        }
      } else {
        // direct epilogue (fp32 outputs, channel tails): 32-bit offsets hoisted per pixel; none/ReLU/LeakyReLU are
        // the branch-free max(v, v*slope); tanh/sigmoid and edge tiles take separate uniformly selected copies.
        const int ybase = ty * TH + wm * TM, xbase = tx * 16 + fg * 4;
        const int pix0 = ((n * p.H + ybase) * p.W + xbase) * p.Cout + col0;
        const bool interior = (ty + 1) * TH <= p.H && (tx + 1) * 16 <= p.W && n0 + BN <= p.Cout;
        auto epilogue = [&](auto check_tag, auto slow_tag) {
          constexpr bool CHECK = decltype(check_tag)::value, SLOW = decltype(slow_tag)::value;
#pragma unroll
          for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int off = pix0 + (i * p.W + r) * p.Cout;
              const bool pok = !CHECK || (ybase + i < p.H && xbase + r < p.W);
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                if (CHECK && !(pok && col0 + j * 16 < p.Cout)) continue;
                const int idx = off + j * 16;
                float v = acc[i][j][r] + bv[j];
                if constexpr (SLOW) v = act_fwd(v, p.act, p.act_alpha);
                else v = fmaxf(v, v * p.nslope);
                if (has_res) v += Elem<TOut>::ld(gres + idx);
                if (has_aux) v *= Elem<TOut>::ld(gaux + idx) > 0.f ? 1.f : p.mslope;
                if (ABL & 1) { asm volatile("" ::"v"(v)); } else Elem<TOut>::st(gout + idx, v);
              }
            }
          }
        };
        if (p.act >= TG_ACT_TANH) epilogue(std::true_type{}, std::true_type{});
        else if (interior) epilogue(std::false_type{}, std::false_type{});
        else epilogue(std::true_type{}, std::false_type{});
      }
    }
    first = false;
    tile = ntile;
    chunk = nchunk_i;
  }
}

template <typename TIn, typename TOut, int TH, int BN, int ABL = 0, int PACK = 1>
static void launch3(Conv3P p, hipStream_t st) {
  constexpr int LDS = ((TH + 2) * (PACK > 1 ? 16 + PACK + 1 : 18) + 9 * BN) * 144;
  auto kern = conv3x3_tile_kernel<TIn, TOut, TH, BN, ABL, PACK>;
  static std::once_flag attr_once;               // one-time, thread-safe: raise the dynamic-LDS limit of this instantiation
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  p.tiles_y = (p.H + TH - 1) / TH;
  p.tiles_x = PACK > 1 ? 1 : (p.W + 15) / 16;
  p.ntiles = ((p.N + PACK - 1) / PACK) * p.tiles_y * p.tiles_x;
  const int nt = (p.Cout + BN - 1) / BN;
  int gx = p.ntiles;
  const int per = tg_num_cus() / nt > 0 ? tg_num_cus() / nt : 1;        // one workgroup per CU (LDS-bound residency)
  if (gx > per) gx = per;
  static const char* const pname = [] {
    static char b[96];
    if constexpr (PACK > 1)
      snprintf(b, sizeof(b), "conv3x3_tile<%s,%s,%d,%d,pack%d>", sizeof(TIn) == 2 ? "bf16" : "f32",
               sizeof(TOut) == 2 ? "bf16" : "f32", TH, BN, PACK);
    else
      snprintf(b, sizeof(b), "conv3x3_tile<%s,%s,%d,%d>", sizeof(TIn) == 2 ? "bf16" : "f32", sizeof(TOut) == 2 ? "bf16" : "f32",
               TH, BN);
    return (const char*)b;
  }();
  const double px = (double)p.N * p.H * p.W;
  TG_LAUNCH(pname, 2.0 * px * p.Cout * 9.0 * p.Cin,
            px * (p.Cin * sizeof(TIn) + p.Cout * sizeof(TOut) * (1 + (p.res != nullptr) + (p.aux != nullptr))) +
                9.0 * p.Cin * p.Cout * sizeof(TIn),
            kern, dim3(gx, nt), dim3(256), LDS, st, p);
}

template <typename TIn, typename TOut>
static void launch3_typed(const Conv3P& p, hipStream_t st, bool coexist) {
  const int64_t pix = (int64_t)p.N * p.H * p.W;
  const int nt64 = (p.Cout + 63) / 64;
  if constexpr (sizeof(TIn) == 2 && sizeof(TOut) == 2) {
    // narrow images: several per tile row
    // ... once the layer is big enough to fill the chip anyway (pixels x channel tiles >= 16k: VGG conv5, the 72-pair
    // FNet of the TecoGAN step); below that the packed tiling only cuts the workgroup count of a latency-bound launch
    // (FRVSR step 3.96 -> 4.06 ms with packing everywhere)
    if (!p.direct_epi && p.W <= 8 && p.N >= 2 && (p.Cout & 7) == 0 && p.Cout >= 64 && pix * nt64 >= 16384) {
      if (p.W <= 4 && p.N >= 4) return launch3<TIn, TOut, 4, 64, 0, 4>(p, st);
      if (p.H <= 4) return launch3<TIn, TOut, 4, 64, 0, 2>(p, st);
      return launch3<TIn, TOut, 8, 64, 0, 2>(p, st);
    }
  }
  if (p.Cout <= 16) {
    if (!coexist && pix >= 16 * 16 * 256) launch3<TIn, TOut, 16, 16>(p, st);
    else if (pix >= 8 * 16 * 256) launch3<TIn, TOut, 8, 16>(p, st);
    else launch3<TIn, TOut, 4, 16>(p, st);
  } else if (p.Cout <= 32) {
    if (!coexist && pix >= 16 * 16 * 256) launch3<TIn, TOut, 16, 32>(p, st);
    else if (pix >= 8 * 16 * 256) launch3<TIn, TOut, 8, 32>(p, st);
    else launch3<TIn, TOut, 4, 32>(p, st);
  } else if (p.Cin * (int)sizeof(TIn) > 128) {
    // multi-chunk (Cin > one 128-B chunk): the weight panel is re-staged per (tile, chunk) -> largest pixel tile
    // TG_CONV_COEXIST: <8,64> instead of <16,64> -- 109 KB LDS / ~300 registers instead of 130 KB / ~400, which leaves room
    // on the CU for a co-resident workgroup of the latency-bound <4,16> chain kernel (36 KB / 120 registers)
    const int maxth = coexist ? 8 : 16;
    if (maxth >= 16 && pix * nt64 >= (int64_t)16 * 16 * 256) launch3<TIn, TOut, 16, 64>(p, st);
    else if (pix * nt64 >= (int64_t)8 * 16 * 256) launch3<TIn, TOut, 8, 64>(p, st);
    else if (pix * nt64 >= (int64_t)4 * 16 * 256) launch3<TIn, TOut, 4, 64>(p, st);
    else launch3<TIn, TOut, 2, 64>(p, st);
  } else {
    // single chunk, weights stationary: the largest tile that still gives every CU a workgroup.  Narrow
    // channel tiles (BN 32/16) shrink the weight panel each workgroup stages and were measured faster in situ
    // for the small recurrent-chain shapes (FRVSR step 6.10 -> 5.61 ms with <4,16>, same-session A/B in round 1).
    auto blocks = [&](int th, int bn) {
      return (int64_t)p.N * ((p.H + th - 1) / th) * ((p.W + 15) / 16) * ((p.Cout + bn - 1) / bn);
    };
    const int maxth = coexist ? 8 : 16;
    if (maxth >= 16 && blocks(16, 64) >= 256) launch3<TIn, TOut, 16, 64>(p, st);
    else if (maxth < 16 && blocks(8, 64) >= 256) launch3<TIn, TOut, 8, 64>(p, st);
    else if (blocks(16, 32) >= 256) launch3<TIn, TOut, 16, 32>(p, st);
    else if (blocks(8, 32) >= 256) launch3<TIn, TOut, 8, 32>(p, st);
    else if (blocks(8, 16) >= 256) launch3<TIn, TOut, 8, 16>(p, st);
    else if (blocks(4, 32) >= 256) launch3<TIn, TOut, 4, 32>(p, st);
    else launch3<TIn, TOut, 4, 16>(p, st);
  }
}

// ---------------------------------------------------------------------------------------------
// 8-channel inputs (bf16): the 3-channel tensors of the path after channel padding -- the input gradient of the
// generator's output conv (8 -> 64 at HR, once per frame), VGG conv1_1 and FNet's first conv.  The general kernel
// spends a full 64-wide K chunk per tap on them (7/8 zeros).  Here K packs the TAPS: one v_mfma_f32_16x16x32_bf16
// consumes 4 taps x 8 channels (lane group fg <-> tap 4*kk + fg), so 3 MFMAs cover the 9 taps (3 zero-weighted
// slots) instead of 18.  The weights are 3*NT register fragments per lane fetched once (no weight LDS at all); the
// halo tile is 16 B per pixel, so every A fragment is one conflict-free ds_read_b128.  The kernel is then what
// it should be: a streaming epilogue (HBM-bound on the output and mask tensors).
struct C8P {
  const void* in;
  const void* w;      // [9][Cout][8]
  const float* bias;
  const u16* res;
  const u16* aux;
  u16* out;
  int N, H, W, Cout, flip;
  float nslope, mslope;
  int tiles_y, tiles_x;
  unsigned in_bytes, w_bytes;
};

template <int NT>
__global__ __launch_bounds__(256) void conv3x3_c8_kernel(C8P p) {
  constexpr int TH = 4, TW = 32, HW = TW + 2, HALO = (TH + 2) * HW;
  constexpr int SP = NT * 16 + 8;                     // u16 per staged pixel row (+16 B pad)
  __shared__ __attribute__((aligned(16))) unsigned char Xs[HALO * 16];
  __shared__ __attribute__((aligned(16))) u16 stage[TH * TW * SP];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int frow = lane & 15, fg = lane >> 4;
  const int tx = blockIdx.x % p.tiles_x, t1 = blockIdx.x / p.tiles_x;
  const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = blockIdx.y * NT * 16;
  constexpr unsigned OOB = 0x80000000u;
  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  {  // halo tile: one 16-byte pixel per thread, zero outside the image (buffer bounds check)
    const int item = min(tid, HALO - 1);
    const int dy = item / HW, dx = item - dy * HW;
    const int y = y0 - 1 + dy, x = x0 - 1 + dx;
    const bool ok = tid < HALO && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, (int)(ok ? (unsigned)(((n * p.H + y) * p.W + x) * 16) : OOB),
                                                          0, 0);
    if (tid < HALO) *reinterpret_cast<u32x4*>(Xs + item * 16) = v;
  }
  // weight fragments: lane (frow = output channel, fg = tap slot) of K-step kk holds w[tap 4*kk+fg][cout][0..7]
  uint4 bfr[3][NT];
  float bv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = n0 + j * 16 + frow;
    bv[j] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      const int t = 4 * kk + fg;
      const int wt = p.flip ? 8 - t : t;
      const bool ok = t < 9 && co < p.Cout;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (int)(ok ? (unsigned)((wt * p.Cout + co) * 16) : OOB), 0, 0);
      bfr[kk][j] = make_uint4(v.x, v.y, v.z, v.w);
    }
  }
  // per-lane tap offsets inside the halo tile (slots >= 9 re-read tap 8: finite data times zero weights)
  int aoff[3];
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    const int t = min(4 * kk + fg, 8);
    const int kh = t / 3, kw = t - kh * 3;
    aoff[kk] = ((wave + kh) * HW + frow + kw) * 16;
  }
  __syncthreads();
#pragma unroll
  for (int sgm = 0; sgm < TW / 16; ++sgm) {
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      uint4 a = *reinterpret_cast<const uint4*>(Xs + aoff[kk] + sgm * 256);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&a),
                                                         *reinterpret_cast<bf16x8*>(&bfr[kk][j]), acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pl = (wave * TW + sgm * 16 + fg * 4 + r) * SP + frow;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float v = acc[j][r] + bv[j];
        stage[pl + j * 16] = f2bf(fmaxf(v, v * p.nslope));
      }
    }
  }
  __syncthreads();
  constexpr int VPP = NT * 2, NV = TH * TW * VPP;
  const bool has_res = p.res != nullptr, has_aux = p.aux != nullptr;
#pragma unroll
  for (int it0 = 0; it0 < NV; it0 += 256) {
    const int it = it0 + tid;
    const int pl = it / VPP, cv = it % VPP;
    const int y = y0 + pl / TW, x = x0 + pl % TW, c = n0 + cv * 8;
    if (y >= p.H || x >= p.W || c >= p.Cout) continue;
    uint4 o = *reinterpret_cast<const uint4*>(stage + pl * SP + cv * 8);
    const int idx = ((n * p.H + y) * p.W + x) * p.Cout + c;
    if (has_res || has_aux) {
      uint4 rr = make_uint4(0, 0, 0, 0), aa = rr;
      if (has_res) rr = *reinterpret_cast<const uint4*>(p.res + idx);
      if (has_aux) aa = *reinterpret_cast<const uint4*>(p.aux + idx);
      u16* ov = reinterpret_cast<u16*>(&o);
      const u16* rv = reinterpret_cast<const u16*>(&rr);
      const u16* av = reinterpret_cast<const u16*>(&aa);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = bf2f(ov[e]);
        if (has_res) v += bf2f(rv[e]);
        if (has_aux) v *= bf2f(av[e]) > 0.f ? 1.f : p.mslope;
        ov[e] = f2bf(v);
      }
    }
    *reinterpret_cast<uint4*>(p.out + idx) = o;
  }
}

static int conv3x3_c8_try(const tg_conv_desc* d, const void* in, const void* weight, const float* bias, const void* res,
                          const void* aux, void* out, hipStream_t st) {
  const bool enabled = true;
  if (!enabled || d->Cin != 8 || d->in_dtype != TG_BF16 || d->out_dtype != TG_BF16) return 0;
  if (d->act >= TG_ACT_TANH || (d->Cout != 32 && d->Cout % 64 != 0)) return 0;
  if ((((uintptr_t)out | (uintptr_t)res | (uintptr_t)aux) & 15)) return 0;
  // (this kernel's epilogue is the staged one only: a residual or LeakyReLU-mask operand would be added to the ALREADY ROUNDED
  //  tile -- those launches go to the tile kernel's fp32 register epilogue; none is on the timed path)
  if (res != nullptr || (aux != nullptr && d->mask_act == TG_ACT_LRELU)) return 0;
  C8P p;
  p.in = in; p.w = weight; p.bias = bias; p.res = (const u16*)res; p.aux = (const u16*)aux; p.out = (u16*)out;
  p.N = d->N; p.H = d->Hin; p.W = d->Win; p.Cout = d->Cout; p.flip = d->mode == 1;
  p.nslope = d->act == TG_ACT_RELU ? 0.f : (d->act == TG_ACT_LRELU ? d->act_alpha : 1.f);
  p.mslope = d->mask_act == TG_ACT_RELU ? 0.f : (d->mask_act == TG_ACT_LRELU ? d->mask_alpha : 1.f);
  p.tiles_y = (p.H + 3) / 4; p.tiles_x = (p.W + 31) / 32;
  p.in_bytes = (unsigned)((int64_t)d->N * d->Hin * d->Win * 16);
  p.w_bytes = (unsigned)((int64_t)9 * d->Cout * 16);
  const int64_t blocks = (int64_t)p.N * p.tiles_y * p.tiles_x;
  if (blocks >= ((int64_t)1 << 31)) return 0;
  const double px = (double)p.N * p.H * p.W;
  const double fl = 2.0 * px * p.Cout * 9.0 * 8.0;
  const double by = px * (16.0 + 2.0 * p.Cout * (1 + (res != nullptr) + (aux != nullptr))) + 9.0 * 16.0 * p.Cout;
  if (d->Cout == 32) TG_LAUNCH("conv3x3_c8<2>", fl, by, conv3x3_c8_kernel<2>, dim3((unsigned)blocks, 1), dim3(256), 0, st, p);
  else TG_LAUNCH("conv3x3_c8<4>", fl, by, conv3x3_c8_kernel<4>, dim3((unsigned)blocks, d->Cout / 64), dim3(256), 0, st, p);
  return 1;
}

// Returns 1 if the descriptor was handled by this kernel, 0 if the generic engine must take it.
int tg_conv3x3_try(const tg_conv_desc* d, const void* in, const void* weight, const float* bias, const void* res,
                   const void* aux, void* out, hipStream_t st) {
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad_t != 1 || d->pad_l != 1) return 0;
  if (d->Hin != d->Hout || d->Win != d->Wout) return 0;
  const int epv = d->in_dtype == TG_F32 ? 4 : 8;
  if (d->Cin % epv != 0 || (((uintptr_t)in | (uintptr_t)weight) & 15)) return 0;
  if (d->in_dtype == TG_F32 && d->out_dtype == TG_BF16) return 0;
  const int64_t esz = d->in_dtype == TG_F32 ? 4 : 2;
  const int64_t in_bytes = (int64_t)d->N * d->Hin * d->Win * d->Cin * esz, w_bytes = (int64_t)9 * d->Cout * d->Cin * esz;
  if (in_bytes >= ((int64_t)1 << 31) || w_bytes >= ((int64_t)1 << 31)) return 0;   // 32-bit buffer offsets
  if (aux && d->mask_act != TG_ACT_RELU && d->mask_act != TG_ACT_LRELU) return 0;   // generic engine handles others
  if (conv3x3_c8_try(d, in, weight, bias, res, aux, out, st)) return 1;
  if (tg_conv3x3_ws_try(d, in, weight, bias, res, aux, out, st)) return 1;       // one-chunk layers in the throughput regime
  if (tg_conv3x3_dma_try(d, in, weight, bias, res, aux, out, st)) return 1;      // wide layers (Cin > 64) in the throughput regime
  Conv3P p;
  p.in_bytes = (unsigned)in_bytes; p.w_bytes = (unsigned)w_bytes;
  p.in = in; p.w = weight; p.bias = bias; p.res = res; p.aux = aux; p.out = out;
  p.N = d->N; p.H = d->Hin; p.W = d->Win; p.Cin = d->Cin; p.Cout = d->Cout;
  p.flip = d->mode == 1;
  p.act = d->act; p.act_alpha = d->act_alpha;
  // The LDS-staged epilogue rounds the activated tile to bf16 BEFORE the row movers add the residual / apply a LeakyReLU mask
  // and round again: a double rounding, up to a whole bf16 step instead of half of one (what VERDICT r5 found as "2-4 steps off"
  // behind a tolerance relative to the tensor maximum).  Launches with such an operand take the register epilogue -- fp32 until
  // the one rounding at the store.  (A ReLU mask multiplies by 0 or 1: exact, stays on the staged path.)
  const int direct = (res != nullptr || (aux != nullptr && d->mask_act == TG_ACT_LRELU)) ? 1 : 0;
  p.direct_epi = direct;
  // default ON: with throughput kernels of the side stream on the same CU, the chain's waves at s_setprio 3 hide 75 % instead
  // of 48 % of a co-running VGG layer (tools/mb_forktax.py D: 5.14 vs 6.06 ms) and the TecoGAN step gains 2 %
  p.prio = 1;
  p.nslope = d->act == TG_ACT_RELU ? 0.f : (d->act == TG_ACT_LRELU ? d->act_alpha : 1.f);
  p.mslope = d->mask_act == TG_ACT_RELU ? 0.f : (d->mask_act == TG_ACT_LRELU ? d->mask_alpha : 1.f);
  const bool coexist = (d->flags & TG_CONV_COEXIST) != 0;
  if (d->in_dtype == TG_F32) launch3_typed<float, float>(p, st, coexist);
  else if (d->out_dtype == TG_BF16) launch3_typed<u16, u16>(p, st, coexist);
  else launch3_typed<u16, float>(p, st, coexist);
  return 1;
}
