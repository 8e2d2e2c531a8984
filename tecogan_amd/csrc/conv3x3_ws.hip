// 3x3 stride-1 SAME convolution, "weights in registers" variant for one-chunk layers (Cin <= 64, bf16) -- gfx950.
//
// Covers the throughput-regime 64->64 convolutions of the path: the generator's res-block convs at inference
// resolution (reference lib/frvsr.py:50-57 through main.py:204, 33 launches per 1080p frame), VGG-19 conv1_2
// (lib/ops.py:319), the discriminator's input conv (lib/Teco.py:48), FNet's 64-channel layers (lib/frvsr.py:9-31), and
// -- taps mirrored -- their input gradients.  conv3x3.hip keeps the latency regime (small tiles) and Cin > 64.
//
// Why a second kernel.  conv3x3_tile_kernel<16,64> holds the 9x64x64 weight panel in LDS (83 KB) next to one halo tile
// (47 KB): one workgroup per CU, one wave per SIMD, a single LDS buffer -- so halo staging (register prefetch ->
// ds_write_b128 -> barrier), the LDS-staged epilogue (2 more barriers) and the MFMA block run strictly one after the
// other on each SIMD (measured 6.4 us per 256-pixel tile for 1.9 us of MFMA work).  Here:
//   * the weights live in REGISTERS: a wave owns 32 output channels, 9 taps x 2 K-steps x 2 channel groups = 36
//     fragments = 144 VGPRs, loaded once per workgroup (persistent over tiles).  No weight LDS, no weight re-staging;
//   * the input halo tile ((8+2) x 18 pixels x 64 channels, 144-byte pixel pitch: the conflict-free layout of
//     conv3x3.hip) arrives by LDS-DMA (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write pass; lanes
//     outside the image get an out-of-range offset and the hardware writes zeros) into a DOUBLE buffer: the next
//     tile's DMA is issued right after the one barrier of a tile and flies during its MFMA block and epilogue;
//   * each halo fragment is read from LDS ONCE per (row, kw, K-step) and reused by the three vertical taps and both
//     channel groups: 36 ds_read_b128 per 144 MFMAs (the tile kernel: 144 per 288);
//   * the MFMA operands are swapped (A = weights, B = pixels), so a lane's four accumulator rows are four CONSECUTIVE
//     OUTPUT CHANNELS of one pixel: the epilogue is bias/activation/residual/mask in registers and one 8-byte store per
//     accumulator -- no LDS staging, no epilogue barriers;
//   * 52 KB of LDS and <= 256 registers per wave: TWO workgroups per CU (2 waves per SIMD), so one workgroup's
//     barrier, DMA wait and epilogue hide under the other's MFMA block.
#include "common.h"
#include <mutex>
#include <stdlib.h>

struct ConvWsP {
  const void* in;
  const void* w;      // [9][Cout][Cin]
  const float* bias;
  const void* res;
  const void* aux;
  void* out;
  int N, H, W, Cin, Cout;
  int flip;           // 1: taps mirrored (input-gradient form)
  float nslope;       // none: 1, ReLU: 0, LeakyReLU: alpha  -> act(v) = max(v, v*nslope)
  float mslope;       // act-grad mask: aux > 0 ? 1 : mslope
  int tiles_y, tiles_x, ntiles;
  unsigned in_bytes, w_bytes, out_bytes;
};

typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {
// Pixel pitch of the halo tile: WS_PS slots of 16 bytes (8 data + pad).  Under the gfx950 lane grouping of ds_read_b128
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) the model of tools/lds_layout_search.py calls a row of 16 pixels x 4 chunks at
// pitch 144 2-way conflicted and at pitch 160 conflict-free.  Built and measured (-DWS_PS=10 via tools/build_variant.py,
// profiles/r04o_ab.txt): no difference at any shape (16.3 / 96.8 us against 16.2 / 96.3 us) -- with 36 reads per 144 MFMAs
// this kernel is not bound by its fragment reads; the smaller tile stays.
#ifndef WS_PS
#define WS_PS 9
#endif
constexpr int WS_TH = 8, WS_HALO = (WS_TH + 2) * 18, WS_ROWB = WS_PS * 16;
constexpr int WS_SLOTS = WS_HALO * WS_PS;                  // 16-byte slots of one halo tile (8 data + pad per pixel)
constexpr int WS_NDMA = (WS_SLOTS + 63) / 64;              // wave-wide DMA instructions per tile (26)
constexpr int WS_BUF = WS_NDMA * 1024;                     // bytes per halo buffer (26624)
constexpr int WS_WROWB = 144;                              // weight panel rows (read once per workgroup): 8 data + 1 pad slot
constexpr int WS_KPW = (WS_NDMA + 3) / 4;                  // DMA instructions per wave (7)
constexpr int WS_WINST = (9 * 64 * 9 + 63) / 64;           // weight panel: 576 rows x 9 slots (8 data + 1 pad) = 81 instructions
constexpr int WS_WPANEL = WS_WINST * 1024;                 // 82944 bytes
constexpr int WS_WROUNDS = (WS_WINST + 3) / 4;             // 21 DMA rounds of 4 waves
constexpr unsigned WS_OOB = 0x80000000u;
}  // namespace

// Cycle stamps (tools/trace_ws.py builds a private -DTG_WS_TRACE copy of the library; the product build has none of it).
#ifdef TG_WS_TRACE
__device__ unsigned long long tg_ws_trace_buf[64];
#define WS_STAMP(i)                                                                                         \
  do {                                                                                                      \
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0 && (i) < 64) tg_ws_trace_buf[(i)] = (unsigned long long)clock64(); \
  } while (0)
extern "C" int tg_debug_ws_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_ws_trace_buf), sizeof(unsigned long long) * 64);
}
#else
#define WS_STAMP(i) do { } while (0)
#endif

// WLDS: the 9 x 64 x 64 weight panel of the workgroup's channel block comes in ONCE by LDS-DMA (73 KB + row padding) and
// the four waves read their 36 fragments from LDS, instead of every wave pulling its own 36 KB from L2 into registers
// through the texture path (144 KB per workgroup, 10k cycles of the prologue, tools/trace_ws.py) -- for launches with one
// workgroup per CU and a handful of tiles each (the 1080p inference convs: 4 tiles per CU), where the prologue is a
// quarter of the kernel.  83 KB more LDS: not with TG_CONV_COEXIST (no room left for a chain workgroup).
// WFRAG: p.w is the FRAGMENT-ORDER copy of a 64 -> 64 layer ([2 tap + kk][16-channel block][lane][8], tg_pack_weights_frag): each
// of a wave's 36 weight loads is one contiguous KiB = 8 whole cache lines instead of 64 half lines (DESIGN lesson 17: 43 against
// 19 B/clk/CU on the miss path), consumed in issue order by the peeled first tile.  The prologue shrinks enough (9.4k cycles of a
// 32k-cycle launch at the 1080p inference convs, profiles/r04p_trace_ws.txt) for TWO workgroups per CU to pay off at 4 tiles per CU:
// one workgroup's DMA issue and epilogue (1000 + 1500 of 5600 cycles per tile) run under the other's MFMA block.
// (Then issuing the next tile's 7 DMA instructions one by one between the MFMA groups, as conv3x3_dma.hip does, is neutral:
// 14.0 / 16.5 against 13.9 / 16.7 us, profiles/r04v_ab.txt -- the second workgroup already covers the issue gap.  Not kept.)
template <bool HAS_RES, bool HAS_AUX, bool WLDS, bool WFRAG = false>
__global__ __launch_bounds__(256, 2) void conv3x3_ws_kernel(ConvWsP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 2 x WS_BUF [+ WS_WPANEL]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;                  // 2 x 2 waves: 4 pixel rows x 32 channels each
  const int frow = lane & 15, fg = lane >> 4;
  const int cbase = blockIdx.y * 64 + wn * 32;
  const int row_bytes = p.Cin * 2;

  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  const auto rsrcO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
  const auto rsrcR = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_RES ? p.res : p.out), 0, (int)p.out_bytes, 0x00020000);
  const auto rsrcM = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_AUX ? p.aux : p.out), 0, (int)p.out_bytes, 0x00020000);

  u32x4w wf[9][2][2];
  // WFRAG: the weight stream starts before anything else (its addresses are lane * 16 + constants): the 12 fragments of the first
  // tap column go out ahead of the halo tile's DMA descriptors (2.5k cycles of address arithmetic and DMA issue in the trace),
  // the other 24 right after the DMA -- the first tile then waits for vmcnt(24): DMA and first column landed.
  auto load_frag = [&](const int kw_lo, const int kw_hi) {
#pragma unroll
    for (int kw = kw_lo; kw < kw_hi; ++kw)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int tap = kh * 3 + kw;
          const int wt = p.flip ? 8 - tap : tap;
#pragma unroll
          for (int j = 0; j < 2; ++j)
            wf[tap][kk][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, lane * 16, ((wt * 2 + kk) * 4 + wn * 2 + j) * 1024, 0);
        }
  };
  if constexpr (WFRAG) {
    if ((int)blockIdx.x >= p.ntiles) return;
    load_frag(0, 1);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- LDS-DMA slot descriptors of this lane (tile-independent): slot S = (wave + 4k)*64 + lane holds 16-byte chunk
  //      c = S % WS_PS of halo pixel S / WS_PS (chunks >= 8 = row padding; chunks past Cin and slots past the tile read zeros).
  int code[WS_KPW];
#pragma unroll
  for (int k = 0; k < WS_KPW; ++k) {
    const int S = (wave + 4 * k) * 64 + lane;
    const int pix = S / WS_PS, c = S - WS_PS * pix;
    const int dy = pix / 18, dx = pix - 18 * dy;
    const bool valid = S < WS_SLOTS && c < 8 && c * 8 < p.Cin;
    code[k] = dy | (dx << 8) | (c << 16) | (valid ? (1 << 24) : 0);
  }

  auto issue_dma = [&](int tile, int buf) {
    const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
    const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
    const int y0 = ty * WS_TH - 1, x0 = tx * 16 - 1;
    const int base = ((n * p.H + y0) * p.W + x0) * row_bytes;               // wave-uniform
#pragma unroll
    for (int k = 0; k < WS_KPW; ++k) {
      const int inst = wave + 4 * k;
      // only the last round is partial (26 instructions over 4 waves): the others issue unconditionally, branch-free
      if (4 * k + 3 < WS_NDMA || inst < WS_NDMA) {                          // wave-uniform
        const int dy = code[k] & 255, dx = (code[k] >> 8) & 255, c = (code[k] >> 16) & 255;
        const bool ok = (code[k] >> 24) && (unsigned)(y0 + dy) < (unsigned)p.H && (unsigned)(x0 + dx) < (unsigned)p.W;
        const unsigned off = ok ? (unsigned)(base + (dy * p.W + dx) * row_bytes + c * 16) : WS_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(smem + buf * WS_BUF + inst * 1024), 16, (int)off, 0, 0, 0);
      }
    }
  };

  int tile = blockIdx.x;
  if (tile >= p.ntiles) return;
  WS_STAMP(0);
  issue_dma(tile, 0);
  WS_STAMP(1);

  // ---- weights -> registers: lane (frow, fg) of fragment (tap, kk, j) holds w[tap][cbase + 16 j + frow][32 kk + 8 fg .. +7].
  //      Issued in the order the MFMA loop consumes them (kw, kk, kh, j): the loop below starts on the first fragments
  //      while the rest are still streaming in (the compiler places the counted vmcnt waits), instead of draining all 36
  //      loads first -- at kernel start every wave of the chip pulls its 36 KB through the texture path at once and the
  //      full drain cost 13k of the 23k prologue cycles (tools/trace_ws.py).
  if constexpr (WFRAG) {
    __builtin_amdgcn_sched_barrier(0);
    load_frag(1, 3);
  } else if constexpr (WLDS) {
    // panel row R = tap * 64 + channel (of this block), 144-byte pitch as the halo; slot S = instruction * 64 + lane
    unsigned char* wp = smem + 2 * WS_BUF;
#pragma unroll
    for (int k = 0; k < WS_WROUNDS; ++k) {
      const int inst = wave + 4 * k;
      if (k + 1 < WS_WROUNDS || inst < WS_WINST) {                          // wave-uniform
        const int S = inst * 64 + lane;
        const int R = S / 9, c = S - 9 * R;
        const int tap = R >> 6, co = blockIdx.y * 64 + (R & 63);
        const int wt = p.flip ? 8 - tap : tap;
        const bool ok = R < 576 && c < 8 && c * 8 < p.Cin && co < p.Cout;
        const unsigned off = ok ? (unsigned)(((wt * p.Cout + co) * p.Cin + c * 8) * 2) : WS_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_void*)(wp + inst * 1024), 16, (int)off, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's halo and panel slots
    __builtin_amdgcn_s_barrier();                              // everybody's
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            wf[kh * 3 + kw][kk][j] = *reinterpret_cast<const u32x4w*>(
                wp + ((kh * 3 + kw) * 64 + wn * 32 + j * 16 + frow) * WS_WROWB + kk * 64 + fg * 16);
  } else {
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int tap = kh * 3 + kw;
        const int wt = p.flip ? 8 - tap : tap;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int co = cbase + j * 16 + frow, ci = kk * 32 + fg * 8;
          const bool ok = co < p.Cout && ci < p.Cin;
          wf[tap][kk][j] = __builtin_amdgcn_raw_buffer_load_b128(
              rsrcW, (int)(ok ? (unsigned)(((wt * p.Cout + co) * p.Cin + ci) * 2) : WS_OOB), 0, 0);
        }
      }
  }
  float bv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cbase + j * 16 + fg * 4 + r;
      bv[j][r] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
    }

  // Loop-carried synchronisation.  Per tile and wave the VMEM queue holds, in issue order: the next tile's DMA slots,
  // [the residual / mask loads, consumed by the epilogue], the 8 epilogue stores.  Before reading the next tile only the
  // DMA has to have landed, so the wait at the bottom of the loop is vmcnt(8): everything but the 8 youngest operations
  // (vmcnt retires in issue order on gfx9-family parts) -- draining the stores too (vmcnt(0), or the vmcnt(0) that
  // __syncthreads() adds while a DMA is in flight) exposed a full store round trip per tile.  Hence also the raw s_barrier.
  WS_STAMP(2);                                              // weight loads issued
  if constexpr (WFRAG) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // first tile: DMA slots + first tap column (all but the 24 later loads)
  else if constexpr (!WLDS) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");   // first tile: this wave's DMA slots (all but the 36 weight loads)
  WS_STAMP(3);
  int buf = 0;
  [[maybe_unused]] int it = 0;
  // one tile: barrier, next tile's DMA, MFMA block, epilogue.  A lambda so that the FIRST tile is a peeled copy: there the
  // compiler can wait for the weight fragments one by one (counted vmcnt) as the MFMA loop reaches them; inside the loop
  // it would have to drain them all at the loop header.
  auto process = [&](const int tile, const int ntile) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // every wave's slots landed; nobody still reads the other buffer
    WS_STAMP(4 + 5 * it);
    if (ntile < p.ntiles) issue_dma(ntile, buf ^ 1);        // flies during the MFMA block and the epilogue below
    __builtin_amdgcn_sched_barrier(0);
    WS_STAMP(5 + 5 * it);

    const unsigned char* Afrag = smem + buf * WS_BUF + ((wm * 4) * 18 + frow) * WS_ROWB + fg * 16;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        u32x4w af[6];                                       // halo rows 0..5 of this wave at column offset kw
#pragma unroll
        for (int r = 0; r < 6; ++r) af[r] = *reinterpret_cast<const u32x4w*>(Afrag + (r * 18 + kw) * WS_ROWB + kk * 64);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[kh * 3 + kw][kk][j]),
                                                                  __builtin_bit_cast(bf16x8, af[i + kh]), acc[i][j], 0, 0, 0);
      }
    }

    WS_STAMP(6 + 5 * it);                                   // MFMA block issued (the last results may still be in the pipe)
    // ---- epilogue in registers: accumulator r of lane (frow, fg) = pixel column frow, output channel cbase+16j+4fg+r
    {
      const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
      const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
      const int x = tx * 16 + frow;
      unsigned offs[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int y = ty * WS_TH + wm * 4 + i;
        const bool pok = y < p.H && x < p.W;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int co = cbase + j * 16 + fg * 4;
          offs[i][j] = (pok && co < p.Cout) ? (unsigned)((((n * p.H + y) * p.W + x) * p.Cout + co) * 2) : WS_OOB;
        }
      }
      // all residual / mask vectors of the tile in flight at once (8 + 8 loads), then the arithmetic: one exposed round
      // trip per tile instead of one per accumulator (the residual form ran 23.3 us against 17.6 us without)
      u32x2w rr[HAS_RES ? 4 : 1][2], aa[HAS_AUX ? 4 : 1][2];
      if constexpr (HAS_RES) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) rr[i][j] = __builtin_amdgcn_raw_buffer_load_b64(rsrcR, (int)offs[i][j], 0, 0);
      }
      if constexpr (HAS_AUX) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) aa[i][j] = __builtin_amdgcn_raw_buffer_load_b64(rsrcM, (int)offs[i][j], 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[i][j][r] + bv[j][r];
            v[r] = fmaxf(v[r], v[r] * p.nslope);
          }
          if constexpr (HAS_RES) {
            v[0] += __uint_as_float(rr[i][j].x << 16);
            v[1] += __uint_as_float(rr[i][j].x & 0xffff0000u);
            v[2] += __uint_as_float(rr[i][j].y << 16);
            v[3] += __uint_as_float(rr[i][j].y & 0xffff0000u);
          }
          if constexpr (HAS_AUX) {
            v[0] *= __uint_as_float(aa[i][j].x << 16) > 0.f ? 1.f : p.mslope;
            v[1] *= __uint_as_float(aa[i][j].x & 0xffff0000u) > 0.f ? 1.f : p.mslope;
            v[2] *= __uint_as_float(aa[i][j].y << 16) > 0.f ? 1.f : p.mslope;
            v[3] *= __uint_as_float(aa[i][j].y & 0xffff0000u) > 0.f ? 1.f : p.mslope;
          }
          u32x2w o;
          o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
          o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
          __builtin_amdgcn_raw_buffer_store_b64(o, rsrcO, (int)offs[i][j], 0, 0);
        }
      }
    }
    WS_STAMP(7 + 5 * it);                                   // epilogue stores issued
  };
  process(tile, tile + gridDim.x);
  tile += gridDim.x;
  while (tile < p.ntiles) {
    buf ^= 1;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // the DMA of the tile we turn to has landed (stores may still fly)
    WS_STAMP(8 + 5 * it);
    ++it;
    process(tile, tile + gridDim.x);
    tile += gridDim.x;
  }
#ifdef TG_WS_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WS_STAMP(63);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Transposed 3x3 stride-2 SAME convolution (slim.conv2d_transpose, reference lib/ops.py:35-44 as used by the generator's
// conv_tran2highres stage, lib/frvsr.py:73-78) in the same weights-in-registers / LDS-DMA scheme, for the throughput
// regime (the 1080p inference step: [1,270,480,64] -> [1,540,960,64] -> [1,1080,1920,64], 0.26 ms of a 1.44 ms frame on
// the generic engine).  TF alignment ([TF1] SURVEY A.2): y[2i+ky, 2j+kx] += x[i,j] . w[ky,kx], so the four output phases
// of input pixel (i,j) are small stride-1 convolutions over x with NO zero MACs:
//     out(2i  ,2j  ) = x[i,j] w00 + x[i,j-1] w02 + x[i-1,j] w20 + x[i-1,j-1] w22        (4 taps)
//     out(2i  ,2j+1) = x[i,j] w01 + x[i-1,j] w21                                         (2 taps)
//     out(2i+1,2j  ) = x[i,j] w10 + x[i,j-1] w12                                         (2 taps)
//     out(2i+1,2j+1) = x[i,j] w11                                                        (1 tap)
// A tile is 8 x 16 INPUT pixels (+ one halo row / column up-left; the conv's (8+2) x 18 halo tile is reused) and
// 16 x 32 output pixels; per wave 144 MFMAs per tile as in the conv, four register epilogues (bias + ReLU, 8-byte stores).
__global__ __launch_bounds__(256, 2) void deconv3x3s2_ws_kernel(ConvWsP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 2 x WS_BUF
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fg = lane >> 4;
  const int cbase = blockIdx.y * 64 + wn * 32;
  const int row_bytes = p.Cin * 2;
  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  const auto rsrcO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);

  int code[WS_KPW];
#pragma unroll
  for (int k = 0; k < WS_KPW; ++k) {
    const int S = (wave + 4 * k) * 64 + lane;
    const int pix = S / WS_PS, c = S - WS_PS * pix;
    const int dy = pix / 18, dx = pix - 18 * dy;
    const bool valid = S < WS_SLOTS && c < 8 && c * 8 < p.Cin;
    code[k] = dy | (dx << 8) | (c << 16) | (valid ? (1 << 24) : 0);
  }
  auto issue_dma = [&](int tile, int buf) {
    const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
    const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
    const int y0 = ty * WS_TH - 1, x0 = tx * 16 - 1;
    const int base = ((n * p.H + y0) * p.W + x0) * row_bytes;
#pragma unroll
    for (int k = 0; k < WS_KPW; ++k) {
      const int inst = wave + 4 * k;
      if (4 * k + 3 < WS_NDMA || inst < WS_NDMA) {
        const int dy = code[k] & 255, dx = (code[k] >> 8) & 255, c = (code[k] >> 16) & 255;
        const bool ok = (code[k] >> 24) && (unsigned)(y0 + dy) < (unsigned)p.H && (unsigned)(x0 + dx) < (unsigned)p.W;
        const unsigned off = ok ? (unsigned)(base + (dy * p.W + dx) * row_bytes + c * 16) : WS_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(smem + buf * WS_BUF + inst * 1024), 16, (int)off, 0, 0, 0);
      }
    }
  };

  int tile = blockIdx.x;
  if (tile >= p.ntiles) return;
  issue_dma(tile, 0);
  u32x4w wf[9][2][2];                                        // [ky*3+kx][K-step][channel group], TF layout [kh,kw,Cout,Cin]
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int co = cbase + j * 16 + frow, ci = kk * 32 + fg * 8;
        const bool ok = co < p.Cout && ci < p.Cin;
        wf[tap][kk][j] = __builtin_amdgcn_raw_buffer_load_b128(
            rsrcW, (int)(ok ? (unsigned)(((tap * p.Cout + co) * p.Cin + ci) * 2) : WS_OOB), 0, 0);
      }
  float bv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cbase + j * 16 + fg * 4 + r;
      bv[j][r] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int buf = 0;
  const int Ho = 2 * p.H, Wo = 2 * p.W;
  while (true) {
    const int ntile = tile + gridDim.x;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ntile < p.ntiles) issue_dma(ntile, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    // halo pixel of input (row i, column frow) of this wave: row wm*4 + i + 1, column frow + 1; "up" / "left" = -1
    const unsigned char* Afrag = smem + buf * WS_BUF + ((wm * 4) * 18 + frow) * WS_ROWB + fg * 16;
    const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
    const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
    const int xin = tx * 16 + frow;
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int dyi = 0; dyi < (py ? 1 : 2); ++dyi)          // phase row 0 also sees the input row above (ky = 2)
#pragma unroll
            for (int dxi = 0; dxi < (px ? 1 : 2); ++dxi) {      // phase column 0 also sees the input column to the left (kx = 2)
              const int ky = py ? 1 : 2 * dyi, kx = px ? 1 : 2 * dxi;
              u32x4w af[4];
#pragma unroll
              for (int i = 0; i < 4; ++i)
                af[i] = *reinterpret_cast<const u32x4w*>(Afrag + ((i + 1 - dyi) * 18 + 1 - dxi) * WS_ROWB + kk * 64);
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ky * 3 + kx][kk][j]),
                                                                      __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0);
            }
        const int xo = 2 * xin + px;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int yin = ty * WS_TH + wm * 4 + i;
          const int yo = 2 * yin + py;
          const bool pok = yin < p.H && xin < p.W;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int co = cbase + j * 16 + fg * 4;
            const unsigned off = (pok && co < p.Cout) ? (unsigned)((((n * Ho + yo) * Wo + xo) * p.Cout + co) * 2) : WS_OOB;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = acc[i][j][r] + bv[j][r];
              v[r] = fmaxf(v[r], v[r] * p.nslope);
            }
            u32x2w o;
            o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
            o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
            __builtin_amdgcn_raw_buffer_store_b64(o, rsrcO, (int)off, 0, 0);
          }
        }
      }
    tile = ntile;
    if (tile >= p.ntiles) break;
    buf ^= 1;
    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");       // the next tile's DMA has landed; the 32 stores may still fly
  }
}

// Returns 1 if handled (bf16, Cin <= 64, Cout % 64 == 0, k3 s2 pad 0 transposed, >= 256 input tiles), else 0.
int tg_deconv3x3s2_ws_try(const tg_conv_desc* d, const void* in, const void* weight, const float* bias, const void* res,
                          const void* aux, void* out, hipStream_t st) {
  const bool enabled = true;
  if (!enabled || d->mode != 1 || d->KH != 3 || d->KW != 3 || d->stride != 2 || d->pad_t != 0 || d->pad_l != 0) return 0;
  if (d->Hout != 2 * d->Hin || d->Wout != 2 * d->Win || res || aux) return 0;
  if (d->in_dtype != TG_BF16 || d->out_dtype != TG_BF16 || d->act >= TG_ACT_TANH) return 0;
  if (d->Cin % 8 != 0 || d->Cin > 64 || d->Cin < 16 || d->Cout % 64 != 0) return 0;
  if ((((uintptr_t)in | (uintptr_t)weight | (uintptr_t)out) & 15)) return 0;
  const int64_t px = (int64_t)d->N * d->Hin * d->Win;
  const int64_t in_bytes = px * d->Cin * 2, out_bytes = 4 * px * d->Cout * 2, w_bytes = (int64_t)9 * d->Cout * d->Cin * 2;
  if (in_bytes >= ((int64_t)1 << 31) || out_bytes >= ((int64_t)1 << 31)) return 0;
  ConvWsP p;
  p.in = in; p.w = weight; p.bias = bias; p.res = nullptr; p.aux = nullptr; p.out = out;
  p.N = d->N; p.H = d->Hin; p.W = d->Win; p.Cin = d->Cin; p.Cout = d->Cout;
  p.flip = 0;
  p.nslope = d->act == TG_ACT_RELU ? 0.f : (d->act == TG_ACT_LRELU ? d->act_alpha : 1.f);
  p.mslope = 1.f;
  p.tiles_y = (p.H + WS_TH - 1) / WS_TH;
  p.tiles_x = (p.W + 15) / 16;
  const int64_t ntiles = (int64_t)p.N * p.tiles_y * p.tiles_x;
  // selection threshold in input tiles: 256 = one tile per CU.  The training chain's two transposed convs (32 and 128 tiles)
  // stay on conv_igemm: 5.3 / 9.2 us per launch there against 10.8 / 11.8 us here, TecoGAN step 10.96 -> 11.23 ms with them on
  // this kernel (round 4, profiles/r04a_ab.txt: a weights-in-registers prologue is not for 32 workgroups)
  constexpr int min_tiles = 256;
  if (ntiles < min_tiles || ntiles >= ((int64_t)1 << 30)) return 0;
  p.ntiles = (int)ntiles;
  p.in_bytes = (unsigned)in_bytes; p.w_bytes = (unsigned)w_bytes; p.out_bytes = (unsigned)out_bytes;
  constexpr int LDS = 2 * WS_BUF;
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(deconv3x3s2_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  const int nt = p.Cout / 64;
  int gx = p.ntiles;
  const int cap = 2 * tg_num_cus() / nt > 0 ? 2 * tg_num_cus() / nt : 1;
  if (gx > cap) gx = cap;
  TG_LAUNCH("deconv3x3s2_ws", 2.0 * (double)px * 9.0 * p.Cin * p.Cout, (double)px * (p.Cin * 2.0 + 4.0 * p.Cout * 2.0) + 18.0 * p.Cin * p.Cout,
            deconv3x3s2_ws_kernel, dim3(gx, nt), dim3(256), LDS, st, p);
  return 1;
}

template <bool HAS_RES, bool HAS_AUX>
static void launch_ws_wlds(const ConvWsP& p, hipStream_t st) {
  auto kern = conv3x3_ws_kernel<HAS_RES, HAS_AUX, true>;
  constexpr int LDS = 2 * WS_BUF + WS_WPANEL;
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  const int nt = p.Cout / 64;
  int gx = p.ntiles;
  const int cap = tg_num_cus() / nt > 0 ? tg_num_cus() / nt : 1;
  if (gx > cap) gx = cap;
  static const char* const pname = HAS_RES ? (HAS_AUX ? "conv3x3_ws<res,aux>" : "conv3x3_ws<res>")
                                           : (HAS_AUX ? "conv3x3_ws<aux>" : "conv3x3_ws<>");
  const double px = (double)p.N * p.H * p.W;
  TG_LAUNCH(pname, 2.0 * px * p.Cout * 9.0 * p.Cin,
            px * (p.Cin * 2.0 + p.Cout * 2.0 * (1 + HAS_RES + HAS_AUX)) + 18.0 * p.Cin * p.Cout, kern, dim3(gx, nt), dim3(256),
            LDS, st, p);
}

template <bool HAS_RES, bool HAS_AUX>
static void launch_ws(const ConvWsP& p, hipStream_t st, bool coexist) {
  auto kern = conv3x3_ws_kernel<HAS_RES, HAS_AUX, false>;
  constexpr int LDS_MAX = 2 * WS_BUF + 32768;
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX);
  });
  // TG_CONV_COEXIST: ONE workgroup per CU (32 KB of unused LDS push the request past half the CU) -- two of them would
  // take 480 of the SIMD's 512 registers and lock the latency-bound chain kernel out of the CU
  // two workgroups per CU pay the weight prologue twice per CU: worth it from ~4 tiles per workgroup on
  // (measured at 1020 tiles: 16.4 us with one, 17.8 us with two; at 9728 tiles: 97 vs 95 us)
  const int per_cu = coexist ? 1 : (p.ntiles >= 2048 ? 2 : 1);
  // ... from two tiles per workgroup on (the 1080p inference convs: 4); with a single tile per workgroup (FNet's 64-channel
  // layers in the training steps) it measured neutral to slightly slower (12.20 -> 12.26 ms, profiles/r02zzzz_ab.txt)
  if (per_cu == 1 && !coexist && p.ntiles >= 2 * (tg_num_cus() / (p.Cout / 64) > 0 ? tg_num_cus() / (p.Cout / 64) : 1)) {
    launch_ws_wlds<HAS_RES, HAS_AUX>(p, st);
    return;
  }
  const int LDS = per_cu == 1 ? LDS_MAX : 2 * WS_BUF;
  const int nt = p.Cout / 64;
  int gx = p.ntiles;
  const int cap = tg_num_cus() * per_cu / nt > 0 ? tg_num_cus() * per_cu / nt : 1;
  if (gx > cap) gx = cap;
  static const char* const pname = HAS_RES ? (HAS_AUX ? "conv3x3_ws<res,aux>" : "conv3x3_ws<res>")
                                           : (HAS_AUX ? "conv3x3_ws<aux>" : "conv3x3_ws<>");
  const double px = (double)p.N * p.H * p.W;
  TG_LAUNCH(pname, 2.0 * px * p.Cout * 9.0 * p.Cin,
            px * (p.Cin * 2.0 + p.Cout * 2.0 * (1 + HAS_RES + HAS_AUX)) + 18.0 * p.Cin * p.Cout, kern, dim3(gx, nt), dim3(256),
            LDS, st, p);
}

// Returns 1 if the descriptor was handled here, 0 otherwise (the halo-tile kernel of conv3x3.hip takes it).
int tg_conv3x3_ws_try(const tg_conv_desc* d, const void* in, const void* weight, const float* bias, const void* res,
                      const void* aux, void* out, hipStream_t st) {
  const int min_tiles = 256;
  if (d->in_dtype != TG_BF16 || d->out_dtype != TG_BF16) return 0;
  if (d->Cin % 8 != 0 || d->Cin > 64 || d->Cin < 16 || d->Cout % 64 != 0) return 0;
  if (d->act >= TG_ACT_TANH) return 0;
  if ((((uintptr_t)in | (uintptr_t)weight | (uintptr_t)out | (uintptr_t)res | (uintptr_t)aux) & 15)) return 0;
  const int64_t px = (int64_t)d->N * d->Hin * d->Win;
  const int64_t in_bytes = px * d->Cin * 2, out_bytes = px * d->Cout * 2, w_bytes = (int64_t)9 * d->Cout * d->Cin * 2;
  if (in_bytes >= ((int64_t)1 << 31) || out_bytes >= ((int64_t)1 << 31)) return 0;
  ConvWsP p;
  p.in = in; p.w = weight; p.bias = bias; p.res = res; p.aux = aux; p.out = out;
  p.N = d->N; p.H = d->Hin; p.W = d->Win; p.Cin = d->Cin; p.Cout = d->Cout;
  p.flip = d->mode == 1;
  p.nslope = d->act == TG_ACT_RELU ? 0.f : (d->act == TG_ACT_LRELU ? d->act_alpha : 1.f);
  p.mslope = d->mask_act == TG_ACT_RELU ? 0.f : (d->mask_act == TG_ACT_LRELU ? d->mask_alpha : 1.f);
  p.tiles_y = (p.H + WS_TH - 1) / WS_TH;
  p.tiles_x = (p.W + 15) / 16;
  const int64_t ntiles = (int64_t)p.N * p.tiles_y * p.tiles_x;
  if (ntiles < min_tiles || ntiles >= ((int64_t)1 << 30)) return 0;
  p.ntiles = (int)ntiles;
  p.in_bytes = (unsigned)in_bytes; p.w_bytes = (unsigned)w_bytes; p.out_bytes = (unsigned)out_bytes;
  const bool coexist = (d->flags & TG_CONV_COEXIST) != 0;
  if (res && aux) launch_ws<true, true>(p, st, coexist);
  else if (res) launch_ws<true, false>(p, st, coexist);
  else if (aux) launch_ws<false, true>(p, st, coexist);
  else launch_ws<false, false>(p, st, coexist);
  return 1;
}

// The throughput-regime 64 -> 64 conv with the weights handed over in fragment order (the generator's res-block convs at inference
// resolution, reference lib/frvsr.py:50-57 through main.py:204): out = act(conv3x3(x, W) + b) [+ res].  TG_EINVAL below 256 tiles
// of 8x16 pixels (the latency regime: tg_resblock / tg_conv_forward).
extern "C" int tg_conv3x3_c64_frag(const void* x, const void* w_frag, const float* bias, const void* res, void* out, int N, int H,
                                   int W, int act, float act_alpha, void* stream) {
  TG_CHECK_ARG(x && w_frag && out && N > 0 && H > 0 && W > 0, "bad argument");
  TG_CHECK_ARG((((uintptr_t)x | (uintptr_t)w_frag | (uintptr_t)out | (uintptr_t)res) & 15) == 0, "alignment");
  TG_CHECK_ARG(act == TG_ACT_NONE || act == TG_ACT_RELU || act == TG_ACT_LRELU, "activation");
  const int64_t px = (int64_t)N * H * W;
  TG_CHECK_ARG(px * 128 < ((int64_t)1 << 31), "tensor too large for 32-bit buffer offsets");
  ConvWsP p;
  p.in = x; p.w = w_frag; p.bias = bias; p.res = res; p.aux = nullptr; p.out = out;
  p.N = N; p.H = H; p.W = W; p.Cin = 64; p.Cout = 64;
  p.flip = 0;
  p.nslope = act == TG_ACT_RELU ? 0.f : (act == TG_ACT_LRELU ? act_alpha : 1.f);
  p.mslope = 1.f;
  p.tiles_y = (H + WS_TH - 1) / WS_TH;
  p.tiles_x = (W + 15) / 16;
  const int64_t ntiles = (int64_t)N * p.tiles_y * p.tiles_x;
  TG_CHECK_ARG(ntiles >= 256 && ntiles < ((int64_t)1 << 30), "fewer than 256 tiles: latency regime");
  p.ntiles = (int)ntiles;
  p.in_bytes = (unsigned)(px * 128); p.w_bytes = 9 * 64 * 64 * 2; p.out_bytes = (unsigned)(px * 128);
  hipStream_t st = static_cast<hipStream_t>(stream);
  constexpr int per_cu = 2;      // two workgroups per CU: 14.0 against 14.5 us with one (profiles/r04q_ab.txt)
  int gx = p.ntiles;
  if (gx > tg_num_cus() * per_cu) gx = tg_num_cus() * per_cu;
  const int LDS = per_cu == 1 ? 2 * WS_BUF + 32768 : 2 * WS_BUF;
  const double fl = 2.0 * px * 64 * 9.0 * 64, by = px * 128.0 * (2 + (res != nullptr)) + 73728.0;
  if (res) {
    auto kern = conv3x3_ws_kernel<true, false, false, true>;
    static std::once_flag once;
    std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WS_BUF + 32768); });
    TG_LAUNCH("conv3x3_ws<res,frag>", fl, by, kern, dim3(gx, 1), dim3(256), LDS, st, p);
  } else {
    auto kern = conv3x3_ws_kernel<false, false, false, true>;
    static std::once_flag once;
    std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WS_BUF + 32768); });
    TG_LAUNCH("conv3x3_ws<frag>", fl, by, kern, dim3(gx, 1), dim3(256), LDS, st, p);
  }
  TG_CHECK_LAUNCH();
}
