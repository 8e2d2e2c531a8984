// 3x3 stride-1 SAME convolution for the wide layers (Cin a multiple of 32, > 64; bf16) -- gfx950.
//
// Covers where most of the training step's MACs are: VGG-19 conv2_2 ... conv4_4 (reference lib/ops.py:319-327 through
// lib/Teco.py:5-24: 85 % of the perceptual-loss MACs, which are 86 % of the TecoGAN step's),
// FNet's 128 / 256-channel levels (lib/frvsr.py:13-27) and -- taps mirrored -- their input gradients.
//
// Why a third kernel.  conv3x3_tile_kernel<8|16,64> (conv3x3.hip) stages every 64-channel chunk of the input halo AND of
// the 9 x 64 weight panel global -> registers -> ds_write_b128 -> barrier into a SINGLE LDS buffer (130 KB at TH = 16), so
// per chunk the load round trip, 1250 cycles of ds_write traffic, two barriers and the 2300-cycle MFMA block run one
// after the other: 500 TFLOP/s, 20 % of the MFMA peak (profiles/r02g_bench.json).  conv3x3_ws.hip fixed this for
// Cin <= 64 by keeping the weights in registers; with Cin up to 512 they do not fit.  Here:
//   * a pipeline STAGE is one 32-channel chunk (= one K step of v_mfma_f32_16x16x32_bf16) of one 16 x 16 pixel tile:
//     the 18 x 18 halo pixels (20.3 KB) and the 9 x 64 x 32 weight panel (36 KB) of the stage arrive by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write pass; out-of-image lanes get an out-of-range
//     offset and the hardware writes zeros) into a DOUBLE buffer (2 x 57 KB): the DMA of stage s+1 is issued right
//     after the one barrier of stage s and flies during its 72 MFMAs per wave;
//   * rows are 64 bytes (32 channels) with NO padding; the four 16-byte slots of a row are XOR-swizzled by
//     2 * (row bit 2), which makes every ds_read_b128 fragment read of 16 consecutive rows conflict-free under the
//     gfx950 lane grouping (MI355X_MICROARCH.md LDS table) -- the swizzle is applied on the GLOBAL side of the DMA (each
//     lane picks which chunk it fetches), since the LDS side is fixed at base + 16 * lane;
//   * 8 waves (4 x 2): a wave owns 4 pixel rows x 32 output channels; per (kw) it reads 6 halo-row fragments once and
//     reuses them for the three vertical taps: 12 ds_read_b128 per 24 MFMAs (the tile kernel: 6 per 8), two waves per
//     SIMD so one wave's LDS latency and address work hide under the other's MFMAs;
//   * MFMA operands swapped (A = weights, B = pixels) as in conv3x3_ws.hip: register-only epilogue, 8-byte stores.
#include "common.h"
#include <mutex>
#include <stdlib.h>

struct ConvDmaP {
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

typedef unsigned int u32x4d __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_d;

namespace {
constexpr int DM_TH = 16;
// PK = images per tile side.  PK = 1: one 16 x 16 tile of a larger image, 18 x 18 halo.  PK = 2 / 4 (PACKED tiles, images of
// exactly 8 x 8 / 4 x 4 pixels: VGG conv5, FNet's inner levels): the 16 x 16 output pixels are PK x PK whole images; each
// image keeps its own zero border in the halo, so the halo is (16 + 2 PK)^2 pixels and output pixel (r, c) reads halo
// (r + 2 (r / S) + kh, c + 2 (c / S) + kw), S = 16 / PK -- a per-wave row shift and a per-lane column shift on the fragment base.
template <int PK> struct DmGeo {
  static constexpr int S = 16 / PK;                                 // image side (PK > 1)
  static constexpr int HW = 16 + 2 * PK;                            // halo row pitch in pixels (18 / 20 / 24)
  static constexpr int HALO = HW * HW;                              // 324 / 400 / 576 halo pixels
  static constexpr int HALO_INST = (HALO * 4 + 63) / 64;            // 21 / 25 / 36 wave-wide DMA instructions (1 KB each)
  static constexpr int HALO_ROUNDS = (HALO_INST + 7) / 8;           // 3 / 4 / 5 rounds of 8 waves
  static constexpr int HALO_BYTES = HALO_INST * 1024;               // 21504 / 25600 / 36864
  static_assert(HALO_BYTES % 512 == 0, "regions must start on a multiple of 8 rows (swizzle period)");
};
// J = 16-channel groups per wave: the workgroup's channel block is CB = 32 J output channels.  J = 2 everywhere but on packed
// launches that would leave most of the chip idle (J = 1: twice the workgroups, half the weight panel and MFMAs per stage).
template <int J> struct DmW {
  static constexpr int CB = 32 * J;
  static constexpr int INST = 9 * CB * 4 / 64;                           // 36 / 18 wave-wide DMA instructions per stage
  static constexpr int ROUNDS = (INST + 7) / 8;                          // 5 (the last: waves 0..3) / 3 (the last: waves 0..1)
};
constexpr unsigned DM_OOB = 0x80000000u;
template <int PK, int J = 2> constexpr int dm_buf_bytes() { return DmGeo<PK>::HALO_BYTES + DmW<J>::INST * 1024; }   // 58368 / 62464 / 73728
}  // namespace

// Cycle stamps (tools/trace_dma.py builds a private -DTG_DMA_TRACE copy of the library; the product build has none of it).
#ifdef TG_DMA_TRACE
__device__ unsigned long long tg_dma_trace_buf[64];
#define DM_STAMP(i)                                                                                          \
  do {                                                                                                       \
    if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && threadIdx.x == 0 && (i) < 64)                      \
      tg_dma_trace_buf[(i)] = (unsigned long long)clock64();                                                 \
  } while (0)
extern "C" int tg_debug_dma_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_dma_trace_buf), sizeof(unsigned long long) * 64);
}
#else
#define DM_STAMP(i) do { } while (0)
#endif

// (Round 4, session D: the stage's two streams read 64-byte pieces of longer rows -- half cache lines, the pattern that halved the
// weight stream of resblock_lat.hip.  Timing-only hooks that addressed the weight panel as one contiguous 36 KB block per stage
// and the halo as channel-blocked activations bought 2-3 % here (1105 -> 1133 TFLOP/s at [76,32,32,256], profiles/
// r04d_dma_stream.txt): this kernel is not bound by its line requests, and neither layout change was built.)
// (A two-tiles-per-stage variant -- one weight panel for two halos, 39 instead of 57 KB per 256 output pixels -- lost twice:
// alone to tile quantisation (1044 vs 1143 TFLOP/s at [76,32,32,256], profiles/r02t_microbench.txt), and in round 3 also as
// "whole rounds of pairs + a second launch of single tiles for the remainder" (78.8 -> 81.8 us, profiles/r03q_microbench.txt):
// a pair stage costs ~1.65 x a single one, not the 1.35 x its byte count suggests -- the stage is not purely stream-bound.)
template <bool HAS_RES, bool HAS_AUX, int PK, int J = 2>
__global__ __launch_bounds__(512, 1) void conv3x3_dma_kernel(ConvDmaP p) {
  using G = DmGeo<PK>;
  constexpr int CB = DmW<J>::CB, DM_W_INST = DmW<J>::INST, DM_W_ROUNDS = DmW<J>::ROUNDS;
  constexpr int NT = 1;
  constexpr int DM_HW = G::HW, DM_HALO = G::HALO, DM_HALO_INST = G::HALO_INST, DM_HALO_ROUNDS = G::HALO_ROUNDS;
  constexpr int DM_HALO_BYTES = G::HALO_BYTES;
  constexpr int BUF = dm_buf_bytes<PK, J>(), WOFF = DM_HALO_BYTES;
  constexpr int ROUNDS = DM_HALO_ROUNDS + DM_W_ROUNDS;                      // 8 / 9 / 10 DMA rounds per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 2 x BUF
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;                  // 4 x 2 waves: 4 pixel rows x 16 J channels each
  const int frow = lane & 15, fg = lane >> 4;
  // Packed tiles (8x8 / 4x4 images: VGG conv5, FNet's inner levels) are WEIGHT-heavy -- 4.7 MB of weights against 2 MB of
  // activations at conv5 -- and with blockIdx.y = channel block every XCD (linear workgroup id % 8) met all channel blocks and
  // pulled the whole panel into its own L2: 38.7 MB of HBM traffic for 8.7 MB of algorithmic bytes (profiles/r04_pmc_train.json).
  // Here the CHANNEL BLOCK is the fast index of the linear id, so an XCD owns Cout / 64 / 8 of the panel and re-reads the
  // (small) activations instead.  PK = 1 keeps blockIdx.y: its channel blocks of one pixel tile share an XCD's copy of the halo.
  const int lin = blockIdx.x + gridDim.x * blockIdx.y, nblk = gridDim.y;
#ifdef DM_NO_XCD_REMAP
  constexpr bool REMAP = false;
#else
  constexpr bool REMAP = PK != 1;
#endif
  const int bx = REMAP ? lin / nblk : (int)blockIdx.x, by = REMAP ? lin % nblk : (int)blockIdx.y;
  const int n0 = by * CB;
  const int cbase = n0 + wn * 16 * J;
  const int row_bytes = p.Cin * 2;
  const int nchunk = p.Cin >> 5;
  const int nunits = (p.ntiles + NT - 1) / NT;              // a work unit = NT consecutive tiles

  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  const auto rsrcO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
  const auto rsrcR = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_RES ? p.res : p.out), 0, (int)p.out_bytes, 0x00020000);
  const auto rsrcM = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_AUX ? p.aux : p.out), 0, (int)p.out_bytes, 0x00020000);

  // ---- LDS-DMA slot descriptors of this lane (stage-independent).  A wave-wide DMA instruction fills 64 consecutive
  //      16-byte slots; slot S of a region is row q = S / 4, position S % 4, and holds channel group (S % 4) ^ 2*((q >> 2) & 1)
  //      of that row (the swizzle).  Halo regions: row = halo pixel (instruction wave + 8k, k < 3; 21 instructions, the
  //      same descriptors for every tile of the stage); weight region: row = tap * 64 + channel (wave + 8k, k < 5; 36).
  int hrel[DM_HALO_ROUNDS], hcode[DM_HALO_ROUNDS], wrel[DM_W_ROUNDS];
#pragma unroll
  for (int k = 0; k < DM_HALO_ROUNDS; ++k) {
    const int S = (wave + 8 * k) * 64 + lane;
    const int q = S >> 2, ch = (S & 3) ^ (((S >> 4) & 1) << 1);
    const int dy = q / DM_HW, dx = q - DM_HW * dy;
    if constexpr (PK == 1) {
      hrel[k] = (dy * p.W + dx) * row_bytes + ch * 16;
      hcode[k] = dy | (dx << 8) | (q < DM_HALO ? (1 << 24) : 0);
    } else {                  // packed: block (by, bx) = image within the tile, (ry, rx) = pixel of that image (or its border)
      constexpr int S = G::S;
      const int by = dy / (S + 2), ry = dy - by * (S + 2) - 1, bx = dx / (S + 2), rx = dx - bx * (S + 2) - 1;
      const int img = by * PK + bx;
      const bool in = q < DM_HALO && (unsigned)ry < (unsigned)S && (unsigned)rx < (unsigned)S;
      hrel[k] = ((img * S + ry) * S + rx) * row_bytes + ch * 16;           // images are contiguous: S * S pixels each
      hcode[k] = img | (in ? (1 << 24) : 0);
    }
  }
  // The 36 instructions of the weight panel are issued in an order ROTATED per workgroup: every workgroup of a channel
  // block streams the same 36 KB per stage (measured neutral against the fixed order, profiles/r02s_microbench.txt; kept).
  const int rot = (int)(((unsigned)bx * 7u + (unsigned)by * 3u) % (unsigned)DM_W_INST);
  int winst[DM_W_ROUNDS];
#pragma unroll
  for (int k = 0; k < DM_W_ROUNDS; ++k) {
    const int i0 = wave + 8 * k;                            // issue slot; slots >= 36 of the last round are idle
    winst[k] = (i0 + rot) % DM_W_INST;                      // the instruction (1 KB of the panel) this slot fetches
    const int S = winst[k] * 64 + lane;
    const int q = S >> 2, ch = (S & 3) ^ (((S >> 4) & 1) << 1);
    const int tap = q / CB, co = n0 + (q % CB);
    const int wt = p.flip ? 8 - tap : tap;
    wrel[k] = (i0 < DM_W_INST && co < p.Cout) ? ((wt * p.Cout + co) * p.Cin) * 2 + ch * 16 : -1;
  }

  // One DMA round (r compile-time after unrolling) of the stage set up by dma_setup: rounds [3t, 3t+3) = halo of tile t,
  // the last five = the weight panel.
  int d_y0[NT], d_x0[NT], d_base[NT], d_wofs = 0;
  bool d_tok[NT];
  unsigned char* d_dst = smem;
  auto dma_setup = [&](int unit, int chunk, int buf) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int tile = unit * NT + t;
      d_tok[t] = tile < p.ntiles;
      if constexpr (PK == 1) {
        const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
        const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
        d_y0[t] = ty * DM_TH - 1;
        d_x0[t] = tx * 16 - 1;
        d_base[t] = ((n * p.H + d_y0[t]) * p.W + d_x0[t]) * row_bytes + chunk * 64;    // wave-uniform
      } else {                // packed: the tile's first image; d_y0 = images of the batch left from there on
        d_y0[t] = p.N - tile * PK * PK;
        d_x0[t] = 0;
        d_base[t] = tile * PK * PK * G::S * G::S * row_bytes + chunk * 64;
      }
    }
    d_wofs = chunk * 64;
    d_dst = smem + buf * BUF;
  };
  auto dma_round = [&](int r) {
    if (r < NT * DM_HALO_ROUNDS) {
      const int t = r / DM_HALO_ROUNDS, k = r % DM_HALO_ROUNDS;
      const int inst = wave + 8 * k;                                        // wave-uniform
      if (k + 1 < DM_HALO_ROUNDS || inst < DM_HALO_INST) {
        bool ok;
        if constexpr (PK == 1) {
          const int dy = hcode[k] & 255, dx = (hcode[k] >> 8) & 255;
          ok = (hcode[k] >> 24) && d_tok[t] && (unsigned)(d_y0[t] + dy) < (unsigned)p.H && (unsigned)(d_x0[t] + dx) < (unsigned)p.W;
        } else {
          ok = (hcode[k] >> 24) && d_tok[t] && (hcode[k] & 255) < d_y0[t];
        }
        const unsigned off = ok ? (unsigned)(d_base[t] + hrel[k]) : DM_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void_d*)(d_dst + t * DM_HALO_BYTES + inst * 1024), 16, (int)off, 0, 0, 0);
      }
    } else {
      const int k = r - NT * DM_HALO_ROUNDS;
      if (k + 1 < DM_W_ROUNDS || wave + 8 * k < DM_W_INST) {
        const unsigned off = wrel[k] >= 0 ? (unsigned)(wrel[k] + d_wofs) : DM_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_void_d*)(d_dst + WOFF + winst[k] * 1024), 16, (int)off, 0, 0, 0);
      }
    }
  };

  int unit = bx;
  if (unit >= nunits) return;
  DM_STAMP(0);
  dma_setup(unit, 0, 0);
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) dma_round(r);
  DM_STAMP(1);

  float bv[J][4];
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cbase + j * 16 + fg * 4 + r;
      bv[j][r] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
    }

  // ---- fragment addresses.  Halo fragment (row r, tap column kw): pixel q = Q0 + K with Q0 = 72 wm + frow (lane part)
  //      and K = 18 r + kw (compile time).  The swizzle bit (q >> 2) & 1 depends only on (Q0 + K) mod 8, so eight lane
  //      bases cover every K: the read is base[K & 7] + 64 K as an immediate offset -- no address arithmetic per read.
  // packed tiles: + 2 halo rows per image block above this wave's rows (wave-uniform), + 2 columns per block left of the lane's
  const int Q0 = PK == 1 ? wm * 4 * DM_HW + frow
                         : (wm * 4 + 2 * ((wm * 4) / G::S)) * DM_HW + frow + 2 * (frow / G::S);
  int abase[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) abase[d] = Q0 * 64 + ((fg ^ ((((Q0 & 7) + d) >> 2 & 1) << 1)) << 4);
  // weight fragment (tap, j): row tap*64 + 32 wn + 16 j + frow -- the swizzle bit is (frow >> 2) & 1
  const int bbase = WOFF + (wn * 16 * J + frow) * 64 + ((fg ^ (((frow >> 2) & 1) << 1)) << 4);

  f32x4 acc[NT][4][J];
  int chunk = 0, buf = 0;
  [[maybe_unused]] int it = 0;
  bool prev_epi = false;
  while (true) {
    int nunit = unit, nch = chunk + 1;
    if (nch == nchunk) {
      nch = 0;
      nunit = unit + gridDim.x;
    }
    // This wave's DMA slots of the stage have landed.  The VMEM queue holds (oldest first) the stage's DMA and, after an
    // epilogue, that unit's 8 NT stores: vmcnt retires in order, so a counted wait covers the DMA without draining the stores.
    if (prev_epi) {
      if constexpr (NT == 1 && J == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (NT == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DM_STAMP(2 + 4 * it);
    __builtin_amdgcn_s_barrier();                           // every wave's slots landed; nobody still reads the other buffer
    DM_STAMP(3 + 4 * it);
    // The next stage's DMA is issued IN BETWEEN the MFMA groups below (two rounds after each of the first groups of 8
    // MFMAs): issuing the rounds back to back cost each wave ~1000 cycles (an LDS-DMA instruction takes 60-180 cycles to
    // issue, MI355X_MICROARCH.md) during which BOTH waves of a SIMD -- they leave the barrier together -- fed no MFMA.
    const bool has_next = nunit < nunits;
    if (has_next) dma_setup(nunit, nch, buf ^ 1);

    if (chunk == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < J; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const unsigned char* sb = smem + buf * BUF;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      u32x4d af[NT][6];                                     // halo rows 0..5 of this wave at column offset kw, per tile
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const int K = r * DM_HW + kw;
          af[t][r] = *reinterpret_cast<const u32x4d*>(sb + t * DM_HALO_BYTES + abase[K & 7] + K * 64);
        }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        u32x4d bfr[J];
#pragma unroll
        for (int j = 0; j < J; ++j)
          bfr[j] = *reinterpret_cast<const u32x4d*>(sb + bbase + ((kh * 3 + kw) * CB + j * 16) * 64);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j)
              acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[j]),
                                                                     __builtin_bit_cast(bf16x8, af[t][i + kh]), acc[t][i][j], 0, 0, 0);
          const int grp = (kw * 3 + kh) * NT + t;           // compile time: group of 8 MFMAs just issued
          if (2 * grp < ROUNDS) {
            __builtin_amdgcn_sched_barrier(0);
            if (has_next) {
              dma_round(2 * grp);
              if (2 * grp + 1 < ROUNDS) dma_round(2 * grp + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    DM_STAMP(4 + 4 * it);

    prev_epi = chunk == nchunk - 1;
    if (prev_epi) {
      // ---- epilogue in registers: accumulator r of lane (frow, fg) = pixel column frow, output channel cbase+16j+4fg+r
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int tile = unit * NT + t;
        int n, x, ybase;
        if constexpr (PK == 1) {
          const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
          const int ty = t1 % p.tiles_y;
          n = t1 / p.tiles_y;
          x = tx * 16 + frow;
          ybase = ty * DM_TH + wm * 4;
        } else {              // packed: tile row r = 4 wm + i, column frow -> image (r / S) PK + frow / S, pixel (r % S, frow % S)
          n = tile * PK * PK + ((wm * 4) / G::S) * PK + frow / G::S;
          x = frow % G::S;
          ybase = (wm * 4) % G::S;
        }
        unsigned offs[4][J];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int y = ybase + i;
          const bool pok = tile < p.ntiles && y < p.H && x < p.W && n < p.N;
#pragma unroll
          for (int j = 0; j < J; ++j) {
            const int co = cbase + j * 16 + fg * 4;
            offs[i][j] = (pok && co < p.Cout) ? (unsigned)((((n * p.H + y) * p.W + x) * p.Cout + co) * 2) : DM_OOB;
          }
        }
        u32x2d rr[HAS_RES ? 4 : 1][J], aa[HAS_AUX ? 4 : 1][J];
        if constexpr (HAS_RES) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) rr[i][j] = __builtin_amdgcn_raw_buffer_load_b64(rsrcR, (int)offs[i][j], 0, 0);
        }
        if constexpr (HAS_AUX) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) aa[i][j] = __builtin_amdgcn_raw_buffer_load_b64(rsrcM, (int)offs[i][j], 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < J; ++j) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = acc[t][i][j][r] + bv[j][r];
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
            u32x2d o;
            o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
            o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
            __builtin_amdgcn_raw_buffer_store_b64(o, rsrcO, (int)offs[i][j], 0, 0);
          }
        }
      }
    }
    DM_STAMP(5 + 4 * it);
    if (nunit >= nunits) break;
    unit = nunit;
    chunk = nch;
    buf ^= 1;
    ++it;
  }
}

template <bool HAS_RES, bool HAS_AUX, int PK, int J>
static void launch_dma(const ConvDmaP& p, hipStream_t st) {
  auto kern = conv3x3_dma_kernel<HAS_RES, HAS_AUX, PK, J>;
  constexpr int LDS = 2 * dm_buf_bytes<PK, J>();
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  const int nt = p.Cout / DmW<J>::CB;
  const int nunits = p.ntiles;
  // persistent over work units: one workgroup per CU; grid.x a multiple of 8 so that the channel blocks of one pixel tile
  // (workgroup ids x, x + grid.x, ...) land on the SAME XCD and share its L2 copy of the halo
  int gx = tg_num_cus() / nt;
  if (gx < 8) gx = 8;
  gx &= ~7;
  if (gx > nunits) gx = nunits;
  static const char* const pname =
      PK == 1 ? (HAS_RES ? (HAS_AUX ? "conv3x3_dma<res,aux>" : "conv3x3_dma<res>") : (HAS_AUX ? "conv3x3_dma<aux>" : "conv3x3_dma<>"))
      : PK == 2 ? (HAS_RES ? (HAS_AUX ? "conv3x3_dma_pack2<res,aux>" : "conv3x3_dma_pack2<res>")
                           : (HAS_AUX ? "conv3x3_dma_pack2<aux>" : "conv3x3_dma_pack2<>"))
                : (HAS_RES ? (HAS_AUX ? "conv3x3_dma_pack4<res,aux>" : "conv3x3_dma_pack4<res>")
                           : (HAS_AUX ? "conv3x3_dma_pack4<aux>" : "conv3x3_dma_pack4<>"));
  const double px = (double)p.N * p.H * p.W;
  TG_LAUNCH(pname, 2.0 * px * p.Cout * 9.0 * p.Cin,
            px * (p.Cin * 2.0 + p.Cout * 2.0 * (1 + HAS_RES + HAS_AUX)) + 18.0 * p.Cin * p.Cout, kern, dim3(gx, nt), dim3(512),
            LDS, st, p);
}

template <int PK, int J>
static void launch_dma_pk(const ConvDmaP& p, bool res, bool aux, hipStream_t st) {
  if (res && aux) launch_dma<true, true, PK, J>(p, st);
  else if (res) launch_dma<true, false, PK, J>(p, st);
  else if (aux) launch_dma<false, true, PK, J>(p, st);
  else launch_dma<false, false, PK, J>(p, st);
}

// Returns 1 if the descriptor was handled here, 0 otherwise (the halo-tile kernel of conv3x3.hip takes it).
int tg_conv3x3_dma_try(const tg_conv_desc* d, const void* in, const void* weight, const float* bias, const void* res,
                       const void* aux, void* out, hipStream_t st) {
  // selection threshold in workgroup-units (tiles x channel blocks).  Round 2 set 96 ("below a third of the chip the 8 x 64
  // tiles fill it better"); re-measured in round 3, alternating runs on one box (profiles/r03s_ab.txt): the 1080p inference frame
  // 1.070 -> 1.042 ms with 24 (FNet's 33 x 60 / 66 x 120 levels: the tile kernel's K loop is the longer serial chain there),
  // the training steps unchanged (TecoGAN 12.09-12.15 ms with either)
  constexpr int min_wg = 24;
  if (d->in_dtype != TG_BF16 || d->out_dtype != TG_BF16) return 0;
  if (d->Cin % 32 != 0 || d->Cin < 64 || d->Cout % 64 != 0) return 0;
  if (d->act >= TG_ACT_TANH) return 0;
  // images of exactly 8 x 8 / 4 x 4 pixels: packed tiles (4 / 16 whole images per 16 x 16 tile); other small sizes: conv3x3.hip
  constexpr int min_wg_pack = 16;
  const int pk = (d->Hin == 8 && d->Win == 8) ? 2 : ((d->Hin == 4 && d->Win == 4) ? 4 : 1);
  if (pk == 1 && (d->Hin <= 8 || d->Win <= 8)) return 0;
  if ((((uintptr_t)in | (uintptr_t)weight | (uintptr_t)out | (uintptr_t)res | (uintptr_t)aux) & 15)) return 0;
  const int64_t px = (int64_t)d->N * d->Hin * d->Win;
  const int64_t in_bytes = px * d->Cin * 2, out_bytes = px * d->Cout * 2, w_bytes = (int64_t)9 * d->Cout * d->Cin * 2;
  if (in_bytes >= ((int64_t)1 << 31) || out_bytes >= ((int64_t)1 << 31)) return 0;
  ConvDmaP p;
  p.in = in; p.w = weight; p.bias = bias; p.res = res; p.aux = aux; p.out = out;
  p.N = d->N; p.H = d->Hin; p.W = d->Win; p.Cin = d->Cin; p.Cout = d->Cout;
  p.flip = d->mode == 1;
  p.nslope = d->act == TG_ACT_RELU ? 0.f : (d->act == TG_ACT_LRELU ? d->act_alpha : 1.f);
  p.mslope = d->mask_act == TG_ACT_RELU ? 0.f : (d->mask_act == TG_ACT_LRELU ? d->mask_alpha : 1.f);
  p.tiles_y = (p.H + DM_TH - 1) / DM_TH;
  p.tiles_x = (p.W + 15) / 16;
  const int64_t ntiles = pk == 1 ? (int64_t)p.N * p.tiles_y * p.tiles_x : ((int64_t)p.N + pk * pk - 1) / (pk * pk);
  if (ntiles * (p.Cout / 64) < (pk == 1 ? min_wg : min_wg_pack) || ntiles >= ((int64_t)1 << 30)) return 0;
  p.ntiles = (int)ntiles;
  p.in_bytes = (unsigned)in_bytes; p.w_bytes = (unsigned)w_bytes; p.out_bytes = (unsigned)out_bytes;
  // packed launches that would fill less than the chip even with 32-channel blocks: twice the workgroups, each with half the
  // weight panel and half the MFMAs per stage (VGG conv5 at 32 images: 64 -> 128 workgroups; conv5 at 28-48 images 28.1 -> 21.7 us, profiles/r04y_ab.txt)
  const bool j1 = pk > 1 && ntiles * (p.Cout / 32) <= 256;
  if (pk == 2 && j1) launch_dma_pk<2, 1>(p, res != nullptr, aux != nullptr, st);
  else if (pk == 2) launch_dma_pk<2, 2>(p, res != nullptr, aux != nullptr, st);
  else if (pk == 4 && j1) launch_dma_pk<4, 1>(p, res != nullptr, aux != nullptr, st);
  else if (pk == 4) launch_dma_pk<4, 2>(p, res != nullptr, aux != nullptr, st);
  else launch_dma_pk<1, 2>(p, res != nullptr, aux != nullptr, st);
  return 1;
}
