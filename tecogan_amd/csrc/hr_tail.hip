// Fused HR tail of generator_F for the throughput (inference) regime, bf16 -- gfx950:
//
//     t2    = relu(conv2d_transpose_k3s2(t1, W2) + b2)             reference lib/frvsr.py:73-78  (conv_tran2, second stage)
//     c     = conv3x3(t2, W3) + b3                       (64 -> 3)  reference lib/frvsr.py:80-81  (output_stage)
//     frame = (c + bicubic_four(LR)) * 2 - 1                        reference lib/frvsr.py:83-87, lib/ops.py:166-212
//     state = (frame + 1) / 2                                       reference main.py:207 (the inference loop's recurrent state)
//
// in ONE kernel, so that the 64-channel HR tensor t2 never touches HBM: at 1920x1080 it is 265 MB, written by the
// transposed conv at 3.2 TB/s (65 us) and read back by the output conv at 2.6 TB/s (105 us) -- both HBM-bound, together
// with the separate bicubic pass (41 us) 210 us of a 1.11 ms frame (profiles/r02u_infer1080p_bf16_kernel_stats.txt).
//
// Structure (one 256-thread workgroup per CU, persistent over tiles; the transposed-conv part is deconv3x3s2_ws_kernel of
// conv3x3_ws.hip: weights of BOTH convs in registers, input halo tiles by LDS-DMA into a double buffer):
//   * a tile is 8 x 16 pixels of t1 (plus the halo row / column up-left) -> the 16 x 32 block of t2 it determines, staged
//     as bf16 in LDS (73 KB, 144-byte pixel pitch; positions outside the image are stored as ZERO: they are the output
//     conv's SAME padding, not "transposed conv of zero input" = relu(b2));
//   * the output conv then needs a one-pixel ring, so it produces only the block's interior 14 x 30 pixels and tiles
//     advance by 7 x 15 input pixels (14 x 30 output pixels): 22 % of the transposed conv is recomputed at tile borders;
//   * output conv on MFMA with swapped operands: A = W3 padded 3 -> 16 rows (18 fragments in registers), B = 16 consecutive
//     pixels of one row of the staged block; 28 groups of 16 pixels per tile (14 rows x 2 column blocks, the second one
//     overlapping by two columns), 7 per wave, 18 ds_read_b128 + 18 MFMAs each; lanes 0..15 hold the 3 channels of a pixel;
//   * bicubic in the same epilogue: the 4 x 4 LR neighbourhood of a pixel is gathered by the four 16-lane groups of the
//     wave (group g reads LR row g, four 8-byte loads), weighted, and summed across the groups with two xor-shuffles.
//
// Round 3: validated on MI355X (tests/test_kernels_gpu.py::test_hr_tail_fused_matches_the_three_kernel_path, every frame of
// tests/test_infer_gpu.py) and default-on (TG_HR_TAIL=0 is the A/B switch): 1080p frame 1.103 -> 1.068 ms same box
// (profiles/r03a_ab.txt).  rocprofv3: 224 us per 1080p frame (profiles/r03l_infer1080p_bf16_kernel_stats.txt), i.e. ~6900
// cycles per tile against ~4600 cycles of MFMA issue (270 v_mfma_16x16x32 per wave and tile at one wave per SIMD): it is
// MFMA-ISSUE bound, and 126 of the 270 are the output conv with 3 of 16 operand rows in use.  Hoisting the bicubic skip's LR
// loads ahead of the transposed-conv phase changed nothing (224.2 us).  Open: the output conv as ONE [pixel] x [tap, channel]
// product (27 -> 32 columns, 4 MFMAs per 16 pixels instead of 18) followed by a shifted 9-tap sum of the partial products.
#include "common.h"
#include <mutex>
#include <stdlib.h>

struct HrTailP {
  const void* t1;       // [N, h2, w2, 64] bf16
  const void* wd;       // transposed conv: [9][64][64] bf16, TF conv2d_transpose layout [kh,kw,Cout,Cin]
  const float* bd;      // [64]
  const void* wo;       // output conv: [9][3][64] bf16  ([tap][Cout][Cin])
  const float* bo;      // [3]
  const void* gen_in;   // [N, h, w, Cpad] bf16: LR frame in channels 0..2 (h = h2 / 2, w = w2 / 2)
  float* out;           // [N, 2 h2, 2 w2, 3] fp32 in [-1,1], nullable
  float* state;         // same shape, (out + 1) / 2, nullable
  int N, h2, w2, Cpad;
  int tiles_y, tiles_x, ntiles;
  unsigned t1_bytes, lr_bytes, o_bytes;
};

typedef unsigned int u32x4h __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3h __attribute__((ext_vector_type(3)));
typedef unsigned int u32x2h __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_h;

namespace {
constexpr int HT_TH = 8, HT_HALO = (HT_TH + 2) * 18, HT_ROWB = 144;
constexpr int HT_SLOTS = HT_HALO * 9;                      // 16-byte slots of one halo tile (8 data + 1 pad per pixel)
constexpr int HT_NDMA = (HT_SLOTS + 63) / 64;              // 26 wave-wide DMA instructions per tile
constexpr int HT_BUF = HT_NDMA * 1024;                     // 26624
constexpr int HT_KPW = (HT_NDMA + 3) / 4;                  // 7
constexpr int HT_STAGE = 16 * 32 * HT_ROWB;                // staged t2 block: 73728 bytes
constexpr int HT_LDS = 2 * HT_BUF + HT_STAGE;              // 126976
constexpr int HT_SY = 14, HT_SX = 30;                      // output pixels a tile finishes
constexpr unsigned HT_OOB = 0x80000000u;
__constant__ float kBicubicH[4][4] = {{0.f, 1.f, 0.f, 0.f},
                                      {-0.10546875f, 0.87890625f, 0.26171875f, -0.03515625f},
                                      {-0.09375f, 0.59375f, 0.59375f, -0.09375f},
                                      {-0.03515625f, 0.26171875f, 0.87890625f, -0.10546875f}};
}  // namespace

__global__ __launch_bounds__(256, 1) void hr_tail_kernel(HrTailP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 2 x HT_BUF | HT_STAGE
  unsigned char* stage = smem + 2 * HT_BUF;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;                  // transposed conv: 4 input rows x 32 channels per wave
  const int frow = lane & 15, fg = lane >> 4;
  const int cbase = wn * 32;
  const int Ho = 2 * p.h2, Wo = 2 * p.w2, h = p.h2 >> 1, w = p.w2 >> 1;

  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.t1), 0, (int)p.t1_bytes, 0x00020000);
  const auto rsrcW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wd), 0, 9 * 64 * 64 * 2, 0x00020000);
  const auto rsrcW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wo), 0, 9 * 3 * 64 * 2, 0x00020000);
  const auto rsrcL = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.gen_in), 0, (int)p.lr_bytes, 0x00020000);
  const auto rsrcO = __builtin_amdgcn_make_buffer_rsrc(p.out ? (void*)p.out : (void*)p.state, 0, (int)p.o_bytes, 0x00020000);
  const auto rsrcS = __builtin_amdgcn_make_buffer_rsrc(p.state ? (void*)p.state : (void*)p.out, 0, (int)p.o_bytes, 0x00020000);

  // ---- LDS-DMA slot descriptors (as conv3x3_ws.hip): slot S = (wave + 4k)*64 + lane = chunk S % 9 of halo pixel S / 9
  int code[HT_KPW];
#pragma unroll
  for (int k = 0; k < HT_KPW; ++k) {
    const int S = (wave + 4 * k) * 64 + lane;
    const int pix = S / 9, c = S - 9 * pix;
    const int dy = pix / 18, dx = pix - 18 * dy;
    code[k] = dy | (dx << 8) | (c << 16) | ((S < HT_SLOTS && c < 8) ? (1 << 24) : 0);
  }
  // tile (ty, tx): input rows [7 ty - 1, 7 ty + 7), columns [15 tx - 1, 15 tx + 15); halo origin one further up-left
  auto issue_dma = [&](int tile, int buf) {
    const int tx = tile % p.tiles_x, t1 = tile / p.tiles_x;
    const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
    const int y0 = ty * (HT_SY / 2) - 2, x0 = tx * (HT_SX / 2) - 2;
    const int base = ((n * p.h2 + y0) * p.w2 + x0) * 128;                   // wave-uniform
#pragma unroll
    for (int k = 0; k < HT_KPW; ++k) {
      const int inst = wave + 4 * k;
      if (4 * k + 3 < HT_NDMA || inst < HT_NDMA) {
        const int dy = code[k] & 255, dx = (code[k] >> 8) & 255, c = (code[k] >> 16) & 255;
        const bool ok = (code[k] >> 24) && (unsigned)(y0 + dy) < (unsigned)p.h2 && (unsigned)(x0 + dx) < (unsigned)p.w2;
        const unsigned off = ok ? (unsigned)(base + (dy * p.w2 + dx) * 128 + c * 16) : HT_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void_h*)(smem + buf * HT_BUF + inst * 1024), 16, (int)off, 0, 0, 0);
      }
    }
  };

  int tile = blockIdx.x;
  if (tile >= p.ntiles) return;
  issue_dma(tile, 0);

  // ---- weights -> registers.  Transposed conv: lane (frow, fg) of fragment (tap, kk, j) = W2[tap][cbase+16j+frow][32kk+8fg..].
  u32x4h wf[9][2][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wf[tap][kk][j] = __builtin_amdgcn_raw_buffer_load_b128(
            rsrcW2, ((tap * 64 + cbase + j * 16 + frow) * 64 + kk * 32 + fg * 8) * 2, 0, 0);
  // Output conv: A operand rows = output channels, 3 real ones: lane (frow, fg) of (tap, kk) = W3[tap][frow][32kk+8fg..], 0 for frow >= 3
  u32x4h w3f[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      w3f[tap][kk] = __builtin_amdgcn_raw_buffer_load_b128(
          rsrcW3, (int)(frow < 3 ? (unsigned)(((tap * 3 + frow) * 64 + kk * 32 + fg * 8) * 2) : HT_OOB), 0, 0);
  float bv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[j][r] = p.bd ? p.bd[cbase + j * 16 + fg * 4 + r] : 0.f;
  const float b3[3] = {p.bo ? p.bo[0] : 0.f, p.bo ? p.bo[1] : 0.f, p.bo ? p.bo[2] : 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int buf = 0;
  while (true) {
    const int ntile = tile + gridDim.x;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // A: this tile's halo landed; everybody is done reading the previous staged block
    if (ntile < p.ntiles) issue_dma(ntile, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    const int tx = tile % p.tiles_x, tq = tile / p.tiles_x;
    const int ty = tq % p.tiles_y, n = tq / p.tiles_y;
    const int yb = ty * HT_SY - 2, xb = tx * HT_SX - 2;      // image coordinates of the staged block's pixel (0, 0)

    // ---- transposed conv (deconv3x3s2_ws_kernel) into the staged block: block pixel (2 (4 wm + i) + py, 2 frow + px)
    const unsigned char* Afrag = smem + buf * HT_BUF + ((wm * 4) * 18 + frow) * HT_ROWB + fg * 16;
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
          for (int dyi = 0; dyi < (py ? 1 : 2); ++dyi)
#pragma unroll
            for (int dxi = 0; dxi < (px ? 1 : 2); ++dxi) {
              const int ky = py ? 1 : 2 * dyi, kx = px ? 1 : 2 * dxi;
              u32x4h af[4];
#pragma unroll
              for (int i = 0; i < 4; ++i)
                af[i] = *reinterpret_cast<const u32x4h*>(Afrag + ((i + 1 - dyi) * 18 + 1 - dxi) * HT_ROWB + kk * 64);
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ky * 3 + kx][kk][j]),
                                                                      __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0);
            }
        const int xl = 2 * frow + px, xo = xb + xl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int yl = 2 * (wm * 4 + i) + py, yo = yb + yl;
          const bool inside = (unsigned)yo < (unsigned)Ho && (unsigned)xo < (unsigned)Wo;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = inside ? fmaxf(acc[i][j][r] + bv[j][r], 0.f) : 0.f;
            u32x2h o;
            o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
            o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
            *reinterpret_cast<u32x2h*>(stage + (yl * 32 + xl) * HT_ROWB + (cbase + j * 16 + fg * 4) * 2) = o;
          }
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();           // B: the staged block is complete

    // ---- output conv + bicubic + value ranges: group g = (block row 1 + g / 2, column block g % 2), 7 groups per wave
#pragma unroll 1
    for (int gi = 0; gi < 7; ++gi) {
      const int g = wave + 4 * gi;
      const int yl = 1 + (g >> 1), cb = (g & 1) ? 15 : 1;   // pixel of lane frow: block column cb + frow
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned char* Bfrag = stage + ((yl - 1) * 32 + cb - 1 + frow) * HT_ROWB + fg * 16;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const u32x4h bf = *reinterpret_cast<const u32x4h*>(Bfrag + (kh * 32 + kw) * HT_ROWB + kk * 64);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w3f[kh * 3 + kw][kk]),
                                                          __builtin_bit_cast(bf16x8, bf), acc, 0, 0, 0);
          }
      // lanes fg == 0 hold channels 0..2 of pixel (yl, cb + frow) in acc[0..2]
      const int yo = yb + yl, xo = xb + cb + frow;
      const bool mine = (cb == 1 || frow >= 2) && (unsigned)yo < (unsigned)Ho && (unsigned)xo < (unsigned)Wo;
      // bicubic_four of the LR frame at (yo, xo): 16-lane group fg gathers LR row clamp(yo/4 - 1 + fg), weights Keys(-0.75)
      const int yc = min(max(yo, 0), Ho - 1), xc = min(max(xo, 0), Wo - 1);
      const int li = yc >> 2, lj = xc >> 2;
      const int ry = min(max(li + fg - 1, 0), h - 1);
      const float wy = kBicubicH[yc & 3][fg];
      float part[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int rx = min(max(lj + k - 1, 0), w - 1);
        const u32x2h q = __builtin_amdgcn_raw_buffer_load_b64(rsrcL, ((n * h + ry) * w + rx) * p.Cpad * 2, 0, 0);
        const float wgt = wy * kBicubicH[xc & 3][k];
        part[0] += wgt * __uint_as_float(q.x << 16);
        part[1] += wgt * __uint_as_float(q.x & 0xffff0000u);
        part[2] += wgt * __uint_as_float(q.y << 16);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        part[c] += __shfl_xor(part[c], 16, 64);
        part[c] += __shfl_xor(part[c], 32, 64);
      }
      if (fg == 0 && mine) {
        const unsigned off = (unsigned)((((n * Ho + yo) * Wo + xo) * 3) * 4);
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = acc[c] + b3[c] + part[c];
        if (p.out) {
          u32x3h o = {__float_as_uint(v[0] * 2.f - 1.f), __float_as_uint(v[1] * 2.f - 1.f), __float_as_uint(v[2] * 2.f - 1.f)};
          __builtin_amdgcn_raw_buffer_store_b96(o, rsrcO, (int)off, 0, 0);
        }
        if (p.state) {
          u32x3h o = {__float_as_uint((v[0] * 2.f - 1.f) * 0.5f + 0.5f), __float_as_uint((v[1] * 2.f - 1.f) * 0.5f + 0.5f),
                      __float_as_uint((v[2] * 2.f - 1.f) * 0.5f + 0.5f)};
          __builtin_amdgcn_raw_buffer_store_b96(o, rsrcS, (int)off, 0, 0);
        }
      }
    }
    tile = ntile;
    if (tile >= p.ntiles) break;
    buf ^= 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // next tile's halo (and this tile's few stores)
  }
}

// frame / state = the fused HR tail of generator_F (see the file header).  t1 [N,h2,w2,64] bf16 with h2, w2 even (twice the LR size).
extern "C" int tg_hr_tail_forward(const void* t1, const void* w2 /*[9][64][64] TF transposed-conv layout*/, const float* b2,
                                  const void* w3 /*[9][3][64]*/, const float* b3, const void* gen_in, int Cpad, float* out,
                                  float* state, int N, int h2, int w2dim, void* stream) {
  TG_CHECK_ARG(t1 && w2 && w3 && gen_in && (out || state), "null pointer");
  TG_CHECK_ARG(N > 0 && h2 > 0 && w2dim > 0 && (h2 & 1) == 0 && (w2dim & 1) == 0 && Cpad >= 4 && (Cpad & 3) == 0, "bad shape");
  TG_CHECK_ARG(((((uintptr_t)t1 | (uintptr_t)w2 | (uintptr_t)w3) & 15) == 0) && (((uintptr_t)gen_in & 7) == 0), "alignment");
  const int64_t t1_bytes = (int64_t)N * h2 * w2dim * 128, o_bytes = (int64_t)N * h2 * w2dim * 4 * 3 * 4;
  const int64_t lr_bytes = (int64_t)N * (h2 / 2) * (w2dim / 2) * Cpad * 2;
  TG_CHECK_ARG(t1_bytes < ((int64_t)1 << 31) && o_bytes < ((int64_t)1 << 31), "tensor too large for 32-bit buffer offsets");
  HrTailP p;
  p.t1 = t1; p.wd = w2; p.bd = b2; p.wo = w3; p.bo = b3; p.gen_in = gen_in; p.out = out; p.state = state;
  p.N = N; p.h2 = h2; p.w2 = w2dim; p.Cpad = Cpad;
  p.tiles_y = (2 * h2 + 1 + HT_SY - 1) / HT_SY;             // finished rows of tile ty: [14 ty - 1, 14 ty + 13)
  p.tiles_x = (2 * w2dim + 1 + HT_SX - 1) / HT_SX;
  p.ntiles = N * p.tiles_y * p.tiles_x;
  p.t1_bytes = (unsigned)t1_bytes; p.lr_bytes = (unsigned)lr_bytes; p.o_bytes = (unsigned)o_bytes;
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hr_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, HT_LDS);
  });
  int gx = p.ntiles < 256 ? p.ntiles : 256;
  const double px = (double)N * h2 * w2dim;
  TG_LAUNCH("hr_tail", 2.0 * px * 9.0 * 64 * 64 + 2.0 * 4 * px * 9.0 * 64 * 3, px * 128.0 + 4.0 * px * 12.0 * ((out != nullptr) + (state != nullptr)),
            hr_tail_kernel, dim3(gx), dim3(256), HT_LDS, static_cast<hipStream_t>(stream), p);
  TG_CHECK_LAUNCH();
  return TG_OK;
}
