// TWO consecutive residual blocks of generator_F (reference lib/frvsr.py:50-57, 66-70) -- or the input-gradient chain of two blocks
// -- as ONE launch, latency regime of the training recurrence: 304 launches per TecoGAN step where resblock_lat.hip has 608.
//
// Same construction as resblock_lat.hip (read its header first), four levels instead of two: a workgroup owns a 4x4 output tile and
// recomputes the halo of every intermediate tensor (10x10, 8x8, 6x6 pixels around it); a wave owns 16 channels of all four convs and
// streams its 72 weight fragments from the fragment-order copies as ONE stream with a prefetch distance.  What a node costs is the
// kernel boundary (1.5 us), one memory round trip and its weight stream (4 x 73 KB per CU at ~43 B/clk = 2.9 us): two blocks per
// node share the first two.
//   * LDS: buffer A = the 12x12 input region, updated IN PLACE at its 8x8 centre by level 2 (block output = input + conv: the same
//     lane reads and writes a position); buffer B = the 10x10 region of level 1, later the 6x6 region of level 3.  Both with a
//     160-byte pixel pitch and 12-position rows: conflict-free ds_read_b128 fragment reads for 4x4-pixel MFMA tiles
//     (tools/lds_layout_search.py: "4x4, Ri = 12, P = 160"); 42 KB in all -- the node still fits beside a VGG workgroup (117 KB);
//   * MFMA pixel tiles are 4x4 blocks (lane = pixel (frow / 4, frow % 4) of the block): 9 + 4 + 4 + 1 tiles for the four levels;
//     lanes outside a level's region compute on clamped positions and write to a dump position;
//   * every level stores the tile's own 4x4 pixels to HBM (the weight gradients read all four tensors after the BPTT);
//   * taps and K-steps in the order of conv3x3_tile's chain tiles, operands swapped the same way, bf16 rounding at the same places:
//     BIT-IDENTICAL to four tg_conv_forward launches (and to two tg_resblock launches); tests/test_kernels_gpu.py holds that.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

struct Rb2P {
  const void* x;        // [N,H,W,64] bf16  input of the first block (forward) / gradient w.r.t. the second block's output (backward)
  const void* w[4];     // fragment-order weights of the four convs in the order they are applied
  const float* b[4];    // biases, nullable
  const void* aux1;     // nullable: level-1 result *= (aux1 > 0)
  const void* aux3;     // nullable: level-3 result *= (aux3 > 0)
  const void* aux4;     // nullable: level-4 result *= (aux4 > 0)
  void* o[4];           // [N,H,W,64] results of the four levels (the tile's own pixels), o[0..2] nullable
  int N, H, W;
  int flip;             // 1: taps mirrored (input-gradient form)
  float nslope;         // activation of levels 1 and 3: max(v, v * nslope) (ReLU 0, none 1)
  int tiles_y, tiles_x, ntiles;
  unsigned bytes;
  int prio;
};

typedef unsigned int u32x4q __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2q __attribute__((ext_vector_type(2)));

namespace {
constexpr int R2_P = 160, R2_RP = 12;
constexpr int R2_APOS = 12 * R2_RP + 1, R2_BPOS = 10 * R2_RP + 1;      // + a dump position each
constexpr unsigned R2_OOB = 0x80000000u;
constexpr int R2_DIST = 14;
}  // namespace

template <int I, int N, typename F>
__device__ __forceinline__ void r2_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    r2_static_for<I + 1, N>(f);
  }
}

template <bool HAS_A1, bool HAS_A3, bool HAS_A4>
__global__ __launch_bounds__(256, 2) void resblock2_lat_kernel(Rb2P p) {
  __shared__ __attribute__((aligned(16))) unsigned char As[R2_APOS * R2_P];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[R2_BPOS * R2_P];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fg = lane >> 4;
  const int by = frow >> 2, bx = frow & 3;                 // pixel of a 4x4 MFMA tile
  if (p.prio) __builtin_amdgcn_s_setprio(3);
  int b = blockIdx.x;
  if ((p.ntiles & 7) == 0) b = (b & 7) * (p.ntiles >> 3) + (b >> 3);      // an XCD owns a contiguous range of tiles
  const int tx = b % p.tiles_x, t1 = b / p.tiles_x;
  const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
  const int y0 = ty * 4, x0 = tx * 4;
  const int cbyte = (wave * 16 + fg * 4) * 2;

  const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.bytes, 0x00020000);
  const auto rsW0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w[0]), 0, 9 * 64 * 64 * 2, 0x00020000);
  const auto rsW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w[1]), 0, 9 * 64 * 64 * 2, 0x00020000);
  const auto rsW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w[2]), 0, 9 * 64 * 64 * 2, 0x00020000);
  const auto rsW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w[3]), 0, 9 * 64 * 64 * 2, 0x00020000);

  // ---- global loads, in consumption order; none behind a branch (a null pointer is a zero-length buffer) -----------------------
  float bv[4][4];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b[l]), 0, p.b[l] ? 256 : 0, 0x00020000);
    const u32x4q q = __builtin_amdgcn_raw_buffer_load_b128(rsB, (wave * 16 + fg * 4) * 4, 0, 0);
    bv[l][0] = __uint_as_float(q.x); bv[l][1] = __uint_as_float(q.y); bv[l][2] = __uint_as_float(q.z); bv[l][3] = __uint_as_float(q.w);
  }
  constexpr int XITEMS = 144 * 8, XL = (XITEMS + 255) / 256;          // the 12x12 input region: 1152 16-byte items, 4.5 per thread
  u32x4q xr[XL];
#pragma unroll
  for (int k = 0; k < XL; ++k) {
    const int item = tid + k * 256;
    const int pix = min(item >> 3, 143), c = item & 7;
    const int gy = y0 - 4 + pix / 12, gx = x0 - 4 + pix % 12;
    const bool ok = item < XITEMS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(ok ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + c * 16) : R2_OOB), 0, 0);
  }
  // weight stream: fragment i of 72 = step i % 18 of conv i / 18; lane's 16 bytes at [step][wave][lane] (fragment order)
  u32x4q w0[18], w1[18], w2[18], w3[18];
  const int wlane = wave * 1024 + lane * 16;
  auto wstep = [&](int s) {
    const int tap = s >> 1, kk = s & 1;
    return ((p.flip ? 8 - tap : tap) * 2 + kk) * 4096;
  };
#define R2_WISSUE(i)                                                                                                        \
  do {                                                                                                                      \
    if constexpr ((i) < 18) w0[(i) % 18] = __builtin_amdgcn_raw_buffer_load_b128(rsW0, wlane, wstep((i) % 18), 0);          \
    else if constexpr ((i) < 36) w1[(i) % 18] = __builtin_amdgcn_raw_buffer_load_b128(rsW1, wlane, wstep((i) % 18), 0);     \
    else if constexpr ((i) < 54) w2[(i) % 18] = __builtin_amdgcn_raw_buffer_load_b128(rsW2, wlane, wstep((i) % 18), 0);     \
    else if constexpr ((i) < 72) w3[(i) % 18] = __builtin_amdgcn_raw_buffer_load_b128(rsW3, wlane, wstep((i) % 18), 0);     \
  } while (0)
  r2_static_for<0, R2_DIST>([&](auto i) { R2_WISSUE(decltype(i)::value); });
  // masks (backward): four channels of the pixel this lane finishes in each tile of levels 1, 3 and 4
  const auto rsA1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_A1 ? p.aux1 : p.x), 0, (int)p.bytes, 0x00020000);
  const auto rsA3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_A3 ? p.aux3 : p.x), 0, (int)p.bytes, 0x00020000);
  const auto rsA4 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HAS_A4 ? p.aux4 : p.x), 0, (int)p.bytes, 0x00020000);
  auto goff = [&](int gy, int gx, bool ok) {
    return (int)((ok && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + cbyte) : R2_OOB);
  };
  u32x2q m1[HAS_A1 ? 9 : 1], m3[HAS_A3 ? 4 : 1], m4;
  if constexpr (HAS_A1) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ry = 4 * (t / 3) + by, rx = 4 * (t % 3) + bx;
      m1[t] = __builtin_amdgcn_raw_buffer_load_b64(rsA1, goff(y0 - 3 + ry, x0 - 3 + rx, ry < 10 && rx < 10), 0, 0);
    }
  }
  if constexpr (HAS_A3) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ry = 4 * (t >> 1) + by, rx = 4 * (t & 1) + bx;
      m3[t] = __builtin_amdgcn_raw_buffer_load_b64(rsA3, goff(y0 - 1 + ry, x0 - 1 + rx, ry < 6 && rx < 6), 0, 0);
    }
  }
  if constexpr (HAS_A4) m4 = __builtin_amdgcn_raw_buffer_load_b64(rsA4, goff(y0 + by, x0 + bx, true), 0, 0);
  __builtin_amdgcn_sched_barrier(0);

#pragma unroll
  for (int k = 0; k < XL; ++k) {
    const int item = tid + k * 256;
    if (item < XITEMS) *reinterpret_cast<u32x4q*>(As + (item >> 3) * R2_P + (item & 7) * 16) = xr[k];
  }
  __syncthreads();

  auto unpack = [](const u32x2q& a, float (&f)[4]) {
    f[0] = __uint_as_float(a.x << 16); f[1] = __uint_as_float(a.x & 0xffff0000u);
    f[2] = __uint_as_float(a.y << 16); f[3] = __uint_as_float(a.y & 0xffff0000u);
  };
  auto pack = [](const float (&v)[4]) {
    u32x2q o;
    o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    return o;
  };

  // ---- one level: out (S x S, NB x NB tiles of 4x4) = conv3x3(in) over the region staged at `in` (origin = the level's input
  //      region); FIRST = index of the level's first weight fragment in the 72-fragment stream -----------------------------------
  // (a generic lambda over compile-time level parameters: everything below unrolls)
  auto run_level = [&](auto lv, const unsigned char* in, auto& wreg, auto&& epilogue) {
    constexpr int LV = decltype(lv)::value;
    constexpr int S = LV == 0 ? 10 : LV == 1 ? 8 : LV == 2 ? 6 : 4, NB = (S + 3) / 4, NT = NB * NB, FIRST = LV * 18;
    const unsigned char* base[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ry = min(4 * (t / NB) + by, S - 1), rx = min(4 * (t % NB) + bx, S - 1);     // (clamped: lanes outside the region)
      base[t] = in + (ry * R2_RP + rx) * R2_P + fg * 16;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto frag = [&](int s, int t) {
      const int tap = s >> 1, kk = s & 1;
      return *reinterpret_cast<const uint4*>(base[t] + ((tap / 3) * R2_RP + tap % 3) * R2_P + kk * 64);
    };
    uint4 bf[NT], nbf[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bf[t] = frag(0, t);
    r2_static_for<0, 18>([&](auto sv) {
      constexpr int s = decltype(sv)::value;
      R2_WISSUE(FIRST + s + R2_DIST);
      if constexpr (s < 17) {
#pragma unroll
        for (int t = 0; t < NT; ++t) nbf[t] = frag(s + 1, t);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wreg[s]), *reinterpret_cast<bf16x8*>(&bf[t]), acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t) bf[t] = nbf[t];
    });
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ry = 4 * (t / NB) + by, rx = 4 * (t % NB) + bx;
      epilogue(t, ry, rx, ry < S && rx < S, acc[t]);
    }
  };

  // level 1: h1 (10x10, origin y0-3) = act(conv(x) + b0) [* mask]; zero outside the image; -> B; own pixels (3..6) -> o[0]
  const auto rsO0 = __builtin_amdgcn_make_buffer_rsrc(p.o[0], 0, p.o[0] ? (int)p.bytes : 0, 0x00020000);
  run_level(std::integral_constant<int, 0>{}, As, w0, [&](int t, int ry, int rx, bool valid, const f32x4& a) {
    const int gy = y0 - 3 + ry, gx = x0 - 3 + rx;
    const bool inimg = valid && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = a[r] + bv[0][r];
      v[r] = fmaxf(v[r], v[r] * p.nslope);
    }
    if constexpr (HAS_A1) {
      float m[4];
      unpack(m1[t], m);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= m[r] > 0.f ? 1.f : 0.f;
    }
    u32x2q o = pack(v);
    if (!inimg) o = u32x2q{0u, 0u};
    *reinterpret_cast<u32x2q*>(Bs + (valid ? ry * R2_RP + rx : 10 * R2_RP) * R2_P + cbyte) = o;
    const bool own = inimg && ry >= 3 && ry < 7 && rx >= 3 && rx < 7;
    __builtin_amdgcn_raw_buffer_store_b64(o, rsO0, (int)(own ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + cbyte) : R2_OOB), 0, 0);
  });
  __syncthreads();
  // level 2: a1 (8x8, origin y0-2) = x + conv(h1) + b1, IN PLACE at A's centre; zero outside the image; own pixels (2..5) -> o[1]
  const auto rsO1 = __builtin_amdgcn_make_buffer_rsrc(p.o[1], 0, p.o[1] ? (int)p.bytes : 0, 0x00020000);
  run_level(std::integral_constant<int, 1>{}, Bs, w1, [&](int t, int ry, int rx, bool valid, const f32x4& a) {
    const int gy = y0 - 2 + ry, gx = x0 - 2 + rx;
    const bool inimg = valid && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    unsigned char* pos = As + ((2 + ry) * R2_RP + 2 + rx) * R2_P + cbyte;          // valid for every lane (8x8 = four full tiles)
    float s[4], v[4];
    unpack(*reinterpret_cast<const u32x2q*>(pos), s);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = a[r] + bv[1][r] + s[r];
    u32x2q o = pack(v);
    if (!inimg) o = u32x2q{0u, 0u};
    *reinterpret_cast<u32x2q*>(pos) = o;
    const bool own = inimg && ry >= 2 && ry < 6 && rx >= 2 && rx < 6;
    __builtin_amdgcn_raw_buffer_store_b64(o, rsO1, (int)(own ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + cbyte) : R2_OOB), 0, 0);
  });
  __syncthreads();
  // level 3: h2 (6x6, origin y0-1) = act(conv(a1) + b2) [* mask] from A's centre (offset 2,2) -> B; own pixels (1..4) -> o[2]
  const auto rsO2 = __builtin_amdgcn_make_buffer_rsrc(p.o[2], 0, p.o[2] ? (int)p.bytes : 0, 0x00020000);
  run_level(std::integral_constant<int, 2>{}, As + (2 * R2_RP + 2) * R2_P, w2, [&](int t, int ry, int rx, bool valid, const f32x4& a) {
    const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
    const bool inimg = valid && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = a[r] + bv[2][r];
      v[r] = fmaxf(v[r], v[r] * p.nslope);
    }
    if constexpr (HAS_A3) {
      float m[4];
      unpack(m3[t], m);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= m[r] > 0.f ? 1.f : 0.f;
    }
    u32x2q o = pack(v);
    if (!inimg) o = u32x2q{0u, 0u};
    *reinterpret_cast<u32x2q*>(Bs + (valid ? ry * R2_RP + rx : 10 * R2_RP) * R2_P + cbyte) = o;
    const bool own = inimg && ry >= 1 && ry < 5 && rx >= 1 && rx < 5;
    __builtin_amdgcn_raw_buffer_store_b64(o, rsO2, (int)(own ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + cbyte) : R2_OOB), 0, 0);
  });
  __syncthreads();
  // level 4: out (4x4) = a1 + conv(h2) + b3 [* mask]; the skip is a1 at A (4 + ry, 4 + rx)
  const auto rsO3 = __builtin_amdgcn_make_buffer_rsrc(p.o[3], 0, (int)p.bytes, 0x00020000);
  run_level(std::integral_constant<int, 3>{}, Bs, w3, [&](int t, int ry, int rx, bool valid, const f32x4& a) {
    const int gy = y0 + ry, gx = x0 + rx;
    float s[4], v[4];
    unpack(*reinterpret_cast<const u32x2q*>(As + ((4 + ry) * R2_RP + 4 + rx) * R2_P + cbyte), s);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = a[r] + bv[3][r] + s[r];
    if constexpr (HAS_A4) {
      float m[4];
      unpack(m4, m);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= m[r] > 0.f ? 1.f : 0.f;
    }
    const bool ok = gy < p.H && gx < p.W;
    __builtin_amdgcn_raw_buffer_store_b64(pack(v), rsO3, (int)(ok ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + cbyte) : R2_OOB), 0, 0);
  });
#undef R2_WISSUE
}

// Two residual blocks (mode 0: o0 = relu(conv(x, w0) + b0), o1 = x + conv(o0, w1) + b1, o2 = relu(conv(o1, w2) + b2), o3 = o1 +
// conv(o2, w3) + b3) or the input-gradient chain of two blocks (mode 1: o0 = convT(x, w0) * (aux1 > 0), o1 = x + convT(o0, w1),
// o2 = convT(o1, w2) * (aux3 > 0), o3 = (o1 + convT(o2, w3)) [* (aux4 > 0)]; b = NULL).  All weights in FRAGMENT order
// (tg_pack_weights_frag); o0..o2 nullable.  bf16, 64 channels.
extern "C" int tg_resblock2(int mode, const void* x, const void* const* w4, const float* const* b4, const void* aux1, const void* aux3,
                            const void* aux4, void* const* o4, int N, int H, int W, int C, int dtype, void* stream) {
  TG_CHECK_ARG(mode == 0 || mode == 1, "mode must be 0 (forward) or 1 (input gradient)");
  TG_CHECK_ARG(dtype == TG_BF16 && C == 64, "bf16 tensors with 64 channels only");
  TG_CHECK_ARG(x && w4 && o4 && w4[0] && w4[1] && w4[2] && w4[3] && o4[3] && N > 0 && H > 0 && W > 0, "null pointer / empty tensor");
  const int64_t bytes = (int64_t)N * H * W * 128;
  TG_CHECK_ARG(bytes < ((int64_t)1 << 31), "tensor too large for 32-bit buffer offsets");
  Rb2P p;
  p.x = x;
  for (int l = 0; l < 4; ++l) {
    p.w[l] = w4[l];
    p.b[l] = b4 ? b4[l] : nullptr;
    p.o[l] = o4[l];
    TG_CHECK_ARG((((uintptr_t)w4[l] | (uintptr_t)o4[l]) & 15) == 0, "pointers must be 16-byte aligned");
  }
  TG_CHECK_ARG((((uintptr_t)x | (uintptr_t)aux1 | (uintptr_t)aux3 | (uintptr_t)aux4) & 15) == 0, "pointers must be 16-byte aligned");
  p.aux1 = aux1; p.aux3 = aux3; p.aux4 = aux4;
  p.N = N; p.H = H; p.W = W;
  p.flip = mode;
  p.nslope = mode == 0 ? 0.f : 1.f;
  p.tiles_y = (H + 3) / 4; p.tiles_x = (W + 3) / 4;
  const int64_t nt = (int64_t)N * p.tiles_y * p.tiles_x;
  TG_CHECK_ARG(nt < ((int64_t)1 << 24), "too many tiles: this is the latency-regime kernel");
  p.ntiles = (int)nt;
  p.bytes = (unsigned)bytes;
  static const int prio = getenv("TG_C3_PRIO") ? atoi(getenv("TG_C3_PRIO")) : 1;
  p.prio = prio;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double px = (double)N * H * W;
  const double fl = 4.0 * 2.0 * px * 64.0 * 576.0, by = px * 128.0 * (5 + (aux1 != nullptr) + (aux3 != nullptr) + (aux4 != nullptr)) + 4.0 * 73728.0;
  TG_CHECK_ARG((aux1 != nullptr) == (aux3 != nullptr), "aux1 and aux3 come together (the input-gradient form)");
  if (aux1 && aux4) TG_LAUNCH("resblock2_lat<bwd,mask4>", fl, by, (resblock2_lat_kernel<true, true, true>), dim3(p.ntiles), dim3(256), 0, st, p);
  else if (aux1) TG_LAUNCH("resblock2_lat<bwd>", fl, by, (resblock2_lat_kernel<true, true, false>), dim3(p.ntiles), dim3(256), 0, st, p);
  else TG_LAUNCH("resblock2_lat<fwd>", fl, by, (resblock2_lat_kernel<false, false, false>), dim3(p.ntiles), dim3(256), 0, st, p);
  TG_CHECK_LAUNCH();
}
