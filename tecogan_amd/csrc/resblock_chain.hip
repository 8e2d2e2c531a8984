// The residual TRUNK of generator_F -- `for i in range(1, num_resblock + 1): net = residual_block(net, 64, 1, ...)`, reference
// lib/frvsr.py:50-57,66-70 -- of one frame, or the input-gradient chain of the same blocks (tf.gradients, lib/Teco.py:441-449),
// as ONE persistent launch for the LATENCY regime of the training recurrence ([B,32,32,64]: 256 tiles of 4x4 pixels).
//
// Why.  csrc/resblock_lat.hip runs a block as one launch: 4.5 us per block in a graph chain, of which ~0.3 us is matrix work --
// the rest is the kernel boundary (1.45 us), a cold round trip for the halo tile and the 147 KB per-CU weight stream that only
// starts once the launch has.  19 frames x 16 blocks x 2 directions = 608 such nodes are 58 % of the TecoGAN step.  Here the
// boundary is replaced by a NEIGHBOUR hand-off inside the launch:
//   * workgroup = tile, as in resblock_lat.hip (same levels, same MFMA order: results are BIT-IDENTICAL to nb x tg_resblock);
//     a workgroup keeps its tile for all nb blocks: the block output goes to the centre of the LDS input region in place;
//   * what a block needs from outside is the 2-pixel ring around the tile (48 pixels x 128 B = 6 KB) from up to 8 neighbour
//     workgroups.  Every workgroup PUBLISHES its 4x4 output as 8-byte {two bf16 values, tag} granules with write-through (sc1)
//     stores -- a lane's four channels are one 16-byte store {v01, tag, v23, tag} -- into a two-slot ring (slot = block parity),
//     and SWEEPS the ring pixels of its neighbours with sc1 loads until every tag equals the expected epoch: the data is the
//     flag (cdna_hip_programming.md section 6, Guideline 16, form R2): no fence, no flag word, no L1/L2 maintenance;
//   * write-through stores are what the fabric is slow at (1 MB of them per block over the chip took ~1 us, session D), and a
//     workgroup's neighbours mostly sit on ITS OWN XCD (an XCD owns a contiguous range of tiles): so there are two rings -- P,
//     written with PLAIN stores (they reach the XCD's L2, which is the coherence point of its CUs: an L1-bypassing load of a
//     workgroup on the same XCD sees them, 237 ns against 334 in tools/probe_handoff.hip) and read by same-XCD neighbours only,
//     and S, written through (sc1) by exactly the lanes whose pixel a neighbour on ANOTHER XCD needs.  Which neighbour is where
//     is not assumed: every workgroup publishes its hardware XCC id in a tagged word at launch and reads its neighbours';
//   * two slots suffice: a workgroup writes block k + 1 into slot (k + 1) & 1 = (k - 1) & 1 only after it has read block k
//     of ALL its neighbours, each of which produced block k only after reading this workgroup's block k - 1;
//   * tags are monotonic ACROSS launches (tag = epoch base + block index + 1; the base lives in device memory and is advanced
//     by the last workgroup to arrive at the end of a launch), so the ring is never cleared and a captured graph replays;
//   * the weight stream does not depend on the hand-off: it is ONE stream over all blocks in consumption order with a prefetch
//     distance, so block k + 1's first fragments fly while block k's outputs travel;
//   * every spin is bounded: a workgroup that gives up counts itself in ctrl[2], stops waiting and runs on (garbage, no hang).
// Requires all workgroups co-resident: ntiles <= number of CUs (checked on the host; one workgroup always fits beside the
// capped side-stream kernels, and those terminate on their own).
#include "common.h"
#include "handoff.h"
#include <stdlib.h>

#define RC_MAXB 16

struct RcP {
  const void* x;                // [N,H,W,64] bf16: input of the first block processed
  const void* w1[RC_MAXB];      // per block, in processing order: fragment-order weights of the FIRST conv applied
  const void* w2[RC_MAXB];      // ... of the second
  const float* b1[RC_MAXB];     // nullable
  const float* b2[RC_MAXB];     // nullable
  const void* aux1[RC_MAXB];    // backward: the saved relu(conv_1) output of the block (level-1 mask)
  void* mid[RC_MAXB];           // level-1 result (nullable per block)
  void* out[RC_MAXB];           // block output
  const void* pre_x;            // forward only, nullable: [N,H,W,pre_cpad] bf16 -- the generator input; the input-stage conv + ReLU
  const void* pre_w;            //   (lib/frvsr.py:60-63) runs in this launch, on the 8x8 region the first block needs (no exchange)
  const float* pre_b;           //   pre_w: fragment-order [tap][64][64-padded Cin] copy, pre_out: [N,H,W,64] its output (stored: the
  void* pre_out;                //   weight gradients and the BPTT's mask need it); x is ignored then
  int pre_cpad;
  const void* aux2;             // nullable: mask on the LAST block's output (backward: the ReLU of the input stage)
  unsigned* ctrl;               // [0] epoch base  [1] arrivals  [2] give-ups (sticky)
  unsigned long long* xccw;     // [tiles] {tag, XCC id} words
  void* gran;                   // granule rings P (plain stores), S (write-through): 2 x 2 slots x [tiles] x 4 KB (rc_ring_off)
  int nb, N, H, W, flip;
  float nslope1;
  int tiles_y, tiles_x, ntiles;
  unsigned bytes;               // extent of every [N,H,W,64] tensor
  unsigned gslot;               // bytes of one ring slot = tiles * 4096
  unsigned spin_limit;
  int prio;
  int noweights;                // trace builds: zero-length weight descriptors (timing without the weight stream; results are wrong)
};

namespace {
constexpr int RC_P = 160;                       // bytes per LDS position (64 bf16 + 32 pad): resblock_lat.hip's layout
constexpr int RC_XR = 8, RC_XPOS = 8 * 8 + 2;
constexpr int RC_HR = 12, RC_HPOS = 6 * 12;
constexpr int RC_HL = 6;                        // level-2 LDS fragment look-ahead
}  // namespace

#ifdef TG_RC_TRACE
// [wave 0..7][block 0..15][stamp 0..7] of the middle workgroup
__device__ unsigned long long tg_rc_trace_buf[8 * 16 * 8];
#define RC_STAMP(k, i)                                                                                                  \
  do {                                                                                                                  \
    if (blockIdx.x == gridDim.x / 2 && lane == 0) tg_rc_trace_buf[(wave * 16 + (k)) * 8 + (i)] = (unsigned long long)clock64(); \
  } while (0)
#define RC_STAT(k) ((blockIdx.x == gridDim.x / 2 && lane == 0) ? &tg_rc_trace_buf[(wave * 16 + (k)) * 8 + 7] : nullptr)
extern "C" int tg_debug_rc_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_rc_trace_buf), sizeof(unsigned long long) * 8 * 16 * 8);
}
#else
#define RC_STAMP(k, i) do { } while (0)
#define RC_STAT(k) nullptr
#endif

// The ring of the 8x8 region around a 4x4 tile: 48 positions.  hi -> (row, column) of the region.
__device__ __forceinline__ void rc_ring_pos(int hi, int& ry, int& rx) {
  if (hi < 16) {
    ry = hi >> 3; rx = hi & 7;
  } else if (hi < 32) {
    const int j = hi - 16, c = j & 3;
    ry = 2 + (j >> 2); rx = c < 2 ? c : c + 4;
  } else {
    const int j = hi - 32;
    ry = 6 + (j >> 3); rx = j & 7;
  }
}

// Ring layout (per slot): [tile][wave 4][pixel 16][fg 4] x 16 B -- a publishing wave's store instruction covers ONE contiguous
// KiB (8 whole cache lines; as [pixel][channel] it was 16 half lines per instruction, and a write-through store of part of a line
// is what the fabric is slow at).  Offset of chunk c (= 4 channels: wave c / 4, fg c % 4) of image pixel (gy, gx):
__device__ __forceinline__ unsigned rc_ring_off(int n, int gy, int gx, int c, int tiles_y, int tiles_x) {
  const int t = (n * tiles_y + (gy >> 2)) * tiles_x + (gx >> 2);
  return (unsigned)(t * 4096 + (c >> 2) * 1024 + ((gy & 3) * 4 + (gx & 3)) * 64 + (c & 3) * 16);
}

// HAS_AUX1: level-1 mask (the input-gradient form); DIST: prefetch distance of the weight stream in fragments (36 per block).
// (A fifth wave that only sweeps -- its own memory queue -- was measured and lost: polling from barrier A on, 2 - 4 polls per
//  sweep, every block 3 - 6 % slower, profiles/r06a_mb_chain.txt.)
// TR (trace builds): 1 = the weight loads are not even issued (what the matrix phases cost without the stream; results are wrong)
// PRE: the input-stage conv of generator_F in front of the first block (forward launches of the training recurrence)
template <bool HAS_AUX1, int DIST, int TR = 0, int SM = 0, bool PRE = false>
__global__ __launch_bounds__(256, 2) void resblock_chain_kernel(RcP p) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[RC_XPOS * RC_P];
  __shared__ __attribute__((aligned(16))) unsigned char hs[RC_HPOS * RC_P];
  __shared__ __attribute__((aligned(16))) unsigned char ps_[PRE ? 100 * RC_P : 16];        // 10x10 region of the generator input
  // DIST == 0: ALL of a block's weight loads are issued in level 1 -- step s requests the SECOND conv's fragment s of this block
  // and, once its own fragment has been used, the FIRST conv's fragment s of the next block -- and none in level 2, a chain of
  // 18 dependent MFMAs that hides nothing (1400 -> 650 cycles without loads, profiles/r06k_trace_chain.txt).  Measured: the
  // time moves with the loads (level 1 1900 -> 2500), 6.6k vs 6.7k cycles per block (r06l_trace_chain.txt): kept as a variant.
  static_assert(DIST >= 0 && DIST <= 35, "prefetch distance in fragments");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fg = lane >> 4;

  int b = blockIdx.x;
  if ((p.ntiles & 7) == 0) b = (b & 7) * (p.ntiles >> 3) + (b >> 3);      // an XCD owns a contiguous range of tiles (resblock_lat.hip)
  const int tx = b % p.tiles_x, t1 = b / p.tiles_x;
  const int ty = t1 % p.tiles_y, n = t1 / p.tiles_y;
  const int y0 = ty * 4, x0 = tx * 4;
  const int nb = p.nb;

  // epoch base of this launch (every workgroup reads it before any workgroup can have arrived at the end)
  const unsigned epoch0 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  const auto rsG = __builtin_amdgcn_make_buffer_rsrc(p.gran, 0, (int)(4u * p.gslot), 0x00020000);
  const unsigned my_xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;   // HW_REG_XCC_ID[3:0]: the XCD this workgroup runs on
  if (nb > 1 && tid == 0)
    __hip_atomic_store(p.xccw + b, ((unsigned long long)(epoch0 + 1u) << 32) | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned limit = p.spin_limit;                                            // 0 once this workgroup has given up

  const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.bytes, 0x00020000);
  const int cbyte = (wave * 16 + fg * 4) * 2;       // byte offset of this lane's four output channels inside a pixel

  // ---- loads that do not depend on the block ----------------------------------------------------------------------------------
  u32x4c xr[2];
  if constexpr (!PRE) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int item = tid + k * 256;
      const int pix = item >> 3, ch = item & 7;
      const int gy = y0 - 2 + (pix >> 3), gx = x0 - 2 + (pix & 7);
      const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(ok ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + ch * 16) : RC_OOB), 0, 0);
    }
  }
  // PRE: the 10x10 region of the generator input (pre_cpad channels: the chunks beyond them read zeros), 800 16-byte items, and the
  // input conv's 18 weight fragments -- requested BEFORE the trunk's weight stream starts (they are consumed first)
  u32x4c pr[4], wP[PRE ? 18 : 1];
  u32x4c bqP = u32x4c{0u, 0u, 0u, 0u};
  if constexpr (PRE) {
    const auto rsPX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.pre_x), 0, (int)((unsigned)(p.N * p.H * p.W) * (unsigned)p.pre_cpad * 2u), 0x00020000);
    const int pchunks = p.pre_cpad >> 3;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int item = tid + k * 256;
      const int pix = item >> 3, ch = item & 7;
      const int gy = y0 - 3 + pix / 10, gx = x0 - 3 + pix % 10;
      const bool ok = item < 800 && ch < pchunks && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      pr[k] = __builtin_amdgcn_raw_buffer_load_b128(rsPX, (int)(ok ? (unsigned)(((n * p.H + gy) * p.W + gx) * p.pre_cpad * 2 + ch * 16) : RC_OOB), 0, 0);
    }
    const auto rsPW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.pre_w), 0, 9 * 64 * 64 * 2, 0x00020000);
#pragma unroll
    for (int s2 = 0; s2 < 18; ++s2) wP[s2] = __builtin_amdgcn_raw_buffer_load_b128(rsPW, wave * 1024 + lane * 16, s2 * 4096, 0);
    bqP = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.pre_b), 0, p.pre_b ? 256 : 0, 0x00020000),
                                                (wave * 16 + fg * 4) * 4, 0, 0);
  }
  const int oy = y0 + (frow >> 2), ox = x0 + (frow & 3);
  const bool out_ok = oy < p.H && ox < p.W;
  const unsigned out_off = out_ok ? (unsigned)(((n * p.H + oy) * p.W + ox) * 128 + cbyte) : RC_OOB;
  const unsigned pub_off = out_ok ? rc_ring_off(n, oy, ox, wave * 4 + fg, p.tiles_y, p.tiles_x) : RC_OOB;
  unsigned m1off[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int ry = 2 * t + (frow >> 3), rx = frow & 7;
    const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
    const bool ok = rx < 6 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    m1off[t] = ok ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + cbyte) : RC_OOB;
  }
  // the ring items of this lane (3 per lane) and the tile that owns each
  unsigned goff[3], xoff[6];
  int lpos[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int item = tid + k * 256;
    int ry, rx;
    rc_ring_pos(item >> 4, ry, rx);
    const int gy = y0 - 2 + ry, gx = x0 - 2 + rx;
    const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    goff[k] = ok ? rc_ring_off(n, gy, gx, item & 15, p.tiles_y, p.tiles_x) : RC_OOB;
    xoff[k] = ok ? (unsigned)(((n * p.tiles_y + (gy >> 2)) * p.tiles_x + (gx >> 2)) * 8) : RC_OOB;
    lpos[k] = (ry * RC_XR + rx) * RC_P + (item & 15) * 8;
  }
  // the three neighbour tiles that need this lane's pixel (its quadrant of the tile: vertical, horizontal, diagonal neighbour)
  {
    const int dy = (frow >> 2) < 2 ? -1 : 1, dx = (frow & 3) < 2 ? -1 : 1;
    const bool vy = (unsigned)(ty + dy) < (unsigned)p.tiles_y, vx = (unsigned)(tx + dx) < (unsigned)p.tiles_x;
    xoff[3] = vy ? (unsigned)(((n * p.tiles_y + ty + dy) * p.tiles_x + tx) * 8) : RC_OOB;
    xoff[4] = vx ? (unsigned)(((n * p.tiles_y + ty) * p.tiles_x + tx + dx) * 8) : RC_OOB;
    xoff[5] = vy && vx ? (unsigned)(((n * p.tiles_y + ty + dy) * p.tiles_x + tx + dx) * 8) : RC_OOB;
  }

  // ---- per-block operands: descriptors of block k (cur) and k + 1 (nxt: the weight stream runs ahead) ------------------------
  auto rsrc_w = [&](const void* w) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, w && !p.noweights ? 9 * 64 * 64 * 2 : 0, 0x00020000); };
  auto rsrc_b = [&](const float* q) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q), 0, q ? 256 : 0, 0x00020000); };
  auto rsrc_t = [&](const void* q) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q), 0, q ? (int)p.bytes : 0, 0x00020000); };
  auto rsW1 = rsrc_w(p.w1[0]), rsW2 = rsrc_w(p.w2[0]);
  auto rsW1n = rsrc_w(nb > 1 ? p.w1[1] : nullptr), rsW2n = rsrc_w(nb > 1 ? p.w2[1] : nullptr);

  // weight stream: position i of a block: i < 18 step i of the first conv (wA[i]), else step i - 18 of the second (wB[i - 18]);
  // positions 36 .. 36 + 35 are the NEXT block's (same registers: a slot is re-requested DIST - 36 positions after its last use)
  u32x4c wA[18], wB[18];
  if constexpr (TR >= 1) {
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      wA[i] = wB[i] = u32x4c{(unsigned)lane, 0x3c003c00u, 0x3c003c00u, (unsigned)i};
      asm volatile("" : "+v"(wA[i]), "+v"(wB[i]));
    }
  }
  const int wlane = wave * 1024 + lane * 16;
  auto wload = [&](const auto& rs, int s) {
    const int tap = s >> 1, kk = s & 1;
    const int wtap = p.flip ? 8 - tap : tap;
    return __builtin_amdgcn_raw_buffer_load_b128(rs, wlane, (wtap * 2 + kk) * 4096, 0);
  };
#define RC_WISSUE(i)                                                                                   \
  do {                                                                                                 \
    if constexpr (TR >= 1) break;                                                                      \
    if constexpr ((i) < 18) wA[(i) < 18 ? (i) : 0] = wload(rsW1, (i));                                 \
    else if constexpr ((i) < 36) wB[(i) >= 18 && (i) < 36 ? (i) - 18 : 0] = wload(rsW2, (i) - 18);     \
    else if constexpr ((i) < 54) wA[(i) >= 36 && (i) < 54 ? (i) - 36 : 0] = wload(rsW1n, (i) - 36);    \
    else wB[(i) >= 54 && (i) < 72 ? (i) - 54 : 0] = wload(rsW2n, (i) - 54);                            \
  } while (0)
  rc_static_for<0, (DIST == 0 ? 18 : DIST)>([&](auto i) { RC_WISSUE(decltype(i)::value); });

  // biases and level-1 masks of the first block (those of block k + 1 are requested at the start of level 2 of block k)
  u32x4c bq1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b(p.b1[0]), (wave * 16 + fg * 4) * 4, 0, 0);
  u32x4c bq2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b(p.b2[0]), (wave * 16 + fg * 4) * 4, 0, 0);
  u32x2c m1[3];
  if constexpr (HAS_AUX1) {
    const auto rsA1 = rsrc_t(p.aux1[0]);
#pragma unroll
    for (int t = 0; t < 3; ++t) m1[t] = __builtin_amdgcn_raw_buffer_load_b64(rsA1, (int)m1off[t], 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- who is where: the neighbours' XCC ids (published at their launch; one sweep per launch, behind the first loads) ---------
  unsigned pubS_off = RC_OOB;                      // write-through copy of this lane's granule: only if a cross-XCD neighbour needs it
  if (nb > 1) {
    const auto rsXW = __builtin_amdgcn_make_buffer_rsrc(p.xccw, 0, p.ntiles * 8, 0x00020000);
    u32x2c xw[6];
    bool got = false;
    const unsigned lim = __builtin_amdgcn_readfirstlane(limit);
    for (unsigned spins = 0; spins <= lim; ++spins) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < 6; ++j) xw[j] = __builtin_amdgcn_raw_buffer_load_b64(rsXW, (int)xoff[j], 0, RC_SC1);
      unsigned bad = 0;
#pragma unroll
      for (int j = 0; j < 6; ++j) bad |= (xoff[j] != RC_OOB ? 0xffffffffu : 0u) & (xw[j].y ^ (epoch0 + 1u));
      if (!__any(bad != 0)) { got = true; break; }
      if (spins > 32) __builtin_amdgcn_s_sleep(32);
      else __builtin_amdgcn_s_sleep(1);
    }
    if (!got) {
      limit = 0;
      if (lane == 0) __hip_atomic_fetch_add(p.ctrl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (goff[k] != RC_OOB && (xw[k].x & 15u) != my_xcc) goff[k] += 2u * p.gslot;                 // ring S
    bool needS = false;
#pragma unroll
    for (int j = 3; j < 6; ++j) needS = needS || (xoff[j] != RC_OOB && (xw[j].x & 15u) != my_xcc);
    if (needS) pubS_off = pub_off;
  }
  if (p.prio) __builtin_amdgcn_s_setprio(3);       // (the matrix phases; the sweeps drop to 0)

  // ---- input region of the first block -> LDS (positions outside the image hold zeros from here on) --------------------------
  if constexpr (!PRE) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int item = tid + k * 256;
      *reinterpret_cast<u32x4c*>(xs + (item >> 3) * RC_P + (item & 7) * 16) = xr[k];
    }
    __syncthreads();
  } else {
    // ... computed here: relu(conv3x3(generator input) + b) on the whole 8x8 region (four 2x8 pixel tiles; the ring is recomputed
    // from the 10x10 input region instead of exchanged: 72 MFMAs per wave against a kernel boundary), taps and K halves in
    // conv3x3_tile's order -> bit-identical to the input-stage launch it replaces
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int item = tid + k * 256;
      if (item < 800) *reinterpret_cast<u32x4c*>(ps_ + (item >> 3) * RC_P + (item & 7) * 16) = pr[k];
    }
    __syncthreads();
    const unsigned char* pb = ps_ + ((frow >> 3) * 10 + (frow & 7)) * RC_P + fg * 16;
    f32x4 accp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) accp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    rc_static_for<0, 18>([&](auto sv) {
      constexpr int s2 = decltype(sv)::value, tap = s2 >> 1, kk = s2 & 1;
      uint4 bfp[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bfp[t] = *reinterpret_cast<const uint4*>(pb + ((2 * t + tap / 3) * 10 + tap % 3) * RC_P + kk * 64);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        accp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wP[s2]), *reinterpret_cast<bf16x8*>(&bfp[t]),
                                                          accp[t], 0, 0, 0);
    });
    const auto rsPO = __builtin_amdgcn_make_buffer_rsrc(p.pre_out, 0, (int)p.bytes, 0x00020000);
    const float bvp[4] = {__uint_as_float(bqP.x), __uint_as_float(bqP.y), __uint_as_float(bqP.z), __uint_as_float(bqP.w)};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ry = 2 * t + (frow >> 3), rx = frow & 7;
      const int gy = y0 - 2 + ry, gx = x0 - 2 + rx;
      const bool inimg = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = accp[t][r] + bvp[r];
        v[r] = fmaxf(z, z * p.nslope1);             // (ReLU as conv3x3_tile writes it: -0 for a negative sum, bit for bit; nslope1 = 0 forward)
      }
      u32x2c o = rc_pack4(v);
      if (!inimg) o = u32x2c{0u, 0u};
      *reinterpret_cast<u32x2c*>(xs + (ry * RC_XR + rx) * RC_P + cbyte) = o;
      const bool own = inimg && ry >= 2 && ry <= 5 && rx >= 2 && rx <= 5;
      __builtin_amdgcn_raw_buffer_store_b64(o, rsPO, (int)(own ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + cbyte) : RC_OOB), 0, 0);
    }
    __syncthreads();
  }

  const unsigned char* xb = xs + ((frow >> 3) * RC_XR + (frow & 7)) * RC_P + fg * 16;
  const unsigned char* hb = hs + ((frow >> 2) * RC_HR + (frow & 3)) * RC_P + fg * 16;
  unsigned char* ctr = xs + (((frow >> 2) + 2) * RC_XR + (frow & 3) + 2) * RC_P + cbyte;     // this lane's element of the centre

  for (int k = 0; k < nb; ++k) {
    RC_STAMP(k, 0);
    const bool last = k + 1 >= nb;
    const float bv1[4] = {__uint_as_float(bq1.x), __uint_as_float(bq1.y), __uint_as_float(bq1.z), __uint_as_float(bq1.w)};
    const float bv2[4] = {__uint_as_float(bq2.x), __uint_as_float(bq2.y), __uint_as_float(bq2.z), __uint_as_float(bq2.w)};

    // ---- level 1: first conv on the 6x6 region, three 2x8 pixel tiles (resblock_lat.hip) --------------------------------------
    f32x4 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The three taps of a kernel row read the SAME LDS rows shifted by one / two columns, and a pixel tile's 16 lanes of one
    // K group are two region rows of 8 columns: the fragment of tap column dx is the fragment of column 0 shifted by dx lanes
    // inside its 16-lane row (DPP row_shl; lanes shifted in from the next tile row are what the LDS read at column 8, 9 -- the
    // next row's columns 0, 1 -- returned, lanes without a source belong to the padding columns 6, 7).  One LDS read per (kernel
    // row, K half, tile) instead of three: level 1 was bound by its 216 KB of fragment reads (12 KB per step against 128 B/clk).
    auto xbase = [&](int dy, int kk, int t) {
      return *reinterpret_cast<const uint4*>(xb + ((2 * t + dy) * RC_XR) * RC_P + kk * 64);
    };
    auto shl = [&](const uint4& v, auto dxv) {
      constexpr int dx = decltype(dxv)::value;
      if constexpr (dx == 0) return v;
      else {
        uint4 o;
        o.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x100 + dx, 0xf, 0xf, true);
        o.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x100 + dx, 0xf, 0xf, true);
        o.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x100 + dx, 0xf, 0xf, true);
        o.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x100 + dx, 0xf, 0xf, true);
        return o;
      }
    };
    uint4 base[2][3], nbase[2][3];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int t = 0; t < 3; ++t) base[kk][t] = xbase(0, kk, t);
    rc_static_for<0, 18>([&](auto sv) {
      constexpr int s = decltype(sv)::value;
      constexpr int tap = s >> 1, kk = s & 1, dy = tap / 3, dx = tap % 3;
      if constexpr (DIST == 0) RC_WISSUE(18 + s);
      else RC_WISSUE(s + DIST);
      if constexpr (dx == 0 && dy < 2 && TR != 2) {             // the next kernel row's fragments: a whole row of MFMAs ahead
#pragma unroll
        for (int t = 0; t < 3; ++t) nbase[kk][t] = xbase(dy + 1, kk, t);
      }
      uint4 bf[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) bf[t] = TR == 2 ? base[0][t] : shl(base[kk][t], std::integral_constant<int, dx>{});
#pragma unroll
      for (int t = 0; t < 3; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wA[s]), *reinterpret_cast<bf16x8*>(&bf[t]),
                                                         acc[t], 0, 0, 0);
      if constexpr (DIST == 0) RC_WISSUE(36 + s);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (dx == 2 && kk == 1 && dy < 2 && TR != 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int t = 0; t < 3; ++t) base[q][t] = nbase[q][t];
      }
    });
    RC_STAMP(k, 1);
    const auto rsM = __builtin_amdgcn_make_buffer_rsrc(p.mid[k], 0, p.mid[k] ? (int)p.bytes : 0, 0x00020000);
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc(p.out[k], 0, (int)p.bytes, 0x00020000);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int ry = 2 * t + (frow >> 3), rx = frow & 7;
      const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
      const bool inimg = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[t][r] + bv1[r];
        v[r] = fmaxf(v[r], v[r] * p.nslope1);
      }
      if constexpr (HAS_AUX1) {
        float a[4];
        rc_unpack4(m1[t], a);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= a[r] > 0.f ? 1.f : 0.f;
      }
      u32x2c o = rc_pack4(v);
      if (!inimg) o = u32x2c{0u, 0u};
      *reinterpret_cast<u32x2c*>(hs + (ry * RC_HR + rx) * RC_P + cbyte) = o;
      const bool own = inimg && ry >= 1 && ry <= 4 && rx >= 1 && rx <= 4;
      __builtin_amdgcn_raw_buffer_store_b64(o, rsM, (int)(own ? (unsigned)(((n * p.H + gy) * p.W + gx) * 128 + cbyte) : RC_OOB), 0, 0);
    }
    __syncthreads();                                                         // barrier A
    RC_STAMP(k, 2);

    // the next block's small operands (biases, level-1 masks; nothing for the last block: zero-length descriptors); this block's
    // were consumed above (bv1, m1) or copied (bv2)
    {
      const int kn = last ? k : k + 1;
      bq1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b(last ? nullptr : p.b1[kn]), (wave * 16 + fg * 4) * 4, 0, 0);
      bq2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b(last ? nullptr : p.b2[kn]), (wave * 16 + fg * 4) * 4, 0, 0);
      if constexpr (HAS_AUX1) {
        const auto rsA1n = rsrc_t(last ? nullptr : p.aux1[kn]);
#pragma unroll
        for (int t = 0; t < 3; ++t) m1[t] = __builtin_amdgcn_raw_buffer_load_b64(rsA1n, (int)m1off[t], 0, 0);
      }
    }
    // mask of the last block's output (zero-length descriptor otherwise: reads zeros, ignored below)
    const bool use_m2 = last && p.aux2 != nullptr;
    const u32x2c m2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc_t(use_m2 ? p.aux2 : nullptr), (int)out_off, 0, 0);

    // ---- level 2: second conv on the 4x4 tile, one accumulator, 18 dependent MFMAs --------------------------------------------
    f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto hfrag = [&](int s) {
      return *reinterpret_cast<const uint4*>(hb + ((s / 6) * RC_HR + (s >> 1) % 3) * RC_P + (s & 1) * 64);
    };
    uint4 hf[RC_HL];
#pragma unroll
    for (int s = 0; s < RC_HL; ++s) hf[s] = hfrag(s);
    rc_static_for<0, 18>([&](auto sv) {
      constexpr int s = decltype(sv)::value;
      if constexpr (DIST != 0) RC_WISSUE(18 + s + DIST);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wB[s]), *reinterpret_cast<bf16x8*>(&hf[s % RC_HL]),
                                                     acc2, 0, 0, 0);
      if constexpr (s + RC_HL < 18) hf[s % RC_HL] = hfrag(s + RC_HL);
      __builtin_amdgcn_sched_barrier(0);
    });
    RC_STAMP(k, 3);
    // level-2 epilogue: bias, skip (the centre of the region), mask, store; publish; the centre of the next block's region
    {
      const u32x2c sk = *reinterpret_cast<const u32x2c*>(ctr);
      float s[4], v[4];
      rc_unpack4(sk, s);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc2[r] + bv2[r];
        v[r] += s[r];
      }
      float a[4];
      rc_unpack4(m2, a);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= (a[r] > 0.f || !use_m2) ? 1.f : 0.f;
      const u32x2c o = rc_pack4(v);
      const unsigned tag = epoch0 + (unsigned)k + 1u;
      const u32x4c gr = u32x4c{o.x, tag, o.y, tag};
      // (the slot offsets in the VECTOR offset, soffset = 0: behind a 16-byte store with an SGPR soffset hipcc leaves out the wait
      //  state before the store's data registers are rewritten -- here the centre's v_cndmask one instruction later -- and gfx950
      //  needs it: csrc/resblock_plane.hip, DESIGN lesson 41)
      const unsigned slotP = (unsigned)(k & 1) * p.gslot, slotS = slotP + 2u * p.gslot;
      __builtin_amdgcn_raw_buffer_store_b128(gr, rsG, (int)(last || pubS_off == RC_OOB ? RC_OOB : pubS_off + slotS), 0, RC_SC1);
      __builtin_amdgcn_raw_buffer_store_b128(gr, rsG, (int)(last || pub_off == RC_OOB ? RC_OOB : pub_off + slotP), 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(o, rsO, (int)out_off, 0, 0);
      *reinterpret_cast<u32x2c*>(ctr) = out_ok ? o : u32x2c{0u, 0u};   // (a partial tile's pixels outside the image stay the next conv's zero padding)
    }
    // the weight descriptors move on: what was "next" is current, block k + 2 becomes next
    rsW1 = rsW1n; rsW2 = rsW2n;
    rsW1n = rsrc_w(k + 2 < nb ? p.w1[k + 2 < nb ? k + 2 : 0] : nullptr);
    rsW2n = rsrc_w(k + 2 < nb ? p.w2[k + 2 < nb ? k + 2 : 0] : nullptr);
    RC_STAMP(k, 4);

    // ---- hand-off: the ring of block k + 1's input region from the neighbours' block-k outputs --------------------------------
    if (!last && limit) {
      if (!rc_sweep<3, SM>(rsG, goff, lpos, xs, (unsigned)(k & 1) * p.gslot, epoch0 + (unsigned)k + 1u, limit, RC_STAT(k))) {
        limit = 0;
        if (lane == 0) __hip_atomic_fetch_add(p.ctrl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (p.prio) __builtin_amdgcn_s_setprio(3);
    }
    RC_STAMP(k, 5);
    __syncthreads();                                                         // barrier B
    RC_STAMP(k, 6);
  }
#undef RC_WISSUE

  // ---- arrival: the last workgroup advances the epoch base for the next launch -------------------------------------------------
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(p.ctrl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)p.ntiles - 1u) {
      __hip_atomic_store(p.ctrl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.ctrl, epoch0 + (unsigned)nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


// (Round 6, session Y: an EIGHT-wave form -- two waves per SIMD owning the same 16 output channels, every conv's reduction split
//  between them by K half, partial sums handed over through LDS -- was built on lesson 37 and measured: correct (1.2e-2 of the
//  tensor maximum from the per-block path after 16 blocks: two partial sums per conv), level 1 1700 -> 1060 and level 2 1360 ->
//  970 cycles per wave, but the two hand-over barriers cost ~1000 cycles each because the waves of a SIMD do not interleave --
//  the pair's second wave finishes its half as late as one wave finishes the whole -- 3.65 against 3.34 us per block
//  (profiles/r06y_trace_chain.txt).  Deleted; commit 34adc01 has the code.)

extern "C" int tg_resblock_chain_scratch_bytes(int N, int H, int W, int64_t* bytes) {
  TG_CHECK_ARG(bytes && N > 0 && H > 0 && W > 0, "null pointer / empty tensor");
  const int64_t nt = (int64_t)N * ((H + 3) / 4) * ((W + 3) / 4);
  *bytes = 256 + ((nt * 8 + 255) / 256) * 256 + 4 * nt * 4096;      // control words, XCC words, rings P and S (2 slots each)
  return TG_OK;
}

extern "C" int tg_resblock_chain(int mode, const void* x, int nblocks, const void* const* w1, const float* const* b1,
                                 const void* const* w2, const float* const* b2, const void* const* aux1, const void* aux2_last,
                                 void* const* mid, void* const* out, void* scratch, const void* pre_x, int pre_cpad,
                                 const void* pre_w_frag, const float* pre_b, void* pre_out, int N, int H, int W, int C, int dtype,
                                 int variant, void* stream) {
  TG_CHECK_ARG(mode == 0 || mode == 1, "mode must be 0 (forward) or 1 (input gradient)");
  TG_CHECK_ARG(dtype == TG_BF16 && C == 64, "bf16 tensors with 64 channels only");
  TG_CHECK_ARG(nblocks >= 1 && nblocks <= RC_MAXB, "1 .. 16 blocks per launch");
  TG_CHECK_ARG((x || pre_x) && w1 && w2 && out && scratch && N > 0 && H > 0 && W > 0, "null pointer / empty tensor");
  TG_CHECK_ARG(!pre_x || (mode == 0 && pre_w_frag && pre_out && pre_cpad >= 8 && pre_cpad <= 64 && pre_cpad % 8 == 0),
               "the input-stage conv in front of the trunk: forward launches, fragment-order weights, 8 .. 64 padded input channels");
  TG_CHECK_ARG((((uintptr_t)pre_x | (uintptr_t)pre_w_frag | (uintptr_t)pre_out) & 15) == 0, "pointers must be 16-byte aligned");
  TG_CHECK_ARG((mode == 1) == (aux1 != nullptr), "aux1 (the saved relu(conv_1) outputs) belongs to mode 1");
  TG_CHECK_ARG((((uintptr_t)x | (uintptr_t)scratch | (uintptr_t)aux2_last) & 15) == 0, "pointers must be 16-byte aligned");
  const int64_t bytes = (int64_t)N * H * W * 128;
  TG_CHECK_ARG(bytes * 4 < ((int64_t)1 << 31), "tensor too large for 32-bit buffer offsets");
  RcP p;
  p.x = x;
  for (int k = 0; k < RC_MAXB; ++k) {
    const bool on = k < nblocks;
    p.w1[k] = on ? w1[k] : nullptr; p.w2[k] = on ? w2[k] : nullptr;
    p.b1[k] = on && b1 ? b1[k] : nullptr; p.b2[k] = on && b2 ? b2[k] : nullptr;
    p.aux1[k] = on && aux1 ? aux1[k] : nullptr;
    p.mid[k] = on && mid ? mid[k] : nullptr; p.out[k] = on ? out[k] : nullptr;
    if (on) {
      TG_CHECK_ARG(p.w1[k] && p.w2[k] && p.out[k] && (mode == 0 || p.aux1[k]), "null per-block pointer");
      TG_CHECK_ARG((((uintptr_t)p.w1[k] | (uintptr_t)p.w2[k] | (uintptr_t)p.out[k] | (uintptr_t)p.mid[k] | (uintptr_t)p.aux1[k]) & 15) == 0,
                   "pointers must be 16-byte aligned");
    }
  }
  p.aux2 = aux2_last;
  p.pre_x = pre_x; p.pre_w = pre_w_frag; p.pre_b = pre_b; p.pre_out = pre_out; p.pre_cpad = pre_cpad;
  p.ctrl = static_cast<unsigned*>(scratch);
  p.nb = nblocks; p.N = N; p.H = H; p.W = W;
  p.flip = mode;
  p.nslope1 = mode == 0 ? 0.f : 1.f;
  p.tiles_y = (H + 3) / 4; p.tiles_x = (W + 3) / 4;
  const int64_t nt = (int64_t)N * p.tiles_y * p.tiles_x;
  TG_CHECK_ARG(nt <= tg_num_cus(), "more tiles than compute units: the hand-offs need every workgroup resident (use tg_resblock)");
  p.ntiles = (int)nt;
  p.bytes = (unsigned)bytes;
  p.gslot = (unsigned)(nt * 4096);
  p.xccw = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(scratch) + 256);
  p.gran = static_cast<unsigned char*>(scratch) + 256 + ((nt * 8 + 255) / 256) * 256;
  p.spin_limit = 1u << 16;                      // ~0.2 s of backed-off polling before a workgroup gives up
  p.prio = 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double px = (double)N * H * W;
  const double fl = 2.0 * 2.0 * px * 64.0 * 576.0 * nblocks + (pre_x ? 2.0 * px * 64.0 * 9.0 * pre_cpad : 0.0);
  const double by = nblocks * (px * 128.0 * (2 + (mid != nullptr) + (aux1 != nullptr)) + 2.0 * 73728.0) + (aux2_last ? px * 128.0 : 0.0);
  const int dist = (variant >> 1) & 63;          // 0: default
#ifdef TG_RC_TRACE
  p.noweights = (variant >> 8) & 1;
#else
  p.noweights = 0;
#endif
  auto go = [&](auto atag, auto dtag) {
    constexpr bool A = decltype(atag)::value;
    constexpr int D = decltype(dtag)::value;
    TG_LAUNCH(A ? "resblock_chain<bwd>" : "resblock_chain<fwd>", fl, by, (resblock_chain_kernel<A, D>), dim3(p.ntiles), dim3(256), 0, st, p);
  };
  using T = std::true_type;
  using Fa = std::false_type;
  auto pick = [&](auto atag) {
#ifdef TG_RC_TRACE
    if ((variant >> 11) & 3) {
      constexpr bool A = decltype(atag)::value;
      const int sm = ((variant >> 11) & 3) - 1;
      if (sm == 0) TG_LAUNCH("resblock_chain<sm3>", fl, by, (resblock_chain_kernel<A, 14, 0, 3>), dim3(p.ntiles), dim3(256), 0, st, p);
      else if (sm == 1) TG_LAUNCH("resblock_chain<sm1>", fl, by, (resblock_chain_kernel<A, 14, 0, 1>), dim3(p.ntiles), dim3(256), 0, st, p);
      else TG_LAUNCH("resblock_chain<sm2>", fl, by, (resblock_chain_kernel<A, 14, 0, 2>), dim3(p.ntiles), dim3(256), 0, st, p);
      return;
    }
    if ((variant >> 9) & 1) {
      constexpr bool A = decltype(atag)::value;
      if ((variant >> 10) & 1) TG_LAUNCH("resblock_chain<trace2>", fl, by, (resblock_chain_kernel<A, 14, 2>), dim3(p.ntiles), dim3(256), 0, st, p);
      else TG_LAUNCH("resblock_chain<trace>", fl, by, (resblock_chain_kernel<A, 14, 1>), dim3(p.ntiles), dim3(256), 0, st, p);
      return;
    }
#endif
    if constexpr (!decltype(atag)::value) {
      if (pre_x) {
        TG_LAUNCH("resblock_chain<fwd,in>", fl, by + px * (2.0 * pre_cpad + 128.0), (resblock_chain_kernel<false, 14, 0, 0, true>), dim3(p.ntiles),
                  dim3(256), 0, st, p);
        return;
      }
    }
    if (dist == 14) go(atag, std::integral_constant<int, 14>{});
    else if (dist == 63) go(atag, std::integral_constant<int, 0>{});
    else if (dist == 4) go(atag, std::integral_constant<int, 4>{});
    else if (dist == 35) go(atag, std::integral_constant<int, 35>{});
    else go(atag, std::integral_constant<int, 28>{});
  };
  if (mode == 1) pick(T{});
  else pick(Fa{});
  TG_CHECK_LAUNCH();
}
