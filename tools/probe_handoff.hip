// Probe of the workgroup -> workgroup hand-off latency on gfx950, the quantity that decides whether a kernel boundary of the
// recurrent chain can be replaced by an in-launch exchange (csrc/resblock_chain.hip).  A ping-pong between two workgroups over
// two 16-byte tagged granules (the form the chain kernel uses: the data is the flag), 2000 round trips, timed with the shader
// clock and the 100 MHz wall clock of workgroup A; one-way latency = round trip / 2 (poll turnaround included).  Pairs: the
// partner on the same XCD (block 8, if block b runs on XCD b % 8) or on another (block 1); the XCC ids are read from the hardware
// register and printed.  Store / load flavours:
//   sc1 / sc1        write-through store, L1-bypassing load: correct at any placement (the chain kernel's form)
//   plain / sc1      plain store (write-through L1, the line stays in the XCD's L2), L1-bypassing load: same-XCD only
//   sc0sc1 / sc0sc1  system scope both sides
// "loaded": the other 254 workgroups stream a buffer meanwhile (every CU's memory queue busy).
//   build:  hipcc --offload-arch=gfx950 -O2 tools/probe_handoff.hip -o tools/_trace/probe_handoff      run (GPU): tools/_trace/probe_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__device__ __forceinline__ void st16(const __amdgpu_buffer_rsrc_t& rs, unsigned off, unsigned tag) {
  __builtin_amdgcn_raw_buffer_store_b128(u32x4{tag ^ 0x5a5a5a5au, tag, tag ^ 0xa5a5a5a5u, tag}, rs, (int)off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ bool wait16(const __amdgpu_buffer_rsrc_t& rs, unsigned off, unsigned tag, unsigned& polls) {
  for (unsigned spins = 0; spins < (1u << 20); ++spins) {
    asm volatile("" ::: "memory");
    const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, AUX);
    ++polls;
    if (g.y == tag && g.w == tag) return true;
  }
  return false;
}

struct Out {
  unsigned long long cycles, wall, polls;
  unsigned xcc_a, xcc_b, ok, pad;
};

// blocks: 0 = A, partner = B, everyone else: idle or streaming load until A raises `stop`
template <int ST, int LD>
__global__ __launch_bounds__(64) void pingpong(unsigned char* slots, Out* out, int partner, int iters, unsigned epoch0, int load,
                                               const u32x4* stream_src, unsigned stream_n, unsigned* stop, u32x4* sink) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15;      // HW_REG_XCC_ID = 20, bits 3:0
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(slots, 0, 4096, 0x00020000);
  if (b == 0) {
    unsigned polls = 0;
    bool ok = true;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters && ok; ++i) {
      if (lane == 0) {
        st16<ST>(rs, 0, epoch0 + i + 1);
        ok = wait16<LD>(rs, 1024, epoch0 + i + 1, polls);
      }
      ok = __shfl(ok ? 1 : 0, 0) != 0;
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (lane == 0) {
      out->cycles = c1 - c0; out->wall = w1 - w0; out->polls = polls; out->xcc_a = xcc; out->ok = ok;
      __hip_atomic_store(stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (b == partner) {
    unsigned polls = 0;
    bool ok = true;
    for (int i = 0; i < iters && ok; ++i) {
      if (lane == 0) {
        ok = wait16<LD>(rs, 0, epoch0 + i + 1, polls);
        st16<ST>(rs, 1024, epoch0 + i + 1);
      }
      ok = __shfl(ok ? 1 : 0, 0) != 0;
    }
    if (lane == 0) out->xcc_b = xcc;
  } else if (load) {
    // streaming load: 8 x 16-byte loads in flight per lane until A is done (bounded)
    u32x4 acc = u32x4{0, 0, 0, 0};
    unsigned idx = (unsigned)(b * 64 + lane);
    for (int r = 0; r < 200000; ++r) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc ^= stream_src[idx % stream_n];
        idx += 256u * 64u;
      }
      if ((r & 15) == 0 && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    if (acc.x == 0x12345678u) sink[b * 64 + lane] = acc;
  }
}

int main() {
  unsigned char* slots;
  Out* out;
  unsigned* stop;
  u32x4 *src, *sink;
  const unsigned stream_n = 64u << 20 >> 4;          // 64 MB
  if (hipMalloc(&slots, 4096) != hipSuccess) { printf("no device\n"); return 1; }
  hipMalloc(&out, sizeof(Out)); hipMalloc(&stop, 4); hipMalloc(&src, (size_t)stream_n * 16); hipMalloc(&sink, 256 * 64 * 16);
  hipMemset(slots, 0, 4096); hipMemset(src, 1, (size_t)stream_n * 16);
  const int iters = 2000;
  unsigned epoch = 0;
  printf("workgroup -> workgroup hand-off, 16-byte tagged granule ping-pong, %d round trips; one-way = round trip / 2\n", iters);
  printf("%-18s %-10s %-8s %9s %9s %9s %7s  xcc\n", "store / load", "pair", "chip", "cycles", "ns", "polls/hop", "ok");
  for (int load = 0; load < 2; ++load)
    for (int pair = 0; pair < 2; ++pair)
      for (int flav = 0; flav < 3; ++flav) {
        const int partner = pair == 0 ? 8 : 1;
        hipMemset(out, 0, sizeof(Out)); hipMemset(stop, 0, 4);
        void (*k)(unsigned char*, Out*, int, int, unsigned, int, const u32x4*, unsigned, unsigned*, u32x4*) =
            flav == 0 ? pingpong<16, 16> : flav == 1 ? pingpong<0, 16> : pingpong<17, 17>;
        hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, slots, out, partner, iters, epoch, load, src, stream_n, stop, sink);
        hipDeviceSynchronize();
        epoch += iters + 8;
        Out h;
        hipMemcpy(&h, out, sizeof(Out), hipMemcpyDeviceToHost);
        printf("%-18s %-10s %-8s %9.0f %9.0f %9.2f %7u  %u -> %u\n", flav == 0 ? "sc1 / sc1" : flav == 1 ? "plain / sc1" : "sc0sc1 / sc0sc1",
               pair == 0 ? "blocks 0,8" : "blocks 0,1", load ? "loaded" : "idle", (double)h.cycles / iters / 2, (double)h.wall * 10.0 / iters / 2,
               (double)h.polls / iters, h.ok, h.xcc_a, h.xcc_b);
        fflush(stdout);
      }
  return 0;
}
