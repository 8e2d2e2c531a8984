// Probe of gfx950's LDS transpose read (ds_read_b64_tr_b16): which 16-bit LDS elements does lane l receive, for a few
// per-lane address patterns?  Groundwork for a weight-gradient kernel whose K dimension (pixels) is the slow axis of both
// operands in memory: with the transpose read the [pixel][channel] tiles can stay in their natural layout in LDS.
//   build:  hipcc --offload-arch=gfx950 -O2 tools/probe_tr.hip -o tools/_trace/probe_tr      run (GPU): tools/_trace/probe_tr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// pattern 0: lane l -> byte address 8 l                     (64 consecutive 8-byte granules)
// pattern 1: lane l -> row (l % 16), 8-byte column (l / 16) of a [16][ROWB]-byte image   (ROWB = 32)
// pattern 2: lane l -> row (l / 4) % 4 + 4 * (l / 16), 8-byte column l % 4 of a [16][32]-byte image (4 keys x 16 cols per 16 lanes)
__global__ void probe(unsigned short* out, int pattern) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;      // element value = element index
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr;
  if (pattern == 0) addr = 8u * l;
  else if (pattern == 1) addr = (l % 16) * 32u + (l / 16) * 8u;
  else addr = (((l / 4) % 4) + 4 * (l / 16)) * 32u + (l % 4) * 8u;
  addr += (unsigned)(size_t)lds;                                               // LDS byte address
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = (unsigned short)(v.x & 0xffff);
  out[l * 4 + 1] = (unsigned short)(v.x >> 16);
  out[l * 4 + 2] = (unsigned short)(v.y & 0xffff);
  out[l * 4 + 3] = (unsigned short)(v.y >> 16);
}

int main() {
  unsigned short* d;
  if (hipMalloc(&d, 64 * 4 * 2) != hipSuccess) { printf("no device\n"); return 1; }
  for (int pat = 0; pat < 3; ++pat) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pat);
    std::vector<unsigned short> h(256);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    printf("pattern %d: lane: the 4 element indices it received (element = 2 bytes)\n", pat);
    for (int l = 0; l < 64; ++l) printf("  l%02d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l % 4 == 3) ? "\n" : "");
  }
  return 0;
}
