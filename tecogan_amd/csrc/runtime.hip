// Library-level runtime services of libtecogan_hip.so: the built-in launch profiler.
//
// bench.py has to state, for the kernel that dominates a step, its average launch duration measured "live, with
// HIP events on the stream the kernel is launched on".  Inside a replayed hipGraph no event can be placed around
// one node, and host-side event pairs around a 3 us kernel measure the launch gap, not the kernel.  The profiler
// therefore uses the dispatch's own start / stop timestamps: when enabled, every instrumented launch (TG_LAUNCH in
// common.h) goes through hipExtLaunchKernelGGL with a start and a stop event, and tg_prof_collect() turns the pairs
// into per-kernel {calls, total us, algorithmic FLOPs, algorithmic bytes}.  These are the same timestamps
// rocprofv3 --kernel-trace reports, so the figures agree with the committed profiles/ summaries.
//
// Off by default (one relaxed atomic load per launch); only meaningful on eager streams (not under capture).
// The record list is process-global and mutex-protected: profiling is a debugging/measurement mode, every compute
// entry point stays re-entrant.
#include "common.h"
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {
struct Rec {
  const char* name;
  double flops, bytes;
  hipEvent_t e0, e1;
};
std::atomic<int> g_on{0};
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;       // recycled events
}  // namespace

bool tg_det() {
  static const bool on = getenv("TG_DETERMINISTIC") != nullptr && atoi(getenv("TG_DETERMINISTIC")) != 0;
  return on;
}

int tg_num_cus() {
  static std::atomic<int> cached[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  int n = cached[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

bool tg_prof_slot(const char* name, double flops, double bytes, hipEvent_t* e0, hipEvent_t* e1) {
  if (!g_on.load(std::memory_order_relaxed)) return false;
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t ev[2];
  for (int i = 0; i < 2; ++i) {
    if (!g_pool.empty()) {
      ev[i] = g_pool.back();
      g_pool.pop_back();
    } else if (hipEventCreate(&ev[i]) != hipSuccess) {
      return false;
    }
  }
  g_recs.push_back(Rec{name, flops, bytes, ev[0], ev[1]});
  *e0 = ev[0];
  *e1 = ev[1];
  return true;
}

extern "C" int tg_prof_enable(int on) {
  g_on.store(on ? 1 : 0, std::memory_order_relaxed);
  return TG_OK;
}

extern "C" int tg_prof_collect(tg_prof_entry* out, int max_entries, int* count) {
  TG_CHECK_ARG(count != nullptr && (out != nullptr || max_entries == 0), "null pointer");
  std::lock_guard<std::mutex> lk(g_mu);
  std::map<std::string, tg_prof_entry> agg;
  std::map<std::string, std::vector<double>> samples;
  int rc = TG_OK;
  for (const Rec& r : g_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) {
      (void)hipGetLastError();
      rc = TG_ELAUNCH;
    } else {
      tg_prof_entry& e = agg[r.name];
      if (e.calls == 0) {
        memset(&e, 0, sizeof(e));
        strncpy(e.name, r.name, sizeof(e.name) - 1);
      }
      e.calls += 1;
      samples[r.name].push_back((double)ms * 1e3);
      e.flops += r.flops;
      e.bytes += r.bytes;
    }
    g_pool.push_back(r.e0);
    g_pool.push_back(r.e1);
  }
  g_recs.clear();
  // total = sum of the launches' durations, a launch whose event pair reads more than 8 x the kernel's median counted AS the median:
  // one pair in ~10^4 comes back with a start stamp tens of milliseconds old (met once: 54 launches of 12 us summed to 81 ms,
  // profiles/r06zy_bench.json) and would otherwise own the kernel's average
  for (auto& kv : samples) {
    std::vector<double> v = kv.second;
    std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
    const double med = v[v.size() / 2];
    double tot = 0.0;
    for (double d : kv.second) tot += (v.size() >= 3 && d > 8.0 * med) ? med : d;
    agg[kv.first].total_us = tot;
  }
  int n = 0;
  for (auto& kv : agg) {
    if (n < max_entries) out[n] = kv.second;
    ++n;
  }
  *count = n;
  if (rc != TG_OK) tg_set_error("tg_prof_collect: an event pair could not be read (launch under stream capture?)");
  return rc;
}

// Device-side wall-clock stamp (100 MHz constant clock) as a one-thread kernel: the only timing probe that works INSIDE a
// captured hipGraph.  The engine places one at the start and end of every segment (TG_SEG_STAMPS=1) to read the real
// segment schedule of a replayed step without a tracing tool in the way (tools/seg_timeline.py).
__global__ void prof_stamp_kernel(unsigned long long* dst) { *dst = wall_clock64(); }

extern "C" int tg_prof_stamp(void* dst, void* stream) {
  TG_CHECK_ARG(dst != nullptr && (((uintptr_t)dst) & 7) == 0, "dst must be an 8-byte aligned device pointer");
  hipLaunchKernelGGL(prof_stamp_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), (unsigned long long*)dst);
  TG_CHECK_LAUNCH();
  return TG_OK;
}

// Node census of a captured hipGraph (the handle torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph() returns): total nodes and
// kernel nodes.  The multi-GPU engine asserts with it that a captured exchange segment really carries its RCCL kernels -- a
// one-rank communicator elides them and the captured graph is EMPTY, which an 8-GPU node must not discover by a wrong result.
extern "C" int tg_graph_node_count(void* graph, int* total, int* kernels) {
  TG_CHECK_ARG(graph != nullptr && total != nullptr && kernels != nullptr, "null pointer");
  size_t n = 0;
  if (hipGraphGetNodes(static_cast<hipGraph_t>(graph), nullptr, &n) != hipSuccess) {
    tg_set_error("tg_graph_node_count: hipGraphGetNodes failed: %s", hipGetErrorString(hipGetLastError()));
    return TG_ELAUNCH;
  }
  std::vector<hipGraphNode_t> nodes(n);
  int k = 0;
  if (n > 0) {
    if (hipGraphGetNodes(static_cast<hipGraph_t>(graph), nodes.data(), &n) != hipSuccess) {
      tg_set_error("tg_graph_node_count: hipGraphGetNodes failed: %s", hipGetErrorString(hipGetLastError()));
      return TG_ELAUNCH;
    }
    for (size_t i = 0; i < n; ++i) {
      hipGraphNodeType t;
      if (hipGraphNodeGetType(nodes[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel) ++k;
    }
  }
  *total = (int)n;
  *kernels = k;
  return TG_OK;
}
