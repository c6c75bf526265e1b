// Host-side plumbing shared by the engine translation units: error reporting, device buffers, weight registry.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/lemas_hip.h"
#include "common.h"
#include "kernels.h"

namespace lemas {

void set_error(const char* fmt, ...);
// once per process: kernel attributes (dynamic-LDS opt-ins) that must not be set on a launch path.  Every *_create and
// every lemas_k_* entry point calls it; returns 0 or a negative HIP code.
int kernels_init();
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define HIP_TRY(expr)                                                        \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) return ::lemas::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

#define RC_TRY(expr)          \
  do {                        \
    int _rc = (expr);         \
    if (_rc != 0) return _rc; \
  } while (0)

// Zero-fill that is COMPLETE when it returns.  hipMemset on device memory only enqueues the fill on the NULL stream
// (measured on MI355X / ROCm 7.2: the call returns in 3 us for 2 GiB) and the NULL stream does not order against the
// engines' non-blocking streams, so a plain hipMemset after hipMalloc can land AFTER the first kernels that write the
// buffer (tools/exp/memset_race_probe.hip: 191 of 200 trials lose the kernel's data).  That was the round-1 "box-dependent"
// config3 failure: tables and workspaces partly re-zeroed under load.  Allocation is rare, so waiting here is free.
inline hipError_t zero_fill_sync(void* p, size_t n) {
  hipError_t e = hipMemsetAsync(p, 0, n, nullptr);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(nullptr);
}

// Growable device allocation.  `*moved` (the owning engine's counter, if set) bumps on every (re)allocation so that the
// owner's cached hipGraphs, which baked the old address, can be invalidated -- per engine, not process-wide.
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  unsigned long long* moved = nullptr;
  int ensure(size_t n) {
    if (n <= bytes) return 0;
    if (p) HIP_TRY(hipFree(p));
    p = nullptr;
    bytes = 0;
    n = (n + 255) & ~(size_t)255;
    HIP_TRY(hipMalloc(&p, n));
    HIP_TRY(zero_fill_sync(p, n));  // pad regions must stay finite (attention v^T tail)
    bytes = n;
    if (moved) ++*moved;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <typename T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

struct Tensor {
  float* dev = nullptr;  // fp32 copy on device
  std::vector<int64_t> shape;
  size_t numel = 0;
};

// name -> fp32 device tensor, with a declared schema for strict loading (utils_infer.py:237 load_state_dict strict)
struct WeightStore {
  std::map<std::string, std::vector<int64_t>> schema;
  std::map<std::string, Tensor> t;
  char* arena = nullptr;                       // one allocation for every declared tensor (WeightStore::load)
  size_t arena_bytes = 0;
  std::map<std::string, size_t> slot;          // byte offset of each tensor's slot in the arena
  void declare(const std::string& name, std::vector<int64_t> shape) { schema[name] = std::move(shape); }
  // `src` is host memory, or (on_device) a device address on the current device, e.g. a view of the flat buffer that arrived
  // by RCCL broadcast (parallel.py): then the tensor never touches the host
  int load(const char* name, const float* src, const int64_t* shape, int ndim, bool on_device = false);
  int check_complete() const;
  const Tensor* find(const std::string& name) const {
    auto it = t.find(name);
    return it == t.end() ? nullptr : &it->second;
  }
  float* ptr(const std::string& name) const {
    const Tensor* x = find(name);
    return x ? x->dev : nullptr;
  }
  void release();
};

}  // namespace lemas
