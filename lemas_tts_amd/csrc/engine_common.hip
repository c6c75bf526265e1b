// Error reporting and the strict fp32 weight registry shared by the DiT and Vocos engines.
#include "engine_common.h"

#include <cstring>
#include <mutex>

namespace lemas {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) at %s:%d in %s", (int)e, hipGetErrorString(e), file, line, what);
  (void)hipGetLastError();
  return -(int)e;
}

// The > 64 KB dynamic-LDS opt-in of the GEMM kernels is a per-DEVICE function attribute: once per device a process touches (one
// process per GPU is the deployment model, but nothing stops a caller from building engines on two devices).
int kernels_init() {
  static std::mutex mu;
  static bool done[64] = {};
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (dev < 0 || dev >= 64) { set_error("kernels_init: device index %d out of range", dev); return LEMAS_E_STATE; }
  if (done[dev]) return 0;
  const hipError_t err = gemm_bf16_init();
  if (err != hipSuccess) return hip_fail(err, "gemm_bf16_init()", __FILE__, __LINE__);
  done[dev] = true;
  return 0;
}

int WeightStore::load(const char* name, const float* src, const int64_t* shape, int ndim, bool on_device) {
  auto it = schema.find(name);
  if (it == schema.end()) {
    set_error("unexpected tensor '%s' (strict load)", name);
    return LEMAS_E_WEIGHT;
  }
  const std::vector<int64_t>& want = it->second;
  bool ok = (int)want.size() == ndim;
  size_t numel = 1;
  for (int i = 0; i < ndim && ok; ++i) {
    ok = want[i] == shape[i];
    numel *= (size_t)shape[i];
  }
  if (!ok) {
    set_error("tensor '%s' has the wrong shape (ndim %d)", name, ndim);
    return LEMAS_E_WEIGHT;
  }
  // ONE device allocation for the whole schema, made (and zero-filled, once) when the first tensor arrives: a checkpoint is ~370
  // tensors, and a hipMalloc + a synchronised fill each was most of the load time.  Every tensor owns a fixed 256-B aligned slot
  // with 64 floats of slack behind it (the fp32 GEMM reads whole float4 groups of offset sub-matrices: input_embed.proj columns);
  // a reload writes the same slot.
  if (!arena) {
    size_t total = 0;
    for (const auto& kv : schema) {
      size_t n = 1;
      for (int64_t d : kv.second) n *= (size_t)d;
      slot[kv.first] = total;
      total += ((n + 64) * sizeof(float) + 255) & ~(size_t)255;
    }
    HIP_TRY(hipMalloc((void**)&arena, total));
    HIP_TRY(zero_fill_sync(arena, total));
    arena_bytes = total;
  }
  // a RE-load overwrites the slot in place: nothing of the owning engine may still be reading it (a decode / encode in flight on a
  // non-blocking stream is not ordered against the NULL-stream copy below)
  if (t.find(name) != t.end()) HIP_TRY(hipDeviceSynchronize());
  Tensor& x = t[name];
  x.dev = reinterpret_cast<float*>(arena + slot[name]);
  // synchronous on the NULL stream in both cases; a device source written on another stream must be complete before the call
  HIP_TRY(hipMemcpy(x.dev, src, numel * sizeof(float), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  if (on_device) HIP_TRY(hipStreamSynchronize(nullptr));
  x.shape.assign(shape, shape + ndim);
  x.numel = numel;
  return 0;
}

int WeightStore::check_complete() const {
  for (const auto& kv : schema)
    if (t.find(kv.first) == t.end()) {
      set_error("missing tensor '%s' (strict load)", kv.first.c_str());
      return LEMAS_E_WEIGHT;
    }
  return 0;
}

void WeightStore::release() {
  if (arena) (void)hipFree(arena);
  arena = nullptr;
  arena_bytes = 0;
  slot.clear();
  t.clear();
}

}  // namespace lemas

extern "C" {
const char* lemas_last_error(void) { return lemas::g_err; }
int lemas_version(void) { return 200; }   // 200: lemas_sample_args carries struct_size (include/lemas_hip.h)
}
