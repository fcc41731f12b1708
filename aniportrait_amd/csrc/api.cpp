// Host-side glue of libaniportrait_hip.so: version, thread-local error string, device query.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/aniportrait_hip.h"

static thread_local char g_err[512] = "";

void anip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int anip_version(void) { return ANIP_ABI_VERSION; }

extern "C" const char* anip_last_error(void) { return g_err; }

extern "C" int anip_device_info(char* arch, int arch_len, int* num_cu) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    anip_set_error("anip_device_info: hipGetDevice: %s", hipGetErrorString(e));
    return -2;
  }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    anip_set_error("anip_device_info: hipGetDeviceProperties: %s", hipGetErrorString(e));
    return -2;
  }
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  if (num_cu) *num_cu = prop.multiProcessorCount;
  return 0;
}

// ---- per-device caches (any device ordinal: a CPX-partitioned node exposes up to 64) ------------------------------------------
namespace {
std::mutex g_dev_mu;
std::map<std::pair<const void*, int>, int> g_lds_limit;   // (kernel, device) -> dynamic LDS limit already granted
std::map<int, int> g_cu_count;
}  // namespace

int anip_raise_lds_limit(const void* kernel, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  std::lock_guard<std::mutex> lk(g_dev_mu);
  int& have = g_lds_limit[std::make_pair(kernel, dev)];
  if (have >= bytes) return 0;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  have = bytes;
  return 0;
}

int anip_cu_count() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  std::lock_guard<std::mutex> lk(g_dev_mu);
  auto it = g_cu_count.find(dev);
  if (it != g_cu_count.end()) return it->second;
  hipDeviceProp_t prop;
  const int n = (hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : -1;
  g_cu_count[dev] = n;
  return n;
}

// ---- per-kernel HIP-event profiling -------------------------------------------------------------
namespace {
struct ProfRec {
  int kid;
  hipEvent_t a, b;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
const char* const kKernelNames[ANIP_K_COUNT] = {
    "gemm_kernel<false>", "gemm_kernel<true> (conv3x3)", "gn_stats_kernel", "gn_apply_kernel", "layernorm_kernel",
    "ref_attn_kernel", "temporal_attn_kernel", "softmax_rows_kernel", "conv_small_kernel", "linear_small_kernel",
    "elementwise", "batchnorm_kernels"};
}  // namespace

void anip_prof_begin(int kid, hipStream_t s) {
  if (!g_prof_on) return;
  ProfRec r;
  r.kid = kid;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, s);
  g_prof.push_back(r);
}

void anip_prof_end(int kid, hipStream_t s) {
  if (!g_prof_on || g_prof.empty()) return;
  ProfRec& r = g_prof.back();
  if (r.kid == kid) (void)hipEventRecord(r.b, s);
}

extern "C" int anip_profile_enable(int on) {
  g_prof_on = on != 0;
  return 0;
}

extern "C" int anip_profile_collect(int max_ids, int64_t* launches, double* total_ms) {
  return anip_profile_collect_records(max_ids, launches, total_ms, 0, nullptr, nullptr, nullptr);
}

extern "C" int anip_profile_collect_records(int max_ids, int64_t* launches, double* total_ms, int64_t max_records,
                                            int* rec_kid, float* rec_ms, int64_t* n_records) {
  for (int i = 0; i < max_ids; ++i) {
    launches[i] = 0;
    total_ms[i] = 0.0;
  }
  int rc = 0;
  int64_t nrec = 0;
  for (ProfRec& r : g_prof) {
    float ms = 0.f;
    hipError_t e = hipEventSynchronize(r.b);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.a, r.b);
    if (e != hipSuccess) {
      anip_set_error("anip_profile_collect: %s", hipGetErrorString(e));
      rc = -2;
    } else if (r.kid >= 0 && r.kid < max_ids) {
      launches[r.kid] += 1;
      total_ms[r.kid] += (double)ms;
    }
    if (nrec < max_records && rec_kid != nullptr && rec_ms != nullptr) {
      rec_kid[nrec] = r.kid;
      rec_ms[nrec] = (e == hipSuccess) ? ms : -1.f;
    }
    ++nrec;
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  if (n_records != nullptr) *n_records = nrec;
  g_prof.clear();
  return rc;
}

extern "C" const char* anip_profile_kernel_name(int kid) {
  return (kid >= 0 && kid < ANIP_K_COUNT) ? kKernelNames[kid] : "?";
}
