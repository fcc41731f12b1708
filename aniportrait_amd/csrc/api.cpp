// Host-side glue of libaniportrait_hip.so: version, thread-local error string, device query.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

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
