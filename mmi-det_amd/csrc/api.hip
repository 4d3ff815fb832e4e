// Library-level entry points: version and per-thread error text.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void mmi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mmi_version(void) { return 100; }
extern "C" const char* mmi_last_error(void) { return g_err; }

// ---- fill (see common.h: used instead of hipMemsetAsync everywhere in the library) -----------------------------------------
namespace {
__global__ void fill_u32_kernel(uint32_t* __restrict__ p, uint32_t v, size_t nwords) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) p[i] = v;
}
}  // namespace

int mmi_fill_bytes(void* ptr, int value, size_t bytes, hipStream_t stream) {
  MMI_CHECK_ARG(ptr != nullptr && ((uintptr_t)ptr & 3) == 0 && (bytes & 3) == 0, "mmi_fill_bytes: pointer and size must be multiples of 4");
  if (bytes == 0) return MMI_OK;
  const uint32_t b = (uint32_t)(value & 0xFF), v = b | (b << 8) | (b << 16) | (b << 24);
  const size_t nwords = bytes / 4;
  size_t blocks = (nwords + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (uint32_t*)ptr, v, nwords);
  MMI_CHECK_LAUNCH("mmi_fill_bytes");
  return MMI_OK;
}
