// capi.hip -- library-level entry points of libeda_hip.so (include/eda_hip.h).
#include "eda_common.h"

#include <stdarg.h>

int g_eda_fma_mode = 0;

static thread_local char g_err[512] = "";

void eda_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int eda_version(void) { return EDA_HIP_ABI_VERSION; }
extern "C" const char *eda_last_error_string(void) { return g_err; }
extern "C" int eda_set_fma_mode(int mode) {
  if (mode != 0 && mode != 1) {
    eda_set_error("eda_set_fma_mode: mode must be 0 or 1");
    return EDA_ERR_INVALID_ARG;
  }
  g_eda_fma_mode = mode;
  return 0;
}
extern "C" int eda_get_fma_mode(void) { return g_eda_fma_mode; }
