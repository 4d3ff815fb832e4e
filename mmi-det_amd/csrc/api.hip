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
