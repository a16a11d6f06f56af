// Library-wide C-ABI helpers: version, thread-local error text.
#include <stdarg.h>
#include <stdio.h>

#include "qt_common.h"

namespace {
thread_local char g_err[512] = "";
}

void qt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int qt_version(void) { return 100; }
extern "C" const char* qt_last_error(void) { return g_err; }
