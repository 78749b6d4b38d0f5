// Error reporting + library identity for the vlni C-ABI (see include/vlni.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

void vlni_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vlni_last_error(void) { return g_err; }
extern "C" int vlni_version(void) { return 1; }
