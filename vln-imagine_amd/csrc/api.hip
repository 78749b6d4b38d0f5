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

// Optional device-resident offset added to every dropout seed (all kernels that take a drop_seed). A captured hipGraph bakes
// its launch arguments in; with the seeds' base in device memory (advanced by a node of the graph) every replay draws new masks,
// and the backward nodes of the same replay regenerate the forward masks because they read the same value.
static const unsigned* g_seed_base = nullptr;
const unsigned* vlni_seed_base() { return g_seed_base; }
extern "C" int vlni_set_dropout_seed_base(const unsigned* device_ptr) {
  g_seed_base = device_ptr;
  return VLNI_OK;
}
