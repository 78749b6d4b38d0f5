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

// Optional device-resident counter of out-of-range table indices (embedding gathers / scatters). The kernels that take caller-supplied
// row indices (vlni_embed_combine_fwd, vlni_scatter_add_rows*) skip a row whose index is outside its table - no out-of-bounds access -
// and count it here when a counter is registered, so the host can raise like nn.Embedding does (ops.index_errors()), at a time of
// its choosing instead of a device-to-host sync per call.
static int* g_index_errors = nullptr;
int* vlni_index_error_counter() { return g_index_errors; }
extern "C" int vlni_set_index_error_counter(int* device_ptr) {
  g_index_errors = device_ptr;
  return VLNI_OK;
}

// Host -> device copy of a small table on `stream`. From PINNED host memory this is legal while the stream is being captured (it becomes a
// memcpy node that re-reads the host bytes on every replay: the caller keeps them alive and unchanged); torch's own copy_ is not.
extern "C" int vlni_upload(void* dst, const void* src_host, long bytes, void* stream) {
  VLNI_CHECK(dst && src_host && bytes > 0, VLNI_EINVAL, "upload: dst=%p src=%p bytes=%ld", dst, src_host, bytes);
  hipError_t e = hipMemcpyAsync(dst, src_host, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
  VLNI_CHECK(e == hipSuccess, VLNI_ELAUNCH, "upload: %s", hipGetErrorString(e));
  return VLNI_OK;
}
