// Host-side glue of libtok_gfx950.so: thread-local error string, version.
#include "tok_common.h"

static thread_local char g_err[512] = "";

void tok_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* tok_last_error(void) { return g_err; }
extern "C" int tok_version(void) { return 1; }
