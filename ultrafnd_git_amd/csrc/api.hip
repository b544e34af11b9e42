// error plumbing + ABI version
#include <stdarg.h>
#include <string.h>

#include "common.hpp"

static thread_local char g_err[512] = "";

void ufnd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ufnd_last_error(void) { return g_err; }
extern "C" int ufnd_abi_version(void) { return UFND_ABI_VERSION; }
