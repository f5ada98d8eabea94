// Error plumbing + version of libla_hip.so (plain host C++; the kernels live in the .hip files).
#include <cstdarg>
#include <cstdio>

#include "../../include/la_hip.h"

static thread_local char g_err[512] = "";

void la_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* la_last_error(void) { return g_err; }
extern "C" int la_version(void) { return 1; }
