// ABI version + thread-local error string of libmevi_hip.so.
#include "common.h"

#include <string.h>

namespace mevi {
namespace {
thread_local char g_err[512] = "";
}
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mevi

extern "C" int mevi_abi_version(void) { return 1; }
extern "C" const char *mevi_last_error(void) { return mevi::g_err; }
