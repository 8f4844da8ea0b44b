// Error reporting + library identification for the acktr_hip C-ABI.
#include <stdarg.h>
#include <string.h>

#include "common.hpp"
#include "../../include/acktr_hip.h"

namespace acktr {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace acktr

extern "C" const char* acktr_last_error(void) { return acktr::g_err; }
extern "C" int acktr_abi_version(void) { return ACKTR_ABI_VERSION; }
extern "C" const char* acktr_target_arch(void) { return "gfx950"; }
