// Error reporting + ABI version for the PULPo HIP library.
#include "common.h"
#include "../../include/pulpo_hip.h"
#include <stdarg.h>

namespace pulpo {
char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code == 0 ? -1 : code;
}
}  // namespace pulpo

PULPO_API int pulpo_abi_version(void) { return PULPO_ABI_VERSION; }
PULPO_API const char* pulpo_last_error(void) { return pulpo::err_buf(); }
