// Error plumbing and version entry points of libmmbidaf_hip.so.
#include <stdarg.h>

#include "common.h"

namespace mmb {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace mmb

extern "C" int mmb_version(void) { return MMB_VERSION; }
extern "C" const char* mmb_last_error(void) { return mmb::err_buf(); }
