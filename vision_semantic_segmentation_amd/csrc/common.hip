// libavl_hip.so: version + per-thread error string.
#include "avl_common.h"

namespace avl {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace avl

extern "C" const char* avl_version(void) { return "avl_hip 0.1 (gfx950)"; }

extern "C" int avl_last_error(char* buf, int len) {
    const char* e = avl::err_buf();
    int n = (int)strlen(e);
    if (buf && len > 0) {
        int c = n < len - 1 ? n : len - 1;
        memcpy(buf, e, c);
        buf[c] = 0;
    }
    return n;
}
