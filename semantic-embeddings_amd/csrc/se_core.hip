// se_core.hip -- library-level entry points of include/sehip.h (version, error text).
#include "se_common.h"

namespace se {

char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace se

extern "C" int se_version(void) { return 300; /* 0.3.0: se_retrieve_topk takes K-blocks and ldg (fused distance + top-k) */ }
extern "C" const char *se_last_error(void) { return se::err_buf(); }
extern "C" const char *se_build_arch(void) { return "gfx950"; }
