// Error plumbing of libvq2 (thread-local, see include/vq2.h).
#include "vq2_common.h"

namespace vq2 {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return VQ2_OK;
    return set_error(VQ2_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace vq2

extern "C" int vq2_version(void) { return 1; }
extern "C" const char *vq2_last_error(void) { return vq2::g_err; }
