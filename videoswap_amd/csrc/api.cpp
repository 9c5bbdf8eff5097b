// Host-side plumbing of libvsx: ABI version, thread-local error string, launch check.
#include "common.h"

#include <string.h>

namespace {
thread_local char g_err[512] = "";
}

int vsx_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int vsx_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vsx_fail(VSX_E_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return VSX_OK;
}

#ifndef VSX_SOURCE_DIGEST
#define VSX_SOURCE_DIGEST "unstamped"
#endif

extern "C" int vsx_abi_version(void) { return VSX_ABI_VERSION; }
// sha256 of csrc/ + include/vsx.h + the compiler flags this binary was built from (videoswap_amd/build.py): the
// loader refuses a library whose digest differs from the sources next to it (stale kernels after a pull).
// The string is stored behind a marker so that build.py can read it from the FILE (no dlopen: a stale image that was
// loaded once stays mapped by name, and the rebuilt library would then report the old digest).
static const char g_digest[] = "@vsx-source-digest:" VSX_SOURCE_DIGEST;
extern "C" const char* vsx_source_digest(void) { return g_digest + 19; }
extern "C" const char* vsx_last_error(void) { return g_err; }
