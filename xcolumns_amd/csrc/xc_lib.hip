// xc_lib.hip -- library-level entry points: version, errors, device query.
#include <stdarg.h>
#include <string.h>

#include "xc_host.h"

namespace xc {

char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail_arg(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int fail_hip(hipError_t e, const char *what) {
    snprintf(err_buf(), 512, "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
    (void)hipGetLastError(); // clear the sticky error
    return (int)e;
}

static int g_cu_count = -1;
static int g_waves_per_cu = 32;

static int query_device() {
    if (g_cu_count > 0) return 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail_hip(e, "hipGetDevice");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return fail_hip(e, "hipGetDeviceProperties");
    g_cu_count = prop.multiProcessorCount;
    g_waves_per_cu = prop.maxThreadsPerMultiProcessor / 64;
    return 0;
}

int default_row_waves(int64_t n_rows) {
    if (query_device() != 0 || g_cu_count <= 0) return 1024;
    // memory-bound row loops: fill every wave slot, grid-stride the rest
    int64_t cap = (int64_t)g_cu_count * g_waves_per_cu;
    int64_t w = n_rows < cap ? n_rows : cap;
    return (int)(w < 1 ? 1 : w);
}

} // namespace xc

extern "C" {

int xc_abi_version(void) { return XC_ABI_VERSION; }

const char *xc_last_error(void) { return xc::err_buf(); }

int xc_device_info(int *cu_count, int *waves_per_cu, char *arch, int arch_len) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return xc::fail_arg(XC_ERR_NO_DEVICE, "no HIP device visible");
    }
    int dev = 0;
    XC_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    XC_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (waves_per_cu) *waves_per_cu = prop.maxThreadsPerMultiProcessor / 64;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return XC_OK;
}

} // extern "C"
