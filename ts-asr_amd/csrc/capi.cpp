// Host-only part of the C-ABI: error string, version, device probe.
#include <hip/hip_runtime_api.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/tsasr_hip.h"

static thread_local char g_err[512] = "";

void tsasr_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
const char *tsasr_last_error(void) { return g_err; }
int tsasr_version(void) { return 1; }
int tsasr_device_ok(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
}
