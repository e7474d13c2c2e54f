// ABI version / error string / device probe.
#include "common.hpp"

extern "C" int lav_abi_version(void) { return LAV_ABI_VERSION; }

extern "C" const char *lav_last_error(void) { return lav::error_buffer(); }

extern "C" int lav_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return lav::fail(LAV_EHIP, "hipGetDeviceCount -> %s", hipGetErrorString(e));
    }
    return n;
}
