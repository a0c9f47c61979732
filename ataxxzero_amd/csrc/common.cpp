// Error reporting and device selection for libataxxzero_hip.so.
#include "azh_host.h"

static thread_local char g_error[512] = "";

int azh_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *azh_last_error(void) { return g_error; }

extern "C" int azh_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess)
        return azh_fail(-100 - (int)e, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

extern "C" int azh_set_device(int device)
{
    AZH_HIP(hipSetDevice(device));
    return 0;
}

// "dddd:bb:dd.f" of a device — which physical GPU an ordinal is (two ranks that were handed the same card have
// different ordinals only if HIP_VISIBLE_DEVICES differs between them: the bus id shows it)
extern "C" int azh_device_pci_bus_id(int device, char *buf, int cap)
{
    if (!buf || cap < 16)
        return azh_fail(-1, "azh_device_pci_bus_id: buffer of at least 16 bytes needed");
    AZH_HIP(hipDeviceGetPCIBusId(buf, cap, device));
    return 0;
}

int azh_require_device(void)
{
    int n = azh_device_count();
    if (n <= 0) {
        if (n == 0)
            azh_fail(-3, "no HIP device visible: the MI355X path has no CPU fallback");
        return -3;
    }
    return 0;
}
