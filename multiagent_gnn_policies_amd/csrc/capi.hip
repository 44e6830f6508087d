// Library-level entry points of libmgp.so: version, error strings, device probe.
#include <string.h>
#include "mgp_common.h"

thread_local int mgp_tls_hip_error = 0;
thread_local void* mgp_tls_launch_events[2] = {nullptr, nullptr};

extern "C" int mgp_version(void) { return MGP_VERSION; }

extern "C" const char* mgp_last_hip_error(void)
{
    return hipGetErrorString((hipError_t)mgp_tls_hip_error);
}

extern "C" const char* mgp_strerror(int code)
{
    switch (code) {
        case MGP_OK: return "ok";
        case MGP_EINVAL: return "invalid argument (size, null pointer or unsupported combination)";
        case MGP_EALIGN: return "pointer is not sufficiently aligned";
        case MGP_ELAUNCH: return "HIP kernel launch failed";
        case MGP_ENODEV: return "no HIP device available";
        case MGP_EUNSUPPORTED: return "shape not covered by this kernel (use the composed ops)";
        default: return "unknown mgp error code";
    }
}

extern "C" int mgp_device_info(char* name, int cap)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MGP_ENODEV;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return MGP_ENODEV;
    if (name != nullptr && cap > 0) {
        strncpy(name, prop.gcnArchName, (size_t)cap - 1);
        name[cap - 1] = '\0';
    }
    return prop.multiProcessorCount;
}

extern "C" int mgp_set_launch_events(void* start_event, void* stop_event)
{
    mgp_tls_launch_events[0] = start_event;
    mgp_tls_launch_events[1] = stop_event;
    return MGP_OK;
}
