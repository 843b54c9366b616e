// Library identification entry points (see include/scarches_hip.h).
#include <hip/hip_runtime.h>
#include <string.h>

extern "C" int sca_abi_version(void) { return 1; }

extern "C" int sca_device_name(char* name, int len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return (int)e;
    if (name && len > 0) {
        strncpy(name, prop.gcnArchName, (size_t)len - 1);
        name[len - 1] = 0;
    }
    return 0;
}
