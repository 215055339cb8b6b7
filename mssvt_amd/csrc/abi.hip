// abi.hip -- library identification and status strings of the C ABI.
#include "common.hip.h"

extern "C" int mssvt_hip_abi_version(void) { return 100; }

extern "C" const char *mssvt_hip_status_string(int status) {
    if (status == MSSVT_OK) return "ok";
    if (status == MSSVT_E_BADARG) return "mssvt: bad argument (null pointer or non-positive size)";
    if (status == MSSVT_E_TOOLARGE) return "mssvt: size exceeds what the kernel supports";
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "mssvt: unknown status";
}
