// ABI bookkeeping for libsparsemarg_hip.so.
#include "common.hpp"

namespace smlvm {
thread_local int g_last_hip_error = 0;
}

extern "C" int smlvm_abi_version(void) { return SMLVM_ABI_VERSION; }

extern "C" int smlvm_last_hip_error(void) { return smlvm::g_last_hip_error; }

extern "C" const char* smlvm_error_string(int code)
{
    switch (code) {
        case SMLVM_OK: return "ok";
        case SMLVM_ERR_INVALID: return "invalid argument";
        case SMLVM_ERR_UNSUPPORTED: return "size not supported by the gfx950 kernels";
        case SMLVM_ERR_HIP: return "HIP launch failed (see smlvm_last_hip_error)";
        case SMLVM_ERR_WORKSPACE: return "workspace missing or too small";
        default: return "unknown error";
    }
}
