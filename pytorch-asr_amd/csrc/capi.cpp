// Library-level C ABI helpers (version, error strings).
#include "../../include/asr_amd.h"

extern "C" int asr_abi_version(void) { return 22; }

extern "C" const char *asr_strerror(int code) {
    switch (code) {
        case ASR_OK: return "ok";
        case ASR_EINVAL: return "invalid argument (shape or null pointer)";
        case ASR_EUNSUPPORTED: return "shape not supported by the gfx950 kernels";
        case ASR_ELAUNCH: return "HIP kernel launch failed";
        default: return "unknown error";
    }
}
