// Version / error-string entry points of the C ABI.
#include "../../include/mlqem_hip.h"

extern "C" int mlqem_abi_version(void) { return MLQEM_ABI_VERSION; }

extern "C" const char* mlqem_error_string(int code) {
  switch (code) {
    case MLQEM_OK: return "ok";
    case MLQEM_ERR_BAD_ARG: return "bad argument (null pointer, negative size or leading dimension too small)";
    case MLQEM_ERR_UNSUPPORTED: return "unsupported shape for this kernel";
    case MLQEM_ERR_LAUNCH: return "HIP launch or runtime error";
    case MLQEM_ERR_WORKSPACE: return "workspace missing or too small";
    default: return "unknown error code";
  }
}
