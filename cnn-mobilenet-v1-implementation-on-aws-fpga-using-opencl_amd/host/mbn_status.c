/* mbn_status.c — status codes of the C-ABI as text (the reference prints "Error: Failed to ..." and exits,
 * MobileNet.c:156-160,261-265; here every call returns a code and never exits). */
#include "mbn.h"

const char *mbn_version(void) { return "mbn-mi355x 0.1 (gfx950)"; }

const char *mbn_strerror(int code)
{
    switch (code) {
    case MBN_OK: return "ok";
    case MBN_EINVAL: return "invalid argument";
    case MBN_ENOMEM: return "out of memory";
    case MBN_EDEVICE: return "HIP runtime error";
    case MBN_EIO: return "I/O error";
    case MBN_EFORMAT: return "bad file format";
    case MBN_ENOTFOUND: return "object not found";
    case MBN_ESHAPE: return "shape mismatch";
    case MBN_EUNSUPPORTED: return "unsupported";
    case MBN_ENODEVICE: return "no HIP device";
    default: return "unknown error";
    }
}
