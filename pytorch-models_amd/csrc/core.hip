// core.hip - ABI version and error strings of libpm_mi355x.so.
#include "common.h"

extern "C" int pm_abi_version(void) { return PM_ABI_VERSION; }

extern "C" const char* pm_strerror(int code) {
  switch (code) {
    case PM_OK: return "ok";
    case PM_EINVAL: return "invalid argument (null pointer, negative size or leading dimension smaller than the row)";
    case PM_EUNSUPPORTED:
      return "shape not covered by the gfx950 kernels (linear: K % 8 == 0; layernorm: d % 8 == 0, "
             "d <= 4096; attention: head_dim 8..128 (% 8); vit_tokens: image sides % patch == 0)";
    case PM_ELAUNCH: return "hip kernel launch failed";
    case PM_EALIGN: return "pointer or leading dimension not aligned for 16-byte vector access";
    default: return "unknown pm_mi355x error code";
  }
}
