// Version / error-string entry points of libuavgnn (see include/uavgnn.h).
#include "common.h"

extern "C" int uavgnn_version(void) { return UAVGNN_VERSION; }

extern "C" const char* uavgnn_strerror(int code) {
  if (code == 0) return "ok";
  if (code == UAVGNN_EINVAL) return "uavgnn: invalid argument (null pointer or bad size)";
  if (code == UAVGNN_EUNSUPPORTED) return "uavgnn: shape outside the compiled instantiations";
  if (code == UAVGNN_EWORKSPACE) return "uavgnn: workspace too small";
  if (code < 0) return hipGetErrorString(static_cast<hipError_t>(-code));
  return "uavgnn: unknown code";
}
