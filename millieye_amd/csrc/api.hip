// libmillieye_hip.so - library-level entry points (version, errors, device query).
#include <stdarg.h>
#include <string.h>
#include "common.h"

namespace me {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace me

extern "C" {

int me_abi_version(void) { return ME_ABI_VERSION; }

const char* me_last_error(void) { return me::g_err; }

int me_device_query(int32_t* cu_count, int32_t* clock_khz, int32_t* lds_bytes) {
  int dev = 0;
  ME_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  ME_HIP(hipGetDeviceProperties(&prop, dev));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  if (lds_bytes) *lds_bytes = (int32_t)prop.sharedMemPerBlock;
  return 0;
}

int32_t me_sizeof(int32_t which) {
  switch (which) {
    case 0: return (int32_t)sizeof(me_conv_desc);
    case 1: return (int32_t)sizeof(me_pool_desc);
    case 2: return (int32_t)sizeof(me_yolo_desc);
    case 3: return (int32_t)sizeof(me_nms_desc);
    case 4: return (int32_t)sizeof(me_heads_desc);
    case 5: return (int32_t)sizeof(me_heads_weights);
    case 6: return (int32_t)sizeof(me_conv16_desc);
    case 7: return (int32_t)sizeof(me_pack_desc);
    case 8: return (int32_t)sizeof(me_bneck16_desc);
    case 9: return (int32_t)sizeof(me_adam_desc);
    default: return -1;
  }
}

}  // extern "C"
