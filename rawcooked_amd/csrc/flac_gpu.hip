// flac_gpu.hip -- FLAC encoder entry points (device kernels land in this file).
#include "rc_common.h"
extern "C" int rcgpu_flac_create(const rcgpu_flac_config*, rcgpu_flac**) { return rc::fail(300, "flac: encoder not built yet"); }
extern "C" void rcgpu_flac_destroy(rcgpu_flac*) {}
extern "C" int rcgpu_flac_encode_host(rcgpu_flac*, const uint8_t*, uint64_t, uint8_t*, size_t, uint32_t*, uint32_t, uint32_t*) { return rc::fail(300, "flac: encoder not built yet"); }
extern "C" size_t rcgpu_flac_codec_private(const rcgpu_flac*, uint8_t*, size_t) { return 0; }
