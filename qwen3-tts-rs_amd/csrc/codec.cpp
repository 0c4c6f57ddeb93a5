// placeholder until the codec kernels land
#include "generate.hpp"
namespace qtts {
struct CodecState {};
void finalize_codec(Model&) {}
void run_codec_decode(Model&, const uint32_t*, const int*, int, int, int, int, float*, int64_t, int64_t*) {
    throw Error(ERR_STATE, "codec decode is not built into this library yet");
}
}  // namespace qtts
