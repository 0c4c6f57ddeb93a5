// placeholder until the mel / speaker-encoder kernels land
#include "generate.hpp"
namespace qtts {
struct SpeakerState {};
void finalize_speaker(Model&) {}
void run_mel(Model&, const float*, const int*, int, int, float*, int, int*) { throw Error(ERR_STATE, "mel front end is not built into this library yet"); }
void run_speaker_embed(Model&, const float*, const int*, int, int, float*) { throw Error(ERR_STATE, "speaker encoder is not built into this library yet"); }
}  // namespace qtts
