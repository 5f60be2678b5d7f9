// mot_linear.hip -- CONCAT_LINEAR mode of the fused front-end (placeholder until the MFMA kernel lands).
#include "mot_internal.hpp"

namespace mot {

size_t embed_mix_linear_workspace_bytes(const MotEmbedMixDesc &) { return 0; }

int launch_embed_mix_linear(const MotEmbedMixDesc &, hipStream_t) {
    return set_error(MOT_EUNSUPPORTED, "embed_mix: CONCAT_LINEAR is not built yet");
}

}  // namespace mot
