// hko_media.h — CPU ORACLE (test infrastructure): participating media.
// Round-1 status: records + hooks only.  Homogeneous / Grid / RGBGrid / NanoVDB delta tracking
// (src/integrators/volpath/delta-tracking.jl, media.jl, nanovdb.jl, medium-scatter.jl) are SURVEY §8
// rows a27-a30 and land with the media widening; until then a scene with n_media > 0 is rejected by
// hko_scene_create so nothing silently renders without its media.
#pragma once
#include "hikari_mi355x.h"
#include "hko_spectral.h"

namespace hko {

struct MediaCtx {
    const hk_medium* media = nullptr;
    int32_t n = 0;
    const RGB2SpecTable* table = nullptr;
};
inline void init_media(MediaCtx& m, const hk_scene_desc* d, const RGB2SpecTable* t) {
    m.media = d->media;
    m.n = d->n_media;
    m.table = t;
}
// compute_transmittance_ratio_tracking  intersection.jl:422-542
inline void transmittance_ratio_tracking(const MediaCtx&, int32_t, V3, V3, float, const Wavelengths&, Spec& T, Spec& ru, Spec& rl, uint64_t&) {
    T = Spec(1.0f);
    ru = Spec(1.0f);
    rl = Spec(1.0f);
}

}  // namespace hko
