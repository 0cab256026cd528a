// vit.hip — Seam A (placeholder while the kernels are brought up).
#include "common.h"
using namespace mi;
extern "C" {
int mi_clip_load(const char*, int, int, mi_clip** out) { return guarded([&] { if (out) *out = nullptr; fail(MI_ERR_UNSUPPORTED, "ViT path not built yet"); }); }
void mi_clip_free(mi_clip*) {}
int mi_clip_info(const mi_clip*, uint32_t*) { return guarded([&] { fail(MI_ERR_UNSUPPORTED, "ViT path not built yet"); }); }
int mi_clip_embed(mi_clip*, const float*, size_t, float*) { return guarded([&] { fail(MI_ERR_UNSUPPORTED, "ViT path not built yet"); }); }
int mi_clip_embed_device(mi_clip*, const float*, size_t, float*, void*) { return guarded([&] { fail(MI_ERR_UNSUPPORTED, "ViT path not built yet"); }); }
int mi_clip_embed_rgb8(mi_clip*, const uint8_t*, size_t, float*) { return guarded([&] { fail(MI_ERR_UNSUPPORTED, "ViT path not built yet"); }); }
}
