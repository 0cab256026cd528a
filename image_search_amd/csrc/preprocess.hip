// preprocess.hip — image_prepare_resnet (server/src/clip.rs:153-175) behind the C ABI, standalone:
// host RGB8 image of any size -> (device: CatmullRom resize [+ normalisation]) -> host result.
// The batched form that feeds the tower directly is mi_clip_embed_images in vit.hip.
#include <vector>

#include "common.h"
#include "preprocess_kernels.h"

using namespace mi;

namespace {
struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) { HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16)); }
    ~DevBuf() { (void)hipFree(p); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

template <bool TO_CHW>
void run(int device, const uint8_t* src, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, void* dst) {
    if (!src || !dst) fail(MI_ERR_INVALID, "null buffer");
    const char* why = nullptr;
    if (!resize_supported(w, h, nw, nh, &why)) fail(MI_ERR_UNSUPPORTED, "resize %ux%u -> %ux%u: %s", w, h, nw, nh, why);
    DeviceGuard g(device);
    const size_t in_b = (size_t)w * h * 3, out_n = (size_t)nw * nh * 3;
    DevBuf d_src(in_b), d_tmp((size_t)nh * w * 3 * 4), d_out(out_n * (TO_CHW ? 4 : 1));
    hipStream_t s = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{s};
    HIP_CHECK(hipMemcpyAsync(d_src.p, src, in_b, hipMemcpyHostToDevice, s));
    resize_catmullrom_launch<TO_CHW>((const uint8_t*)d_src.p, w, h, nw, nh, (float*)d_tmp.p, TO_CHW ? nullptr : (uint8_t*)d_out.p,
                                     TO_CHW ? (float*)d_out.p : nullptr, s);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(dst, d_out.p, out_n * (TO_CHW ? 4 : 1), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
}
}  // namespace

extern "C" {

int mi_resize_catmullrom_rgb8(int device, const uint8_t* rgb8, uint32_t width, uint32_t height, uint32_t new_width,
                              uint32_t new_height, uint8_t* out) {
    return guarded([&] { run<false>(device, rgb8, width, height, new_width, new_height, out); });
}

int mi_image_prepare_resnet(int device, const uint8_t* rgb8, uint32_t width, uint32_t height, float* chw) {
    return guarded([&] { run<true>(device, rgb8, width, height, 224, 224, chw); });
}

}  // extern "C"
