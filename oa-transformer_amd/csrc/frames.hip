// Input pipeline, device side: decoded frames -> the clip tensor the encoder reads.
// The reference does this per sample on CPU worker processes: frames.float() / 255 (base_dataset.py:519,533,544), then
// torchvision transforms on the tensor - Resize / CenterCrop / RandomResizedCrop / RandomHorizontalFlip / Normalize
// (data_loader/transforms.py:4-31; the OA datasets: Resize((224, 224)) + Normalize, base_dataset_global_local.py:251-257).
// On tensors torchvision's Resize is torch.nn.functional.interpolate(mode="bilinear", align_corners=False) (no
// antialiasing in the torchvision the reference pins), which is what one launch computes here together with the crop,
// the flip, the /255 and the normalisation, for all frames of a batch: uint8 HWC in, bf16 / fp32 CHW out.
#include "common.h"

namespace oat {

struct FrameArgs {
  const void* in; int in_u8_hwc;       // 1: uint8 [F, H, W, 3]; 0: float [F, 3, H, W]
  int F, H, W;
  float x0, y0, cw, ch;                // crop box in source pixels (cw, ch > 0)
  int flip;                            // horizontal flip of the OUTPUT
  void* out; int out_bf16; int OH, OW; // [F, 3, OH, OW]
  float scale;                         // applied to the sampled value (1 / 255 for uint8 frames)
  float mean[3], inv_std[3];           // (v - mean) * inv_std ; mean 0 / inv_std 1: no normalisation
};

OAT_DEV float frame_at(const FrameArgs& a, int f, int c, int y, int x) {
  if (a.in_u8_hwc) return (float)reinterpret_cast<const uint8_t*>(a.in)[(((size_t)f * a.H + y) * a.W + x) * 3 + c];
  return reinterpret_cast<const float*>(a.in)[(((size_t)f * 3 + c) * a.H + y) * a.W + x];
}

// one thread per output pixel (all 3 channels: the HWC source bytes of a pixel are adjacent)
__global__ void frames_resize_kernel(FrameArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)a.F * a.OH * a.OW;
  if (i >= total) return;
  const int ox = (int)(i % a.OW), oy = (int)((i / a.OW) % a.OH), f = (int)(i / ((long long)a.OW * a.OH));
  const int sxo = a.flip ? a.OW - 1 - ox : ox;
  // torch's area_pixel_compute_source_index, align_corners = False, applied inside the crop box
  const float sy = fmaxf(((float)oy + 0.5f) * (a.ch / (float)a.OH) - 0.5f, 0.f);
  const float sx = fmaxf(((float)sxo + 0.5f) * (a.cw / (float)a.OW) - 0.5f, 0.f);
  const int chh = (int)a.ch, cww = (int)a.cw, bx = (int)a.x0, by = (int)a.y0;
  const int y0 = min((int)sy, chh - 1), x0 = min((int)sx, cww - 1);
  const int y1 = min(y0 + 1, chh - 1), x1 = min(x0 + 1, cww - 1);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v00 = frame_at(a, f, c, by + y0, bx + x0), v01 = frame_at(a, f, c, by + y0, bx + x1);
    const float v10 = frame_at(a, f, c, by + y1, bx + x0), v11 = frame_at(a, f, c, by + y1, bx + x1);
    // torch's order of operations: (1-ly) * ((1-lx) v00 + lx v01) + ly * ((1-lx) v10 + lx v11)
    float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    v = (v * a.scale - a.mean[c]) * a.inv_std[c];
    const size_t o = (((size_t)f * 3 + c) * a.OH + oy) * a.OW + ox;
    if (a.out_bf16) reinterpret_cast<bf16*>(a.out)[o] = f2bf(v);
    else reinterpret_cast<float*>(a.out)[o] = v;
  }
}

}  // namespace oat

using namespace oat;

// frames: uint8 [F, H, W, 3] (in_u8_hwc = 1) or float [F, 3, H, W] (0).  crop = {x0, y0, w, h} in source pixels (NULL: the
// whole frame).  out: [F, 3, OH, OW] bf16 (out_bf16 = 1) or fp32.  scale multiplies the sampled value (1/255 for uint8
// frames); mean / std: 3 floats each or NULL (no normalisation).
extern "C" int oat_frames_resize(const void* frames, int in_u8_hwc, int F, int H, int W, const float* crop, int flip, void* out,
                                 int out_bf16, int OH, int OW, float scale, const float* mean, const float* std, void* stream) {
  if (F <= 0 || OH <= 0 || OW <= 0) return 0;
  if (!frames || !out || H <= 0 || W <= 0) { set_error("frames_resize: null pointer / empty frame"); return -4; }
  FrameArgs a{frames, in_u8_hwc, F, H, W, 0.f, 0.f, (float)W, (float)H, flip, out, out_bf16, OH, OW, scale, {0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}};
  if (crop) {
    a.x0 = crop[0]; a.y0 = crop[1]; a.cw = crop[2]; a.ch = crop[3];
    if (a.x0 < 0 || a.y0 < 0 || a.cw < 1 || a.ch < 1 || a.x0 + a.cw > W || a.y0 + a.ch > H) { set_error("frames_resize: crop box outside the frame"); return -3; }
  }
  if (mean && std) for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.inv_std[c] = 1.f / std[c]; }
  const long long total = (long long)F * OH * OW;
  OAT_LAUNCH(frames_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("frames_resize");
}
