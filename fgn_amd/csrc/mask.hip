// Mask-head tail: logits + sigmoid, and full-image paste + threshold.
//   mmdet FCNMaskHead: ... upsample (ConvTranspose2d 2x2/2) -> ReLU -> conv_logits 1x1
//   (fgn_roi_head.py:380), then get_seg_masks/_do_paste_mask (fgn_roi_head.py:668-671).
//
// The 2x2/stride-2 transposed conv is four independent 1x1 convs, one per output
// sub-position; it runs on the MFMA conv kernel as ONE 1x1 conv with 4*C output
// channels ordered n = (dy*2+dx)*C + co, leaving [D,7,7,(dy,dx),C] in memory.  Because
// conv_logits is point-wise it is applied directly on that layout, and the 14x14 pixel
// shuffle is folded into this kernel's output index - no shuffle pass.
#include "post_common.h"

// x : [D][P*P][4][C]  (post-ReLU upsample output);  w : [C], bias scalar
// prob/logit out : [D][2P][2P]
__global__ __launch_bounds__(256) void mask_logits_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float bias, float* __restrict__ logits,
                                                          float* __restrict__ prob,
                                                          const int32_t* __restrict__ n_dev, int n_det, int P,
                                                          int C) {
    int D = n_det;
    if (n_dev) D = min(D, *n_dev);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int total = D * P * P * 4;
    if (wave >= total) return;
    const float* px = x + (size_t)wave * C;
    float acc = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(px + c);
        const float4 k = *reinterpret_cast<const float4*>(w + c);
        acc += (v.x * k.x + v.y * k.y) + (v.z * k.z + v.w * k.w);
    }
    acc = wave_reduce_sum(acc);
    if (lane == 0) {
        const int sub = wave & 3;
        const int cell = (wave >> 2) % (P * P);
        const int d = (wave >> 2) / (P * P);
        const int y = (cell / P) * 2 + (sub >> 1), xo = (cell % P) * 2 + (sub & 1);
        const float v = acc + bias;
        const size_t o = ((size_t)d * 2 * P + y) * 2 * P + xo;
        logits[o] = v;
        prob[o] = sigmoid32(v);
    }
}

extern "C" int fgn_mask_logits_f32(const float* x, const float* w, float bias, float* logits, float* prob,
                                   const int32_t* n_dev, int n_det, int roi_size, int C, hipStream_t stream) {
    if (!x || !w || !logits || !prob) return FGN_ERR_ARG;
    if (C % 4) return FGN_ERR_SHAPE;
    if (n_det == 0) return FGN_OK;
    const int waves = n_det * roi_size * roi_size * 4;
    hipLaunchKernelGGL(mask_logits_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, stream, x, w, bias, logits, prob,
                       n_dev, n_det, roi_size, C);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ----------------------------------------------------------------------------------------------
// paste: out[d][y][x] = bilinear(prob[d], grid(x,y)) >= thr inside the CPU path's
// integer-expanded box (floor(x0)-1 .. ceil(x1)+1, clipped), 0 elsewhere
// (_do_paste_mask with skip_empty=True, one mask per chunk).  grid_sample semantics:
// align_corners=False, zero padding.  HBM-bound: D*H*W bytes written, 4 pixels per lane.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float paste_sample(const float* __restrict__ m, int MS, float gx, float gy) {
    // unnormalise: ((g + 1) * size - 1) / 2
    const float ix = ((gx + 1.f) * (float)MS - 1.f) / 2.f;
    const float iy = ((gy + 1.f) * (float)MS - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wx0 = 1.f - wx1;
    const float wy1 = iy - fy, wy0 = 1.f - wy1;
    auto at = [&](int yy, int xx) -> float {
        return ((unsigned)yy < (unsigned)MS && (unsigned)xx < (unsigned)MS) ? m[yy * MS + xx] : 0.f;
    };
    return at(y0, x0) * (wx0 * wy0) + at(y0, x1) * (wx1 * wy0) + at(y1, x0) * (wx0 * wy1) + at(y1, x1) * (wx1 * wy1);
}

__global__ __launch_bounds__(256) void mask_paste_kernel(const float* __restrict__ prob,
                                                         const float* __restrict__ boxes, int box_stride,
                                                         uint8_t* __restrict__ out,
                                                         const int32_t* __restrict__ n_dev, int n_det, int H, int W,
                                                         int MS, float thr) {
    int D = n_det;
    if (n_dev) D = min(D, *n_dev);
    const long long HW = (long long)H * W;
    const long long total4 = ((long long)n_det * HW + 3) / 4;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < total4;
         q += (long long)gridDim.x * blockDim.x) {
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = q * 4 + k;
            if (i >= (long long)n_det * HW) break;
            const int d = (int)(i / HW);
            if (d >= D) continue;
            const int rem = (int)(i - (long long)d * HW);
            const int y = rem / W, x = rem - y * W;
            const float* b = boxes + (size_t)d * box_stride;
            const float bx0 = b[0], by0 = b[1], bx1 = b[2], by1 = b[3];
            const int x0i = max((int)floorf(bx0) - 1, 0), y0i = max((int)floorf(by0) - 1, 0);
            const int x1i = min((int)ceilf(bx1) + 1, W), y1i = min((int)ceilf(by1) + 1, H);
            if (x < x0i || x >= x1i || y < y0i || y >= y1i) continue;
            float gx = ((float)x + 0.5f - bx0) / (bx1 - bx0) * 2.f - 1.f;
            float gy = ((float)y + 0.5f - by0) / (by1 - by0) * 2.f - 1.f;
            if (isinf(gx)) gx = 0.f;
            if (isinf(gy)) gy = 0.f;
            const float v = paste_sample(prob + (size_t)d * MS * MS, MS, gx, gy);
            if (v >= thr) packed |= 1u << (8 * k);
        }
        if ((q + 1) * 4 <= (long long)n_det * HW) {
            reinterpret_cast<uint32_t*>(out)[q] = packed;
        } else {
            for (int k = 0; k < 4 && q * 4 + k < (long long)n_det * HW; ++k) out[q * 4 + k] = (packed >> (8 * k)) & 0xff;
        }
    }
}

extern "C" int fgn_mask_paste_u8(const float* prob, const float* boxes, int box_stride, uint8_t* out,
                                 const int32_t* n_dev, int n_det, int img_h, int img_w, int mask_size, float thr,
                                 hipStream_t stream) {
    if (!prob || !boxes || !out) return FGN_ERR_ARG;
    if (n_det == 0) return FGN_OK;
    const long long total4 = ((long long)n_det * img_h * img_w + 3) / 4;
    const int grid = (int)std::min<long long>((total4 + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(mask_paste_kernel, dim3(grid), dim3(256), 0, stream, prob, boxes, box_stride, out, n_dev,
                       n_det, img_h, img_w, mask_size, thr);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
