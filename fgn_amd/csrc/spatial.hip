// HBM-bound spatial kernels of the FGN path: input layout conversion, stem max-pool,
// RoIAlign (both flavours the reference uses) and the support-set reductions.
// All tensors are NHWC fp32; every lane moves 16 B and a wave covers contiguous
// channels, so global accesses are fully coalesced.
#include "common.h"

// ----------------------------------------------------------------------------------
// NCHW [B,3,H,W] -> NHWC4 [B,H,W,4] (4th channel zero) so the 7x7 stem conv can stage
// one filter tap per 16-byte load.  (input side of fgn.py:212,215)
// ----------------------------------------------------------------------------------
__global__ void nchw3_to_nhwc4_kernel(const float* __restrict__ x, float4* __restrict__ y, int HW,
                                      long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / HW;
        const int p = (int)(i - b * HW);
        const float* s = x + b * 3 * HW + p;
        y[i] = make_float4(s[0], s[HW], s[2 * (size_t)HW], 0.f);
    }
}

extern "C" int fgn_nchw3_to_nhwc4_f32(const float* x, float* y, int n_img, int H, int W,
                                      hipStream_t stream) {
    if (!x || !y) return FGN_ERR_ARG;
    const long long total = (long long)n_img * H * W;
    if (total == 0) return FGN_OK;
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(grid), dim3(256), 0, stream, x,
                       reinterpret_cast<float4*>(y), H * W, total);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ----------------------------------------------------------------------------------
// 3x3 / stride 2 / pad 1 max-pool, NHWC (mmdet ResNet stem; floor mode, -inf padding)
// ----------------------------------------------------------------------------------
__global__ void maxpool3x3s2_kernel(const float4* __restrict__ x, float4* __restrict__ y, int H, int W,
                                    int C4, int Ho, int Wo, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const long long b = r / Ho;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const float4 v = x[((b * H + iy) * W + ix) * C4 + c];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y);
                m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        y[i] = m;
    }
}

extern "C" int fgn_maxpool3x3s2_nhwc_f32(const float* x, float* y, int n_img, int H, int W, int C,
                                         hipStream_t stream) {
    if (!x || !y) return FGN_ERR_ARG;
    if (C % 4) return FGN_ERR_SHAPE;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)n_img * Ho * Wo * (C / 4);
    if (total == 0) return FGN_OK;
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), H, W, C / 4, Ho,
                       Wo, total);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ----------------------------------------------------------------------------------
// RoIAlign, average pooling, NHWC.  One workgroup per (roi, bin); lanes cover channels.
//   aligned=1, sampling_ratio=0 : mmcv.ops.RoIAlign as configured at
//                                 fgn_r50_c4_densecl.py:69-73 (fgn_roi_head.py:331,366)
//   aligned=0, sampling_ratio<=0: torchvision.ops.roi_align as called at
//                                 fgn_roi_head.py:429,432 (roi size clamped to >= 1)
// rois [R,5] = (batch_idx, x1, y1, x2, y2).  Sampling follows the upstream
// bilinear_interpolate: outside [-1,size] -> 0, coordinates clamped at 0, the last
// row/column snaps.
// ----------------------------------------------------------------------------------
struct AxisSample {
    int lo, hi;
    float l, h;
    bool valid;
};

__device__ __forceinline__ AxisSample axis_sample(float c, int size) {
    AxisSample s;
    s.valid = !(c < -1.0f || c > (float)size);
    if (c <= 0.f) c = 0.f;
    int lo = (int)c;
    int hi;
    if (lo >= size - 1) {
        hi = lo = size - 1;
        c = (float)lo;
    } else {
        hi = lo + 1;
    }
    s.lo = lo; s.hi = hi;
    s.l = c - (float)lo;
    s.h = 1.f - s.l;
    return s;
}

__global__ __launch_bounds__(256) void roi_align_kernel(const float* __restrict__ fmap,
                                                        const float* __restrict__ rois,
                                                        float* __restrict__ out,
                                                        const int32_t* __restrict__ n_rois_dev, int n_rois,
                                                        int H, int W, int C, int P, float spatial_scale,
                                                        int sampling_ratio, int aligned,
                                                        const float* __restrict__ post_shift, int relu,
                                                        const float* __restrict__ fmap2, float* __restrict__ out2, int C2,
                                                        const float* __restrict__ post_shift2, int relu2) {
    // (fmap2 / out2: an optional second map of the same spatial size pooled at the same RoIs by the same launch - the C4
    // map and the map of the shared head's commuted first conv; channel quads [C/4, (C + C2)/4) of the thread loop)
    const int bin = blockIdx.x;
    const int r = bin / (P * P);
    int nr = n_rois;
    if (n_rois_dev) nr = min(nr, *n_rois_dev);
    if (r >= nr) return;
    const int pb = bin - r * P * P;
    const int ph = pb / P, pw = pb - ph * P;

    const float* roi = rois + (size_t)r * 5;
    const int b = (int)roi[0];
    const float off = aligned ? 0.5f : 0.f;
    const float x1 = roi[1] * spatial_scale - off;
    const float y1 = roi[2] * spatial_scale - off;
    const float x2 = roi[3] * spatial_scale - off;
    const float y2 = roi[4] * spatial_scale - off;
    float rw = x2 - x1, rh = y2 - y1;
    if (!aligned) {
        rw = fmaxf(rw, 1.f);
        rh = fmaxf(rh, 1.f);
    }
    const float bin_h = rh / (float)P, bin_w = rw / (float)P;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
    const float count = (float)max(gh * gw, 1);
    const float* base1 = fmap + (size_t)b * H * W * C;
    const float* base2 = fmap2 ? fmap2 + (size_t)b * H * W * C2 : nullptr;
    const int c_all = C + (fmap2 ? C2 : 0);
    const int C1 = C;

    // The gh x gw samples of a bin form a grid and bilinear weights are separable, so the bin's value is
    //     sum_py sum_px WY[py] * WX[px] * v[py][px],   WY[py] = sum of the row weights of the samples touching row py
    // (likewise WX): (gh + 1) x (gw + 1) loads per channel quad instead of 4 * gh * gw (adaptive sampling takes up to
    // 8 x 8 samples per bin of a large RoI: 100 instead of 256 loads).  The two weight vectors are built once per bin
    // by two threads; samples outside [-1, size] carry no weight, as in the per-sample form.
    // (Measured on the 300 proposals of a cfg3 episode, two maps: 62.3 -> 55.0 us.  One workgroup per ROW of bins -
    // 2100 instead of 14 700 workgroups - was tried next and took 144 us: 10 serial items per thread.)
    constexpr int MAXS = 32;
    __shared__ float wy_sh[MAXS], wx_sh[MAXS];
    __shared__ int span_sh[4];       // y0, ny, x0, nx
    if (threadIdx.x < 2) {
        const bool is_y = threadIdx.x == 0;
        const int g = is_y ? gh : gw, size = is_y ? H : W;
        const float start = (is_y ? y1 + (float)ph * bin_h : x1 + (float)pw * bin_w), step = is_y ? bin_h : bin_w;
        float* w = is_y ? wy_sh : wx_sh;
        // (an empty sampling grid - a box of no extent - has no pixels: the bin is 0, as in the per-sample form)
        const int first = g > 0 ? axis_sample(start + 0.5f * step / (float)g, size).lo : 0;
        const int last = g > 0 ? axis_sample(start + ((float)(g - 1) + 0.5f) * step / (float)g, size).hi : -1;
        const int n = last - first + 1;
        span_sh[is_y ? 0 : 2] = first;
        span_sh[is_y ? 1 : 3] = n;
        if (n <= MAXS) {
            for (int i = 0; i < n; ++i) w[i] = 0.f;
            for (int i = 0; i < g; ++i) {
                const AxisSample sp = axis_sample(start + ((float)i + 0.5f) * step / (float)g, size);
                if (sp.valid) {
                    w[sp.lo - first] += sp.h;
                    w[sp.hi - first] += sp.l;
                }
            }
        }
    }
    __syncthreads();
    const int y0 = span_sh[0], ny = span_sh[1], x0 = span_sh[2], nx = span_sh[3];
    const bool separable = ny <= MAXS && nx <= MAXS;

    for (int cc = threadIdx.x * 4; cc < c_all; cc += blockDim.x * 4) {
        const bool snd = cc >= C1;
        const float* base = snd ? base2 : base1;
        const int c = snd ? cc - C1 : cc;
        C = snd ? C2 : C1;
        if (snd) { out = out2; post_shift = post_shift2; relu = relu2; }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (separable) {
            for (int py = 0; py < ny; ++py) {
                const float wyv = wy_sh[py];
                const float* row = base + ((size_t)(y0 + py) * W + x0) * C + c;
                float4 racc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int px = 0; px < nx; ++px) {
                    const float wv = wx_sh[px];
                    const float4 v = *reinterpret_cast<const float4*>(row + (size_t)px * C);
                    racc.x += wv * v.x; racc.y += wv * v.y; racc.z += wv * v.z; racc.w += wv * v.w;
                }
                acc.x += wyv * racc.x; acc.y += wyv * racc.y; acc.z += wyv * racc.z; acc.w += wyv * racc.w;
            }
        } else {
            for (int iy = 0; iy < gh; ++iy) {
                const float y = y1 + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
                const AxisSample sy = axis_sample(y, H);
                for (int ix = 0; ix < gw; ++ix) {
                    const float x = x1 + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw;
                    const AxisSample sx = axis_sample(x, W);
                    if (!(sy.valid && sx.valid)) continue;
                    const float w1 = sy.h * sx.h, w2 = sy.h * sx.l, w3 = sy.l * sx.h, w4 = sy.l * sx.l;
                    const float4 v1 = *reinterpret_cast<const float4*>(base + ((size_t)sy.lo * W + sx.lo) * C + c);
                    const float4 v2 = *reinterpret_cast<const float4*>(base + ((size_t)sy.lo * W + sx.hi) * C + c);
                    const float4 v3 = *reinterpret_cast<const float4*>(base + ((size_t)sy.hi * W + sx.lo) * C + c);
                    const float4 v4 = *reinterpret_cast<const float4*>(base + ((size_t)sy.hi * W + sx.hi) * C + c);
                    acc.x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
                    acc.y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
                    acc.z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
                    acc.w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
                }
            }
        }
        acc.x /= count; acc.y /= count; acc.z /= count; acc.w /= count;
        if (post_shift) {
            const float4 sh = *reinterpret_cast<const float4*>(post_shift + c);
            acc.x += sh.x; acc.y += sh.y; acc.z += sh.z; acc.w += sh.w;
        }
        if (relu) {
            acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
        }
        *reinterpret_cast<float4*>(out + ((size_t)r * P * P + pb) * C + c) = acc;
    }
}

// single-channel variant for the support masks (uint8/bool input [B,H,W]); one wavefront per bin,
// lanes stride over the (adaptive, up to ~37x37) sampling grid and combine with a shuffle reduction
__global__ __launch_bounds__(256) void roi_align_mask_kernel(const uint8_t* __restrict__ mask,
                                                             const float* __restrict__ rois,
                                                             float* __restrict__ out, int n_rois, int H, int W,
                                                             int P, float spatial_scale, int sampling_ratio,
                                                             int aligned) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= n_rois * P * P) return;
    const int r = i / (P * P);
    const int pb = i - r * P * P;
    const int ph = pb / P, pw = pb - ph * P;
    const float* roi = rois + (size_t)r * 5;
    const int b = (int)roi[0];
    const float off = aligned ? 0.5f : 0.f;
    const float x1 = roi[1] * spatial_scale - off;
    const float y1 = roi[2] * spatial_scale - off;
    const float x2 = roi[3] * spatial_scale - off;
    const float y2 = roi[4] * spatial_scale - off;
    float rw = x2 - x1, rh = y2 - y1;
    if (!aligned) {
        rw = fmaxf(rw, 1.f);
        rh = fmaxf(rh, 1.f);
    }
    const float bin_h = rh / (float)P, bin_w = rw / (float)P;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
    const float count = (float)max(gh * gw, 1);
    const uint8_t* base = mask + (size_t)b * H * W;
    float acc = 0.f;
    for (int s = lane; s < gh * gw; s += 64) {
        const int iy = s / gw, ix = s - iy * gw;
        const float y = y1 + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
        const float x = x1 + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw;
        const AxisSample sy = axis_sample(y, H);
        const AxisSample sx = axis_sample(x, W);
        if (!(sy.valid && sx.valid)) continue;
        const float v1 = base[(size_t)sy.lo * W + sx.lo] ? 1.f : 0.f;
        const float v2 = base[(size_t)sy.lo * W + sx.hi] ? 1.f : 0.f;
        const float v3 = base[(size_t)sy.hi * W + sx.lo] ? 1.f : 0.f;
        const float v4 = base[(size_t)sy.hi * W + sx.hi] ? 1.f : 0.f;
        acc += sy.h * sx.h * v1 + sy.h * sx.l * v2 + sy.l * sx.h * v3 + sy.l * sx.l * v4;
    }
    acc = wave_reduce_sum(acc);
    if (lane == 0) out[i] = acc / count;
}

extern "C" int fgn_roi_align_nhwc_f32(const float* fmap, const float* rois, float* out,
                                      const int32_t* n_rois_dev, int n_rois, int n_img, int H, int W, int C,
                                      int out_size, float spatial_scale, int sampling_ratio, int aligned,
                                      const float* post_shift, int relu, hipStream_t stream) {
    if (!fmap || !rois || !out) return FGN_ERR_ARG;
    if (C % 4 || out_size <= 0) return FGN_ERR_SHAPE;
    (void)n_img;
    if (n_rois == 0) return FGN_OK;
    const int threads = C >= 1024 ? 256 : (C >= 512 ? 128 : 64);
    hipLaunchKernelGGL(roi_align_kernel, dim3(n_rois * out_size * out_size), dim3(threads), 0, stream, fmap,
                       rois, out, n_rois_dev, n_rois, H, W, C, out_size, spatial_scale, sampling_ratio,
                       aligned, post_shift, relu, (const float*)nullptr, (float*)nullptr, 0, (const float*)nullptr, 0);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// Two maps of the same spatial size pooled at the same RoIs by one launch (fgn_roi_head.py:331,366 with the shared
// head's first 1x1 conv commuted in front of the pooling): out [R,P,P,C] from fmap, out2 [R,P,P,C2] from fmap2
// (+ post_shift2 [C2], ReLU).  Same arithmetic per channel as two fgn_roi_align_nhwc_f32 calls: identical bytes.
extern "C" int fgn_roi_align2_nhwc_f32(const float* fmap, const float* fmap2, const float* rois, float* out, float* out2,
                                       const int32_t* n_rois_dev, int n_rois, int n_img, int H, int W, int C, int C2,
                                       int out_size, float spatial_scale, int sampling_ratio, int aligned,
                                       const float* post_shift2, int relu2, hipStream_t stream) {
    if (!fmap || !fmap2 || !rois || !out || !out2) return FGN_ERR_ARG;
    if (C % 4 || C2 % 4 || out_size <= 0) return FGN_ERR_SHAPE;
    (void)n_img;
    if (n_rois == 0) return FGN_OK;
    // threads per bin, measured on the 300 proposals of a cfg3 episode (1024 + 512 channels = 384 quads; us per call):
    // 64: 92, 128: 61, 192: 52, 256: 55, 384: 67 - two whole items per thread on three waves
    const int quads = (C + C2) / 4;
    const int threads = quads % 192 == 0 ? 192 : (quads >= 256 ? 256 : (quads >= 128 ? 128 : 64));
    hipLaunchKernelGGL(roi_align_kernel, dim3(n_rois * out_size * out_size), dim3(threads), 0, stream, fmap,
                       rois, out, n_rois_dev, n_rois, H, W, C, out_size, spatial_scale, sampling_ratio,
                       aligned, (const float*)nullptr, 0, fmap2, out2, C2, post_shift2, relu2);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

extern "C" int fgn_roi_align_mask_u8(const uint8_t* mask, const float* rois, float* out, int n_rois, int n_img,
                                     int H, int W, int out_size, float spatial_scale, int sampling_ratio,
                                     int aligned, hipStream_t stream) {
    if (!mask || !rois || !out) return FGN_ERR_ARG;
    (void)n_img;
    if (n_rois == 0) return FGN_OK;
    const int total = n_rois * out_size * out_size;
    hipLaunchKernelGGL(roi_align_mask_kernel, dim3(cdiv(total, 4)), dim3(256), 0, stream, mask, rois, out,
                       n_rois, H, W, out_size, spatial_scale, sampling_ratio, aligned);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ----------------------------------------------------------------------------------
// Support-set reductions.
//   class vectors  out[g][c] = 1/(K*P) * sum_k sum_p x[g*K+k][p][c] * (w ? w[g*K+k][p] : 1)
//     - AG-RPN class-attentive vector, mean over (K,h,w)       fgn_ag_rpn_head.py:38-41
//     - mask-pooled vector, mean over (K,7,7) of feat*mask     fgn_roi_head.py:444-447
//   k-mean         out[g][p][c] = 1/K * sum_k x[g*K+k][p][c]   fgn_roi_head.py:439-442
// One workgroup per (group, 256-channel slab): lanes own 4 channels each, waves split
// the (k,p) range and combine through LDS.
// ----------------------------------------------------------------------------------
constexpr int CV_WAVES = 16;      // 12 workgroups in all at cfg3 and a serial K*P loop per wave: many waves keep it short
__global__ __launch_bounds__(64 * CV_WAVES) void class_vector_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ w,
                                                                     float* __restrict__ out, int K, int P, int C) {
    __shared__ float4 part[CV_WAVES][64];
    const int g = blockIdx.x;
    const int c = (blockIdx.y * 64 + (threadIdx.x & 63)) * 4;
    const int wv = threadIdx.x >> 6;
    const int KP = K * P;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        for (int i = wv; i < KP; i += CV_WAVES) {
            const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)g * KP + i) * C + c);
            const float s = w ? w[(size_t)g * KP + i] : 1.f;
            acc.x += v.x * s; acc.y += v.y * s; acc.z += v.z * s; acc.w += v.w * s;
        }
    }
    part[wv][threadIdx.x & 63] = acc;
    __syncthreads();
    if (wv == 0 && c < C) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < CV_WAVES; ++k) {            // fixed order
            const float4 a = part[k][threadIdx.x];
            o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        const float inv = 1.f / (float)KP;
        o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
        *reinterpret_cast<float4*>(out + (size_t)g * C + c) = o;
    }
}

extern "C" int fgn_support_class_vectors_f32(const float* x, const float* weights, float* out, int n_groups,
                                             int K, int P, int C, hipStream_t stream) {
    if (!x || !out) return FGN_ERR_ARG;
    if (C % 4) return FGN_ERR_SHAPE;
    if (n_groups == 0) return FGN_OK;
    hipLaunchKernelGGL(class_vector_kernel, dim3(n_groups, cdiv(C, 256)), dim3(64 * CV_WAVES), 0, stream, x, weights,
                       out, K, P, C);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

__global__ void kmean_kernel(const float4* __restrict__ x, float4* __restrict__ out, int K, long long PC4,
                             long long total) {
    const float inv = 1.f / (float)K;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long g = i / PC4, r = i - g * PC4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < K; ++k) {
            const float4 v = x[(g * K + k) * PC4 + r];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        out[i] = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
}

extern "C" int fgn_support_kmean_f32(const float* x, float* out, int n_groups, int K, int P, int C,
                                     hipStream_t stream) {
    if (!x || !out) return FGN_ERR_ARG;
    if (C % 4) return FGN_ERR_SHAPE;
    const long long pc4 = (long long)P * C / 4, total = pc4 * n_groups;
    if (total == 0) return FGN_OK;
    const int grid = (int)std::min<long long>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(kmean_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<float4*>(out), K, pc4, total);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// gather rows: out[i][:] = table[idx_base[i]][:]   (label -> support vector gather,
// fgn_roi_head.py:707-714; idx = label + n_ways * img)
__global__ void gather_rows_kernel(const float4* __restrict__ table, const int64_t* __restrict__ labels,
                                   const float* __restrict__ rois, float4* __restrict__ out,
                                   const int32_t* __restrict__ n_dev, int n, int n_ways, int C4) {
    int cnt = n;
    if (n_dev) cnt = min(cnt, *n_dev);
    const int i = blockIdx.x;
    if (i >= cnt) return;
    const int img = rois ? (int)rois[(size_t)i * 5] : 0;
    const long long row = labels[i] + (long long)n_ways * img;
    for (int c = threadIdx.x; c < C4; c += blockDim.x) out[(size_t)i * C4 + c] = table[row * C4 + c];
}

extern "C" int fgn_gather_support_vectors_f32(const float* table, const int64_t* labels, const float* rois,
                                              float* out, const int32_t* n_dev, int n, int n_ways, int C,
                                              hipStream_t stream) {
    if (!table || !labels || !out) return FGN_ERR_ARG;
    if (C % 4) return FGN_ERR_SHAPE;
    if (n == 0) return FGN_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(table), labels, rois, reinterpret_cast<float4*>(out),
                       n_dev, n, n_ways, C / 4);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// out[n][p][c] = x[n / div][p][c] * v[n][c]: the AG-RPN guidance multiply materialised
// (fgn_ag_rpn_head.py:44-46) so that the 3x3 RPN conv can run on the LDS-DMA kernel when it does not take the Winograd form,
// which has no register stage to scale in.  HBM-bound: 1 read of x per class + 1 write.
__global__ void scale_channels_kernel(const float4* __restrict__ x, const float4* __restrict__ v,
                                      float4* __restrict__ out, int div, long long PC4, int C4, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / PC4, r = i - n * PC4;
        const float4 a = x[(n / div) * PC4 + r];
        const float4 s = v[n * C4 + (r % C4)];
        out[i] = make_float4(a.x * s.x, a.y * s.y, a.z * s.z, a.w * s.w);
    }
}

extern "C" int fgn_scale_channels_f32(const float* x, const float* v, float* out, int n_out, int div, int P, int C,
                                      hipStream_t stream) {
    if (!x || !v || !out) return FGN_ERR_ARG;
    if (C % 4 || div < 1) return FGN_ERR_SHAPE;
    const long long pc4 = (long long)P * C / 4, total = pc4 * n_out;
    if (total == 0) return FGN_OK;
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(scale_channels_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<const float4*>(v), reinterpret_cast<float4*>(out), div, pc4, C / 4, total);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
