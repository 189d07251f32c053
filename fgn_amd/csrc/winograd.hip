// Winograd transforms for the 3x3 / stride 1 / pad 1 convolutions whose reduction is deep
// enough to be MFMA-bound (AG-RPN conv, fgn_ag_rpn_head.py:48; the 3x3 of the shared_head bottlenecks,
// fgn_roi_head.py:236).  The convolution becomes
//     V = B^T d B   (input transform,  this file: HBM-bound, 16 B per lane)
//     Mo[g] = V[g] U[g]^T, g = 0..15 / 0..35   (one grouped GEMM launch, conv_igemm.hip)
//     y = A^T Mo A + shift, ReLU       (output transform, this file)
// with 2.25x fewer multiply-adds than the direct form.  fp32 throughout; F(2x2,3x3) uses only
// +-1 and 1/2 coefficients, its rounding error stays within a few ulp of the direct sum.
// Tile t = (img * ty + y) * tx + x covers outputs [2y, 2y+2) x [2x, 2x+2); V / Mo are laid out
// [16][t_pad][C] so each of the 16 positions is a contiguous row-major GEMM operand.
#include "common.h"
#include <cstdlib>

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

// x [n_in, H, W, C]; logical image i reads x[i / a_img_div], scaled per channel by in_scale[i][c]
// (the AG-RPN guidance multiply, fgn_ag_rpn_head.py:44, fused here).
__global__ __launch_bounds__(256) void wg_input_kernel(const float4* __restrict__ x, const float4* __restrict__ in_scale,
                                                       float4* __restrict__ V, const int32_t* __restrict__ n_img_dev,
                                                       int n_img, int a_img_div, int H, int W, int C4, int ty, int tx,
                                                       int t_pad, long long total) {
    if (n_img_dev) n_img = min(n_img, *n_img_dev);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const int t = (int)(i / C4);
        const int xx = t % tx;
        const int r = t / tx;
        const int yy = r % ty;
        const int img = r / ty;
        if (img >= n_img) break;                      // images are the slowest index: nothing left for this thread
        const float4* src = x + (size_t)(img / a_img_div) * H * W * C4 + c;
        const int iy0 = 2 * yy - 1, ix0 = 2 * xx - 1;
        float4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int iy = iy0 + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ix = ix0 + b;
                d[a][b] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                              ? src[((size_t)iy * W + ix) * C4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (in_scale) {
            const float4 s = in_scale[(size_t)img * C4 + c];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) d[a][b] = f4mul(d[a][b], s);
        }
        float4 tt[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {                 // B^T d  (columns)
            tt[0][b] = f4sub(d[0][b], d[2][b]);
            tt[1][b] = f4add(d[1][b], d[2][b]);
            tt[2][b] = f4sub(d[2][b], d[1][b]);
            tt[3][b] = f4sub(d[1][b], d[3][b]);
        }
        float4* dst = V + (size_t)t * C4 + c;
        const size_t gs = (size_t)t_pad * C4;
#pragma unroll
        for (int a = 0; a < 4; ++a) {                 // (B^T d) B  (rows)
            dst[(a * 4 + 0) * gs] = f4sub(tt[a][0], tt[a][2]);
            dst[(a * 4 + 1) * gs] = f4add(tt[a][1], tt[a][2]);
            dst[(a * 4 + 2) * gs] = f4sub(tt[a][2], tt[a][1]);
            dst[(a * 4 + 3) * gs] = f4sub(tt[a][1], tt[a][3]);
        }
    }
}

// Mo [16][t_pad][C] -> y [n_img, H, W, C] = A^T Mo A + shift (ReLU); odd H / W drop the last row / column
__global__ __launch_bounds__(256) void wg_output_kernel(const float4* __restrict__ Mo, float4* __restrict__ y,
                                                        const float4* __restrict__ shift,
                                                        const int32_t* __restrict__ n_img_dev, int n_img, int H, int W,
                                                        int C4, int ty, int tx, int t_pad, int relu, long long total) {
    if (n_img_dev) n_img = min(n_img, *n_img_dev);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const int t = (int)(i / C4);
        const int xx = t % tx;
        const int r = t / tx;
        const int yy = r % ty;
        const int img = r / ty;
        if (img >= n_img) break;
        const float4* src = Mo + (size_t)t * C4 + c;
        const size_t gs = (size_t)t_pad * C4;
        float4 s0[4], s1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {                 // A^T m (columns)
            const float4 m0 = src[(0 * 4 + b) * gs], m1 = src[(1 * 4 + b) * gs], m2 = src[(2 * 4 + b) * gs],
                         m3 = src[(3 * 4 + b) * gs];
            s0[b] = f4add(f4add(m0, m1), m2);
            s1[b] = f4sub(f4sub(m1, m2), m3);
        }
        float4 o[2][2];
        o[0][0] = f4add(f4add(s0[0], s0[1]), s0[2]);
        o[0][1] = f4sub(f4sub(s0[1], s0[2]), s0[3]);
        o[1][0] = f4add(f4add(s1[0], s1[1]), s1[2]);
        o[1][1] = f4sub(f4sub(s1[1], s1[2]), s1[3]);
        const float4 sh = shift ? shift[c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = 2 * yy + a;
            if (oy >= H) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ox = 2 * xx + b;
                if (ox >= W) continue;
                float4 v = f4add(o[a][b], sh);
                if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                y[(((size_t)img * H + oy) * W + ox) * C4 + c] = v;
            }
        }
    }
}

extern "C" int fgn_winograd_input_f32(const float* x, const float* in_scale, float* V, const int32_t* n_img_dev,
                                      int n_img, int a_img_div, int H, int W, int C, int t_pad, hipStream_t stream) {
    if (!x || !V) return FGN_ERR_ARG;
    if (n_img <= 0) return FGN_OK;
    const int ty = (H + 1) / 2, tx = (W + 1) / 2;
    if (C % 4 != 0 || a_img_div < 1 || H < 1 || W < 1 || (long long)n_img * ty * tx > t_pad) return FGN_ERR_SHAPE;
    const long long total = (long long)n_img * ty * tx * (C / 4);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    FGN_LAUNCH_TIMED(wg_input_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<const float4*>(in_scale), reinterpret_cast<float4*>(V), n_img_dev, n_img,
                       a_img_div, H, W, C / 4, ty, tx, t_pad, total);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

extern "C" int fgn_winograd_output_f32(const float* Mo, float* y, const float* shift, const int32_t* n_img_dev,
                                       int n_img, int H, int W, int C, int t_pad, int relu, hipStream_t stream) {
    if (!Mo || !y) return FGN_ERR_ARG;
    if (n_img <= 0) return FGN_OK;
    const int ty = (H + 1) / 2, tx = (W + 1) / 2;
    if (C % 4 != 0 || (long long)n_img * ty * tx > t_pad) return FGN_ERR_SHAPE;
    const long long total = (long long)n_img * ty * tx * (C / 4);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    FGN_LAUNCH_TIMED(wg_output_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(Mo),
                       reinterpret_cast<float4*>(y), reinterpret_cast<const float4*>(shift), n_img_dev, n_img, H, W,
                       C / 4, ty, tx, t_pad, relu, total);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ------------------------------------------------------------------------------------------------
// F(4x4, 3x3): 36 products per 4x4 outputs instead of 144 (4x fewer multiply-adds than the direct form, 1.78x fewer
// than F(2x2,3x3)); V / Mo are [36][t_pad][C], 2.25 values per pixel instead of 4.  Cook-Toom with the interpolation
// points {0, 1, -1, 1/2, -2, inf}: every coefficient of B^T and A^T is a small dyadic rational (exact in fp32) and
// the point set keeps the fp32 error of a 512-deep 3x3 layer at 2.5e-6 of the output scale (the textbook points
// {0, +-1, +-2} give 7e-6, F(2x2) 4e-7, the direct form 2e-7); end to end the detector's scores move by 1.6e-6 and
// its mask probabilities by 2e-5 (tests/test_hip_e2e.py holds them to 1e-4).
//   B^T = [ 1 -3/2  -2   3/2   1   0 ]      A^T = [ 1  1   1   1     1   0 ]
//         [ 0  -1   1/2  5/2   1   0 ]            [ 0  1  -1   1/2  -2   0 ]
//         [ 0   1  -5/2  1/2   1   0 ]            [ 0  1   1   1/4   4   0 ]
//         [ 0  -2   -1    2    1   0 ]            [ 0  1  -1   1/8  -8   1 ]
//         [ 0  1/2  -1  -1/2   1   0 ]
//         [ 0   1  -3/2  -2   3/2  1 ]      (G, with thirds and fifteenths, is applied to the weights on the host in fp64)
// Tile t = (img * ty + y) * tx + x covers outputs [4y, 4y+4) x [4x, 4x+4) and reads inputs [4y-1, 4y+5) x [4x-1, 4x+5).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 f4fma(float k, float4 a, float4 acc) {
    return make_float4(fmaf(k, a.x, acc.x), fmaf(k, a.y, acc.y), fmaf(k, a.z, acc.z), fmaf(k, a.w, acc.w));
}
// r = B^T d for one column / row of six
__device__ __forceinline__ void wg4_bt(const float4 (&d)[6], float4 (&r)[6]) {
    r[0] = f4add(f4fma(1.5f, f4sub(d[3], d[1]), f4fma(-2.f, d[2], d[0])), d[4]);
    r[1] = f4add(f4fma(2.5f, d[3], f4fma(0.5f, d[2], f4sub(d[4], d[1]))), make_float4(0.f, 0.f, 0.f, 0.f));
    r[2] = f4fma(0.5f, d[3], f4fma(-2.5f, d[2], f4add(d[1], d[4])));
    r[3] = f4fma(2.f, f4sub(d[3], d[1]), f4sub(d[4], d[2]));
    r[4] = f4fma(0.5f, f4sub(d[1], d[3]), f4sub(d[4], d[2]));
    r[5] = f4add(f4fma(1.5f, f4sub(d[4], d[2]), f4fma(-2.f, d[3], d[1])), d[5]);
}
// r = A^T m for one column / row of six -> four
__device__ __forceinline__ void wg4_at(const float4 (&m)[6], float4 (&r)[4]) {
    const float4 s12 = f4add(m[1], m[2]), d12 = f4sub(m[1], m[2]);
    r[0] = f4add(f4add(m[0], s12), f4add(m[3], m[4]));
    r[1] = f4fma(-2.f, m[4], f4fma(0.5f, m[3], d12));
    r[2] = f4fma(4.f, m[4], f4fma(0.25f, m[3], s12));
    r[3] = f4add(f4fma(-8.f, m[4], f4fma(0.125f, m[3], d12)), m[5]);
}

template <bool EAGER>
__global__ __launch_bounds__(256) void wg4_input_kernel(const float4* __restrict__ x, const float4* __restrict__ in_scale,
                                                        float4* __restrict__ V, const int32_t* __restrict__ n_img_dev,
                                                        int n_img, int a_img_div, int H, int W, int C4, int ty, int tx,
                                                        int t_pad, long long total) {
    if (n_img_dev) n_img = min(n_img, *n_img_dev);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const int t = (int)(i / C4);
        const int xx = t % tx;
        const int r = t / tx;
        const int yy = r % ty;
        const int img = r / ty;
        if (img >= n_img) break;
        const float4* src = x + (size_t)(img / a_img_div) * H * W * C4 + c;
        const int iy0 = 4 * yy - 1, ix0 = 4 * xx - 1;
        const float4 s = in_scale ? in_scale[(size_t)img * C4 + c] : make_float4(1.f, 1.f, 1.f, 1.f);
        float4 tt[6][6];
        if (EAGER) {
            // all 36 loads are issued before the first use (small layers are latency-bound: six dependent batches
            // of six loads cost six memory round trips)
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const int iy = iy0 + a;
#pragma unroll
                for (int b = 0; b < 6; ++b) {
                    const int ix = ix0 + b;
                    tt[a][b] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                                   ? src[((size_t)iy * W + ix) * C4] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int b = 0; b < 6; ++b) {             // tt[.][b] = B^T (d * s)[.][b], in place
                float4 d[6], rr[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) d[a] = f4mul(tt[a][b], s);
                wg4_bt(d, rr);
#pragma unroll
                for (int a = 0; a < 6; ++a) tt[a][b] = rr[a];
            }
        } else {
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const int ix = ix0 + b;
                float4 d[6], rr[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const int iy = iy0 + a;
                    d[a] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                               ? f4mul(src[((size_t)iy * W + ix) * C4], s) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                wg4_bt(d, rr);
#pragma unroll
                for (int a = 0; a < 6; ++a) tt[a][b] = rr[a];
            }
        }
        float4* dst = V + (size_t)t * C4 + c;
        const size_t gs = (size_t)t_pad * C4;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            float4 rr[6];
            wg4_bt(tt[a], rr);                         // (B^T d) B, row a
#pragma unroll
            for (int b = 0; b < 6; ++b) dst[(a * 6 + b) * gs] = rr[b];
        }
    }
}

// Mo [36][t_pad][C] -> y [n_img, H, W, C] = A^T Mo A + shift (ReLU); outputs beyond H / W are dropped
__global__ __launch_bounds__(256) void wg4_output_kernel(const float4* __restrict__ Mo, float4* __restrict__ y,
                                                         const float4* __restrict__ shift,
                                                         const int32_t* __restrict__ n_img_dev, int n_img, int H, int W,
                                                         int C4, int ty, int tx, int t_pad, int relu, long long total) {
    if (n_img_dev) n_img = min(n_img, *n_img_dev);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const int t = (int)(i / C4);
        const int xx = t % tx;
        const int r = t / tx;
        const int yy = r % ty;
        const int img = r / ty;
        if (img >= n_img) break;
        const float4* src = Mo + (size_t)t * C4 + c;
        const size_t gs = (size_t)t_pad * C4;
        float4 mm[6][6];                               // all 36 loads in flight at once
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) mm[a][b] = src[(a * 6 + b) * gs];
        float4 st[4][6];                               // st[i][b] = (A^T m)[i][b]
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            float4 m[6], rr[4];
#pragma unroll
            for (int a = 0; a < 6; ++a) m[a] = mm[a][b];
            wg4_at(m, rr);
#pragma unroll
            for (int k = 0; k < 4; ++k) st[k][b] = rr[k];
        }
        const float4 sh = shift ? shift[c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oy = 4 * yy + a;
            float4 rr[4];
            wg4_at(st[a], rr);
            if (oy >= H) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ox = 4 * xx + b;
                if (ox >= W) continue;
                float4 v = f4add(rr[b], sh);
                if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                y[(((size_t)img * H + oy) * W + ox) * C4 + c] = v;
            }
        }
    }
}

extern "C" int fgn_winograd4_input_f32(const float* x, const float* in_scale, float* V, const int32_t* n_img_dev,
                                       int n_img, int a_img_div, int H, int W, int C, int t_pad, hipStream_t stream) {
    if (!x || !V) return FGN_ERR_ARG;
    if (n_img <= 0) return FGN_OK;
    const int ty = (H + 3) / 4, tx = (W + 3) / 4;
    if (C % 4 != 0 || a_img_div < 1 || H < 1 || W < 1 || (long long)n_img * ty * tx > t_pad) return FGN_ERR_SHAPE;
    const long long total = (long long)n_img * ty * tx * (C / 4);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    // small (latency-bound) layers issue all 36 loads first; large (bandwidth-bound) ones load column by column at twice
    // the occupancy (measured, tools/wg4_ab.py: 10.3 -> 8.6 us on layer3, 39 -> 45 us on the AG-RPN map)
    static const int eager_thr = getenv("FGN_WG4_EAGER") ? atoi(getenv("FGN_WG4_EAGER")) : 64000;
    if (total < (long long)eager_thr)
        FGN_LAUNCH_TIMED(wg4_input_kernel<true>, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(x),
                         reinterpret_cast<const float4*>(in_scale), reinterpret_cast<float4*>(V), n_img_dev, n_img,
                         a_img_div, H, W, C / 4, ty, tx, t_pad, total);
    else
        FGN_LAUNCH_TIMED(wg4_input_kernel<false>, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(x),
                         reinterpret_cast<const float4*>(in_scale), reinterpret_cast<float4*>(V), n_img_dev, n_img,
                         a_img_div, H, W, C / 4, ty, tx, t_pad, total);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

extern "C" int fgn_winograd4_output_f32(const float* Mo, float* y, const float* shift, const int32_t* n_img_dev,
                                        int n_img, int H, int W, int C, int t_pad, int relu, hipStream_t stream) {
    if (!Mo || !y) return FGN_ERR_ARG;
    if (n_img <= 0) return FGN_OK;
    const int ty = (H + 3) / 4, tx = (W + 3) / 4;
    if (C % 4 != 0 || (long long)n_img * ty * tx > t_pad) return FGN_ERR_SHAPE;
    const long long total = (long long)n_img * ty * tx * (C / 4);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    FGN_LAUNCH_TIMED(wg4_output_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(Mo),
                     reinterpret_cast<float4*>(y), reinterpret_cast<const float4*>(shift), n_img_dev, n_img, H, W,
                     C / 4, ty, tx, t_pad, relu, total);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
