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
// The F(4x4) transform kernels are templated on the channel vector width VEC of a thread (1, 2 or 4 floats): a thread owns
// (tile, VEC channels), loads the 6x6 patch / the 36 products and stores 36 / 16 values.  Most layers of an episode
// are small (70 k .. 800 k (tile, channel) elements): with 16-byte vectors they start 17 k .. 200 k threads on a chip
// that holds 524 k, and the kernel is a chain of dependent memory round trips - latency-bound at 2 TB/s.  With one
// channel per thread the same layer runs 4x the waves (a wave still moves 256 contiguous bytes per instruction) at a
// quarter of the registers; 16-byte vectors remain for the launches that fill the chip anyway (batched AG-RPN maps).
template <int V>
struct VF {
    typedef float type __attribute__((ext_vector_type(V)));
};
template <int V>
using vf = typename VF<V>::type;

template <int V>
__device__ __forceinline__ vf<V> vfma(float k, vf<V> a, vf<V> acc) {
    vf<V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = fmaf(k, a[i], acc[i]);
    return r;
}
template <int V>
__device__ __forceinline__ vf<V> vzero() {
    vf<V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = 0.f;
    return r;
}
template <int V>
__device__ __forceinline__ vf<V> vrelu(vf<V> a) {
#pragma unroll
    for (int i = 0; i < V; ++i) a[i] = fmaxf(a[i], 0.f);
    return a;
}
// r = B^T d for one column / row of six
template <int V>
__device__ __forceinline__ void wg4_bt(const vf<V> (&d)[6], vf<V> (&r)[6]) {
    r[0] = (vfma<V>(1.5f, d[3] - d[1], vfma<V>(-2.f, d[2], d[0]))) + d[4];
    r[1] = vfma<V>(2.5f, d[3], vfma<V>(0.5f, d[2], d[4] - d[1]));
    r[2] = vfma<V>(0.5f, d[3], vfma<V>(-2.5f, d[2], d[1] + d[4]));
    r[3] = vfma<V>(2.f, d[3] - d[1], d[4] - d[2]);
    r[4] = vfma<V>(0.5f, d[1] - d[3], d[4] - d[2]);
    r[5] = (vfma<V>(1.5f, d[4] - d[2], vfma<V>(-2.f, d[3], d[1]))) + d[5];
}
// r = A^T m for one column / row of six -> four
template <int V>
__device__ __forceinline__ void wg4_at(const vf<V> (&m)[6], vf<V> (&r)[4]) {
    const vf<V> s12 = m[1] + m[2], d12 = m[1] - m[2];
    r[0] = (m[0] + s12) + (m[3] + m[4]);
    r[1] = vfma<V>(-2.f, m[4], vfma<V>(0.5f, m[3], d12));
    r[2] = vfma<V>(4.f, m[4], vfma<V>(0.25f, m[3], s12));
    r[3] = (vfma<V>(-8.f, m[4], vfma<V>(0.125f, m[3], d12))) + m[5];
}

// A second tensor (`seg`: its own image count and size, no input scale, no device count) may follow the first in the
// tile index space - tiles [tiles0, ...) of V: the query and the support maps of one backbone layer share one launch.
struct WgSeg {
    const float* x;       // second input / output tensor (nullptr: none)
    int n_img, H, W, ty, tx;
    int tiles0;           // tiles of the first tensor
};

template <int V, bool EAGER>
__global__ __launch_bounds__(256) void wg4_input_kernel(const float* __restrict__ x_, const float* __restrict__ in_scale_,
                                                        float* __restrict__ V_, const int32_t* __restrict__ n_img_dev,
                                                        int n_img, int a_img_div, int H, int W, int CV, int ty, int tx,
                                                        int t_pad, long long total, const WgSeg seg) {
    typedef vf<V> T;
    const T* __restrict__ x = reinterpret_cast<const T*>(x_);
    const T* __restrict__ in_scale = reinterpret_cast<const T*>(in_scale_);
    T* __restrict__ Vo = reinterpret_cast<T*>(V_);
    if (n_img_dev) n_img = min(n_img, *n_img_dev);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV);
        const int t = (int)(i / CV);
        const bool second = seg.x && t >= seg.tiles0;
        if (second) {
            x = reinterpret_cast<const T*>(seg.x);
            H = seg.H; W = seg.W; ty = seg.ty; tx = seg.tx; n_img = seg.n_img; a_img_div = 1;
        }
        const int tl = second ? t - seg.tiles0 : t;
        const int xx = tl % tx;
        const int r = tl / tx;
        const int yy = r % ty;
        const int img = r / ty;
        if (img >= n_img) {
            if (seg.x) continue;
            break;                                    // single tensor: images are the slowest index, nothing is left
        }
        const T* src = x + (size_t)(img / a_img_div) * H * W * CV + c;
        const int iy0 = 4 * yy - 1, ix0 = 4 * xx - 1;
        T s;
        if (in_scale) s = in_scale[(size_t)img * CV + c];
        else {
#pragma unroll
            for (int k = 0; k < V; ++k) s[k] = 1.f;
        }
        T tt[6][6];
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
                                   ? src[((size_t)iy * W + ix) * CV] : vzero<V>();
                }
            }
#pragma unroll
            for (int b = 0; b < 6; ++b) {             // tt[.][b] = B^T (d * s)[.][b], in place
                T d[6], rr[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) d[a] = tt[a][b] * s;
                wg4_bt<V>(d, rr);
#pragma unroll
                for (int a = 0; a < 6; ++a) tt[a][b] = rr[a];
            }
        } else {
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const int ix = ix0 + b;
                T d[6], rr[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const int iy = iy0 + a;
                    d[a] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                               ? src[((size_t)iy * W + ix) * CV] * s : vzero<V>();
                }
                wg4_bt<V>(d, rr);
#pragma unroll
                for (int a = 0; a < 6; ++a) tt[a][b] = rr[a];
            }
        }
        T* dst = Vo + (size_t)t * CV + c;
        const size_t gs = (size_t)t_pad * CV;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            T rr[6];
            wg4_bt<V>(tt[a], rr);                      // (B^T d) B, row a
#pragma unroll
            for (int b = 0; b < 6; ++b) dst[(a * 6 + b) * gs] = rr[b];
        }
    }
}

// Mo [36][t_pad][C] -> y [n_img, H, W, C] = A^T Mo A + shift (ReLU); outputs beyond H / W are dropped
template <int V>
__global__ __launch_bounds__(256) void wg4_output_kernel(const float* __restrict__ Mo_, float* __restrict__ y_,
                                                         const float* __restrict__ shift_,
                                                         const int32_t* __restrict__ n_img_dev, int n_img, int H, int W,
                                                         int CV, int ty, int tx, int t_pad, int relu, long long total,
                                                         const WgSeg seg) {
    typedef vf<V> T;
    const T* __restrict__ Mo = reinterpret_cast<const T*>(Mo_);
    const T* __restrict__ shift = reinterpret_cast<const T*>(shift_);
    T* __restrict__ y = reinterpret_cast<T*>(y_);
    if (n_img_dev) n_img = min(n_img, *n_img_dev);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV);
        const int t = (int)(i / CV);
        const bool second = seg.x && t >= seg.tiles0;
        if (second) {
            y = reinterpret_cast<T*>(const_cast<float*>(seg.x));
            H = seg.H; W = seg.W; ty = seg.ty; tx = seg.tx; n_img = seg.n_img;
        }
        const int tl = second ? t - seg.tiles0 : t;
        const int xx = tl % tx;
        const int r = tl / tx;
        const int yy = r % ty;
        const int img = r / ty;
        if (img >= n_img) {
            if (seg.x) continue;
            break;
        }
        const T* src = Mo + (size_t)t * CV + c;
        const size_t gs = (size_t)t_pad * CV;
        T mm[6][6];                                    // all 36 loads in flight at once
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) mm[a][b] = src[(a * 6 + b) * gs];
        T st[4][6];                                    // st[i][b] = (A^T m)[i][b]
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            T m[6], rr[4];
#pragma unroll
            for (int a = 0; a < 6; ++a) m[a] = mm[a][b];
            wg4_at<V>(m, rr);
#pragma unroll
            for (int k = 0; k < 4; ++k) st[k][b] = rr[k];
        }
        const T sh = shift ? shift[c] : vzero<V>();
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oy = 4 * yy + a;
            T rr[4];
            wg4_at<V>(st[a], rr);
            if (oy >= H) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ox = 4 * xx + b;
                if (ox >= W) continue;
                T v = rr[b] + sh;
                if (relu) v = vrelu<V>(v);
                y[(((size_t)img * H + oy) * W + ox) * CV + c] = v;
            }
        }
    }
}

// Channel vector width of a thread for `elems` = tiles * channels (tile, channel) pairs, and whether the input transform
// issues its 36 loads up front.  Measured on MI355X at the cfg3 layer shapes (tools/wg4_ab.py, r03): loads up front wins
// at every size and width (AG-RPN map 55 -> 36 us at 16-byte vectors, shared_head on 300 RoIs 36 -> 20, layer1 31 -> 14:
// the column-by-column form is a chain of six dependent memory round trips); with it, 8-byte vectors are best or within
// 1.5 us of the best at every layer of an episode (34.0 / 21.7 / 16.5 / 13.7 us on the four large ones, 6-7 us - the
// floor of a launch whose every thread does one load and one store round trip - on the rest); 16-byte vectors are kept
// for launches with several million elements (batched AG-RPN maps).  (The A/B was driven by environment variables, removed
// since: the library reads none.)
static int wg4_vec(long long elems) { return elems >= (4ll << 20) ? 4 : 2; }
static bool wg4_eager(long long) { return true; }
// vec * 10 + eager of the input transform / vec * 10 of the output transform for a layer (lets a profiler name the
// kernel instance a launch uses)
extern "C" int fgn_winograd4_variant(int tiles_total, int C, int is_output) {
    const int v = wg4_vec((long long)tiles_total * C);
    return v * 10 + ((!is_output && wg4_eager((long long)tiles_total * (C / v))) ? 1 : 0);
}

#define WG4_IN_LAUNCH(VV, EE)                                                                                          \
    FGN_LAUNCH_TIMED((wg4_input_kernel<VV, EE>), dim3(grid), dim3(256), 0, stream, x, in_scale, V, n_img_dev, n_img,  \
                     a_img_div, H, W, C / VV, ty, tx, t_pad, total, seg)

static int wg4_input_launch(const float* x, const float* in_scale, float* V, const int32_t* n_img_dev, int n_img,
                            int a_img_div, int H, int W, int C, int t_pad, const float* x1, int n_img1, int H1, int W1,
                            hipStream_t stream) {
    if (!x || !V) return FGN_ERR_ARG;
    if (n_img <= 0) return FGN_OK;
    const int ty = (H + 3) / 4, tx = (W + 3) / 4;
    WgSeg seg;
    seg.x = x1; seg.n_img = n_img1; seg.H = H1; seg.W = W1; seg.ty = (H1 + 3) / 4; seg.tx = (W1 + 3) / 4;
    seg.tiles0 = n_img * ty * tx;
    const long long tiles_all = (long long)seg.tiles0 + (x1 ? (long long)n_img1 * seg.ty * seg.tx : 0);
    if (C % 4 != 0 || a_img_div < 1 || H < 1 || W < 1 || tiles_all > t_pad) return FGN_ERR_SHAPE;
    if (x1 && (in_scale || n_img_dev || a_img_div != 1 || n_img1 < 1 || H1 < 1 || W1 < 1)) return FGN_ERR_SHAPE;
    const int vec = wg4_vec(tiles_all * C);
    const long long total = tiles_all * (C / vec);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    const bool eager = wg4_eager(total);
    switch (vec * 10 + (eager ? 1 : 0)) {
        case 11: WG4_IN_LAUNCH(1, true); break;
        case 10: WG4_IN_LAUNCH(1, false); break;
        case 21: WG4_IN_LAUNCH(2, true); break;
        case 20: WG4_IN_LAUNCH(2, false); break;
        case 41: WG4_IN_LAUNCH(4, true); break;
        default: WG4_IN_LAUNCH(4, false); break;
    }
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

extern "C" int fgn_winograd4_input_f32(const float* x, const float* in_scale, float* V, const int32_t* n_img_dev,
                                       int n_img, int a_img_div, int H, int W, int C, int t_pad, hipStream_t stream) {
    return wg4_input_launch(x, in_scale, V, n_img_dev, n_img, a_img_div, H, W, C, t_pad, nullptr, 0, 1, 1, stream);
}
// two tensors (e.g. the query map and the support maps of a backbone layer) into consecutive tile ranges of one V
extern "C" int fgn_winograd4_input2_f32(const float* x0, int n_img0, int H0, int W0, const float* x1, int n_img1, int H1,
                                        int W1, float* V, int C, int t_pad, hipStream_t stream) {
    if (!x1) return FGN_ERR_ARG;
    return wg4_input_launch(x0, nullptr, V, nullptr, n_img0, 1, H0, W0, C, t_pad, x1, n_img1, H1, W1, stream);
}

#define WG4_OUT_LAUNCH(VV)                                                                                             \
    FGN_LAUNCH_TIMED((wg4_output_kernel<VV>), dim3(grid), dim3(256), 0, stream, Mo, y, shift, n_img_dev, n_img, H, W, \
                     C / VV, ty, tx, t_pad, relu, total, seg)

static int wg4_output_launch(const float* Mo, float* y, const float* shift, const int32_t* n_img_dev, int n_img, int H,
                             int W, int C, int t_pad, int relu, float* y1, int n_img1, int H1, int W1, hipStream_t stream) {
    if (!Mo || !y) return FGN_ERR_ARG;
    if (n_img <= 0) return FGN_OK;
    const int ty = (H + 3) / 4, tx = (W + 3) / 4;
    WgSeg seg;
    seg.x = y1; seg.n_img = n_img1; seg.H = H1; seg.W = W1; seg.ty = (H1 + 3) / 4; seg.tx = (W1 + 3) / 4;
    seg.tiles0 = n_img * ty * tx;
    const long long tiles_all = (long long)seg.tiles0 + (y1 ? (long long)n_img1 * seg.ty * seg.tx : 0);
    if (C % 4 != 0 || tiles_all > t_pad) return FGN_ERR_SHAPE;
    if (y1 && (n_img_dev || n_img1 < 1 || H1 < 1 || W1 < 1)) return FGN_ERR_SHAPE;
    const int vec = wg4_vec(tiles_all * C);
    const long long total = tiles_all * (C / vec);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    switch (vec) {
        case 1: WG4_OUT_LAUNCH(1); break;
        case 2: WG4_OUT_LAUNCH(2); break;
        default: WG4_OUT_LAUNCH(4); break;
    }
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

extern "C" int fgn_winograd4_output_f32(const float* Mo, float* y, const float* shift, const int32_t* n_img_dev,
                                        int n_img, int H, int W, int C, int t_pad, int relu, hipStream_t stream) {
    return wg4_output_launch(Mo, y, shift, n_img_dev, n_img, H, W, C, t_pad, relu, nullptr, 0, 1, 1, stream);
}
extern "C" int fgn_winograd4_output2_f32(const float* Mo, const float* shift, float* y0, int n_img0, int H0, int W0,
                                         float* y1, int n_img1, int H1, int W1, int C, int t_pad, int relu,
                                         hipStream_t stream) {
    if (!y1) return FGN_ERR_ARG;
    return wg4_output_launch(Mo, y0, shift, nullptr, n_img0, H0, W0, C, t_pad, relu, y1, n_img1, H1, W1, stream);
}

// ----------------------------------------------------------------------------------------------
// Weight side of the Winograd forms: U[g = a * R + b][co][ci] = sum_ij G[a][i] * w[co][ci][i][j] * G[b][j], R = m + 2,
// computed in fp64 from the fp32 weights and rounded once - the transform `ops.pack_winograd` does with torch ops on
// the host when a model is built.  A training loop re-derives U from the updated master weights after EVERY step
// (fgn_amd.train.Trainer.refresh): there it was ~40 small torch kernels and 300 MB of fp64 intermediates per layer;
// here one launch writes U in place (rows [cout, cout_pad) of a group keep the zeros of the first pack).
// G [R][3] fp64 is handed over by the caller (the host's table is the single definition of the interpolation points).
// ----------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void wg_pack_weights_kernel(const float* __restrict__ w, const double* __restrict__ G,
                                                              float* __restrict__ U, int cout, int cin, int cout_pad) {
    __shared__ double g[R][3];
    if (threadIdx.x < R * 3) g[threadIdx.x / 3][threadIdx.x % 3] = G[threadIdx.x];
    __syncthreads();
    const long long total = (long long)cout * cin;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(q / cin), ci = (int)(q - (long long)co * cin);
        double k[3][3];
#pragma unroll
        for (int i = 0; i < 9; ++i) k[i / 3][i % 3] = (double)w[q * 9 + i];
        double t[R][3];
#pragma unroll
        for (int a = 0; a < R; ++a)
#pragma unroll
            for (int j = 0; j < 3; ++j) t[a][j] = g[a][0] * k[0][j] + g[a][1] * k[1][j] + g[a][2] * k[2][j];
#pragma unroll
        for (int a = 0; a < R; ++a)
#pragma unroll
            for (int b = 0; b < R; ++b)
                U[((size_t)(a * R + b) * cout_pad + co) * cin + ci] =
                    (float)(t[a][0] * g[b][0] + t[a][1] * g[b][1] + t[a][2] * g[b][2]);
    }
}

extern "C" int fgn_winograd_pack_weights_f32(const float* w, const double* G, float* U, int cout, int cin, int cout_pad,
                                             int m, hipStream_t stream) {
    if (!w || !G || !U) return FGN_ERR_ARG;
    if ((m != 2 && m != 4) || cout < 1 || cin < 1 || cout_pad < cout) return FGN_ERR_SHAPE;
    const long long total = (long long)cout * cin;
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 16);
    if (m == 4)
        hipLaunchKernelGGL(wg_pack_weights_kernel<6>, dim3(grid), dim3(256), 0, stream, w, G, U, cout, cin, cout_pad);
    else
        hipLaunchKernelGGL(wg_pack_weights_kernel<4>, dim3(grid), dim3(256), 0, stream, w, G, U, cout, cin, cout_pad);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
