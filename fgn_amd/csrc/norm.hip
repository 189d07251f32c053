// GroupNorm and the average-pool shortcut of the from-scratch backbone variant
// (fgn_r50_c4_scratch.py:16-23: deep_stem, avg_down, norm_cfg GN(32)).  GroupNorm statistics
// depend on the whole image, so unlike eval-mode BatchNorm it cannot be folded into the conv
// epilogue: three HBM-bound passes per layer (partial sums, finalise, apply), NHWC fp32, every
// lane moves 16 B.  Sums are fp32 per lane over <= a few hundred values, fp64 across lanes /
// chunks (deterministic: fixed chunking, no atomics).
#include "common.h"

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_CHUNKS = 128;

static inline int gn_chunks(int HW) { return std::max(1, std::min(GN_MAX_CHUNKS, cdiv(HW, 32))); }

// partial[(n*chunks + chunk)*G + g] = (sum, sum of squares) of the chunk's pixels x the group's channels
__global__ __launch_bounds__(GN_THREADS) void gn_partial_kernel(const float4* __restrict__ x,
                                                                double2* __restrict__ partial, int HW, int C,
                                                                int G, int chunks) {
    __shared__ float2 red[GN_THREADS * 4];
    const int qc = C >> 2;                       // float4 columns per pixel
    const int stripes = GN_THREADS / qc;          // pixel stripes of this block (>= 1: C <= 1024)
    const int t = threadIdx.x;
    const int cq = t % qc, st = t / qc;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int per = (HW + chunks - 1) / chunks;
    const int p0 = chunk * per, p1 = min(HW, p0 + per);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    if (st < stripes) {
        const float4* base = x + (size_t)n * HW * qc + cq;
        for (int p = p0 + st; p < p1; p += stripes) {
            const float4 v = base[(size_t)p * qc];
            s[0] += v.x; q[0] += v.x * v.x;
            s[1] += v.y; q[1] += v.y * v.y;
            s[2] += v.z; q[2] += v.z * v.z;
            s[3] += v.w; q[3] += v.w * v.w;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[t * 4 + j] = make_float2(s[j], q[j]);   // [stripe][channel]
    __syncthreads();
    const int cpg = C / G;
    for (int g = t; g < G; g += GN_THREADS) {
        double a = 0.0, b = 0.0;
        for (int sidx = 0; sidx < stripes; ++sidx)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
                const float2 v = red[(sidx * qc + (c >> 2)) * 4 + (c & 3)];
                a += (double)v.x;
                b += (double)v.y;
            }
        partial[((size_t)n * chunks + chunk) * G + g] = make_double2(a, b);
    }
}

// stats[n*G + g] = (mean, rstd)
__global__ void gn_finalize_kernel(const double2* __restrict__ partial, float2* __restrict__ stats, int G,
                                   int chunks, double count, float eps) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (g >= G) return;
    double a = 0.0, b = 0.0;
    for (int c = 0; c < chunks; ++c) {
        const double2 v = partial[((size_t)n * chunks + c) * G + g];
        a += v.x;
        b += v.y;
    }
    const double mean = a / count;
    const double var = fmax(b / count - mean * mean, 0.0);
    stats[(size_t)n * G + g] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
}

// y = x * (rstd*gamma) + (beta - mean*rstd*gamma)  [+ residual]  [ReLU]
__global__ void gn_apply_kernel(const float4* __restrict__ x, float4* __restrict__ y,
                                const float2* __restrict__ stats, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float4* __restrict__ residual, int HW,
                                int C, int G, int relu, long long total) {
    const int qc = C >> 2;
    const int cpg = C / G;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int cq = (int)(i % qc);
        const long long n = i / ((long long)HW * qc);
        float4 v = x[i];
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = cq * 4 + j;
            const float2 st = stats[n * G + c / cpg];
            const float sc = st.y * gamma[c];
            o[j] = o[j] * sc + (beta[c] - st.x * sc);
        }
        if (residual) {
            const float4 r = residual[i];
            o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
        }
        y[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" size_t fgn_group_norm_workspace_bytes(int n_img, int HW, int C, int groups) {
    if (n_img <= 0 || HW <= 0 || groups <= 0) return 0;
    return (size_t)n_img * gn_chunks(HW) * groups * sizeof(double2) + (size_t)n_img * groups * sizeof(float2);
}

extern "C" int fgn_group_norm_nhwc_f32(const float* x, float* y, const float* gamma, const float* beta,
                                       const float* residual, void* ws, size_t ws_bytes, int n_img, int HW,
                                       int C, int groups, float eps, int relu, hipStream_t stream) {
    if (!x || !y || !gamma || !beta || !ws) return FGN_ERR_ARG;
    if (n_img <= 0 || HW <= 0) return FGN_OK;
    if (C % 4 != 0 || C > 4 * GN_THREADS || groups < 1 || C % groups != 0) return FGN_ERR_SHAPE;
    if (ws_bytes < fgn_group_norm_workspace_bytes(n_img, HW, C, groups)) return FGN_ERR_ARG;
    const int chunks = gn_chunks(HW);
    double2* partial = reinterpret_cast<double2*>(ws);
    float2* stats = reinterpret_cast<float2*>(partial + (size_t)n_img * chunks * groups);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(chunks, n_img), dim3(GN_THREADS), 0, stream,
                       reinterpret_cast<const float4*>(x), partial, HW, C, groups, chunks);
    FGN_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(groups, 64), n_img), dim3(64), 0, stream, partial, stats,
                       groups, chunks, (double)HW * (C / groups), eps);
    FGN_LAUNCH_CHECK();
    const long long total = (long long)n_img * HW * (C / 4);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<float4*>(y), stats, gamma, beta,
                       reinterpret_cast<const float4*>(residual), HW, C, groups, relu, total);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ----------------------------------------------------------------------------------
// AvgPool2d(kernel=2, stride=2, ceil_mode=True, count_include_pad=False): the shortcut of a
// strided bottleneck under avg_down (mmdet ResLayer); a partial last window averages the
// pixels that exist.
// ----------------------------------------------------------------------------------
__global__ void avgpool2x2_kernel(const float4* __restrict__ x, float4* __restrict__ y, int H, int W, int C4,
                                  int Ho, int Wo, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const long long b = r / Ho;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        int cnt = 0;
#pragma unroll
        for (int ky = 0; ky < 2; ++ky) {
            const int iy = oy * 2 + ky;
            if (iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 2; ++kx) {
                const int ix = ox * 2 + kx;
                if (ix >= W) continue;
                const float4 v = x[((b * H + iy) * W + ix) * C4 + c];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                ++cnt;
            }
        }
        const float d = (float)cnt;
        y[i] = make_float4(a.x / d, a.y / d, a.z / d, a.w / d);
    }
}

extern "C" int fgn_avgpool2x2_nhwc_f32(const float* x, float* y, int n_img, int H, int W, int C,
                                       hipStream_t stream) {
    if (!x || !y) return FGN_ERR_ARG;
    if (C % 4 != 0 || H < 1 || W < 1) return FGN_ERR_SHAPE;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long total = (long long)n_img * Ho * Wo * (C / 4);
    if (total == 0) return FGN_OK;
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(avgpool2x2_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<float4*>(y), H, W, C / 4, Ho, Wo, total);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
