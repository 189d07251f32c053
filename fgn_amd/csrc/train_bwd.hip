// Backward kernels of the trainable heads (AG-RPN head, shared head, relation / box head, mask head) for
// FGN.forward_train (fgn.py:125-185; the reference obtains these gradients from torch.autograd).  The backbone is
// frozen (frozen_stages=4 + torch.no_grad in extract_feat, fgn_r50_c4_densecl.py:31, fgn.py:67-73), so no gradient
// flows below RoIAlign.  Here:
//   loss gradients (sigmoid CE, smooth L1, softmax CE), BatchNorm(train) backward, the fused relation-head backward
//   (fc -> avg-pool -> ReLU -> GroupNorm -> split 1x1 conv sum), mask-logit backward, im2col for the 3x3 weight
//   gradients, the weight-gradient GEMM dW = dY^T . X on fp32 MFMA, column sums (bias gradients), the Adagrad
//   update (fgn_train_schedule.py:5-13).
// Data gradients run on the forward convolution kernel (conv_igemm.hip) with transposed (1x1) or flipped and
// transposed (3x3) weights (host side, train.py).
#include "common.h"

// ---------------------------------------------------------------------------------------------------------
// loss gradients: d(sum_i w_i loss_i / avg_factor) / d(prediction), times `scale` (upstream gradient, e.g. 1/N)
// ---------------------------------------------------------------------------------------------------------
__global__ void bce_grad_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ w,
                                long long n, float y_thr, float scale, float* __restrict__ dx) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float yv = y[i];
        if (y_thr >= 0.f) yv = yv >= y_thr ? 1.f : 0.f;
        const float s = (float)(1.0 / (1.0 + exp(-(double)x[i])));
        dx[i] = (s - yv) * (w ? w[i] : 1.f) * scale;
    }
}

__global__ void smooth_l1_grad_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                      const float* __restrict__ w, long long n, float beta, float scale,
                                      float* __restrict__ dpred) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = pred[i] - tgt[i];
        const float g = fabsf(d) < beta ? d / beta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        dpred[i] = g * (w ? w[i] : 1.f) * scale;
    }
}

__global__ void softmax_ce_grad_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                       const float* __restrict__ w, int n, int C, float scale, float* __restrict__ dl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = logits + (size_t)i * C;
    float* o = dl + (size_t)i * C;
    const int64_t lab = labels[i];
    if (lab < 0 || lab >= C) {
        for (int c = 0; c < C; ++c) o[c] = 0.f;
        return;
    }
    float m = r[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, r[c]);
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += exp((double)(r[c] - m));
    const float k = (w ? w[i] : 1.f) * scale;
    for (int c = 0; c < C; ++c) {
        const float p = (float)(exp((double)(r[c] - m)) / s);
        o[c] = (p - (c == lab ? 1.f : 0.f)) * k;
    }
}

static inline int grid_for(long long n) {
    const long long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int fgn_bce_logits_grad_f32(const float* x, const float* y, const float* w, long long n, float y_threshold,
                                       float scale, float* dx, hipStream_t stream) {
    if (n > 0 && (!x || !y || !dx)) return FGN_ERR_ARG;
    if (n <= 0) return FGN_OK;
    hipLaunchKernelGGL(bce_grad_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, w, n, y_threshold, scale, dx);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

extern "C" int fgn_smooth_l1_grad_f32(const float* pred, const float* target, const float* w, long long n, float beta,
                                      float scale, float* dpred, hipStream_t stream) {
    if (n > 0 && (!pred || !target || !dpred)) return FGN_ERR_ARG;
    if (!(beta > 0.f)) return FGN_ERR_SHAPE;
    if (n <= 0) return FGN_OK;
    hipLaunchKernelGGL(smooth_l1_grad_kernel, dim3(grid_for(n)), dim3(256), 0, stream, pred, target, w, n, beta, scale,
                       dpred);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

extern "C" int fgn_softmax_ce_grad_f32(const float* logits, const int64_t* labels, const float* w, int n,
                                       int n_classes, float scale, float* dlogits, hipStream_t stream) {
    if (n > 0 && (!logits || !labels || !dlogits)) return FGN_ERR_ARG;
    if (n_classes < 1) return FGN_ERR_SHAPE;
    if (n <= 0) return FGN_OK;
    hipLaunchKernelGGL(softmax_ce_grad_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, logits, labels, w, n,
                       n_classes, scale, dlogits);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ReLU backward: out = dy * [y > 0]  (y = the ReLU's output)
__global__ void relu_backward_kernel(const float4* __restrict__ dy, const float4* __restrict__ y, float4* __restrict__ out,
                                     long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 d = dy[i], v = y[i];
        out[i] = make_float4(v.x > 0.f ? d.x : 0.f, v.y > 0.f ? d.y : 0.f, v.z > 0.f ? d.z : 0.f, v.w > 0.f ? d.w : 0.f);
    }
}

extern "C" int fgn_relu_backward_f32(const float* dy, const float* y, float* out, long long n, hipStream_t stream) {
    if (n > 0 && (!dy || !y || !out)) return FGN_ERR_ARG;
    if (n % 4) return FGN_ERR_SHAPE;
    if (n <= 0) return FGN_OK;
    hipLaunchKernelGGL(relu_backward_kernel, dim3(grid_for(n / 4)), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(y),
                       reinterpret_cast<float4*>(out), n / 4);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Column sums of x [R, C] (bias gradients, reductions over RoIs): fp64 partials over row chunks, fixed order.
// ---------------------------------------------------------------------------------------------------------
constexpr int CS_CHUNKS = 64;

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long long R, int C,
                                                             double* __restrict__ partial) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const long long per = (R + CS_CHUNKS - 1) / CS_CHUNKS;
    const long long r0 = blockIdx.y * per, r1 = r0 + per < R ? r0 + per : R;
    double s = 0.0;
    for (long long r = r0; r < r1; ++r) s += (double)x[r * C + c];
    partial[(size_t)blockIdx.y * C + c] = s;
}

__global__ void colsum_final_kernel(const double* __restrict__ partial, int C, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int k = 0; k < CS_CHUNKS; ++k) s += partial[(size_t)k * C + c];
    out[c] = accumulate ? out[c] + (float)s : (float)s;
}

extern "C" size_t fgn_colsum_scratch_bytes(int C) { return (size_t)CS_CHUNKS * C * sizeof(double); }

extern "C" int fgn_colsum_f32(const float* x, long long R, int C, void* scratch, float* out, int accumulate,
                              hipStream_t stream) {
    if (!out || !scratch || (R > 0 && !x)) return FGN_ERR_ARG;
    if (C <= 0) return FGN_OK;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(C, 256), CS_CHUNKS), dim3(256), 0, stream, x, R < 0 ? 0 : R, C,
                       reinterpret_cast<double*>(scratch));
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream,
                       reinterpret_cast<const double*>(scratch), C, out, accumulate);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm2d (training mode) backward on NHWC rows [P, C]:
//   g = dy * [y_post > 0]                      (the ReLU that follows the norm; y_post optional)
//   dbeta = sum g, dgamma = sum g * xhat,  xhat = (x - mean) * rstd
//   dx = gamma * rstd * (g - dbeta / P - xhat * dgamma / P)
// `g_out` (optional) receives g: the gradient of the residual branch of relu(bn3(conv3) + identity).
// ---------------------------------------------------------------------------------------------------------
constexpr int BNB_CHUNKS = 64;

__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float4* __restrict__ x, const float4* __restrict__ y_post,
                                                             const float4* __restrict__ dy, const float4* __restrict__ mean,
                                                             const float4* __restrict__ var, float eps, int P, int C4,
                                                             double* __restrict__ partial) {
    __shared__ double red[4][64][8];
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + lane;
    const int rows_per = (P + BNB_CHUNKS - 1) / BNB_CHUNKS;
    const int r0 = blockIdx.y * rows_per, r1 = min(P, r0 + rows_per);
    double sb[4] = {0, 0, 0, 0}, sg[4] = {0, 0, 0, 0};
    if (c4 < C4) {
        const float4 m = mean[c4], v = var[c4];
        const float rs[4] = {1.f / sqrtf(v.x + eps), 1.f / sqrtf(v.y + eps), 1.f / sqrtf(v.z + eps), 1.f / sqrtf(v.w + eps)};
        const float mm[4] = {m.x, m.y, m.z, m.w};
        for (int r = r0 + ph; r < r1; r += 4) {
            const size_t o = (size_t)r * C4 + c4;
            const float4 xv = x[o], dv = dy[o];
            float g[4] = {dv.x, dv.y, dv.z, dv.w};
            if (y_post) {
                const float4 yv = y_post[o];
                if (!(yv.x > 0.f)) g[0] = 0.f;
                if (!(yv.y > 0.f)) g[1] = 0.f;
                if (!(yv.z > 0.f)) g[2] = 0.f;
                if (!(yv.w > 0.f)) g[3] = 0.f;
            }
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sb[j] += g[j];
                sg[j] += (double)g[j] * (double)((xs[j] - mm[j]) * rs[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[ph][lane][j] = sb[j]; red[ph][lane][4 + j] = sg[j]; }
    __syncthreads();
    if (ph == 0 && c4 < C4) {
        double* o = partial + ((size_t)blockIdx.y * C4 + c4) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = ((red[0][lane][j] + red[1][lane][j]) + red[2][lane][j]) + red[3][lane][j];
    }
}

__global__ void bn_bwd_final_kernel(const double* __restrict__ partial, int C, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int C4 = C / 4, c4 = c >> 2, j = c & 3;
    double b = 0.0, g = 0.0;
    for (int k = 0; k < BNB_CHUNKS; ++k) {
        const double* o = partial + ((size_t)k * C4 + c4) * 8;
        b += o[j];
        g += o[4 + j];
    }
    dbeta[c] = (float)b;
    dgamma[c] = (float)g;
}

__global__ void bn_bwd_apply_kernel(const float4* __restrict__ x, const float4* __restrict__ y_post,
                                    const float4* __restrict__ dy, const float4* __restrict__ mean,
                                    const float4* __restrict__ var, const float4* __restrict__ gamma,
                                    const float4* __restrict__ dgamma, const float4* __restrict__ dbeta, float eps,
                                    float inv_p, float4* __restrict__ dx, float4* __restrict__ g_out, long long total4,
                                    int C4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const float4 xv = x[i], dv = dy[i], m = mean[c], v = var[c], ga = gamma[c], dg = dgamma[c], db = dbeta[c];
        float g[4] = {dv.x, dv.y, dv.z, dv.w};
        if (y_post) {
            const float4 yv = y_post[i];
            if (!(yv.x > 0.f)) g[0] = 0.f;
            if (!(yv.y > 0.f)) g[1] = 0.f;
            if (!(yv.z > 0.f)) g[2] = 0.f;
            if (!(yv.w > 0.f)) g[3] = 0.f;
        }
        if (g_out) g_out[i] = make_float4(g[0], g[1], g[2], g[3]);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, mm[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w};
        const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, dgg[4] = {dg.x, dg.y, dg.z, dg.w}, dbb[4] = {db.x, db.y, db.z, db.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float rs = 1.f / sqrtf(vv[j] + eps);
            const float xh = (xs[j] - mm[j]) * rs;
            o[j] = gg[j] * rs * (g[j] - dbb[j] * inv_p - xh * dgg[j] * inv_p);
        }
        dx[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" size_t fgn_bn_train_backward_scratch_bytes(int C) { return (size_t)BNB_CHUNKS * C * 2 * sizeof(double); }

extern "C" int fgn_bn_train_backward_f32(const float* x_pre, const float* y_post, const float* dy, const float* mean,
                                         const float* var, const float* gamma, float eps, int P, int C, void* scratch,
                                         float* dx, float* g_out, float* dgamma, float* dbeta, hipStream_t stream) {
    if (!x_pre || !dy || !mean || !var || !gamma || !scratch || !dx || !dgamma || !dbeta) return FGN_ERR_ARG;
    if (C % 4 || P <= 0) return FGN_ERR_SHAPE;
    const int C4 = C / 4;
    double* partial = reinterpret_cast<double*>(scratch);
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(cdiv(C4, 64), BNB_CHUNKS), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(x_pre), reinterpret_cast<const float4*>(y_post),
                       reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(mean),
                       reinterpret_cast<const float4*>(var), eps, P, C4, partial);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, partial, C, dgamma, dbeta);
    const long long total4 = (long long)P * C4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(x_pre), reinterpret_cast<const float4*>(y_post),
                       reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(mean),
                       reinterpret_cast<const float4*>(var), reinterpret_cast<const float4*>(gamma),
                       reinterpret_cast<const float4*>(dgamma), reinterpret_cast<const float4*>(dbeta), eps,
                       1.f / (float)P, reinterpret_cast<float4*>(dx), reinterpret_cast<float4*>(g_out), total4, C4);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Relation / box head backward, the mirror of relation_head_kernel (relation.hip): one workgroup per RoI, a wave
// owns 32 channels x 49 pixels.  Per (RoI r, class n), recomputed from Q and S exactly like the forward:
//   x = Q[r] + S[img, n];  xhat = (x - mean) * rstd;  y = relu(gamma * xhat + beta);  pooled = mean_p y
//   dpooled[c] = sum_j d6[r, n, j] * fcw[j, c]                    (fc_cls | fc_reg backward)
//   g = [y > 0] * dpooled / 49 ;  gg = g * gamma
//   dx = rstd * (gg - mean_grp(gg) - xhat * mean_grp(gg * xhat))  (GroupNorm backward)
// Outputs: dQ[r] = sum_n dx; dZ[r, n] = dx (reduced over the RoIs of an image into dS by a column sum);
// pooled[r, n, c] (for the fc weight gradient, a plain GEMM); per-RoI partials of dgamma / dbeta.
// ---------------------------------------------------------------------------------------------------------
constexpr int RELB_MAX_N = 8;
constexpr int RELB_WAVES = 8;

__global__ __launch_bounds__(64 * RELB_WAVES) void relation_head_backward_kernel(
    const float* __restrict__ Q, const float* __restrict__ S, const float* __restrict__ rois,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ fcw,
    const float* __restrict__ d6, float* __restrict__ dQ, float* __restrict__ dZ, float* __restrict__ pooled,
    float* __restrict__ dgamma_part, float* __restrict__ dbeta_part, int n_rois, int n_ways, int C, int gw, float eps) {
    constexpr int P = 49;
    auto group_sum = [gw](float v) {
#pragma unroll
        for (int off = 32; off >= 8; off >>= 1) v += __shfl_xor(v, off, 64);
        for (int off = (gw >> 3); off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        return v;
    };
    auto slot_sum = [](float v) {                        // over the 8 pixel slots only (same channel quad)
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
        return v;
    };
    const int r = blockIdx.x;
    if (r >= n_rois) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int quad = lane & 7, slot = lane >> 3;
    const int img = (int)rois[(size_t)r * 5];
    const float inv_cnt = 1.f / ((float)gw * (float)P);
    const int groups = C / 32;
    for (int g = wv; g < groups; g += RELB_WAVES) {
        const int c = g * 32 + quad * 4;
        float4 q[7], dq[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int p = slot + 8 * i;
            q[i] = (p < P) ? *reinterpret_cast<const float4*>(Q + ((size_t)r * P + p) * C + c)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            dq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
        const float4 be = *reinterpret_cast<const float4*>(beta + c);
        float4 dga = make_float4(0.f, 0.f, 0.f, 0.f), dbe = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int n = 0; n < n_ways; ++n) {
            const size_t row = (size_t)r * n_ways + n;
            const float* Sn = S + ((size_t)(img * n_ways + n) * P) * C + c;
            float4 x[7];
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int p = slot + 8 * i;
                if (p < P) {
                    const float4 s = *reinterpret_cast<const float4*>(Sn + (size_t)p * C);
                    x[i] = make_float4(q[i].x + s.x, q[i].y + s.y, q[i].z + s.z, q[i].w + s.w);
                    sum += (x[i].x + x[i].y) + (x[i].z + x[i].w);
                } else {
                    x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            const float mean = group_sum(sum) * inv_cnt;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int p = slot + 8 * i;
                if (p < P) {
                    const float a = x[i].x - mean, b = x[i].y - mean, d = x[i].z - mean, e = x[i].w - mean;
                    sq += (a * a + b * b) + (d * d + e * e);
                }
            }
            const float var = group_sum(sq) * inv_cnt;
            const float rstd = 1.f / sqrtf(var + eps);
            // upstream: dpooled[c .. c+3] / 49
            float4 dp = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float dj = d6[row * 6 + j];
                const float4 fwj = *reinterpret_cast<const float4*>(fcw + (size_t)j * C + c);
                dp.x += dj * fwj.x; dp.y += dj * fwj.y; dp.z += dj * fwj.z; dp.w += dj * fwj.w;
            }
            const float ip = 1.f / (float)P;
            dp.x *= ip; dp.y *= ip; dp.z *= ip; dp.w *= ip;
            // pass 1: xhat (kept in x), masked upstream g (kept in gy), group sums of gg and gg * xhat, pooled
            float4 gy[7];
            float s1 = 0.f, s2 = 0.f;
            float4 pool = make_float4(0.f, 0.f, 0.f, 0.f), sga = make_float4(0.f, 0.f, 0.f, 0.f),
                   sbe = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int p = slot + 8 * i;
                gy[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p < P) {
                    const float4 xh = make_float4((x[i].x - mean) * rstd, (x[i].y - mean) * rstd, (x[i].z - mean) * rstd,
                                                  (x[i].w - mean) * rstd);
                    const float4 y = make_float4(fmaxf(xh.x * ga.x + be.x, 0.f), fmaxf(xh.y * ga.y + be.y, 0.f),
                                                 fmaxf(xh.z * ga.z + be.z, 0.f), fmaxf(xh.w * ga.w + be.w, 0.f));
                    pool.x += y.x; pool.y += y.y; pool.z += y.z; pool.w += y.w;
                    const float4 gv = make_float4(y.x > 0.f ? dp.x : 0.f, y.y > 0.f ? dp.y : 0.f, y.z > 0.f ? dp.z : 0.f,
                                                  y.w > 0.f ? dp.w : 0.f);
                    sbe.x += gv.x; sbe.y += gv.y; sbe.z += gv.z; sbe.w += gv.w;
                    sga.x += gv.x * xh.x; sga.y += gv.y * xh.y; sga.z += gv.z * xh.z; sga.w += gv.w * xh.w;
                    const float4 gg = make_float4(gv.x * ga.x, gv.y * ga.y, gv.z * ga.z, gv.w * ga.w);
                    s1 += (gg.x + gg.y) + (gg.z + gg.w);
                    s2 += (gg.x * xh.x + gg.y * xh.y) + (gg.z * xh.z + gg.w * xh.w);
                    gy[i] = gg;
                    x[i] = xh;
                }
            }
            const float m1 = group_sum(s1) * inv_cnt, m2 = group_sum(s2) * inv_cnt;
            pool.x = slot_sum(pool.x) * ip; pool.y = slot_sum(pool.y) * ip; pool.z = slot_sum(pool.z) * ip;
            pool.w = slot_sum(pool.w) * ip;
            dga.x += slot_sum(sga.x); dga.y += slot_sum(sga.y); dga.z += slot_sum(sga.z); dga.w += slot_sum(sga.w);
            dbe.x += slot_sum(sbe.x); dbe.y += slot_sum(sbe.y); dbe.z += slot_sum(sbe.z); dbe.w += slot_sum(sbe.w);
            if (slot == 0) *reinterpret_cast<float4*>(pooled + row * C + c) = pool;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int p = slot + 8 * i;
                if (p < P) {
                    const float4 dx = make_float4(rstd * (gy[i].x - m1 - x[i].x * m2), rstd * (gy[i].y - m1 - x[i].y * m2),
                                                  rstd * (gy[i].z - m1 - x[i].z * m2), rstd * (gy[i].w - m1 - x[i].w * m2));
                    *reinterpret_cast<float4*>(dZ + (row * P + p) * C + c) = dx;
                    dq[i].x += dx.x; dq[i].y += dx.y; dq[i].z += dx.z; dq[i].w += dx.w;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int p = slot + 8 * i;
            if (p < P) *reinterpret_cast<float4*>(dQ + ((size_t)r * P + p) * C + c) = dq[i];
        }
        if (slot == 0) {
            *reinterpret_cast<float4*>(dgamma_part + (size_t)r * C + c) = dga;
            *reinterpret_cast<float4*>(dbeta_part + (size_t)r * C + c) = dbe;
        }
    }
}

extern "C" int fgn_relation_gn_head_backward_f32(const float* Q, const float* S, const float* rois, const float* gn_weight,
                                                 const float* gn_bias, const float* fc_weight, const float* d_out6,
                                                 float* dQ, float* dZ, float* pooled, float* dgamma_part,
                                                 float* dbeta_part, int n_rois, int n_ways, int C, int gn_groups,
                                                 int roi_size, float eps, hipStream_t stream) {
    if (!Q || !S || !rois || !gn_weight || !gn_bias || !fc_weight || !d_out6 || !dQ || !dZ || !pooled || !dgamma_part ||
        !dbeta_part)
        return FGN_ERR_ARG;
    if (roi_size != 7 || gn_groups <= 0 || C % 32 != 0 || C % gn_groups != 0 || n_ways < 1 || n_ways > RELB_MAX_N)
        return FGN_ERR_SHAPE;
    const int gw = C / gn_groups;
    if (gw != 8 && gw != 16 && gw != 32) return FGN_ERR_SHAPE;
    if (n_rois == 0) return FGN_OK;
    hipLaunchKernelGGL(relation_head_backward_kernel, dim3(n_rois), dim3(64 * RELB_WAVES), 0, stream, Q, S, rois,
                       gn_weight, gn_bias, fc_weight, d_out6, dQ, dZ, pooled, dgamma_part, dbeta_part, n_rois, n_ways, C,
                       gw, eps);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Mask logits backward (FCNMaskHead: ReLU(deconv) -> conv_logits 1x1, one class).  up [D, P*P, 4, C] is the
// un-shuffled deconv output (sub-position major, as fgn_mask_logits_f32 reads it); dlogit [D, 2P, 2P].
//   d_up[d, px, sub, c] = dlogit[d, 2*i + dy, 2*j + dx] * w[c] * [up > 0]
//   dw_part[d, c] = sum_{px, sub} dlogit * up          (column-summed over d afterwards); dbias = sum dlogit
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_logits_backward_kernel(const float* __restrict__ up, const float* __restrict__ dlogit,
                                                                   const float* __restrict__ w, float* __restrict__ d_up,
                                                                   float* __restrict__ dw_part, int P, int C) {
    const int d = blockIdx.x;
    const int M = 2 * P;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float wc = w[c];
        float acc = 0.f;
        for (int px = 0; px < P * P; ++px) {
            const int i = px / P, j = px - i * P;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const float dl = dlogit[((size_t)d * M + 2 * i + (sub >> 1)) * M + 2 * j + (sub & 1)];
                const size_t o = (((size_t)d * P * P + px) * 4 + sub) * C + c;
                const float u = up[o];
                acc += dl * u;
                d_up[o] = u > 0.f ? dl * wc : 0.f;
            }
        }
        dw_part[(size_t)d * C + c] = acc;
    }
}

extern "C" int fgn_mask_logits_backward_f32(const float* up, const float* dlogit, const float* w, float* d_up,
                                            float* dw_part, int n_det, int roi_size, int C, hipStream_t stream) {
    if (n_det > 0 && (!up || !dlogit || !w || !d_up || !dw_part)) return FGN_ERR_ARG;
    if (n_det <= 0) return FGN_OK;
    hipLaunchKernelGGL(mask_logits_backward_kernel, dim3(n_det), dim3(256), 0, stream, up, dlogit, w, d_up, dw_part,
                       roi_size, C);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// im2col of a 3x3 / stride 1 / pad 1 convolution input, NHWC: out [n*h*w, 9*C], column = (ky*3 + kx)*C + ci.
// The weight gradient is then the plain GEMM dW[co, (ky,kx,ci)] = dY^T . out.
// ---------------------------------------------------------------------------------------------------------
__global__ void im2col3x3_kernel(const float4* __restrict__ x, float4* __restrict__ out, int n, int H, int W, int C4) {
    const long long total = (long long)n * H * W * 9 * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long r = i / C4;
        const int tap = (int)(r % 9);
        r /= 9;
        const int xw = (int)(r % W);
        r /= W;
        const int yh = (int)(r % H);
        const int b = (int)(r / H);
        const int sy = yh + tap / 3 - 1, sx = xw + tap % 3 - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = x[(((size_t)b * H + sy) * W + sx) * C4 + c];
        out[i] = v;
    }
}

extern "C" int fgn_im2col3x3_f32(const float* x, float* out, int n, int H, int W, int C, hipStream_t stream) {
    if (n > 0 && (!x || !out)) return FGN_ERR_ARG;
    if (C % 4) return FGN_ERR_SHAPE;
    if (n <= 0) return FGN_OK;
    const long long total = (long long)n * H * W * 9 * (C / 4);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for(total)), dim3(256), 0, stream, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<float4*>(out), n, H, W, C / 4);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM  C[M,N] = A[R,M]^T . B[R,N]  (A = dY rows x Cout, B = X or im2col(X) rows x K): both operands
// are row-major with the REDUCTION index R as the slow one, which is exactly the lane layout of
// v_mfma_f32_16x16x4_f32 (lane = (k = lane/16, m or n = lane%16)): a K-chunk of 32 rows x 64 columns of each operand is
// staged in LDS as it lies in memory (coalesced float4 loads, rows padded by 16 floats so that the two 16-lane groups
// of a half-wave hit disjoint banks) and every MFMA operand is one conflict-free ds_read_b32.  64x64 output tile per
// workgroup, 4 waves x (2x2 tiles of 16x16) - or 128x128 with 4x4 tiles per wave when those alone fill the chip -,
// global loads of chunk k+1 in flight while chunk k is multiplied.
// Few output tiles (Cout x Cin of a 1x1 conv) -> the rows are split into slabs; partial tiles are reduced in slab
// order by gemm_tn_reduce_kernel (bit-reproducible, no atomics).
// ---------------------------------------------------------------------------------------------------------
template <int BM, int BN, int TN_BK>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ C, int R, int M, int N, int n_tiles_n,
                                                      int rows_per_split) {
    constexpr int LDA = BM + 16, LDB = BN + 16;          // +16 floats: rows k and k+1 start 16 banks apart
    constexpr int TM = BM / 32, TN = BN / 32;            // 16x16 MFMA tiles per wave (wave tile BM/2 x BN/2)
    constexpr int A4 = BM / 4, B4 = BN / 4;              // float4 per chunk row
    constexpr int A_LD = TN_BK * A4 / 256, B_LD = TN_BK * B4 / 256;      // float4 loads per thread and chunk
    extern __shared__ __attribute__((aligned(16))) float tn_smem[];
    float* As = tn_smem;                                 // [2][TN_BK][LDA]
    float* Bs = tn_smem + 2 * TN_BK * LDA;               // [2][TN_BK][LDB]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int tile_m = blockIdx.x / n_tiles_n, tile_n = blockIdx.x - tile_m * n_tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int r0 = blockIdx.y * rows_per_split, r1 = min(R, r0 + rows_per_split);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 ra[A_LD], rb[B_LD];
    auto load = [&](int k0) {
#pragma unroll
        for (int h = 0; h < A_LD; ++h) {
            const int e = t + 256 * h, row = e / A4, c = (e - row * A4) * 4, r = k0 + row;
            ra[h] = (m0 + c < M && r < r1) ? *reinterpret_cast<const float4*>(A + (size_t)r * M + m0 + c)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int h = 0; h < B_LD; ++h) {
            const int e = t + 256 * h, row = e / B4, c = (e - row * B4) * 4, r = k0 + row;
            rb[h] = (n0 + c < N && r < r1) ? *reinterpret_cast<const float4*>(B + (size_t)r * N + n0 + c)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store = [&](int st) {
#pragma unroll
        for (int h = 0; h < A_LD; ++h) {
            const int e = t + 256 * h, row = e / A4, c = (e - row * A4) * 4;
            *reinterpret_cast<float4*>(As + (st * TN_BK + row) * LDA + c) = ra[h];
        }
#pragma unroll
        for (int h = 0; h < B_LD; ++h) {
            const int e = t + 256 * h, row = e / B4, c = (e - row * B4) * 4;
            *reinterpret_cast<float4*>(Bs + (st * TN_BK + row) * LDB + c) = rb[h];
        }
    };
    const int r16 = lane & 15, g16 = lane >> 4;
    load(r0);
    store(0);
    __syncthreads();
    int cur = 0;
    for (int k0 = r0; k0 < r1; k0 += TN_BK) {
        const bool more = k0 + TN_BK < r1;
        if (more) load(k0 + TN_BK);
        const float* Ac = As + cur * TN_BK * LDA + wm * (BM / 2) + r16;
        const float* Bc = Bs + cur * TN_BK * LDB + wn * (BN / 2) + r16;
#pragma unroll
        for (int ks = 0; ks < TN_BK / 4; ++ks) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Ac[(ks * 4 + g16) * LDA + 16 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bc[(ks * 4 + g16) * LDB + 16 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    float* Cs = C + (size_t)blockIdx.y * M * N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + 16 * j + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * (BM / 2) + 16 * i + 4 * g16 + r;
                if (m < M && n < N) Cs[(size_t)m * N + n] = acc[i][j][r];
            }
        }
}

__global__ void gemm_tn_reduce_kernel(const float4* __restrict__ part, float4* __restrict__ out, long long n4, int S) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 s = part[i];
        for (int k = 1; k < S; ++k) {
            const float4 v = part[(size_t)k * n4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        out[i] = s;
    }
}

// 128x128 tiles (half the L2 traffic per MFMA, 2 workgroups per CU) for the large products, 64x64 tiles otherwise;
// few output tiles -> row slabs
#include <cstdlib>
static inline bool tn_big(int R, int M, int N) {
    return M >= 1024 && N >= 1024 && R >= 1024;      // measured: 96 vs 87 TFLOP/s at 1024 x 1024 x 6272 rows, a loss below
}

static inline int tn_splits(int R, int M, int N) {
    const bool big = tn_big(R, M, N);
    const int tiles = big ? cdiv(M, 128) * cdiv(N, 128) : cdiv(M, 64) * cdiv(N, 64);
    int S = cdiv(big ? 512 : 1024, tiles);          // about two (128x128) / four (64x64) workgroups per CU in flight
    const int max_s = R / 256 > 1 ? R / 256 : 1;    // at least 256 rows per slab
    if (S > max_s) S = max_s;
    if (S > 64) S = 64;
    return S < 1 ? 1 : S;
}

extern "C" size_t fgn_gemm_tn_workspace_bytes(int R, int M, int N) {
    const int S = tn_splits(R, M, N);
    return S > 1 ? (size_t)S * M * N * sizeof(float) : 0;
}

template <int BM, int BN, int TN_BK>
static int tn_launch(const float* A, const float* B, float* C, int R, int M, int N, void* workspace, hipStream_t stream) {
    const int S = tn_splits(R, M, N);
    if (S > 1 && !workspace) return FGN_ERR_ARG;
    const int ntn = cdiv(N, BN);
    int rows_per = cdiv(R, S);
    rows_per = cdiv(rows_per, TN_BK) * TN_BK;
    const int S_eff = cdiv(R, rows_per);
    float* dst = S_eff > 1 ? reinterpret_cast<float*>(workspace) : C;
    constexpr size_t lds = (size_t)2 * TN_BK * ((BM + 16) + (BN + 16)) * sizeof(float);      // 40 KB / 73.7 KB
    static unsigned long long ok = 0ull;
    const hipError_t attr = fgn_allow_full_lds(reinterpret_cast<const void*>(gemm_tn_kernel<BM, BN, TN_BK>), &ok);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, TN_BK>), dim3(cdiv(M, BM) * ntn, S_eff), dim3(256), lds, stream, A, B, dst, R, M, N,
                       ntn, rows_per);
    if (S_eff > 1) {
        const long long n4 = (long long)M * N / 4;
        hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(grid_for(n4)), dim3(256), 0, stream,
                           reinterpret_cast<const float4*>(dst), reinterpret_cast<float4*>(C), n4, S_eff);
    }
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// C [M,N] = A [R,M]^T B [R,N]; M % 4 == 0, N % 4 == 0; workspace: fgn_gemm_tn_workspace_bytes(R, M, N)
extern "C" int fgn_gemm_tn_f32(const float* A, const float* B, float* C, int R, int M, int N, void* workspace,
                               hipStream_t stream) {
    if (!A || !B || !C) return FGN_ERR_ARG;
    if (R <= 0 || M <= 0 || N <= 0 || M % 4 || N % 4) return FGN_ERR_SHAPE;
    // (64-row chunks at half the occupancy were 8 % slower on every shape of tools/gemm_tn_bench.py)
    return tn_big(R, M, N) ? tn_launch<128, 128, 32>(A, B, C, R, M, N, workspace, stream)
                           : tn_launch<64, 64, 32>(A, B, C, R, M, N, workspace, stream);
}

// ---------------------------------------------------------------------------------------------------------
// torch.optim.Adagrad step (fgn_train_schedule.py:5-13; lr_decay 0, eps 1e-10):
//   g += weight_decay * p;  state += g * g;  p -= lr * g / (sqrt(state) + eps)
// ---------------------------------------------------------------------------------------------------------
__global__ void adagrad_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ state, long long n,
                               float lr, float wd, float eps) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float pv = p[i];
        const float gv = g[i] + wd * pv;
        const float st = state[i] + gv * gv;
        state[i] = st;
        p[i] = pv - lr * gv / (sqrtf(st) + eps);
    }
}

// ------------------------------------------------------------------------------------------------
// Small products the MFMA kernels do not take (an operand dimension that is not a multiple of 4 / 32: the 6-row fc
// weight gradient, the data gradient through the 6-row fc and through the 75-channel AG-RPN head).  One thread per
// output element, the reduction index walked in order (bit-reproducible), fp32 fma.  A few MFLOP each.
//   trans_a = 0 :  C[M,N] = A[M,K]   * B[K,N]
//   trans_a = 1 :  C[M,N] = A[K,M]^T * B[K,N]      (lda / ldb / ldc = row strides in floats)
// ------------------------------------------------------------------------------------------------
__global__ void gemm_small_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
                                  int N, int K, int lda, int ldb, int ldc, int trans_a) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)M * N) return;
    const int m = (int)(i / N), n = (int)(i - (long long)m * N);
    float acc = 0.f;
    if (trans_a)
        for (int k = 0; k < K; ++k) acc = fmaf(A[(size_t)k * lda + m], B[(size_t)k * ldb + n], acc);
    else
        for (int k = 0; k < K; ++k) acc = fmaf(A[(size_t)m * lda + k], B[(size_t)k * ldb + n], acc);
    C[(size_t)m * ldc + n] = acc;
}

extern "C" int fgn_gemm_small_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                  int trans_a, hipStream_t stream) {
    if (!A || !B || !C) return FGN_ERR_ARG;
    if (M < 0 || N < 0 || K < 0 || ldb < N || ldc < N || lda < (trans_a ? M : K)) return FGN_ERR_SHAPE;
    if (M == 0 || N == 0) return FGN_OK;
    const long long total = (long long)M * N;
    hipLaunchKernelGGL(gemm_small_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, A, B, C, M, N, K,
                       lda, ldb, ldc, trans_a);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// The same update for up to FGN_ADAGRAD_MAX_TENSORS parameter tensors in ONE launch (the heads have ~40 tensors from 1 to
// 9.4 M elements; one launch each was launch-rate-bound: 0.4 ms of a 14 ms step).  The table travels as a kernel argument;
// block b works on chunk b of the concatenation of all tensors cut into 4096-element chunks (first_chunk = prefix sums).
// Element-wise, so the result is that of the one-tensor launches bit for bit.
constexpr int FGN_ADAGRAD_MAX_TENSORS = 64;
struct AdagradTable {
    float* p[FGN_ADAGRAD_MAX_TENSORS];
    const float* g[FGN_ADAGRAD_MAX_TENSORS];
    float* s[FGN_ADAGRAD_MAX_TENSORS];
    long long n[FGN_ADAGRAD_MAX_TENSORS];
    int first_chunk[FGN_ADAGRAD_MAX_TENSORS + 1];
    float lr[FGN_ADAGRAD_MAX_TENSORS];
    int count;
};
__global__ __launch_bounds__(256) void adagrad_multi_kernel(const AdagradTable t, float wd, float eps) {
    const int chunk = blockIdx.x;
    int k = 0;
    while (k + 1 < t.count && t.first_chunk[k + 1] <= chunk) ++k;          // block-uniform, <= 64 steps over SGPRs
    const long long base = (long long)(chunk - t.first_chunk[k]) * 4096;
    float* __restrict__ p = t.p[k];
    const float* __restrict__ g = t.g[k];
    float* __restrict__ state = t.s[k];
    const long long end = min(t.n[k], base + 4096);
    const float lr = t.lr[k];
    for (long long i = base + threadIdx.x; i < end; i += 256) {
        const float pv = p[i];
        const float gv = g[i] + wd * pv;
        const float st = state[i] + gv * gv;
        state[i] = st;
        p[i] = pv - lr * gv / (sqrtf(st) + eps);
    }
}

extern "C" int fgn_adagrad_multi_f32(float* const* params, const float* const* grads, float* const* state_sums,
                                     const long long* n, const float* lr, int count, float weight_decay, float eps,
                                     hipStream_t stream) {
    if (count < 0 || (count > 0 && (!params || !grads || !state_sums || !n || !lr))) return FGN_ERR_ARG;
    // `i` walks the caller's list ONCE across the tables: empty tensors are skipped without taking a slot, so a table may
    // cover more than FGN_ADAGRAD_MAX_TENSORS list entries - the next table starts where this one stopped (a fixed
    // stride would hand the entries beyond it to two tables: two updates in one step)
    int i = 0;
    while (i < count) {
        AdagradTable t;
        t.count = 0;
        int chunks = 0;
        for (; i < count && t.count < FGN_ADAGRAD_MAX_TENSORS; ++i) {
            if (n[i] <= 0) continue;
            if (!params[i] || !grads[i] || !state_sums[i]) return FGN_ERR_ARG;
            const int k = t.count++;
            t.p[k] = params[i]; t.g[k] = grads[i]; t.s[k] = state_sums[i]; t.n[k] = n[i]; t.lr[k] = lr[i];
            t.first_chunk[k] = chunks;
            chunks += (int)((n[i] + 4095) / 4096);
        }
        t.first_chunk[t.count] = chunks;
        if (t.count == 0) continue;
        hipLaunchKernelGGL(adagrad_multi_kernel, dim3(chunks), dim3(256), 0, stream, t, weight_decay, eps);
        FGN_LAUNCH_CHECK();
    }
    return FGN_OK;
}

extern "C" int fgn_adagrad_step_f32(float* param, const float* grad, float* state_sum, long long n, float lr,
                                    float weight_decay, float eps, hipStream_t stream) {
    if (n > 0 && (!param || !grad || !state_sum)) return FGN_ERR_ARG;
    if (n <= 0) return FGN_OK;
    hipLaunchKernelGGL(adagrad_kernel, dim3(grid_for(n)), dim3(256), 0, stream, param, grad, state_sum, n, lr,
                       weight_decay, eps);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
