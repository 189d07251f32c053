// Device kernels of FGN.forward_train (fgn.py:125-185): the pieces the inference path does not have.
//   box_assign      MaxIoUAssigner (the reference vendors mmdet's file: my_max_iou_assigner.py:126-213) on top of
//                   mmdet's bbox_overlaps, for the AG-RPN anchors and for the RoI head's proposals
//   bbox2delta      DeltaXYWHBBoxCoder.encode of the sampled positives
//   loss sums       sigmoid CE (AG-RPN objectness, mask BCE), smooth L1 (both box losses), softmax CE (box head),
//                   each with per-element weights and an avg_factor (mmdet weight_reduce_loss)
//   bn_train        BatchNorm in TRAINING mode for the shared head (fgn_roi_head.py:202-238: norm_cfg BN with
//                   requires_grad=True inside a module that is in train()): batch statistics + running update
// Compiled with -ffp-contract=off: the IoU is the oracle's fp32 expression bit for bit, so the threshold tests
// (>= pos_iou_thr, < neg_iou_thr, == the per-GT maximum) select the same boxes.
#include "common.h"

// mmdet bbox_overlaps(mode='iou', eps=1e-6): overlap / max(area1 + area2 - overlap, eps), fp32, left to right
__device__ __forceinline__ float iou_exact(const float4 g, float garea, const float4 b, float barea) {
    const float lx = fmaxf(g.x, b.x), ly = fmaxf(g.y, b.y);
    const float rx = fminf(g.z, b.z), ry = fminf(g.w, b.w);
    const float w = fmaxf(rx - lx, 0.f), h = fmaxf(ry - ly, 0.f);
    const float overlap = w * h;
    float uni = (garea + barea) - overlap;
    uni = fmaxf(uni, 1e-6f);
    return overlap / uni;
}

constexpr int ASSIGN_MAX_GT = 256;

// pass 1: per box the maximum IoU over the GTs and its (first) arg-max; per GT the maximum over the boxes
// (atomicMax on the bit pattern: IoUs are >= 0, where the IEEE order is the integer order)
__global__ __launch_bounds__(256) void assign_max_kernel(const float* __restrict__ boxes, int box_stride,
                                                         const uint8_t* __restrict__ inside, const float4* __restrict__ gts,
                                                         int n, int k, float* __restrict__ max_ov, int32_t* __restrict__ argmax,
                                                         uint32_t* __restrict__ gt_max_bits) {
    __shared__ float4 sg[ASSIGN_MAX_GT];
    __shared__ float sa[ASSIGN_MAX_GT];
    __shared__ uint32_t smax[ASSIGN_MAX_GT];
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const float4 g = gts[i];
        sg[i] = g;
        sa[i] = (g.z - g.x) * (g.w - g.y);
        smax[i] = 0u;
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (!inside || inside[i])) {
        const float* bp = boxes + (size_t)i * box_stride;
        const float4 b = make_float4(bp[0], bp[1], bp[2], bp[3]);
        const float barea = (b.z - b.x) * (b.w - b.y);
        float best = -1.f;
        int bi = 0;
        for (int j = 0; j < k; ++j) {
            const float v = iou_exact(sg[j], sa[j], b, barea);
            if (v > best) { best = v; bi = j; }
            atomicMax(&smax[j], __float_as_uint(v));
        }
        max_ov[i] = best;
        argmax[i] = bi;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += blockDim.x)
        if (smax[j]) atomicMax(&gt_max_bits[j], smax[j]);
}

// pass 2: the assignment steps 1-4 of MaxIoUAssigner.assign_wrt_overlaps.  gt_inds: -2 = not a candidate (outside
// anchor), -1 = ignored, 0 = negative, i + 1 = positive for GT i.  Low-quality matching walks the GTs in order, so a
// later GT overwrites an earlier one exactly like the reference's loop.
__global__ __launch_bounds__(256) void assign_final_kernel(const float* __restrict__ boxes, int box_stride,
                                                           const uint8_t* __restrict__ inside, const float4* __restrict__ gts,
                                                           int n, int k, float pos_thr, float neg_thr, float min_pos,
                                                           int low_quality, const float* __restrict__ max_ov,
                                                           const int32_t* __restrict__ argmax,
                                                           const uint32_t* __restrict__ gt_max_bits,
                                                           int32_t* __restrict__ gt_inds) {
    __shared__ float4 sg[ASSIGN_MAX_GT];
    __shared__ float sa[ASSIGN_MAX_GT];
    __shared__ float smax[ASSIGN_MAX_GT];
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const float4 g = gts[i];
        sg[i] = g;
        sa[i] = (g.z - g.x) * (g.w - g.y);
        smax[i] = __uint_as_float(gt_max_bits[i]);
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (inside && !inside[i]) { gt_inds[i] = -2; return; }
    if (k == 0) { gt_inds[i] = 0; return; }
    const float mo = max_ov[i];
    int g = -1;
    if (mo >= 0.f && mo < neg_thr) g = 0;
    if (mo >= pos_thr) g = argmax[i] + 1;
    if (low_quality) {
        const float* bp = boxes + (size_t)i * box_stride;
        const float4 b = make_float4(bp[0], bp[1], bp[2], bp[3]);
        const float barea = (b.z - b.x) * (b.w - b.y);
        for (int j = 0; j < k; ++j)
            if (smax[j] >= min_pos && iou_exact(sg[j], sa[j], b, barea) == smax[j]) g = j + 1;
    }
    gt_inds[i] = g;
}

static inline size_t assign_gt_bytes(int k) { return ((size_t)(k > 0 ? k : 1) * 4 + 63) / 64 * 64; }

extern "C" size_t fgn_box_assign_scratch_bytes(int n, int k) { return assign_gt_bytes(k) + (size_t)(n > 0 ? n : 0) * 8; }

// gt_inds [n] int32; max_overlaps [n] optional.  scratch: fgn_box_assign_scratch_bytes(n, k), uninitialised.
extern "C" int fgn_box_assign_f32(const float* boxes, int box_stride, const uint8_t* inside, const float* gts, int n,
                                  int k, float pos_iou_thr, float neg_iou_thr, float min_pos_iou,
                                  int match_low_quality, void* scratch, int32_t* gt_inds, float* max_overlaps,
                                  hipStream_t stream) {
    if (!boxes || !gt_inds || !scratch || (k > 0 && !gts)) return FGN_ERR_ARG;
    if (k > ASSIGN_MAX_GT || box_stride < 4) return FGN_ERR_SHAPE;
    if (n <= 0) return FGN_OK;
    unsigned char* sb = reinterpret_cast<unsigned char*>(scratch);
    uint32_t* gt_max = reinterpret_cast<uint32_t*>(sb);
    float* mo = max_overlaps ? max_overlaps : reinterpret_cast<float*>(sb + assign_gt_bytes(k));
    int32_t* am = reinterpret_cast<int32_t*>(sb + assign_gt_bytes(k) + (size_t)n * 4);
    const hipError_t me = hipMemsetAsync(gt_max, 0, assign_gt_bytes(k), stream);
    if (me != hipSuccess) return (int)me;
    const dim3 grid(cdiv(n, 256));
    if (k > 0) {
        hipLaunchKernelGGL(assign_max_kernel, grid, dim3(256), 0, stream, boxes, box_stride, inside,
                           reinterpret_cast<const float4*>(gts), n, k, mo, am, gt_max);
    }
    hipLaunchKernelGGL(assign_final_kernel, grid, dim3(256), 0, stream, boxes, box_stride, inside,
                       reinterpret_cast<const float4*>(gts), n, k, pos_iou_thr, neg_iou_thr, min_pos_iou,
                       match_low_quality, mo, am, gt_max, gt_inds);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// DeltaXYWHBBoxCoder.encode (mmdet 2.18 bbox2delta): the logarithm is evaluated in fp64 and rounded once
// ---------------------------------------------------------------------------------------------------------
__global__ void bbox2delta_kernel(const float4* __restrict__ props, const float4* __restrict__ gts, float4* __restrict__ out,
                                  int n, float4 mean, float4 stdv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = props[i], g = gts[i];
    const float px = (p.x + p.z) * 0.5f, py = (p.y + p.w) * 0.5f;
    const float pw = p.z - p.x, ph = p.w - p.y;
    const float gx = (g.x + g.z) * 0.5f, gy = (g.y + g.w) * 0.5f;
    const float gw = g.z - g.x, gh = g.w - g.y;
    const float dx = (gx - px) / pw, dy = (gy - py) / ph;
    const float dw = (float)log((double)(gw / pw)), dh = (float)log((double)(gh / ph));
    out[i] = make_float4((dx - mean.x) / stdv.x, (dy - mean.y) / stdv.y, (dw - mean.z) / stdv.z, (dh - mean.w) / stdv.w);
}

extern "C" int fgn_bbox2delta_f32(const float* proposals, const float* gts, float* out, int n, const float* means4,
                                  const float* stds4, hipStream_t stream) {
    if (!means4 || !stds4 || (n > 0 && (!proposals || !gts || !out))) return FGN_ERR_ARG;
    if (n <= 0) return FGN_OK;
    hipLaunchKernelGGL(bbox2delta_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(proposals), reinterpret_cast<const float4*>(gts),
                       reinterpret_cast<float4*>(out), n, make_float4(means4[0], means4[1], means4[2], means4[3]),
                       make_float4(stds4[0], stds4[1], stds4[2], stds4[3]));
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Weighted loss sums: one workgroup, fixed reduction order (bit-reproducible), fp64 accumulation.
// out[0] = sum_i w_i * loss_i / avg_factor.
// ---------------------------------------------------------------------------------------------------------
constexpr int LOSS_THREADS = 1024;

__device__ __forceinline__ void loss_block_finish(double acc, double avg_factor, float* out) {
    __shared__ double part[LOSS_THREADS / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < LOSS_THREADS / 64; ++w) s += part[w];
        out[0] = (float)(s / avg_factor);
    }
}

// F.binary_cross_entropy_with_logits(x, y): max(x, 0) - x*y + log1p(exp(-|x|)); y may be a probability target that
// is binarised at `y_thr` first (mask targets: (roi_align(gt) >= 0.5), mmdet mask_target_single), y_thr < 0: as is
__global__ __launch_bounds__(LOSS_THREADS) void bce_sum_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               const float* __restrict__ w, long long n, float y_thr,
                                                               double avg_factor, float* __restrict__ out) {
    double acc = 0.0;
    for (long long i = threadIdx.x; i < n; i += LOSS_THREADS) {
        const double xv = (double)x[i];
        double yv = (double)y[i];
        if (y_thr >= 0.f) yv = y[i] >= y_thr ? 1.0 : 0.0;
        const double l = fmax(xv, 0.0) - xv * yv + log1p(exp(-fabs(xv)));
        acc += (w ? (double)w[i] : 1.0) * (double)(float)l;
    }
    loss_block_finish(acc, avg_factor, out);
}

// mmdet smooth_l1_loss(beta): |d| < beta ? 0.5 d^2 / beta : |d| - 0.5 beta, element-wise fp32 like the reference
__global__ __launch_bounds__(LOSS_THREADS) void smooth_l1_sum_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                                     const float* __restrict__ w, long long n, float beta,
                                                                     double avg_factor, float* __restrict__ out) {
    double acc = 0.0;
    for (long long i = threadIdx.x; i < n; i += LOSS_THREADS) {
        const float d = fabsf(pred[i] - tgt[i]);
        const float l = d < beta ? 0.5f * d * d / beta : d - 0.5f * beta;
        acc += (double)(w ? w[i] * l : l);
    }
    loss_block_finish(acc, avg_factor, out);
}

// F.cross_entropy(logits [n,C], labels): logsumexp(row) - row[label]
__global__ __launch_bounds__(LOSS_THREADS) void softmax_ce_sum_kernel(const float* __restrict__ logits,
                                                                      const int64_t* __restrict__ labels,
                                                                      const float* __restrict__ w, int n, int C,
                                                                      double avg_factor, float* __restrict__ out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += LOSS_THREADS) {
        const float* r = logits + (size_t)i * C;
        float m = r[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, r[c]);
        double s = 0.0;
        for (int c = 0; c < C; ++c) s += exp((double)(r[c] - m));
        const int64_t lab = labels[i];
        if (lab < 0 || lab >= C) continue;                       // ignore_index
        const float l = (float)((double)m + log(s) - (double)r[lab]);
        acc += (double)(w ? w[i] * l : l);
    }
    loss_block_finish(acc, avg_factor, out);
}

extern "C" int fgn_bce_logits_sum_f32(const float* x, const float* y, const float* w, long long n, float y_threshold,
                                      double avg_factor, float* out, hipStream_t stream) {
    if (!out || (n > 0 && (!x || !y))) return FGN_ERR_ARG;
    hipLaunchKernelGGL(bce_sum_kernel, dim3(1), dim3(LOSS_THREADS), 0, stream, x, y, w, n, y_threshold, avg_factor, out);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

extern "C" int fgn_smooth_l1_sum_f32(const float* pred, const float* target, const float* w, long long n, float beta,
                                     double avg_factor, float* out, hipStream_t stream) {
    if (!out || (n > 0 && (!pred || !target))) return FGN_ERR_ARG;
    if (!(beta > 0.f)) return FGN_ERR_SHAPE;
    hipLaunchKernelGGL(smooth_l1_sum_kernel, dim3(1), dim3(LOSS_THREADS), 0, stream, pred, target, w, n, beta,
                       avg_factor, out);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

extern "C" int fgn_softmax_ce_sum_f32(const float* logits, const int64_t* labels, const float* w, int n, int n_classes,
                                      double avg_factor, float* out, hipStream_t stream) {
    if (!out || (n > 0 && (!logits || !labels))) return FGN_ERR_ARG;
    if (n_classes < 1) return FGN_ERR_SHAPE;
    hipLaunchKernelGGL(softmax_ce_sum_kernel, dim3(1), dim3(LOSS_THREADS), 0, stream, logits, labels, w, n, n_classes,
                       avg_factor, out);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm2d in training mode on NHWC rows x [P, C] (P = samples x pixels).
//   stats    per-channel sum and sum of squares in fp64 over row chunks (HBM-bound: one read of x), partials
//            [chunks][C][2] reduced in chunk order -> mean, biased variance; running estimates updated with
//            momentum (unbiased variance, torch.nn.BatchNorm2d)
//   apply    y = (x - mean) * rsqrt(var + eps) * gamma + beta (+ residual) (ReLU): one read, one write
// ---------------------------------------------------------------------------------------------------------
constexpr int BN_CHUNKS = 64;

__global__ __launch_bounds__(256) void bn_partial_kernel(const float4* __restrict__ x, int P, int C4,
                                                         double* __restrict__ partial) {
    // 64 lanes x 4 channels across, 4 row phases down
    __shared__ double red[4][64][8];
    const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c4 = blockIdx.x * 64 + lane;
    const int rows_per = (P + BN_CHUNKS - 1) / BN_CHUNKS;
    const int r0 = blockIdx.y * rows_per, r1 = min(P, r0 + rows_per);
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (c4 < C4) {
        for (int r = r0 + ph; r < r1; r += 4) {
            const float4 v = x[(size_t)r * C4 + c4];
            s[0] += v.x; q[0] += (double)v.x * v.x;
            s[1] += v.y; q[1] += (double)v.y * v.y;
            s[2] += v.z; q[2] += (double)v.z * v.z;
            s[3] += v.w; q[3] += (double)v.w * v.w;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[ph][lane][j] = s[j]; red[ph][lane][4 + j] = q[j]; }
    __syncthreads();
    if (ph == 0 && c4 < C4) {
        double* o = partial + ((size_t)blockIdx.y * C4 + c4) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = ((red[0][lane][j] + red[1][lane][j]) + red[2][lane][j]) + red[3][lane][j];
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ partial, int P, int C, float* __restrict__ mean,
                                   float* __restrict__ var, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int C4 = C / 4, c4 = c >> 2, j = c & 3;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < BN_CHUNKS; ++k) {
        const double* o = partial + ((size_t)k * C4 + c4) * 8;
        s += o[j];
        q += o[4 + j];
    }
    const double m = s / P;
    double v = q / P - m * m;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)m;
    var[c] = (float)v;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    if (running_var) {
        const double unbiased = P > 1 ? v * ((double)P / (double)(P - 1)) : v;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

__global__ void bn_apply_kernel(const float4* __restrict__ x, const float4* __restrict__ mean, const float4* __restrict__ var,
                                const float4* __restrict__ gamma, const float4* __restrict__ beta, float eps,
                                const float4* __restrict__ residual, int relu, float4* __restrict__ out, long long total4,
                                int C4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const float4 v = x[i], m = mean[c], s2 = var[c], g = gamma[c], b = beta[c];
        float4 y;
        y.x = (v.x - m.x) * (1.f / sqrtf(s2.x + eps)) * g.x + b.x;
        y.y = (v.y - m.y) * (1.f / sqrtf(s2.y + eps)) * g.y + b.y;
        y.z = (v.z - m.z) * (1.f / sqrtf(s2.z + eps)) * g.z + b.z;
        y.w = (v.w - m.w) * (1.f / sqrtf(s2.w + eps)) * g.w + b.w;
        if (residual) {
            const float4 r = residual[i];
            y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
        }
        if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
        out[i] = y;
    }
}

extern "C" size_t fgn_bn_train_scratch_bytes(int C) { return (size_t)BN_CHUNKS * C * 2 * sizeof(double); }

// x, out [P, C] (out may alias x); mean / var [C] receive the batch statistics; running_* (optional) are updated
extern "C" int fgn_bn_train_f32(const float* x, int P, int C, const float* gamma, const float* beta, float eps,
                                float momentum, float* running_mean, float* running_var, const float* residual,
                                int relu, void* scratch, float* mean, float* var, float* out, hipStream_t stream) {
    if (!x || !gamma || !beta || !scratch || !mean || !var || !out) return FGN_ERR_ARG;
    if (C % 4 || P <= 0) return FGN_ERR_SHAPE;
    const int C4 = C / 4;
    double* partial = reinterpret_cast<double*>(scratch);
    hipLaunchKernelGGL(bn_partial_kernel, dim3(cdiv(C4, 64), BN_CHUNKS), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(x), P, C4, partial);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, partial, P, C, mean, var,
                       running_mean, running_var, momentum);
    const long long total4 = (long long)P * C4;
    const int grid = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<const float4*>(mean), reinterpret_cast<const float4*>(var),
                       reinterpret_cast<const float4*>(gamma), reinterpret_cast<const float4*>(beta), eps,
                       reinterpret_cast<const float4*>(residual), relu, reinterpret_cast<float4*>(out), total4, C4);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
