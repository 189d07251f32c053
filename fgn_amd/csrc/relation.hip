// Relation-guided box head, fused tail:
//   x[r,n]  = Q[r] + S[img(r)*N + n]          (the two halves of the 2048->1024 1x1 conv)
//   y       = ReLU(GroupNorm32(x))            fgn_roi_head.py:272-274
//   pooled  = mean_{7x7}(y)                   mmdet BBoxHead.forward (with_avg_pool)
//   cls,reg = fc_cls(pooled), fc_reg(pooled)  -> [R*N,2], [R*N,4]   fgn_roi_head.py:338
//
// The reference concatenates (RoI feature, class-mean support feature) on channels and
// runs one 2048->1024 1x1 conv per (RoI, class) (fgn_roi_head.py:260-270).  The conv is
// linear, so W = [Wq | Ws] is split: Q = Wq * roi_feat is computed once per RoI and
// S = Ws * support + bias once per class by the MFMA conv kernel; this kernel does the
// per-(RoI,class) part, which is HBM/L2-bound: it reads Q once (not N times), never
// materialises the [R*N,2048,7,7] concat or the [R*N,1024,7,7] normalised tensor, and
// writes 6 floats per (RoI, class).
//
// (mapping of work to waves: see relation_head_kernel)
#include "common.h"

constexpr int REL_MAX_N = 8;
constexpr int REL_WAVES = 8;     // 32-channel slabs per workgroup: one per wave

// Mapping: a wave owns one 32-channel slab (one GroupNorm group of the reference's GN(32, 1024); 2 or 4 groups for
// narrower heads) x 49 pixels of one RoI: lane = (pixel slot 0..7, channel quad 0..7), 7 float4 per lane stay in
// registers across the N classes.  A workgroup holds REL_WAVES slabs of one RoI; grid = R x (C / 32 / REL_WAVES), i.e.
// 1200 workgroups for 300 RoIs x 1024 channels - round 2 ran one workgroup per RoI with every wave looping over four
// slabs x N classes, a serial chain of dependent load -> shuffle-reduce -> normalise steps that took 90 us for 60 MB.
// Statistics are two-pass in registers (mean, then centred sum of squares) with wavefront xor-shuffle reductions.
// The fc products are reduced per workgroup in a fixed order and written as partials [R][chunks][N][6];
// relation_fc_finalize_kernel adds the chunks in chunk order and the bias (bit-reproducible: no float atomics).
__global__ __launch_bounds__(64 * REL_WAVES) void relation_head_kernel(
    const float* __restrict__ Q, const float* __restrict__ S, const float* __restrict__ rois,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ fcw,
    float* __restrict__ fc_part, const int32_t* __restrict__ n_rois_dev, int n_rois, int n_ways, int C, int gw, float eps,
    int chunks, float* __restrict__ rel_out) {
    constexpr int P = 49;
    // a wave covers 32 consecutive channels = 32 / gw GroupNorm groups (gw = channels per group: 8, 16 or 32);
    // statistics are reduced over the lanes of one group: all 8 pixel slots (lane bits 3..5) and the channel
    // quads of the group (lane bits below log2(gw / 4))
    auto group_sum = [gw](float v) {
#pragma unroll
        for (int off = 32; off >= 8; off >>= 1) v += __shfl_xor(v, off, 64);
        for (int off = (gw >> 3); off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        return v;
    };
    __shared__ float fc_acc[REL_WAVES][REL_MAX_N][6];
    const int r = blockIdx.x / chunks, chunk = blockIdx.x - r * chunks;
    int nr = n_rois;
    if (n_rois_dev) nr = min(nr, *n_rois_dev);
    if (r >= nr) return;

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int quad = lane & 7, slot = lane >> 3;
    const int img = (int)rois[(size_t)r * 5];
    const float inv_cnt = 1.f / ((float)gw * (float)P);
    const int g = chunk * REL_WAVES + wv;       // this wave's 32-channel slab
    const bool active = g * 32 < C;
    if (active) {
        const int c = g * 32 + quad * 4;
        float4 q[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int p = slot + 8 * i;
            q[i] = (p < P) ? *reinterpret_cast<const float4*>(Q + ((size_t)r * P + p) * C + c)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
        const float4 be = *reinterpret_cast<const float4*>(beta + c);
        float4 fw[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) fw[j] = *reinterpret_cast<const float4*>(fcw + (size_t)j * C + c);

        for (int n = 0; n < n_ways; ++n) {
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
            float4 pool = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int p = slot + 8 * i;
                if (p < P) {
                    const float4 y = make_float4(fmaxf((x[i].x - mean) * rstd * ga.x + be.x, 0.f),
                                                 fmaxf((x[i].y - mean) * rstd * ga.y + be.y, 0.f),
                                                 fmaxf((x[i].z - mean) * rstd * ga.z + be.z, 0.f),
                                                 fmaxf((x[i].w - mean) * rstd * ga.w + be.w, 0.f));
                    pool.x += y.x; pool.y += y.y; pool.z += y.z; pool.w += y.w;
                    // parity tests only: the relation feature map the reference materialises (fgn_roi_head.py:274)
                    if (rel_out) *reinterpret_cast<float4*>(rel_out + (((size_t)r * n_ways + n) * P + p) * C + c) = y;
                }
            }
            // reduce over the 8 pixel slots (lanes differing in bits 3..5), then over quads
            float dots[6];
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                pool.x += __shfl_xor(pool.x, off, 64);
                pool.y += __shfl_xor(pool.y, off, 64);
                pool.z += __shfl_xor(pool.z, off, 64);
                pool.w += __shfl_xor(pool.w, off, 64);
            }
            const float ip = 1.f / (float)P;
            pool.x *= ip; pool.y *= ip; pool.z *= ip; pool.w *= ip;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float d = (pool.x * fw[j].x + pool.y * fw[j].y) + (pool.z * fw[j].z + pool.w * fw[j].w);
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                d += __shfl_xor(d, 4, 64);
                dots[j] = d;
            }
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < 6; ++j) fc_acc[wv][n][j] = dots[j];
            }
        }
    }
    __syncthreads();
    if (t < n_ways * 6) {
        const int n = t / 6, j = t - n * 6;
        const int waves = min(REL_WAVES, C / 32 - chunk * REL_WAVES);
        float v = 0.f;
        for (int w = 0; w < waves; ++w) v += fc_acc[w][n][j];     // fixed order
        fc_part[(((size_t)r * chunks + chunk) * n_ways + n) * 6 + j] = v;
    }
}

// cls / reg of one (RoI, class) = bias + the channel-chunk partials in chunk order
__global__ void relation_fc_finalize_kernel(const float* __restrict__ fc_part, const float* __restrict__ fcb,
                                            float* __restrict__ cls_out, float* __restrict__ reg_out,
                                            const int32_t* __restrict__ n_rois_dev, int n_rois, int n_ways, int chunks) {
    int nr = n_rois;
    if (n_rois_dev) nr = min(nr, *n_rois_dev);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nr * n_ways * 6) return;
    const int j = i % 6, row = i / 6;               // row = r * n_ways + n
    const int r = row / n_ways, n = row - r * n_ways;
    float v = 0.f;
    for (int ch = 0; ch < chunks; ++ch) v += fc_part[(((size_t)r * chunks + ch) * n_ways + n) * 6 + j];
    v += fcb[j];
    if (j < 2) cls_out[(size_t)row * 2 + j] = v;
    else reg_out[(size_t)row * 4 + (j - 2)] = v;
}

extern "C" size_t fgn_relation_gn_head_scratch_bytes(int n_rois, int n_ways, int C) {
    const int chunks = cdiv(C / 32, REL_WAVES);
    return (size_t)n_rois * chunks * n_ways * 6 * sizeof(float);
}

extern "C" int fgn_relation_gn_head_f32(const float* Q, const float* S, const float* rois, const float* gn_weight,
                                        const float* gn_bias, const float* fc_weight, const float* fc_bias,
                                        float* cls_out, float* reg_out, const int32_t* n_rois_dev, int n_rois,
                                        int n_ways, int C, int gn_groups, int roi_size, float eps,
                                        float* rel_out_debug, float* scratch, hipStream_t stream) {
    if (!Q || !S || !rois || !gn_weight || !gn_bias || !fc_weight || !fc_bias || !cls_out || !reg_out || !scratch)
        return FGN_ERR_ARG;
    if (roi_size != 7 || gn_groups <= 0 || C % 32 != 0 || C % gn_groups != 0 || n_ways < 1 || n_ways > REL_MAX_N)
        return FGN_ERR_SHAPE;
    const int gw = C / gn_groups;              // channels per GroupNorm group
    if (gw != 8 && gw != 16 && gw != 32) return FGN_ERR_SHAPE;
    if (n_rois == 0) return FGN_OK;
    const int chunks = cdiv(C / 32, REL_WAVES);
    hipLaunchKernelGGL(relation_head_kernel, dim3(n_rois * chunks), dim3(64 * REL_WAVES), 0, stream, Q, S, rois, gn_weight,
                       gn_bias, fc_weight, scratch, n_rois_dev, n_rois, n_ways, C, gw, eps, chunks, rel_out_debug);
    FGN_LAUNCH_CHECK();
    const int total = n_rois * n_ways * 6;
    hipLaunchKernelGGL(relation_fc_finalize_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, scratch, fc_bias, cls_out,
                       reg_out, n_rois_dev, n_rois, n_ways, chunks);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
