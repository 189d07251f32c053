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
constexpr int REL_WAVES = 4;     // 32-channel slabs per workgroup: one per wave

// Lane exchanges through the DPP path of the VALU (no LDS crossbar): the lane whose index differs in bit 0 / bit 1
// (quad permutes), the other quad of an aligned group of 8 (half-row mirror: exact once the 4 lanes of every quad
// hold equal values), the lane 8 on in its row of 16.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_xor1(float v) { return dpp_f32<0xB1>(v); }        // quad_perm:[1,0,3,2]
__device__ __forceinline__ float lane_xor2(float v) { return dpp_f32<0x4E>(v); }        // quad_perm:[2,3,0,1]
__device__ __forceinline__ float other_quad(float v) { return dpp_f32<0x141>(v); }      // row_half_mirror
__device__ __forceinline__ float lane_xor8(float v) { return dpp_f32<0x128>(v); }       // row_ror:8
// sum over the four rows of 16 lanes, every lane of a row holding the row's value: scalar reads, the result is uniform
__device__ __forceinline__ float sum_rows(float v) {
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
// v + the value of the lane 16 / 32 on or back (butterfly step across rows / halves): the gfx950 row and half swaps
typedef unsigned rel_v2u __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float add_xor16(float v) {
    const rel_v2u r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float add_xor32(float v) {
    const rel_v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Mapping: a wave owns one 32-channel slab (one GroupNorm group of the reference's GN(32, 1024); 2 or 4 groups for
// narrower heads) x 49 pixels of one RoI: lane = (pixel slot 0..7, channel quad 0..7), 7 float4 per lane stay in
// registers across the N classes.  A workgroup holds REL_WAVES slabs of one RoI; grid = R x (C / 32 / REL_WAVES), i.e.
// 2400 workgroups of 4 waves for 300 RoIs x 1024 channels.  History: one workgroup per RoI with every wave looping
// over four slabs x N classes took 90 us for 60 MB (round 2); one slab per wave in 8-wave workgroups 48 us (168
// VGPRs: one workgroup per CU, every class a serial load -> LDS-crossbar reductions chain).  Now the classes are
// unrolled (template NW) with the NEXT class's support slab in flight while the current one is normalised, the
// reductions run on DPP + scalar lane reads (no ds_bpermute where the group is the whole wave), the fc products are
// taken per lane BEFORE the reduction (six wave sums instead of a pooled vector + six dot reductions), and the register
// budget is held at 168 (3 waves per SIMD; 150 used at three classes).
// Statistics are two-pass in registers (mean, then centred sum of squares).
// The fc products are reduced per workgroup in a fixed order and written as partials [R][chunks][N][6];
// relation_fc_finalize_kernel adds the chunks in chunk order and the bias (bit-reproducible: no float atomics).
template <int NW, bool REL_OUT>
__global__ __launch_bounds__(64 * REL_WAVES, NW <= 4 ? 3 : 2) void relation_head_kernel(
    const float* __restrict__ Q, const float* __restrict__ S, const float* __restrict__ rois,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ fcw,
    float* __restrict__ fc_part, const int32_t* __restrict__ n_rois_dev, int n_rois, int C, int gw, float eps,
    int chunks, float* __restrict__ rel_out) {
    constexpr int P = 49;
    // a wave covers 32 consecutive channels = 32 / gw GroupNorm groups (gw = channels per group: 8, 16 or 32);
    // statistics are reduced over the lanes of one group: all 8 pixel slots (lane bits 3..5) and the channel
    // quads of the group (lane bits below log2(gw / 4))
    auto group_sum = [gw](float v) {
        v += lane_xor1(v);                       // gw >= 8: two quads
        if (gw >= 16) v += lane_xor2(v);
        if (gw == 32) v += other_quad(v);
        v += lane_xor8(v);                       // pixel slots 0..7: lane bits 3..5
        if (gw == 32) return sum_rows(v);        // the group is the whole wave
        return add_xor32(add_xor16(v));
    };
    __shared__ float fc_acc[REL_WAVES][NW][6];
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
        // pixel 48 = slot 0 of the seventh sweep; the other slots have 6 pixels and carry zeros (masked by m7) there
        const bool last = slot == 0;
        const float m7s = last ? 1.f : 0.f;
        const v2f m7 = {m7s, m7s};
        struct Slab { v2f lo[7], hi[7]; };      // 7 pixels x 4 channels as pairs: the arithmetic below is v_pk_*_f32
        auto load_slab = [&](const float* base, Slab& o) {
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < 6 || last) v = *reinterpret_cast<const float4*>(base + (size_t)(slot + 8 * i) * C);
                o.lo[i] = v2f{v.x, v.y};
                o.hi[i] = v2f{v.z, v.w};
            }
        };
        Slab q, sbuf[2];
        load_slab(Q + (size_t)r * P * C + c, q);
        load_slab(S + (size_t)(img * NW) * P * C + c, sbuf[0]);
        const float4 ga4 = *reinterpret_cast<const float4*>(gamma + c);
        const float4 be4 = *reinterpret_cast<const float4*>(beta + c);
        const v2f ga_lo = {ga4.x, ga4.y}, ga_hi = {ga4.z, ga4.w}, be_lo = {be4.x, be4.y}, be_hi = {be4.z, be4.w};
        const v2f zero2 = {0.f, 0.f};

#pragma unroll
        for (int n = 0; n < NW; ++n) {
            Slab& x = sbuf[n & 1];              // x = q + s in place; the other buffer receives the next class
            if (n + 1 < NW) load_slab(S + (size_t)(img * NW + n + 1) * P * C + c, sbuf[(n + 1) & 1]);
            v2f sum2 = zero2;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                x.lo[i] += q.lo[i];
                x.hi[i] += q.hi[i];
                sum2 += x.lo[i];
                sum2 += x.hi[i];
            }
            const float mean = group_sum(sum2.x + sum2.y) * inv_cnt;
            const v2f mean2 = {mean, mean};
            v2f sq2 = zero2;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                x.lo[i] -= mean2;
                x.hi[i] -= mean2;
                if (i == 6) {                   // no pixel there for slots 1..7
                    x.lo[i] *= m7;
                    x.hi[i] *= m7;
                }
                sq2 = __builtin_elementwise_fma(x.lo[i], x.lo[i], sq2);
                sq2 = __builtin_elementwise_fma(x.hi[i], x.hi[i], sq2);
            }
            const float var = group_sum(sq2.x + sq2.y) * inv_cnt;
            const float rstd = 1.f / sqrtf(var + eps);
            const v2f rg_lo = ga_lo * rstd, rg_hi = ga_hi * rstd;
            v2f pool_lo = zero2, pool_hi = zero2;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                v2f y_lo = __builtin_elementwise_max(__builtin_elementwise_fma(x.lo[i], rg_lo, be_lo), zero2);
                v2f y_hi = __builtin_elementwise_max(__builtin_elementwise_fma(x.hi[i], rg_hi, be_hi), zero2);
                if (i == 6) {
                    y_lo *= m7;
                    y_hi *= m7;
                }
                pool_lo += y_lo;
                pool_hi += y_hi;
                // parity tests only: the relation feature map the reference materialises (fgn_roi_head.py:274)
                if (REL_OUT && (i < 6 || last))
                    *reinterpret_cast<float4*>(rel_out + (((size_t)r * NW + n) * P + slot + 8 * i) * C + c) =
                        make_float4(y_lo.x, y_lo.y, y_hi.x, y_hi.y);
            }
            // average pool: sum over the 8 pixel slots (lane bits 3..5) - every lane ends with the pooled values of its
            // four channels -, then fc_cls / fc_reg: 4 products per lane and a sum over the 8 channel quads (lane bits 0..2)
            float pl[4] = {pool_lo.x, pool_lo.y, pool_hi.x, pool_hi.y};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pl[k] += lane_xor8(pl[k]);
                pl[k] = add_xor16(pl[k]);
                pl[k] = add_xor32(pl[k]);
            }
            const float ip = 1.f / (float)P;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float4 fw = *reinterpret_cast<const float4*>(fcw + (size_t)j * C + c);
                float d = (pl[0] * fw.x + pl[1] * fw.y) + (pl[2] * fw.z + pl[3] * fw.w);
                d += lane_xor1(d);
                d += lane_xor2(d);
                d += other_quad(d);
                if (lane == 0) fc_acc[wv][n][j] = d * ip;
            }
        }
    }
    __syncthreads();
    if (t < NW * 6) {
        const int n = t / 6, j = t - n * 6;
        const int waves = min(REL_WAVES, C / 32 - chunk * REL_WAVES);
        float v = 0.f;
        for (int w = 0; w < waves; ++w) v += fc_acc[w][n][j];     // fixed order
        fc_part[(((size_t)r * chunks + chunk) * NW + n) * 6 + j] = v;
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
#define REL_LAUNCH(NW)                                                                                                  \
    case NW:                                                                                                            \
        if (rel_out_debug)                                                                                              \
            hipLaunchKernelGGL((relation_head_kernel<NW, true>), dim3(n_rois * chunks), dim3(64 * REL_WAVES), 0, stream, Q, S, \
                               rois, gn_weight, gn_bias, fc_weight, scratch, n_rois_dev, n_rois, C, gw, eps, chunks,    \
                               rel_out_debug);                                                                          \
        else                                                                                                            \
            hipLaunchKernelGGL((relation_head_kernel<NW, false>), dim3(n_rois * chunks), dim3(64 * REL_WAVES), 0, stream, Q, S, \
                               rois, gn_weight, gn_bias, fc_weight, scratch, n_rois_dev, n_rois, C, gw, eps, chunks,    \
                               rel_out_debug);                                                                          \
        break;
    switch (n_ways) {
        REL_LAUNCH(1) REL_LAUNCH(2) REL_LAUNCH(3) REL_LAUNCH(4) REL_LAUNCH(5) REL_LAUNCH(6) REL_LAUNCH(7) REL_LAUNCH(8)
    }
#undef REL_LAUNCH
    FGN_LAUNCH_CHECK();
    const int total = n_rois * n_ways * 6;
    hipLaunchKernelGGL(relation_fc_finalize_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, scratch, fc_bias, cls_out,
                       reg_out, n_rois_dev, n_rois, n_ways, chunks);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
