// Box-head post-processing for one image, on device, no host sync:
//   count_modified_cls_bbox  (fgn_roi_head.py:302-326)  class logits = column 1 of each of the
//        N guided passes, background logit = column 0 of the arg-max pass
//   softmax over (c_0..c_{N-1}, bg), delta2bbox with stds (.1,.1,.2,.2), clip to image
//        (mmdet BBoxHead.get_bboxes, reached from fgn_roi_head.py:606-613)
//   multiclass_nms: score > score_thr, class-aware NMS through the coordinate-offset trick
//        (boxes + label*(max_coordinate+1), fp32), keep the first max_per_img
// Softmax is evaluated in fp64 and rounded once (oracle convention); every other op is
// fp32 in the oracle's order (-ffp-contract=off).
#include "post_common.h"

constexpr int DET_MAX_N = 8;

// One workgroup per image (blockIdx.x); every pointer is the first image's, image i sits i * (its per-image extent) on.
struct DetParams {
    const float* rois;       // [B][R][5] (the proposals of each image)
    const float* cls_raw;    // [B][R*N][2]
    const float* reg_raw;    // [B][R*N][4]
    const int32_t* n_rois_dev;   // optional [B]
    float4* cand_boxes;      // scratch [B][2][cap]: decoded boxes in candidate order (r*N+n) | offset boxes in score order
    float* det_bboxes;       // out [B][max_out][5]
    float* mask_rois;        // optional out [B][max_out][5] = (img_index, x1, y1, x2, y2): the mask branch's bbox2roi (fgn_roi_head.py:654)
    int img_index0;          // image index of the first image (image i carries img_index0 + i)
    int64_t* det_labels;     // out [B][max_out]
    int32_t* n_dets;         // out [B]
    float* dbg_scores;       // optional out [R][N+1] softmax scores of image 0 (+ 16 stamp words), or null
    int n_rois, N, cap;
    int xch_off;             // byte offset of the sort's exchange buffer in LDS (8-byte aligned)
    float img_h, img_w;
    float mean[4], stdv[4];
    float max_ratio, score_thr, iou_thr;
    int max_out;
};

// diagnostic phase stamps (100 MHz realtime counter) written behind the debug score table
#define DET_STAMP(slot)                                                                          \
    do {                                                                                         \
        if (dbg_scores && threadIdx.x == 0)                                                    \
            reinterpret_cast<int32_t*>(dbg_scores)[(size_t)p.n_rois * (p.N + 1) + (slot)] =      \
                (int32_t)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);                        \
    } while (0)

__global__ __launch_bounds__(POST_THREADS) void det_post_kernel(const DetParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(lds_raw);
    int* misc = reinterpret_cast<int*>(keys + p.cap);   // [0] counter, [2] kept count, [4..20) wave scratch
    unsigned long long* sup = reinterpret_cast<unsigned long long*>(misc + 64);   // 8-byte aligned
    NmsBox* kept = reinterpret_cast<NmsBox*>(sup + NMS_ROUND * NMS_WORDS);
    NmsBox* cand = kept + p.max_out;
    int* flags = reinterpret_cast<int*>(cand + NMS_ROUND);
    int* keep = flags + NMS_ROUND + 2;
    float* cand_score = reinterpret_cast<float*>(keep + p.max_out);   // [cap]
    uint64_t* xch = reinterpret_cast<uint64_t*>(lds_raw + p.xch_off);  // [2][POST_THREADS] exchange buffer of the register sort

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int img = blockIdx.x;
    const int N = p.N;
    int R = p.n_rois;
    if (p.n_rois_dev) R = min(R, p.n_rois_dev[img]);
    const float* rois = p.rois + (size_t)img * p.n_rois * 5;
    const float* cls_raw = p.cls_raw + (size_t)img * p.n_rois * N * 2;
    const float* reg_raw = p.reg_raw + (size_t)img * p.n_rois * N * 4;
    float4* cand_boxes = p.cand_boxes + (size_t)img * 2 * p.cap;
    float4* sorted_nms_boxes = cand_boxes + p.cap;
    float* det_bboxes = p.det_bboxes + (size_t)img * p.max_out * 5;
    int64_t* det_labels = p.det_labels + (size_t)img * p.max_out;
    float* dbg_scores = img == 0 ? p.dbg_scores : nullptr;
    const int n_cand = R * N;

    DET_STAMP(0);
    if (t == 0) misc[0] = 0;
    for (int i = t; i < p.cap; i += POST_THREADS) keys[i] = ~0ull;
    __syncthreads();

    // ---- 1. scores + decoded boxes per (roi, class); local max of valid box coordinates -------
    float local_max = -INFINITY;
    for (int r = t; r < R; r += POST_THREADS) {
        float logit[DET_MAX_N + 1];
        int best = 0;
#pragma unroll
        for (int n = 0; n < DET_MAX_N; ++n) {
            if (n < N) {
                logit[n] = cls_raw[((size_t)r * N + n) * 2 + 1];
                if (n > 0 && logit[n] > logit[best]) best = n;   // first maximal index
            }
        }
        float bg = 0.f;
#pragma unroll
        for (int n = 0; n < DET_MAX_N; ++n)
            if (n == best) bg = cls_raw[((size_t)r * N + n) * 2 + 0];
        // softmax in fp64 over N+1 logits
        double mx = (double)bg;
#pragma unroll
        for (int n = 0; n < DET_MAX_N; ++n)
            if (n < N) mx = fmax(mx, (double)logit[n]);
        double e[DET_MAX_N + 1];
        double sum = 0.0;
#pragma unroll
        for (int n = 0; n < DET_MAX_N; ++n)
            if (n < N) {
                e[n] = exp((double)logit[n] - mx);
                sum += e[n];
            }
        const double ebg = exp((double)bg - mx);
        sum += ebg;
        if (dbg_scores) dbg_scores[(size_t)r * (N + 1) + N] = (float)(ebg / sum);

        const float* roi = rois + (size_t)r * 5;
        const float rx1 = roi[1], ry1 = roi[2], rx2 = roi[3], ry2 = roi[4];
        const float pcx = (rx1 + rx2) * 0.5f, pcy = (ry1 + ry2) * 0.5f;
        const float pw = rx2 - rx1, ph = ry2 - ry1;
#pragma unroll
        for (int n = 0; n < DET_MAX_N; ++n) {
            if (n >= N) continue;
            const float score = (float)(e[n] / sum);
            if (dbg_scores) dbg_scores[(size_t)r * (N + 1) + n] = score;
            const float* d = reg_raw + ((size_t)r * N + n) * 4;
            const float dx = d[0] * p.stdv[0] + p.mean[0];
            const float dy = d[1] * p.stdv[1] + p.mean[1];
            float dw = d[2] * p.stdv[2] + p.mean[2];
            float dh = d[3] * p.stdv[3] + p.mean[3];
            const float dxw = pw * dx, dyh = ph * dy;
            dw = fminf(fmaxf(dw, -p.max_ratio), p.max_ratio);
            dh = fminf(fmaxf(dh, -p.max_ratio), p.max_ratio);
            const float gcx = pcx + dxw, gcy = pcy + dyh;
            const float gw = pw * exp32(dw), gh = ph * exp32(dh);
            const float hw = gw * 0.5f, hh = gh * 0.5f;
            float x1 = gcx - hw, y1 = gcy - hh, x2 = gcx + hw, y2 = gcy + hh;
            x1 = fminf(fmaxf(x1, 0.f), p.img_w); x2 = fminf(fmaxf(x2, 0.f), p.img_w);
            y1 = fminf(fmaxf(y1, 0.f), p.img_h); y2 = fminf(fmaxf(y2, 0.f), p.img_h);
            const int ci = r * N + n;
            cand_boxes[ci] = make_float4(x1, y1, x2, y2);
            cand_score[ci] = score;
            if (score > p.score_thr) {
                keys[atomicAdd(&misc[0], 1)] = sort_key(score, (uint32_t)ci);
                local_max = fmaxf(local_max, fmaxf(fmaxf(x1, y1), fmaxf(x2, y2)));
            }
        }
    }
    // block max
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local_max = fmaxf(local_max, __shfl_xor(local_max, off, 64));
    float* wave_max = reinterpret_cast<float*>(misc + 4);
    if (lane == 0) wave_max[wv] = local_max;
    __syncthreads();
    float max_coord = wave_max[0];
    for (int w = 1; w < POST_WAVES; ++w) max_coord = fmaxf(max_coord, wave_max[w]);
    const int n_valid = misc[0];
    (void)n_cand;
    DET_STAMP(1);

    // ---- 2. stable score-descending order ---------------------------------------------------------
    if (n_valid <= POST_THREADS) {
        // one key per thread, sorted in registers (cfg3: 900 candidates; the all-LDS network took 26 of this kernel's
        // 54 us)
        const uint64_t mine = keys[t];
        const uint64_t sorted = block_bitonic_sort_regs(mine, xch);
        keys[t] = sorted;
        __syncthreads();
    } else {
        int sort_n = POST_THREADS;
        while (sort_n < n_valid) sort_n <<= 1;
        block_bitonic_sort(keys, sort_n);
    }
    DET_STAMP(2);

    // ---- 3. class-aware offset boxes in sorted order ----------------------------------------------
    const float off_unit = max_coord + 1.0f;
    for (int i = t; i < n_valid; i += POST_THREADS) {
        const uint32_t ci = key_index(keys[i]);
        const float4 b = cand_boxes[ci];
        const float o = (float)(ci % N) * off_unit;
        sorted_nms_boxes[i] = make_float4(b.x + o, b.y + o, b.z + o, b.w + o);
    }
    __syncthreads();

    DET_STAMP(3);
    // ---- 4. NMS + output --------------------------------------------------------------------------
    const int n_keep = nms_sorted_block(sorted_nms_boxes, n_valid, p.iou_thr, p.max_out, keep, kept, cand, sup, flags);
    DET_STAMP(6);
    for (int i = t; i < p.max_out; i += POST_THREADS) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        float s = 0.f;
        int64_t lab = 0;
        if (i < n_keep) {
            const uint32_t ci = key_index(keys[keep[i]]);
            b = cand_boxes[ci];
            s = cand_score[ci];
            lab = ci % N;
        }
        det_bboxes[i * 5 + 0] = b.x; det_bboxes[i * 5 + 1] = b.y;
        det_bboxes[i * 5 + 2] = b.z; det_bboxes[i * 5 + 3] = b.w;
        det_bboxes[i * 5 + 4] = s;
        det_labels[i] = lab;
        if (p.mask_rois) {
            float* r = p.mask_rois + ((size_t)img * p.max_out + i) * 5;
            r[0] = (float)(p.img_index0 + img); r[1] = b.x; r[2] = b.y; r[3] = b.z; r[4] = b.w;
        }
    }
    if (t == 0) p.n_dets[img] = n_keep;
    DET_STAMP(4);
    if (dbg_scores && t == 0) reinterpret_cast<int32_t*>(dbg_scores)[(size_t)p.n_rois * (p.N + 1) + 5] = n_valid;
}

extern "C" size_t fgn_det_post_scratch_bytes(int max_rois, int n_ways) {
    int cap = POST_THREADS;
    while (cap < max_rois * n_ways) cap <<= 1;
    return (size_t)cap * 2 * sizeof(float4);
}

extern "C" int fgn_det_post_f32(const float* rois, const float* cls_raw, const float* reg_raw,
                                const int32_t* n_rois_dev, void* scratch, float* det_bboxes, float* mask_rois_out, int img_index,
                                int64_t* det_labels,
                                int32_t* n_dets, float* dbg_scores, int batch, int n_rois, int n_ways, float img_h, float img_w,
                                const float* means4, const float* stds4, float max_ratio, float score_thr,
                                float iou_thr, int max_per_img, hipStream_t stream) {
    if (!rois || !cls_raw || !reg_raw || !scratch || !det_bboxes || !det_labels || !n_dets || !means4 || !stds4)
        return FGN_ERR_ARG;
    if (n_ways < 1 || n_ways > DET_MAX_N || max_per_img < 1 || max_per_img > 1024 || batch < 1) return FGN_ERR_SHAPE;
    int cap = POST_THREADS;
    while (cap < n_rois * n_ways) cap <<= 1;
    if (cap > 8192) return FGN_ERR_SHAPE;
    DetParams p;
    p.rois = rois; p.cls_raw = cls_raw; p.reg_raw = reg_raw; p.n_rois_dev = n_rois_dev;
    p.cand_boxes = reinterpret_cast<float4*>(scratch);
    p.det_bboxes = det_bboxes; p.mask_rois = mask_rois_out; p.img_index0 = img_index;
    p.det_labels = det_labels; p.n_dets = n_dets; p.dbg_scores = dbg_scores;
    p.n_rois = n_rois; p.N = n_ways; p.cap = cap; p.img_h = img_h; p.img_w = img_w;
    for (int i = 0; i < 4; ++i) { p.mean[i] = means4[i]; p.stdv[i] = stds4[i]; }
    p.max_ratio = max_ratio; p.score_thr = score_thr; p.iou_thr = iou_thr; p.max_out = max_per_img;
    size_t lds = (size_t)cap * 8 + 64 * 4 + (size_t)max_per_img * sizeof(NmsBox) +
                 NMS_ROUND * sizeof(NmsBox) + NMS_ROUND * NMS_WORDS * 8 + (NMS_ROUND + 2) * 4 + (size_t)max_per_img * 4 + (size_t)cap * 4;
    lds = (lds + 7) / 8 * 8;
    p.xch_off = (int)lds;
    lds += 2 * POST_THREADS * sizeof(uint64_t);
    static unsigned long long lds_ok = 0ull;
    const hipError_t attr = fgn_allow_full_lds(reinterpret_cast<const void*>(det_post_kernel), &lds_ok);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL(det_post_kernel, dim3(batch), dim3(POST_THREADS), lds, stream, p);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
