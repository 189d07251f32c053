// AG-RPN output merge and proposal generation, all on device.
//   rpn_merge   : per-anchor arg-max over the N guided passes + sigmoid
//                 (AGRPNHead.forward_single, fgn_ag_rpn_head.py:81-113; sigmoid of
//                  mmdet RPNHead._get_bboxes_single, reached from fgn.py:229-235)
//   proposals   : top nms_pre by (score desc, index asc) -> delta2bbox -> w,h > min ->
//                 greedy NMS -> first max_per_img      (mmdet 2.18 RPNHead semantics)
// The reference runs this stage on CPU tensors (img_metas_cpu, fgn.py:209,233); here it
// never leaves the GPU and never synchronises with the host: counts stay in device memory.
// Compiled with -ffp-contract=off: fp32 op order is the oracle's.
#include "post_common.h"

// head : [B*N][HW][CH] NHWC output of the fused 1x1 conv; channels [0,A) = objectness
// logits, channels [A, 5A) = deltas (a*4+d).
__global__ void rpn_merge_kernel(const float* __restrict__ head, float* __restrict__ logits,
                                 float* __restrict__ scores, float4* __restrict__ deltas, int B, int N, int HW,
                                 int A, int CH) {
    const long long total = (long long)B * HW * A;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int a = (int)(i % A);
        const long long r = i / A;
        const int px = (int)(r % HW);
        const int b = (int)(r / HW);
        const float* p0 = head + (((size_t)b * N) * HW + px) * CH;
        float best = p0[a];
        int bn = 0;
        for (int n = 1; n < N; ++n) {
            const float v = p0[(size_t)n * HW * CH + a];
            if (v > best) {   // torch.argmax: first maximal index wins
                best = v;
                bn = n;
            }
        }
        const float* pd = p0 + (size_t)bn * HW * CH + A + a * 4;
        logits[i] = best;
        scores[i] = sigmoid32(best);
        deltas[i] = make_float4(pd[0], pd[1], pd[2], pd[3]);
    }
}

extern "C" int fgn_rpn_merge_f32(const float* head, float* logits, float* scores, float* deltas, int batch,
                                 int n_ways, int HW, int n_anchors, int head_channels, hipStream_t stream) {
    if (!head || !logits || !scores || !deltas) return FGN_ERR_ARG;
    if (head_channels < 5 * n_anchors) return FGN_ERR_SHAPE;
    const long long total = (long long)batch * HW * n_anchors;
    if (total == 0) return FGN_OK;
    const int grid = (int)std::min<long long>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(rpn_merge_kernel, dim3(grid), dim3(256), 0, stream, head, logits, scores,
                       reinterpret_cast<float4*>(deltas), batch, n_ways, HW, n_anchors, head_channels);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ----------------------------------------------------------------------------------------------
// proposals: one workgroup (1024 threads) per image.
// ----------------------------------------------------------------------------------------------
constexpr int RPN_EPT = 64;
constexpr int RPN_FAST_SEL = 1536;   // candidates ranked by the fast first attempt
constexpr int RPN_FAST_CAP = 2048;   // its sort buffer   // scores cached per thread: n_total <= 65536

struct ProposalParams {
    const float* scores;     // [B][n_total]
    const float4* deltas;    // [B][n_total]
    const float4* base_anchors;  // [A]
    float4* sorted_boxes;    // scratch [B][cap]  decoded + filtered boxes in score order
    float* sorted_scores;    // scratch [B][cap]
    float* proposals;        // out [B][max_out][5]
    float* rois;             // optional out [B*max_out][5] = (image index, x1, y1, x2, y2): bbox2roi (fgn_roi_head.py:556)
    int32_t* n_props;        // out [B]
    int32_t* dbg_topk_idx;   // optional out [B][cap] (selected anchor indices, sorted) or null
    int n_total, A, feat_w, stride;
    int nms_pre, cap;        // cap = pow2 >= min(nms_pre, n_total)
    float img_h, img_w;
    float mean[4], stdv[4];
    float max_ratio, min_size, iou_thr;
    int max_out;
};

// diagnostic phase stamps (100 MHz realtime counter) written behind the top-k debug buffer
#define RPN_STAMP(slot)                                                                         \
    do {                                                                                        \
        if (p.dbg_topk_idx && threadIdx.x == 0)                                                 \
            p.dbg_topk_idx[(size_t)blockIdx.x * 8192 + 8192 - 16 + (slot)] =                    \
                (int32_t)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);                       \
    } while (0)

__global__ __launch_bounds__(POST_THREADS) void rpn_proposals_kernel(const ProposalParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    // LDS carve: keys [cap] u64 | hist[256] | misc ; after the sort the key area is dead
    // and the NMS scratch (kept boxes, chunk boxes) is carved after it.
    uint64_t* keys = reinterpret_cast<uint64_t*>(lds_raw);
    int* hist = reinterpret_cast<int*>(keys + p.cap);      // [POST_WAVES][256] per-wave histograms
    int* misc = hist + POST_WAVES * 256;   // [0] counter/digit, [1] rank carry, [2] kept count, [3..] wave sums
    unsigned long long* sup = reinterpret_cast<unsigned long long*>(misc + 64);   // 8-byte aligned
    NmsBox* kept = reinterpret_cast<NmsBox*>(sup + NMS_ROUND * NMS_WORDS);
    NmsBox* cand = kept + p.max_out;
    int* flags = reinterpret_cast<int*>(cand + NMS_ROUND);
    int* keep = flags + NMS_ROUND + 2;

    const int b = blockIdx.x, t = threadIdx.x;
    const float* scores = p.scores + (size_t)b * p.n_total;
    const float4* deltas = p.deltas + (size_t)b * p.n_total;
    const int n_sel_full = min(p.nms_pre, p.n_total);
    RPN_STAMP(0);

    // ---- 1. exact k-th key by 8-bit radix select over the 64-bit composite key ------------
    // The scores are read from global memory ONCE into registers (<= 64 per thread); the eight
    // digit passes then touch only registers and LDS.  Histograms are private per wave (16
    // copies) so that the first pass, where most keys share the exponent byte, does not
    // serialise 1024 threads on one LDS word.
    // The scores (<= 256 KB, L2 resident) are re-read by every sweep, 8 independent loads per thread
    // at a time; caching all 64 per thread in registers spilled heavily (1024-thread workgroups
    // cap at 128 VGPRs).  h = high word of the composite key (= ~ordered(score)), the low word is
    // the anchor index i.  All digit passes are 32-bit: passes 7..4 walk h, passes 3..0 walk the
    // index among elements whose h equals the threshold.
#define RPN_SWEEP_N 16    /* independent loads in flight per thread and batch: 4 memory round trips per sweep of 63 000 */
#define RPN_SWEEP(BODY)                                                      \
    for (int j0 = 0; j0 < RPN_EPT; j0 += RPN_SWEEP_N) {                      \
        if (j0 * POST_THREADS >= p.n_total) break;                           \
        float sv[RPN_SWEEP_N];                                               \
        _Pragma("unroll") for (int jj = 0; jj < RPN_SWEEP_N; ++jj) {         \
            const int ii = (j0 + jj) * POST_THREADS + t;                     \
            sv[jj] = ii < p.n_total ? scores[ii] : 0.f;                      \
        }                                                                    \
        _Pragma("unroll") for (int jj = 0; jj < RPN_SWEEP_N; ++jj) {         \
            const uint32_t i = (uint32_t)((j0 + jj) * POST_THREADS + t);     \
            const bool in_range = i < (uint32_t)p.n_total;                   \
            const uint32_t h = ~f32_ordered(sv[jj]);                         \
            BODY                                                             \
        }                                                                    \
    }
    const int lane = t & 63, wv = t >> 6;
    RPN_STAMP(1);
    // Greedy NMS consumes candidates in score order and stops at max_out kept boxes, so the
    // result only depends on a prefix of the ranking.  Attempt 0 ranks just the best
    // RPN_FAST_SEL candidates (sort of 2048 instead of 8192 keys); if NMS cannot fill max_out from
    // them, attempt 1 redoes the stage with the full nms_pre - identical output either way.
    int n_keep = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool fast = attempt == 0 && n_sel_full > RPN_FAST_SEL;
        if (attempt == 1 && !(n_sel_full > RPN_FAST_SEL)) break;
        const int n_sel = fast ? RPN_FAST_SEL : n_sel_full;
        const int cap = fast ? RPN_FAST_CAP : p.cap;
        uint32_t kth_hi = 0xffffffffu, kth_lo = 0xffffffffu, kth_mask_hi = 0xffffffffu, kth_mask_lo = 0xffffffffu;
        if (n_sel < p.n_total) {
            uint32_t pre_hi = 0, msk_hi = 0, pre_lo = 0, msk_lo = 0;
            int k = n_sel;   // 1-based rank wanted
            for (int pass = 7; pass >= 0; --pass) {
                const int shift = (pass & 3) * 8;
                const bool hi_pass = pass >= 4;
                for (int i = t; i < POST_WAVES * 256; i += POST_THREADS) hist[i] = 0;
                __syncthreads();
                int* my_hist = hist + wv * 256;
                // run-length aggregation per thread: consecutive equal digits (the common case in
                // the exponent-byte pass) cost one LDS atomic per run instead of one per element
                uint32_t run_d = 0xffffffffu;
                int run_n = 0;
                RPN_SWEEP({
                    bool match;
                    uint32_t digit;
                    if (hi_pass) {
                        match = (h & msk_hi) == pre_hi;
                        digit = (h >> shift) & 0xffu;
                    } else {
                        match = (h == pre_hi) && ((i & msk_lo) == pre_lo);
                        digit = (i >> shift) & 0xffu;
                    }
                    if (match && in_range) {
                        if (digit == run_d) {
                            ++run_n;
                        } else {
                            if (run_n) atomicAdd(&my_hist[run_d], run_n);
                            run_d = digit;
                            run_n = 1;
                        }
                    }
                })
                if (run_n) atomicAdd(&my_hist[run_d], run_n);
                __syncthreads();
                // bin totals over the 16 wave copies, then an inclusive scan over the 256 bins
                int tot = 0;
                if (t < 256) {
    #pragma unroll
                    for (int w = 0; w < POST_WAVES; ++w) tot += hist[w * 256 + t];
                }
                int incl = tot;
    #pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int v = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += v;
                }
                if (t < 256 && lane == 63) misc[8 + wv] = incl;
                __syncthreads();
                if (t < 256) {
                    int base = 0;
                    for (int w = 0; w < wv; ++w) base += misc[8 + w];
                    incl += base;
                    const int excl = incl - tot;
                    if (excl < k && k <= incl) {   // exactly one bin holds rank k
                        misc[0] = t;
                        misc[1] = k - excl;
                        misc[5] = (n_sel - k) + incl;   // #keys whose known prefix is <= the chosen one
                    }
                }
                __syncthreads();
                const uint32_t d = (uint32_t)misc[0];
                if (hi_pass) {
                    pre_hi |= d << shift;
                    msk_hi |= 0xffu << shift;
                } else {
                    pre_lo |= d << shift;
                    msk_lo |= 0xffu << shift;
                }
                k = misc[1];
                // Early exit: once the keys with prefix <= chosen fit the sort buffer, select them all;
                // the sort puts the wanted n_sel first.  Typically after 2 of the 8 passes.
                if (misc[5] <= cap) break;
            }
            kth_hi = pre_hi;
            kth_lo = pre_lo;
            kth_mask_hi = msk_hi;
            kth_mask_lo = msk_lo;
        }
        // ---- 2. compaction of the selected keys into LDS (wave-aggregated), pad, sort ----------
        RPN_STAMP(2);
        for (int i = t; i < cap; i += POST_THREADS) keys[i] = ~0ull;
        {
            // each thread owns a contiguous output range found by one block scan: no atomics
            int mine = 0;
            RPN_SWEEP({
                const uint32_t hm = h & kth_mask_hi;
                mine += (in_range && (hm < kth_hi || (hm == kth_hi && (i & kth_mask_lo) <= kth_lo))) ? 1 : 0;
            })
            int incl = mine;
    #pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            if (lane == 63) misc[8 + wv] = incl;
            __syncthreads();          // also orders the ~0 fill above before the writes below
            int pos = incl - mine;
            for (int w = 0; w < wv; ++w) pos += misc[8 + w];
            RPN_SWEEP({
                const uint32_t hm = h & kth_mask_hi;
                if (in_range && (hm < kth_hi || (hm == kth_hi && (i & kth_mask_lo) <= kth_lo)))
                    keys[pos++] = ((uint64_t)h << 32) | i;
            })
        }
        __syncthreads();
        RPN_STAMP(3);
        block_bitonic_sort(keys, cap);
        RPN_STAMP(4);

        // ---- 3. decode (delta2bbox), min-size filter, order-preserving compaction --------------
        float4* out_boxes = p.sorted_boxes + (size_t)b * cap;
        float* out_scores = p.sorted_scores + (size_t)b * cap;
        const int per_thread = cap / POST_THREADS > 0 ? cap / POST_THREADS : 1;
        const int i0 = t * per_thread;
        float4 bx[8];
        float sc[8];
        int valid_bits = 0, cnt = 0;
    #pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + j;
            if (j >= per_thread || i >= n_sel || i >= cap) continue;
            const uint64_t key = keys[i];
            const uint32_t idx = key_index(key);
            if (p.dbg_topk_idx) p.dbg_topk_idx[(size_t)b * 8192 + i] = (int32_t)idx;
            const int a = idx % p.A;
            const int px = idx / p.A;
            const int gx = px % p.feat_w, gy = px / p.feat_w;
            const float4 ba = p.base_anchors[a];
            const float sx = (float)(gx * p.stride), sy = (float)(gy * p.stride);
            const float ax1 = ba.x + sx, ay1 = ba.y + sy, ax2 = ba.z + sx, ay2 = ba.w + sy;
            const float4 d = deltas[idx];
            const float dx = d.x * p.stdv[0] + p.mean[0];
            const float dy = d.y * p.stdv[1] + p.mean[1];
            float dw = d.z * p.stdv[2] + p.mean[2];
            float dh = d.w * p.stdv[3] + p.mean[3];
            const float pcx = (ax1 + ax2) * 0.5f, pcy = (ay1 + ay2) * 0.5f;
            const float pw = ax2 - ax1, ph = ay2 - ay1;
            const float dxw = pw * dx, dyh = ph * dy;
            dw = fminf(fmaxf(dw, -p.max_ratio), p.max_ratio);
            dh = fminf(fmaxf(dh, -p.max_ratio), p.max_ratio);
            const float gcx = pcx + dxw, gcy = pcy + dyh;
            const float gw = pw * exp32(dw), gh = ph * exp32(dh);
            const float hw = gw * 0.5f, hh = gh * 0.5f;
            float x1 = gcx - hw, y1 = gcy - hh, x2 = gcx + hw, y2 = gcy + hh;
            x1 = fminf(fmaxf(x1, 0.f), p.img_w); x2 = fminf(fmaxf(x2, 0.f), p.img_w);
            y1 = fminf(fmaxf(y1, 0.f), p.img_h); y2 = fminf(fmaxf(y2, 0.f), p.img_h);
            bx[j] = make_float4(x1, y1, x2, y2);
            sc[j] = key_score(key);
            bool ok = true;
            if (p.min_size >= 0.f) ok = ((x2 - x1) > p.min_size) && ((y2 - y1) > p.min_size);
            if (ok) {
                valid_bits |= 1 << j;
                ++cnt;
            }
        }
        // block exclusive scan of cnt
        int incl = cnt;
    #pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) misc[3 + wv] = incl;
        __syncthreads();
        int wave_off = 0, total_valid = 0;
        for (int w = 0; w < POST_WAVES; ++w) {
            const int s = misc[3 + w];
            if (w < wv) wave_off += s;
            total_valid += s;
        }
        int pos = wave_off + incl - cnt;
    #pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (valid_bits & (1 << j)) {
                out_boxes[pos] = bx[j];
                out_scores[pos] = sc[j];
                ++pos;
            }
        }
        __syncthreads();   // global writes of this workgroup are visible to it after the barrier

        RPN_STAMP(5);
        // ---- 4. greedy NMS, keep the first max_out -------------------------------------------------
        n_keep = nms_sorted_block(out_boxes, total_valid, p.iou_thr, p.max_out, keep, kept, cand, sup, flags);
        float* props = p.proposals + (size_t)b * p.max_out * 5;
        for (int i = t; i < p.max_out; i += POST_THREADS) {
            if (i < n_keep) {
                const int s = keep[i];
                const float4 v = out_boxes[s];
                props[i * 5 + 0] = v.x; props[i * 5 + 1] = v.y; props[i * 5 + 2] = v.z; props[i * 5 + 3] = v.w;
                props[i * 5 + 4] = out_scores[s];
            } else {
                props[i * 5 + 0] = 0.f; props[i * 5 + 1] = 0.f; props[i * 5 + 2] = 0.f; props[i * 5 + 3] = 0.f;
                props[i * 5 + 4] = 0.f;
            }
            if (p.rois) {
                float* r = p.rois + ((size_t)b * p.max_out + i) * 5;
                r[0] = (float)b; r[1] = props[i * 5 + 0]; r[2] = props[i * 5 + 1]; r[3] = props[i * 5 + 2]; r[4] = props[i * 5 + 3];
            }
        }

        __syncthreads();
        if (!fast || n_keep >= p.max_out) break;
    }
    if (t == 0) p.n_props[b] = n_keep;
    RPN_STAMP(6);
}

extern "C" size_t fgn_rpn_proposals_scratch_bytes(int batch, int n_total, int nms_pre) {
    int n_sel = nms_pre < n_total ? nms_pre : n_total;
    int cap = POST_THREADS;
    while (cap < n_sel) cap <<= 1;
    return (size_t)batch * cap * (sizeof(float4) + sizeof(float));
}

extern "C" int fgn_rpn_proposals_f32(const float* scores, const float* deltas, const float* base_anchors,
                                     void* scratch, float* proposals, float* rois_out, int32_t* n_props, int32_t* dbg_topk_idx,
                                     int batch, int feat_h, int feat_w, int n_anchors, int stride, float img_h,
                                     float img_w, const float* means4, const float* stds4, float max_ratio,
                                     int nms_pre, float min_bbox_size, float iou_thr, int max_per_img,
                                     hipStream_t stream) {
    if (!scores || !deltas || !base_anchors || !scratch || !proposals || !n_props || !means4 || !stds4)
        return FGN_ERR_ARG;
    ProposalParams p;
    p.n_total = feat_h * feat_w * n_anchors;
    if (p.n_total <= 0 || batch <= 0) return FGN_OK;
    const int n_sel = nms_pre > 0 && nms_pre < p.n_total ? nms_pre : p.n_total;
    int cap = POST_THREADS;
    while (cap < n_sel) cap <<= 1;
    if (cap > 8192 || max_per_img > 1024 || max_per_img < 1) return FGN_ERR_SHAPE;
    if (p.n_total > RPN_EPT * POST_THREADS) return FGN_ERR_SHAPE;   // scores are cached in registers
    p.scores = scores; p.deltas = reinterpret_cast<const float4*>(deltas);
    p.base_anchors = reinterpret_cast<const float4*>(base_anchors);
    p.sorted_boxes = reinterpret_cast<float4*>(scratch);
    p.sorted_scores = reinterpret_cast<float*>(p.sorted_boxes + (size_t)batch * cap);
    p.proposals = proposals; p.rois = rois_out; p.n_props = n_props; p.dbg_topk_idx = dbg_topk_idx;
    p.A = n_anchors; p.feat_w = feat_w; p.stride = stride;
    p.nms_pre = n_sel; p.cap = cap;
    p.img_h = img_h; p.img_w = img_w;
    for (int i = 0; i < 4; ++i) { p.mean[i] = means4[i]; p.stdv[i] = stds4[i]; }
    p.max_ratio = max_ratio; p.min_size = min_bbox_size; p.iou_thr = iou_thr; p.max_out = max_per_img;
    const size_t lds = (size_t)cap * 8 + POST_WAVES * 256 * 4 + 64 * 4 + (size_t)max_per_img * sizeof(NmsBox) +
                       NMS_ROUND * sizeof(NmsBox) + NMS_ROUND * NMS_WORDS * 8 + (NMS_ROUND + 2) * 4 + (size_t)max_per_img * 4;
    static unsigned long long lds_ok = 0ull;
    const hipError_t attr = fgn_allow_full_lds(reinterpret_cast<const void*>(rpn_proposals_kernel), &lds_ok);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL(rpn_proposals_kernel, dim3(batch), dim3(POST_THREADS), lds, stream, p);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
