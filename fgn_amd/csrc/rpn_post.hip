// AG-RPN output merge and proposal generation, all on device.
//   rpn_merge   : per-anchor arg-max over the N guided passes + sigmoid
//                 (AGRPNHead.forward_single, fgn_ag_rpn_head.py:81-113; sigmoid of
//                  mmdet RPNHead._get_bboxes_single, reached from fgn.py:229-235)
//   proposals   : top nms_pre by (score desc, index asc) -> delta2bbox -> w,h > min ->
//                 greedy NMS -> first max_per_img      (mmdet 2.18 RPNHead semantics)
// The reference runs this stage on CPU tensors (img_metas_cpu, fgn.py:209,233); here it
// never leaves the GPU and never synchronises with the host: counts stay in device memory.
// Compiled with -ffp-contract=off: fp32 op order is the oracle's.
//
// Launch sequence of fgn_rpn_proposals_f32 (inference sizes), all behind one C call:
//   rpn_hist<1>, rpn_hist<2>, rpn_compact   the best >= 1536, <= 2048 keys of the 63 000 (two histogram levels)
//   rpn_ranksort                            their order (rank = number of smaller keys) + the decoded box of each rank
//   rpn_iou_matrix                          suppression bits between the ranked boxes, on the whole chip
//   rpn_matrix_nms                          greedy resolution + outputs by ONE wavefront; marks the image finished
//   rpn_proposals                           returns at once for finished images; otherwise (ties / saturation made the
//                                           pre-selection invalid, or the ranked prefix ran dry) the whole stage in one
//                                           workgroup: radix select, bitonic sort, decode, round-of-128 NMS
// Identical outputs on every route (tests/test_hip_stages.py: bit-exact against the oracle, incl. the fallbacks).
#include "post_common.h"
#include <cstdlib>

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
// proposals, single-workgroup form (the fallback of the launch sequence above): one workgroup (1024 threads) per image.
// ----------------------------------------------------------------------------------------------
constexpr int RPN_EPT = 64;
constexpr int RPN_FAST_SEL = 1536;   // candidates ranked by the fast first attempt
constexpr int RPN_FAST_CAP = 2048;   // its sort buffer   // scores cached per thread: n_total <= 65536
constexpr int RPN_MT_WORDS = RPN_FAST_CAP / 64;   // u64 words of one row of the suppression matrix

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
    // multi-workgroup pre-selection (rpn_hist<1>, rpn_hist<2>, rpn_compact, rpn_ranksort kernels): the best
    // pre_info[b][3] <= RPN_FAST_CAP keys of image b, sorted, in pre_sorted[b]; pre_info[b][4] = 1 when valid
    const uint64_t* pre_sorted;   // [B][RPN_FAST_CAP] or null
    int32_t* pre_info;            // [B][8]: b1, count before b1, b2, candidate count, ok flag, [5] = finished by rpn_matrix_nms_kernel
    // ... and, from rpn_ranksort / rpn_iou_matrix, their decoded boxes and the suppression bits between them
    // (null: the proposal kernel decodes and tests the boxes itself)
    const float4* pre_boxes;      // [B][RPN_FAST_CAP] box of the candidate of each rank
    const int32_t* pre_valid;     // [B][RPN_FAST_CAP] 1 = passes the min-size test
    const unsigned long long* pre_mt;   // [B][RPN_FAST_CAP][RPN_MT_WORDS] bit j of row c: the EARLIER candidate j suppresses c
    int n_total, A, feat_w, stride;
    int nms_pre, cap;        // cap = pow2 >= min(nms_pre, n_total)
    float img_h, img_w;
    float mean[4], stdv[4];
    float max_ratio, min_size, iou_thr;
    int max_out;
};


// delta2bbox of anchor `idx` (mmdet 2.18 DeltaXYWHBBoxCoder.decode, fp32 op by op, clipped to the image) and the
// min-size test of RPNHead._bbox_post_process
__device__ __forceinline__ float4 rpn_decode_box(const ProposalParams& p, uint32_t idx, const float4* __restrict__ deltas,
                                                 bool* ok) {
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
    *ok = p.min_size >= 0.f ? ((x2 - x1) > p.min_size) && ((y2 - y1) > p.min_size) : true;
    return make_float4(x1, y1, x2, y2);
}

// proposals [max_out][5] (zero rows after the kept ones) and their bbox2roi form; whole workgroup
__device__ __forceinline__ void rpn_write_outputs(const ProposalParams& p, int b, int n_keep, const int* keep,
                                                  const float4* __restrict__ out_boxes,
                                                  const float* __restrict__ out_scores) {
    float* props = p.proposals + (size_t)b * p.max_out * 5;
    for (int i = threadIdx.x; i < p.max_out; i += POST_THREADS) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        float sc = 0.f;
        if (i < n_keep) {
            const int s = keep[i];
            v = out_boxes[s];
            sc = out_scores[s];
        }
        props[i * 5 + 0] = v.x; props[i * 5 + 1] = v.y; props[i * 5 + 2] = v.z; props[i * 5 + 3] = v.w;
        props[i * 5 + 4] = sc;
        if (p.rois) {
            float* r = p.rois + ((size_t)b * p.max_out + i) * 5;
            r[0] = (float)b; r[1] = v.x; r[2] = v.y; r[3] = v.z; r[4] = v.w;
        }
    }
}

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
    // rpn_matrix_nms_kernel ran in front on the ranked prefix: finished this image, or could not fill max_out from the
    // prefix - then attempt 0 here would fail the same way and the stage starts at the full nms_pre
    if (p.pre_mt && p.pre_info[b * 8 + 5] == 1) return;
    const int first_attempt = (p.pre_mt && p.pre_info[b * 8 + 4] == 1 && n_sel_full > RPN_FAST_SEL) ? 1 : 0;
    for (int attempt = first_attempt; attempt < 2; ++attempt) {
        const bool fast = attempt == 0 && n_sel_full > RPN_FAST_SEL;
        if (attempt == 1 && !(n_sel_full > RPN_FAST_SEL)) break;
        int n_sel = fast ? RPN_FAST_SEL : n_sel_full;
        const int cap = fast ? RPN_FAST_CAP : p.cap;
        // attempt 0 from the multi-workgroup pre-selection: the keys arrive ranked, steps 1-2 and the sort are skipped
        const bool pre = fast && p.pre_sorted && p.pre_info[b * 8 + 4] == 1;
        if (pre) n_sel = min(p.pre_info[b * 8 + 3], n_sel_full);
        uint32_t kth_hi = 0xffffffffu, kth_lo = 0xffffffffu, kth_mask_hi = 0xffffffffu, kth_mask_lo = 0xffffffffu;
        if (!pre && n_sel < p.n_total) {
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
        for (int i = t; i < cap; i += POST_THREADS)
            keys[i] = (pre && i < n_sel) ? p.pre_sorted[(size_t)b * RPN_FAST_CAP + i] : ~0ull;
        if (!pre) {
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
        if (!pre) block_bitonic_sort(keys, cap);
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
            bool ok = false;
            bx[j] = rpn_decode_box(p, idx, deltas, &ok);
            sc[j] = key_score(key);
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
        rpn_write_outputs(p, b, n_keep, keep, out_boxes, out_scores);

        __syncthreads();
        if (!fast || n_keep >= p.max_out) break;
    }
    if (t == 0) p.n_props[b] = n_keep;
    RPN_STAMP(6);
}


// The proposal stage of an image whose best candidates arrive ranked, decoded and IoU-tested from the multi-workgroup
// kernels in front (rpn_hist / rpn_compact / rpn_ranksort / rpn_iou_matrix): the greedy resolution and the outputs,
// nothing else - ONE WAVEFRONT, no LDS, no barrier.  (The single workgroup of rpn_proposals_kernel spent 60 of its
// 76 us on 330 000 IoU tests on one CU and on a walk that took 125 cycles per kept box.)
// Candidates are taken 64 at a time, one per lane, in rank order:
//   - lane c is dead when a box kept in an EARLIER chunk suppresses it: its row of `mt` AND the kept bitsets
//     (uniform: scalar registers; the chunk loop is fully unrolled so every index is static),
//   - inside the chunk the greedy order is resolved by passes over the whole wavefront: an undecided lane is dead
//     when a kept lane before it suppresses it, kept when every lane before it that overlaps it is decided dead; the
//     first undecided lane is always decided, so a pass per link of the longest suppression chain (2-4 in practice,
//     64 at worst) instead of a step per kept box,
//   - the kept lanes write their proposal rows at once: slot = boxes kept before.
// Rows, box and key of a chunk are loaded three chunks ahead (the matrix was written by other XCDs: 1-2 us away).
// Greedy NMS is a prefix computation, so cutting the last chunk at max_out kept boxes is exact.
// When the ranked prefix fills max_out (or is the whole nms_pre) the image is finished (pre_info[5] = 1) and
// rpn_proposals_kernel behind it returns at once; otherwise that kernel redoes the stage on the full nms_pre.
constexpr int MN_CHUNKS = RPN_FAST_CAP / 64;
constexpr int MN_AHEAD = 3;
struct MnChunk {
    ulonglong2 w[MN_CHUNKS / 2];   // words [0, chunk] of the candidate's row of mt
    float4 box;
    uint64_t key;
    int ok;
};
template <int K>
__device__ __forceinline__ void mn_fetch(MnChunk& o, const unsigned long long* __restrict__ mt, const float4* __restrict__ boxes,
                                         const uint64_t* __restrict__ keys, const int32_t* __restrict__ valid, int lane) {
    if (K >= MN_CHUNKS) return;
    const int c = K * 64 + lane;
    const ulonglong2* row = reinterpret_cast<const ulonglong2*>(mt + (size_t)c * RPN_MT_WORDS);
#pragma unroll
    for (int i = 0; i < MN_CHUNKS / 2; ++i)
        if (2 * i <= K) o.w[i] = row[i];
    o.box = boxes[c];
    o.key = keys[c];
    o.ok = valid[c];
}
__device__ __forceinline__ unsigned long long mn_word(const MnChunk& c, int w) { return (w & 1) ? c.w[w >> 1].y : c.w[w >> 1].x; }

struct MnState {
    unsigned long long kept[MN_CHUNKS];
    int kept_cnt;
    bool stop;
};
template <int K>
__device__ __forceinline__ void mn_chunk(MnState& st, const MnChunk& c, const ProposalParams& p, int b, int n, int lane) {
    if (K >= MN_CHUNKS || st.stop) return;
    if (K * 64 >= n || st.kept_cnt >= p.max_out) { st.stop = true; return; }
    unsigned long long hit = 0ull;
#pragma unroll
    for (int w = 0; w < K; ++w) hit |= mn_word(c, w) & st.kept[w];
    const bool alive = c.ok != 0 && K * 64 + lane < n && hit == 0ull;
    const unsigned long long E = mn_word(c, K);        // the lanes before this one that overlap it
    unsigned long long U = __ballot(alive), Kp = 0ull;
    while (U != 0ull) {
        const bool und = ((U >> lane) & 1ull) != 0ull;
        const bool dead = (E & Kp) != 0ull;
        const bool keep = !dead && (E & U) == 0ull;
        const unsigned long long nk = __ballot(und && keep), nd = __ballot(und && dead);
        Kp |= nk;
        U &= ~(nk | nd);
    }
    const int before = __popcll(Kp & ((1ull << lane) - 1ull));
    const bool mine = ((Kp >> lane) & 1ull) != 0ull && st.kept_cnt + before < p.max_out;
    Kp = __ballot(mine);
    if (mine) {
        const int i = st.kept_cnt + before;
        float* o = p.proposals + ((size_t)b * p.max_out + i) * 5;
        o[0] = c.box.x; o[1] = c.box.y; o[2] = c.box.z; o[3] = c.box.w;
        o[4] = key_score(c.key);
        if (p.rois) {
            float* r = p.rois + ((size_t)b * p.max_out + i) * 5;
            r[0] = (float)b; r[1] = c.box.x; r[2] = c.box.y; r[3] = c.box.z; r[4] = c.box.w;
        }
    }
    st.kept[K] = Kp;
    st.kept_cnt += __popcll(Kp);
}
// chunks K, K+1, K+2 on the three ring slots, then the next three (compile-time recursion: static indices everywhere)
template <int K>
__device__ __forceinline__ void mn_run(MnState& st, MnChunk& r0, MnChunk& r1, MnChunk& r2, const ProposalParams& p,
                                       const unsigned long long* mt, const float4* boxes, const uint64_t* keys,
                                       const int32_t* valid, int b, int n, int lane) {
    if constexpr (K < MN_CHUNKS) {
        mn_chunk<K>(st, r0, p, b, n, lane);
        if (!st.stop) mn_fetch<K + MN_AHEAD>(r0, mt, boxes, keys, valid, lane);
        mn_chunk<K + 1>(st, r1, p, b, n, lane);
        if (!st.stop) mn_fetch<K + 1 + MN_AHEAD>(r1, mt, boxes, keys, valid, lane);
        mn_chunk<K + 2>(st, r2, p, b, n, lane);
        if (!st.stop) mn_fetch<K + 2 + MN_AHEAD>(r2, mt, boxes, keys, valid, lane);
        if (!st.stop) mn_run<K + 3>(st, r0, r1, r2, p, mt, boxes, keys, valid, b, n, lane);
    }
}

__global__ __launch_bounds__(64) void rpn_matrix_nms_kernel(const ProposalParams p) {
    const int b = blockIdx.x, lane = threadIdx.x;
    RPN_STAMP(5);
    if (p.pre_info[b * 8 + 4] != 1) return;
    const int n = min(p.pre_info[b * 8 + 3], p.nms_pre);
    const unsigned long long* mt = p.pre_mt + (size_t)b * RPN_FAST_CAP * RPN_MT_WORDS;
    const float4* boxes = p.pre_boxes + (size_t)b * RPN_FAST_CAP;
    const uint64_t* keys = p.pre_sorted + (size_t)b * RPN_FAST_CAP;
    const int32_t* valid = p.pre_valid + (size_t)b * RPN_FAST_CAP;
    MnState st;
#pragma unroll
    for (int w = 0; w < MN_CHUNKS; ++w) st.kept[w] = 0ull;
    st.kept_cnt = 0;
    st.stop = false;
    MnChunk r0, r1, r2;
    mn_fetch<0>(r0, mt, boxes, keys, valid, lane);
    mn_fetch<1>(r1, mt, boxes, keys, valid, lane);
    mn_fetch<2>(r2, mt, boxes, keys, valid, lane);
    mn_run<0>(st, r0, r1, r2, p, mt, boxes, keys, valid, b, n, lane);
    const int n_keep = st.kept_cnt;
    if (n_keep < p.max_out && n < min(p.nms_pre, p.n_total)) return;   // the prefix ran dry: the full stage follows
    for (int i = n_keep + lane; i < p.max_out; i += 64) {             // zero rows after the kept ones
        float* o = p.proposals + ((size_t)b * p.max_out + i) * 5;
        o[0] = o[1] = o[2] = o[3] = o[4] = 0.f;
        if (p.rois) {
            float* r = p.rois + ((size_t)b * p.max_out + i) * 5;
            r[0] = (float)b; r[1] = r[2] = r[3] = r[4] = 0.f;
        }
    }
    if (lane == 0) {
        p.n_props[b] = n_keep;
        p.pre_info[b * 8 + 5] = 1;
        if (p.dbg_topk_idx) p.dbg_topk_idx[(size_t)b * 8192 + 8192 - 16 + 9] = 1;
    }
    RPN_STAMP(6);
}

// ----------------------------------------------------------------------------------------------
// Multi-workgroup pre-selection for attempt 0 of rpn_proposals_kernel (the single-workgroup radix select,
// compaction and bitonic sort of ~63 000 scores were 90 us of that kernel's 170 us).  Two histogram levels over the
// descending key h = ~ordered(score): bits [31:20], then bits [19:8] inside the level-1 threshold bin (sigmoid
// scores saturate: at cfg3 2 855 keys share the 16 high bits at rank 1536, 83 share the 22 high bits):
//   rpn_hist1    4096-bin histogram of h >> 20, privatised in LDS per workgroup, non-empty bins flushed with atomics
//   rpn_hist2    every workgroup scans level 1 (first bin b1 whose cumulative count reaches RPN_FAST_SEL, count
//                before it), then histograms (h >> 8) & 4095 of the keys in bin b1
//   rpn_compact  every workgroup scans level 2 (first bin b2 that reaches the target); candidates = ALL keys with
//                h >> 20 < b1, or == b1 and (h >> 8) & 4095 <= b2: a prefix of the ranking with >= RPN_FAST_SEL keys;
//                more than RPN_FAST_CAP (ties / saturation inside 24 equal bits) -> not valid, the proposal kernel
//                runs its own select; keys appended through a counter (order arbitrary)
//   rpn_ranksort rank of every candidate = number of smaller keys (keys are unique: the anchor index is the low
//                word) -> written at its rank: the exact sorted prefix the greedy NMS consumes
// Integer work only: the same set and order the in-kernel path produces (tests/test_hip_stages.py, bit-exact).
// ----------------------------------------------------------------------------------------------
constexpr int RPN_BINS = 4096;
constexpr int RPN_HIST_EPT = 1;                    // anchors per thread of the histogram / compaction kernels

// first bin whose inclusive cumulative count reaches `want` (-1: none) and the count before it; whole workgroup
__device__ inline void rpn_find_bin(const uint32_t* __restrict__ hist, int want, int base, int* wave_sums, int* out) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    constexpr int PER = RPN_BINS / POST_THREADS;   // 4 consecutive bins per thread
    const uint4 v = *reinterpret_cast<const uint4*>(hist + t * PER);
    const int c[4] = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    const int mine = c[0] + c[1] + c[2] + c[3];
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (t == 0) { out[0] = -1; out[1] = 0; }
    __syncthreads();
    if (lane == 63) wave_sums[wv] = incl;
    __syncthreads();
    int before = base + incl - mine;
    for (int w = 0; w < wv; ++w) before += wave_sums[w];
    if (before < want && want <= before + mine) {   // exactly one thread holds the crossing
        int cum = before;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (cum < want && want <= cum + c[j]) { out[0] = t * PER + j; out[1] = cum; }
            cum += c[j];
        }
    }
    __syncthreads();
}

template <int LEVEL>
__global__ __launch_bounds__(POST_THREADS) void rpn_hist_kernel(const float* __restrict__ scores, uint32_t* __restrict__ hist1,
                                                                uint32_t* __restrict__ hist2, int32_t* __restrict__ info,
                                                                int n_total, int want) {
    __shared__ uint32_t lh[RPN_BINS];
    __shared__ int wave_sums[POST_WAVES];
    __shared__ int found[2];
    const int b = blockIdx.y, t = threadIdx.x;
    for (int i = t; i < RPN_BINS; i += POST_THREADS) lh[i] = 0u;
    int b1 = 0;
    if (LEVEL == 2) {
        rpn_find_bin(hist1 + (size_t)b * RPN_BINS, want, 0, wave_sums, found);
        b1 = found[0];
        if (blockIdx.x == 0 && t == 0) { info[b * 8 + 0] = found[0]; info[b * 8 + 1] = found[1]; }
        if (b1 < 0) return;
    }
    __syncthreads();
    const int i0 = blockIdx.x * (POST_THREADS * RPN_HIST_EPT) + t;
#pragma unroll
    for (int j = 0; j < RPN_HIST_EPT; ++j) {
        const int i = i0 + j * POST_THREADS;
        if (i < n_total) {
            const uint32_t h = ~f32_ordered(scores[(size_t)b * n_total + i]);
            if (LEVEL == 1) atomicAdd(&lh[h >> 20], 1u);
            else if ((int)(h >> 20) == b1) atomicAdd(&lh[(h >> 8) & 4095u], 1u);
        }
    }
    __syncthreads();
    uint32_t* gh = (LEVEL == 1 ? hist1 : hist2) + (size_t)b * RPN_BINS;
    for (int i = t; i < RPN_BINS; i += POST_THREADS)
        if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

__global__ __launch_bounds__(POST_THREADS) void rpn_compact_kernel(const float* __restrict__ scores,
                                                                   const uint32_t* __restrict__ hist2,
                                                                   int32_t* __restrict__ info, int32_t* __restrict__ counter,
                                                                   uint64_t* __restrict__ cand, int n_total, int want, int cap) {
    __shared__ int wave_sums[POST_WAVES];
    __shared__ int found[2];
    const int b = blockIdx.y, t = threadIdx.x;
    const int b1 = info[b * 8 + 0], before1 = info[b * 8 + 1];
    if (b1 < 0) return;
    rpn_find_bin(hist2 + (size_t)b * RPN_BINS, want, before1, wave_sums, found);
    const int b2 = found[0];
    if (b2 < 0) return;
    // candidates = everything up to and including bin (b1, b2)
    const int n_cand = found[1] + (int)hist2[(size_t)b * RPN_BINS + b2];
    const bool ok = n_cand <= cap;
    if (blockIdx.x == 0 && t == 0) { info[b * 8 + 2] = b2; info[b * 8 + 3] = n_cand; info[b * 8 + 4] = ok ? 1 : 0; }
    if (!ok) return;
    const int i0 = blockIdx.x * (POST_THREADS * RPN_HIST_EPT) + t;
#pragma unroll
    for (int j = 0; j < RPN_HIST_EPT; ++j) {
        const int i = i0 + j * POST_THREADS;
        if (i < n_total) {
            const uint32_t h = ~f32_ordered(scores[(size_t)b * n_total + i]);
            const int k1 = (int)(h >> 20), k2 = (int)((h >> 8) & 4095u);
            if (k1 < b1 || (k1 == b1 && k2 <= b2)) {
                const int pos = atomicAdd(&counter[b], 1);
                if (pos < cap) cand[(size_t)b * cap + pos] = ((uint64_t)h << 32) | (uint32_t)i;
            }
        }
    }
}

// 8 threads per candidate, each counting the smaller keys in its eighth of the list (n^2 = 2.5 M comparisons spread
// over 64 workgroups; one thread per candidate took 68 us)
constexpr int RANK_SPLIT = 8;
// The thread that learns a candidate's rank also decodes its box (delta2bbox + min-size test) to that rank:
// boxes[rank], valid[rank] feed rpn_iou_matrix_kernel and the proposal kernel (null: keys only).
__global__ __launch_bounds__(256) void rpn_ranksort_kernel(const uint64_t* __restrict__ cand, const int32_t* __restrict__ info,
                                                           uint64_t* __restrict__ sorted, int cap, const ProposalParams p,
                                                           float4* __restrict__ boxes, int32_t* __restrict__ valid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(lds_raw);
    const int b = blockIdx.y;
    if (info[b * 8 + 4] != 1) return;
    const int n = info[b * 8 + 3];
    const int n4 = (n + 3) & ~3;
    for (int j = threadIdx.x; j < n4; j += blockDim.x) keys[j] = j < n ? cand[(size_t)b * cap + j] : ~0ull;
    __syncthreads();
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) / RANK_SPLIT;      // candidate
    const int part = threadIdx.x & (RANK_SPLIT - 1);
    const uint64_t k = i < n ? keys[i] : 0ull;
    int rank = 0;
    // part p scans the 4-key groups p, p + 8, ...: the 8 lanes of a candidate read 8 different groups, the 8
    // candidates of a wave read the same ones (broadcast)
    for (int j = part * 4; j < n4; j += RANK_SPLIT * 4) {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(keys + j);
        const ulonglong2 c = *reinterpret_cast<const ulonglong2*>(keys + j + 2);
        rank += (a.x < k) + (a.y < k) + (c.x < k) + (c.y < k);
    }
    rank += __shfl_xor(rank, 1, 64);
    rank += __shfl_xor(rank, 2, 64);
    rank += __shfl_xor(rank, 4, 64);
    if (part == 0 && i < n) {
        sorted[(size_t)b * cap + rank] = k;
        if (boxes) {
            bool ok = false;
            boxes[(size_t)b * cap + rank] = rpn_decode_box(p, key_index(k), p.deltas + (size_t)b * p.n_total, &ok);
            valid[(size_t)b * cap + rank] = ok ? 1 : 0;
        }
    }
}

// Pairwise suppression bits of the ranked candidates of an image: thread (c, w) tests candidate c against the 64
// candidates j of word w and sets bit j of mt[c][w] when the EARLIER candidate j < c suppresses c (both pass the
// min-size test).  Grid (word, row block of 64, image); blocks above the diagonal leave at once.
// iou_gt(earlier, later) as in nms_sorted_block.
constexpr int IOU_PARTS = 4;      // threads per (candidate, word): 16 tests each (one wave per block took 19 us in the episode)
__global__ __launch_bounds__(64 * IOU_PARTS) void rpn_iou_matrix_kernel(const int32_t* __restrict__ info,
                                                                        const float4* __restrict__ boxes,
                                                                        const int32_t* __restrict__ valid,
                                                                        unsigned long long* __restrict__ mt, int cap,
                                                                        int nms_pre, float iou_thr) {
    __shared__ NmsBox col[64];
    __shared__ int col_ok[64];
    __shared__ unsigned long long part_bits[IOU_PARTS][64];
    const int w = blockIdx.x, rb = blockIdx.y, b = blockIdx.z, lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    if (info[b * 8 + 4] != 1) return;
    const int n = min(info[b * 8 + 3], nms_pre);
    if (rb * 64 >= n || w > rb) return;
    const size_t o = (size_t)b * cap;
    if (part == 0) {
        const int j = w * 64 + lane;
        const float4 v = j < n ? boxes[o + j] : make_float4(0.f, 0.f, 0.f, 0.f);
        col[lane] = make_nms_box(v.x, v.y, v.z, v.w);
        col_ok[lane] = j < n && valid[o + j] != 0;
    }
    __syncthreads();
    const int c = min(rb * 64 + lane, n - 1);
    const float4 v = boxes[o + c];
    const NmsBox cb = make_nms_box(v.x, v.y, v.z, v.w);
    unsigned long long before = 0ull;
    if (valid[o + c] != 0) {
        constexpr int PER = 64 / IOU_PARTS;
#pragma unroll 4
        for (int k = 0; k < PER; ++k) {
            const int jj = part * PER + k, j = w * 64 + jj;
            if (j < c && col_ok[jj] && iou_gt(col[jj], cb, iou_thr)) before |= 1ull << jj;
        }
    }
    part_bits[part][lane] = before;
    __syncthreads();
    if (part == 0 && rb * 64 + lane < n) {
#pragma unroll
        for (int q = 1; q < IOU_PARTS; ++q) before |= part_bits[q][lane];
        mt[(o + c) * RPN_MT_WORDS + w] = before;
    }
}

extern "C" size_t fgn_rpn_proposals_scratch_bytes(int batch, int n_total, int nms_pre) {
    int n_sel = nms_pre < n_total ? nms_pre : n_total;
    int cap = POST_THREADS;
    while (cap < n_sel) cap <<= 1;
    const size_t base = ((size_t)batch * cap * (sizeof(float4) + sizeof(float)) + 255) / 256 * 256;
    // pre-selection: candidate keys + sorted keys (the zeroed histograms / counters come in `pre_zeroed`), then the
    // boxes, min-size flags and suppression bits of the ranked candidates
    return base + (size_t)batch * RPN_FAST_CAP * (2 * 8 + sizeof(float4) + 4 + RPN_MT_WORDS * 8);
}

// zero-initialised workspace of the pre-selection per call: two 4096-bin histograms + info + counter per image
extern "C" size_t fgn_rpn_proposals_zeroed_bytes(int batch) { return (size_t)batch * (2 * RPN_BINS * 4 + 64); }

extern "C" int fgn_rpn_proposals_f32(const float* scores, const float* deltas, const float* base_anchors,
                                     void* scratch, void* pre_zeroed, float* proposals, float* rois_out, int32_t* n_props,
                                     int32_t* dbg_topk_idx,
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
    p.pre_sorted = nullptr; p.pre_info = nullptr;
    p.pre_boxes = nullptr; p.pre_valid = nullptr; p.pre_mt = nullptr;
    constexpr int use_matrix = 1;    // (0: IoU tests inside the proposal kernel - round 3's A/B)
    constexpr int multi = 1;         // (0: single-workgroup path only)
    if (multi && pre_zeroed && n_sel > RPN_FAST_SEL && p.n_total > RPN_FAST_CAP) {
        unsigned char* z = reinterpret_cast<unsigned char*>(pre_zeroed);
        uint32_t* hist1 = reinterpret_cast<uint32_t*>(z);
        uint32_t* hist2 = hist1 + (size_t)batch * RPN_BINS;
        int32_t* info = reinterpret_cast<int32_t*>(hist2 + (size_t)batch * RPN_BINS);
        int32_t* counter = info + batch * 8;
        unsigned char* base = reinterpret_cast<unsigned char*>(scratch) +
                              ((size_t)batch * cap * (sizeof(float4) + sizeof(float)) + 255) / 256 * 256;
        uint64_t* candk = reinterpret_cast<uint64_t*>(base);
        uint64_t* sortedk = candk + (size_t)batch * RPN_FAST_CAP;
        const dim3 ga(cdiv(p.n_total, POST_THREADS * RPN_HIST_EPT), batch);
        hipLaunchKernelGGL(rpn_hist_kernel<1>, ga, dim3(POST_THREADS), 0, stream, scores, hist1, hist2, info, p.n_total,
                           RPN_FAST_SEL);
        hipLaunchKernelGGL(rpn_hist_kernel<2>, ga, dim3(POST_THREADS), 0, stream, scores, hist1, hist2, info, p.n_total,
                           RPN_FAST_SEL);
        hipLaunchKernelGGL(rpn_compact_kernel, ga, dim3(POST_THREADS), 0, stream, scores, hist2, info, counter, candk,
                           p.n_total, RPN_FAST_SEL, RPN_FAST_CAP);
        float4* boxes = reinterpret_cast<float4*>(sortedk + (size_t)batch * RPN_FAST_CAP);
        unsigned long long* mt = reinterpret_cast<unsigned long long*>(boxes + (size_t)batch * RPN_FAST_CAP);
        int32_t* valid = reinterpret_cast<int32_t*>(mt + (size_t)batch * RPN_FAST_CAP * RPN_MT_WORDS);
        if (!use_matrix) boxes = nullptr;
        hipLaunchKernelGGL(rpn_ranksort_kernel, dim3(RPN_FAST_CAP * RANK_SPLIT / 256, batch), dim3(256), RPN_FAST_CAP * 8, stream,
                           candk, info, sortedk, RPN_FAST_CAP, p, boxes, valid);
        if (boxes) {
            hipLaunchKernelGGL(rpn_iou_matrix_kernel, dim3(RPN_MT_WORDS, RPN_FAST_CAP / 64, batch), dim3(64 * IOU_PARTS), 0, stream, info,
                               boxes, valid, mt, RPN_FAST_CAP, n_sel, iou_thr);
            p.pre_boxes = boxes; p.pre_valid = valid; p.pre_mt = mt;
        }
        FGN_LAUNCH_CHECK();
        p.pre_sorted = sortedk; p.pre_info = info;
        if (boxes) {
            hipLaunchKernelGGL(rpn_matrix_nms_kernel, dim3(batch), dim3(64), 0, stream, p);
            FGN_LAUNCH_CHECK();
        }
    }
    hipLaunchKernelGGL(rpn_proposals_kernel, dim3(batch), dim3(POST_THREADS), lds, stream, p);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ----------------------------------------------------------------------------------------------
// Proposal stage at TRAINING sizes (fgn.py:161-167 with train_cfg.rpn_proposal: nms_pre 12000, 2000 kept,
// fgn_r50_c4_densecl.py:153-157): the candidate list no longer fits the LDS of one workgroup, so the ranking is a
// global bitonic sort of all composite keys over many workgroups, and one workgroup per image then decodes the
// first nms_pre of them chunk by chunk and runs the same greedy NMS with its kept list (<= 4096 boxes) in LDS.
// Same keys, same decode arithmetic, same NMS routine as rpn_proposals_kernel: identical selections.
//   rpn_keys_kernel        keys[b][i] = (~ordered(score) << 32) | i, padded with ~0 to n_pow2
//   rpn_sort_local_kernel  FIRST: sorts each chunk of SORT_CH keys (all stages k <= SORT_CH);
//                          otherwise finishes stage k with the strides j < SORT_CH in LDS
//   rpn_sort_global_kernel one compare-exchange pass of stride j >= SORT_CH
//   rpn_sorted_nms_kernel  decode + min-size filter + order-preserving compaction + greedy NMS + outputs
// ----------------------------------------------------------------------------------------------
constexpr int SORT_CH = 4096;

__global__ __launch_bounds__(256) void rpn_keys_kernel(const float* __restrict__ scores, uint64_t* __restrict__ keys,
                                                       int n_total, int n_pow2) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pow2) return;
    keys[(size_t)b * n_pow2 + i] = i < n_total ? sort_key(scores[(size_t)b * n_total + i], (uint32_t)i) : ~0ull;
}

template <bool FIRST>
__global__ __launch_bounds__(POST_THREADS) void rpn_sort_local_kernel(uint64_t* __restrict__ keys, int n_pow2, int ch,
                                                                     int k_stage) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint64_t* s = reinterpret_cast<uint64_t*>(lds_raw);
    const int t = threadIdx.x;
    const size_t base = (size_t)blockIdx.y * n_pow2 + (size_t)blockIdx.x * ch;
    const int g0 = blockIdx.x * ch;                       // global position of s[0]: decides the direction bits
    for (int i = t; i < ch; i += POST_THREADS) s[i] = keys[base + i];
    __syncthreads();
    const int half = ch >> 1;
    for (int k = FIRST ? 2 : k_stage; k <= (FIRST ? ch : k_stage); k <<= 1) {
        for (int j = min(k >> 1, half); j >= 1; j >>= 1) {
            for (int c = t; c < half; c += POST_THREADS) {
                const int i = ((c / j) * (j << 1)) + (c % j);
                const uint64_t a = s[i], bb = s[i + j];
                const bool up = ((g0 + i) & k) == 0;
                if ((a > bb) == up) { s[i] = bb; s[i + j] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = t; i < ch; i += POST_THREADS) keys[base + i] = s[i];
}

__global__ __launch_bounds__(256) void rpn_sort_global_kernel(uint64_t* __restrict__ keys, int n_pow2, int j, int k) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (n_pow2 >> 1)) return;
    uint64_t* kb = keys + (size_t)blockIdx.y * n_pow2;
    const int i = ((c / j) * (j << 1)) + (c % j);
    const uint64_t a = kb[i], b = kb[i + j];
    const bool up = (i & k) == 0;
    if ((a > b) == up) { kb[i] = b; kb[i + j] = a; }
}

__global__ __launch_bounds__(POST_THREADS) void rpn_sorted_nms_kernel(const ProposalParams p, const uint64_t* __restrict__ sorted,
                                                                     int n_pow2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    int* misc = reinterpret_cast<int*>(lds_raw);                                   // [64]: wave sums, running offset
    unsigned long long* sup = reinterpret_cast<unsigned long long*>(misc + 64);
    NmsBox* kept = reinterpret_cast<NmsBox*>(sup + NMS_ROUND * NMS_WORDS);
    NmsBox* cand = kept + p.max_out;
    int* flags = reinterpret_cast<int*>(cand + NMS_ROUND);
    int* keep = flags + NMS_ROUND + 2;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const float4* deltas = p.deltas + (size_t)b * p.n_total;
    const uint64_t* keys = sorted + (size_t)b * n_pow2;
    const int n_sel = min(p.nms_pre, p.n_total);
    float4* out_boxes = p.sorted_boxes + (size_t)b * p.cap;
    float* out_scores = p.sorted_scores + (size_t)b * p.cap;
    int total_valid = 0;
    for (int c0 = 0; c0 < n_sel; c0 += POST_THREADS) {
        const int i = c0 + t;
        bool ok = false;
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        float sc = 0.f;
        if (i < n_sel) {
            const uint64_t key = keys[i];
            bx = rpn_decode_box(p, key_index(key), deltas, &ok);
            sc = key_score(key);
        }
        const unsigned long long m = __ballot(ok);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) misc[wv] = __popcll(m);
        __syncthreads();
        int off = total_valid, chunk = 0;
        for (int w = 0; w < POST_WAVES; ++w) {
            const int sN = misc[w];
            if (w < wv) off += sN;
            chunk += sN;
        }
        if (ok) {
            out_boxes[off + before] = bx;
            out_scores[off + before] = sc;
        }
        total_valid += chunk;
        __syncthreads();
    }
    const int n_keep = nms_sorted_block(out_boxes, total_valid, p.iou_thr, p.max_out, keep, kept, cand, sup, flags);
    rpn_write_outputs(p, b, n_keep, keep, out_boxes, out_scores);
    if (t == 0) p.n_props[b] = n_keep;
}

extern "C" size_t fgn_rpn_proposals_large_scratch_bytes(int batch, int n_total, int nms_pre) {
    int n_pow2 = 2048;
    while (n_pow2 < n_total) n_pow2 <<= 1;
    const int n_sel = nms_pre > 0 && nms_pre < n_total ? nms_pre : n_total;
    return (size_t)batch * ((size_t)n_pow2 * 8 + (size_t)n_sel * (sizeof(float4) + sizeof(float))) + 256;
}

extern "C" int fgn_rpn_proposals_large_f32(const float* scores, const float* deltas, const float* base_anchors,
                                           void* scratch, float* proposals, float* rois_out, int32_t* n_props,
                                           int batch, int feat_h, int feat_w, int n_anchors, int stride, float img_h,
                                           float img_w, const float* means4, const float* stds4, float max_ratio,
                                           int nms_pre, float min_bbox_size, float iou_thr, int max_per_img,
                                           hipStream_t stream) {
    if (!scores || !deltas || !base_anchors || !scratch || !proposals || !n_props || !means4 || !stds4)
        return FGN_ERR_ARG;
    ProposalParams p;
    p.n_total = feat_h * feat_w * n_anchors;
    if (p.n_total <= 0 || batch <= 0) return FGN_OK;
    if (max_per_img < 1 || max_per_img > 4096) return FGN_ERR_SHAPE;     // kept list lives in LDS
    const int n_sel = nms_pre > 0 && nms_pre < p.n_total ? nms_pre : p.n_total;
    int n_pow2 = 2048;
    while (n_pow2 < p.n_total) n_pow2 <<= 1;
    const int ch = n_pow2 < SORT_CH ? n_pow2 : SORT_CH;
    uint64_t* keys = reinterpret_cast<uint64_t*>(scratch);
    p.scores = scores; p.deltas = reinterpret_cast<const float4*>(deltas);
    p.base_anchors = reinterpret_cast<const float4*>(base_anchors);
    p.sorted_boxes = reinterpret_cast<float4*>(keys + (size_t)batch * n_pow2);
    p.sorted_scores = reinterpret_cast<float*>(p.sorted_boxes + (size_t)batch * n_sel);
    p.proposals = proposals; p.rois = rois_out; p.n_props = n_props; p.dbg_topk_idx = nullptr;
    p.pre_sorted = nullptr; p.pre_info = nullptr;
    p.A = n_anchors; p.feat_w = feat_w; p.stride = stride;
    p.nms_pre = n_sel; p.cap = n_sel;
    p.img_h = img_h; p.img_w = img_w;
    for (int i = 0; i < 4; ++i) { p.mean[i] = means4[i]; p.stdv[i] = stds4[i]; }
    p.max_ratio = max_ratio; p.min_size = min_bbox_size; p.iou_thr = iou_thr; p.max_out = max_per_img;
    hipLaunchKernelGGL(rpn_keys_kernel, dim3(cdiv(n_pow2, 256), batch), dim3(256), 0, stream, scores, keys, p.n_total,
                       n_pow2);
    const dim3 gl(n_pow2 / ch, batch);
    hipLaunchKernelGGL(rpn_sort_local_kernel<true>, gl, dim3(POST_THREADS), (size_t)ch * 8, stream, keys, n_pow2, ch, 0);
    for (int k = ch << 1; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j >= ch; j >>= 1)
            hipLaunchKernelGGL(rpn_sort_global_kernel, dim3(cdiv(n_pow2 >> 1, 256), batch), dim3(256), 0, stream, keys,
                               n_pow2, j, k);
        hipLaunchKernelGGL(rpn_sort_local_kernel<false>, gl, dim3(POST_THREADS), (size_t)ch * 8, stream, keys, n_pow2,
                           ch, k);
    }
    FGN_LAUNCH_CHECK();
    const size_t lds = 64 * 4 + NMS_ROUND * NMS_WORDS * 8 + (size_t)max_per_img * sizeof(NmsBox) +
                       NMS_ROUND * sizeof(NmsBox) + (NMS_ROUND + 2) * 4 + (size_t)max_per_img * 4;
    static unsigned long long lds_ok = 0ull;
    const hipError_t attr = fgn_allow_full_lds(reinterpret_cast<const void*>(rpn_sorted_nms_kernel), &lds_ok);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL(rpn_sorted_nms_kernel, dim3(batch), dim3(POST_THREADS), lds, stream, p, keys, n_pow2);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
