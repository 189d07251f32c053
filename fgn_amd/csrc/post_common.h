// Device building blocks for the selection stages (proposal top-k / NMS / detection NMS).
// Everything here is on the bit-exact path: fp32 ops are kept in the oracle's order and the
// translation unit is compiled with -ffp-contract=off so no mul+add is fused.
#pragma once
#include "common.h"

constexpr int POST_THREADS = 1024;
constexpr int POST_WAVES = POST_THREADS / 64;

// monotone map float -> uint32 (larger float -> larger uint)
__device__ __forceinline__ uint32_t f32_ordered(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_f32(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
// ascending composite key  <=>  (score descending, index ascending): the oracle's stable
// descending sort (np.argsort(-scores, kind='stable')).
__device__ __forceinline__ uint64_t sort_key(float score, uint32_t idx) {
    return ((uint64_t)(~f32_ordered(score)) << 32) | idx;
}
__device__ __forceinline__ float key_score(uint64_t k) { return ordered_f32(~(uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t key_index(uint64_t k) { return (uint32_t)k; }

// correctly rounded fp32 sigmoid / exp (evaluated in fp64, rounded once) -- the oracle's
// exp32 / sigmoid32 convention (oracle/fgn_ref_cpu.py).
__device__ __forceinline__ float sigmoid32(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }
__device__ __forceinline__ float exp32(float x) { return (float)exp((double)x); }

// In-LDS bitonic sort of n_pow2 uint64 keys, ascending; all POST_THREADS threads call.
__device__ inline void block_bitonic_sort(uint64_t* keys, int n_pow2) {
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n_pow2; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        keys[i] = b;
                        keys[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// IoU test of mmcv's nms (offset 0): inter / (area_a + area_b - inter) > thr
struct NmsBox {
    float x1, y1, x2, y2, area;
};
__device__ __forceinline__ NmsBox make_nms_box(float x1, float y1, float x2, float y2) {
    NmsBox b;
    b.x1 = x1; b.y1 = y1; b.x2 = x2; b.y2 = y2;
    b.area = (x2 - x1) * (y2 - y1);
    return b;
}
__device__ __forceinline__ bool iou_gt(const NmsBox& a, const NmsBox& b, float thr) {
    const float xx1 = fmaxf(a.x1, b.x1), yy1 = fmaxf(a.y1, b.y1);
    const float xx2 = fminf(a.x2, b.x2), yy2 = fminf(a.y2, b.y2);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    const float ovr = inter / ((a.area + b.area) - inter);
    return ovr > thr;
}

// Greedy NMS over boxes already sorted by (score desc, index asc).
//   boxes   : global, n x float4 (x1,y1,x2,y2) in sorted order
//   keep    : LDS/global int array [max_out] receiving sorted positions of kept boxes
//   kept    : LDS scratch NmsBox[max_out]
//   chunk_* : LDS scratch [POST_THREADS]
// Exactly the sequential greedy result: a box is kept iff no previously kept box has
// IoU > thr with it; stops after max_out kept boxes.  Returns the number kept (uniform).
// Parallel structure: 1024 candidates per round are tested against the kept list by all
// 16 waves (phase A); wave 0 then resolves the round 64 boxes at a time with ballot /
// readlane only (phase B) -- no barriers inside the serial part.
__device__ inline int nms_sorted_block(const float4* __restrict__ boxes, int n, float thr, int max_out,
                                       int* keep, NmsBox* kept, NmsBox* chunk_box, int* chunk_alive,
                                       int* kept_cnt_sh) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) *kept_cnt_sh = 0;
    __syncthreads();
    int kept_cnt = 0;
    for (int base = 0; base < n && kept_cnt < max_out; base += POST_THREADS) {
        const int i = base + t;
        bool alive = i < n;
        NmsBox b = make_nms_box(0.f, 0.f, 0.f, 0.f);
        if (alive) {
            const float4 v = boxes[i];
            b = make_nms_box(v.x, v.y, v.z, v.w);
        }
        const int K0 = kept_cnt;
        for (int j = 0; j < K0 && alive; ++j)
            if (iou_gt(kept[j], b, thr)) alive = false;
        chunk_box[t] = b;
        chunk_alive[t] = alive ? 1 : 0;
        __syncthreads();
        if (wv == 0) {
            int kcur = K0;
            for (int c = 0; c < POST_WAVES && kcur < max_out; ++c) {
                const int ci = c * 64 + lane;
                if (base + c * 64 >= n) break;
                const NmsBox cb = chunk_box[ci];
                bool und = chunk_alive[ci] != 0;
                for (int j = K0; j < kcur && und; ++j)
                    if (iou_gt(kept[j], cb, thr)) und = false;
                while (kcur < max_out) {
                    const unsigned long long m = __ballot(und);
                    if (m == 0ull) break;
                    const int f = __ffsll((long long)m) - 1;
                    NmsBox fb;
                    fb.x1 = __shfl(cb.x1, f, 64); fb.y1 = __shfl(cb.y1, f, 64);
                    fb.x2 = __shfl(cb.x2, f, 64); fb.y2 = __shfl(cb.y2, f, 64);
                    fb.area = __shfl(cb.area, f, 64);
                    if (lane == f) {
                        kept[kcur] = cb;
                        keep[kcur] = base + ci;
                        und = false;
                    } else if (und && iou_gt(fb, cb, thr)) {
                        und = false;
                    }
                    ++kcur;
                }
            }
            if (lane == 0) *kept_cnt_sh = kcur;
        }
        __syncthreads();
        kept_cnt = *kept_cnt_sh;
    }
    return kept_cnt;
}
