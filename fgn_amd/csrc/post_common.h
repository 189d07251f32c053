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

// Workgroup barrier that orders LDS traffic only: global loads issued earlier (prefetches into registers) stay in
// flight across it.  (__syncthreads() drains them: s_waitcnt vmcnt(0) in front of the s_barrier.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Bitonic sort of POST_THREADS uint64 keys, ONE PER THREAD, ascending by thread index; returns this thread's key of
// the sorted sequence.  Strides below 64 exchange through the wavefront (no LDS storage, no barrier: 45 of the 55
// stages), strides of 64 and more through `xch` (LDS, 2 x POST_THREADS keys, double-buffered: one barrier per stage).
// 4-5 us for 1024 keys against 26 us for the all-LDS network below.
__device__ inline uint64_t block_bitonic_sort_regs(uint64_t key, uint64_t* xch) {
    const int i = threadIdx.x;
    int buf = 0;
    for (int k = 2; k <= POST_THREADS; k <<= 1) {
        const bool up = (i & k) == 0;
        for (int j = k >> 1; j >= 1; j >>= 1) {
            uint64_t other;
            if (j >= 64) {
                uint64_t* x = xch + buf * POST_THREADS;
                x[i] = key;
                lds_barrier();
                other = x[i ^ j];
                buf ^= 1;
            } else {
                const unsigned lo = __shfl_xor((unsigned)key, j, 64), hi = __shfl_xor((unsigned)(key >> 32), j, 64);
                other = ((uint64_t)hi << 32) | lo;
            }
            const bool lower = (i & j) == 0;
            const uint64_t mn = key < other ? key : other, mx = key < other ? other : key;
            key = (lower == up) ? mn : mx;
        }
    }
    return key;
}

// In-LDS bitonic sort of n_pow2 (>= 1024, power of two) uint64 keys, ascending; all
// POST_THREADS threads call.  Each wave owns a contiguous segment of n_pow2/16 keys: every
// compare-exchange with stride j < segment stays inside one wave's segment, and a wave's LDS
// operations execute in order, so those stages need no workgroup barrier - only the strides
// that cross segments do (10 of the 91 stages at n = 8192).
__device__ __forceinline__ void bitonic_cmpx(uint64_t* keys, int i, int j, int k) {
    const uint64_t a = keys[i], b = keys[i + j];
    const bool up = (i & k) == 0;
    if ((a > b) == up) {
        keys[i] = b;
        keys[i + j] = a;
    }
}
__device__ inline void block_bitonic_sort(uint64_t* keys, int n_pow2) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int seg = n_pow2 / POST_WAVES;        // keys per wave segment (>= 64)
    const int seg_half = seg >> 1;
    const int half = n_pow2 >> 1;
    bool need_block_sync = false;
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int lj = 31 - __builtin_clz(k >> 1); lj >= 0; --lj) {
            const int j = 1 << lj;
            if (j >= seg) {
                // cross-segment stride: comparators spread over the whole workgroup
                __syncthreads();
                for (int c = t; c < half; c += POST_THREADS)
                    bitonic_cmpx(keys, ((c >> lj) << (lj + 1)) + (c & (j - 1)), j, k);
                need_block_sync = true;
            } else {
                if (need_block_sync) {
                    __syncthreads();
                    need_block_sync = false;
                }
                // wave-local stride: this wave's seg/2 comparators, in-order LDS within the wave.
                // All reads of a stage are issued before any write (min/max stores, no
                // divergence), so LDS latency is paid once per stage, not once per comparator.
                const int base = wv * seg;
                uint64_t lo[4], hi[4];
                int idx[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = lane + 64 * u;
                    idx[u] = base + ((c >> lj) << (lj + 1)) + (c & (j - 1));
                    if (c < seg_half) {
                        lo[u] = keys[idx[u]];
                        hi[u] = keys[idx[u] + j];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = lane + 64 * u;
                    if (c < seg_half) {
                        const bool up = (idx[u] & k) == 0;
                        const uint64_t mn = lo[u] < hi[u] ? lo[u] : hi[u];
                        const uint64_t mx = lo[u] < hi[u] ? hi[u] : lo[u];
                        keys[idx[u]] = up ? mn : mx;
                        keys[idx[u] + j] = up ? mx : mn;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    __syncthreads();
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
// Exactly `inter / ((area_a + area_b) - inter) > thr` (the oracle's fp32 expression), but the
// IEEE division is only executed when a multiplication test cannot decide with a 4e-6 relative
// margin (the quotient and the products are each within 1 ulp): the division is ~10x the cost
// of everything else in the test and sits on the serial chain of the greedy loop.
__device__ __forceinline__ bool iou_gt(const NmsBox& a, const NmsBox& b, float thr) {
    const float xx1 = fmaxf(a.x1, b.x1), yy1 = fmaxf(a.y1, b.y1);
    const float xx2 = fminf(a.x2, b.x2), yy2 = fminf(a.y2, b.y2);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    const float uni = (a.area + b.area) - inter;
    const float tu = thr * uni;
    if (inter > tu * 1.000004f && uni > 0.f && tu < 3.0e38f) return true;
    if (inter < tu * 0.999996f && uni > 0.f) return false;
    const float ovr = inter / uni;
    return ovr > thr;
}

// Greedy NMS over boxes already sorted by (score desc, index asc).
//   boxes : global, n x float4 (x1,y1,x2,y2) in sorted order
//   keep  : LDS int [max_out]  sorted positions of the kept boxes (output)
//   kept  : LDS NmsBox [max_out], cand : LDS NmsBox [NMS_ROUND], sup : LDS u64 [NMS_ROUND*NMS_WORDS],
//   flags : LDS int [NMS_ROUND + 1]  (alive flags, last = kept counter)
// Exactly the sequential greedy result: a box is kept iff no previously kept box has IoU > thr
// with it; stops after max_out kept boxes.  Returns the number kept (uniform).
// Rounds of NMS_ROUND = 128 candidates (fewer IoU tests in total than 256-wide rounds):
//   A. all 1024 threads test the candidates against the boxes kept in earlier rounds
//      (8 threads per candidate split the kept list),
//   M. all threads build the 128x128 suppression bit matrix of the round (16 tests each),
//   S. wave 0 walks the round in score order using only scalar bit operations and
//      v_readlane on register-held matrix rows: ~50 cycles per kept box, no LDS, no barrier.
constexpr int NMS_ROUND = 128;                       // candidates per round
constexpr int NMS_WORDS = NMS_ROUND / 64;            // u64 words per suppression row
constexpr int NMS_PARTS = POST_THREADS / NMS_ROUND;  // threads per candidate
constexpr int NMS_PART_BITS = NMS_ROUND / NMS_PARTS; // matrix bits each thread computes (16)
static_assert(NMS_PART_BITS == 16, "suppression rows are stored as 16-bit pieces");

__device__ inline int nms_sorted_block(const float4* __restrict__ boxes, int n, float thr, int max_out,
                                       int* keep, NmsBox* kept, NmsBox* cand, unsigned long long* sup,
                                       int* flags) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int* kept_cnt_sh = flags + NMS_ROUND;
    unsigned short* sup16 = reinterpret_cast<unsigned short*>(sup);   // [NMS_ROUND][NMS_PARTS] == rows of NMS_WORDS u64
    if (t == 0) *kept_cnt_sh = 0;
    __syncthreads();
    int kept_cnt = 0;
    for (int base = 0; base < n && kept_cnt < max_out; base += NMS_ROUND) {
        const int nb = min(NMS_ROUND, n - base);
        if (t < NMS_ROUND) {
            NmsBox b = make_nms_box(0.f, 0.f, 0.f, 0.f);
            if (t < nb) {
                const float4 v = boxes[base + t];
                b = make_nms_box(v.x, v.y, v.z, v.w);
            }
            cand[t] = b;
            flags[t] = t < nb ? 1 : 0;
        }
        __syncthreads();
        const int c = t & (NMS_ROUND - 1), part = t / NMS_ROUND;   // NMS_PARTS threads per candidate
        const NmsBox cb = cand[c];
        // ---- A: against boxes kept in earlier rounds
        {
            bool dead = false;
            for (int j = part; j < kept_cnt; j += NMS_PARTS)
                if (iou_gt(kept[j], cb, thr)) dead = true;
            if (dead) flags[c] = 0;    // benign race: every writer stores 0
        }
        // ---- M: bit c2 of row c  <=>  box c suppresses the later box c2 of this round
        {
            unsigned bits = 0u;
            const int c2_0 = part * NMS_PART_BITS;
            if (c2_0 + NMS_PART_BITS - 1 > c && c < nb) {
#pragma unroll 4
                for (int b2 = 0; b2 < NMS_PART_BITS; ++b2) {
                    const int c2 = c2_0 + b2;
                    if (c2 > c && c2 < nb && iou_gt(cb, cand[c2], thr)) bits |= 1u << b2;
                }
            }
            sup16[c * NMS_PARTS + part] = (unsigned short)bits;
        }
        __syncthreads();
        // ---- S: serial resolution by wave 0
        if (wv == 0) {
            int kcur = kept_cnt;
            unsigned long long removed[NMS_WORDS];
#pragma unroll
            for (int q = 0; q < NMS_WORDS; ++q) removed[q] = ~__ballot(flags[q * 64 + lane] != 0);
#pragma unroll
            for (int q = 0; q < NMS_WORDS; ++q) {
                if (q * 64 >= nb || kcur >= max_out) break;
                // this chunk's matrix rows, one candidate per lane, in registers
                unsigned rl[NMS_WORDS], rh[NMS_WORDS];
#pragma unroll
                for (int w = 0; w < NMS_WORDS; ++w) {
                    const unsigned long long r = sup[(q * 64 + lane) * NMS_WORDS + w];
                    rl[w] = (unsigned)r;
                    rh[w] = (unsigned)(r >> 32);
                }
                unsigned long long avail = ~removed[q];
                while (avail != 0ull && kcur < max_out) {
                    const int f = __builtin_amdgcn_readfirstlane(__ffsll((long long)avail) - 1);
                    if (lane == 0) keep[kcur] = base + q * 64 + f;
                    ++kcur;
#pragma unroll
                    for (int w = 0; w < NMS_WORDS; ++w) {
                        const unsigned long long r =
                            ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)rh[w], f) << 32) |
                            (unsigned)__builtin_amdgcn_readlane((int)rl[w], f);
                        removed[w] |= r;
                    }
                    removed[q] |= 1ull << f;
                    avail = ~removed[q];
                }
            }
            if (lane == 0) *kept_cnt_sh = kcur;
        }
        __syncthreads();
        const int new_cnt = *kept_cnt_sh;
        // publish the boxes kept in this round for phase A of the next rounds
        for (int j = kept_cnt + t; j < new_cnt; j += POST_THREADS) kept[j] = cand[keep[j] - base];
        kept_cnt = new_cnt;
        __syncthreads();
    }
    return kept_cnt;
}
