// Mask-head tail: logits + sigmoid, and full-image paste + threshold.
//   mmdet FCNMaskHead: ... upsample (ConvTranspose2d 2x2/2) -> ReLU -> conv_logits 1x1
//   (fgn_roi_head.py:380), then get_seg_masks/_do_paste_mask (fgn_roi_head.py:668-671).
//
// The 2x2/stride-2 transposed conv is four independent 1x1 convs, one per output
// sub-position; it runs on the MFMA conv kernel as ONE 1x1 conv with 4*C output
// channels ordered n = (dy*2+dx)*C + co, leaving [D,7,7,(dy,dx),C] in memory.  Because
// conv_logits is point-wise it is applied directly on that layout, and the 14x14 pixel
// shuffle is folded into this kernel's output index - no shuffle pass.
#include "post_common.h"

// x : [D][P*P][4][C]  (post-ReLU upsample output);  w : [C], bias scalar
// prob/logit out : [D][2P][2P]
__global__ __launch_bounds__(256) void mask_logits_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float bias, const float* __restrict__ bias_dev,
                                                          float* __restrict__ logits, float* __restrict__ prob,
                                                          const int32_t* __restrict__ n_dev, int n_det, int P,
                                                          int C) {
    if (bias_dev) bias = *bias_dev;          // device-resident bias (training: no host read of the updated parameter)
    int D = n_det;
    if (n_dev) D = min(D, *n_dev);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int total = D * P * P * 4;
    if (wave >= total) return;
    const float* px = x + (size_t)wave * C;
    float acc = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(px + c);
        const float4 k = *reinterpret_cast<const float4*>(w + c);
        acc += (v.x * k.x + v.y * k.y) + (v.z * k.z + v.w * k.w);
    }
    acc = wave_reduce_sum(acc);
    if (lane == 0) {
        const int sub = wave & 3;
        const int cell = (wave >> 2) % (P * P);
        const int d = (wave >> 2) / (P * P);
        const int y = (cell / P) * 2 + (sub >> 1), xo = (cell % P) * 2 + (sub & 1);
        const float v = acc + bias;
        const size_t o = ((size_t)d * 2 * P + y) * 2 * P + xo;
        logits[o] = v;
        prob[o] = sigmoid32(v);
    }
}

extern "C" int fgn_mask_logits_f32(const float* x, const float* w, float bias, const float* bias_dev, float* logits, float* prob,
                                   const int32_t* n_dev, int n_det, int roi_size, int C, hipStream_t stream) {
    if (!x || !w || !logits || !prob) return FGN_ERR_ARG;
    if (C % 4) return FGN_ERR_SHAPE;
    if (n_det == 0) return FGN_OK;
    const int waves = n_det * roi_size * roi_size * 4;
    hipLaunchKernelGGL(mask_logits_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, stream, x, w, bias, bias_dev, logits, prob,
                       n_dev, n_det, roi_size, C);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ----------------------------------------------------------------------------------------------
// paste: out[d][y][x] = bilinear(prob[d], grid(x,y)) >= thr.  grid_sample semantics: align_corners=False, zero
// padding.  HBM-bound: D*H*W bytes written, 4 pixels per lane.  Two region semantics, as in mmdet's _do_paste_mask:
//   skip_empty = 1 (its CPU path, one mask per chunk): only inside the integer-expanded box
//     (floor(x0)-1 .. ceil(x1)+1, clipped), 0 elsewhere - the oracle's and north_star's "CPU reference";
//   skip_empty = 0 (its CUDA path: the reference runs on cuda:0, main.py:365): the grid spans the whole image.  A
//     sample is non-zero only while its source coordinate lies inside (-1, M), i.e. within half a mask pixel =
//     box_w / (2 M) of the box, so the region computed here is that band (+1 px of slack), not the image; the two
//     semantics agree for thr >= 0.5 on boxes of positive width and height (the value on the box edge is half the
//     border pixel) and differ below it.
//     (thr <= 0 sets every pixel of the image under this semantic: the region is then the image.)
// ----------------------------------------------------------------------------------------------
// One mask sample, separable form shared by the dense paste kernel and the RLE kernel (so both
// produce the same bits): horizontal lerp of the two neighbouring mask rows, then vertical lerp.
// grid_sample semantics: pixel = ((g + 1) * size - 1) / 2, bilinear, zeros outside the map.
struct PasteBox {
    float bx0, by0, bx1, by1;
    int x0i, y0i, x1i, y1i;   // integer-expanded region [x0i,x1i) x [y0i,y1i)
};
__device__ __forceinline__ PasteBox make_paste_box(const float* b, int H, int W, int MS, int skip_empty, float thr) {
    PasteBox p;
    p.bx0 = b[0]; p.by0 = b[1]; p.bx1 = b[2]; p.by1 = b[3];
    if (skip_empty) {
        p.x0i = max((int)floorf(p.bx0) - 1, 0);
        p.y0i = max((int)floorf(p.by0) - 1, 0);
        p.x1i = min((int)ceilf(p.bx1) + 1, W);
        p.y1i = min((int)ceilf(p.by1) + 1, H);
        return p;
    }
    // whole-image grid: the band in which a sample can be non-zero; a degenerate axis (width <= 0: the grid
    // coordinate is inf -> 0, NaN where 0 / 0) samples the mask's centre line everywhere -> the whole axis
    const float bw = p.bx1 - p.bx0, bh = p.by1 - p.by0;
    const float mx = bw / (2.f * (float)MS), my = bh / (2.f * (float)MS);
    const bool all = !(thr > 0.f);
    const bool fx = all || !(bw > 0.f), fy = all || !(bh > 0.f);
    p.x0i = fx ? 0 : max((int)floorf(p.bx0 - mx) - 1, 0);
    p.x1i = fx ? W : min((int)ceilf(p.bx1 + mx) + 1, W);
    p.y0i = fy ? 0 : max((int)floorf(p.by0 - my) - 1, 0);
    p.y1i = fy ? H : min((int)ceilf(p.by1 + my) + 1, H);
    return p;
}
struct AxisLerp {
    int lo;        // low index (may be -1 .. MS-1); high = lo + 1
    float w_hi;    // weight of the high neighbour; low weight = 1 - w_hi
};
__device__ __forceinline__ AxisLerp paste_axis(int pix, float b0, float b1, int MS) {
    float g = ((float)pix + 0.5f - b0) / (b1 - b0) * 2.f - 1.f;
    if (isinf(g)) g = 0.f;
    const float c = ((g + 1.f) * (float)MS - 1.f) / 2.f;
    const float f = floorf(c);
    AxisLerp a;
    // clamp far-outside coordinates so the int conversion is defined; they contribute 0 anyway
    a.lo = (int)fminf(fmaxf(f, -2.f), (float)MS);
    a.w_hi = c - f;
    return a;
}
__device__ __forceinline__ float paste_row_lerp(const float* __restrict__ m, int MS, int row, const AxisLerp& ax) {
    if ((unsigned)row >= (unsigned)MS) return 0.f;
    const float v0 = ((unsigned)ax.lo < (unsigned)MS) ? m[row * MS + ax.lo] : 0.f;
    const float v1 = ((unsigned)(ax.lo + 1) < (unsigned)MS) ? m[row * MS + ax.lo + 1] : 0.f;
    return v0 * (1.f - ax.w_hi) + v1 * ax.w_hi;
}
__device__ __forceinline__ float paste_value(const float* __restrict__ m, int MS, const AxisLerp& ax,
                                             const AxisLerp& ay) {
    const float r0 = paste_row_lerp(m, MS, ay.lo, ax);
    const float r1 = paste_row_lerp(m, MS, ay.lo + 1, ax);
    return r0 * (1.f - ay.w_hi) + r1 * ay.w_hi;
}

__global__ __launch_bounds__(256) void mask_paste_kernel(const float* __restrict__ prob,
                                                         const float* __restrict__ boxes, int box_stride,
                                                         uint8_t* __restrict__ out,
                                                         const int32_t* __restrict__ n_dev, int n_det, int H, int W,
                                                         int MS, float thr, int skip_empty) {
    int D = n_det;
    if (n_dev) D = min(D, *n_dev);
    const long long HW = (long long)H * W;
    const long long total4 = ((long long)n_det * HW + 3) / 4;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < total4;
         q += (long long)gridDim.x * blockDim.x) {
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long i = q * 4 + k;
            if (i >= (long long)n_det * HW) break;
            const int d = (int)(i / HW);
            if (d >= D) continue;
            const int rem = (int)(i - (long long)d * HW);
            const int y = rem / W, x = rem - y * W;
            const PasteBox pb = make_paste_box(boxes + (size_t)d * box_stride, H, W, MS, skip_empty, thr);
            if (x < pb.x0i || x >= pb.x1i || y < pb.y0i || y >= pb.y1i) continue;
            const AxisLerp ax = paste_axis(x, pb.bx0, pb.bx1, MS);
            const AxisLerp ay = paste_axis(y, pb.by0, pb.by1, MS);
            if (paste_value(prob + (size_t)d * MS * MS, MS, ax, ay) >= thr) packed |= 1u << (8 * k);
        }
        if ((q + 1) * 4 <= (long long)n_det * HW) {
            reinterpret_cast<uint32_t*>(out)[q] = packed;
        } else {
            for (int k = 0; k < 4 && q * 4 + k < (long long)n_det * HW; ++k) out[q * 4 + k] = (packed >> (8 * k)) & 0xff;
        }
    }
}

extern "C" int fgn_mask_paste_u8(const float* prob, const float* boxes, int box_stride, uint8_t* out,
                                 const int32_t* n_dev, int n_det, int img_h, int img_w, int mask_size, float thr,
                                 int skip_empty, hipStream_t stream) {
    if (!prob || !boxes || !out) return FGN_ERR_ARG;
    if (n_det == 0) return FGN_OK;
    const long long total4 = ((long long)n_det * img_h * img_w + 3) / 4;
    const int grid = (int)std::min<long long>((total4 + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(mask_paste_kernel, dim3(grid), dim3(256), 0, stream, prob, boxes, box_stride, out, n_dev,
                       n_det, img_h, img_w, mask_size, thr, skip_empty);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ----------------------------------------------------------------------------------------------
// paste + threshold + COCO RLE, fused: the D x H x W masks are never materialised.
// Replaces get_seg_masks -> .cpu().numpy() -> pycocotools encode (fgn_roi_head.py:668-671,
// fgn.py:267,281): instead of writing D*H*W bytes, copying them over PCIe and run-length
// encoding on a host core, one workgroup per detection
//   1. counts the value transitions of every image column inside the pasted box
//      (column-major = pycocotools' Fortran order), one thread per column,
//   2. block-scans the counts and writes the transition positions in order,
//   3. turns positions into run lengths, delta-codes them against the run two back and
//      emits the COCO 5-bit/continuation ASCII string, again through a block scan.
// Only the strings (a few hundred bytes per detection) cross PCIe.
// Overflow of either cap sets overflow[d]; the host then falls back to the dense kernel for
// that detection, so results never depend on the caps.
// ----------------------------------------------------------------------------------------------
constexpr int RLE_THREADS = 1024;

__device__ inline int block_exclusive_scan(int v, int* wave_sums, int* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    __syncthreads();   // protects wave_sums reuse across calls
    if (lane == 63) wave_sums[wv] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < RLE_THREADS / 64; ++w) {
        const int s = wave_sums[w];
        if (w < wv) base += s;
        tot += s;
    }
    *total = tot;
    return base + incl - v;
}

__device__ __forceinline__ int rle_char_count(long long x) {
    int n = 0;
    bool more = true;
    while (more) {
        const int c = (int)(x & 0x1f);
        x >>= 5;
        more = (c & 0x10) ? (x != -1) : (x != 0);
        ++n;
    }
    return n;
}

// Step 3 of both RLE kernels: sorted transition positions tr[0..T) of the column-major pixel sequence -> run
// lengths -> COCO string (rleToString of pycocotools: delta against the run two back from the 4th run on, 5 data
// bits + continuation bit per character, +48).  Whole workgroup (RLE_THREADS); writes *out_len / *overflow.
__device__ inline void emit_coco_string(const uint32_t* __restrict__ tr, int T, long long HWl, uint8_t* __restrict__ ob,
                                        int byte_cap, int32_t* out_len, int32_t* overflow, int* wave_sums) {
    const int t = threadIdx.x;
    auto run_len = [&](int i) -> long long {   // i in [0, T]
        const long long hi = (i < T) ? (long long)tr[i] : HWl;
        const long long lo = (i > 0) ? (long long)tr[i - 1] : 0;
        return hi - lo;
    };
    const int n_runs = T + 1;
    const int rpt = (n_runs + RLE_THREADS - 1) / RLE_THREADS;
    int nchar = 0;
    for (int j = 0; j < rpt; ++j) {
        const int i = t * rpt + j;
        if (i < n_runs) {
            long long x = run_len(i);
            if (i > 2) x -= run_len(i - 2);
            nchar += rle_char_count(x);
        }
    }
    int total_chars;
    int o = block_exclusive_scan(nchar, wave_sums, &total_chars);
    if (total_chars > byte_cap) {
        if (t == 0) { *overflow = 1; *out_len = 0; }
        return;
    }
    for (int j = 0; j < rpt; ++j) {
        const int i = t * rpt + j;
        if (i < n_runs) {
            long long x = run_len(i);
            if (i > 2) x -= run_len(i - 2);
            bool more = true;
            while (more) {
                int c = (int)(x & 0x1f);
                x >>= 5;
                more = (c & 0x10) ? (x != -1) : (x != 0);
                if (more) c |= 0x20;
                ob[o++] = (uint8_t)(c + 48);
            }
        }
    }
    if (t == 0) { *out_len = total_chars; *overflow = 0; }
}

__global__ __launch_bounds__(RLE_THREADS) void mask_rle_kernel(
    const float* __restrict__ prob, const float* __restrict__ boxes, int box_stride, uint32_t* __restrict__ trans,
    uint8_t* __restrict__ out_bytes, int32_t* __restrict__ out_len, int32_t* __restrict__ overflow,
    const int32_t* __restrict__ n_dev, int n_det, int H, int W, int MS, float thr, int trans_cap, int byte_cap,
    int skip_empty) {
    __shared__ float m[32 * 32];
    __shared__ int wave_sums[RLE_THREADS / 64];
    const int d = blockIdx.x, t = threadIdx.x;
    int D = n_det;
    if (n_dev) D = min(D, *n_dev);
    if (d >= D) {
        if (t == 0) { out_len[d] = 0; overflow[d] = 0; }
        return;
    }
    for (int i = t; i < MS * MS; i += RLE_THREADS) m[i] = prob[(size_t)d * MS * MS + i];
    __syncthreads();
    const PasteBox pb = make_paste_box(boxes + (size_t)d * box_stride, H, W, MS, skip_empty, thr);
    // columns x0i .. min(x1i, W-1): one past the region closes a run that wraps a full-height column
    const int xs = pb.x0i, xe = min(pb.x1i, W - 1);
    const int ncols = max(xe - xs + 1, 0);
    const int cpt = (ncols + RLE_THREADS - 1) / RLE_THREADS;   // contiguous columns per thread
    // one row past the region closes a run; a region touching the bottom edge wraps into the next
    // column's row 0, so such columns are scanned from row 0
    const int ys = (pb.y1i >= H) ? 0 : pb.y0i, ye = min(pb.y1i, H - 1);

    auto value_at = [&](int x, int y, const AxisLerp& ax) -> int {
        if (x < pb.x0i || x >= pb.x1i || y < pb.y0i || y >= pb.y1i) return 0;
        const AxisLerp ay = paste_axis(y, pb.by0, pb.by1, MS);
        return paste_value(m, MS, ax, ay) >= thr ? 1 : 0;
    };
    auto scan_column = [&](int x, uint32_t* dst) -> int {   // dst == nullptr: count only
        const AxisLerp ax = paste_axis(x, pb.bx0, pb.bx1, MS);
        int prev = 0;
        if (ys == 0 && x > 0) {
            const AxisLerp axp = paste_axis(x - 1, pb.bx0, pb.bx1, MS);
            prev = value_at(x - 1, H - 1, axp);
        }
        int n = 0;
        for (int y = ys; y <= ye; ++y) {
            const int v = value_at(x, y, ax);
            if (v != prev) {
                if (dst) dst[n] = (uint32_t)x * (uint32_t)H + (uint32_t)y;
                ++n;
                prev = v;
            }
        }
        return n;
    };

    // ---- 1. count transitions per thread (contiguous columns) -------------------------------
    int cnt = 0;
    for (int j = 0; j < cpt; ++j) {
        const int c = t * cpt + j;
        if (c < ncols) cnt += scan_column(xs + c, nullptr);
    }
    int T;
    const int off = block_exclusive_scan(cnt, wave_sums, &T);
    if (T > trans_cap) {
        if (t == 0) { overflow[d] = 1; out_len[d] = 0; }
        return;
    }
    // ---- 2. write positions in order ----------------------------------------------------------
    uint32_t* tr = trans + (size_t)d * trans_cap;
    {
        int o = off;
        for (int j = 0; j < cpt; ++j) {
            const int c = t * cpt + j;
            if (c < ncols) o += scan_column(xs + c, tr + o);
        }
    }
    __syncthreads();
    emit_coco_string(tr, T, (long long)H * W, out_bytes + (size_t)d * byte_cap, byte_cap, out_len + d, overflow + d,
                     wave_sums);
}

extern "C" int fgn_mask_rle(const float* prob, const float* boxes, int box_stride, uint32_t* trans_scratch,
                            uint8_t* out_bytes, int32_t* out_len, int32_t* overflow, const int32_t* n_dev,
                            int n_det, int img_h, int img_w, int mask_size, float thr, int trans_cap, int byte_cap,
                            int skip_empty, hipStream_t stream) {
    if (!prob || !boxes || !trans_scratch || !out_bytes || !out_len || !overflow) return FGN_ERR_ARG;
    if (mask_size > 32 || mask_size < 1 || trans_cap < 1 || byte_cap < 8) return FGN_ERR_SHAPE;
    if ((long long)img_h * img_w >= (1ll << 32)) return FGN_ERR_SHAPE;
    if (n_det == 0) return FGN_OK;
    hipLaunchKernelGGL(mask_rle_kernel, dim3(n_det), dim3(RLE_THREADS), 0, stream, prob, boxes, box_stride,
                       trans_scratch, out_bytes, out_len, overflow, n_dev, n_det, img_h, img_w, mask_size, thr,
                       trans_cap, byte_cap, skip_empty);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ----------------------------------------------------------------------------------------------
// COCO RLE of DENSE binary masks: the ground-truth masks of the query image, which the reference moves to the
// GPU with the rest of the batch (fgn.py:92-99) and run-length encodes on the host at the end of simple_test
// (fgn.py:298, `qry_isegmaps_rle`).  Here they stay on the device: 4 host milliseconds of numpy per episode become
// four small kernels beside the network, and only the strings cross PCIe.
//   1. mask_to_columns_kernel: [n][H][W] bytes -> column-major [n][W][Hp] (Hp = H rounded up to 16, the pad rows
//      repeat the column's last pixel), i.e. pycocotools' Fortran scan order as contiguous 16-byte chunks;
//   2. dense_rle_walk_kernel (count, then emit; 32 workgroups per mask): value changes against the preceding pixel (the
//      pad makes "previous pixel of row 0" the last row of the previous column), positions x*H + y written in order;
//   3. dense_rle_string_kernel, one workgroup per mask: the shared string emitter.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_to_columns_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                               int H, int W, int Hp) {
    __shared__ uint8_t tile[64][65];
    const int n = blockIdx.z;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
    const uint8_t* src = in + (size_t)n * H * W;
    uint8_t* dst = out + (size_t)n * W * Hp;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int y = min(y0 + r, H - 1), x = x0 + tx;          // rows past H repeat row H-1 (the pad)
        tile[r][tx] = (x < W) ? (src[(size_t)y * W + x] != 0) : 0;
    }
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {
        const int x = x0 + c, y = y0 + tx;
        if (x < W && y < Hp) dst[(size_t)x * Hp + y] = tile[tx][c];
    }
}

// The walk over a mask's change bits, split over DENSE_SEGS workgroups per mask (round 5: one 1024-thread workgroup per
// mask kept 4 of the 256 CUs busy for ~100 us - 66 steps of ~150 instructions per wave, two integer divisions among them -
// on the caller stream of the pipelined serving loop).  Three launches:
//   dense_rle_walk_kernel<false>  per (segment, mask): counts the transitions of its contiguous chunk range
//   dense_rle_walk_kernel<true>   ...: its offset = the counts of the segments before it; writes its transitions
//   dense_rle_string_kernel       one workgroup per mask: transitions -> COCO string (emit_coco_string)
// Inside a segment every wave owns a contiguous chunk range and walks it 64 chunks at a time (one contiguous kilobyte per
// wave-load, DENSE_U loads issued ahead); the column / chunk-in-column of a chunk advance incrementally (no division).
constexpr int DENSE_SEGS = 32;
constexpr int DENSE_THREADS = 256;
constexpr int DENSE_U = 4;

template <bool EMIT>
__global__ __launch_bounds__(DENSE_THREADS) void dense_rle_walk_kernel(const uint8_t* __restrict__ cols,
                                                                         uint32_t* __restrict__ trans,
                                                                         int32_t* __restrict__ seg_counts,
                                                                         int H, int W, int Hp, int trans_cap) {
    __shared__ int wave_sums[DENSE_THREADS / 64];
    const int seg = blockIdx.x, d = blockIdx.y, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint8_t* m = cols + (size_t)d * W * Hp;
    const int cpc = Hp / 16;                                   // chunks per column
    const int n_chunks = W * cpc;
    constexpr int NW = DENSE_THREADS / 64;
    const int per_seg = ((n_chunks + DENSE_SEGS - 1) / DENSE_SEGS + 64 * NW - 1) / (64 * NW) * (64 * NW);
    const int per_wave = per_seg / NW;                         // a multiple of 64
    const int w_lo = min(seg * per_seg + wv * per_wave, n_chunks), w_hi = min(w_lo + per_wave, n_chunks);
    int base = 0;
    if (EMIT) {
        int total = 0;
        for (int s2 = 0; s2 < DENSE_SEGS; ++s2) {
            const int c = seg_counts[d * DENSE_SEGS + s2];
            if (s2 < seg) base += c;
            total += c;
        }
        if (total > trans_cap) return;                         // (dense_rle_string_kernel flags the overflow)
    }
    auto load_chunk = [&](int c) -> uint4 {
        return c < w_hi ? *reinterpret_cast<const uint4*>(m + (size_t)c * 16) : make_uint4(0u, 0u, 0u, 0u);
    };
    // change bits of chunk c = chunk k of column x (its 16 bytes in v): bit r set iff pixel (x, 16 k + r) differs from its
    // predecessor in scan order (the last pixel of the previous chunk: the neighbouring lane's, or for lane 0 one byte load)
    auto chunk_bits = [&](int c, int k, const uint4& v) -> unsigned {
        const bool in = c < w_hi;
        // the 16 bytes are 0 / 1 (mask_to_columns_kernel normalises them): a multiply gathers the four low bits of a
        // word into one nibble (b0 | b1 << 1 | b2 << 2 | b3 << 3 lands in bits 24..27, the partial products never
        // carry), and "differs from its predecessor" is one XOR against the value shifted by a pixel
        const unsigned px = ((v.x * 0x01020408u) >> 24 & 0xfu) | ((v.y * 0x01020408u) >> 20 & 0xf0u) |
                            ((v.z * 0x01020408u) >> 16 & 0xf00u) | ((v.w * 0x01020408u) >> 12 & 0xf000u);
        unsigned prev = __shfl_up(px >> 15, 1, 64);
        if (lane == 0) prev = (in && c > 0) ? (unsigned)(m[(size_t)c * 16 - 1] & 1) : 0u;
        if (!in) return 0u;
        const unsigned bits = (px ^ ((px << 1) | prev)) & 0xffffu;
        const int valid = min(16, H - 16 * k);                  // pad rows never differ, but mask them anyway
        return valid >= 16 ? bits : (bits & ((1u << valid) - 1u));
    };
    // (x, k) of this lane's chunk, advanced by 64 chunks per step without a division
    int c = w_lo + lane;
    int x = c / cpc, k = c - x * cpc;
    const int dx = 64 / cpc, dk = 64 - dx * cpc;
    uint32_t* tr = trans + (size_t)d * trans_cap;
    int cnt = 0, run = 0;
    // ---- pass A (both variants): this thread's count (the emitting variant needs it for its offsets inside the segment)
    {
        int ca = c, ka = k;
        for (int c0 = w_lo; c0 < w_hi; c0 += 64 * DENSE_U) {
            uint4 v[DENSE_U];
#pragma unroll
            for (int u = 0; u < DENSE_U; ++u) v[u] = load_chunk(c0 + 64 * u + lane);
#pragma unroll
            for (int u = 0; u < DENSE_U; ++u) {
                if (c0 + 64 * u < w_hi) cnt += __popc(chunk_bits(ca, ka, v[u]));
                ca += 64; ka += dk;
                if (ka >= cpc) ka -= cpc;
            }
        }
    }
    // workgroup-level exclusive scan of the per-thread counts (thread order = wave-major); a wave's base = lane 0's
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wave_sums[wv] = incl;
    __syncthreads();
    int wave_base = 0, seg_total = 0;
    for (int w = 0; w < NW; ++w) {
        if (w < wv) wave_base += wave_sums[w];
        seg_total += wave_sums[w];
    }
    if (!EMIT) {
        if (t == 0) seg_counts[d * DENSE_SEGS + seg] = seg_total;
        return;
    }
    // ---- pass B: emit, in (step, lane) order inside the wave's range
    run = base + wave_base;
    for (int c0 = w_lo; c0 < w_hi; c0 += 64 * DENSE_U) {
        uint4 v[DENSE_U];
#pragma unroll
        for (int u = 0; u < DENSE_U; ++u) v[u] = load_chunk(c0 + 64 * u + lane);
#pragma unroll
        for (int u = 0; u < DENSE_U; ++u) {
            if (c0 + 64 * u < w_hi) {                            // (wave-uniform)
                unsigned bits = chunk_bits(c, k, v[u]);
                const int n = __popc(bits);
                int sc = n;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int up = __shfl_up(sc, off, 64);
                    if (lane >= off) sc += up;
                }
                int o = run + sc - n;
                run += __shfl(sc, 63, 64);
                while (bits) {
                    const int r = __ffs(bits) - 1;
                    bits &= bits - 1;
                    tr[o++] = (uint32_t)x * (uint32_t)H + (uint32_t)(16 * k + r);
                }
            }
            c += 64; x += dx; k += dk;
            if (k >= cpc) { k -= cpc; ++x; }
        }
    }
}

__global__ __launch_bounds__(RLE_THREADS) void dense_rle_string_kernel(const uint32_t* __restrict__ trans,
                                                                       const int32_t* __restrict__ seg_counts,
                                                                       uint8_t* __restrict__ out_bytes,
                                                                       int32_t* __restrict__ out_len,
                                                                       int32_t* __restrict__ overflow, int H, int W,
                                                                       int trans_cap, int byte_cap) {
    __shared__ int wave_sums[RLE_THREADS / 64];
    const int d = blockIdx.x, t = threadIdx.x;
    int T = 0;
    for (int s2 = 0; s2 < DENSE_SEGS; ++s2) T += seg_counts[d * DENSE_SEGS + s2];
    if (T > trans_cap) {
        if (t == 0) { overflow[d] = 1; out_len[d] = 0; }
        return;
    }
    emit_coco_string(trans + (size_t)d * trans_cap, T, (long long)H * W, out_bytes + (size_t)d * byte_cap, byte_cap,
                     out_len + d, overflow + d, wave_sums);
}

extern "C" size_t fgn_dense_rle_scratch_bytes(int n_masks, int img_h, int img_w, int trans_cap) {
    const size_t hp = (size_t)(img_h + 15) / 16 * 16;
    return (size_t)n_masks * img_w * hp + (size_t)n_masks * trans_cap * sizeof(uint32_t) + 256 +
           (size_t)n_masks * DENSE_SEGS * sizeof(int32_t) + 256;
}

extern "C" int fgn_dense_mask_rle(const uint8_t* masks, void* scratch, size_t scratch_bytes, uint8_t* out_bytes,
                                  int32_t* out_len, int32_t* overflow, int n_masks, int img_h, int img_w, int trans_cap,
                                  int byte_cap, hipStream_t stream) {
    if (!masks || !scratch || !out_bytes || !out_len || !overflow) return FGN_ERR_ARG;
    if (n_masks == 0) return FGN_OK;
    if (img_h < 1 || img_w < 1 || trans_cap < 1 || byte_cap < 8 || (long long)img_h * img_w >= (1ll << 32) ||
        n_masks > 65535)
        return FGN_ERR_SHAPE;
    if (scratch_bytes < fgn_dense_rle_scratch_bytes(n_masks, img_h, img_w, trans_cap)) return FGN_ERR_ARG;
    const int hp = (img_h + 15) / 16 * 16;
    uint8_t* cols = reinterpret_cast<uint8_t*>(scratch);
    const size_t cols_bytes = ((size_t)n_masks * img_w * hp + 255) / 256 * 256;
    uint32_t* trans = reinterpret_cast<uint32_t*>(cols + cols_bytes);
    const size_t trans_bytes = ((size_t)n_masks * trans_cap * sizeof(uint32_t) + 255) / 256 * 256;
    int32_t* seg_counts = reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(trans) + trans_bytes);
    hipLaunchKernelGGL(mask_to_columns_kernel, dim3(cdiv(img_w, 64), cdiv(hp, 64), n_masks), dim3(256), 0, stream, masks,
                       cols, img_h, img_w, hp);
    FGN_LAUNCH_CHECK();
    hipLaunchKernelGGL(dense_rle_walk_kernel<false>, dim3(DENSE_SEGS, n_masks), dim3(DENSE_THREADS), 0, stream, cols, trans,
                       seg_counts, img_h, img_w, hp, trans_cap);
    FGN_LAUNCH_CHECK();
    hipLaunchKernelGGL(dense_rle_walk_kernel<true>, dim3(DENSE_SEGS, n_masks), dim3(DENSE_THREADS), 0, stream, cols, trans,
                       seg_counts, img_h, img_w, hp, trans_cap);
    FGN_LAUNCH_CHECK();
    hipLaunchKernelGGL(dense_rle_string_kernel, dim3(n_masks), dim3(RLE_THREADS), 0, stream, trans, seg_counts, out_bytes,
                       out_len, overflow, img_h, img_w, trans_cap, byte_cap);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}
