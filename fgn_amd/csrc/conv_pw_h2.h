// conv_pw_h2_kernel - the point-wise / grouped GEMM of conv_pw_x3_kernel with THREE f16 MFMA products per f32 product
// instead of six bf16 ones.  Included by conv_igemm.hip after conv_pw_x3.h (ConvParams, make_rsrc, lds_dma16_s, BK).
//
// Why.  conv_pw_x3_kernel is power-bound: what it costs is the number of MFMA products per f32 product.  An f32 value
// scaled by a power of two into the f16 range is h + l + e with h = f16(x), l = f16(x - h) (both round-to-nearest, the
// subtraction exact) and |e| <= 2^-23 |x|: l keeps 11 of the at most 13 bits h leaves, so the two planes hold x to one ulp
// of the f32 value at worst, a quarter of it on average.  Products of two f16 values are exact in f32 (22 significant
// bits), so a * b = ha hb + ha lb + la hb + (la lb <= 2^-22 |a b|, left out): each product within 2^-21 |a b| at worst and
// ~2^-24 |a b| rms - the order of the rounding an f32 FMA chain makes at every step: three v_mfma_f32_16x16x32_f16 per f32
// MFMA's worth of K,
// accumulated in f32 by the matrix pipe.  Against fp64 the result is as close as the f32 kernels' (the accumulation's own
// rounding dominates all three arithmetics: tests/test_hip_conv.py::test_h2_*, tools/x3_probe.py).
//
// Range.  f16 has 5 exponent bits, so the operands are scaled by powers of two (exact) and the scales taken out again:
//  * the weights per output column at pack time (ops.pack_h2: column maximum into [2^14, 2^15); the inverse scales, [group]
//    [npad] floats, lie behind the image and multiply the C tile in the epilogue);
//  * the activations by a scale the kernel finds itself, per WAVE and OUTPUT TILE: the first K-tile of the tile whose
//    32 RB x 32 fragment holds a non-zero element sets S so that the fragment's largest |x| lands in [2^13, 2^14); every
//    later K-tile is looked at once (max |x| of the raw fragment against 65504 / S: `v_max3_f32` with |.| modifiers, 0.25 vector instruction per element),
//    and one that would leave the f16 range - an element 4x .. 8x above what S was chosen for - picks a new S from its
//    own maximum and multiplies the accumulators by the ratio (a power of two: exact).  S comes out of the accumulators
//    when they go to the C tile.  An element within 2^15 of the maximum S was chosen for keeps both planes at full
//    precision; a smaller one loses low bits of its l plane (f16 subnormals): an absolute error below 2^-38 of that
//    maximum per element.  The splits are scale-invariant otherwise: row tiles / stage counts differ only through such
//    elements (tests: within 1e-7 of the range of each other).  No producer has to supply a maximum: the form that
//    took one from device memory (written by the Winograd input transform with atomics) ran the GEMM 5-10 % faster and
//    the transform 3x slower (profiles/r05_ab_h2_100steps.jsonl, r05_h2_probe_record_vs_own_scale.jsonl; DESIGN appendix A).
//
// Structure: conv_pw_x3_kernel's (16x16x32 form): BM = 32 RB WMW rows x 128 columns, WMW x 2 waves, wave tile 32 RB x 64,
// a ring of NST LDS stages of (BM * 128 + 16384) bytes running across output tiles (the C tile goes through the last
// stage one wave row at a time: 32 RB rows x 512 bytes fit it for either row tile), the f32 activations split in
// registers after the ds_read (2 vector instructions per element: v_fma_mixlo / mixhi_f16), the weight image
// [group][K-tile][plane][npad][32] f16 with the k order and chunk swizzle of ops.pack_x3's 16x16x32 form.  64 rows: 48 KB
// of LDS, three workgroups per CU; 128 rows: 64 KB, two.
#pragma once

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

struct H2Frag { f16x8_t hi, lo; };

// four f32, scaled by the power of two s (wave-uniform) -> hi = f16(x s), lo = f16(x s - hi) as two packed pairs each.
// v_fma_mixlo / mixhi_f16 take f32 and f16 sources in one fma and round the f32 result to f16 once: x s is exact (a power
// of two), x s - hi is exact in f32, so each plane is one instruction per element (the compiler's own sequence for the
// same C expressions is three).  The pairs are interleaved so that no instruction reads a register in the slot right
// behind a half-register write to it (gfx940+ dst-forwarding rule; the hazard recognizer does not look inside an asm block).
__device__ __forceinline__ void h2_split4(const float x0, const float x1, const float x2, const float x3, const float s,
                                          unsigned& h01, unsigned& h23, unsigned& l01, unsigned& l23) {
    asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
        "v_fma_mixlo_f16 %1, %6, %8, 0\n\t"
        "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
        "v_fma_mixhi_f16 %1, %7, %8, 0\n\t"
        "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 1"        // an MFMA may read the planes next: VALU write -> MFMA read needs two wait states (not inserted for asm)
        : "=&v"(h01), "=&v"(h23), "=&v"(l01), "=&v"(l23)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(s));
}

// eight f32 -> two planes of eight f16
__device__ __forceinline__ H2Frag h2_split(const float4& a, const float4& b, const float s) {
    unsigned h[4], l[4];
    h2_split4(a.x, a.y, a.z, a.w, s, h[0], h[1], l[0], l[1]);
    h2_split4(b.x, b.y, b.z, b.w, s, h[2], h[3], l[2], l[3]);
    H2Frag f;
    f.hi = __builtin_bit_cast(f16x8_t, make_uint4(h[0], h[1], h[2], h[3]));
    f.lo = __builtin_bit_cast(f16x8_t, make_uint4(l[0], l[1], l[2], l[3]));
    return f;
}

// max |x| over sixteen f32 values: eight v_max3_f32 with |.| source modifiers (the compiler's own sequence for the same
// fmaxf / fabsf tree is three times as long: it does not fold the inner maximum)
__device__ __forceinline__ float h2_absmax16(const float4& a, const float4& b, const float4& c, const float4& d) {
    float m;
    asm("v_max3_f32 %0, |%1|, |%2|, 0\n\t"
        "v_max3_f32 %0, |%3|, |%4|, %0\n\t"
        "v_max3_f32 %0, |%5|, |%6|, %0\n\t"
        "v_max3_f32 %0, |%7|, |%8|, %0\n\t"
        "v_max3_f32 %0, |%9|, |%10|, %0\n\t"
        "v_max3_f32 %0, |%11|, |%12|, %0\n\t"
        "v_max3_f32 %0, |%13|, |%14|, %0\n\t"
        "v_max3_f32 %0, |%15|, |%16|, %0"
        : "=&v"(m)
        : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w),
          "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w), "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w));
    return m;
}

constexpr int H2_BN = 128;                       // columns the weight image is padded to

// Second tensor / spatial structure of an IM2COL launch (conv_pw_h2_kernel<..., true>): a KH x KW / stride / pad convolution
// (KW 1 or 3, Cin / 32 a power of two) as an implicit GEMM over the rows of ONE or TWO NHWC tensors with the same weights
// (the query map and the support maps of a backbone layer): rows [0, M0) are the output pixels of tensor 0 (ConvParams:
// n_img, H, W, Ho, Wo; input at x + x_off0 bytes, output y), rows [M0, M0 + M1) those of tensor 1 (input at x + x_off1,
// output y1).  Both inputs lie inside the one buffer descriptor [x, x + x_bytes).
struct H2Im2col {
    float* y1;
    unsigned x_off0, x_off1;
    int M0, M1;
    int H1, W1, Ho1, Wo1;
    int cin_shift;            // log2(Cin / 32)
};

template <int WMW, int WNW, int RB, int NST, bool IM2COL>
__global__ __launch_bounds__(64 * WMW * WNW, 2) void conv_pw_h2_kernel(const ConvParams p, const H2Im2col q2, const int total_tiles) {
    constexpr int BM = 32 * RB * WMW, BN = 64 * WNW, NTHR = 64 * WMW * WNW, NW = WMW * WNW;
    constexpr int A_LD = BM / 8 / NW;                       // activation wave-instructions per wave per K-tile (8 rows each)
    constexpr int A_STAGE = BM * 128;                       // bytes
    constexpr int B_STAGE = 2 * BN * 64;                    // bytes of one K-tile of the weight image for BN columns
    constexpr int STAGE = A_STAGE + B_STAGE;                // bytes
    constexpr int BQ = BN / 16;                             // weight wave-instructions per plane per K-tile (16 rows each)
    constexpr int B_LD = 2 * BQ / NW;                       // weight wave-instructions per wave per K-tile
    static_assert(A_LD * 8 * NW == BM && B_LD * NW == 2 * BQ, "tile / wave split");
    constexpr int PER = A_LD + B_LD;                        // LDS-DMA wave-instructions per wave per K-tile
    constexpr int D = NST - 1;                              // K-tiles in flight ahead of the one being multiplied
    constexpr int ROWS_PER_PASS = NTHR / 8;                 // A rows one pass of the workgroup's DMAs covers
    static_assert(NST == 2 || NST == 3, "ring of 2 or 3 stages");      // (three: measured 5-12 % slower, DESIGN appendix A row 53)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h2[];

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv / WNW, wn = wv % WNW;
    const int M = IM2COL ? q2.M0 + q2.M1 : (p.n_img_dev ? min(p.n_img, *p.n_img_dev) : p.n_img) * p.Ho * p.Wo;
    int grp_valid = p.grp_valid;
    if (p.grp_rows && p.grp_count_dev) grp_valid = min(grp_valid, min(p.grp_items, *p.grp_count_dev) * p.grp_rows_per_item);

    float a_s = 1.f, a_inv = 1.f;                   // the wave's activation scale and its inverse (powers of two)

    const int col4 = t & 7, row0 = t >> 3;
    const int src_c4 = col4 ^ ((row0 >> 1) & 7);
    const i32x4 x_rs = make_rsrc(p.x, p.x_bytes);
    const i32x4 w_rs = make_rsrc(p.w3, p.w3_bytes);
    const bool dual = !IM2COL && p.x2 != nullptr;
    const i32x4 x2_rs = make_rsrc(dual ? p.x2 : p.x, dual ? p.x2_bytes : p.x_bytes);
    const int KT = p.K / BK;                                                  // >= D (checked by the launcher)
    const unsigned kt_bytes = (unsigned)(2 * p.npad3 * 64);          // one K-tile of the image, both planes, all rows
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<size_t>(smem_h2));
    const unsigned wave_row_bytes = __builtin_amdgcn_readfirstlane(wv) * 8 * 128;
    constexpr unsigned OOB = 0x7ffffff0u;

    const int nq = total_tiles >> 3, nr = total_tiles & 7;
    auto coords = [&](int tile, int& m0, int& n0) -> bool {
        const int xcd = tile & 7, idx = tile >> 3;
        const int bid = (xcd < nr ? xcd * (nq + 1) : nr * (nq + 1) + (xcd - nr) * nq) + idx;
        int tile_m = bid / p.n_tiles_n;
        int tile_n = bid - tile_m * p.n_tiles_n;
        if (p.band_nt > 0) {
            const int per_grp = p.band_mt * p.n_tiles_n;
            const int grp = bid / per_grp;
            int r = bid - grp * per_grp;
            const int per_band = p.band_mt * p.band_nt;
            const int band = r / per_band;
            r -= band * per_band;
            const int mi = r / p.band_nt;
            tile_m = grp * p.band_mt + mi;
            tile_n = band * p.band_nt + (r - mi * p.band_nt);
        }
        m0 = tile_m * BM;
        n0 = tile_n * BN;
        if (m0 >= M) return false;
        if (p.grp_rows && m0 - (m0 / p.grp_rows) * p.grp_rows >= grp_valid) return false;
        return true;
    };
    auto next_active = [&](int tile, int& m0, int& n0) -> int {
        for (; tile < total_tiles; tile += gridDim.x)
            if (coords(tile, m0, n0)) return tile;
        return -1;
    };

    // IM2COL: a[i] = byte offset of filter tap (0, 0) of the row's receptive field (may lie before the tensor: wraps),
    // a2[i] = bit (ky KW + kx) set when the tap is inside the image (and the row < M) | bit 31 when the row is of tensor 1
    struct Offs { unsigned a[A_LD], a2[A_LD], b[B_LD]; };
    unsigned b_lds[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const int q = wv + NW * i;                     // wave-instruction q of 2 BQ: plane q / BQ, rows 16 * (q % BQ) ..
        b_lds[i] = __builtin_amdgcn_readfirstlane((unsigned)(A_STAGE + (q / BQ) * (BN * 64) + (q % BQ) * 1024));
    }
    auto offsets = [&](int m0, int n0) -> Offs {
        Offs o;
        // rows past M, and the rows of a group past its valid ones (the unwritten tail of a Winograd V), are fetched out of
        // bounds - zeros: a wave's rows share one scale, so whatever lies in memory there must not reach the fragment
        const int g_end = p.grp_rows ? (m0 / p.grp_rows) * p.grp_rows + grp_valid : M;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int m = m0 + row0 + ROWS_PER_PASS * i;
            if constexpr (IM2COL) {
                o.a[i] = 0u; o.a2[i] = 0u;
                if (m < M) {
                    const bool t1 = m >= q2.M0;
                    const int mm = t1 ? m - q2.M0 : m;
                    const int H = t1 ? q2.H1 : p.H, W = t1 ? q2.W1 : p.W, Wo = t1 ? q2.Wo1 : p.Wo;
                    const int HoWo = (t1 ? q2.Ho1 : p.Ho) * Wo;
                    const int img = mm / HoWo;
                    const int rem = mm - img * HoWo;
                    const int oy = rem / Wo;
                    const int ox = rem - oy * Wo;
                    const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
                    o.a[i] = (t1 ? q2.x_off1 : q2.x_off0) + (unsigned)((((img * H + iy0) * W + ix0) * p.Cin + src_c4 * 4) * 4);
                    unsigned tm = t1 ? 0x80000000u : 0u;
                    int tp = 0;
                    for (int ky = 0; ky < p.KH; ++ky) {
                        const bool y_ok = (unsigned)(iy0 + ky) < (unsigned)H;
                        for (int kx = 0; kx < p.KW; ++kx, ++tp)
                            if (y_ok && (unsigned)(ix0 + kx) < (unsigned)W) tm |= 1u << tp;
                    }
                    o.a2[i] = tm;
                }
                continue;
            }
            const bool in = m < M && m < g_end;
            o.a[i] = in ? (unsigned)((m * p.Cin + src_c4 * 4) * 4) : OOB;
            o.a2[i] = OOB;
            if (dual && in) o.a2[i] = (unsigned)(((p.x2_rows ? p.x2_rows[m] : m) * p.cin2 + src_c4 * 4) * 4);
        }
        unsigned g0 = 0;
        if (p.grp_rows) g0 = (unsigned)(m0 / p.grp_rows) * (unsigned)KT * kt_bytes;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int q = wv + NW * i;
            const int plane = q / BQ, row = (q % BQ) * 16 + (lane >> 2);
            o.b[i] = g0 + (unsigned)(((plane * p.npad3 + n0 + row) * 4 + (lane & 3)) * 16);
        }
        return o;
    };
    auto issue = [&](const Offs& o, int kt, int stage) {
        const unsigned st = lds_base + stage * STAGE;
        const unsigned sa = st + wave_row_bytes;
        const unsigned ko = (unsigned)(kt * BK * 4);
        if constexpr (IM2COL) {
            // K-tile -> (filter tap, channel slice), wave-uniform; the tap's byte offset differs with the tensor's width
            const int tap = kt >> q2.cin_shift;
            const int c0 = (kt - (tap << q2.cin_shift)) * BK;
            const int ky = p.KW == 3 ? (tap * 43) >> 7 : tap;             // tap / 3 for tap <= 8; KW == 1: one column
            const int kx = tap - ky * p.KW;
            const unsigned off0 = (unsigned)(((ky * p.W + kx) * p.Cin + c0) * 4);
            const unsigned off1 = (unsigned)(((ky * q2.W1 + kx) * p.Cin + c0) * 4);
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const bool ok = (o.a2[i] >> tap) & 1u;
                const unsigned voff = ok ? o.a[i] + ((o.a2[i] >> 31) ? off1 : off0) : OOB;     // out-of-image taps: zeros
                lds_dma16_s(x_rs, sa + i * ROWS_PER_PASS * 128, voff, 0u);
            }
        } else if (dual && kt >= p.kt1) {
            const unsigned ko2 = (unsigned)((kt - p.kt1) * BK * 4);
#pragma unroll
            for (int i = 0; i < A_LD; ++i) lds_dma16_s(x2_rs, sa + i * ROWS_PER_PASS * 128, o.a2[i], ko2);
        } else {
#pragma unroll
            for (int i = 0; i < A_LD; ++i) lds_dma16_s(x_rs, sa + i * ROWS_PER_PASS * 128, o.a[i], ko);
        }
        const unsigned kb = (unsigned)kt * kt_bytes;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) lds_dma16_s(w_rs, st + b_lds[i], o.b[i], kb);
    };

    // 16x16x32: lane (r = lane & 15, g = lane >> 4) holds 8 operand positions of row / column r of its block; the k order
    // (lane group g: the f32 chunks g and g + 4 of an activation row) and the chunk swizzles are conv_pw_x3_kernel's
    const int r16 = lane & 15, g4 = lane >> 4;
    const int a16_row = wm * 32 * RB + r16;       // (+ 16 i for row block i: (row >> 1) & 7 is the same)
    const unsigned a16_sw = (unsigned)((a16_row >> 1) & 7);
    const unsigned a16_rd0 = (unsigned)(a16_row * 128) + ((((unsigned)g4) ^ a16_sw) << 4);
    const unsigned a16_rd1 = (unsigned)(a16_row * 128) + ((((unsigned)(g4 + 4)) ^ a16_sw) << 4);
    const int n16 = wn * 64 + r16;                // (+ 16 j for column block j: (n >> 2) & 3 is the same)
    const unsigned tau16 = (0x1230u >> (4 * ((n16 >> 2) & 3))) & 3u;           // {0, 3, 2, 1}
    const unsigned b16_rd = (unsigned)(A_STAGE + n16 * 64) + ((((unsigned)g4) ^ tau16) << 4);

    if (p.stamp && t == 0 && blockIdx.x == 0) atomicExch(p.stamp, __builtin_amdgcn_s_memrealtime());
    auto leave = [&]() {
        if (!p.stamp || t != 0) return;
        const unsigned shard = blockIdx.x & 7u;
        const unsigned long long in_shard = (gridDim.x - shard + 7u) / 8u;
        unsigned long long* const sc = p.stamp + 8 * (1 + shard);
        if (atomicAdd(sc, 1ull) != in_shard - 1) return;
        atomicExch(sc, 0ull);
        const unsigned long long shards = gridDim.x < 8u ? gridDim.x : 8u;
        if (atomicAdd(p.stamp + 2, 1ull) != shards - 1) return;
        const unsigned long long d = __builtin_amdgcn_s_memrealtime() - atomicExch(p.stamp, 0ull);
        atomicExch(p.stamp + 2, 0ull);
        atomicAdd(p.stamp + 1, d);
        atomicAdd(p.stamp + 3, 1ull);
        atomicMin(p.stamp + 4, d);
        atomicMax(p.stamp + 5, d);
    };

    int m0, n0, nm0 = 0, nn0 = 0;
    int tile = next_active(blockIdx.x, m0, n0);
    if (tile < 0) { leave(); return; }
    Offs cur = offsets(m0, n0);
    int ntile = next_active(tile + gridDim.x, nm0, nn0);
    Offs nxt = cur;
    if (ntile >= 0) nxt = offsets(nm0, nn0);
    // issue cursor: the next K-tile to be requested is K-tile ic_kt of the current (ic_next = false) or the next output tile
    int ic_kt = 0, ic_stage = 0, ahead = 0;
    bool ic_next = false;
    auto issue_one = [&]() {
        if (!ic_next) {
            issue(cur, ic_kt, ic_stage);
            if (++ic_kt == KT) { ic_next = true; ic_kt = 0; }
        } else {
            if (ntile < 0 || ic_kt >= KT) return;
            issue(nxt, ic_kt, ic_stage);
            ++ic_kt;
        }
        ic_stage = ic_stage + 1 == NST ? 0 : ic_stage + 1;
        ++ahead;
    };
#pragma unroll
    for (int i = 0; i < D; ++i) issue_one();
    int stage = 0;                                  // stage of the K-tile being multiplied

    while (true) {
        f32x4 acc[2 * RB][4];
#pragma unroll
        for (int i = 0; i < 2 * RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bool have_s = false;                        // no scale chosen yet for this output tile
        float a_lim = 0.f;                          // |x| above this leaves the f16 range under the current scale (no scale yet:
                                                    // any non-zero element; an all-zero fragment - padded rows - passes as it is)
        a_s = 1.f; a_inv = 1.f;
        for (int kt = 0; kt < KT; ++kt) {
            // K-tile `kt` has landed: everything (first K-tile of an output tile: the previous epilogue's stores share
            // the counter and return in no fixed order with the loads), or all but the K-tile requested after it
            if (D == 2 && kt > 0 && ahead == 2) x3_wait_vm<PER>(); else x3_wait_vm<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();            // ... for every wave; and every wave is done with the stage before
            --ahead;
            issue_one();
            asm volatile("" ::: "memory");
            const unsigned char* const S = smem_h2 + stage * STAGE;
            float4 alo[2 * RB], ahi[2 * RB];
            f16x8_t bq[4][2];
#pragma unroll
            for (int i = 0; i < 2 * RB; ++i) {
                alo[i] = *reinterpret_cast<const float4*>(S + a16_rd0 + i * (16 * 128));
                ahi[i] = *reinterpret_cast<const float4*>(S + a16_rd1 + i * (16 * 128));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    bq[j][q] = *reinterpret_cast<const f16x8_t*>(S + b16_rd + j * (16 * 64) + q * (BN * 64));
            __builtin_amdgcn_sched_barrier(0);       // every read of the K-tile in flight before the first split
            // a new scale from the maximum of this K-tile's fragment (wave-uniform); the accumulators follow it
            auto adapt = [&]() {
                float m = 0.f;
#pragma unroll
                for (int i = 0; i < 2 * RB; ++i) {
                    m = fmaxf(m, fmaxf(fmaxf(fabsf(alo[i].x), fabsf(alo[i].y)), fmaxf(fabsf(alo[i].z), fabsf(alo[i].w))));
                    m = fmaxf(m, fmaxf(fmaxf(fabsf(ahi[i].x), fabsf(ahi[i].y)), fmaxf(fabsf(ahi[i].z), fabsf(ahi[i].w))));
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
                const unsigned mb = __builtin_amdgcn_readfirstlane(__float_as_uint(m));
                const int e = (int)((mb >> 23) & 0xffu);
                if (e < 27 || e == 255) return;             // all zero (or tiny / non-finite): keep the scale
                const float ns = __uint_as_float((unsigned)(267 - e) << 23);         // 2^(13 - (e - 127))
                const float ratio = ns * a_inv;                                      // new / old, a power of two
                if (have_s) {
#pragma unroll
                    for (int i = 0; i < 2 * RB; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] *= ratio;
                }
                a_s = ns;
                a_inv = __uint_as_float((unsigned)(e - 13) << 23);
                a_lim = 65504.f * a_inv;
                have_s = true;
            };
            {                                        // one look at the raw fragment per K-tile, then straight-line code
                float big = h2_absmax16(alo[0], ahi[0], alo[1], ahi[1]);
                if constexpr (RB == 2) big = fmaxf(big, h2_absmax16(alo[2], ahi[2], alo[3], ahi[3]));
                // (a_lim = 0 until a scale is chosen: the first K-tile with a non-zero element chooses it)
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(big <= a_lim)) != 0ull, 0)) adapt();
            }
#pragma unroll
            for (int i = 0; i < 2 * RB; ++i) {
                const H2Frag a = h2_split(alo[i], ahi[i], a_s);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo, bq[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, bq[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi, bq[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
            }
            stage = stage + 1 == NST ? 0 : stage + 1;
        }
        // the stage of the last K-tile (the one before `stage` in the ring) takes the C tile once every wave has read it
        float* const cbase = reinterpret_cast<float*>(smem_h2 + (stage == 0 ? NST - 1 : stage - 1) * STAGE);
        const int em0 = m0, en0 = n0;
        const float* const winv = p.w_inv + (p.grp_rows ? (size_t)(em0 / p.grp_rows) * p.npad3 : 0);
        // one pass per wave row: PR = 32 RB rows x 128 floats (16 / 32 KB: fits the stage for either row tile)
        constexpr int PR = 32 * RB;
#pragma unroll
        for (int pass = 0; pass < WMW; ++pass) {
            __syncthreads();
            if (wm == pass) {                        // 16x16 C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
                for (int i = 0; i < 2 * RB; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float* cw = cbase + (16 * i + 4 * g4) * BN + wn * 64 + 16 * j + r16;
#pragma unroll
                        for (int r = 0; r < 4; ++r) cw[r * BN] = acc[i][j][r] * a_inv;      // the wave's scale out again
                    }
            }
            __syncthreads();
            constexpr int C4 = BN / 4, RPP = NTHR / C4, NPASS = PR / RPP, SW = NPASS < 4 ? NPASS : 4;
            static_assert(NPASS % SW == 0, "row sweeps in groups of SW");
            const int c4 = t % C4, rr = t / C4;
            const int n = en0 + c4 * 4;
            if (n < p.Cout) {
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + n);
                if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + n);
                const float4 iv = *reinterpret_cast<const float4*>(winv + n);     // the column scales out again (exact)
#pragma unroll
                for (int k0 = 0; k0 < NPASS; k0 += SW) {
                    float4 res[SW];
#pragma unroll
                    for (int k = 0; k < SW; ++k) {
                        const int m = em0 + PR * pass + rr + RPP * (k0 + k);
                        res[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (p.residual && m < M) res[k] = *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.Cout + n);
                    }
#pragma unroll
                    for (int k = 0; k < SW; ++k) {
                        const int row = rr + RPP * (k0 + k);
                        const int m = em0 + PR * pass + row;
                        if (m >= M) continue;
                        float4 v = *reinterpret_cast<const float4*>(cbase + row * BN + c4 * 4);
                        v.x *= iv.x; v.y *= iv.y; v.z *= iv.z; v.w *= iv.w;
                        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                        v.x += res[k].x; v.y += res[k].y; v.z += res[k].z; v.w += res[k].w;
                        if (p.relu) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        }
                        float* const yrow = (IM2COL && m >= q2.M0) ? q2.y1 + (size_t)(m - q2.M0) * p.Cout : p.y + (size_t)m * p.Cout;
                        *reinterpret_cast<float4*>(yrow + n) = v;
                    }
                }
            }
        }
        if (ntile < 0) break;
        // the next output tile becomes the current one (its first K-tiles are requested already: ic_kt of them)
        tile = ntile; m0 = nm0; n0 = nn0;
        cur = nxt;
        ic_next = ic_kt >= KT;                       // (KT == D: the whole new current tile is requested already)
        if (ic_next) ic_kt = 0;
        ntile = next_active(tile + gridDim.x, nm0, nn0);
        if (ntile >= 0) nxt = offsets(nm0, nn0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    leave();
}
