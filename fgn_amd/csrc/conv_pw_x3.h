// conv_pw_x3_kernel - the point-wise / grouped GEMM of conv_pw_persist_kernel with its products on the BF16 matrix pipe.
// Included by conv_igemm.hip (ConvParams, make_rsrc, lds_dma16_s, BK are defined there).
//
// Why.  gfx950 has no xf32 / TF32 form; its f32-input MFMA runs at the VECTOR rate, 64 FLOP / clk / SIMD, 1/16 of the
// bf16 MFMA (MI355X_MICROARCH.md, matrix cores).  conv_pw_persist_kernel holds 0.84-0.87 of that f32 peak on its large
// GEMMs back to back: the remaining head-room of this path is the pipe, not the kernel.  An f32 value is the
// EXACT sum of three bf16 values (8 significant bits each: x1 = x with the low 16 bits cleared, x2 the same of x - x1,
// x3 = x - x1 - x2, each subtraction exact), and the product of two bf16 values is exact in f32 (16 significant bits).
// a * b = sum of nine bf16 products; the three smallest (a2 b3, a3 b2, a3 b3) are below 2^-23 |a b| - the rounding an f32
// FMA chain makes at every step - and are left out: six bf16 MFMAs per f32 MFMA's worth of K, accumulated in f32 by the
// matrix pipe, at 16x the rate: 6/16 of the f32 pipe's time.  Measured against fp64 the result is as close as the f32
// MFMA kernel's (tests/test_hip_conv.py::test_x3_*; NT = 9 takes all nine terms).
//
// Operands.  A: f32 activations exactly as conv_pw_persist_kernel reads them (rows of Cin floats, LDS-DMA, the same
// XOR-swizzled 128-byte tile rows, optional second operand / row table); split into its three bf16 planes in REGISTERS,
// after the ds_read (~5.5 vector instructions per element; a wave's rows are used against 64 output columns, so the
// split hides under the MFMAs of the other waves).  B: the weights, split ONCE at pack time (fgn_amd/ops.py::pack_x3) into an
// image that is the LDS image tile by tile: [group][K-tile][plane][Npad][32] bf16, 64 bytes per row - a K-tile of one
// plane for 128 output columns is 8 KB of contiguous memory, fetched by 8 wave-instructions with no address arithmetic.
// MFMA: v_mfma_f32_16x16x32_bf16 (SH16, the product instances), lane group g = lane >> 4 holding k = 8 consecutive
// operand positions.  ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: the order
// of k inside a K-tile is chosen so that both operands read conflict-free (lane group g takes the f32 chunks g and g + 4 of
// an activation row; the image holds the same k order per chunk and swizzles its chunks with tau[(n >> 2) & 3]).  The
// v_mfma_f32_32x32x16_bf16 form (two 16-deep sub-steps per K-tile, k in order) is kept for the experiments build: equal
// cycles per FLOP, 2-8 % more time on the large GEMMs.
// Tile: BM = 32 * RB * WMW rows x 128 columns, 64 * 2 * WMW threads (WMW waves along M x 2 along N), wave tile 32 RB x 64.
// Product instances: 64 rows (WMW 2, RB 1) and 128 rows (WMW 2, RB 2: 0.078 B / MAC through L2 -> LDS against 0.109),
// both 2 LDS stages and 2 workgroups per CU; fgn_x3_row_tile picks per launch.
// LDS: NST stages x (BM * 128 + 24576) bytes in a ring that runs ACROSS output tiles: K-tile g of the workgroup's
// sequence lives in stage g % NST, the LDS-DMA of K-tile g + NST - 1 (of this output tile or the next one) is issued
// right after the barrier that ends the reads of K-tile g - 1, and is waited for with a counted vmcnt.  The C tile of the
// epilogue (64 rows x 128 floats per pass) goes through the stage the last K-tile was read from.  (Three stages at one
// 512-thread workgroup per CU - more bytes in flight - measured 4-25 % slower: its waves run their phases in lockstep.)
// The kernel is POWER-bound (1.5-1.8 GHz against 2.4 for the f32 kernels): instances with 14 % fewer or 13 % more cycles
// per MAC take the same time; what separates them is energy per MAC (tools/x3_probe.py --phases).
#pragma once

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

struct X3Frag { bf16x8_t p1, p2, p3; };

// eight f32 -> three planes of eight bf16 (truncating split: every plane has the sign of x, x = p1 + p2 + p3 exactly)
__device__ __forceinline__ X3Frag x3_split(const float4& lo, const float4& hi) {
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned q1[4], q2[4], q3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned u0 = __float_as_uint(x[2 * i]), u1 = __float_as_uint(x[2 * i + 1]);
        q1[i] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);                       // hi16(u0) | hi16(u1) << 16
        const float r0 = x[2 * i] - __uint_as_float(u0 & 0xffff0000u);
        const float r1 = x[2 * i + 1] - __uint_as_float(u1 & 0xffff0000u);
        const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
        q2[i] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
        const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u);
        const float s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
        q3[i] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
    X3Frag f;
    f.p1 = __builtin_bit_cast(bf16x8_t, make_uint4(q1[0], q1[1], q1[2], q1[3]));
    f.p2 = __builtin_bit_cast(bf16x8_t, make_uint4(q2[0], q2[1], q2[2], q2[3]));
    f.p3 = __builtin_bit_cast(bf16x8_t, make_uint4(q3[0], q3[1], q3[2], q3[3]));
    return f;
}

constexpr int X3_BN = 128;
constexpr int X3_B_STAGE = 3 * X3_BN * 64;       // bytes of one K-tile of the weight image for 128 columns

// phase clocks of one wave (tools/x3_probe.py --phases; -DX3_PHASES): cycles in [wait + barrier | issue | LDS reads
// landed | split + MFMA | epilogue], written by wave 0 of workgroups 0 and 1 to p.ws
#ifdef X3_PHASES
#define X3_T(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph_[i] += now_ - last_; last_ = now_; } while (0)
#else
#define X3_T(i) do { } while (0)
#endif

template <int N> __device__ __forceinline__ void x3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int WMW, int RB, int NT, int NST, bool SH16>
__global__ __launch_bounds__(128 * WMW, 2) void conv_pw_x3_kernel(const ConvParams p, const int total_tiles) {
    // WMW waves along M x 2 along N; a wave's tile is 32 RB rows x 64 columns: (SH16) 2 RB x 4 blocks of
    // v_mfma_f32_16x16x32_bf16 (one MFMA spans the K-tile), or RB x 2 blocks of v_mfma_f32_32x32x16_bf16 (two 16-deep
    // sub-steps per K-tile)
    constexpr int BM = 32 * RB * WMW, BN = X3_BN, NTHR = 128 * WMW, NW = 2 * WMW;
    constexpr int A_LD = BM / 8 / NW;                       // activation wave-instructions per wave per K-tile (8 rows each)
    constexpr int A_STAGE = BM * 128;                       // bytes
    constexpr int STAGE = A_STAGE + X3_B_STAGE;             // bytes
    constexpr int B_LD = 24 / NW;                           // weight wave-instructions per wave per K-tile
    constexpr int PER = A_LD + B_LD;                        // LDS-DMA wave-instructions per wave per K-tile
    constexpr int D = NST - 1;                              // K-tiles in flight ahead of the one being multiplied
    constexpr int ROWS_PER_PASS = NTHR / 8;                 // A rows one pass of the workgroup's DMAs covers
    static_assert(NST == 2 || NST == 3, "ring of 2 or 3 stages");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x3[];

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int M = (p.n_img_dev ? min(p.n_img, *p.n_img_dev) : p.n_img) * p.Ho * p.Wo;
    int grp_valid = p.grp_valid;
    if (p.grp_rows && p.grp_count_dev) grp_valid = min(grp_valid, min(p.grp_items, *p.grp_count_dev) * p.grp_rows_per_item);

    const int col4 = t & 7, row0 = t >> 3;
    const int src_c4 = col4 ^ ((row0 >> 1) & 7);
    const i32x4 x_rs = make_rsrc(p.x, p.x_bytes);
    const i32x4 w_rs = make_rsrc(p.w3, p.w3_bytes);
    const bool dual = p.x2 != nullptr;
    const i32x4 x2_rs = make_rsrc(dual ? p.x2 : p.x, dual ? p.x2_bytes : p.x_bytes);
    const int KT = p.K / BK;                                                  // >= D (checked by the launcher)
    const unsigned kt_bytes = (unsigned)(3 * p.npad3 * 64);          // one K-tile of the image, all planes, all rows
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<size_t>(smem_x3));
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

    struct Offs { unsigned a[A_LD], a2[A_LD], b[B_LD]; };
    unsigned b_lds[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const int q = wv + NW * i;                     // wave-instruction q of 24: plane q / 8, rows 16 * (q % 8) ..
        b_lds[i] = __builtin_amdgcn_readfirstlane((unsigned)(A_STAGE + (q >> 3) * (BN * 64) + (q & 7) * 1024));
    }
    auto offsets = [&](int m0, int n0) -> Offs {
        Offs o;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int m = m0 + row0 + ROWS_PER_PASS * i;
            o.a[i] = m < M ? (unsigned)((m * p.Cin + src_c4 * 4) * 4) : OOB;
            o.a2[i] = OOB;
            if (dual && m < M) o.a2[i] = (unsigned)(((p.x2_rows ? p.x2_rows[m] : m) * p.cin2 + src_c4 * 4) * 4);
        }
        unsigned g0 = 0;
        if (p.grp_rows) g0 = (unsigned)(m0 / p.grp_rows) * (unsigned)KT * kt_bytes;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int q = wv + NW * i;
            const int plane = q >> 3, row = (q & 7) * 16 + (lane >> 2);
            o.b[i] = g0 + (unsigned)(((plane * p.npad3 + n0 + row) * 4 + (lane & 3)) * 16);
        }
        return o;
    };
    auto issue = [&](const Offs& o, int kt, int stage) {
        const unsigned st = lds_base + stage * STAGE;
        const unsigned sa = st + wave_row_bytes;
        const unsigned ko = (unsigned)(kt * BK * 4);
        if (dual && kt >= p.kt1) {
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

    // fragment addresses (bytes within a stage).  32x32x16: lane (r = lane & 31, h = lane >> 5) holds k = 8h .. 8h + 7
    const int r32 = lane & 31, h = lane >> 5;
    const int a_row = wm * 32 * RB + r32;         // (+ 32 i for row block i: the swizzle term (row >> 1) & 7 is the same)
    const unsigned a_sw = (unsigned)((a_row >> 1) & 7);
    unsigned a_rd[2][2], b_rd[2][2];              // [sub-step][chunk] / [sub-step][column block], swizzle applied
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const unsigned ca = (unsigned)(4 * s + 2 * h);
        a_rd[s][0] = (unsigned)(a_row * 128) + ((ca ^ a_sw) << 4);
        a_rd[s][1] = (unsigned)(a_row * 128) + (((ca + 1) ^ a_sw) << 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = wn * 64 + 32 * j + r32;
            b_rd[s][j] = (unsigned)(A_STAGE + n * 64) + ((((unsigned)(2 * s + h)) ^ (unsigned)((n >> 2) & 3)) << 4);
        }
    }

    // 16x16x32: lane (r = lane & 15, g = lane >> 4) holds k = 8g .. 8g + 7 of row / column r of its block
    const int r16 = lane & 15, g4 = lane >> 4;
    const int a16_row = wm * 32 * RB + r16;       // (+ 16 i for row block i: (row >> 1) & 7 is the same)
    const unsigned a16_sw = (unsigned)((a16_row >> 1) & 7);
    // ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): a group
    // holds rows 0-3 / 12-15 of lane group g and rows 4-11 of lane group g ^ 1.  With lane group g on the 16-byte chunks
    // 2g, 2g+1 of its row those collide two-way (measured: SQ_LDS_BANK_CONFLICT = half of the LDS cycles).  The order of k
    // inside a K-tile is free as long as A and B agree: lane group g takes the chunks g and g + 4 (k = 4g..4g+3, 16+4g..),
    // the weight image of this form (ops.pack_x3(sh16=True)) holds the same k order in its chunk g and swizzles its
    // chunks with tau[(n >> 2) & 3], tau = {0, 3, 2, 1}: conflict-free for both operands.
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
#ifdef X3_PHASES
    unsigned long long ph_[6] = {0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
    const unsigned long long begin_ = last_;
#endif

    while (true) {
        f32x16 acc[RB][2];
        f32x4 acc16[2 * RB][4];
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int i = 0; i < 2 * RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt) {
            // K-tile `kt` has landed: everything (first K-tile of an output tile: the previous epilogue's stores share
            // the counter and return in no fixed order with the loads), or all but the K-tile requested after it
            if (D == 2 && kt > 0 && ahead == 2) x3_wait_vm<PER>(); else x3_wait_vm<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();            // ... for every wave; and every wave is done with the stage before
            X3_T(0);
            --ahead;
            issue_one();
            asm volatile("" ::: "memory");
            X3_T(1);
            const unsigned char* const S = smem_x3 + stage * STAGE;
            if constexpr (SH16) {
                float4 alo[2 * RB], ahi[2 * RB];
                bf16x8_t bq[4][3];
#pragma unroll
                for (int i = 0; i < 2 * RB; ++i) {
                    alo[i] = *reinterpret_cast<const float4*>(S + a16_rd0 + i * (16 * 128));
                    ahi[i] = *reinterpret_cast<const float4*>(S + a16_rd1 + i * (16 * 128));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        bq[j][q] = *reinterpret_cast<const bf16x8_t*>(S + b16_rd + j * (16 * 64) + q * (BN * 64));
                __builtin_amdgcn_sched_barrier(0);
#ifdef X3_PHASES
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                X3_T(2);
#endif
#pragma unroll
                for (int i = 0; i < 2 * RB; ++i) {
                    const X3Frag a = x3_split(alo[i], ahi[i]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 c = acc16[i][j];
                        if (NT == 9) {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p3, bq[j][2], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p3, bq[j][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p2, bq[j][2], c, 0, 0, 0);
                        }
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p3, bq[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p1, bq[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p2, bq[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p2, bq[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p1, bq[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.p1, bq[j][0], c, 0, 0, 0);
                        acc16[i][j] = c;
                    }
                }
            } else {
            float4 alo[2][RB], ahi[2][RB];
            bf16x8_t bq[2][2][3];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    alo[s][i] = *reinterpret_cast<const float4*>(S + a_rd[s][0] + i * (32 * 128));
                    ahi[s][i] = *reinterpret_cast<const float4*>(S + a_rd[s][1] + i * (32 * 128));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        bq[s][j][q] = *reinterpret_cast<const bf16x8_t*>(S + b_rd[s][j] + q * (BN * 64));
            }
            __builtin_amdgcn_sched_barrier(0);       // all 16 reads of the K-tile in flight before the first split
#ifdef X3_PHASES
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            X3_T(2);
#endif
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const X3Frag a = x3_split(alo[s][i], ahi[s][i]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x16 c = acc[i][j];
                        if (NT == 9) {
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p3, bq[s][j][2], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p3, bq[s][j][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p2, bq[s][j][2], c, 0, 0, 0);
                        }
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p3, bq[s][j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, bq[s][j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p2, bq[s][j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p2, bq[s][j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, bq[s][j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p1, bq[s][j][0], c, 0, 0, 0);
                        acc[i][j] = c;
                    }
                }
            }
            }
            stage = stage + 1 == NST ? 0 : stage + 1;
#ifdef X3_PHASES
            if constexpr (SH16) asm volatile("s_nop 0" ::"v"(acc16[0][0][0]), "v"(acc16[2 * RB - 1][3][3]));
            else asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[RB - 1][1][15]));
            X3_T(3);
            ++ph_[5];
#endif
        }
        // the stage of the last K-tile (the one before `stage` in the ring) takes the C tile once every wave has read it
        float* const cbase = reinterpret_cast<float*>(smem_x3 + (stage == 0 ? NST - 1 : stage - 1) * STAGE);
        const int em0 = m0, en0 = n0;
        // 32x32 C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); 64 rows per pass
#pragma unroll
        for (int pass = 0; pass < BM / 64; ++pass) {
            __syncthreads();
            if ((wm * RB) / 2 == pass) {
                if constexpr (SH16) {            // 16x16 C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
                    for (int i = 0; i < 2 * RB; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float* cw = cbase + (((wm * RB) & 1) * 32 + 16 * i + 4 * g4) * BN + wn * 64 + 16 * j + r16;
#pragma unroll
                            for (int r = 0; r < 4; ++r) cw[r * BN] = acc16[i][j][r];
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < RB; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            float* cw = cbase + ((((wm * RB) & 1) + i) * 32 + 4 * h) * BN + wn * 64 + 32 * j + r32;
#pragma unroll
                            for (int r = 0; r < 16; ++r) cw[((r & 3) + 8 * (r >> 2)) * BN] = acc[i][j][r];
                        }
                }
            }
            __syncthreads();
            constexpr int C4 = BN / 4, RPP = NTHR / C4, NPASS = 64 / RPP;
            const int c4 = t % C4, rr = t / C4;
            const int n = en0 + c4 * 4;
            if (n < p.Cout) {
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + n);
                if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + n);
#pragma unroll
                for (int k0 = 0; k0 < NPASS; k0 += 4) {
                    float4 res[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int m = em0 + 64 * pass + rr + RPP * (k0 + k);
                        res[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (p.residual && m < M) res[k] = *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.Cout + n);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int row = rr + RPP * (k0 + k);
                        const int m = em0 + 64 * pass + row;
                        if (m >= M) continue;
                        float4 v = *reinterpret_cast<const float4*>(cbase + row * BN + c4 * 4);
                        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                        v.x += res[k].x; v.y += res[k].y; v.z += res[k].z; v.w += res[k].w;
                        if (p.relu) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        }
                        *reinterpret_cast<float4*>(p.y + (size_t)m * p.Cout + n) = v;
                    }
                }
            }
        }
        X3_T(4);
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
#ifdef X3_PHASES
    if (p.ws && lane == 0 && wv == 0 && blockIdx.x < 2) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ws) + 8 * blockIdx.x;
        for (int i = 0; i < 6; ++i) o[i] = ph_[i];
        o[6] = __builtin_amdgcn_s_memtime() - begin_;
    }
#endif
    leave();
}
