// NHWC fp32 implicit-GEMM convolution on the CDNA4 fp32 matrix pipe.
//
// Replaces the cuDNN conv + BN + ReLU calls the reference reaches through
// mmdet's ResNet / RPNHead / ResLayer / FCNMaskHead (call sites
// fgn.py:212-215, fgn_ag_rpn_head.py:44-48, fgn_roi_head.py:236,369,380).
//
//   GEMM view :  C[M,N] = A[M,K] * B[K,N]
//     M = n_img*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin
//     A = im2col of the NHWC activation (gathered on the fly, never materialised)
//     B = weights packed [CoutPad][KH][KW][Cin]  (K contiguous per output channel)
//   MFMA      :  v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 64 FLOP/clk/SIMD).
//     Lane l feeds A[row l&31][k = l>>5]; we give lane-half h the 4 consecutive
//     k's  8t+4h .. 8t+4h+3 with one ds_read_b128 and issue 4 MFMAs from it, so
//     MFMA j contracts k in {8t+j, 8t+4+j}: a permutation of the K order that is
//     applied identically to A and B.
//   Loader    :  LDS-DMA (`buffer_load_dwordx4 ... lds`, 16 B per lane straight into LDS) in every kernel but
//     conv_igemm_kernel; tile rows are the unpadded 128 B, bank conflicts are removed by XOR-swizzling the
//     source chunk and the fragment read alike.  Two LDS stages, one barrier per K-tile.
//   Epilogue  :  y = acc*scale[n] + shift[n] (+ residual) (ReLU)  -- folded eval-mode BN or conv bias.
//     The accumulators take a round trip through LDS so that every lane moves 16 B of one output row.
//   Fusions   :  optional per-(image, cin) input scale applied while staging A (register-staged
//     kernel only): the mask-head support-vector multiply (fgn_roi_head.py:379); `a_img_div` lets several
//     output images read one input image (AG-RPN guidance, fgn_ag_rpn_head.py:44, when the direct form is
//     used; the Winograd form applies it in its input transform).
//   Kernels   :  conv_pw_persist_kernel (point-wise launches with more output tiles than resident workgroups: the
//     dominant kernel of an episode), conv_igemm_dma_kernel (64x64 / 128x128 tile, optional split-K; modes generic /
//     point-wise + grouped Winograd GEMM / stem), conv_igemm_dma_pair_kernel (one layer on two tensors),
//     conv_igemm_kernel (register-staged loader, [rows][32 + 4 pad] LDS layout, for the fused input scale when a
//     layer does not take the Winograd form); the dispatcher at the bottom of the file picks one per launch.
//   Everything here is on the default path.  The kernels that were built, measured and NOT adopted (Stream-K, the
//   generalised persistent kernel in eight tile shapes, their tuning entry point) live in
//   tools/micro/conv_pw_experiments.inc and are compiled only with -DFGN_EXPERIMENTS (tools/micro/build_experiments.sh).
#include "common.h"
#include <cstdlib>

struct ConvParams {
    const float* x;
    const float* w;
    float* y;
    const float* scale;
    const float* shift;
    const float* residual;
    const float* in_scale;
    const int32_t* n_img_dev;
    int n_img, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int a_img_div;
    int relu;
    int K;          // padded reduction length (multiple of 32)
    int n_tiles_n;  // Cout tiles
    // banded tile raster (LDS-DMA kernel): when the weight matrix of a launch (Cout x K) does not fit an XCD's L2
    // next to the activations, the plain order (all Cout tiles of one row tile, then the next row tile) re-streams
    // it from the Infinity Cache for every row tile (measured: 1.53 GB fetched by the AG-RPN Winograd GEMM against
    // 0.28 GB of operands).  Tiles are therefore walked band by band: `band_nt` Cout tiles (<= ~2 MB of weights)
    // x all `band_mt` row tiles of the group (grouped GEMM) or launch, then the next band.  0 = plain order.
    int band_nt, band_mt;
    // split-K: blockIdx.y owns K-tiles [y*kt_per_split, (y+1)*kt_per_split) and writes its raw
    // partial tile to slab y of `ws` ([splits][n_img*Ho*Wo][Cout]); splitk_epilogue_kernel sums
    // the slabs in a fixed order (deterministic) and applies the epilogue.
    float* ws;
    int splits, kt_per_split;
#ifdef FGN_EXPERIMENTS      // tools/micro/conv_pw_experiments.inc (not part of libfgn_hip.so)
    int32_t* tickets;        // Stream-K: one zero-initialised word per remaining tile
    int32_t* sched;          // tile scheduler of conv_pw_persist2_kernel: zero-initialised counters
    int sk_U, sk_dp;         // Stream-K launch: tiles [0, sk_dp) whole, the rest in ranges of sk_U K-tiles
#endif
    unsigned x_bytes, w_bytes;   // extents for the buffer descriptors of the LDS-DMA kernel
    // grouped GEMM (Winograd; 64x64 kernel, point-wise mode): rows [g*grp_rows, (g+1)*grp_rows) use the weight
    // matrix at w + g*grp_w_stride floats; within a group only the first `valid` rows are computed, valid =
    // grp_valid, or min(grp_items, *grp_count_dev) * grp_rows_per_item when the item count lives on the
    // device.  grp_rows == 0: plain convolution.
    int grp_rows, grp_valid, grp_items, grp_rows_per_item, grp_w_stride;
    const int32_t* grp_count_dev;
    // launch record (conv_pw_persist_kernel; fgn_profile_stamps): FGN_STAMP_WORDS x uint64 in device memory or nullptr
    unsigned long long* stamp;
    // second A operand of conv_pw_persist_kernel (fgn_conv1x1_dual_nhwc_f32): the K-tiles from kt1 on are read from x2
    // (rows of cin2 floats) instead of x (rows of Cin floats) - two 1x1 convolutions on the same pixels summed in one
    // K loop (a bottleneck's conv3 and the 1x1 / stride 1 shortcut of its stage's first block).  nullptr: one operand.
    const float* x2;
    unsigned x2_bytes;
    int kt1, cin2;
    const int32_t* x2_rows;      // optional: row m of the output reads row x2_rows[m] of x2 (a strided shortcut); nullptr: row m
    // conv_pw_x3_kernel (conv_pw_x3.h): the weights as three bf16 planes, [group][K-tile][plane][npad3][32], or nullptr
    const void* w3 = nullptr;
    unsigned w3_bytes = 0;
    int npad3 = 0;
    // conv_pw_h2_kernel (conv_pw_h2.h): w3 holds two f16 planes of the column-scaled weights, w_inv the inverse column
    // scales [group][npad3]
    const float* w_inv = nullptr;
};

#ifndef CONV_DMA_STAGES
#define CONV_DMA_STAGES 2
#endif
constexpr int BK = 32;
constexpr int LDS_STRIDE = 36;  // floats

// Diagnostic build only (-DCONV_CLOCK_STAMPS, tools/micro/gemm_clock.hip; the product library never defines it): one
// workgroup-level pair of (s_memtime = shader cycles, s_memrealtime = 100 MHz) stamps around a kernel's work, written
// to a buffer of their own that nothing else reads - the in-kernel clock of MI355X_MICROARCH.md "DVFS give-back" (6).
#ifdef CONV_CLOCK_STAMPS
__device__ unsigned long long g_clock_stamps[16384 * 6];
#define CLOCK_STAMP_BEGIN()                                                \
    const unsigned long long cs_t0 = __builtin_amdgcn_s_memtime();         \
    const unsigned long long cs_r0 = __builtin_amdgcn_s_memrealtime();     \
    __builtin_amdgcn_s_waitcnt(0xC07F)
#define CLOCK_STAMP_END(slot)                                              \
    do {                                                                   \
        const unsigned long long cs_t1 = __builtin_amdgcn_s_memtime();     \
        const unsigned long long cs_r1 = __builtin_amdgcn_s_memrealtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                \
        unsigned cs_xcc, cs_hw;                                            \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(cs_xcc)); \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(cs_hw));   \
        if (threadIdx.x == 0 && (slot) < 16384) {                          \
            g_clock_stamps[(slot) * 6 + 0] = cs_t0;                        \
            g_clock_stamps[(slot) * 6 + 1] = cs_t1;                        \
            g_clock_stamps[(slot) * 6 + 2] = cs_r0;                        \
            g_clock_stamps[(slot) * 6 + 3] = cs_r1;                        \
            g_clock_stamps[(slot) * 6 + 4] = cs_xcc;                       \
            g_clock_stamps[(slot) * 6 + 5] = cs_hw;                        \
        }                                                                  \
    } while (0)
#else
#define CLOCK_STAMP_BEGIN() do { } while (0)
#define CLOCK_STAMP_END(slot) do { } while (0)
#endif

// Per-thread staging state, fixed for the whole K loop: for each A row this thread loads, the
// element offset of the (ky=0,kx=0) tap and a bit mask of the filter taps that fall inside the
// image.  Per K-tile the address is then `off + uniform tap offset` (one VALU add) and the
// bounds test is one bit test - the im2col index arithmetic is out of the hot loop.
template <int A_LD>
struct AStage {
    int off[A_LD];                    // ((img*H + iy0)*W + ix0)*Cin + col4*4
    unsigned long long taps[A_LD];    // bit (ky*KW+kx): tap inside the image and row < M
    int soff[A_LD];                   // in_scale row offset (img*Cin + col4*4)
};

// Register staging buffers are ext_vector SSA values (not arrays): hipcc leaves float4 arrays
// that cross a scheduling fence in scratch memory, which serialises every load.
template <int N>
struct Pack {
    typedef float type __attribute__((ext_vector_type(N)));
};
#define PACK_SET4(pk, i, v)      \
    do {                         \
        (pk)[4 * (i) + 0] = (v).x; \
        (pk)[4 * (i) + 1] = (v).y; \
        (pk)[4 * (i) + 2] = (v).z; \
        (pk)[4 * (i) + 3] = (v).w; \
    } while (0)
#define PACK_GET4(pk, i) make_float4((pk)[4 * (i)], (pk)[4 * (i) + 1], (pk)[4 * (i) + 2], (pk)[4 * (i) + 3])

// Issue the global loads of one K-tile.  Nothing here consumes a loaded value: zero-fill of
// out-of-image taps and the input-scale multiply happen in finish_tile(), after the MFMA block,
// so the compiler places its s_waitcnt there and the loads fly under the MFMAs.
template <int A_LD, int B_LD, bool IN_SCALE>
__device__ __forceinline__ unsigned load_tile(const ConvParams& p, const AStage<A_LD>& st, const float* b_base,
                                              int kt, int cin_tiles, int col4,
                                              typename Pack<4 * A_LD>::type& a_reg,
                                              typename Pack<4 * A_LD>::type& s_reg,
                                              typename Pack<4 * B_LD>::type& b_reg) {
    unsigned ok_mask = 0;
    const int tap = kt / cin_tiles;                       // wave-uniform (scalar unit)
    const int c0 = (kt - tap * cin_tiles) * BK;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int tap_off = (ky * p.W + kx) * p.Cin + c0;     // uniform
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const bool ok = (st.taps[i] >> tap) & 1ull;
        const int off = ok ? st.off[i] + tap_off : 0;     // invalid taps read a valid dummy address
        const float4 v = *reinterpret_cast<const float4*>(p.x + off);
        PACK_SET4(a_reg, i, v);
        if (IN_SCALE) {
            const float4 sv = *reinterpret_cast<const float4*>(p.in_scale + st.soff[i] + c0);
            PACK_SET4(s_reg, i, sv);
        }
        ok_mask |= ok ? (1u << i) : 0u;
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(b_base + (size_t)(32 * i) * p.K + kt * BK);
        PACK_SET4(b_reg, i, v);
    }
    return ok_mask;
}

template <int A_LD, bool IN_SCALE>
__device__ __forceinline__ void finish_tile(unsigned ok_mask, typename Pack<4 * A_LD>::type& a_reg,
                                            const typename Pack<4 * A_LD>::type& s_reg) {
    if (IN_SCALE) a_reg *= s_reg;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        if (!(ok_mask & (1u << i))) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            PACK_SET4(a_reg, i, z);
        }
    }
}

template <int BM, int BN, int WM, int WN, bool IN_SCALE, int MIN_WAVES>
__global__ __launch_bounds__(256, MIN_WAVES) void conv_igemm_kernel(const ConvParams p) {
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_LD = BM * 8 / 256;  // float4 loads per thread per K-tile
    constexpr int B_LD = BN * 8 / 256;
    constexpr int STAGE = (BM + BN) * LDS_STRIDE;

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = t >> 6;
    const int wm = wv / WAVES_N, wn = wv % WAVES_N;

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so
    // give every XCD a contiguous run of logical tiles; consecutive logical tiles
    // share the same A rows (n fastest), which then hit in that XCD's L2.
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid / p.n_tiles_n;
    const int tile_n = bid - tile_m * p.n_tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const int HoWo = p.Ho * p.Wo;
    int n_img = p.n_img;
    if (p.n_img_dev) n_img = min(n_img, *p.n_img_dev);
    const int M = n_img * HoWo;
    if (m0 >= M) return;

    // ---- per-thread staging coordinates (fixed across the K loop) ---------------
    const int col4 = t & 7;    // which float4 of the 32-float K-tile row
    const int row0 = t >> 3;   // 0..31, rows row0 + 32*i
    AStage<A_LD> st;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m0 + row0 + 32 * i;
        st.off[i] = 0; st.taps[i] = 0ull; st.soff[i] = 0;
        if (m < M) {
            const int img = m / HoWo;
            const int rem = m - img * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            st.off[i] = (((img / p.a_img_div) * p.H + iy0) * p.W + ix0) * p.Cin + col4 * 4;
            st.soff[i] = img * p.Cin + col4 * 4;
            unsigned long long tm = 0ull;
            int tp = 0;
            for (int ky = 0; ky < p.KH; ++ky) {
                const bool y_ok = (unsigned)(iy0 + ky) < (unsigned)p.H;
                for (int kx = 0; kx < p.KW; ++kx, ++tp)
                    if (y_ok && (unsigned)(ix0 + kx) < (unsigned)p.W) tm |= 1ull << tp;
            }
            st.taps[i] = tm;
        }
    }
    const float* b_base = p.w + (size_t)(n0 + row0) * p.K + col4 * 4;

    const int KT_all = p.K / BK;
    const int kt0 = blockIdx.y * p.kt_per_split;
    const int KT = min(KT_all, kt0 + p.kt_per_split);     // this block's K-tiles: [kt0, KT)
    const int cin_tiles = p.Cin / BK;

    typename Pack<4 * A_LD>::type a_reg, s_reg;
    typename Pack<4 * B_LD>::type b_reg;
    s_reg = 0.f;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;
    float* const st_a = smem + row0 * LDS_STRIDE + col4 * 4;
    float* const st_b = st_a + BM * LDS_STRIDE;
    const float* const rd_a = smem + (wm * WM + frag_row) * LDS_STRIDE + frag_k;
    const float* const rd_b = smem + BM * LDS_STRIDE + (wn * WN + frag_row) * LDS_STRIDE + frag_k;

    // ---- prologue: tile kt0 -> LDS[0]; tile kt0+1 -> registers -------------------------------
    unsigned okm = load_tile<A_LD, B_LD, IN_SCALE>(p, st, b_base, kt0, cin_tiles, col4, a_reg, s_reg, b_reg);
    finish_tile<A_LD, IN_SCALE>(okm, a_reg, s_reg);
#pragma unroll
    for (int i = 0; i < A_LD; ++i) *reinterpret_cast<float4*>(st_a + 32 * i * LDS_STRIDE) = PACK_GET4(a_reg, i);
#pragma unroll
    for (int i = 0; i < B_LD; ++i) *reinterpret_cast<float4*>(st_b + 32 * i * LDS_STRIDE) = PACK_GET4(b_reg, i);
    okm = load_tile<A_LD, B_LD, IN_SCALE>(p, st, b_base, min(kt0 + 1, KT - 1), cin_tiles, col4, a_reg, s_reg,
                                                b_reg);
    __syncthreads();

    // ---- main loop, three-stage software pipeline ------------------------------------------
    //   registers hold tile t+1 (loaded one iteration ago), LDS[cur] holds tile t.
    //   [MFMAs of k-slice 0] -> [regs(t+1) -> LDS[cur^1]; issue global loads of tile t+2]
    //   -> [MFMAs of k-slices 1..3] -> barrier.
    // The LDS stores and the global loads sit in the shadow of 3/4 of the tile's MFMAs, so the
    // only pipe bubble per K-tile is the barrier plus the first ds_read of the next tile.
    int cur = 0;
    for (int kt = kt0; kt < KT; ++kt) {
        const float* As = rd_a + cur * STAGE;
        const float* Bs = rd_b + cur * STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const float4*>(As + i * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const float4*>(Bs + j * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
            if (kk == 0) {
                // stage tile t+1 (in registers since the previous iteration) into the other buffer,
                // then issue the loads of tile t+2.  Fences pin this block between k-slice 0 and 1.
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" : "+v"(okm));   // consumers of the loaded registers stay below
                finish_tile<A_LD, IN_SCALE>(okm, a_reg, s_reg);
                float* sa = st_a + (cur ^ 1) * STAGE;
                float* sb = st_b + (cur ^ 1) * STAGE;
#pragma unroll
                for (int i = 0; i < A_LD; ++i)
                    *reinterpret_cast<float4*>(sa + 32 * i * LDS_STRIDE) = PACK_GET4(a_reg, i);
#pragma unroll
                for (int i = 0; i < B_LD; ++i)
                    *reinterpret_cast<float4*>(sb + 32 * i * LDS_STRIDE) = PACK_GET4(b_reg, i);
                asm volatile("" ::: "memory");
                okm = load_tile<A_LD, B_LD, IN_SCALE>(p, st, b_base, min(kt + 2, KT - 1), cin_tiles, col4, a_reg,
                                                      s_reg, b_reg);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: C/D layout col = lane&31 (-> n), row = (r&3)+8*(r>>2)+4*(lane>>5) (-> m)
    const int half = lane >> 5;
    if (p.splits > 1) {
        float* slab = p.ws + (size_t)blockIdx.y * ((size_t)p.n_img * HoWo) * p.Cout;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + frag_row;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + wm * WM + i * 32 + 4 * half;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (n < p.Cout && m < M) slab[(size_t)m * p.Cout + n] = acc[i][j][r];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + frag_row;
        const bool n_ok = n < p.Cout;
        const float sc = (n_ok && p.scale) ? p.scale[n] : 1.f;
        const float sh = (n_ok && p.shift) ? p.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * WM + i * 32 + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (n_ok && m < M) {
                    const size_t o = (size_t)m * p.Cout + n;
                    float v = acc[i][j][r] * sc + sh;
                    if (p.residual) v += p.residual[o];
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.y[o] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (no input scale, Cin >= 32): the K-tiles go global -> LDS directly with
// `buffer_load_dwordx4 ... lds` (16 B per lane, 1 KiB per wave-instruction), no VGPR staging,
// no ds_write, no zero-fill selects:
//   * out-of-image filter taps use an out-of-range buffer offset, for which the hardware
//     writes zeros into LDS (probed on gfx950: tools_scratch/t_ldsdma.hip);
//   * an LDS-DMA destination is lane-linear (wave base + lane*16), so tile rows are the
//     unpadded 128 bytes and bank conflicts are removed by an XOR swizzle applied to the
//     SOURCE chunk index and to the ds_read address alike: chunk c of row r lives at
//     c ^ ((r>>1)&7); 16 consecutive rows at one logical chunk then cover all 16 four-bank
//     groups of the 64-bank LDS;
//   * NSTAGE LDS buffers: the DMA of tile t+NSTAGE-1 is in flight while tile t is multiplied;
//     completion is tracked with a counted s_waitcnt vmcnt and a raw s_barrier.
// ------------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));

// One LDS-DMA wave-instruction: 64 lanes x 16 B from buffer `rs` at per-lane byte offset `voff`
// to LDS bytes [lds_base, lds_base + 1024).  Issued through inline asm on purpose: for the
// builtin form hipcc conservatively emits `s_waitcnt vmcnt(0)` before the next ds_read of the
// same __shared__ array (it cannot prove the buffers distinct), which would serialise the
// pipeline; asm loads are invisible to its counters, so completion is waited for by hand
// (counted s_waitcnt vmcnt before the barrier).  M0 is saved/restored inside the statement.
__device__ __forceinline__ void lds_dma16(const i32x4& rs, unsigned lds_base, unsigned voff) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(lds_base), "s"(rs)
        : "memory");
}
// The same with the K offset in the instruction's scalar offset and M0 left as written (no save / restore: nothing else
// in these kernels reads M0): per K-tile no vector arithmetic at all - `voff` is the tile's row offset (or the
// out-of-range constant: the range check looks at voff alone) and `soff` the byte offset of the K-tile.
__device__ __forceinline__ void lds_dma16_s(const i32x4& rs, unsigned lds_base, unsigned voff, unsigned soff) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %0, %2, %3 offen lds"
        :
        : "v"(voff), "s"(lds_base), "s"(rs), "s"(soff)
        : "memory", "m0");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));   // stride 0
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// MODE 1 = point-wise fast path (1x1, stride 1, no padding, a_img_div 1): im2col is the identity, so
// the per-row index decode (integer divisions) and the tap masks are compiled out.  These layers
// have K of 64..1024, i.e. 2..32 K-tiles, and are otherwise dominated by prologue instructions.
// MODE 2 = stem: Cin = 4 (NHWC4 image), KW <= 8.  A K-tile is one filter ROW: the 8 consecutive pixels
// ix0..ix0+7 of input row iy0+ky are 128 contiguous bytes, i.e. exactly one LDS-DMA tile row; weights are packed
// [Cout][KH][8][4] with zeros beyond KW (K = 32*KH).  Lane chunk = pixel: validity is per lane (x) and per
// K-tile (y).
// (the body: workgroup (bx, by) of a launch of nwg_in x splits workgroups - the kernels below pass their own block
// index and grid size, or the position of the workgroup inside ITS half of a two-convolution launch)
template <int BM, int BN, int WM, int WN, int NSTAGE, int MODE>
__device__ __forceinline__ void conv_igemm_dma_body(const ConvParams& p, const int bx, const int nwg_in, const int by) {
    constexpr bool PW = MODE == 1, STEM = MODE == 2;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_LD = BM * 8 / 256;
    constexpr int B_LD = BN * 8 / 256;
    constexpr int STAGE = (BM + BN) * BK;          // floats per stage, rows are 32 floats (128 B)
    constexpr int LOADS = A_LD + B_LD;
    constexpr unsigned OOB = 0x7ffffff0u;

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wv = t >> 6;
    const int wm = wv / WAVES_N, wn = wv % WAVES_N;

    const int nwg = nwg_in;
    int bid = bx;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
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
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const int HoWo = p.Ho * p.Wo;
    int n_img = p.n_img;
    if (p.n_img_dev) n_img = min(n_img, *p.n_img_dev);
    const int M = n_img * HoWo;
    if (m0 >= M) return;

    const int col4 = t & 7;
    const int row0 = t >> 3;
    const int src_c4 = col4 ^ ((row0 >> 1) & 7);      // swizzle on the source chunk (rows row0+32i share it)
    int a_off[A_LD];
    unsigned long long a_taps[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m0 + row0 + 32 * i;
        a_off[i] = 0; a_taps[i] = 0ull;
        if (PW) {
            a_off[i] = (m * p.Cin + src_c4 * 4) * 4;
            a_taps[i] = m < M ? 1ull : 0ull;
        } else if (STEM) {
            if (m < M) {
                const int img = m / HoWo;
                const int rem = m - img * HoWo;
                const int oy = rem / p.Wo;
                const int ox = rem - oy * p.Wo;
                const int iy0 = oy * p.stride - p.pad, ix = ox * p.stride - p.pad + src_c4;
                a_off[i] = (((img / p.a_img_div) * p.H + iy0) * p.W + ix) * 16;      // bytes, 16 B per pixel
                unsigned long long tm = 0ull;
                if (src_c4 < p.KW && (unsigned)ix < (unsigned)p.W)
                    for (int ky = 0; ky < p.KH; ++ky)
                        if ((unsigned)(iy0 + ky) < (unsigned)p.H) tm |= 1ull << ky;
                a_taps[i] = tm;
            }
        } else if (m < M) {
            const int img = m / HoWo;
            const int rem = m - img * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            a_off[i] = ((((img / p.a_img_div) * p.H + iy0) * p.W + ix0) * p.Cin + src_c4 * 4) * 4;   // bytes
            unsigned long long tm = 0ull;
            int tp = 0;
            for (int ky = 0; ky < p.KH; ++ky) {
                const bool y_ok = (unsigned)(iy0 + ky) < (unsigned)p.H;
                for (int kx = 0; kx < p.KW; ++kx, ++tp)
                    if (y_ok && (unsigned)(ix0 + kx) < (unsigned)p.W) tm |= 1ull << tp;
            }
            a_taps[i] = tm;
        }
    }
    int b_off0 = ((n0 + row0) * p.K + src_c4 * 4) * 4;   // bytes; rows +32i add 32*K*4
    if (PW && p.grp_rows) {      // grouped GEMM (Winograd): per-group weight matrix, rows beyond the valid count skipped
        const int grp = m0 / p.grp_rows;
        int valid = p.grp_valid;
        if (p.grp_count_dev) valid = min(valid, min(p.grp_items, *p.grp_count_dev) * p.grp_rows_per_item);
        if (m0 - grp * p.grp_rows >= valid) return;          // workgroup-uniform, before any barrier
        b_off0 += grp * p.grp_w_stride * 4;
    }

    const i32x4 x_rs = make_rsrc(p.x, p.x_bytes);
    const i32x4 w_rs = make_rsrc(p.w, p.w_bytes);

    const int KT_all = p.K / BK;
    const int kt0 = by * p.kt_per_split;
    const int KT = min(KT_all, kt0 + p.kt_per_split);
    const int cin_tiles = p.Cin / BK;

    // wave-uniform LDS destinations: this wave writes rows [32i + 8*wv, +8) of the A / B tile
    // LDS byte offset of the dynamic segment = low 32 bits of its generic address
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<size_t>(smem));
    const unsigned wave_row_bytes = __builtin_amdgcn_readfirstlane(wv) * 8 * 128;

    // K-tile -> (filter tap, byte offset of the tap's channel slice), wave-uniform.  Tiles are issued in
    // ascending order, so the decode is incremental (two integer divisions per K-tile were ~60 scalar
    // instructions against 16 MFMAs): consecutive channel tiles and consecutive kx are 128 B apart, a wrap
    // of kx jumps to the next input row.
    int nx_tap = 0, nx_c0t = 0, nx_kx = 0, nx_off = 0;
    if (!PW && !STEM) {
        nx_tap = kt0 / cin_tiles;
        nx_c0t = kt0 - nx_tap * cin_tiles;
        const int ky = nx_tap / p.KW;
        nx_kx = nx_tap - ky * p.KW;
        nx_off = ((ky * p.W + nx_kx) * p.Cin + nx_c0t * BK) * 4;
    }
    auto issue_tile = [&](int kt, int stage) {
        int tap = 0, tap_off = kt * BK * 4;
        if (STEM) {
            tap = kt;
            tap_off = kt * p.W * 16;
        } else if (!PW && NSTAGE == 2) {
            tap = nx_tap;
            tap_off = nx_off;
            nx_off += BK * 4;
            if (++nx_c0t == cin_tiles) {
                nx_c0t = 0;
                ++nx_tap;
                if (++nx_kx == p.KW) {
                    nx_kx = 0;
                    nx_off += (p.W - p.KW) * p.Cin * 4;
                }
            }
        } else if (!PW) {
            tap = kt / cin_tiles;
            const int c0 = (kt - tap * cin_tiles) * BK;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            tap_off = ((ky * p.W + kx) * p.Cin + c0) * 4;           // bytes, wave-uniform
        }
        const unsigned sa = lds_base + stage * (STAGE * 4) + wave_row_bytes;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const bool ok = (a_taps[i] >> tap) & 1ull;
            if (PW) {       // one tap, a_off >= 0: the K-tile's offset rides in the scalar offset (the range check sees voff alone)
                lds_dma16_s(x_rs, sa + i * 32 * 128, ok ? (unsigned)a_off[i] : OOB, (unsigned)tap_off);
                continue;
            }
            const unsigned voff = ok ? (unsigned)(a_off[i] + tap_off) : OOB;   // OOB lanes are zero-filled
            lds_dma16(x_rs, sa + i * 32 * 128, voff);
        }
        const unsigned sb = sa + BM * 128;
        const unsigned bko = (unsigned)(kt * BK * 4);
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            lds_dma16_s(w_rs, sb + i * 32 * 128, (unsigned)(b_off0 + i * 32 * p.K * 4), bko);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_row = lane & 31;
    const int half = lane >> 5;
    const int rswz = (frag_row >> 1) & 7;
    const float* const rd_a = smem + (wm * WM + frag_row) * BK;
    const float* const rd_b = smem + BM * BK + (wn * WN + frag_row) * BK;

    // prologue: NSTAGE-1 tiles in flight
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) issue_tile(min(kt0 + s, KT - 1), s);
    if (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    __builtin_amdgcn_s_barrier();

    int cur = 0;
    for (int kt = kt0; kt < KT; ++kt) {
        int nxt = cur + NSTAGE - 1;
        if (nxt >= NSTAGE) nxt -= NSTAGE;
        // (with two stages the last iteration has nothing to prefetch; deeper pipelines keep the clamped
        // re-load so that the counted s_waitcnt below stays exact)
        if (NSTAGE > 2 || kt + 1 < KT) issue_tile(min(kt + NSTAGE - 1, KT - 1), nxt);
        asm volatile("" ::: "memory");

        const float* As = rd_a + cur * STAGE;
        const float* Bs = rd_b + cur * STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int pc = ((kk * 2 + half) ^ rswz) * 4;
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(As + i * 32 * BK + pc);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(Bs + j * 32 * BK + pc);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
            // stem (MODE 2): a 16-byte chunk is one PIXEL of the NHWC4 image and .w its fourth channel, which is zero in the
            // image and in the packed weights alike - a quarter of the MFMAs of the 7x7 stem would multiply zeros
            if (!STEM) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
            }
        }
        // tile t+1 must have landed (own DMAs counted; the barrier covers the other waves');
        // with NSTAGE == 3 the DMAs of tile t+2 stay in flight across the barrier.
        // lgkmcnt(0): this wave's LDS reads of stage `cur` must have RETURNED before it signals the barrier -
        // the waves released by the barrier DMA the next tile into this very stage.  (hipcc hoists the raw
        // s_barrier above the last MFMAs and above the lgkmcnt wait it inserts for them; without the explicit
        // wait a read still queued in the LDS pipe was overtaken by the next tile's data about once per 1e5
        // tiles at 4-5 workgroups per CU: one wave's last k-slice of a K-tile wrong.)
        if (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LOADS) : "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the clamped tail prefetches

    // ---- epilogue.  64x64 tile: accumulators take a round trip through the (now idle) LDS stages so that
    // every lane moves 16 B of one output row - 4x fewer, fully coalesced global instructions than storing
    // the MFMA layout directly (each half-wave 128 B); this is what bounds the small-K 1x1 layers, whose
    // epilogue (68 MB residual read + 68 MB store at layer1) outweighs their K loop.
    constexpr int PITCH = BN + 4;                     // floats; rows keep b128 alignment, shift banks by 4
    // (the launcher sizes the dynamic LDS to max(stages, C tile): 128x128 needs 67.6 KB for the C tile)
    if ((p.Cout & 3) == 0) {
        float* const cbase = smem;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float* cw = cbase + (wm * WM + i * 32 + 4 * half) * PITCH + wn * WN + j * 32 + frag_row;
#pragma unroll
                for (int r = 0; r < 16; ++r) cw[((r & 3) + 8 * (r >> 2)) * PITCH] = acc[i][j][r];
            }
        __syncthreads();
        constexpr int C4 = BN / 4;                    // float4 per tile row
        constexpr int RPP = 256 / C4;                 // rows per pass
        const int c4 = t % C4, rr = t / C4;
        const int n = n0 + c4 * 4;
        if (n < p.Cout) {
            const bool raw = p.splits > 1;
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!raw && p.scale) sc = *reinterpret_cast<const float4*>(p.scale + n);
            if (!raw && p.shift) sh = *reinterpret_cast<const float4*>(p.shift + n);
            float* dst = raw ? p.ws + (size_t)by * ((size_t)p.n_img * HoWo) * p.Cout : p.y;
#pragma unroll
            for (int k0 = 0; k0 < BM / RPP; k0 += 4) {
                float4 res[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {         // residual loads first: their latency overlaps the LDS reads
                    const int m = m0 + rr + RPP * (k0 + k);
                    res[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (!raw && p.residual && m < M)
                        res[k] = *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.Cout + n);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = rr + RPP * (k0 + k);
                    const int m = m0 + row;
                    if (m >= M) continue;
                    float4 v = *reinterpret_cast<const float4*>(cbase + row * PITCH + c4 * 4);
                    if (!raw) {
                        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                        v.x += res[k].x; v.y += res[k].y; v.z += res[k].z; v.w += res[k].w;
                        if (p.relu) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        }
                    }
                    *reinterpret_cast<float4*>(dst + (size_t)m * p.Cout + n) = v;
                }
            }
        }
        return;
    }
    if (p.splits > 1) {
        float* slab = p.ws + (size_t)by * ((size_t)p.n_img * HoWo) * p.Cout;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + frag_row;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + wm * WM + i * 32 + 4 * half;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (n < p.Cout && m < M) slab[(size_t)m * p.Cout + n] = acc[i][j][r];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + frag_row;
        const bool n_ok = n < p.Cout;
        const float sc = (n_ok && p.scale) ? p.scale[n] : 1.f;
        const float sh = (n_ok && p.shift) ? p.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * WM + i * 32 + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (n_ok && m < M) {
                    const size_t o = (size_t)m * p.Cout + n;
                    float v = acc[i][j][r] * sc + sh;
                    if (p.residual) v += p.residual[o];
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.y[o] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int MIN_WAVES, int MODE>
__global__ __launch_bounds__(256, MIN_WAVES) void conv_igemm_dma_kernel(const ConvParams p) {
    CLOCK_STAMP_BEGIN();
    conv_igemm_dma_body<BM, BN, WM, WN, NSTAGE, MODE>(p, blockIdx.x, gridDim.x, blockIdx.y);
#ifdef CONV_CLOCK_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    CLOCK_STAMP_END(blockIdx.x);
}

// Two convolutions with the SAME weights on two tensors of different geometry (the query map and the support maps of
// one backbone layer whose stride keeps them from sharing rows: the strided 3x3 / 1x1 convolutions and the stem) as
// ONE launch: workgroups [0, n0) run p0, the rest p1.  The support half alone is a small grid that needed split-K
// slabs and a reduce launch to fill the chip; behind the query half it needs neither.  n0 is a multiple of 8, so the
// XCD-contiguous tile walk of each half keeps its meaning.  No split-K in this form.
template <int BM, int BN, int WM, int WN, int NSTAGE, int MIN_WAVES, int MODE>
__global__ __launch_bounds__(256, MIN_WAVES) void conv_igemm_dma_pair_kernel(const ConvParams p0, const ConvParams p1,
                                                                             const int n0) {
    if ((int)blockIdx.x < n0)
        conv_igemm_dma_body<BM, BN, WM, WN, NSTAGE, MODE>(p0, blockIdx.x, n0, 0);
    else
        conv_igemm_dma_body<BM, BN, WM, WN, NSTAGE, MODE>(p1, (int)blockIdx.x - n0, (int)gridDim.x - n0, 0);
}


// ------------------------------------------------------------------------------------------------
// Persistent point-wise kernel (the dominant kernel of an episode): the 64x64 LDS-DMA kernel in MODE 1 (1x1 / stride 1
// convolutions and the grouped Winograd GEMM) with workgroups that walk several output tiles (tile, tile + grid, ...;
// 1024 workgroups = 4 per CU, measured best against 512 / 768 / 1280 in round 3), used when a launch has more tiles
// than that.  In the one-tile-per-workgroup kernel every workgroup of a round runs its prologue (index math, first
// DMA, its latency) and its epilogue (accumulators -> LDS -> 16 B stores) at the same time as its neighbours, so the
// matrix pipe idles ~20 % of a launch.  Here the first K-tile of the NEXT output tile is in flight (LDS stage 0) while
// the epilogue of the current one drains through stage 1, and workgroups drift out of phase after their first tile.
// LDS: [stage 0: 16 KB][stage 1: 16 KB]; the C tile of the epilogue (64 x 64 floats) lives in stage 1: 32 KB per
// workgroup.  Wave tile 32x32 as 2x2 tiles of v_mfma_f32_16x16x4_f32: lane group g = lane >> 4 reads the 16-byte
// chunk 4*kk + g of its row, MFMA j contracts k in {j, 4+j, 8+j, 12+j} of the 16-deep step (same cycles per FLOP as
// the 32x32x2 form, +3..5 % on the large GEMMs: another power / clock point, MI355X_MICROARCH.md "DVFS give-back" 7).
//   * Optional second A operand (fgn_conv1x1_dual_nhwc_f32): the K-tiles from p.kt1 on come from p.x2.
//   * Optional launch record (fgn_profile_stamps): an execution's span = first workgroup start -> arrival of the last
//     workgroup on the 100 MHz clock, folded into the record by that workgroup - works inside a replayed hipGraph,
//     where no HIP event can be placed around one kernel.  One returning atomic per workgroup at its exit (the step
//     pays 0.8 % with every launch recorded, r05: bench.py arms them only with --launch-records).
//   * A form with a PRODUCER wave (a fifth wave issues every LDS-DMA, the four MFMA waves none) is 1.9 % faster on the
//     large GEMMs in isolation and 2.7 % SLOWER in the pipelined step (20 instead of 16 waves per CU leave the other
//     episode's kernels less room): tools/micro/conv_pw_experiments.inc, profiles/r05_producer_wave.txt.
// Where its rate goes (in-kernel stamps, Stream-K, decomposition by diagnostic builds, r04): DESIGN.md 4.1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 5) void conv_pw_persist_kernel(const ConvParams p, const int total_tiles) {
    constexpr int BM = 64, BN = 64, WN = 32, WM = 32;
    constexpr int A_LD = 2, B_LD = 2;
    constexpr int STAGE = (BM + BN) * BK;          // floats
    constexpr int PITCH = BN;       // unpadded: ds_write_b32 halves and the 16-lane groups of ds_read_b128 hit distinct banks
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int M = (p.n_img_dev ? min(p.n_img, *p.n_img_dev) : p.n_img) * p.Ho * p.Wo;
    int grp_valid = p.grp_valid;
    if (p.grp_rows && p.grp_count_dev) grp_valid = min(grp_valid, min(p.grp_items, *p.grp_count_dev) * p.grp_rows_per_item);

    const int col4 = t & 7, row0 = t >> 3;
    const int src_c4 = col4 ^ ((row0 >> 1) & 7);
    const i32x4 x_rs = make_rsrc(p.x, p.x_bytes);
    const i32x4 w_rs = make_rsrc(p.w, p.w_bytes);
    const bool dual = p.x2 != nullptr;                         // (launch-uniform) second A operand from K-tile p.kt1 on
    const i32x4 x2_rs = make_rsrc(dual ? p.x2 : p.x, dual ? p.x2_bytes : p.x_bytes);
    const int KT = p.K / BK;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<size_t>(smem));
    const unsigned wave_row_bytes = __builtin_amdgcn_readfirstlane(wv) * 8 * 128;
    constexpr unsigned OOB = 0x7ffffff0u;

    // tile -> (m0, n0): XCD-contiguous runs of logical tiles (the grid is a multiple of 8), optional banded raster
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
    // first tile of this workgroup that has work; the xcd/idx form needs tile < total_tiles
    auto next_active = [&](int tile, int& m0, int& n0) -> int {
        for (; tile < total_tiles; tile += gridDim.x)
            if (coords(tile, m0, n0)) return tile;
        return -1;
    };

    unsigned a_voff[A_LD], a2_voff[A_LD], b_voff[B_LD];
    auto set_offsets = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int m = m0 + row0 + 32 * i;
            a_voff[i] = m < M ? (unsigned)((m * p.Cin + src_c4 * 4) * 4) : OOB;
            a2_voff[i] = OOB;
            if (dual && m < M) a2_voff[i] = (unsigned)(((p.x2_rows ? p.x2_rows[m] : m) * p.cin2 + src_c4 * 4) * 4);
        }
        int b0 = ((n0 + row0) * p.K + src_c4 * 4) * 4;
        if (p.grp_rows) b0 += (m0 / p.grp_rows) * p.grp_w_stride * 4;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) b_voff[i] = (unsigned)(b0 + i * 32 * p.K * 4);
    };
    auto issue_tile = [&](int kt, int stage) {
        const unsigned sa = lds_base + stage * (STAGE * 4) + wave_row_bytes;
        const unsigned ko = (unsigned)(kt * BK * 4);
        // the K-tile's byte offset rides in the instruction's scalar offset: no vector arithmetic per K-tile (round 4:
        // +1..2.5 % on the large GEMMs against `voff + ko` with its select for out-of-range rows and M0 save / restore)
        if (dual && kt >= p.kt1) {                             // its own rows, K offset counted from kt1
            const unsigned ko2 = (unsigned)((kt - p.kt1) * BK * 4);
#pragma unroll
            for (int i = 0; i < A_LD; ++i) lds_dma16_s(x2_rs, sa + i * 32 * 128, a2_voff[i], ko2);
        } else {
#pragma unroll
            for (int i = 0; i < A_LD; ++i) lds_dma16_s(x_rs, sa + i * 32 * 128, a_voff[i], ko);
        }
        const unsigned sb = sa + BM * 128;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) lds_dma16_s(w_rs, sb + i * 32 * 128, b_voff[i], ko);
    };

    const int r16 = lane & 15, g16 = lane >> 4;
    const float* const rd_a = smem + (wm * WM + r16) * BK;
    const float* const rd_b = smem + BM * BK + (wn * WN + r16) * BK;
    float* const cbase = smem + STAGE;              // stage 1

    // launch record (FGN_STAMP_WORDS x uint64).  Start: workgroup 0 alone (a 1024-workgroup grid starts within ~0.5 us).
    // End: arrivals are counted in eight shards on cache lines of their own (blockIdx & 7: 128 arrivals per address,
    // spread over the tail of the launch - 1024 arrivals on ONE address queue for ~12 us, which the first form of this
    // code paid at the start of every launch: 4-7 % of the step), the last arriver of a shard reports to the top counter,
    // the last of those folds the span into the record and re-arms it for the next replay.
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
    int m0, n0;
    int tile = next_active(blockIdx.x, m0, n0);
    if (tile < 0) { leave(); return; }
    CLOCK_STAMP_BEGIN();
    set_offsets(m0, n0);
    issue_tile(0, 0);

    while (true) {
        f32x4 acc4[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // K-tile 0 of this output tile has landed (own DMAs counted, the barrier covers the other waves'); the same
        // barrier orders the previous epilogue's reads of the C tile before this tile's DMA into stage 1
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        for (int kt = 0; kt < KT; ++kt) {
            if (kt + 1 < KT) issue_tile(kt + 1, cur ^ 1);
            asm volatile("" ::: "memory");
            const float* As = rd_a + cur * STAGE;
            const float* Bs = rd_b + cur * STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                float4 af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = r16 + 16 * i;                       // row within the wave's 32
                    const int pc = ((kk * 4 + g16) ^ ((row >> 1) & 7)) * 4;
                    af[i] = *reinterpret_cast<const float4*>(As + 16 * i * BK + pc);
                    bf[i] = *reinterpret_cast<const float4*>(Bs + 16 * i * BK + pc);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc4[i][j], 0, 0, 0);
                        acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc4[i][j], 0, 0, 0);
                        acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bf[j].z, acc4[i][j], 0, 0, 0);
                        acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bf[j].w, acc4[i][j], 0, 0, 0);
                    }
            }
            // see conv_igemm_dma_kernel: reads of stage `cur` must have returned before the barrier is signalled
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cur ^= 1;
        }
        // both stages are free now.  Next output tile: its first K-tile goes to stage 0 while the epilogue below
        // drains this tile through stage 1.
        const int em0 = m0, en0 = n0;
        int nm0 = 0, nn0 = 0;
        const int next = next_active(tile + gridDim.x, nm0, nn0);
        if (next >= 0) {
            set_offsets(nm0, nn0);
            issue_tile(0, 0);
        }
        // 16x16 C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float* cw = cbase + (wm * WM + 16 * i + 4 * g16) * PITCH + wn * WN + 16 * j + r16;
#pragma unroll
                for (int r = 0; r < 4; ++r) cw[r * PITCH] = acc4[i][j][r];
            }
        __syncthreads();
        {
            constexpr int C4 = BN / 4, RPP = 256 / C4;       // 16 float4 per row, 16 rows per pass
            const int c4 = t % C4, rr = t / C4;
            const int n = en0 + c4 * 4;
            if (n < p.Cout) {
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + n);
                if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + n);
                float4 res[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int m = em0 + rr + RPP * k;
                    res[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.residual && m < M) res[k] = *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.Cout + n);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = rr + RPP * k;
                    const int m = em0 + row;
                    if (m >= M) continue;
                    float4 v = *reinterpret_cast<const float4*>(cbase + row * PITCH + c4 * 4);
                    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                    v.x += res[k].x; v.y += res[k].y; v.z += res[k].z; v.w += res[k].w;
                    if (p.relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    *reinterpret_cast<float4*>(p.y + (size_t)m * p.Cout + n) = v;
                }
            }
        }
        if (next < 0) break;
        tile = next; m0 = nm0; n0 = nn0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    leave();
    CLOCK_STAMP_END(blockIdx.x);
}

#include "conv_pw_x3.h"
#include "conv_pw_h2.h"

#ifdef FGN_EXPERIMENTS
#define FGN_EXP_PART 1       // kernels and tuning state
#include "../../tools/micro/conv_pw_experiments.inc"
#undef FGN_EXP_PART
#endif

__global__ void splitk_epilogue_kernel(const ConvParams p) {
    const int HoWo = p.Ho * p.Wo;
    int n_img = p.n_img;
    if (p.n_img_dev) n_img = min(n_img, *p.n_img_dev);
    const size_t total4 = (size_t)n_img * HoWo * p.Cout / 4;
    const size_t slab4 = (size_t)p.n_img * HoWo * p.Cout / 4;
    const int c4n = p.Cout / 4;
    const float4* ws = reinterpret_cast<const float4*>(p.ws);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = ws[i];
        for (int z = 1; z < p.splits; ++z) {
            const float4 b = ws[i + (size_t)z * slab4];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const int c = (int)(i % c4n) * 4;
        if (p.scale) {
            const float4 sc = *reinterpret_cast<const float4*>(p.scale + c);
            a.x *= sc.x; a.y *= sc.y; a.z *= sc.z; a.w *= sc.w;
        }
        if (p.shift) {
            const float4 sh = *reinterpret_cast<const float4*>(p.shift + c);
            a.x += sh.x; a.y += sh.y; a.z += sh.z; a.w += sh.w;
        }
        if (p.residual) {
            const float4 r = reinterpret_cast<const float4*>(p.residual)[i];
            a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
        }
        if (p.relu) {
            a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
        }
        reinterpret_cast<float4*>(p.y)[i] = a;
    }
}

// persistent grid of conv_pw_persist_kernel: 4 workgroups per CU (measured best, r03: 512 / 768 / 1280 workgroups lose
// 7 / 3 / 3 % of the pipelined step), a multiple of 8 (XCDs)
static constexpr int persist_blocks() { return 1024; }

// split-K plan for the 64x64 tile: used when the plain grid would leave most of the 256 CUs idle
static int plan_splits(long long M, int Cout, int KT, int tile_hint, bool pw = false) {
    if (tile_hint < 0) return 1;                         // negative hint: never split (tests)
    if (Cout % 4) return 1;
#ifdef FGN_EXPERIMENTS
    if (g_sk_mode == 3 && tile_hint == 0 && pw) return 1;   // Stream-K for every eligible 1x1 launch (tools/)
#endif
    const long long blocks = ((M + 63) / 64) * cdiv(Cout, 64);
    // Measured on the point-wise layers of a cfg3 episode (tools/split_time.py, r03): a 16-deep K loop (Cin 512) never
    // repays the slabs and the reduce launch (support layer2 conv1 22.3 us unsplit / 26.9 split 4, the 9-RoI shared
    // head's conv3 14.1 / 15.2); with 32 K-tiles a split pays only below ~320 workgroups (query layer3 conv1, 264
    // workgroups: 40.0 us unsplit, 36.0 split 3, 36.7 split 4; AG-RPN head conv, 394 workgroups: 41.1 unsplit, 43.7 split 3).
    if (blocks >= 512 || KT < 32) return 1;
    if (KT == 32) {
        if (blocks >= 320) return 1;
        const int s32 = std::min((int)((768 + blocks - 1) / blocks), 8);
        return s32 < 2 ? 1 : s32;
    }
    int s = (int)((1024 + blocks - 1) / blocks);
    s = std::min(s, KT / 4);
    s = std::min(s, 16);
    return s < 2 ? 1 : s;
}

// banded raster when the launch's weights exceed the L2 budget (see ConvParams::band_nt)
static void set_band(ConvParams& p, int BM, int BN, int m_tiles) {
    constexpr long long budget = 2048 * 1024;       // bytes of weights per band: half an XCD's L2
    const long long per_nt = (long long)BN * (p.K / p.splits) * 4;
    p.band_nt = 0; p.band_mt = 0;
    if (budget > 0 && per_nt * p.n_tiles_n > budget) {
        int nb = (int)std::max<long long>(1, budget / per_nt);
        while (nb > 1 && p.n_tiles_n % nb) --nb;
        const int mt = p.grp_rows ? p.grp_rows / BM : m_tiles;
        if (nb < p.n_tiles_n && mt > 0 && m_tiles % mt == 0) { p.band_nt = nb; p.band_mt = mt; }
    }
}

#ifdef FGN_EXPERIMENTS
#define FGN_EXP_PART 2       // launchers and the hooks the dispatcher below calls
#include "../../tools/micro/conv_pw_experiments.inc"
#undef FGN_EXP_PART
#endif

template <int BM, int BN, int WM, int WN, int MW>
static int launch_cfg(const ConvParams& p0, int M_max, bool cin4, hipStream_t stream) {
    ConvParams p = p0;
    p.n_tiles_n = cdiv(p.Cout, BN);
    const int m_tiles = cdiv(M_max, BM);
    const dim3 grid(m_tiles * p.n_tiles_n, p.splits);
    set_band(p, BM, BN, m_tiles);
    const size_t lds = 2 * (BM + BN) * LDS_STRIDE * sizeof(float);
    static unsigned long long lds_ok[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
    hipError_t attr = fgn_allow_full_lds(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WM, WN, true, MW>), &lds_ok[0]);
    if (attr == hipSuccess)
        attr = fgn_allow_full_lds(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WM, WN, false, MW>), &lds_ok[1]);
    if (attr != hipSuccess) return (int)attr;
    if (!p.in_scale && p.x_bytes != 0) {
        constexpr int NST = (BM + BN >= 256) ? 2 : CONV_DMA_STAGES;   // 128x128 keeps 2 blocks/CU
        const size_t dlds = std::max((size_t)NST * (BM + BN) * BK, (size_t)BM * (BN + 4)) * sizeof(float);
        attr = fgn_allow_full_lds(reinterpret_cast<const void*>(conv_igemm_dma_kernel<BM, BN, WM, WN, NST, MW, 0>), &lds_ok[2]);
        if (attr == hipSuccess)
            attr = fgn_allow_full_lds(reinterpret_cast<const void*>(conv_igemm_dma_kernel<BM, BN, WM, WN, NST, MW, 1>), &lds_ok[3]);
        if (attr == hipSuccess)
            attr = fgn_allow_full_lds(reinterpret_cast<const void*>(conv_igemm_dma_kernel<BM, BN, WM, WN, NST, MW, 2>), &lds_ok[4]);
        if (attr != hipSuccess) return (int)attr;
        const bool pw = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.a_img_div == 1;
        if (cin4)
            FGN_LAUNCH_TIMED((conv_igemm_dma_kernel<BM, BN, WM, WN, NST, MW, 2>), grid, dim3(256), dlds, stream, p);
#ifdef FGN_EXPERIMENTS
        else if (pw && BM == 64 && BN == 64 && p.splits == 1 && (p.Cout & 3) == 0 && p.sk_U > 0 && p.ws && p.tickets) {
            const int rc = fgn_exp_launch_streamk(p, (int)grid.x, stream);
            if (rc != FGN_OK) return rc;
        }
#endif
        else if (pw && BM == 64 && BN == 64 && p.splits == 1 && (p.Cout & 3) == 0 && (int)grid.x > persist_blocks()) {
            // more output tiles than resident workgroups: persistent workgroups walk them (conv_pw_persist_kernel)
            static unsigned long long pk_ok = 0ull;
            attr = fgn_allow_full_lds(reinterpret_cast<const void*>(conv_pw_persist_kernel), &pk_ok);
            if (attr != hipSuccess) return (int)attr;
            const size_t plds = (size_t)2 * (64 + 64) * BK * sizeof(float);
#ifdef FGN_EXPERIMENTS
            { int rc = FGN_OK; if (fgn_exp_launch_persist_ws(p, (int)grid.x, stream, &rc)) return rc; }
#endif
            p.stamp = fgn_next_stamp_record();
            FGN_LAUNCH_TIMED(conv_pw_persist_kernel, dim3(persist_blocks()), dim3(256), plds, stream, p, (int)grid.x);
        } else if (pw)
            FGN_LAUNCH_TIMED((conv_igemm_dma_kernel<BM, BN, WM, WN, NST, MW, 1>), grid, dim3(256), dlds, stream, p);
        else
            FGN_LAUNCH_TIMED((conv_igemm_dma_kernel<BM, BN, WM, WN, NST, MW, 0>), grid, dim3(256), dlds, stream, p);
    } else if (cin4)
        return FGN_ERR_SHAPE;      // the stem runs on the LDS-DMA kernel only (input < 2 GiB)
    else if (p.in_scale)
        FGN_LAUNCH_TIMED((conv_igemm_kernel<BM, BN, WM, WN, true, MW>), grid, dim3(256), lds, stream, p);
    else
        FGN_LAUNCH_TIMED((conv_igemm_kernel<BM, BN, WM, WN, false, MW>), grid, dim3(256), lds, stream, p);
    FGN_LAUNCH_CHECK();
    if (p.splits > 1) {
        const size_t total4 = (size_t)M_max * p.Cout / 4;
        const int eg = (int)std::min<size_t>((total4 + 255) / 256, 2048);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(eg), dim3(256), 0, stream, p);
        FGN_LAUNCH_CHECK();
    }
    return FGN_OK;
}


// tile choice (measured on MI355X, tools/conv_bench.py): 64x64 everywhere, except when the 128x128 grid is one nearly
// full round of 2 workgroups per CU (the 1024 -> 512 conv on 300 RoIs: 460 tiles), where half the L2 traffic per
// MAC is worth ~8 %.  1 = 128x128, 2 = 64x128, 3 = 128x64, 4 = 64x64 (tile_hint forces one; tests).
static int pick_tile(long long M, int Cout, bool has_residual, int tile_hint) {
    int tile = tile_hint < 0 ? -tile_hint : tile_hint;
    if (tile >= 100) tile -= 100;
    if (tile == 0) {
        const long long b128 = ((M + 127) / 128) * cdiv(Cout, 128);
        tile = (b128 >= 420 && b128 <= 512 && !has_residual && (Cout % 4) == 0) ? 1 : 4;
    }
    return tile;
}

// Which kernel the dispatcher launches for a layer: tile * 10 + mode, mode 0 = LDS-DMA generic, 1 = LDS-DMA
// point-wise, 2 = LDS-DMA stem, 3 = register-staged (fused input scale, or operands beyond the 2 GiB buffer
// descriptors), 4 = conv_pw_persist_kernel (point-wise, more output tiles than resident workgroups).  Lets a profiler
// attribute a launch to the kernel name rocprofv3 reports, e.g. 41 = conv_igemm_dma_kernel<64, 64, 32, 32, 2, 4, 1>.
extern "C" int fgn_conv2d_kernel_id(int n_img, int H, int W, int Cin, int Cout, int cout_pad, int KH, int KW, int stride,
                                    int pad, int a_img_div, int has_in_scale, int has_residual, int tile_hint) {
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0 || n_img <= 0 || a_img_div < 1) return FGN_ERR_SHAPE;
    const bool cin4 = Cin == 4;
    const long long M = (long long)n_img * Ho * Wo;
    const int K = cin4 ? KH * BK : cdiv(KH * KW * Cin, BK) * BK;
    const long long xb = (long long)((n_img + a_img_div - 1) / a_img_div) * H * W * Cin * 4;
    const long long wb = (long long)cout_pad * K * 4;
    const bool use_dma = (tile_hint < 100 || cin4) && xb < 0x7fffff00ll && wb < 0x7fffff00ll;
    const int tile = pick_tile(M, Cout, has_residual != 0, tile_hint);
    int mode = 3;
    if (use_dma && !has_in_scale)
        mode = cin4 ? 2 : (KH == 1 && KW == 1 && stride == 1 && pad == 0 && a_img_div == 1) ? 1 : 0;
#ifdef FGN_EXPERIMENTS
    {   // conv_pw_persist2_kernel: tile code * 10 + 5; Stream-K launches: tile * 10 + 6
        const int id = fgn_exp_kernel_id(mode, tile, M, Cout, K, tile_hint, has_residual);
        if (id) return id;
    }
#endif
    // point-wise launches with more 64x64 output tiles than resident workgroups run on conv_pw_persist_kernel
    if (mode == 1 && tile == 4 && (Cout & 3) == 0 &&
        ((M + 63) / 64) * cdiv(Cout, 64) > persist_blocks() && plan_splits(M, Cout, K / BK, tile_hint, true) == 1)
        mode = 4;
    return tile * 10 + mode;
}

extern "C" size_t fgn_conv2d_workspace_bytes(int n_img, int H, int W, int Cin, int Cout, int KH, int KW,
                                             int stride, int pad, int tile_hint) {
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0 || n_img <= 0) return 0;
    if (tile_hint >= 100) tile_hint -= 100;
    const long long M = (long long)n_img * Ho * Wo;
    const int KT = cdiv(KH * KW * Cin, BK);
    if (tile_hint > 0 && tile_hint != 4) return 0;
    const bool pw = KH == 1 && KW == 1 && stride == 1 && pad == 0;
    const int s = plan_splits(M, Cout, KT, tile_hint, pw);
    if (s > 1) return (size_t)s * M * Cout * sizeof(float);
    return 0;
}

extern "C" int fgn_conv2d_nhwc_f32(const float* x, const float* w_packed, float* y, const float* scale,
                                   const float* shift, const float* residual, const float* in_scale,
                                   const int32_t* n_img_dev, int n_img, int H, int W, int Cin, int Cout,
                                   int cout_pad, int KH, int KW, int stride, int pad, int a_img_div,
                                   int relu, int tile_hint, float* splitk_ws, size_t splitk_ws_bytes,
                                   hipStream_t stream) {
    if (!x || !w_packed || !y) return FGN_ERR_ARG;
    if (n_img <= 0) return FGN_OK;
    const bool cin4 = (Cin == 4);
    if (!cin4 && (Cin % BK) != 0) return FGN_ERR_SHAPE;
    if (cin4 && in_scale) return FGN_ERR_SHAPE;
    if (a_img_div < 1 || stride < 1 || cout_pad % 128 != 0 || cout_pad < Cout) return FGN_ERR_SHAPE;
    ConvParams p;
    p.x = x; p.w = w_packed; p.y = y; p.scale = scale; p.shift = shift; p.residual = residual;
    p.in_scale = in_scale; p.n_img_dev = n_img_dev; p.stamp = nullptr; p.x2 = nullptr; p.x2_bytes = 0; p.kt1 = 0; p.cin2 = 0; p.x2_rows = nullptr;
#ifdef FGN_EXPERIMENTS
    p.tickets = nullptr; p.sched = nullptr; p.sk_U = 0; p.sk_dp = 0;
#endif
    p.n_img = n_img; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW;
    p.stride = stride; p.pad = pad; p.a_img_div = a_img_div; p.relu = relu;
    p.grp_rows = 0; p.grp_valid = 0; p.grp_items = 0; p.grp_rows_per_item = 0; p.grp_w_stride = 0;
    p.grp_count_dev = nullptr;
    p.Ho = (H + 2 * pad - KH) / stride + 1;
    p.Wo = (W + 2 * pad - KW) / stride + 1;
    if (p.Ho <= 0 || p.Wo <= 0) return FGN_ERR_SHAPE;
    if (KH * KW > 64) return FGN_ERR_SHAPE;                   // tap validity is a 64-bit mask
    const int k_raw = KH * KW * Cin;
    p.K = cdiv(k_raw, BK) * BK;
    if (cin4) {                                   // stem layout: one K-tile per filter row, [KH][8 pixels][4]
        if (KW > 8 || in_scale) return FGN_ERR_SHAPE;
        p.K = KH * BK;
    }
    // offsets are 32-bit element indices
    if ((long long)(n_img / a_img_div + 1) * H * W * Cin >= (1ll << 31)) return FGN_ERR_SHAPE;
    const long long M = (long long)n_img * p.Ho * p.Wo;
    if (M * (long long)Cout >= (1ll << 31) * 4) return FGN_ERR_SHAPE;

    int tile;
    p.ws = nullptr; p.splits = 1; p.kt_per_split = p.K / BK;
    {
        // descriptor extents (< 4 GiB checked below); tile_hint >= 100 forces the register-staged kernel
        const long long xb = (long long)((n_img + a_img_div - 1) / a_img_div) * H * W * Cin * 4;
        const long long wb = (long long)cout_pad * p.K * 4;
        const bool use_dma = (tile_hint < 100 || cin4) && xb < 0x7fffff00ll && wb < 0x7fffff00ll;
        p.x_bytes = use_dma ? (unsigned)xb : 0u;
        p.w_bytes = use_dma ? (unsigned)wb : 0u;
        if (tile_hint >= 100) tile_hint -= 100;
    }
    tile = pick_tile(M, Cout, residual != nullptr, tile_hint);
    if (tile == 4 && splitk_ws) {
        const int KT = p.K / BK;
        const int sp = plan_splits(M, Cout, KT, tile_hint, KH == 1 && KW == 1 && stride == 1 && pad == 0 && a_img_div == 1 && !in_scale);
        if (sp > 1 && splitk_ws_bytes >= (size_t)sp * M * Cout * sizeof(float)) {
            p.ws = splitk_ws;
            p.kt_per_split = cdiv(KT, sp);
            p.splits = cdiv(KT, p.kt_per_split);
        }
    }
#ifdef FGN_EXPERIMENTS
    if (p.splits == 1 && !in_scale && p.x_bytes != 0 && !cin4 && KH == 1 && KW == 1 && stride == 1 && pad == 0 &&
        a_img_div == 1 && (Cout & 3) == 0 && tile_hint == 0) {
        int rc = FGN_OK;
        if (fgn_exp_pointwise(p, M, tile, false, 0, stream, &rc)) return rc;
    }
#endif
    switch (tile) {
        case 1: return launch_cfg<128, 128, 64, 64, 2>(p, (int)M, cin4, stream);
        case 2: return launch_cfg<64, 128, 32, 64, 3>(p, (int)M, cin4, stream);
        case 3: return launch_cfg<128, 64, 64, 32, 3>(p, (int)M, cin4, stream);
        case 4: return launch_cfg<64, 64, 32, 32, 4>(p, (int)M, cin4, stream);
        default: return FGN_ERR_ARG;
    }
}

// One launch for the same convolution (weights, BN epilogue) on two tensors of different geometry - the query map and
// the support maps of a backbone layer that looks at the spatial structure with a stride (3x3 / stride 2, the 1x1 /
// stride 2 shortcut, the stem): conv_igemm_dma_pair_kernel, 64x64 tiles, no split-K, no residual / input scale.
// Per tensor the arithmetic is that of fgn_conv2d_nhwc_f32 without split-K.
extern "C" int fgn_conv2d_pair_nhwc_f32(const float* x0, float* y0, int n_img0, int H0, int W0, const float* x1, float* y1,
                                        int n_img1, int H1, int W1, const float* w_packed, const float* scale,
                                        const float* shift, int Cin, int Cout, int cout_pad, int KH, int KW, int stride,
                                        int pad, int relu, hipStream_t stream) {
    if (!x0 || !y0 || !x1 || !y1 || !w_packed) return FGN_ERR_ARG;
    if (n_img0 <= 0 || n_img1 <= 0) return FGN_ERR_SHAPE;
    const bool cin4 = (Cin == 4);
    if (!cin4 && (Cin % BK) != 0) return FGN_ERR_SHAPE;
    if (stride < 1 || cout_pad % 128 != 0 || cout_pad < Cout || (Cout & 3) != 0 || KH * KW > 64 || (cin4 && KW > 8))
        return FGN_ERR_SHAPE;
    ConvParams ps[2];
    int tiles[2];
    const float* xs[2] = {x0, x1};
    float* ys[2] = {y0, y1};
    const int ns[2] = {n_img0, n_img1}, Hs[2] = {H0, H1}, Ws[2] = {W0, W1};
    for (int i = 0; i < 2; ++i) {
        ConvParams& p = ps[i];
        p.x = xs[i]; p.w = w_packed; p.y = ys[i]; p.scale = scale; p.shift = shift; p.residual = nullptr;
        p.in_scale = nullptr; p.n_img_dev = nullptr; p.stamp = nullptr; p.x2 = nullptr; p.x2_bytes = 0; p.kt1 = 0; p.cin2 = 0; p.x2_rows = nullptr;
#ifdef FGN_EXPERIMENTS
        p.tickets = nullptr; p.sched = nullptr; p.sk_U = 0; p.sk_dp = 0;
#endif
        p.n_img = ns[i]; p.H = Hs[i]; p.W = Ws[i]; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW;
        p.stride = stride; p.pad = pad; p.a_img_div = 1; p.relu = relu;
        p.grp_rows = 0; p.grp_valid = 0; p.grp_items = 0; p.grp_rows_per_item = 0; p.grp_w_stride = 0;
        p.grp_count_dev = nullptr;
        p.Ho = (p.H + 2 * pad - KH) / stride + 1;
        p.Wo = (p.W + 2 * pad - KW) / stride + 1;
        if (p.Ho <= 0 || p.Wo <= 0) return FGN_ERR_SHAPE;
        p.K = cin4 ? KH * BK : cdiv(KH * KW * Cin, BK) * BK;
        if ((long long)(p.n_img + 1) * p.H * p.W * Cin >= (1ll << 31)) return FGN_ERR_SHAPE;
        const long long M = (long long)p.n_img * p.Ho * p.Wo;
        if (M * (long long)Cout >= (1ll << 31) * 4) return FGN_ERR_SHAPE;
        const long long xb = (long long)p.n_img * p.H * p.W * Cin * 4, wb = (long long)cout_pad * p.K * 4;
        if (xb >= 0x7fffff00ll || wb >= 0x7fffff00ll) return FGN_ERR_SHAPE;
        p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
        p.ws = nullptr; p.splits = 1; p.kt_per_split = p.K / BK;
        p.n_tiles_n = cdiv(Cout, 64);
        const int m_tiles = cdiv((int)M, 64);
        set_band(p, 64, 64, m_tiles);
        tiles[i] = (m_tiles * p.n_tiles_n + 7) / 8 * 8;       // workgroups past the last tile leave at once
    }
    constexpr int NST = CONV_DMA_STAGES;
    const size_t dlds = std::max((size_t)NST * (64 + 64) * BK, (size_t)64 * (64 + 4)) * sizeof(float);
    static unsigned long long lds_ok[2] = {0ull, 0ull};
    hipError_t attr = fgn_allow_full_lds(reinterpret_cast<const void*>(conv_igemm_dma_pair_kernel<64, 64, 32, 32, NST, 4, 0>), &lds_ok[0]);
    if (attr == hipSuccess)
        attr = fgn_allow_full_lds(reinterpret_cast<const void*>(conv_igemm_dma_pair_kernel<64, 64, 32, 32, NST, 4, 2>), &lds_ok[1]);
    if (attr != hipSuccess) return (int)attr;
    const dim3 grid(tiles[0] + tiles[1]);
    if (cin4)
        FGN_LAUNCH_TIMED((conv_igemm_dma_pair_kernel<64, 64, 32, 32, NST, 4, 2>), grid, dim3(256), dlds, stream, ps[0], ps[1], tiles[0]);
    else
        FGN_LAUNCH_TIMED((conv_igemm_dma_pair_kernel<64, 64, 32, 32, NST, 4, 0>), grid, dim3(256), dlds, stream, ps[0], ps[1], tiles[0]);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// y = (x * W1^T + x2 * W2^T) + shift (ReLU) on the same rows: two 1x1 / stride 1 convolutions summed in ONE K loop of
// conv_pw_persist_kernel - a bottleneck's conv3 (+BN) and the 1x1 / stride 1 shortcut (+BN) of the first block of a stage
// whose stride is 1 (mmdet ResNet layer1.0: out = relu(bn3(conv3(y)) + bn_d(conv_d(x)))), with the two BatchNorm scales
// folded into the packed weights [cout_pad][Cin1 + Cin2] and the two shifts added.  Saves the shortcut's launch and the
// write + re-read of its [rows, Cout] output (106 MB at cfg3).  x [rows, Cin1], y [rows, Cout]; x2 [x2_total_rows, Cin2]:
// output row m reads row x2_rows[m] of it (int32 on the device: the 1x1 / STRIDE 2 shortcut of layer2.0 / layer3.0 reads
// every second pixel of every second row of the stage's input), or row m when x2_rows is NULL (then x2_total_rows = rows).
extern "C" int fgn_conv1x1_dual_nhwc_f32(const float* x, const float* x2, const int32_t* x2_rows, int x2_total_rows,
                                         const float* w_packed, float* y, const float* shift, int rows, int Cin1, int Cin2,
                                         int Cout, int cout_pad, int relu, hipStream_t stream) {
    if (!x || !x2 || !w_packed || !y) return FGN_ERR_ARG;
    if (rows <= 0) return FGN_OK;
    if (Cin1 % BK || Cin2 % BK || Cin1 <= 0 || Cin2 <= 0 || (Cout & 3) || cout_pad % 128 || cout_pad < Cout) return FGN_ERR_SHAPE;
    if ((!x2_rows && x2_total_rows != rows) || x2_total_rows < 1) return FGN_ERR_ARG;
    const long long K = (long long)Cin1 + Cin2;
    const long long xb = (long long)rows * Cin1 * 4, x2b = (long long)x2_total_rows * Cin2 * 4, wb = (long long)cout_pad * K * 4;
    if (xb >= 0x7fffff00ll || x2b >= 0x7fffff00ll || wb >= 0x7fffff00ll || (long long)rows * Cout >= (1ll << 31)) return FGN_ERR_SHAPE;
    ConvParams p;
    p.x = x; p.w = w_packed; p.y = y; p.scale = nullptr; p.shift = shift; p.residual = nullptr; p.in_scale = nullptr;
    p.n_img_dev = nullptr; p.stamp = nullptr;
#ifdef FGN_EXPERIMENTS
    p.tickets = nullptr; p.sched = nullptr; p.sk_U = 0; p.sk_dp = 0;
#endif
    p.x2 = x2; p.x2_bytes = (unsigned)x2b; p.kt1 = Cin1 / BK; p.cin2 = Cin2; p.x2_rows = x2_rows;
    p.n_img = rows; p.H = 1; p.W = 1; p.Cin = Cin1; p.Ho = 1; p.Wo = 1; p.Cout = Cout; p.KH = 1; p.KW = 1;
    p.stride = 1; p.pad = 0; p.a_img_div = 1; p.relu = relu; p.K = (int)K;
    p.ws = nullptr; p.splits = 1; p.kt_per_split = (int)K / BK;
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
    p.grp_rows = 0; p.grp_valid = 0; p.grp_items = 0; p.grp_rows_per_item = 0; p.grp_w_stride = 0; p.grp_count_dev = nullptr;
    p.n_tiles_n = cdiv(Cout, 64);
    const int m_tiles = cdiv(rows, 64);
    set_band(p, 64, 64, m_tiles);
    const int tiles = m_tiles * p.n_tiles_n;
    static unsigned long long pk_ok = 0ull;
    hipError_t attr = fgn_allow_full_lds(reinterpret_cast<const void*>(conv_pw_persist_kernel), &pk_ok);
    if (attr != hipSuccess) return (int)attr;
    const size_t plds = (size_t)2 * (64 + 64) * BK * sizeof(float);
    const int grid = std::min(persist_blocks(), (tiles + 7) / 8 * 8);
    p.stamp = fgn_next_stamp_record();
    FGN_LAUNCH_TIMED(conv_pw_persist_kernel, dim3(grid), dim3(256), plds, stream, p, tiles);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// ------------------------------------------------------------------------------------------------
// conv_pw_x3_kernel launches (conv_pw_x3.h): p describes a point-wise / grouped GEMM as for conv_pw_persist_kernel and
// carries the bf16-plane image of its weights (w3, w3_bytes, npad3).  bm: 0 = choose, 64 / 128 = force the row tile;
// nterms 6 (default) or 9.  The persistent grid is 2 workgroups per CU (LDS: 64 / 80 KB per workgroup).
// ------------------------------------------------------------------------------------------------
// Row tile of a launch: M rows in all (grouped: `groups` x grp_rows, the first `valid` rows of a group computed), K deep.
// bm 64 / 128: forced (tests, tools); 0: chosen - 128, 64, or 0 = leave the launch to the f32 MFMA kernels.  Measured per
// launch of a cfg3 episode, both arithmetics on one box (tools/per_launch.py, tools/x3_probe.py):
//  * the 128-column tile loses where much of it is padding (Cout 64: 37 -> 48-53 us; Cout 76: 35.6 -> 39.0), and a launch of
//    a few dozen tiles loses to the f32 path's split-K (441 x 512 x 1024: 9.6 -> 30.6 us); from ~200 tiles on it wins;
//  * 128 rows (4 waves of 64 x 64) beat 64 rows by 5-8 % on the large, deep GEMMs (relation Q 173 -> 162.5 us, the 300-RoI
//    Winograd GEMM 143.7 -> 132.7) and lose 5-50 % where the launch has few tiles, a shallow K loop or rows that fill
//    128-row tiles badly (400 valid rows: 59.8 -> 66.8).
static int x3_pick_bm(long long M, int Cout, int K, int grp_rows, int grp_valid, int bm) {
    if (bm == 64 || bm == 128) return (grp_rows && grp_rows % bm) ? 0 : bm;
    if (grp_rows && grp_rows % 64) return 0;
    const int nt = cdiv(Cout, X3_BN);
    if (Cout * 10 < nt * X3_BN * 7) return 0;
    if (((M + 63) / 64) * nt < 192) return 0;
    const long long groups = grp_rows ? M / grp_rows : 1;
    const long long valid = grp_rows ? std::min<long long>(grp_valid > 0 ? grp_valid : grp_rows, grp_rows) : M;
    const long long rows64 = (valid + 63) / 64 * 64, rows128 = (valid + 127) / 128 * 128;
    if (K >= 256 && (!grp_rows || grp_rows % 128 == 0) && rows128 * 100 <= rows64 * 108 && groups * (rows128 / 128) * nt >= 400)
        return 128;
    return 64;
}

template <int WMW, int RB, int NT, int NST, bool SH16>
static int launch_x3_cfg(ConvParams& p, int M_max, hipStream_t stream) {
    constexpr int BM = 32 * RB * WMW;
    const int m_tiles = cdiv(M_max, BM);
    {   // banded raster: <= 2 MB of weight image per band (see ConvParams::band_nt)
        const long long per_nt = (long long)X3_BN * p.K * 6;
        p.band_nt = 0; p.band_mt = 0;
        if (per_nt * p.n_tiles_n > 2048 * 1024) {
            int nb = (int)std::max<long long>(1, 2048 * 1024 / per_nt);
            while (nb > 1 && p.n_tiles_n % nb) --nb;
            const int mt = p.grp_rows ? p.grp_rows / BM : m_tiles;
            if (nb < p.n_tiles_n && mt > 0 && m_tiles % mt == 0) { p.band_nt = nb; p.band_mt = mt; }
        }
    }
    const int tiles = m_tiles * p.n_tiles_n;
    static unsigned long long ok = 0ull;
    hipError_t attr = fgn_allow_full_lds(reinterpret_cast<const void*>(conv_pw_x3_kernel<WMW, RB, NT, NST, SH16>), &ok);
    if (attr != hipSuccess) return (int)attr;
    const size_t lds = (size_t)NST * (BM * 128 + X3_B_STAGE);
    const int per_cu = (int)(160 * 1024 / lds);                       // resident workgroups per CU (LDS-bound)
    const int grid = std::min(256 * per_cu, (tiles + 7) / 8 * 8);
    p.stamp = fgn_next_stamp_record();
    FGN_LAUNCH_TIMED((conv_pw_x3_kernel<WMW, RB, NT, NST, SH16>), dim3(grid), dim3(128 * WMW), lds, stream, p, tiles);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

// bm: 0 = choose (fgn_x3_row_tile), 64 / 128 = force the row tile; nterms 6 or 9 (nine: 64-row tile only).  The product
// instances use v_mfma_f32_16x16x32_bf16 (weight image of ops.pack_x3).  The experiments build (tools/micro/
// build_experiments.sh) adds what was measured and not chosen (DESIGN Appendix A rows 45-47): bm + 2000 = the
// v_mfma_f32_32x32x16_bf16 form of the 64 / 128-row tiles (image of ops.pack_x3(mfma32=True)), 2129 = 128 rows as 8 waves
// x 3 stages in that form.
static int launch_x3(const ConvParams& p0, int M_max, int bm, int nterms, hipStream_t stream) {
    ConvParams p = p0;
    if (!p.w3 || p.npad3 % X3_BN || p.npad3 < p.Cout || (p.Cout & 3) || p.K % BK || p.K < 2 * BK || p.splits != 1) return FGN_ERR_SHAPE;
    p.n_tiles_n = cdiv(p.Cout, X3_BN);
#ifdef FGN_EXPERIMENTS
    if (bm >= 2000) {
        const int b = bm - 2000;
        if (x3_pick_bm(M_max, p.Cout, p.K, p.grp_rows, p.grp_valid, b == 129 ? 128 : b) == 0) return FGN_ERR_SHAPE;
        if (b == 129) return nterms == 9 ? launch_x3_cfg<4, 1, 9, 3, false>(p, M_max, stream) : launch_x3_cfg<4, 1, 6, 3, false>(p, M_max, stream);
        if (b == 128) return launch_x3_cfg<2, 2, 6, 2, false>(p, M_max, stream);
        if (b == 64) return nterms == 9 ? launch_x3_cfg<2, 1, 9, 2, false>(p, M_max, stream) : launch_x3_cfg<2, 1, 6, 2, false>(p, M_max, stream);
        return FGN_ERR_SHAPE;
    }
#endif
    if (bm != 0 && bm != 64 && bm != 128) return FGN_ERR_SHAPE;
    const int BM = x3_pick_bm(M_max, p.Cout, p.K, p.grp_rows, p.grp_valid, bm);
    if (BM == 0) return FGN_ERR_SHAPE;
    if (nterms == 9) return BM == 64 ? launch_x3_cfg<2, 1, 9, 2, true>(p, M_max, stream) : FGN_ERR_SHAPE;
    return BM == 128 ? launch_x3_cfg<2, 2, 6, 2, true>(p, M_max, stream) : launch_x3_cfg<2, 1, 6, 2, true>(p, M_max, stream);
}

extern "C" size_t fgn_x3_image_bytes(int K, int npad, int n_groups) { return (size_t)n_groups * K * npad * 6; }
// the row tile launch_x3 chooses for a GEMM of M rows x K (grouped: grp_rows per group, grp_valid of them computed; 0, 0
// otherwise): 64 / 128, or 0 = not supported / not profitable: the caller then uses the f32 MFMA entry point (the x3 entry
// points return FGN_ERR_SHAPE for such a launch)
extern "C" int fgn_x3_row_tile(long long M, int Cout, int K, int grp_rows, int grp_valid) {
    return x3_pick_bm(M, Cout, K, grp_rows, grp_valid, 0);
}

#ifdef X3_PHASES        // tools/micro/build_x3_phases.sh: phase clocks of wave 0 of workgroups 0 / 1 (conv_pw_x3.h)
static unsigned long long* g_x3_ph = nullptr;
extern "C" int fgn_x3_phases(unsigned long long* host16) {
    if (!g_x3_ph) return FGN_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return FGN_ERR_ARG;
    if (hipMemcpy(host16, g_x3_ph, 128, hipMemcpyDeviceToHost) != hipSuccess) return FGN_ERR_ARG;
    return hipMemset(g_x3_ph, 0, 128) == hipSuccess ? FGN_OK : FGN_ERR_ARG;
}
#endif

// y[rows, Cout] = relu?(x[rows, K] * W^T + shift + residual) with W given as its bf16-plane image (ops.pack_x3);
// grouped: rows = n_groups * grp_rows, group g uses image g and computes its first grp_valid rows.  The direct entry
// of conv_pw_x3_kernel (tests, tools); the convolution entry points take the image as an optional argument.
extern "C" int fgn_gemm_x3_f32(const float* x, const void* w3, float* y, const float* shift, const float* residual,
                               int rows, int K, int Cout, int npad, int relu, int grp_rows, int grp_valid, int n_groups,
                               int bm, int nterms, hipStream_t stream) {
    if (!x || !w3 || !y) return FGN_ERR_ARG;
    if (rows <= 0) return FGN_OK;
    if (K % BK || K <= 0 || (Cout & 3) || npad % X3_BN || npad < Cout || n_groups < 1) return FGN_ERR_SHAPE;
    if (n_groups > 1 && (grp_rows <= 0 || (long long)n_groups * grp_rows != rows || grp_valid > grp_rows)) return FGN_ERR_SHAPE;
    const long long xb = (long long)rows * K * 4, wb = (long long)fgn_x3_image_bytes(K, npad, n_groups);
    if (xb >= 0x7fffff00ll || wb >= 0x7fffff00ll || (long long)rows * Cout >= (1ll << 31)) return FGN_ERR_SHAPE;
    ConvParams p;
    p.x = x; p.w = nullptr; p.y = y; p.scale = nullptr; p.shift = shift; p.residual = residual; p.in_scale = nullptr;
    p.n_img_dev = nullptr; p.stamp = nullptr; p.x2 = nullptr; p.x2_bytes = 0; p.kt1 = 0; p.cin2 = 0; p.x2_rows = nullptr;
#ifdef FGN_EXPERIMENTS
    p.tickets = nullptr; p.sched = nullptr; p.sk_U = 0; p.sk_dp = 0;
#endif
    p.n_img = rows; p.H = 1; p.W = 1; p.Cin = K; p.Ho = 1; p.Wo = 1; p.Cout = Cout; p.KH = 1; p.KW = 1;
    p.stride = 1; p.pad = 0; p.a_img_div = 1; p.relu = relu; p.K = K;
    p.ws = nullptr; p.splits = 1; p.kt_per_split = K / BK;
    p.x_bytes = (unsigned)xb; p.w_bytes = 0;
    p.grp_rows = n_groups > 1 ? grp_rows : 0; p.grp_valid = grp_valid; p.grp_items = 0; p.grp_rows_per_item = 0;
    p.grp_w_stride = 0; p.grp_count_dev = nullptr;
    p.band_nt = 0; p.band_mt = 0; p.n_tiles_n = 0;
    p.w3 = w3; p.w3_bytes = (unsigned)wb; p.npad3 = npad;
#ifdef X3_PHASES
    if (!g_x3_ph && (hipMalloc(&g_x3_ph, 128) != hipSuccess || hipMemset(g_x3_ph, 0, 128) != hipSuccess)) return FGN_ERR_ARG;
    p.ws = reinterpret_cast<float*>(g_x3_ph);
#endif
    return launch_x3(p, rows, bm, nterms, stream);
}

// The three GEMM-shaped entry points of the detector on conv_pw_x3_kernel.  Arguments as their f32-MFMA counterparts
// (fgn_conv2d_nhwc_f32 for a 1x1 / stride 1 / unpadded convolution, fgn_conv1x1_dual_nhwc_f32, fgn_winograd_gemm_f32),
// with the weights given as the bf16-plane image of ops.pack_x3 ([groups][K / 32][3][cout_pad][32] bf16) instead of
// [cout_pad][K] floats.  Results: the same f32 values to within the rounding of an f32 accumulation (conv_pw_x3.h).
static void x3_base_params(ConvParams& p) {
    p.w = nullptr; p.scale = nullptr; p.shift = nullptr; p.residual = nullptr; p.in_scale = nullptr; p.n_img_dev = nullptr;
    p.stamp = nullptr; p.x2 = nullptr; p.x2_bytes = 0; p.kt1 = 0; p.cin2 = 0; p.x2_rows = nullptr;
#ifdef FGN_EXPERIMENTS
    p.tickets = nullptr; p.sched = nullptr; p.sk_U = 0; p.sk_dp = 0;
#endif
    p.H = 1; p.W = 1; p.Ho = 1; p.Wo = 1; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.a_img_div = 1; p.relu = 0;
    p.ws = nullptr; p.splits = 1; p.w_bytes = 0;
    p.grp_rows = 0; p.grp_valid = 0; p.grp_items = 0; p.grp_rows_per_item = 0; p.grp_w_stride = 0; p.grp_count_dev = nullptr;
    p.band_nt = 0; p.band_mt = 0; p.n_tiles_n = 0;
}

extern "C" int fgn_conv1x1_x3_nhwc_f32(const float* x, const void* w_x3, float* y, const float* scale, const float* shift,
                                       const float* residual, const int32_t* n_img_dev, int n_img, int H, int W, int Cin,
                                       int Cout, int cout_pad, int relu, hipStream_t stream) {
    if (!x || !w_x3 || !y) return FGN_ERR_ARG;
    if (n_img <= 0) return FGN_OK;
    if (Cin % BK || Cin < 2 * BK || (Cout & 3) || cout_pad % X3_BN || cout_pad < Cout || H <= 0 || W <= 0) return FGN_ERR_SHAPE;
    const long long M = (long long)n_img * H * W;
    const long long xb = M * Cin * 4, wb = (long long)fgn_x3_image_bytes(Cin, cout_pad, 1);
    if (xb >= 0x7fffff00ll || wb >= 0x7fffff00ll || M * Cout >= (1ll << 31)) return FGN_ERR_SHAPE;
    ConvParams p;
    x3_base_params(p);
    p.x = x; p.y = y; p.scale = scale; p.shift = shift; p.residual = residual; p.n_img_dev = n_img_dev;
    p.n_img = n_img; p.H = H; p.W = W; p.Ho = H; p.Wo = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu; p.K = Cin;
    p.kt_per_split = Cin / BK; p.x_bytes = (unsigned)xb;
    p.w3 = w_x3; p.w3_bytes = (unsigned)wb; p.npad3 = cout_pad;
    return launch_x3(p, (int)M, 0, 6, stream);
}

extern "C" int fgn_conv1x1_dual_x3_nhwc_f32(const float* x, const float* x2, const int32_t* x2_rows, int x2_total_rows,
                                            const void* w_x3, float* y, const float* shift, int rows, int Cin1, int Cin2,
                                            int Cout, int cout_pad, int relu, hipStream_t stream) {
    if (!x || !x2 || !w_x3 || !y) return FGN_ERR_ARG;
    if (rows <= 0) return FGN_OK;
    if (Cin1 % BK || Cin2 % BK || Cin1 <= 0 || Cin2 <= 0 || (Cout & 3) || cout_pad % X3_BN || cout_pad < Cout) return FGN_ERR_SHAPE;
    if ((!x2_rows && x2_total_rows != rows) || x2_total_rows < 1) return FGN_ERR_ARG;
    const int K = Cin1 + Cin2;
    const long long xb = (long long)rows * Cin1 * 4, x2b = (long long)x2_total_rows * Cin2 * 4;
    const long long wb = (long long)fgn_x3_image_bytes(K, cout_pad, 1);
    if (xb >= 0x7fffff00ll || x2b >= 0x7fffff00ll || wb >= 0x7fffff00ll || (long long)rows * Cout >= (1ll << 31)) return FGN_ERR_SHAPE;
    ConvParams p;
    x3_base_params(p);
    p.x = x; p.y = y; p.shift = shift;
    p.x2 = x2; p.x2_bytes = (unsigned)x2b; p.kt1 = Cin1 / BK; p.cin2 = Cin2; p.x2_rows = x2_rows;
    p.n_img = rows; p.Cin = Cin1; p.Cout = Cout; p.relu = relu; p.K = K;
    p.kt_per_split = K / BK; p.x_bytes = (unsigned)xb;
    p.w3 = w_x3; p.w3_bytes = (unsigned)wb; p.npad3 = cout_pad;
    return launch_x3(p, rows, 0, 6, stream);
}

extern "C" int fgn_winograd_gemm_x3_f32(const float* V, const void* U_x3, float* Mo, const int32_t* n_img_dev, int n_img,
                                        int tiles_per_img, int t_pad, int Cin, int Cout, int cout_pad, int n_groups,
                                        hipStream_t stream) {
    if (!V || !U_x3 || !Mo) return FGN_ERR_ARG;
    if (n_img <= 0) return FGN_OK;
    if (Cin % BK != 0 || Cin < 2 * BK || Cout % 4 != 0 || cout_pad % X3_BN != 0 || cout_pad < Cout || t_pad % 64 != 0 ||
        (n_groups != 16 && n_groups != 36) || (long long)n_img * tiles_per_img > t_pad)
        return FGN_ERR_SHAPE;
    const long long rows = (long long)n_groups * t_pad;
    const long long xb = rows * Cin * 4, wb = (long long)fgn_x3_image_bytes(Cin, cout_pad, n_groups);
    if (xb >= 0x7fffff00ll || wb >= 0x7fffff00ll || rows * Cout >= (1ll << 31)) return FGN_ERR_SHAPE;
    ConvParams p;
    x3_base_params(p);
    p.x = V; p.y = Mo;
    p.n_img = (int)rows; p.Cin = Cin; p.Cout = Cout; p.K = Cin; p.kt_per_split = Cin / BK; p.x_bytes = (unsigned)xb;
    p.grp_rows = t_pad; p.grp_valid = n_img * tiles_per_img; p.grp_items = n_img;
    p.grp_rows_per_item = tiles_per_img; p.grp_count_dev = n_img_dev;
    p.w3 = U_x3; p.w3_bytes = (unsigned)wb; p.npad3 = cout_pad;
    return launch_x3(p, (int)rows, 0, 6, stream);
}

// ------------------------------------------------------------------------------------------------
// conv_pw_h2_kernel launches (conv_pw_h2.h): the same GEMMs with three f16 MFMA products per f32 product.  p carries the
// two-plane f16 image of the column-scaled weights (ops.pack_h2: image, then the inverse column scales).
// Tile of a launch (h2_pick; 0 = leave the launch to the f32 kernels): x3_pick_bm's rule for the 128-column tiles
// (64 / 128 rows), and for at most 64 output channels - where half of a 128-column tile would be padding - 128 rows x 64
// columns as four waves along M (code 264): layer1's conv1 / 3x3 convolutions.
// ------------------------------------------------------------------------------------------------
static int h2_pick(long long M, int Cout, int K, int grp_rows, int grp_valid, int bm) {
    if (bm == 264) return (grp_rows && grp_rows % 128) ? 0 : 264;
    if (bm != 0) return x3_pick_bm(M, Cout, K, grp_rows, grp_valid, bm);
    if (Cout <= 64) {
        if (grp_rows && grp_rows % 128) return 0;
        if (Cout < 48 || (M + 127) / 128 < 256) return 0;       // mostly padding / too few tiles for a persistent grid
        return 264;
    }
    // 128 rows: x3_pick_bm's conditions, for the grouped (Winograd) launches only - measured with this kernel
    // (profiles/r05_h2_probe_record_vs_own_scale.jsonl, own scale): the 300-RoI / AG-RPN Winograd GEMMs 99.1 -> 94.2 / 246.6 ->
    // 239.6 us on 128 rows, the plain 1x1 launches equal or slower (shared-head conv3 62.5 -> 66.5, conv1 62.4 -> 70.6, relation Q
    // 109.1 -> 109.0): three 64-row workgroups per CU hide more of each other's phases than two 128-row ones
    const int bm_x3 = x3_pick_bm(M, Cout, K, grp_rows, grp_valid, 0);
    return (bm_x3 == 128 && !grp_rows) ? 64 : bm_x3;
}

template <int WMW, int WNW, int RB, int NST, bool IM2COL>
static int launch_h2_cfg(ConvParams& p, const H2Im2col& q2, int M_max, hipStream_t stream) {
    constexpr int BM = 32 * RB * WMW, BN = 64 * WNW;
    const int m_tiles = cdiv(M_max, BM);
    p.n_tiles_n = cdiv(p.Cout, BN);
    {   // banded raster: <= 2 MB of weight image per band (see ConvParams::band_nt)
        const long long per_nt = (long long)BN * p.K * 4;
        p.band_nt = 0; p.band_mt = 0;
        if (per_nt * p.n_tiles_n > 2048 * 1024) {
            int nb = (int)std::max<long long>(1, 2048 * 1024 / per_nt);
            while (nb > 1 && p.n_tiles_n % nb) --nb;
            const int mt = p.grp_rows ? p.grp_rows / BM : m_tiles;
            if (nb < p.n_tiles_n && mt > 0 && m_tiles % mt == 0) { p.band_nt = nb; p.band_mt = mt; }
        }
    }
    const int tiles = m_tiles * p.n_tiles_n;
    static unsigned long long ok = 0ull;
    hipError_t attr = fgn_allow_full_lds(reinterpret_cast<const void*>(conv_pw_h2_kernel<WMW, WNW, RB, NST, IM2COL>), &ok);
    if (attr != hipSuccess) return (int)attr;
    const size_t lds = (size_t)NST * (BM * 128 + 2 * BN * 64);
    const int per_cu = std::min(3, (int)(160 * 1024 / lds));         // resident workgroups per CU (LDS-bound)
    const int grid = std::min(256 * per_cu, (tiles + 7) / 8 * 8);
    p.stamp = fgn_next_stamp_record();
    FGN_LAUNCH_TIMED((conv_pw_h2_kernel<WMW, WNW, RB, NST, IM2COL>), dim3(grid), dim3(64 * WMW * WNW), lds, stream, p, q2, tiles);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

extern "C" size_t fgn_h2_image_bytes(int K, int npad, int n_groups) { return (size_t)n_groups * npad * ((size_t)K * 4 + 4); }
// the tile conv_pw_h2_kernel runs a GEMM of this shape on: 64 / 128 = rows of a 128-column tile, 264 = 128 rows x 64
// columns, 0 = not supported / not profitable (the h2 entry points return FGN_ERR_SHAPE: use the f32 entry point)
extern "C" int fgn_h2_row_tile(long long M, int Cout, int K, int grp_rows, int grp_valid) {
    return h2_pick(M, Cout, K, grp_rows, grp_valid, 0);
}

// bm: 0 = choose (h2_pick), 64 / 128 / 264 = force the tile
static int launch_h2(const ConvParams& p0, int M_max, int n_groups, int bm, hipStream_t stream, const H2Im2col* im = nullptr) {
    ConvParams p = p0;
    if (!p.w3 || p.npad3 % H2_BN || p.npad3 < p.Cout || (p.Cout & 3) || p.K % BK || p.K < 2 * BK || p.splits != 1) return FGN_ERR_SHAPE;
    const size_t img = (size_t)n_groups * p.K * p.npad3 * 4;
    p.w3_bytes = (unsigned)img;
    p.w_inv = reinterpret_cast<const float*>(static_cast<const char*>(p.w3) + img);
    H2Im2col none;
    none.y1 = nullptr; none.x_off0 = none.x_off1 = 0u; none.M0 = none.M1 = 0; none.H1 = none.W1 = none.Ho1 = none.Wo1 = 1; none.cin_shift = 0;
    if (bm != 0 && bm != 64 && bm != 128 && bm != 264) return FGN_ERR_SHAPE;
    const int cfg = h2_pick(M_max, p.Cout, p.K, p.grp_rows, p.grp_valid, bm);
    if (cfg == 0) return FGN_ERR_SHAPE;
    if (im) {
        if (cfg == 264) return launch_h2_cfg<4, 1, 1, 2, true>(p, *im, M_max, stream);
        return cfg == 128 ? launch_h2_cfg<2, 2, 2, 2, true>(p, *im, M_max, stream) : launch_h2_cfg<2, 2, 1, 2, true>(p, *im, M_max, stream);
    }
    if (cfg == 264) return launch_h2_cfg<4, 1, 1, 2, false>(p, none, M_max, stream);
    return cfg == 128 ? launch_h2_cfg<2, 2, 2, 2, false>(p, none, M_max, stream) : launch_h2_cfg<2, 2, 1, 2, false>(p, none, M_max, stream);
}

// A KH x KW (1x1 or 3x3) / stride / pad convolution with folded scale / shift / ReLU on ONE or TWO NHWC tensors that
// share the weights (x1 == nullptr: one) as an implicit GEMM on conv_pw_h2_kernel: the im2col of conv_igemm_dma_kernel
// (per row the offset of filter tap (0, 0) and a bit mask of the taps inside the image; per K-tile one wave-uniform tap
// offset; out-of-image taps fetched out of bounds = zeros) in front of the f16-plane products.  w_h2 = ops.pack_h2 of the
// packed weights [cout_pad][KH KW Cin] (K order: tap, channel).  Cin / 32 a power of two.  Both inputs must lie within
// 2 GiB of each other (they do: the query map and the support maps of a layer share one buffer).
extern "C" int fgn_conv2d_pair_h2_nhwc_f32(const float* x0, int n_img0, int H0, int W0, const float* x1, int n_img1, int H1,
                                           int W1, const void* w_h2, float* y0, float* y1, const float* scale,
                                           const float* shift, int Cin, int Cout, int cout_pad, int KH, int KW, int stride,
                                           int pad, int relu, hipStream_t stream) {
    if (!x0 || !w_h2 || !y0 || (x1 && !y1)) return FGN_ERR_ARG;
    if (n_img0 <= 0 || (x1 && n_img1 <= 0)) return FGN_ERR_SHAPE;
    const int ct = Cin / BK;
    if (Cin % BK || (ct & (ct - 1)) || KH != KW || (KW != 1 && KW != 3) || stride < 1 || pad < 0 || (Cout & 3) ||
        cout_pad % H2_BN || cout_pad < Cout || H0 <= 0 || W0 <= 0)
        return FGN_ERR_SHAPE;
    const int Ho0 = (H0 + 2 * pad - KH) / stride + 1, Wo0 = (W0 + 2 * pad - KW) / stride + 1;
    const int Ho1 = x1 ? (H1 + 2 * pad - KH) / stride + 1 : 1, Wo1 = x1 ? (W1 + 2 * pad - KW) / stride + 1 : 1;
    if (Ho0 <= 0 || Wo0 <= 0 || Ho1 <= 0 || Wo1 <= 0) return FGN_ERR_SHAPE;
    const long long M0 = (long long)n_img0 * Ho0 * Wo0, M1 = x1 ? (long long)n_img1 * Ho1 * Wo1 : 0;
    const long long b0 = (long long)n_img0 * H0 * W0 * Cin * 4, b1 = x1 ? (long long)n_img1 * H1 * W1 * Cin * 4 : 0;
    const char* lo = reinterpret_cast<const char*>(x0);
    if (x1 && reinterpret_cast<const char*>(x1) < lo) lo = reinterpret_cast<const char*>(x1);
    const long long off0 = reinterpret_cast<const char*>(x0) - lo, off1 = x1 ? reinterpret_cast<const char*>(x1) - lo : 0;
    const long long span = std::max(off0 + b0, off1 + b1);
    const int K = KH * KW * Cin;
    const long long wb = (long long)fgn_h2_image_bytes(K, cout_pad, 1);
    if (span >= 0x7fffff00ll || wb >= 0x7fffff00ll || (M0 + M1) * Cout >= (1ll << 31)) return FGN_ERR_SHAPE;
    ConvParams p;
    x3_base_params(p);
    p.x = reinterpret_cast<const float*>(lo); p.x_bytes = (unsigned)span; p.y = y0; p.scale = scale; p.shift = shift;
    p.n_img = n_img0; p.H = H0; p.W = W0; p.Ho = Ho0; p.Wo = Wo0; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
    p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.K = K; p.kt_per_split = K / BK;
    p.w3 = w_h2; p.npad3 = cout_pad;
    H2Im2col q;
    q.y1 = y1; q.x_off0 = (unsigned)off0; q.x_off1 = (unsigned)off1; q.M0 = (int)M0; q.M1 = (int)M1;
    q.H1 = x1 ? H1 : 1; q.W1 = x1 ? W1 : 1; q.Ho1 = Ho1; q.Wo1 = Wo1;
    q.cin_shift = 0;
    while ((1 << q.cin_shift) < ct) ++q.cin_shift;
    return launch_h2(p, (int)(M0 + M1), 1, 0, stream, &q);
}

// y[rows, Cout] = relu?(x[rows, K] * W^T + shift + residual) with W given as its two-plane f16 image (ops.pack_h2);
// grouped as fgn_gemm_x3_f32.  The direct entry of conv_pw_h2_kernel (tests, tools).
extern "C" int fgn_gemm_h2_f32(const float* x, const void* w_h2, float* y, const float* shift, const float* residual,
                               int rows, int K, int Cout, int npad, int relu, int grp_rows, int grp_valid, int n_groups,
                               int bm, hipStream_t stream) {
    if (!x || !w_h2 || !y) return FGN_ERR_ARG;
    if (rows <= 0) return FGN_OK;
    if (K % BK || K <= 0 || (Cout & 3) || npad % H2_BN || npad < Cout || n_groups < 1) return FGN_ERR_SHAPE;
    if (n_groups > 1 && (grp_rows <= 0 || (long long)n_groups * grp_rows != rows || grp_valid > grp_rows)) return FGN_ERR_SHAPE;
    const long long xb = (long long)rows * K * 4, wb = (long long)fgn_h2_image_bytes(K, npad, n_groups);
    if (xb >= 0x7fffff00ll || wb >= 0x7fffff00ll || (long long)rows * Cout >= (1ll << 31)) return FGN_ERR_SHAPE;
    ConvParams p;
    x3_base_params(p);
    p.x = x; p.y = y; p.shift = shift; p.residual = residual;
    p.n_img = rows; p.Cin = K; p.Cout = Cout; p.relu = relu; p.K = K; p.kt_per_split = K / BK; p.x_bytes = (unsigned)xb;
    p.grp_rows = n_groups > 1 ? grp_rows : 0; p.grp_valid = grp_valid;
    p.w3 = w_h2; p.npad3 = npad;
    return launch_h2(p, rows, n_groups, bm, stream);
}

// The three GEMM-shaped entry points of the detector on conv_pw_h2_kernel: arguments as fgn_winograd_gemm_x3_f32 /
// fgn_conv1x1_x3_nhwc_f32 / fgn_conv1x1_dual_x3_nhwc_f32, the weights as their two-plane f16 image (ops.pack_h2)
extern "C" int fgn_winograd_gemm_h2_f32(const float* V, const void* U_h2, float* Mo, const int32_t* n_img_dev, int n_img,
                                        int tiles_per_img, int t_pad, int Cin, int Cout, int cout_pad, int n_groups,
                                        hipStream_t stream) {
    if (!V || !U_h2 || !Mo) return FGN_ERR_ARG;
    if (n_img <= 0) return FGN_OK;
    if (Cin % BK != 0 || Cin < 2 * BK || Cout % 4 != 0 || cout_pad % H2_BN != 0 || cout_pad < Cout || t_pad % 64 != 0 ||
        (n_groups != 16 && n_groups != 36) || (long long)n_img * tiles_per_img > t_pad)
        return FGN_ERR_SHAPE;
    const long long rows = (long long)n_groups * t_pad;
    const long long xb = rows * Cin * 4, wb = (long long)fgn_h2_image_bytes(Cin, cout_pad, n_groups);
    if (xb >= 0x7fffff00ll || wb >= 0x7fffff00ll || rows * Cout >= (1ll << 31)) return FGN_ERR_SHAPE;
    ConvParams p;
    x3_base_params(p);
    p.x = V; p.y = Mo;
    p.n_img = (int)rows; p.Cin = Cin; p.Cout = Cout; p.K = Cin; p.kt_per_split = Cin / BK; p.x_bytes = (unsigned)xb;
    p.grp_rows = t_pad; p.grp_valid = n_img * tiles_per_img; p.grp_items = n_img;
    p.grp_rows_per_item = tiles_per_img; p.grp_count_dev = n_img_dev;
    p.w3 = U_h2; p.npad3 = cout_pad;
    return launch_h2(p, (int)rows, n_groups, 0, stream);
}

extern "C" int fgn_conv1x1_h2_nhwc_f32(const float* x, const void* w_h2, float* y, const float* scale, const float* shift,
                                       const float* residual, const int32_t* n_img_dev, int n_img, int H, int W, int Cin,
                                       int Cout, int cout_pad, int relu, hipStream_t stream) {
    if (!x || !w_h2 || !y) return FGN_ERR_ARG;
    if (n_img <= 0) return FGN_OK;
    if (Cin % BK || Cin < 2 * BK || (Cout & 3) || cout_pad % H2_BN || cout_pad < Cout || H <= 0 || W <= 0) return FGN_ERR_SHAPE;
    const long long M = (long long)n_img * H * W;
    const long long xb = M * Cin * 4, wb = (long long)fgn_h2_image_bytes(Cin, cout_pad, 1);
    if (xb >= 0x7fffff00ll || wb >= 0x7fffff00ll || M * Cout >= (1ll << 31)) return FGN_ERR_SHAPE;
    ConvParams p;
    x3_base_params(p);
    p.x = x; p.y = y; p.scale = scale; p.shift = shift; p.residual = residual; p.n_img_dev = n_img_dev;
    p.n_img = n_img; p.H = H; p.W = W; p.Ho = H; p.Wo = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu; p.K = Cin;
    p.kt_per_split = Cin / BK; p.x_bytes = (unsigned)xb;
    p.w3 = w_h2; p.npad3 = cout_pad;
    return launch_h2(p, (int)M, 1, 0, stream);
}

extern "C" int fgn_conv1x1_dual_h2_nhwc_f32(const float* x, const float* x2, const int32_t* x2_rows, int x2_total_rows,
                                            const void* w_h2, float* y, const float* shift, int rows, int Cin1, int Cin2,
                                            int Cout, int cout_pad, int relu, hipStream_t stream) {
    if (!x || !x2 || !w_h2 || !y) return FGN_ERR_ARG;
    if (rows <= 0) return FGN_OK;
    if (Cin1 % BK || Cin2 % BK || Cin1 <= 0 || Cin2 <= 0 || (Cout & 3) || cout_pad % H2_BN || cout_pad < Cout) return FGN_ERR_SHAPE;
    if ((!x2_rows && x2_total_rows != rows) || x2_total_rows < 1) return FGN_ERR_ARG;
    const int K = Cin1 + Cin2;
    const long long xb = (long long)rows * Cin1 * 4, x2b = (long long)x2_total_rows * Cin2 * 4;
    const long long wb = (long long)fgn_h2_image_bytes(K, cout_pad, 1);
    if (xb >= 0x7fffff00ll || x2b >= 0x7fffff00ll || wb >= 0x7fffff00ll || (long long)rows * Cout >= (1ll << 31)) return FGN_ERR_SHAPE;
    ConvParams p;
    x3_base_params(p);
    p.x = x; p.y = y; p.shift = shift;
    p.x2 = x2; p.x2_bytes = (unsigned)x2b; p.kt1 = Cin1 / BK; p.cin2 = Cin2; p.x2_rows = x2_rows;
    p.n_img = rows; p.Cin = Cin1; p.Cout = Cout; p.relu = relu; p.K = K;
    p.kt_per_split = K / BK; p.x_bytes = (unsigned)xb;
    p.w3 = w_h2; p.npad3 = cout_pad;
    return launch_h2(p, rows, 1, 0, stream);
}

// ------------------------------------------------------------------------------------------------
// The 16 / 36 GEMMs of a Winograd F(2x2,3x3) / F(4x4,3x3) convolution (winograd.hip holds the transforms):
//   Mo[g][t][n] = sum_c V[g][t][c] * U[g][n][c],   g = position in the 4x4 / 6x6 transformed tile
// run as ONE launch of the 64x64 kernel in point-wise mode over the stacked rows [n_groups * t_pad] with a
// per-group weight matrix.  t_pad is a multiple of 128 rows so no 64- or 128-row tile straddles two groups (rows past the
// valid count of a group are skipped tile-wise).  (A grouped 128x128 persistent variant measured
// equal on the AG-RPN GEMM and 5 % slower on the 300-RoI one, and much slower on everything smaller.)
// ------------------------------------------------------------------------------------------------
extern "C" int fgn_winograd_t_pad(int tiles_total) { return (tiles_total + 127) / 128 * 128; }   // whole 64- and 128-row tiles per group

extern "C" int fgn_winograd_gemm_f32(const float* V, const float* U, float* Mo, const int32_t* n_img_dev, int n_img,
                                     int tiles_per_img, int t_pad, int Cin, int Cout, int cout_pad, int n_groups,
                                     hipStream_t stream) {
    if (!V || !U || !Mo) return FGN_ERR_ARG;
    if (n_img <= 0) return FGN_OK;
    if (Cin % BK != 0 || Cout % 4 != 0 || cout_pad % 128 != 0 || cout_pad < Cout || t_pad % 64 != 0 ||
        (n_groups != 16 && n_groups != 36) || (long long)n_img * tiles_per_img > t_pad)
        return FGN_ERR_SHAPE;
    const long long rows = (long long)n_groups * t_pad;
    const long long xb = rows * Cin * 4, wb = (long long)n_groups * cout_pad * Cin * 4;
    if (xb >= 0x7fffff00ll || wb >= 0x7fffff00ll || rows * Cout >= (1ll << 31)) return FGN_ERR_SHAPE;
    ConvParams p;
    p.x = V; p.w = U; p.y = Mo; p.scale = nullptr; p.shift = nullptr; p.residual = nullptr; p.in_scale = nullptr;
    p.n_img_dev = nullptr; p.stamp = nullptr; p.x2 = nullptr; p.x2_bytes = 0; p.kt1 = 0; p.cin2 = 0; p.x2_rows = nullptr;
#ifdef FGN_EXPERIMENTS
    p.tickets = nullptr; p.sched = nullptr; p.sk_U = 0; p.sk_dp = 0;
#endif
    p.n_img = (int)rows; p.H = 1; p.W = 1; p.Cin = Cin; p.Ho = 1; p.Wo = 1; p.Cout = Cout; p.KH = 1; p.KW = 1;
    p.stride = 1; p.pad = 0; p.a_img_div = 1; p.relu = 0; p.K = Cin;
    p.ws = nullptr; p.splits = 1; p.kt_per_split = Cin / BK;
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
    p.grp_rows = t_pad; p.grp_valid = n_img * tiles_per_img; p.grp_items = n_img;
    p.grp_rows_per_item = tiles_per_img; p.grp_w_stride = cout_pad * Cin; p.grp_count_dev = n_img_dev;
    p.n_tiles_n = 0;
#ifdef FGN_EXPERIMENTS
    {
        int rc = FGN_OK;
        if (fgn_exp_pointwise(p, rows, 4, true, t_pad, stream, &rc)) return rc;
    }
#endif
    return launch_cfg<64, 64, 32, 32, 4>(p, (int)rows, false, stream);
}
