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
//   LDS       :  [rows][32 + 4 pad] floats; the 144-byte row stride makes both the
//     ds_write_b128 staging stores and the ds_read_b128 fragment reads
//     conflict-free (36*r mod 64 is a bijection on r mod 16).
//   Pipeline  :  register prefetch of K-tile t+1 is issued before the MFMAs of
//     tile t, written to the other LDS buffer after them; one barrier per K-tile.
//   Epilogue  :  y = acc*scale[n] + shift[n] (+ residual) (ReLU)  -- folded
//     eval-mode BN or conv bias -- written straight from the accumulators
//     (each half-wave stores 128 contiguous bytes of one output pixel).
//   Fusions   :  optional per-(image, cin) input scale applied while staging A:
//     the AG-RPN guidance multiply (fgn_ag_rpn_head.py:44) and the mask-head
//     support-vector multiply (fgn_roi_head.py:379) never materialise their
//     [N*B,1024,H,W] product; `a_img_div` lets N guided passes share one query map.
#include "common.h"

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
};

constexpr int BK = 32;
constexpr int LDS_STRIDE = 36;  // floats

template <int BM, int BN, int WM, int WN, bool CIN4, bool IN_SCALE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
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
    const float* a_base[A_LD];
    const float* s_base[A_LD];
    int iy0[A_LD], ix0[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m0 + row0 + 32 * i;
        if (m < M) {
            const int img = m / HoWo;
            const int rem = m - img * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            iy0[i] = oy * p.stride - p.pad;
            ix0[i] = ox * p.stride - p.pad;
            a_base[i] = p.x + (size_t)(img / p.a_img_div) * p.H * p.W * p.Cin;
            s_base[i] = IN_SCALE ? p.in_scale + (size_t)img * p.Cin : nullptr;
        } else {
            iy0[i] = -(1 << 28);  // forces the bounds test to fail
            ix0[i] = -(1 << 28);
            a_base[i] = p.x;
            s_base[i] = p.in_scale;
        }
    }
    const float* b_base = p.w + (size_t)(n0 + row0) * p.K + col4 * 4;

    const int KT = p.K / BK;
    const int cin_tiles = CIN4 ? 1 : p.Cin / BK;

    float4 a_reg[A_LD], b_reg[B_LD];

    auto load_tile = [&](int kt) {
        if (CIN4) {
            // Cin == 4: one float4 is one filter tap; 8 taps per K-tile.
            const int tap = kt * 8 + col4;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            const bool tap_ok = tap < p.KH * p.KW;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const int iy = iy0[i] + ky, ix = ix0[i] + kx;
                const bool ok = tap_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                a_reg[i] = ok ? *reinterpret_cast<const float4*>(a_base[i] + ((size_t)iy * p.W + ix) * 4)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int tap = kt / cin_tiles;
            const int c0 = (kt - tap * cin_tiles) * BK + col4 * 4;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const int iy = iy0[i] + ky, ix = ix0[i] + kx;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    v = *reinterpret_cast<const float4*>(a_base[i] + ((size_t)iy * p.W + ix) * p.Cin + c0);
                    if (IN_SCALE) {
                        const float4 s = *reinterpret_cast<const float4*>(s_base[i] + c0);
                        v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
                    }
                }
                a_reg[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            b_reg[i] = *reinterpret_cast<const float4*>(b_base + (size_t)(32 * i) * p.K + kt * BK);
    };

    auto store_tile = [&](int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + BM * LDS_STRIDE;
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            *reinterpret_cast<float4*>(As + (row0 + 32 * i) * LDS_STRIDE + col4 * 4) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            *reinterpret_cast<float4*>(Bs + (row0 + 32 * i) * LDS_STRIDE + col4 * 4) = b_reg[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) load_tile(kt + 1);

        const float* As = smem + cur * STAGE + (wm * WM + frag_row) * LDS_STRIDE + frag_k;
        const float* Bs = smem + cur * STAGE + BM * LDS_STRIDE + (wn * WN + frag_row) * LDS_STRIDE + frag_k;
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
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: C/D layout col = lane&31 (-> n), row = (r&3)+8*(r>>2)+4*(lane>>5) (-> m)
    const int half = lane >> 5;
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

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const ConvParams& p0, int M_max, bool cin4, hipStream_t stream) {
    ConvParams p = p0;
    p.n_tiles_n = cdiv(p.Cout, BN);
    const int grid = cdiv(M_max, BM) * p.n_tiles_n;
    const size_t lds = 2 * (BM + BN) * LDS_STRIDE * sizeof(float);
    static const hipError_t attr_once = [] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WM, WN, true, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WM, WN, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WM, WN, false, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return e;
    }();
    if (attr_once != hipSuccess) return (int)attr_once;
    if (cin4)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, true, false>), dim3(grid), dim3(256), lds, stream, p);
    else if (p.in_scale)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false, true>), dim3(grid), dim3(256), lds, stream, p);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false, false>), dim3(grid), dim3(256), lds, stream, p);
    FGN_LAUNCH_CHECK();
    return FGN_OK;
}

extern "C" int fgn_conv2d_nhwc_f32(const float* x, const float* w_packed, float* y, const float* scale,
                                   const float* shift, const float* residual, const float* in_scale,
                                   const int32_t* n_img_dev, int n_img, int H, int W, int Cin, int Cout,
                                   int cout_pad, int KH, int KW, int stride, int pad, int a_img_div,
                                   int relu, int tile_hint, hipStream_t stream) {
    if (!x || !w_packed || !y) return FGN_ERR_ARG;
    if (n_img <= 0) return FGN_OK;
    const bool cin4 = (Cin == 4);
    if (!cin4 && (Cin % BK) != 0) return FGN_ERR_SHAPE;
    if (cin4 && in_scale) return FGN_ERR_SHAPE;
    if (a_img_div < 1 || stride < 1 || cout_pad % 128 != 0 || cout_pad < Cout) return FGN_ERR_SHAPE;
    ConvParams p;
    p.x = x; p.w = w_packed; p.y = y; p.scale = scale; p.shift = shift; p.residual = residual;
    p.in_scale = in_scale; p.n_img_dev = n_img_dev;
    p.n_img = n_img; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW;
    p.stride = stride; p.pad = pad; p.a_img_div = a_img_div; p.relu = relu;
    p.Ho = (H + 2 * pad - KH) / stride + 1;
    p.Wo = (W + 2 * pad - KW) / stride + 1;
    if (p.Ho <= 0 || p.Wo <= 0) return FGN_ERR_SHAPE;
    const int k_raw = KH * KW * Cin;
    p.K = cdiv(k_raw, BK) * BK;
    const long long M = (long long)n_img * p.Ho * p.Wo;
    if (M * (long long)Cout >= (1ll << 31) * 4) return FGN_ERR_SHAPE;

    // tile choice: the largest tile that still yields >= ~2 blocks per CU, else smaller.
    int tile = tile_hint;
    if (tile == 0) {
        const long long b128 = ((M + 127) / 128) * cdiv(Cout, 128);
        const long long b64x128 = ((M + 63) / 64) * cdiv(Cout, 128);
        if (Cout <= 64) tile = 3;
        else if (b128 >= 384) tile = 1;
        else if (b64x128 >= 384) tile = 2;
        else tile = 4;
    }
    switch (tile) {
        case 1: return launch_cfg<128, 128, 64, 64>(p, (int)M, cin4, stream);
        case 2: return launch_cfg<64, 128, 32, 64>(p, (int)M, cin4, stream);
        case 3: return launch_cfg<128, 64, 64, 32>(p, (int)M, cin4, stream);
        case 4: return launch_cfg<64, 64, 32, 32>(p, (int)M, cin4, stream);
        default: return FGN_ERR_ARG;
    }
}
