// In-kernel clock of the GEMM kernels under sustained load (MI355X_MICROARCH.md "DVFS give-back" item 6): a diagnostic
// build of csrc/conv_igemm.hip (-DCONV_CLOCK_STAMPS: one s_memtime / s_memrealtime pair around each workgroup's work,
// written to a buffer nothing else reads), >= 2.5 s of back-to-back launches of ONE GEMM shape on random data, then
//   clock  = median over workgroups of  d(s_memtime) / d(s_memrealtime) * 100 MHz
//   pipe   = MFMA issue cycles per SIMD (32 per v_mfma_f32_16x16x4_f32, 64 per 32x32x2) / median workgroup span in
//            shader cycles of the LAST launch - the share of the span in which a SIMD's matrix pipe was issuing
// build (here or on the box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCONV_CLOCK_STAMPS -mllvm -amdgpu-atomic-optimizer-strategy=None tools/micro/gemm_clock.hip -o tools/micro/gemm_clock
// usage: gemm_clock [seconds]      prints one JSON object per (shape, kernel variant)
#include "../../fgn_amd/csrc/conv_igemm.hip"
#include "../../fgn_amd/csrc/abi.hip"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static float* dev_random(size_t n, float scale, unsigned seed) {
    std::vector<float> h(n);
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    for (auto& v : h) v = d(g) * scale;
    float* p = nullptr;
    if (hipMalloc(&p, n * 4) != hipSuccess) return nullptr;
    (void)hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice);
    return p;
}

struct Shape { const char* name; bool grouped; int n, tiles, rows, cin, cout; };

static int n_variants = 1;
static int run(const Shape& sh, int variant, double seconds) {
    hipStream_t st; CK(hipStreamCreate(&st));
    float *A = nullptr, *W = nullptr, *Y = nullptr;
    int rc = 0;
    double flop = 0;
    const int cout_pad = (sh.cout + 127) / 128 * 128;
    int t_pad = 0;
    if (sh.grouped) {
        t_pad = fgn_winograd_t_pad(sh.n * sh.tiles);
        A = dev_random((size_t)36 * t_pad * sh.cin, 1.f, 1);
        W = dev_random((size_t)36 * cout_pad * sh.cin, 0.03f, 2);
        CK(hipMalloc(&Y, (size_t)36 * t_pad * sh.cout * 4));
        flop = 2.0 * 36 * sh.n * sh.tiles * sh.cin * sh.cout;
    } else {
        A = dev_random((size_t)sh.rows * sh.cin, 1.f, 1);
        W = dev_random((size_t)cout_pad * sh.cin, 0.03f, 2);
        CK(hipMalloc(&Y, (size_t)sh.rows * sh.cout * 4));
        flop = 2.0 * sh.rows * sh.cin * sh.cout;
    }
    if (!A || !W) return 1;
    // 1001..1003: conv_pw_streamk_kernel mode 1..3; 1013: mode 3 with the pieces dropped (timing only, wrong results)
#ifdef FGN_EXPERIMENTS     // the kernel variants of tools/micro/conv_pw_experiments.inc (build with -DFGN_EXPERIMENTS)
    // 1..48: conv_pw_persist2_kernel tile codes; 1001..1013: Stream-K modes; 2001: the producer-wave form of the persistent kernel
    fgn_conv2d_tune(0, variant >= 1000 ? 0 : variant);
    fgn_conv2d_tune(2, variant >= 1000 && variant < 2000 ? (variant - 1000) % 10 : 0);
    fgn_conv2d_tune(5, variant >= 1000 && variant < 2000 ? (variant - 1000) / 10 : 0);
    fgn_conv2d_tune(6, variant == 2001 ? 1 : 0);
#else
    if (variant != 0) { fprintf(stderr, "variant %d needs a -DFGN_EXPERIMENTS build\n", variant); return 1; }
#endif
    float* skws = nullptr;
    const size_t skws_bytes = (size_t)64 << 20;
    CK(hipMalloc(&skws, skws_bytes));
    auto launch = [&]() -> int {
        return sh.grouped ? fgn_winograd_gemm_f32(A, W, Y, nullptr, sh.n, sh.tiles, t_pad, sh.cin, sh.cout, cout_pad, 36, st)
                          : fgn_conv2d_nhwc_f32(A, W, Y, nullptr, nullptr, nullptr, nullptr, nullptr, sh.rows, 1, 1, sh.cin, sh.cout,
                                                cout_pad, 1, 1, 1, 0, 1, 0, 0, skws, skws_bytes, st);
    };
    for (int i = 0; i < 3; ++i) rc |= launch();
    CK(hipStreamSynchronize(st));
    if (rc) { fprintf(stderr, "launch rc %d\n", rc); return 1; }
    // >= `seconds` of back-to-back launches; the stamps of the last launch are read
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double last_batch_ms = 0; int batch = 50;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < batch; ++i) launch();
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        last_batch_ms = ms; launches += batch;
    }
    std::vector<unsigned long long> h(16384 * 6);
    CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_clock_stamps), h.size() * 8));
    struct WgStamp { double clk, span, r0, r1; unsigned xcc, cu; };
    std::vector<WgStamp> ws;
    unsigned long long rmin = ~0ull, rmax = 0;
    for (int b = 0; b < 16384; ++b) {
        const unsigned long long t0s = h[b * 6], t1s = h[b * 6 + 1], r0 = h[b * 6 + 2], r1 = h[b * 6 + 3];
        if (t1s > t0s && r1 > r0 && r1 - r0 > 200) {       // >= 2 us of work
            const unsigned xcc = (unsigned)h[b * 6 + 4] & 0xf, hw = (unsigned)h[b * 6 + 5];
            // HW_ID (gfx9): cu_id [11:8], sh_id [12], se_id [15:13]
            ws.push_back({(double)(t1s - t0s) / (double)(r1 - r0) * 0.1, (double)(t1s - t0s), (double)r0, (double)r1, xcc,
                          (xcc << 8) | ((hw >> 8) & 0xff)});
            rmin = std::min(rmin, r0); rmax = std::max(rmax, r1);
        }
    }
    const std::vector<unsigned long long> raw = h;
    std::fill(h.begin(), h.end(), 0ull);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_clock_stamps), h.data(), h.size() * 8));
    if (ws.empty()) { fprintf(stderr, "%s v%d: no stamps\n", sh.name, variant); return 1; }
    auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[std::min(v.size() - 1, (size_t)(f * v.size()))]; };
    std::vector<double> clk, span, start_us, end_us;
    for (auto& w : ws) { clk.push_back(w.clk); span.push_back(w.span); start_us.push_back((w.r0 - (double)rmin) * 0.01); end_us.push_back(((double)rmax - w.r1) * 0.01); }
    if (getenv("GEMM_CLOCK_PLACEMENT")) {      // block -> (XCC, CU key) of the first blocks of XCD share 0 (blocks 0, 8, 16, ...)
        fprintf(stderr, "%s v%d placement of blocks b = 8 i (xcc:cu):", sh.name, variant);
        for (int i = 0; i < 48; ++i) {
            const int b = 8 * i;
            fprintf(stderr, " %llu:%02llx", raw[b * 6 + 4] & 0xf, (raw[b * 6 + 5] >> 8) & 0xff);
        }
        fprintf(stderr, "\n  blocks 0..15:");
        for (int b = 0; b < 16; ++b) fprintf(stderr, " %llu:%02llx", raw[b * 6 + 4] & 0xf, (raw[b * 6 + 5] >> 8) & 0xff);
        fprintf(stderr, "\n");
    }
    if (getenv("GEMM_CLOCK_TAIL_BLOCKS")) {
        // the workgroups that own one output tile more than the others (blocks < R of a fixed walk): on how many CUs do
        // they sit, and how much longer is their span?  R: comma-separated, one per shape, 0 = skip
        static int shape_no = 0;
        std::vector<int> rs;
        for (const char* c = getenv("GEMM_CLOCK_TAIL_BLOCKS"); *c;) { rs.push_back(atoi(c)); while (*c && *c != ',') ++c; if (*c) ++c; }
        const int R = rs.empty() ? 0 : rs[std::min<size_t>(shape_no / (int)std::max<size_t>(1, (size_t)n_variants), rs.size() - 1)];
        ++shape_no;
        if (R > 0) {
            std::vector<unsigned> tail_cu; std::vector<double> s_tail, s_rest;
            for (int b = 0; b < 16384; ++b) {
                const unsigned long long t0s = raw[b * 6], t1s = raw[b * 6 + 1];
                if (t1s <= t0s) continue;
                const unsigned key = (((unsigned)raw[b * 6 + 4] & 0xf) << 8) | (((unsigned)raw[b * 6 + 5] >> 8) & 0xff);
                if (b < R) { tail_cu.push_back(key); s_tail.push_back((double)(t1s - t0s)); } else s_rest.push_back((double)(t1s - t0s));
            }
            std::sort(tail_cu.begin(), tail_cu.end());
            int th[8] = {0}; size_t ncu = 0;
            for (size_t i = 0; i < tail_cu.size();) { size_t j = i; while (j < tail_cu.size() && tail_cu[j] == tail_cu[i]) ++j; th[std::min<size_t>(j - i, 7)]++; ++ncu; i = j; }
            fprintf(stderr, "%s v%d: blocks < %d sit on %zu CUs (per CU: 1:%d 2:%d 3:%d 4:%d 5+:%d); span p50 %0.f cycles vs %0.f of the other %zu blocks\n",
                    sh.name, variant, R, ncu, th[1], th[2], th[3], th[4], th[5] + th[6] + th[7], s_tail.empty() ? 0. : q(s_tail, 0.5),
                    s_rest.empty() ? 0. : q(s_rest, 0.5), s_rest.size());
        }
    }
    // workgroups per CU (as placed in the last launch), and the median span of a workgroup by how crowded its CU was
    std::vector<unsigned> cus; for (auto& w : ws) cus.push_back(w.cu);
    std::sort(cus.begin(), cus.end());
    int hist[12] = {0}; size_t n_cu = 0;
    for (size_t i = 0; i < cus.size();) { size_t j = i; while (j < cus.size() && cus[j] == cus[i]) ++j; hist[std::min<size_t>(j - i, 11)]++; ++n_cu; i = j; }
    char hbuf[256]; int hp = 0; hbuf[0] = 0;
    for (int k = 1; k < 12; ++k) if (hist[k]) hp += snprintf(hbuf + hp, sizeof(hbuf) - hp, "%s\"%d\": %d", hp ? ", " : "", k, hist[k]);
    double xcd_span[8] = {0}; char xbuf[256]; int xp = 0; xbuf[0] = 0;
    for (unsigned x = 0; x < 8; ++x) { std::vector<double> v; for (auto& w : ws) if (w.xcc == x) v.push_back(w.span); xcd_span[x] = v.empty() ? 0 : q(v, 0.5); xp += snprintf(xbuf + xp, sizeof(xbuf) - xp, "%s%.0f", x ? ", " : "", xcd_span[x]); }
    const double us = last_batch_ms * 1e3 / batch;
    const double mfma_cycles_per_simd = flop / 64.0 / 1024.0;     // flop / (64 FLOP per cycle per SIMD) / 1024 SIMDs
    const double wall_cycles = us * 1e-6 * q(clk, 0.5) * 1e9;
    printf("{\"shape\": \"%s\", \"variant\": %d, \"launches\": %ld, \"us_per_launch_back_to_back\": %.1f, \"tflops\": %.1f, "
           "\"frac_of_157.3\": %.3f, \"clock_ghz_median\": %.3f, \"clock_ghz_p10\": %.3f, \"clock_ghz_p90\": %.3f, "
           "\"workgroups_stamped\": %zu, \"cus_seen\": %zu, \"workgroups_per_cu_histogram\": {%s}, "
           "\"wg_span_cycles\": {\"p10\": %.0f, \"p50\": %.0f, \"p90\": %.0f, \"max\": %.0f}, \"wg_span_p50_by_xcd\": [%s], "
           "\"wg_start_after_first_us\": {\"p50\": %.1f, \"p90\": %.1f, \"max\": %.1f}, \"wg_end_before_last_us\": {\"p10\": %.1f, \"p50\": %.1f, \"p90\": %.1f}, "
           "\"first_start_to_last_end_us\": %.1f, \"launch_wall_cycles_at_median_clock\": %.0f, "
           "\"mfma_issue_cycles_per_simd\": %.0f, \"mfma_pipe_share_of_wall\": %.3f, \"mfma_pipe_share_of_wg_span_p50\": %.3f}\n",
           sh.name, variant, launches, us, flop / us / 1e6, flop / us / 1e6 / 157.3, q(clk, 0.5), q(clk, 0.1), q(clk, 0.9), ws.size(), n_cu, hbuf,
           q(span, 0.1), q(span, 0.5), q(span, 0.9), q(span, 0.9999), xbuf, q(start_us, 0.5), q(start_us, 0.9), q(start_us, 0.9999),
           q(end_us, 0.1), q(end_us, 0.5), q(end_us, 0.9), ((double)rmax - (double)rmin) * 0.01, wall_cycles, mfma_cycles_per_simd,
           mfma_cycles_per_simd / wall_cycles, mfma_cycles_per_simd / q(span, 0.5));
    fflush(stdout);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(Y); (void)hipStreamDestroy(st);
    return 0;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.5;
    const char* vlist = argc > 2 ? argv[2] : "0,1";
    std::vector<int> variants;
    for (const char* c = vlist; *c;) { variants.push_back(atoi(c)); while (*c && *c != ',') ++c; if (*c) ++c; }
    const Shape shapes[] = {
        {"wino agrpn 36x(3x273) 1024>1024", true, 3, 273, 0, 1024, 1024},
        {"wino sh300 36x(300x4) 512>512", true, 300, 4, 0, 512, 512},
        {"relq 14700x1024>1024", false, 0, 0, 14700, 1024, 1024},
        {"sh conv1 14700x1024>512 (r3: 128x128 DMA kernel)", false, 0, 0, 14700, 1024, 512},
        {"perfect 16384x1024>1024 (4 whole rounds)", false, 0, 0, 16384, 1024, 1024},
    };
    n_variants = (int)variants.size();
    const char* only = getenv("GEMM_CLOCK_SHAPES");        // substring of the shape names to run
    int rc = 0;
    for (const auto& sh : shapes)
        if (!only || strstr(sh.name, only))
            for (int v : variants) rc |= run(sh, v, seconds);
    return rc;
}
