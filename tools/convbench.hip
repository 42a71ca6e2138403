// Standalone micro-benchmark + correctness check for the conv kernels of
// libsemseg_hip.so (no Python / torch: starts in milliseconds on the GPU box).
//   tools/bin/convbench [iters]
// For every shape of the HRNet-OCR-MScale census it runs
// the implicit-GEMM kernel (auto tile and, with -a, every tile config), the
// halo-tile kernel where supported, checks both against a naive direct
// convolution and prints time, TFLOP/s and algorithmic GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include "../include/semseg_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned short bf16_t;
static inline bf16_t f2bf_h(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static inline float bf2f_h(bf16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void ref_conv(const bf16_t* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin,
                         int ldx, int Ho, int Wo, int Cout, int K, int stride, int pad, int dil, int Cin_w) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  long n = (long)B * Ho * Wo * Cout;
  if (i >= n) return;
  int co = i % Cout; long p = i / Cout;
  int ox = p % Wo; p /= Wo; int oy = p % Ho; int b = p / Ho;
  float acc = bias ? bias[co] : 0.f;
  for (int kh = 0; kh < K; ++kh) for (int kw = 0; kw < K; ++kw) {
    int iy = oy * stride - pad + kh * dil, ix = ox * stride - pad + kw * dil;
    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
    const bf16_t* xp = x + ((long)(b * H + iy) * W + ix) * ldx;
    for (int ci = 0; ci < Cin_w; ++ci) {
      float wv = w[(((long)co * Cin_w + ci) * K + kh) * K + kw];
      uint32_t wu = __float_as_uint(wv); wu += 0x7fffu + ((wu >> 16) & 1u); wu &= 0xffff0000u;
      acc += __uint_as_float(((uint32_t)xp[ci]) << 16) * __uint_as_float(wu);
    }
  }
  y[i] = acc;
}

__global__ void cmp_bf16(const bf16_t* y, int ldy, const float* ref, long P, int C, float* out) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= P * C) return;
  long p = i / C; int c = i % C;
  float a = __uint_as_float(((uint32_t)y[p * ldy + c]) << 16), r = ref[i];
  float e = fabsf(a - r);
  atomicMax((int*)&out[0], __float_as_int(e));
  atomicMax((int*)&out[1], __float_as_int(fabsf(r)));
}

__global__ void ref_wgrad(const bf16_t* x, const bf16_t* dy, float* dw, int B, int H, int W, int Cin, int Cout) {
  // dw[co][ci][kh][kw], 3x3 stride 1 pad 1; one thread per element
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  long n = (long)Cout * Cin * 9;
  if (i >= n) return;
  int kw = i % 3, kh = (i / 3) % 3, ci = (i / 9) % Cin, co = i / (9L * Cin);
  float acc = 0.f;
  for (int b = 0; b < B; ++b)
    for (int oy = 0; oy < H; ++oy) {
      int iy = oy + kh - 1;
      if (iy < 0 || iy >= H) continue;
      for (int ox = 0; ox < W; ++ox) {
        int ix = ox + kw - 1;
        if (ix < 0 || ix >= W) continue;
        acc += __uint_as_float(((uint32_t)dy[((long)(b * H + oy) * W + ox) * Cout + co]) << 16) *
               __uint_as_float(((uint32_t)x[((long)(b * H + iy) * W + ix) * Cin + ci]) << 16);
      }
    }
  dw[i] = acc;
}

__global__ void cmp_f32(const float* a, const float* r, long n, float* out) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  atomicMax((int*)&out[0], __float_as_int(fabsf(a[i] - r[i])));
  atomicMax((int*)&out[1], __float_as_int(fabsf(r[i])));
}

struct Case { int B, H, W, Cin, Cout, K, stride; const char* name; };

int main(int argc, char** argv) {
  int iters = 20; bool all = false;
  for (int i = 1; i < argc; ++i) { if (!strcmp(argv[i], "-a")) all = true; else iters = atoi(argv[i]); }
  std::vector<Case> cases = {
    {1, 256, 256, 48, 48, 3, 1, "branch0 1.0x"}, {1, 128, 128, 48, 48, 3, 1, "branch0 0.5x"},
    {1, 128, 128, 96, 96, 3, 1, "branch1 1.0x"}, {1, 64, 64, 96, 96, 3, 1, "branch1 0.5x"},
    {1, 64, 64, 192, 192, 3, 1, "branch2 1.0x"}, {1, 32, 32, 192, 192, 3, 1, "branch2 0.5x"},
    {1, 32, 32, 384, 384, 3, 1, "branch3 1.0x"}, {1, 16, 16, 384, 384, 3, 1, "branch3 0.5x"},
    {1, 256, 256, 64, 64, 3, 1, "layer1 conv2"}, {1, 256, 256, 720, 512, 3, 1, "conv3x3_ocr"},
    {1, 128, 128, 720, 512, 3, 1, "conv3x3_ocr 0.5x"}, {1, 256, 256, 512, 256, 3, 1, "attn conv0"},
    {1, 256, 256, 256, 256, 3, 1, "attn conv1"}, {1, 256, 256, 1024, 512, 1, 1, "conv_bn_dropout"},
    {1, 256, 256, 720, 720, 1, 1, "aux_head.0"}, {1, 256, 256, 512, 256, 1, 1, "f_pixel.0"},
    {1, 256, 256, 64, 256, 1, 1, "layer1 conv3"}, {1, 256, 256, 48, 96, 3, 2, "fuse down 48-96"},
    {1, 37, 45, 48, 48, 3, 1, "ragged 48"}, {2, 40, 24, 96, 96, 3, 1, "ragged 96 B2"},
    {1, 20, 12, 64, 64, 3, 1, "ragged 64"}, {2, 21, 45, 192, 192, 3, 1, "ragged 192 B2"},
    {1, 9, 33, 384, 384, 3, 1, "ragged 384"}, {1, 130, 131, 200, 136, 3, 1, "ragged head 3x3"},
    {2, 100, 97, 264, 72, 1, 1, "ragged head 1x1"},
  };
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = ssa_bn_stat_replicas();
  printf("%-22s %-14s %9s %9s %9s %10s\n", "case", "kernel", "us", "TFLOP/s", "GB/s", "maxerr/ref");
  for (const Case& c : cases) {
    const int pad = c.K / 2;
    const int Ho = (c.H + 2 * pad - c.K) / c.stride + 1, Wo = (c.W + 2 * pad - c.K) / c.stride + 1;
    const long Pin = (long)c.B * c.H * c.W, Pout = (long)c.B * Ho * Wo;
    std::vector<bf16_t> hx(Pin * c.Cin);
    std::vector<float> hw((long)c.Cout * c.Cin * c.K * c.K);
    srand(1234);
    for (auto& v : hx) v = f2bf_h((rand() / (float)RAND_MAX) * 2.f - 1.f);
    const float ws = 1.f / sqrtf((float)c.Cin * c.K * c.K);
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * ws;
    bf16_t *dx, *dy; float *dw, *dref, *derr; void* dwp; double* dstats;
    CK(hipMalloc(&dx, hx.size() * 2)); CK(hipMalloc(&dy, Pout * c.Cout * 2 + 64));
    CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dref, Pout * c.Cout * 4)); CK(hipMalloc(&derr, 8));
    const int Kflat = c.K * c.K * c.Cin, Kpad = (Kflat + 31) / 32 * 32;
    const long rows_pad = (c.Cout + 31) / 32 * 32;
    CK(hipMalloc(&dwp, (size_t)rows_pad * Kpad * 2)); CK(hipMalloc(&dstats, sizeof(double) * reps * 2 * c.Cout));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    const long n = Pout * c.Cout;
    hipLaunchKernelGGL(ref_conv, dim3((n + 255) / 256), dim3(256), 0, st, dx, dw, (const float*)nullptr, dref, c.B, c.H,
                       c.W, c.Cin, c.Cin, Ho, Wo, c.Cout, c.K, c.stride, pad, 1, c.Cin);
    CK(hipStreamSynchronize(st));
    const double flops = 2.0 * Pout * c.Cout * Kflat;
    const double bytes = 2.0 * (Pin * c.Cin + Pout * c.Cout) + 2.0 * c.Cout * Kflat;
    ssa_conv_desc d = {c.B, c.H, c.W, c.Cin, c.Cin, Ho, Wo, c.Cout, c.Cout, c.K, c.K, c.stride, pad, 1, 0, Kpad, 0, -1};
    auto check = [&](const char* kname, float us) {
      CK(hipMemsetAsync(derr, 0, 8, st));
      hipLaunchKernelGGL(cmp_bf16, dim3((n + 255) / 256), dim3(256), 0, st, dy, c.Cout, dref, Pout, c.Cout, derr);
      float herr[2]; CK(hipMemcpyAsync(herr, derr, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
      printf("%-22s %-14s %9.2f %9.1f %9.1f %10.5f%s\n", c.name, kname, us, flops / us * 1e-6, bytes / us * 1e-3,
             herr[0] / (herr[1] + 1e-30f), herr[0] / (herr[1] + 1e-30f) > 0.02f ? "  <-- MISMATCH" : "");
    };
    // `iters` launches captured into one hipGraph: GPU-side time per launch without
    // the host's launch rate (8 us per eager launch on the test box) in the way
    auto timeit = [&](auto&& fn) -> float {
      for (int i = 0; i < 2; ++i) fn();
      CK(hipStreamSynchronize(st));
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int i = 0; i < iters; ++i) fn();
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
      return ms * 1e3f / iters;
    };
    // implicit GEMM
    if (ssa_pack_filter(dw, dwp, c.Cout, c.Cin, c.K, c.K, c.Cin, 0, Kpad, 0, st)) { printf("pack failed\n"); return 1; }
    for (int cfg = -1; cfg < (all ? 6 : 0); ++cfg) {
      d.cfg = cfg;
      CK(hipMemsetAsync(dy, 0, Pout * c.Cout * 2, st));
      int rc = 0;
      float us = timeit([&] { rc |= ssa_conv2d_igemm(&d, dx, dwp, nullptr, dy, st); });
      if (rc) { printf("%-22s igemm cfg %d failed rc=%d\n", c.name, cfg, rc); continue; }
      char nm[32]; snprintf(nm, sizeof nm, "igemm cfg%d(%d)", cfg, ssa_conv2d_igemm_tile(&d));
      check(nm, us);
    }
    d.cfg = -1;
    if (ssa_conv2d_tile_supported(&d)) {
      if (ssa_pack_filter(dw, dwp, c.Cout, c.Cin, c.K, c.K, c.Cin, 0, Kflat, 2, st)) { printf("pack2 failed\n"); return 1; }
      int rc = 0;
      float us = 0;
      for (int cfg = 0; cfg < 4; ++cfg) {
        d.cfg = cfg;
        CK(hipMemsetAsync(dy, 0, Pout * c.Cout * 2, st));
        rc = 0;
        us = timeit([&] { rc |= ssa_conv2d_tile(&d, dx, dwp, nullptr, dy, nullptr, st); });
        char nm[32]; snprintf(nm, sizeof nm, "tile cfg%d", cfg);
        if (rc) printf("%-22s %s failed rc=%d\n", c.name, nm, rc); else check(nm, us);
      }
      d.cfg = -1;
      CK(hipMemsetAsync(dstats, 0, sizeof(double) * reps * 2 * c.Cout, st));
      CK(hipMemsetAsync(dy, 0, Pout * c.Cout * 2, st));
      us = timeit([&] { rc |= ssa_conv2d_tile(&d, dx, dwp, nullptr, dy, dstats, st); });
      if (!rc) check("tile+stats", us);
      // statistics check: sum over replicas / launches vs host sum of y
      std::vector<double> hs((size_t)reps * 2 * c.Cout);
      std::vector<bf16_t> hy(Pout * c.Cout);
      CK(hipMemcpy(hs.data(), dstats, hs.size() * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost));
      double worst = 0;
      for (int ch = 0; ch < c.Cout; ++ch) {
        double s = 0, q = 0, gs = 0, gq = 0;
        for (long p = 0; p < Pout; ++p) { double v = bf2f_h(hy[p * c.Cout + ch]); s += v; q += v * v; }
        for (int r = 0; r < reps; ++r) { gs += hs[(size_t)r * 2 * c.Cout + ch]; gq += hs[(size_t)r * 2 * c.Cout + c.Cout + ch]; }
        gs /= (2 * iters + 2); gq /= (2 * iters + 2);
        worst = fmax(worst, fabs(gs - s) / (fabs(s) + 1e-3 * Pout));
        worst = fmax(worst, fabs(gq - q) / (fabs(q) + 1e-9));
      }
      printf("%-22s %-14s stats worst rel err %.3g%s\n", c.name, "tile+stats", worst, worst > 1e-3 ? "  <-- MISMATCH" : "");
      if (c.Cin == c.Cout) {
        // the same launch with a fused epilogue tile (conv_tile_aux.hip), as the data gradients of a residual
        // block use it: what the extra tile costs per launch.  Correctness: tests/test_fuse_bwd_gpu.py.
        std::vector<bf16_t> haux(Pout * c.Cout);
        for (auto& v : haux) v = f2bf_h((rand() / (float)RAND_MAX) * 2.f - 1.f);
        std::vector<float> hcoef(4 * (size_t)c.Cout);
        for (int ch = 0; ch < c.Cout; ++ch) {
          hcoef[ch] = (rand() / (float)RAND_MAX) * 2.f - 1.f;                       // mask scale
          hcoef[c.Cout + ch] = (rand() / (float)RAND_MAX) - 0.5f;                   // mask shift
          hcoef[2 * c.Cout + ch] = ((rand() / (float)RAND_MAX) - 0.5f) * 0.2f;      // mean
          hcoef[3 * c.Cout + ch] = 0.5f + (rand() / (float)RAND_MAX);               // invstd
        }
        bf16_t* daux; float* dcoef;
        CK(hipMalloc(&daux, haux.size() * 2)); CK(hipMalloc(&dcoef, hcoef.size() * 4));
        CK(hipMemcpy(daux, haux.data(), haux.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dcoef, hcoef.data(), hcoef.size() * 4, hipMemcpyHostToDevice));
        rc = 0;
        us = timeit([&] { rc |= ssa_conv2d_tile_aux(&d, dx, dwp, nullptr, dy, nullptr, daux, c.Cout, nullptr, 1, st); });
        if (rc) printf("%-22s tile+add failed rc=%d\n", c.name, rc);
        else printf("%-22s %-14s %9.2f %9.1f %9.1f\n", c.name, "tile+add", us, flops / us * 1e-6, (bytes + 2.0 * Pout * c.Cout) / us * 1e-3);
        CK(hipMemsetAsync(dstats, 0, sizeof(double) * reps * 2 * c.Cout, st));
        rc = 0;
        us = timeit([&] { rc |= ssa_conv2d_tile_aux(&d, dx, dwp, nullptr, dy, dstats, daux, c.Cout, dcoef, 2, st); });
        if (rc) printf("%-22s tile+bnsums failed rc=%d\n", c.name, rc);
        else printf("%-22s %-14s %9.2f %9.1f %9.1f\n", c.name, "tile+bnsums", us, flops / us * 1e-6, (bytes + 2.0 * Pout * c.Cout) / us * 1e-3);
        CK(hipFree(daux)); CK(hipFree(dcoef));
      }
    }
    if (ssa_conv2d_halo_supported(&d)) {
      if (ssa_pack_filter(dw, dwp, c.Cout, c.Cin, c.K, c.K, c.Cin, 0, Kflat, 2, st)) { printf("pack2 failed\n"); return 1; }
      CK(hipMemsetAsync(dy, 0, Pout * c.Cout * 2, st));
      int rc = 0;
      float us = timeit([&] { rc |= ssa_conv2d_halo(&d, dx, dwp, nullptr, dy, nullptr, st); });
      if (rc) printf("%-22s halo failed rc=%d\n", c.name, rc); else check("halo", us);
    }
    // ---- weight gradient: K-pipelined kernel vs halo-staged tile kernel, both checked against a naive sum
    {
      int ns_t = 0; size_t ws_t = 0;
      if (c.K == 3 && c.stride == 1 && c.Cin == c.Cout && ssa_conv2d_wgrad_tile_plan(&d, c.Cout, &ns_t, &ws_t) == 0) {
        std::vector<bf16_t> hdy(Pout * c.Cout);
        for (auto& v : hdy) v = f2bf_h((rand() / (float)RAND_MAX) * 2.f - 1.f);
        bf16_t* ddy; float *dwr, *dwo, *part; float* derr2;
        const long nw = (long)c.Cout * c.Cin * 9;
        int ns_o = 0; size_t ws_o = 0;
        ssa_conv2d_wgrad_plan(&d, c.Cout, &ns_o, &ws_o);
        CK(hipMalloc(&ddy, hdy.size() * 2)); CK(hipMalloc(&dwr, nw * 4)); CK(hipMalloc(&dwo, nw * 4));
        CK(hipMalloc(&part, ws_o > ws_t ? ws_o : ws_t)); CK(hipMalloc(&derr2, 8));
        CK(hipMemcpy(ddy, hdy.data(), hdy.size() * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(ref_wgrad, dim3((nw + 255) / 256), dim3(256), 0, st, dx, ddy, dwr, c.B, c.H, c.W, c.Cin, c.Cout);
        auto checkw = [&](const char* nm, float us) {
          CK(hipMemsetAsync(derr2, 0, 8, st));
          hipLaunchKernelGGL(cmp_f32, dim3((nw + 255) / 256), dim3(256), 0, st, dwo, dwr, nw, derr2);
          float he[2]; CK(hipMemcpyAsync(he, derr2, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
          printf("%-22s %-14s %9.2f %9.1f %9s %10.5f%s\n", c.name, nm, us, flops / us * 1e-6, "-", he[0] / (he[1] + 1e-30f),
                 he[0] / (he[1] + 1e-30f) > 5e-3f ? "  <-- MISMATCH" : "");
        };
        int rc = 0;
        float us = timeit([&] { rc |= ssa_conv2d_wgrad(&d, dx, ddy, c.Cout, c.Cout, ns_o, part, st);
                                rc |= ssa_conv2d_wgrad_reduce(part, ns_o, c.Cout, c.Cout, c.Cin, c.Cin, 3, 3, dwo, 0, st); });
        char nm[40]; snprintf(nm, sizeof nm, "wgrad old s%d", ns_o);
        if (rc) printf("%-22s wgrad old failed rc=%d\n", c.name, rc); else checkw(nm, us);
        CK(hipMemsetAsync(dwo, 0, nw * 4, st));
        rc = 0;
        us = timeit([&] { rc |= ssa_conv2d_wgrad_tile(&d, dx, ddy, c.Cout, c.Cout, ns_t, part, st);
                          rc |= ssa_conv2d_wgrad_reduce(part, ns_t, c.Cout, c.Cout, c.Cin, c.Cin, 3, 3, dwo, 0, st); });
        snprintf(nm, sizeof nm, "wgrad tile s%d", ns_t);
        if (rc) printf("%-22s wgrad tile failed rc=%d\n", c.name, rc); else checkw(nm, us);
        CK(hipFree(ddy)); CK(hipFree(dwr)); CK(hipFree(dwo)); CK(hipFree(part)); CK(hipFree(derr2));
      }
    }
    // ---- head weight gradient (large channels): new persistent kernel vs the K-pipelined one
    {
      int ns_h = 0; size_t ws_h = 0;
      if (c.stride == 1 && ssa_conv2d_wgrad_head_plan(&d, c.Cout, &ns_h, &ws_h) == 0) {
        std::vector<bf16_t> hdy(Pout * c.Cout);
        for (auto& v : hdy) v = f2bf_h((rand() / (float)RAND_MAX) * 2.f - 1.f);
        bf16_t* ddy; float *dw_old, *dw_new, *part, *derr2;
        const long nw = (long)c.Cout * c.Cin * c.K * c.K;
        int ns_o = 0; size_t ws_o = 0;
        ssa_conv2d_wgrad_plan(&d, c.Cout, &ns_o, &ws_o);
        CK(hipMalloc(&ddy, hdy.size() * 2)); CK(hipMalloc(&dw_old, nw * 4)); CK(hipMalloc(&dw_new, nw * 4));
        CK(hipMalloc(&part, ws_o > ws_h ? ws_o : ws_h)); CK(hipMalloc(&derr2, 8));
        CK(hipMemcpy(ddy, hdy.data(), hdy.size() * 2, hipMemcpyHostToDevice));
        int rc = 0;
        float us_old = timeit([&] { rc |= ssa_conv2d_wgrad(&d, dx, ddy, c.Cout, c.Cout, ns_o, part, st);
                                    rc |= ssa_conv2d_wgrad_reduce(part, ns_o, c.Cout, c.Cout, c.Cin, c.Cin, c.K, c.K, dw_old, 0, st); });
        if (rc) printf("%-22s wgrad old failed rc=%d\n", c.name, rc);
        rc = 0;
        float us_new = timeit([&] { rc |= ssa_conv2d_wgrad_head(&d, dx, ddy, c.Cout, c.Cout, ns_h, part, st);
                                    rc |= ssa_conv2d_wgrad_reduce(part, ns_h, c.Cout, c.Cout, c.Cin, c.Cin, c.K, c.K, dw_new, 0, st); });
        CK(hipMemsetAsync(derr2, 0, 8, st));
        hipLaunchKernelGGL(cmp_f32, dim3((nw + 255) / 256), dim3(256), 0, st, dw_new, dw_old, nw, derr2);
        float he[2]; CK(hipMemcpyAsync(he, derr2, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        printf("%-22s wgrad old s%-4d  %9.2f %9.1f\n", c.name, ns_o, us_old, flops / us_old * 1e-6);
        if (rc) printf("%-22s wgrad head failed rc=%d\n", c.name, rc);
        else printf("%-22s wgrad head s%-3d  %9.2f %9.1f   max|new-old|/max|old| %.6f%s\n", c.name, ns_h, us_new,
                    flops / us_new * 1e-6, he[0] / (he[1] + 1e-30f), he[0] / (he[1] + 1e-30f) > 2e-3f ? "  <-- MISMATCH" : "");
        CK(hipFree(ddy)); CK(hipFree(dw_old)); CK(hipFree(dw_new)); CK(hipFree(part)); CK(hipFree(derr2));
      }
    }
    CK(hipFree(dx)); CK(hipFree(dy)); CK(hipFree(dw)); CK(hipFree(dref)); CK(hipFree(derr)); CK(hipFree(dwp)); CK(hipFree(dstats));
  }
  return 0;
}
