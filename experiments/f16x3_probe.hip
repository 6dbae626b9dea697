// Numerics probe (not part of the product): how accurate is an fp32 GEMM emulated with 3 fp16 MFMAs on hi/lo splits?
// Build: hipcc --offload-arch=gfx950 -O3 -o f16x3_probe f16x3_probe.hip ; run on the MI355X.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// one wave: C[32][32] = A[32][K] . B[32][K]^T, four ways
__global__ void probe(const float* A, const float* B, int K, float* C32, float* C16, float* C16x3, float* C16x3u,
                      float scale_lo, float* C16x3p) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc32 = {0}, acc1 = {0}, accH = {0}, accL = {0}, accU = {0}, accP = {0};
  for (int k0 = 0; k0 < K; k0 += 2) acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k0 + h], B[r * K + k0 + h], acc32, 0, 0, 0);
  for (int k0 = 0; k0 < K; k0 += 16) {
    h8 ah, al, bh, bl, alu, blu, pah, pal, pbh, pbl;
    for (int j = 0; j < 8; ++j) {
      const float a = A[r * K + k0 + 8 * h + j], b = B[r * K + k0 + 8 * h + j];
      ah[j] = (_Float16)a; al[j] = (_Float16)((a - (float)ah[j]) * scale_lo); alu[j] = (_Float16)(a - (float)ah[j]);
      { const float as = a * 8.0f, bs = b * 4096.0f; pah[j] = (_Float16)as; pal[j] = (_Float16)(as - (float)pah[j]); pbh[j] = (_Float16)bs; pbl[j] = (_Float16)(bs - (float)pbh[j]); }
      bh[j] = (_Float16)b; bl[j] = (_Float16)((b - (float)bh[j]) * scale_lo); blu[j] = (_Float16)(b - (float)bh[j]);
    }
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
    accH = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accH, 0, 0, 0);
    accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accL, 0, 0, 0);
    accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accL, 0, 0, 0);
    accU = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accU, 0, 0, 0);
    accU = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blu, accU, 0, 0, 0);
    accU = __builtin_amdgcn_mfma_f32_32x32x16_f16(alu, bh, accU, 0, 0, 0);
    accP = __builtin_amdgcn_mfma_f32_32x32x16_f16(pal, pbh, accP, 0, 0, 0);
    accP = __builtin_amdgcn_mfma_f32_32x32x16_f16(pah, pbl, accP, 0, 0, 0);
    accP = __builtin_amdgcn_mfma_f32_32x32x16_f16(pah, pbh, accP, 0, 0, 0);
  }
  for (int q = 0; q < 16; ++q) {
    const int row = (q & 3) + 8 * (q >> 2) + 4 * h, col = r;   // C[row=A row][col=B row]
    C32[row * 32 + col] = acc32[q];
    C16[row * 32 + col] = acc1[q];
    C16x3[row * 32 + col] = accH[q] + accL[q] * (1.0f / scale_lo);
    C16x3u[row * 32 + col] = accU[q];
    C16x3p[row * 32 + col] = accP[q] * (1.0f / 32768.0f);
  }
}

static double urand() { return (double)rand() / RAND_MAX * 2.0 - 1.0; }
static double nrand() { double s = 0; for (int i = 0; i < 12; ++i) s += (double)rand() / RAND_MAX; return s - 6.0; }

int main() {
  for (int cas = 0; cas < 4; ++cas) {
    const int K = (cas == 1) ? 1024 : 512;
    std::vector<float> A(32 * K), B(32 * K);
    for (auto& v : A) v = (cas == 2) ? (float)(nrand() * 30.0) : (cas == 3 ? (float)(urand() * 1e-3) : (float)nrand());
    for (auto& v : B) v = (float)(urand() / sqrt((double)K));
    float *dA, *dB, *d[5];
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4);
    for (auto& p : d) hipMalloc(&p, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, K, d[0], d[1], d[2], d[3], 2048.0f, d[4]);
    hipDeviceSynchronize();
    std::vector<float> out[5];
    for (int i = 0; i < 5; ++i) { out[i].resize(1024); hipMemcpy(out[i].data(), d[i], 4096, hipMemcpyDeviceToHost); }
    double ref[1024], mag = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[j * K + k];
      ref[i * 32 + j] = s; mag = fmax(mag, fabs(s));
    }
    // fp32 sequential fmaf chain on the host for comparison
    double e_host = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      float s = 0; for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[j * K + k], s);
      e_host = fmax(e_host, fabs((double)s - ref[i * 32 + j]));
    }
    const char* names[5] = {"mfma_f32 ", "f16 x1   ", "f16x3 2^11", "f16x3 noscale", "f16x3 prescale1acc"};
    printf("case %d K=%d |C|max=%.3g  host fmaf chain max-err %.3e\n", cas, K, mag, e_host);
    for (int v = 0; v < 5; ++v) {
      double mx = 0, rms = 0;
      for (int i = 0; i < 1024; ++i) { double e = fabs((double)out[v][i] - ref[i]); mx = fmax(mx, e); rms += e * e; }
      printf("   %-18s max-err %.3e  rms %.3e  (rel to |C|max: %.2e)\n", names[v], mx, sqrt(rms / 1024), mx / mag);
    }
  }
  return 0;
}
