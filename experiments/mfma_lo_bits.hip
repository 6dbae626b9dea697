// Does the matrix pipe draw less when the LO halves of the F16X3 operands carry fewer significant bits?  (experiments/NOTES.md 000, energy
// book: the three-MFMA products are about half of the dynamic energy of a sampling on a power-capped card.)  Register loops only, the
// production order of mfma_order.hip (a_lo b_hi, a_hi b_lo, a_hi b_hi per n-tile), 2 waves per SIMD; the lo operands keep their top
// NB mantissa bits (10 = what the engine uses; 0 bits = a power of two; "zero" = lo halves all zero: the floor of what operand content can give).
// A lever only if large: fewer lo bits cost accuracy (4 bits = the fp8 study of NOTES 0.13: 1.3e-4, over the gate).
// hipcc --offload-arch=gfx950 -O3 experiments/mfma_lo_bits.hip -o experiments/mfma_lo_bits && experiments/mfma_lo_bits
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define MMA(B, A, C) C = __builtin_amdgcn_mfma_f32_16x16x32_f16(B, A, C, 0, 0, 0)

__global__ __launch_bounds__(512) void loop(const _Float16* in, float* out, int iters) {
  h8 ah[2], al[2], bh[4], bl[4];
  const _Float16* p = in + (size_t)(threadIdx.x & 63) * 8 * 12;
  for (int i = 0; i < 2; ++i) { ah[i] = *reinterpret_cast<const h8*>(p + 8 * i); al[i] = *reinterpret_cast<const h8*>(p + 8 * (2 + i)); }
  for (int j = 0; j < 4; ++j) { bh[j] = *reinterpret_cast<const h8*>(p + 8 * (4 + j)); bl[j] = *reinterpret_cast<const h8*>(p + 8 * (8 + j)); }
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const h8 &AH = ah[g & 1], &AL = al[g & 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) { MMA(bh[j], AL, acc[g][j]); MMA(bl[j], AH, acc[g][j]); MMA(bh[j], AH, acc[g][j]); }
      __builtin_amdgcn_sched_barrier(0);
    }
    if ((it & 63) == 63)
      for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] *= 1e-3f;
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) s += acc[i][j][q];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  auto gauss = []() { float u = (rand() + 1.f) / (RAND_MAX + 2.f), v = (rand() + 1.f) / (RAND_MAX + 2.f); return sqrtf(-2 * logf(u)) * cosf(6.2831853f * v); };
  _Float16* din; float* dout;
  hipMalloc(&din, 64 * 8 * 12 * 2); hipMalloc(&dout, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 60000;
  const int variants[] = {10, 7, 5, 3, 0, -1, 10};   // mantissa bits kept in the lo halves; -1: lo halves zero; 10 again last (drift check)
  for (int round = 0; round < 2; ++round)
    for (int nb : variants) {
      std::vector<_Float16> h(64 * 8 * 12);
      srand(7);
      for (int lane = 0; lane < 64; ++lane)
        for (int f = 0; f < 12; ++f)
          for (int e = 0; e < 8; ++e) {
            const float x = 8.f * gauss();
            const _Float16 hi = (_Float16)x;
            _Float16 lo = (_Float16)(x - (float)hi);
            if (nb < 0) lo = (_Float16)0.f;
            else if (nb < 10) { unsigned short b; memcpy(&b, &lo, 2); b &= (unsigned short)(0xFFFFu << (10 - nb)); memcpy(&lo, &b, 2); }
            const bool is_lo = (f >= 2 && f < 4) || f >= 8;
            h[(lane * 12 + f) * 8 + e] = is_lo ? lo : hi;
          }
      hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
      float ms = 0.f;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(loop, dim3(256), dim3(512), 0, 0, din, dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const double flops = 256.0 * 8 * iters * 96.0 * 16384.0;
      printf("round %d lo mantissa bits %2d: %.3f ms  %.0f TFLOP/s fp16\n", round, nb, ms, flops / ms / 1e9);
    }
  return 0;
}
