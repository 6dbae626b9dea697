// Does the ORDER of the F16X3 MFMAs of an m-tile group change what the power-limited chip sustains?  (experiments/NOTES.md 0.4)
// Register loops only (no memory, no LDS): 8 waves per workgroup, one workgroup per CU, the 12 MFMAs of a group = 4 n-tiles x
// (a_lo b_hi, a_hi b_lo, a_hi b_hi) on v_mfma_f32_16x16x32_f16, operands with the statistics of real hi / lo halves.
//   order 0: triples per n-tile (the production order): consecutive MFMAs share at most one operand, every third none
//   order 1: A-stationary snake: (bh0..3, al) (bl3..0, ah) (bh3..0, ah) -- every consecutive pair shares one operand, the A side
//            changes twice per group; per accumulator the same three products in the same order (bit-identical)
//   order 4: pairs of n-tiles interleaved (dependent MFMAs one apart), production order per accumulator (bit-identical)
//   order 2: one operand pair for everything (lower bound of operand toggling; wrong arithmetic)
// hipcc --offload-arch=gfx950 -O3 experiments/mfma_order.hip -o experiments/mfma_order && experiments/mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define MMA(B, A, C) C = __builtin_amdgcn_mfma_f32_16x16x32_f16(B, A, C, 0, 0, 0)

template <int ORDER>
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
      if (ORDER == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { MMA(bh[j], AL, acc[g][j]); MMA(bl[j], AH, acc[g][j]); MMA(bh[j], AH, acc[g][j]); }
      } else if (ORDER == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) MMA(bh[j], AL, acc[g][j]);
#pragma unroll
        for (int j = 3; j >= 0; --j) MMA(bl[j], AH, acc[g][j]);
#pragma unroll
        for (int j = 3; j >= 0; --j) MMA(bh[j], AH, acc[g][j]);
      } else if (ORDER == 4) {   // pairs of n-tiles interleaved: a dependent MFMA never follows its producer directly; production order per accumulator
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
          MMA(bh[jp], AL, acc[g][jp]); MMA(bh[jp + 1], AL, acc[g][jp + 1]);
          MMA(bl[jp], AH, acc[g][jp]); MMA(bl[jp + 1], AH, acc[g][jp + 1]);
          MMA(bh[jp], AH, acc[g][jp]); MMA(bh[jp + 1], AH, acc[g][jp + 1]);
        }
      } else if (ORDER == 3) {   // W-stationary: (bh_j, al) (bh_j, ah) (bl_j, ah) -- NOT the production order of additions
#pragma unroll
        for (int j = 0; j < 4; ++j) { MMA(bh[j], AL, acc[g][j]); MMA(bh[j], AH, acc[g][j]); MMA(bl[j], AH, acc[g][j]); }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { MMA(bh[0], ah[0], acc[g][j]); MMA(bh[0], ah[0], acc[g][j]); MMA(bh[0], ah[0], acc[g][j]); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // keep the accumulators bounded (values stay "alive": no constant folding, negligible VALU)
    if ((it & 63) == 63)
      for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] *= 1e-3f;
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) s += acc[i][j][q];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  // per lane 12 fragments of 8 halves: hi halves ~ N(0, 1) * 8 (activations) / weights alike, lo halves = the rounding residue
  std::vector<_Float16> h(64 * 8 * 12);
  srand(7);
  auto gauss = []() { float u = (rand() + 1.f) / (RAND_MAX + 2.f), v = (rand() + 1.f) / (RAND_MAX + 2.f); return sqrtf(-2 * logf(u)) * cosf(6.2831853f * v); };
  for (int lane = 0; lane < 64; ++lane)
    for (int f = 0; f < 12; ++f)
      for (int e = 0; e < 8; ++e) {
        const float x = 8.f * gauss();
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        const bool is_lo = (f >= 2 && f < 4) || f >= 8;
        h[(lane * 12 + f) * 8 + e] = is_lo ? lo : hi;
      }
  _Float16* din; float* dout;
  hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 256 * 512 * 4);
  hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 60000;
  for (int threads : {512, 256})     // 2 waves per SIMD (the GEMM's occupancy) / 1 wave per SIMD (a wave running alone at a phase's end)
    for (int round = 0; round < 2; ++round)
      for (int order : {0, 1, 4, 3, 2}) {
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (order == 0) hipLaunchKernelGGL(loop<0>, dim3(256), dim3(threads), 0, 0, din, dout, iters);
          else if (order == 1) hipLaunchKernelGGL(loop<1>, dim3(256), dim3(threads), 0, 0, din, dout, iters);
          else if (order == 4) hipLaunchKernelGGL(loop<4>, dim3(256), dim3(threads), 0, 0, din, dout, iters);
          else if (order == 3) hipLaunchKernelGGL(loop<3>, dim3(256), dim3(threads), 0, 0, din, dout, iters);
          else hipLaunchKernelGGL(loop<2>, dim3(256), dim3(threads), 0, 0, din, dout, iters);
          hipEventRecord(e1); hipEventSynchronize(e1);
          hipEventElapsedTime(&ms, e0, e1);
        }
        const double flops = 256.0 * (threads / 64) * iters * 96.0 * 16384.0;
        printf("waves/SIMD %d round %d order %d: %.3f ms  %.0f TFLOP/s fp16  (%.1f cycles per MFMA and SIMD at 1.9 GHz)\n", threads / 256, round, order, ms,
               flops / ms / 1e9, ms * 1e-3 * 1.9e9 / ((double)iters * 96.0 * (threads / 256)));
      }
  return 0;
}
