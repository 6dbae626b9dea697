// What MFMA rate does the MI355X sustain on random fp16 operands (DVFS included)?  Pure register loops, no memory.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int SHAPE>   // 0: 32x32x16 (8 accumulators of 16), 1: 16x16x32 (32 accumulators of 4)
__global__ __launch_bounds__(512) void loop(const _Float16* in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  h8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const h8*>(in + ((threadIdx.x * 6 + i) * 8) % 4096);
  for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const h8*>(in + ((threadIdx.x * 6 + 4 + i) * 8) % 4096);
  if (SHAPE == 0) {
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 6; ++rep)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 3; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j & 1], a[i & 3], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) s += acc[i][j][q];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
  (void)lane;
}

int main() {
  std::vector<_Float16> h(4096);
  for (auto& v : h) v = (_Float16)((float)rand() / RAND_MAX * 2.f - 1.f);
  _Float16* din; float* dout;
  hipMalloc(&din, 8192); hipMalloc(&dout, 256 * 8 * 512 * 4);
  hipMemcpy(din, h.data(), 8192, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int shape = 0; shape < 2; ++shape)
    for (int blocks : {256, 512}) {
      const int iters = 4000;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (shape == 0) hipLaunchKernelGGL(loop<0>, dim3(blocks), dim3(512), 0, 0, din, dout, iters);
        else hipLaunchKernelGGL(loop<1>, dim3(blocks), dim3(512), 0, 0, din, dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per wave per iter: shape0: 48 MFMA x 32768 flop; shape1: 96 MFMA x 16384 flop  (same flops)
        const double flops = (double)blocks * 8 * iters * 48.0 * 32768.0;
        if (rep == 1) printf("shape %s blocks %d (%d waves/SIMD): %.3f ms  %.0f TFLOP/s\n", shape ? "16x16x32" : "32x32x16", blocks,
                             blocks / 256 * 2, ms, flops / ms / 1e9);
      }
    }
  return 0;
}
