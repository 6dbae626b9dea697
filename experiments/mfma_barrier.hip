// What does a workgroup barrier cost inside an MFMA-bound loop?  8 waves per workgroup (2 per SIMD), one workgroup per CU,
// 96 v_mfma_f32_16x16x32_f16 per wave between barriers (the k-tile of the 256x256 F16X3 GEMM), registers only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// ORDER 0: the 96 MFMAs of a block go round-robin over 32 accumulators (dependent distance 32)
// ORDER 1: the GEMM's order -- per 16-row group, three products over 4 accumulators (dependent distance 4 = 64 cycles)
// ORDER 2: two groups interleaved (dependent distance 8)
template <int EVERY, int ORDER>   // barrier after every EVERY blocks of 96 MFMAs (0 = never)
__global__ __launch_bounds__(512) void loop(const _Float16* in, float* out, int iters) {
  extern __shared__ unsigned char lds[];
  h8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const h8*>(in + ((threadIdx.x * 6 + i) * 8) % 4096);
  for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const h8*>(in + ((threadIdx.x * 6 + 4 + i) * 8) % 4096);
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (ORDER == 0) {
#pragma unroll
      for (int rep = 0; rep < 3; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j & 1], a[i & 3], acc[i][j], 0, 0, 0);
    } else if (ORDER == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j & 1], a[(i + rep) & 3], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i + ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j & 1], a[(i + ii + rep) & 3], acc[i + ii][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (EVERY && (it % EVERY) == EVERY - 1) __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) s += acc[i][j][q];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (s == 12345.f) lds[threadIdx.x] = 1;
}

int main() {
  std::vector<_Float16> h(4096);
  for (auto& v : h) v = (_Float16)((float)rand() / RAND_MAX * 2.f - 1.f);
  _Float16* din; float* dout;
  hipMalloc(&din, 8192); hipMalloc(&dout, 256 * 512 * 4);
  hipMemcpy(din, h.data(), 8192, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  auto time = [&](auto kfn, const char* what) {
    hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);   // one workgroup per CU, as in the GEMM
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kfn, dim3(256), dim3(512), 131072, 0, din, dout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    const double flops = 256.0 * 8 * iters * 96.0 * 16384.0;
    printf("%-46s %.3f ms  %5.0f TFLOP/s\n", what, best, flops / best / 1e9);
  };
  time(loop<0, 0>, "distance 32, no barrier");
  time(loop<1, 0>, "distance 32, barrier per 96 MFMAs");
  time(loop<0, 1>, "distance 4 (GEMM order), no barrier");
  time(loop<1, 1>, "distance 4 (GEMM order), barrier per 96 MFMAs");
  time(loop<0, 2>, "distance 8, no barrier");
  time(loop<1, 2>, "distance 8, barrier per 96 MFMAs");
  return 0;
}
