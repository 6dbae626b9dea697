// Stand-alone probe (no kernel of the engine): do sub-cache-line stores by workgroups on DIFFERENT XCDs into the SAME 128-byte
// line survive when a second process uses the GPU?
//
//   hipcc --offload-arch=gfx950 -O3 -o false_share_probe false_share_probe.hip
//   ./false_share_probe [rows] [iters] [rows_per_block] &  ./false_share_probe ... &  wait      (two processes at once)
//
// k_rows: one wave per row, lanes 0..2 store three 4-byte values -> 12 bytes per row; rows_per_block = 4 (48 bytes per
// workgroup: lines shared by 3-4 workgroups, which the dispatcher deals round-robin over the 8 XCDs) or 32 (384 bytes = three
// whole lines per workgroup).  k_fill rewrites the buffer with a different pattern in between (whole lines), k_busy is an
// unrelated streaming kernel that keeps kernel boundaries (cache write-back / invalidate) coming.  Every iteration the buffer
// is copied back and compared with the expected pattern on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__host__ __device__ __forceinline__ unsigned mix(unsigned a, unsigned b) { return (a * 2654435761u) ^ (b * 40503u + 0x9E3779B9u); }

template <int RPW>
__global__ __launch_bounds__(256) void k_rows(const float* __restrict__ src, unsigned* __restrict__ out, int rows, int cols, unsigned salt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = 0; r < RPW; ++r) {
    const int row = (blockIdx.x * 4 + wave) * RPW + r;
    if (row >= rows) return;
    // read the row (as the engine's head kernel does: 2 KB per row) so that waves reach their stores at different times
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += src[(size_t)row * cols + c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane < 3) out[(size_t)row * 3 + lane] = mix((unsigned)row * 3u + lane, salt) + (s == 12345.678f ? 1u : 0u);
  }
}
__global__ void k_fill(unsigned* out, size_t n, unsigned v) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = v;
}
__global__ void k_busy(float* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 1377, iters = argc > 2 ? atoi(argv[2]) : 2000, rpb = argc > 3 ? atoi(argv[3]) : 4;
  const int cols = 512;
  float* src; unsigned* out; float* busy;
  const size_t nbusy = 1 << 22;
  CK(hipMalloc(&src, (size_t)rows * cols * 4));
  CK(hipMalloc(&out, (size_t)rows * 3 * 4 + 512));
  CK(hipMalloc(&busy, nbusy * 4));
  CK(hipMemset(src, 0, (size_t)rows * cols * 4));
  CK(hipMemset(busy, 0, nbusy * 4));
  std::vector<unsigned> h((size_t)rows * 3);
  long bad_iters = 0, bad_words = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned salt = 1000u + it;
    hipLaunchKernelGGL(k_fill, dim3((rows * 3 + 255) / 256), dim3(256), 0, 0, out, (size_t)rows * 3, 0xDEADBEEFu);
    hipLaunchKernelGGL(k_busy, dim3(nbusy / 256), dim3(256), 0, 0, busy, nbusy);
    if (rpb == 32) hipLaunchKernelGGL(k_rows<8>, dim3((rows + 31) / 32), dim3(256), 0, 0, src, out, rows, cols, salt);
    else hipLaunchKernelGGL(k_rows<1>, dim3((rows + 3) / 4), dim3(256), 0, 0, src, out, rows, cols, salt);
    hipLaunchKernelGGL(k_busy, dim3(nbusy / 256), dim3(256), 0, 0, busy, nbusy);
    CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    long w = 0; int first = -1; unsigned got = 0;
    for (size_t i = 0; i < h.size(); ++i)
      if (h[i] != mix((unsigned)i, salt)) { if (first < 0) { first = (int)i; got = h[i]; } ++w; }
    if (w) {
      ++bad_iters; bad_words += w;
      if (bad_iters <= 10)
        printf("iter %d: %ld wrong words, first at word %d (row %d, byte offset %d in its 128-B line): got %08x %s\n", it, w, first,
               first / 3, (first * 4) & 127, got, got == 0xDEADBEEFu ? "(the fill pattern: store lost)" : "(other)");
    }
  }
  printf("rows_per_block %d rows %d: %ld of %d iterations with wrong words (%ld words)\n", rpb, rows, bad_iters, iters, bad_words);
  return 0;
}
