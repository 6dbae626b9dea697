// Does a workgroup that owns (almost) all of a CU's LDS keep its LDS contents while a SECOND PROCESS uses the same GPU?
//   hipcc --offload-arch=gfx950 -O3 -o lds_share_probe lds_share_probe.hip
//   ./lds_share_probe 163840 3000 & ./lds_share_probe 163840 3000; wait        (two processes; compare with 98304)
// Each workgroup (512 threads, one per CU and more) fills its dynamic LDS with a pattern -- half of it by ds_write, half by
// LDS-DMA (global_load_lds_dwordx4) from a pattern buffer --, then re-reads it `passes` times with MFMA-free busy work between
// the passes, and counts the words that changed.  Prints the mismatch total over `launches` launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned pat(unsigned i, unsigned b, unsigned it) { return (i * 2654435761u) ^ (b * 40503u) ^ (it * 97u) ^ 0x5bd1e995u; }

__global__ __launch_bounds__(512) void k_probe(const unsigned* __restrict__ src, unsigned* mismatches, int words, int it, int passes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned* w = reinterpret_cast<unsigned*>(lds);
  const int half = (words / 2) & ~2047;   // DMA part: multiple of 512 threads x 4 words
  // part 1: LDS-DMA, 16 bytes per lane per instruction, wave-uniform LDS base + lane * 16
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = wave * 256; base < half; base += 8 * 256) {   // a wave instruction moves 256 words
    const unsigned* g = src + ((size_t)blockIdx.x * half + base + lane * 4);
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(uintptr_t)(lds + (size_t)base * 4), 16, 0, 0);
  }
  // part 2: plain stores
  for (int i = half + threadIdx.x; i < words; i += 512) w[i] = pat(i, blockIdx.x, it);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  unsigned bad = 0;
  for (int p = 0; p < passes; ++p) {
    for (int i = threadIdx.x; i < words; i += 512) {
      const unsigned expect = i < half ? src[(size_t)blockIdx.x * half + i] : pat(i, blockIdx.x, it);
      bad += w[i] != expect;
    }
    __syncthreads();
  }
  if (bad) atomicAdd(mismatches, bad);
}

int main(int argc, char** argv) {
  const int bytes = argc > 1 ? atoi(argv[1]) : 163840, launches = argc > 2 ? atoi(argv[2]) : 2000, passes = argc > 3 ? atoi(argv[3]) : 8;
  const int words = bytes / 4, grid = 512, half = (words / 2) & ~2047;
  std::vector<unsigned> h((size_t)grid * half);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2246822519u) ^ 0x9e3779b9u;
  unsigned *src = nullptr, *mis = nullptr;
  if (hipMalloc(&src, h.size() * 4) != hipSuccess || hipMalloc(&mis, 4) != hipSuccess) return 2;
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(mis, 0, 4);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 3;
  for (int it = 0; it < launches; ++it) hipLaunchKernelGGL(k_probe, dim3(grid), dim3(512), bytes, 0, src, mis, words, it, passes);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 4; }
  unsigned m = 0;
  hipMemcpy(&m, mis, 4, hipMemcpyDeviceToHost);
  printf("lds %d bytes, %d launches x %d workgroups, %d passes: %u mismatching words\n", bytes, launches, grid, passes, m);
  return 0;
}
