// Stand-alone probe (no kernel of the engine): does ds_bpermute_b32 (what __shfl_xor compiles to on gfx950) return wrong
// lanes while ANOTHER wave on the same CU works in LDS -- in particular at byte addresses >= 128 KiB, which only a workgroup
// holding (almost) all of the CU's 160 KiB reaches?
//
//   hipcc --offload-arch=gfx950 -O3 -o bperm_probe bperm_probe.hip && ./bperm_probe [rounds]
//
// k_shfl (no LDS allocation): every wave sums 64 lane values with the xor butterfly of __shfl_xor and compares with the sum
// each lane forms by itself; mismatches are counted and the first few recorded.
// k_hog  (one 512-thread workgroup per CU, dynamic LDS): keeps reading / writing LDS in [lo, hi) -- by ds_write/ds_read or by
// LDS-DMA (global_load_lds_dwordx4) -- until told to stop.
// The two run at once on two streams of ONE process, so their waves share CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__device__ __forceinline__ unsigned f(unsigned lane, unsigned it, unsigned w) { return ((lane * 2654435761u) ^ (it * 40503u + w * 97u)) >> 12; }

__global__ __launch_bounds__(256) void k_shfl(unsigned long long* bad, unsigned* rec, int iters) {
  const unsigned lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int it = 0; it < iters; ++it) {
    unsigned s = f(lane, it, w), t = f(lane ^ 21u, it, w) + 7u, u = f(lane, it + 1000003, w);
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); t += __shfl_xor(t, o, 64); u += __shfl_xor(u, o, 64); }
    unsigned es = 0, et = 0, eu = 0;
    for (unsigned l = 0; l < 64; ++l) { es += f(l, it, w); et += f(l ^ 21u, it, w) + 7u; eu += f(l, it + 1000003, w); }
    if (s != es || t != et || u != eu) {
      const unsigned long long n = atomicAdd(bad, 1ull);
      if (n < 16) { rec[4 * n] = lane; rec[4 * n + 1] = (s != es) | ((t != et) << 1) | ((u != eu) << 2); rec[4 * n + 2] = s ^ es; rec[4 * n + 3] = it; }
    }
  }
}

// mode 0: ds_write + ds_read of [lo, hi);  mode 1: LDS-DMA fills of [lo, hi) from src + ds_read
__global__ __launch_bounds__(512) void k_hog(const unsigned* __restrict__ src, unsigned* sink, volatile int* stop, int lo, int hi, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned acc = 0;
  for (int round = 0; round < (1 << 20); ++round) {
    if (mode == 0) {
      for (int off = lo + tid * 16; off + 16 <= hi; off += 512 * 16)
        *reinterpret_cast<uint4*>(lds + off) = make_uint4(round, off, tid, acc);
    } else {
      for (int off = lo + wave * 1024; off + 1024 <= hi; off += 8 * 1024)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(src) + (size_t)(blockIdx.x & 63) * 65536 + (off & 65535) + lane * 16,
                                         (__attribute__((address_space(3))) void*)(uintptr_t)(lds + off), 16, 0, 0);
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __syncthreads();
    for (int off = lo + tid * 16; off + 16 <= hi; off += 512 * 16) {
      const uint4 v = *reinterpret_cast<const uint4*>(lds + off);
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    __syncthreads();
    if ((round & 15) == 0 && *stop) break;
  }
  if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20;
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  unsigned long long* bad; unsigned *rec, *src, *sink; int* stop;
  CK(hipMalloc(&bad, 8)); CK(hipMalloc(&rec, 16 * 16)); CK(hipMalloc(&src, 64 * 65536 + 4096)); CK(hipMalloc(&sink, 4096));
  CK(hipHostMalloc(&stop, 4, hipHostMallocMapped));
  CK(hipMemset(src, 0x5A, 64 * 65536 + 4096));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hog), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  struct Case { const char* name; int lds, lo, hi, mode; };
  const Case cases[] = {
      {"no hog (control)", 0, 0, 0, -1},
      {"hog 160 KiB, ds ops in [0, 32K)", 160 * 1024, 0, 32 * 1024, 0},
      {"hog 160 KiB, ds ops in [96K, 128K)", 160 * 1024, 96 * 1024, 128 * 1024, 0},
      {"hog 160 KiB, ds ops in [128K, 160K)", 160 * 1024, 128 * 1024, 160 * 1024, 0},
      {"hog 160 KiB, LDS-DMA into [0, 32K)", 160 * 1024, 0, 32 * 1024, 1},
      {"hog 160 KiB, LDS-DMA into [128K, 160K)", 160 * 1024, 128 * 1024, 160 * 1024, 1},
      {"hog 96 KiB, LDS-DMA into [64K, 96K)", 96 * 1024, 64 * 1024, 96 * 1024, 1},
  };
  for (const Case& c : cases) {
    CK(hipMemset(bad, 0, 8));
    *stop = 0;
    if (c.mode >= 0) hipLaunchKernelGGL(k_hog, dim3(256), dim3(512), c.lds, s1, src, sink, stop, c.lo, c.hi, c.mode);
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_shfl, dim3(1024), dim3(256), 0, s2, bad, rec, 2000);
    CK(hipStreamSynchronize(s2));
    *stop = 1;
    CK(hipStreamSynchronize(s1));
    unsigned long long hb = 0; unsigned hr[64];
    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr, rec, sizeof(hr), hipMemcpyDeviceToHost));
    printf("%-44s: %llu wrong butterfly sums of %lld", c.name, hb, (long long)rounds * 1024 * 4 * 2000 * 64);
    for (unsigned i = 0; i < (hb < 4 ? hb : 4); ++i) printf("  [lane %u which %u xor %08x it %u]", hr[4 * i], hr[4 * i + 1], hr[4 * i + 2], hr[4 * i + 3]);
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
