// Machine probes behind d3d_probe_machine (include/d3d.h): what THIS chip sustains, on the day, for the two resources the F16X3
// k-loop is co-limited by (experiments/NOTES.md 0.5) -- so that a bench line can state its roofline fraction
// against the measured ceilings beside the nominal peaks.  Neither probe touches an engine or its workspace.
//   probe 0  sustained fp16 MFMA rate: register loops only (no memory, no LDS), 8 waves per workgroup = 2 per SIMD, one workgroup per
//            CU, v_mfma_f32_16x16x32_f16 in the production order of an m-tile group (a_lo b_hi, a_hi b_lo, a_hi b_hi per n-tile), operands
//            with the statistics of real hi / lo halves (MFMA power draw, and with it the clock the chip holds, depends on them)
//   probe 1  L2 -> LDS staging rate: the k-loop's DMA stream alone -- every wave of a 512-thread workgroup per CU issues the eight 1-KiB
//            global_load_lds_dwordx4 pieces of a 256 x 256 x 32 stage (64 KiB) per k-tile from operand rows that stay in the XCD's L2,
//            one vmcnt(0) + barrier per k-tile as the one-barrier loop has it, no MFMA
#include "d3d_kernels.h"

#include <math.h>
#include <stdlib.h>
#include <vector>

namespace d3d {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k_probe_mfma(const _Float16* __restrict__ in, float* __restrict__ out, int iters) {
  h8 ah[2], al[2], bh[4], bl[4];
  const _Float16* p = in + (size_t)(threadIdx.x & 63) * 8 * 12;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ah[i] = *reinterpret_cast<const h8*>(p + 8 * i);
    al[i] = *reinterpret_cast<const h8*>(p + 8 * (2 + i));
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bh[j] = *reinterpret_cast<const h8*>(p + 8 * (4 + j));
    bl[j] = *reinterpret_cast<const h8*>(p + 8 * (8 + j));
  }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[g & 1], acc[g][j], 0, 0, 0);
        acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[g & 1], acc[g][j], 0, 0, 0);
        acc[g][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[g & 1], acc[g][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if ((it & 63) == 63) {   // keep the sums bounded (and alive)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] *= 1e-3f;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) s += acc[i][j][q];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

constexpr int PS_ROWS = 1024;        // operand rows of the staging probe: 1024 x 2 KiB = 2 MiB, inside one XCD's 4 MiB L2
constexpr int PS_PITCH = 2048;       // bytes per row (K = 512 in the pair layout): 16 k-tiles of 128 B
constexpr int PS_STAGE = 65536;      // one k-tile of a 256 x 256 tile

__global__ __launch_bounds__(512) void k_probe_stage(const char* __restrict__ rows, unsigned* __restrict__ out, int tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  // the GEMM's plan: piece p = 8 rows x 128 B, wave w moves pieces w, w + 8, ..., lane l serves row 8 p + l / 8, chunk l % 8 (swizzled)
  const int lr = lane >> 3, csrc = (lane & 7) ^ (((wave & 1) << 2) | (lr >> 1));
  const unsigned lofs = (unsigned)(lr * PS_PITCH + csrc * 16);
  for (int t = 0; t < tiles; ++t) {
    // 512 rows per tile (256 of "A", 256 of "W"), a different window of the resident rows per workgroup and tile
    const int row0 = (int)(((unsigned)blockIdx.x * 40u + (unsigned)t * 88u) % (unsigned)(PS_ROWS - 512)) & ~7;
    for (int kt = 0; kt < 16; ++kt) {
      const char* base = rows + (size_t)(row0 + wave * 8) * PS_PITCH + (size_t)kt * 128;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(base + (size_t)it * 64 * PS_PITCH);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        const char* sp = reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
        __builtin_amdgcn_global_load_lds(sp + lofs,
                                         (__attribute__((address_space(3))) void*)(uintptr_t)(lds + (kt & 1) * PS_STAGE + (it * 8 + wave) * 1024 + lane * 16),
                                         16, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) out[blockIdx.x] = *reinterpret_cast<const unsigned*>(lds + 64);   // (the stream is observable)
}

}  // namespace

// what 0: *result = TFLOP/s of fp16 MFMA work sustained over ~`ms_target` milliseconds; what 1: *result = GB/s chip-wide of the staging
// stream.  Both time one launch (after an untimed one) with HIP events on `s`.
hipError_t launch_probe_machine(int what, float ms_target, float* result, hipStream_t s) {
  if (!result || (what != 0 && what != 1) || !(ms_target > 0.f)) return hipErrorInvalidValue;
  const int n_cu = device_cu_count();
  if (n_cu <= 0) return hipErrorUnknown;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  void *din = nullptr, *dout = nullptr;
  hipError_t err = hipSuccess;
  auto cleanup = [&]() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(din);
    (void)hipFree(dout);
  };
#define PROBE_TRY(x) do { err = (x); if (err != hipSuccess) { cleanup(); return err; } } while (0)
  PROBE_TRY(hipEventCreate(&e0));
  PROBE_TRY(hipEventCreate(&e1));
  float ms = 0.f;
  if (what == 0) {
    std::vector<_Float16> h(64 * 12 * 8);
    unsigned long long st = 0x9E3779B97F4A7C15ull;   // fixed seed: the same operands on every box
    auto uni = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) + 1) / 16777218.0f; };
    for (int lane = 0; lane < 64; ++lane)
      for (int f = 0; f < 12; ++f)
        for (int e = 0; e < 8; ++e) {
          const float x = 8.f * sqrtf(-2.f * logf(uni())) * cosf(6.2831853f * uni());
          const _Float16 hi = (_Float16)x, lo = (_Float16)(x - (float)hi);
          const bool is_lo = (f >= 2 && f < 4) || f >= 8;
          h[(lane * 12 + f) * 8 + e] = is_lo ? lo : hi;
        }
    PROBE_TRY(hipMalloc(&din, h.size() * 2));
    PROBE_TRY(hipMalloc(&dout, (size_t)n_cu * 512 * 4));
    PROBE_TRY(hipMemcpyAsync(din, h.data(), h.size() * 2, hipMemcpyHostToDevice, s));
    PROBE_TRY(hipStreamSynchronize(s));
    // one iteration = 96 MFMAs of 16384 flop per wave; ~1.6 us at 1.9 GHz
    const int iters = (int)fmaxf(64.f, ms_target * 1000.f / 1.6f);
    for (int rep = 0; rep < 2; ++rep) {
      PROBE_TRY(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_probe_mfma, dim3(n_cu), dim3(512), 0, s, (const _Float16*)din, (float*)dout, iters);
      PROBE_TRY(hipGetLastError());
      PROBE_TRY(hipEventRecord(e1, s));
      PROBE_TRY(hipEventSynchronize(e1));
      PROBE_TRY(hipEventElapsedTime(&ms, e0, e1));
    }
    *result = (float)((double)n_cu * 8 * iters * 96.0 * 16384.0 / ((double)ms * 1e9));
  } else {
    static std::atomic<unsigned long long> attr_set{0};
    PROBE_TRY(lds_optin(reinterpret_cast<const void*>(&k_probe_stage), 2 * PS_STAGE, attr_set));
    PROBE_TRY(hipMalloc(&din, (size_t)PS_ROWS * PS_PITCH));
    PROBE_TRY(hipMalloc(&dout, (size_t)n_cu * 4));
    PROBE_TRY(hipMemsetAsync(din, 0x3c, (size_t)PS_ROWS * PS_PITCH, s));
    // one tile = 16 k-tiles of 64 KiB per workgroup; ~20 us at the rate the GEMM's staging-only build showed
    const int tiles = (int)fmaxf(4.f, ms_target * 1000.f / 20.f);
    for (int rep = 0; rep < 2; ++rep) {
      PROBE_TRY(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_probe_stage, dim3(n_cu), dim3(512), 2 * PS_STAGE, s, (const char*)din, (unsigned*)dout, tiles);
      PROBE_TRY(hipGetLastError());
      PROBE_TRY(hipEventRecord(e1, s));
      PROBE_TRY(hipEventSynchronize(e1));
      PROBE_TRY(hipEventElapsedTime(&ms, e0, e1));
    }
    *result = (float)((double)n_cu * tiles * 16.0 * PS_STAGE / ((double)ms * 1e6));
  }
#undef PROBE_TRY
  cleanup();
  return hipSuccess;
}

}  // namespace d3d
