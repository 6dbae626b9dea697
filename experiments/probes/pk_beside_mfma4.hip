// Fourth probe (see pk_beside_mfma.hip): OTHER instruction families with operand-half selection or cross-lane operands (packed fp16, v_fma_mix*, DPP,
// v_cvt_pkrtz) beside the k-loop-like partner of the third probe -- do they deviate as v_pk_*_f32 with src1 op_sel does?  (derived from the third probe: a partner that looks like the GEMM k-loop (MFMA 16x16x32 f16 + ds_read_b128 + LDS-DMA) next to
// testers that run ONE packed-fp32 form each.  Forms as in pk_beside_mfma2.hip plus the v_pk_fma/v_pk_mul op_sel_hi forms.
//   hipcc --offload-arch=gfx950 -O2 -o pk_probe3.bin experiments/probes/pk_beside_mfma3.hip ; ./pk_probe3.bin [iters] [partner mask]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
static int g_idle_us = 20000;

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

enum { G_PK_ADD_F16_S1_01 = 0, G_PK_FMA_F16_S1_01, G_PK_MUL_F16_S1_HI10, G_MIX_F32_S1HI, G_MIXLO_S0HI, G_MIXHI_S1HI, G_ADD_DPP_ROWSHR, G_ADD_DPP_QUAD,
       G_CVT_PKRTZ, G_PK_ADD_F32_REF, NFORMS };
static const char* form_name[NFORMS] = {"pk_add_f16 src1 op_sel:[0,1]", "pk_fma_f16 src1 op_sel:[0,1,0]", "pk_mul_f16 src1 op_sel_hi:[1,0]",
                                        "fma_mix_f32 src1 hi half (op_sel)", "fma_mixlo_f16 src0 hi half", "fma_mixhi_f16 src1 hi half",
                                        "add_f32 dpp row_shr:1", "add_f32 dpp quad_perm", "cvt_pkrtz_f16_f32", "pk_add_f32 src1 op_sel:[0,1] (reference: deviates)"};

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned h2bits(h2 v) { return __builtin_bit_cast(unsigned, v); }

template <int FORM>
__device__ __forceinline__ void tester(int lane, int wave, int iters, unsigned* bad, unsigned* badlane, int tmask, const float* gsrc, float* gdst,
                                       const unsigned char* lds) {
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const int xi = (lane * 3 + it) & 255, yi = (lane * 5 + 2 * it + 1) & 255, mi = 1 + ((it * 7 + wave) & 31), ji = 40 + ((it * 13 + 77) & 31);
    bool ok = true;
    float got0 = 0, got1 = 0, ex0 = 0, ex1 = 0;
    if (FORM == G_PK_ADD_F16_S1_01 || FORM == G_PK_FMA_F16_S1_01 || FORM == G_PK_MUL_F16_S1_HI10) {
      h2 x, m, g, r;
      x.x = (_Float16)(float)xi; x.y = (_Float16)(float)yi; m.x = (_Float16)(float)ji; m.y = (_Float16)(float)mi; g.x = (_Float16)3.0f; g.y = (_Float16)3.0f;
      if (FORM == G_PK_ADD_F16_S1_01) { asm volatile("v_pk_add_f16 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(m)); ex0 = xi + mi; ex1 = yi + mi; }
      if (FORM == G_PK_FMA_F16_S1_01) { asm volatile("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(x), "v"(m), "v"(g)); ex0 = xi * mi + 3; ex1 = yi * mi + 3; }
      if (FORM == G_PK_MUL_F16_S1_HI10) { asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(m)); ex0 = xi * ji; ex1 = yi * ji; }
      if (FORM == G_PK_FMA_F16_S1_01 || FORM == G_PK_MUL_F16_S1_HI10) { ex0 = (float)(_Float16)ex0; ex1 = (float)(_Float16)ex1; }   // (products up to 255 * 71: round like the hardware)
      got0 = (float)r.x; got1 = (float)r.y;
      ok = got0 == ex0 && got1 == ex1;
    }
    if (FORM == G_MIX_F32_S1HI) {    // r = x.lo * m.hi + 3
      h2 x, m; float r;
      x.x = (_Float16)(float)xi; x.y = (_Float16)(float)yi; m.x = (_Float16)(float)ji; m.y = (_Float16)(float)mi;
      asm volatile("v_fma_mix_f32 %0, %1, %2, 1.0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(x), "v"(m));
      got0 = r; ex0 = (float)(xi * mi + 1); ok = got0 == ex0;
    }
    if (FORM == G_MIXLO_S0HI || FORM == G_MIXHI_S1HI) {   // 16-bit result into one half of r, the other half preserved
      h2 x, m; unsigned r = 0x12345678u;
      x.x = (_Float16)(float)xi; x.y = (_Float16)(float)(yi & 63); m.x = (_Float16)(float)ji; m.y = (_Float16)(float)mi;
      if (FORM == G_MIXLO_S0HI) { asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "+v"(r) : "v"(x), "v"(m)); ex0 = (float)(_Float16)(float)((yi & 63) * ji); got0 = (float)__builtin_bit_cast(h2, r).x; ok = got0 == ex0 && (r >> 16) == 0x1234u; }
      else { asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "+v"(r) : "v"(x), "v"(m)); ex0 = (float)(_Float16)(float)(xi * mi); got0 = (float)__builtin_bit_cast(h2, r).y; ok = got0 == ex0 && (r & 0xffffu) == 0x5678u; }
    }
    if (FORM == G_ADD_DPP_ROWSHR || FORM == G_ADD_DPP_QUAD) {
      float x = (float)xi, y = (float)(lane * 3 + it), r;
      if (FORM == G_ADD_DPP_ROWSHR) { asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(y), "v"(x)); ex0 = ((lane & 15) ? (float)((lane - 1) * 3 + it) : 0.0f) + x; }
      else { asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(y), "v"(x)); ex0 = (float)((lane ^ 1) * 3 + it) + x; }
      got0 = r; ok = got0 == ex0;
    }
    if (FORM == G_CVT_PKRTZ) {
      float a = (float)xi, b = (float)yi; h2 r;
      asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      got0 = (float)r.x; got1 = (float)r.y; ex0 = a; ex1 = b; ok = got0 == ex0 && got1 == ex1;
    }
    if (FORM == G_PK_ADD_F32_REF) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 x, m, r; x.x = (float)xi; x.y = (float)yi; m.x = (float)ji; m.y = (float)mi;
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(m));
      got0 = r.x; got1 = r.y; ex0 = xi + mi; ex1 = yi + mi; ok = got0 == ex0 && got1 == ex1;
    }
    if (!ok) {
      ++nbad;
      atomicAdd(&badlane[lane], 1u);
      const unsigned k = atomicAdd(&bad[2], 1u);
      if (k < 4) { float* rec = reinterpret_cast<float*>(bad + 16 + 8 * k); rec[0] = got0; rec[1] = ex0; rec[2] = got1; rec[3] = ex1; rec[4] = (float)lane; rec[5] = (float)mi; rec[6] = (float)ji; rec[7] = (float)xi; }
    }
  }
  if (nbad) atomicAdd(&bad[0], nbad);
}

// partner: bit 0 MFMAs, bit 1 ds_read_b128 fragment reads, bit 2 LDS-DMA pieces, bit 3 s_sleep 1 + ds_read_b32 polls between groups,
// bit 4 an idle gap (s_sleep 40 = 2560 cycles) after every burst of 8 MFMAs
template <int FORM>
__global__ __launch_bounds__(512) void probe(unsigned* bad, unsigned* badlane, float* sink, const _Float16* src, int iters, int pmask, int tmask, float* gdst) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = 0x2e662e66u;   // fp16 0.1
  __syncthreads();
  if (wave < 4) { tester<FORM>(lane, wave, iters, bad, badlane, tmask, reinterpret_cast<const float*>(src) + blockIdx.x * 16384, gdst + (size_t)blockIdx.x * 65536, lds); return; }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  h8 a = *reinterpret_cast<const h8*>(lds + lane * 16), b = *reinterpret_cast<const h8*>(lds + 1024 + lane * 16);
  const char* base = reinterpret_cast<const char*>(src) + (size_t)blockIdx.x * 65536 + (wave - 4) * 16384;
  const int n = iters / 4;
  for (int it = 0; it < n; ++it) {
    if (pmask & 4) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
        __builtin_amdgcn_global_load_lds(base + ((it * 2 + p) & 15) * 1024 + lane * 16,
                                         (__attribute__((address_space(3))) void*)(uintptr_t)(lds + (wave - 4) * 16384 + ((it * 2 + p) & 15) * 1024), 16, 0, 0);
    }
    if (pmask & 2) {
      a = *reinterpret_cast<const h8*>(lds + ((lane * 16 + it * 1024) & 65520));
      b = *reinterpret_cast<const h8*>(lds + ((lane * 16 + it * 1024 + 32768) & 65520));
    }
    if (pmask & 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    if (pmask & 16) __builtin_amdgcn_s_sleep(40);   // the matrix pipe goes idle between bursts: every burst is a wake-up
    if (pmask & 8) {
      __builtin_amdgcn_s_sleep(1);
      acc[0][0] += (float)*reinterpret_cast<volatile unsigned*>(lds + 60000);
    }
    if ((pmask & 4) && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = 0.f;
  for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][3];
  sink[blockIdx.x * 512 + threadIdx.x] = t;
}

static float* g_dst = nullptr;
template <int FORM>
static void run(unsigned* bad, unsigned* badlane, float* sink, const _Float16* src, int iters, int pmask, int tmask = 0, int reps = 1) {
  if (!g_dst) (void)hipMalloc(&g_dst, (size_t)256 * 65536 * 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  unsigned long long total = 0; int launches_hit = 0; unsigned lo = 64, hi = 0; float rec0[8] = {0};
  for (int r = 0; r < reps; ++r) {
    (void)hipMemset(bad, 0, 4096); (void)hipMemset(badlane, 0, 256);
    (void)hipDeviceSynchronize();
    usleep(g_idle_us);      // the chip idles between launches: every launch is a load step from idle
    hipLaunchKernelGGL(probe<FORM>, dim3(256), dim3(512), 65536, 0, bad, badlane, sink, src, iters, pmask, tmask, g_dst);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    unsigned h[1024], hl[64];
    (void)hipMemcpy(h, bad, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(hl, badlane, 256, hipMemcpyDeviceToHost);
    if (h[0]) {
      if (!total) for (int k = 0; k < 8; ++k) rec0[k] = reinterpret_cast<const float*>(h + 16)[k];
      total += h[0]; ++launches_hit;
      for (unsigned l = 0; l < 64; ++l) if (hl[l]) { lo = l < lo ? l : lo; hi = l > hi ? l : hi; }
    }
  }
  printf("%-52s | partner mask %2d: %8llu wrong lane-results, %3d of %d launches hit (%.2g wave-instructions each)", form_name[FORM], pmask, total,
         launches_hit, reps, (double)iters * 4 * 256);
  if (total) printf("  lanes %u..%u; e.g. got (%.0f, %.0f) expected (%.0f, %.0f)", lo, hi, rec0[0], rec0[2], rec0[1], rec0[3]);
  printf("\n");
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned *bad, *badlane; float* sink; _Float16* src;
  (void)hipMalloc(&bad, 4096); (void)hipMalloc(&badlane, 256); (void)hipMalloc(&sink, 256 * 512 * 4); (void)hipMalloc(&src, 256 * 65536);
  (void)hipMemset(src, 0x2e, 256 * 65536);
  const int pm = argc > 2 ? atoi(argv[2]) : 5, reps = argc > 3 ? atoi(argv[3]) : 200;
  if (argc > 4) g_idle_us = atoi(argv[4]);
  run<9>(bad, badlane, sink, src, iters, pm, 0, reps);
  run<0>(bad, badlane, sink, src, iters, pm, 0, reps); run<1>(bad, badlane, sink, src, iters, pm, 0, reps); run<2>(bad, badlane, sink, src, iters, pm, 0, reps);
  run<3>(bad, badlane, sink, src, iters, pm, 0, reps); run<4>(bad, badlane, sink, src, iters, pm, 0, reps); run<5>(bad, badlane, sink, src, iters, pm, 0, reps);
  run<6>(bad, badlane, sink, src, iters, pm, 0, reps); run<7>(bad, badlane, sink, src, iters, pm, 0, reps); run<8>(bad, badlane, sink, src, iters, pm, 0, reps);
  run<9>(bad, badlane, sink, src, iters, pm, 0, reps);
  return 0;
}
