// Third probe (see pk_beside_mfma.hip): a partner that looks like the GEMM k-loop (MFMA 16x16x32 f16 + ds_read_b128 + LDS-DMA) next to
// testers that run ONE packed-fp32 form each.  Forms as in pk_beside_mfma2.hip plus the v_pk_fma/v_pk_mul op_sel_hi forms.
//   hipcc --offload-arch=gfx950 -O2 -o pk_probe3.bin experiments/probes/pk_beside_mfma3.hip ; ./pk_probe3.bin [iters] [partner mask]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

enum { F_ADD_S1_01 = 0, F_ADD_S1_01_NEG, F_ADD_S1_HI_10, F_MUL_S1_HI_10, F_MUL_S1_01, F_FMA_S2_01, F_ADD_S0_10, F_ADD_PLAIN, F_FMA_S1_01, F_FMA_S1_HI_10, NFORMS };
static const char* form_name[NFORMS] = {"pk_add src1 op_sel:[0,1]", "pk_add src1 op_sel:[0,1] neg", "pk_add src1 op_sel_hi:[1,0]", "pk_mul src1 op_sel_hi:[1,0]",
                                        "pk_mul src1 op_sel:[0,1]", "pk_fma src2 op_sel:[0,0,1]", "pk_add src0 op_sel:[1,0]", "pk_add plain",
                                        "pk_fma src1 op_sel:[0,1,0]", "pk_fma src1 op_sel_hi:[1,0,1]"};

// tmask: 8 = a transcendental just before the packed op (other tester-side traffic was not built)
template <int FORM>
__device__ __forceinline__ void tester(int lane, int wave, int iters, unsigned* bad, unsigned* badlane, int tmask, const float* gsrc, float* gdst,
                                       const unsigned char* lds) {
  unsigned nbad = 0;
  float4 gl = make_float4(0, 0, 0, 0), ll = make_float4(0, 0, 0, 0);
  float tr = 1.0f;
  for (int it = 0; it < iters; ++it) {
    if (tmask & 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(tr));
    const int xi = (lane * 3 + it) & 1023, yi = (lane * 5 + 2 * it + 1) & 1023, mi = 1 + ((it * 7 + wave) & 63), ji = 100 + ((it * 13 + 77) & 63);
    f2 x, m, r, g;
    x.x = (float)xi; x.y = (float)yi;
    m.x = (float)ji; m.y = (float)mi;
    g.x = 3.0f; g.y = 3.0f;
    int ex = 0, ey = 0;
    if (FORM == F_ADD_S1_01) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(m)); ex = xi + mi; ey = yi + mi; }
    if (FORM == F_ADD_S1_01_NEG) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(m)); ex = xi - mi; ey = yi - mi; }
    if (FORM == F_ADD_S1_HI_10) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(m)); ex = xi + ji; ey = yi + ji; }
    if (FORM == F_MUL_S1_HI_10) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(m)); ex = xi * ji; ey = yi * ji; }
    if (FORM == F_MUL_S1_01) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(m)); ex = xi * mi; ey = yi * mi; }
    if (FORM == F_FMA_S2_01) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(x), "v"(g), "v"(m)); ex = 3 * xi + mi; ey = 3 * yi + mi; }
    if (FORM == F_ADD_S0_10) { asm volatile("v_pk_add_f32 %0, %2, %1 op_sel:[1,0]" : "=v"(r) : "v"(x), "v"(m)); ex = xi + mi; ey = yi + mi; }
    if (FORM == F_ADD_PLAIN) { f2 m2; m2.x = m.y; m2.y = m.y; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m2)); ex = xi + mi; ey = yi + mi; }
    if (FORM == F_FMA_S1_01) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(x), "v"(m), "v"(g)); ex = xi * mi + 3; ey = yi * mi + 3; }
    if (FORM == F_FMA_S1_HI_10) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(m), "v"(g)); ex = xi * ji + 3; ey = yi * ji + 3; }
    if (r.x != (float)ex || r.y != (float)ey) {
      ++nbad;
      atomicAdd(&badlane[lane], 1u);
      const unsigned k = atomicAdd(&bad[2], 1u);
      if (k < 4) { float* rec = reinterpret_cast<float*>(bad + 16 + 8 * k); rec[0] = r.x; rec[1] = (float)ex; rec[2] = r.y; rec[3] = (float)ey; rec[4] = (float)lane; rec[5] = (float)mi; rec[6] = (float)ji; rec[7] = (float)xi; }
    }
  }
  if (nbad) atomicAdd(&bad[0], nbad);
  if (gl.x + ll.x + tr == 12345.678f) atomicAdd(&bad[3], 1u);
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
static void run(unsigned* bad, unsigned* badlane, float* sink, const _Float16* src, int iters, int pmask, int tmask = 0) {
  if (!g_dst) (void)hipMalloc(&g_dst, (size_t)256 * 65536 * 4);
  (void)hipMemset(bad, 0, 4096); (void)hipMemset(badlane, 0, 256);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(probe<FORM>, dim3(256), dim3(512), 65536, 0, bad, badlane, sink, src, iters, pmask, tmask, g_dst);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
  unsigned h[1024], hl[64];
  (void)hipMemcpy(h, bad, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(hl, badlane, 256, hipMemcpyDeviceToHost);
  printf("%-30s | tester mask %2d | partner mask %2d: %6u wrong lane-results in %.2g wave-instructions", form_name[FORM], tmask, pmask, h[0], (double)iters * 4 * 256);
  if (h[0]) {
    int lo = 64, hi = -1;
    for (int l = 0; l < 64; ++l) if (hl[l]) { lo = l < lo ? l : lo; hi = l; }
    const float* rec = reinterpret_cast<const float*>(h + 16);
    printf("  lanes %d..%d; e.g. got (%.0f, %.0f) expected (%.0f, %.0f) [x %.0f, m.lo %.0f, m.hi %.0f]", lo, hi, rec[0], rec[2], rec[1], rec[3], rec[7], rec[6], rec[5]);
  }
  printf("\n");
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400000;
  unsigned *bad, *badlane; float* sink; _Float16* src;
  (void)hipMalloc(&bad, 4096); (void)hipMalloc(&badlane, 256); (void)hipMalloc(&sink, 256 * 512 * 4); (void)hipMalloc(&src, 256 * 65536);
  (void)hipMemset(src, 0x2e, 256 * 65536);
  const int pm = argc > 2 ? atoi(argv[2]) : 5;
  { const int pms[] = {1, 17, 5, 21, 20, 16}; for (int i = 0; i < 6; ++i) run<1>(bad, badlane, sink, src, iters, pms[i], 0); }
  const int tm = argc > 3 ? atoi(argv[3]) : 0;
  run<0>(bad, badlane, sink, src, iters, pm, tm); run<2>(bad, badlane, sink, src, iters, pm, tm); run<3>(bad, badlane, sink, src, iters, pm, tm); run<4>(bad, badlane, sink, src, iters, pm, tm);
  run<5>(bad, badlane, sink, src, iters, pm, tm); run<6>(bad, badlane, sink, src, iters, pm, tm); run<7>(bad, badlane, sink, src, iters, pm, tm); run<8>(bad, badlane, sink, src, iters, pm, tm);
  run<9>(bad, badlane, sink, src, iters, pm, tm);
  return 0;
}
