// Second probe (see pk_beside_mfma.hip): WHICH packed-fp32 forms lose a half beside the SIMD partner's MFMAs, and beside which MFMA.
//   hipcc --offload-arch=gfx950 -O2 -o pk_probe2.bin experiments/probes/pk_beside_mfma2.hip ; ./pk_probe2.bin [iters]
// One instruction per check, exactly representable integers, expected value from integer arithmetic.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// forms: r = f(x, m); x = (xi, yi), m = (ji, mi)
enum { F_ADD_S1_01 = 0,      // v_pk_add_f32 op_sel:[0,1]                      -> (xi + mi, yi + mi)
       F_ADD_S1_01_NEG,      // ... neg_lo:[0,1] neg_hi:[0,1]                   -> (xi - mi, yi - mi)
       F_ADD_S1_HI_10,       // v_pk_add_f32 op_sel_hi:[1,0]                   -> (xi + ji, yi + ji)
       F_MUL_S1_HI_10,       // v_pk_mul_f32 op_sel_hi:[1,0]                   -> (xi ji, yi ji)
       F_MUL_S1_01,          // v_pk_mul_f32 op_sel:[0,1]                      -> (xi mi, yi mi)
       F_FMA_S2_01,          // v_pk_fma_f32 x, g, m op_sel:[0,0,1] op_sel_hi:[1,1,1] -> (xi g + mi, yi g + mi)
       F_ADD_S0_10,          // v_pk_add_f32 m, x op_sel:[1,0] (src0 broadcast hi) -> (mi + xi, mi + yi)
       F_ADD_PLAIN,          // v_pk_add_f32 (no op_sel), m2 = (mi, mi)
       F_ADD_SWAP,           // v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]: (xi + mi, yi + ji)  (swapped halves of src1)
       F_MOV_S_01,           // v_pk_mov_b32 r, m, m op_sel:[1,1] op_sel_hi... : broadcast by pk_mov (lo <- m.hi, hi <- m.hi)
       NFORMS };
static const char* form_name[NFORMS] = {"pk_add op_sel:[0,1]", "pk_add op_sel:[0,1] neg", "pk_add op_sel_hi:[1,0]", "pk_mul op_sel_hi:[1,0]",
                                        "pk_mul op_sel:[0,1]", "pk_fma src2 op_sel:[0,0,1]", "pk_add src0 op_sel:[1,0]", "pk_add plain",
                                        "pk_add swapped halves", "pk_mov_b32 broadcast hi"};

template <int FORM>
__device__ __forceinline__ void tester(int lane, int wave, int iters, unsigned* bad, unsigned* badlane) {
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const int xi = (lane * 3 + it) & 1023, yi = (lane * 5 + 2 * it + 1) & 1023, mi = 1 + ((it * 7 + wave) & 63), ji = 100 + ((it * 13 + 77) & 63);
    f2 x, m, r;
    x.x = (float)xi; x.y = (float)yi;
    m.x = (float)ji; m.y = (float)mi;
    int ex = 0, ey = 0;
    if (FORM == F_ADD_S1_01) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(m)); ex = xi + mi; ey = yi + mi; }
    if (FORM == F_ADD_S1_01_NEG) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(m)); ex = xi - mi; ey = yi - mi; }
    if (FORM == F_ADD_S1_HI_10) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(m)); ex = xi + ji; ey = yi + ji; }
    if (FORM == F_MUL_S1_HI_10) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(m)); ex = xi * ji; ey = yi * ji; }
    if (FORM == F_MUL_S1_01) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(m)); ex = xi * mi; ey = yi * mi; }
    if (FORM == F_FMA_S2_01) {
      f2 g; g.x = 3.0f; g.y = 3.0f;
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(x), "v"(g), "v"(m)); ex = 3 * xi + mi; ey = 3 * yi + mi;
    }
    if (FORM == F_ADD_S0_10) { asm volatile("v_pk_add_f32 %0, %2, %1 op_sel:[1,0]" : "=v"(r) : "v"(x), "v"(m)); ex = xi + mi; ey = yi + mi; }
    if (FORM == F_ADD_PLAIN) { f2 m2; m2.x = m.y; m2.y = m.y; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m2)); ex = xi + mi; ey = yi + mi; }
    if (FORM == F_ADD_SWAP) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(m)); ex = xi + mi; ey = yi + ji; }
    if (FORM == F_MOV_S_01) { asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,1]" : "=v"(r) : "v"(m)); ex = mi; ey = mi; }
    if (r.x != (float)ex || r.y != (float)ey) {
      ++nbad;
      atomicAdd(&badlane[lane], 1u);
      const unsigned k = atomicAdd(&bad[2], 1u);
      if (k < 4) { float* rec = reinterpret_cast<float*>(bad + 16 + 8 * k); rec[0] = r.x; rec[1] = (float)ex; rec[2] = r.y; rec[3] = (float)ey; rec[4] = (float)lane; rec[5] = (float)mi; rec[6] = (float)ji; rec[7] = (float)xi; }
    }
  }
  if (nbad) atomicAdd(&bad[0], nbad);
}

// partner modes: 0 same tester, 1 mfma 16x16x32 f16, 2 mfma 32x32x16 f16, 3 mfma 16x16x32 bf16, 4 mfma_f32_16x16x4_f32, 5 VALU loop
template <int FORM>
__global__ __launch_bounds__(512) void probe(unsigned* bad, unsigned* badlane, float* sink, int iters, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < 4 || mode == 0) { tester<FORM>(lane, wave, iters, bad, badlane); return; }
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (float)((lane + i) & 15)); b[i] = (_Float16)(0.02f * (float)((lane * 3 + i) & 15)); }
  float t = 0.f;
  if (mode == 1 || mode == 3) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters / 2; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = mode == 1 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][3];
  } else if (mode == 2) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters / 2; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][15];
  } else if (mode == 4) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters / 2; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[0], (float)b[0], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][3];
  } else {
    float v = (float)lane;
    for (int i = 0; i < iters * 6; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    t = v;
  }
  sink[blockIdx.x * 512 + threadIdx.x] = t;
}

template <int FORM>
static void run(unsigned* bad, unsigned* badlane, float* sink, int iters) {
  const char* mname[] = {"same loop", "mfma 16x16x32 f16", "mfma 32x32x16 f16", "mfma 16x16x32 bf16", "mfma 16x16x4 f32", "VALU loop"};
  for (int mode = 0; mode < 6; ++mode) {
    (void)hipMemset(bad, 0, 4096); (void)hipMemset(badlane, 0, 256);
    hipLaunchKernelGGL(probe<FORM>, dim3(256), dim3(512), 0, 0, bad, badlane, sink, iters, mode);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    unsigned h[1024], hl[64];
    (void)hipMemcpy(h, bad, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(hl, badlane, 256, hipMemcpyDeviceToHost);
    const double winst = (double)iters * (mode == 0 ? 8 : 4) * 256;
    printf("%-28s | partner %-18s: %6u wrong lane-results in %.2g wave-instructions", form_name[FORM], mname[mode], h[0], winst);
    if (h[0]) {
      int lo = 64, hi = -1;
      for (int l = 0; l < 64; ++l) if (hl[l]) { lo = l < lo ? l : lo; hi = l; }
      const float* rec = reinterpret_cast<const float*>(h + 16);
      printf("  lanes %d..%d; e.g. got (%.0f, %.0f) expected (%.0f, %.0f) [x %.0f, m.lo %.0f, m.hi %.0f]", lo, hi, rec[0], rec[2], rec[1], rec[3], rec[7], rec[6], rec[5]);
    }
    printf("\n");
    fflush(stdout);
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 1000000;
  unsigned *bad, *badlane; float* sink;
  (void)hipMalloc(&bad, 4096); (void)hipMalloc(&badlane, 256); (void)hipMalloc(&sink, 256 * 512 * 4);
  run<0>(bad, badlane, sink, iters); run<1>(bad, badlane, sink, iters); run<2>(bad, badlane, sink, iters); run<3>(bad, badlane, sink, iters);
  run<4>(bad, badlane, sink, iters); run<5>(bad, badlane, sink, iters); run<6>(bad, badlane, sink, iters); run<7>(bad, badlane, sink, iters);
  run<8>(bad, badlane, sink, iters); run<9>(bad, badlane, sink, iters);
  return 0;
}
