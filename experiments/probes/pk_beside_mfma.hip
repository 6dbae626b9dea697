// Probe: does a packed-fp32 VALU instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, with and without op_sel) return a wrong
// half in some lanes while the OTHER wave of its SIMD issues MFMAs?  (Round 6: the post-norm epilogue of the barrier-free fc2 kernel
// lost the "- mean" of ONE half of one v_pk_add_f32 ... op_sel:[0,1] in lanes 48-63, ~15 tiles of 2066 per launch, only when the SIMD
// partner was already in the next tile's k-loop; the round-2 head-kernel deviation had the same ingredients.)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pk_probe experiments/probes/pk_beside_mfma.hip && /tmp/pk_probe
// Eight waves per workgroup, one workgroup per CU.  Waves 0-3 ("testers") run chains of packed ops on exactly representable values and
// compare with integer arithmetic; waves 4-7 run, by mode: 0 the same tester loop (lockstep), 1 an MFMA loop, 2 an MFMA loop with LDS
// reads, 3 s_sleep (idle partner), 4 a plain VALU loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int FORM>
__device__ __forceinline__ void tester(int lane, int wave, int iters, unsigned* bad, unsigned* badlane) {
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const int xi = (lane * 3 + it) & 1023, yi = (lane * 5 + 2 * it + 1) & 1023, mi = (it * 7 + wave) & 255, ji = (it * 13 + 77) & 255;
    const int gi = 1 + ((lane + it) & 3), bi = (it + lane) & 31;
    f2 x, m, g, bb, r;
    x.x = (float)xi; x.y = (float)yi;
    m.x = (float)ji; m.y = (float)mi;       // [junk, mean]: op_sel must pick .y for both halves
    g.x = (float)gi; g.y = (float)gi;
    bb.x = (float)bi; bb.y = (float)bi;
    f2 s;
    s.x = 2.0f; s.y = 12345.0f;             // [rstd, junk]: op_sel_hi:[1,0] must pick .x for both halves
    if (FORM == 0) {   // the failing epilogue's sequence
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                   "s_nop 0\n\t"
                   "v_pk_mul_f32 %0, %0, %3 op_sel_hi:[1,0]\n\t"
                   "s_nop 0\n\t"
                   "v_pk_fma_f32 %0, %0, %4, %5"
                   : "=&v"(r) : "v"(x), "v"(m), "v"(s), "v"(g), "v"(bb));
    } else if (FORM == 1) {   // no op_sel: operands splat into real pairs
      f2 m2, s2;
      m2.x = m.y; m2.y = m.y; s2.x = s.x; s2.y = s.x;
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
                   "s_nop 0\n\t"
                   "v_pk_mul_f32 %0, %0, %3\n\t"
                   "s_nop 0\n\t"
                   "v_pk_fma_f32 %0, %0, %4, %5"
                   : "=&v"(r) : "v"(x), "v"(m2), "v"(s2), "v"(g), "v"(bb));
    } else {                  // scalar forms
      asm volatile("v_sub_f32 %0, %2, %4\n\t"
                   "v_sub_f32 %1, %3, %4\n\t"
                   "v_mul_f32 %0, %0, %5\n\t"
                   "v_mul_f32 %1, %1, %5\n\t"
                   "v_fma_f32 %0, %0, %6, %7\n\t"
                   "v_fma_f32 %1, %1, %6, %7"
                   : "=&v"(r.x), "=&v"(r.y) : "v"(x.x), "v"(x.y), "v"(m.y), "v"(s.x), "v"(g.x), "v"(bb.x));
    }
    const int ex = (xi - mi) * 2 * gi + bi, ey = (yi - mi) * 2 * gi + bi;
    if (r.x != (float)ex || r.y != (float)ey) {
      ++nbad;
      atomicAdd(&badlane[lane], 1u);
      if (atomicAdd(&bad[1], 1u) < 8u) {
        const unsigned k = atomicAdd(&bad[2], 1u);
        if (k < 8) { float* rec = reinterpret_cast<float*>(bad + 16 + 8 * k); rec[0] = r.x; rec[1] = (float)ex; rec[2] = r.y; rec[3] = (float)ey; rec[4] = (float)lane; rec[5] = (float)mi; rec[6] = (float)ji; rec[7] = (float)xi; }
      }
    }
  }
  if (nbad) atomicAdd(&bad[0], nbad);
}

template <int FORM>
__global__ __launch_bounds__(512) void probe(unsigned* bad, unsigned* badlane, float* sink, int iters, int mode) {
  __shared__ __attribute__((aligned(16))) _Float16 sm[8192];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += 512) sm[i] = (_Float16)(0.001f * (float)((i * 7) & 255));
  __syncthreads();
  if (wave < 4 || mode == 0) { tester<FORM>(lane, wave, iters, bad, badlane); return; }
  if (mode == 3) { for (int i = 0; i < iters / 8; ++i) __builtin_amdgcn_s_sleep(8); return; }
  if (mode == 4) {
    float v = (float)lane;
    for (int i = 0; i < iters * 6; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    sink[blockIdx.x * 512 + threadIdx.x] = v;
    return;
  }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  h8 a = *reinterpret_cast<const h8*>(sm + lane * 8), b = *reinterpret_cast<const h8*>(sm + 512 + lane * 8);
  for (int it = 0; it < iters / 2; ++it) {
    if (mode == 2) {
      a = *reinterpret_cast<const h8*>(sm + ((lane * 8 + it * 64) & 8184));
      b = *reinterpret_cast<const h8*>(sm + ((lane * 8 + it * 64 + 4096) & 8184));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[blockIdx.x * 512 + threadIdx.x] = t;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200000;
  unsigned *bad, *badlane; float* sink;
  hipMalloc(&bad, 4096); hipMalloc(&badlane, 256); hipMalloc(&sink, 256 * 512 * 4);
  const char* mname[] = {"partner: same tester loop (lockstep)", "partner: MFMA loop", "partner: MFMA loop + LDS reads", "partner: s_sleep", "partner: VALU loop"};
  const char* fname[] = {"packed, op_sel", "packed, no op_sel", "scalar"};
  for (int form = 0; form < 3; ++form)
    for (int mode = 0; mode < 5; ++mode) {
      hipMemset(bad, 0, 4096); hipMemset(badlane, 0, 256);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      if (form == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, bad, badlane, sink, iters, mode);
      else if (form == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, bad, badlane, sink, iters, mode);
      else hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, bad, badlane, sink, iters, mode);
      hipEventRecord(e1, 0);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      unsigned h[1024], hl[64];
      hipMemcpy(h, bad, 4096, hipMemcpyDeviceToHost); hipMemcpy(hl, badlane, 256, hipMemcpyDeviceToHost);
      const double checks = (double)iters * 64 * (mode == 0 ? 8 : 4) * 256;
      printf("[%-17s] %-40s: %u wrong of %.3g checks (%.1f ms)", fname[form], mname[mode], h[0], checks, ms);
      if (h[0]) {
        printf("  lanes:");
        for (int l = 0; l < 64; ++l) if (hl[l]) printf(" %d:%u", l, hl[l]);
        const float* rec = reinterpret_cast<const float*>(h + 16);
        const unsigned n = h[2] < 8 ? h[2] : 8;
        for (unsigned k = 0; k < n && k < 3; ++k)
          printf("\n      got (%.0f, %.0f) expected (%.0f, %.0f) lane %.0f mean %.0f junk %.0f x %.0f", rec[8 * k], rec[8 * k + 2], rec[8 * k + 1], rec[8 * k + 3], rec[8 * k + 4], rec[8 * k + 5], rec[8 * k + 6], rec[8 * k + 7]);
      }
      printf("\n");
    }
  return 0;
}
