// What the epilogues of the F16X3 token GEMM need from their includer (kernels_gemm_x3p.hip, kernels_proj_x3.hip), inside namespace d3d:
// the operand typedefs, the scales, the range guard, the 4-column split store and the patch fence.
#pragma once

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int PBK = 32;                          // k-tile depth (fp16 elements)
constexpr float P_A_SCALE = 8.0f;

// F16X3 range guard (d3d_kernels.h): every plane writer tracks max |scaled value| per lane; a lane whose value left the fp16
// range (|x| > 8188) ORs bit 0 into the launching engine's sticky word (X3Tail::range) once, at the end of its epilogue.
__device__ __forceinline__ void range_note(unsigned* rw, float amax) {
#ifndef D3D_NO_RANGE_GUARD
  if (amax > X3_HALF_MAX) range_raise(rw, RANGE_BIT_ACT);
#endif
}
// Bit 1: a LayerNorm folded into a GEMM met a row whose mean dwarfs its spread.  The folded form has the row's ONE-PASS
// statistics (sum, sum of squares from the producer's epilogue): var = E[x^2] - mean^2 loses relative accuracy like
// eps (1 + mean^2 / var) -- 4x the two-pass error at |mean| = 8 sigma, the 1e-4 parity gate near 25 sigma (measured:
// test_folded_layernorm_statistics_with_offset_rows).  Raised from |mean| > 16 sigma on; remedy as for bit 0: precision fp32.
__device__ __forceinline__ void range_note_stats(unsigned* rw, float mean, float var) {
#ifndef D3D_NO_RANGE_GUARD
  if (mean * mean > 256.0f * var) range_raise(rw, RANGE_BIT_STATS);
#endif
}

// OUTSPLIT: 0 = fp32 C; 1 = hi/lo PLANES of C (two [M][N] fp16 matrices: the temporal attention kernel reads q/k/v
// that way); 2 = PAIR layout (the consumer is another x3 GEMM).  Both carry 8*c (columns < qcols: 1*c, the q third
// of a temporal qkv GEMM, which absorbs the dh^-0.5 = 2^-3 attention scale).
template <int OUTSPLIT, bool GUARD = true>
__device__ __forceinline__ void store_split4(const float (&v)[4], float osc, _Float16* Cht, _Float16* Clt, int off, int poff,
                                             float& amax) {
  h4 hh, ll;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float raw = v[e] * osc;
    if (GUARD) amax = __builtin_fmaxf(amax, __builtin_fabsf(raw));
    const float sc = __builtin_amdgcn_fmed3f(raw, -65504.0f, 65504.0f);
    hh[e] = (_Float16)sc;
    ll[e] = (_Float16)(sc - (float)hh[e]);
  }
  if (OUTSPLIT == 2) {
    *reinterpret_cast<h4*>(Cht + poff) = hh;
    *reinterpret_cast<h4*>(Cht + poff + PAIR_LO) = ll;
  } else {
    *reinterpret_cast<h4*>(Cht + off) = hh;
    *reinterpret_cast<h4*>(Clt + off) = ll;
  }
}

// Between a wave's writes to its LDS patch and its reads of OTHER lanes' rows of it: nothing orders them for the compiler (one
// thread's load does not alias its own stores), LDS itself executes a wave's operations in order.  A compiler-level barrier.
#define D3D_PATCH_FENCE() asm volatile("" ::: "memory")
