// The fc1 epilogue of the F16X3 flow and what it needs (the erfc-series GELU, the hi / lo split, the FX flags), shared by the token GEMM
// (gemm_x3p_epilogue.h) and the dedicated fc1 kernel (kernels_fc1_x3.hip).  Included inside namespace d3d AFTER the includer has defined
// f32x4, h8, P_A_SCALE, range_note(rw, amax) and D3D_PATCH_FENCE.
#pragma once

// Two elements at a time: the epilogues are VALU-bound (the fc1 one: 128 outputs per lane), and gfx950 issues v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32 on register pairs at the rate of the scalar forms.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat2(float a) { return (f2)(a); }
// (a, a) as a register pair the compiler cannot see through -- for every RUN-TIME scalar that meets packed arithmetic.  Reason: a gfx950
// deviation found in round 6 (experiments/probes/pk_beside_mfma*.hip, experiments/NOTES.md section 000): v_pk_{add,mul,fma}_f32 whose
// SRC1 low half selects the HIGH dword of its register pair (op_sel:[.,1,..] -- what the compiler emits when it folds a broadcast of a
// value that sits in an odd register) now and then computes the low result of lanes 48-63 with src1 = 0 while the SIMD's other wave
// STARTS issuing MFMAs after the matrix pipe has been idle.  With the pair opaque there is nothing to fold; tests/test_abi_host.py pins the absence of that form
// in every kernel of the library.
__device__ __forceinline__ f2 splat2_rt(float a) {
#ifdef D3D_EXP_PLAIN_SPLAT   // (A/B of what the opaque pair costs: experiments/build_variant.sh plain "-DD3D_EXP_PLAIN_SPLAT")
  return (f2)(a);
#endif
  f2 r;
  r.x = a; r.y = a;
  asm volatile("" : "+v"(r));
  return r;
}
// GELU(x) = 0.5 x + |x| (0.5 - q),  q = 0.5 erfc(|x| / sqrt 2) = t (a1 + t (a2 + ... a5 t)) exp(-x^2 / 2) / 2,  t = 1 / (1 + p |x| / sqrt 2)
// (Abramowitz-Stegun 7.1.26, coefficients halved).  Per pair of values: 2 + 2 scalar instructions that take |x| as a source
// modifier, 9 packed ones, v_rcp and v_exp twice -- four issue slots fewer than the form max(x, 0) - (|x| / 2) (p t) e, which
// had to materialise |x| for its packed multiplies.  Absolute error < 1e-7 |x|.
__device__ __forceinline__ f2 gelu_fast2(f2 x) {
  f2 t, e, r;
  constexpr float KP = 0.3275911f * 0.70710678118654752440f;
  t.x = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x.x), KP, 1.0f));
  t.y = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(x.y), KP, 1.0f));
  f2 p = fma2(splat2(0.5f * 1.061405429f), t, splat2(0.5f * -1.453152027f));
  p = fma2(p, t, splat2(0.5f * 1.421413741f));
  p = fma2(p, t, splat2(0.5f * -0.284496736f));
  p = fma2(p, t, splat2(0.5f * 0.254829592f));
  const f2 xx = (x * x) * (-0.5f * 1.44269504088896340736f);
  e.x = __builtin_amdgcn_exp2f(xx.x); e.y = __builtin_amdgcn_exp2f(xx.y);
  const f2 w = fma2(-(p * t), e, splat2(0.5f));
  const f2 hx = x * 0.5f;
  r.x = __builtin_fmaf(__builtin_fabsf(x.x), w.x, hx.x);
  r.y = __builtin_fmaf(__builtin_fabsf(x.y), w.y, hx.y);
  return r;
}
// 8 values -> fp16 (hi, lo) of osc * v (osc a power of two).  v_fma_mixlo / mixhi_f16 scale, subtract the fp16 hi half (read
// in place) and convert in one instruction: hi = fp16(osc v), lo = fp16(osc v - hi), both single roundings of exact fp32
// quantities -- 2 instructions per value where multiply, clamp, convert, convert back, subtract, convert took 5.  Nothing is
// clamped: beyond the fp16 range hi becomes inf (and lo NaN), and that is exactly where the range guard fires -- amax is the
// largest |v| seen (UNSCALED: the caller notes amax * osc).
template <bool GUARD = true>
__device__ __forceinline__ void split8_x3(const f2 (&v)[4], float osc, h8& oh, h8& ol, float& amax) {
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  u32x4_ hv, lv;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned hi, lo;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v[e].x), "v"(osc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v[e].y), "v"(osc));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(v[e].x), "v"(osc), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v[e].y), "v"(osc), "v"(hi));
    hv[e] = hi; lv[e] = lo;
    if (GUARD) amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v[e].x)), __builtin_fabsf(v[e].y));
  }
  oh = __builtin_bit_cast(h8, hv);
  ol = __builtin_bit_cast(h8, lv);
}
// FX flags of the folded forms (X3Fold in d3d_kernels.h)
constexpr int FX_LNF = 1;   // LayerNorm folded into this GEMM: per-row (rstd, -mean rstd) from LDS, csum per column
constexpr int FX_RP = 2;    // residual from pair-layout planes
constexpr int FX_SO = 4;    // per-row (sum, sum of squares) of the output rows -> st_out
constexpr int FX_PN = 8;    // the tile spans whole rows: post-norm of the new rows in the epilogue (X3PostNorm)
constexpr int FX_BF16 = 16; // bf16 operand mode (D3D_PREC_BF16): operands are plain bf16 rows, a 128-byte line = 64 k values, ONE bf16
                            // MFMA per product (two per line); OUTSPLIT 3 = bf16 row-major output (the next GEMM's / the attention's operand)

// GELU + pair output straight from the accumulators (fc1 -> hidden activation).  The hidden activation is only ever the A
// operand of the fc2 GEMM, so its k order inside a 32-column group is free: "accumulator order" (pair_col_acc, d3d_kernels.h;
// the fc2 weight is split in the same order at commit) makes the 8 values a lane holds of a group one 16-byte piece.  No LDS
// transpose, no barrier: lane (m = lane & 15, q = lane >> 4) owns row 16 i + m of m-tile i and columns 16 j + 4 q + r.
template <int TM, int WM, int WN, int FX, bool CHECK>
__device__ __forceinline__ void x3q_epilogue_acc(f32x4 (&acc)[TM][4], unsigned char* lds_x, const float* __restrict__ bias,
                                                 _Float16* Cht, const float* __restrict__ csum, int mt0, int nt0, int rbase, int lane,
                                                 int M, int N, int gl, int gh, const float P_OUT_SCALE, unsigned* rw) {
  const int m16 = lane & 15, q4 = lane >> 4;
  f2 bb[4][2], cs[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = nt0 + 16 * j + 4 * q4;
    bb[j][0] = bb[j][1] = cs[j][0] = cs[j][1] = splat2(0.f);
    if (!CHECK || n < N) {
      if (bias) {
        const float4 t = *reinterpret_cast<const float4*>(bias + n);
        bb[j][0].x = t.x; bb[j][0].y = t.y; bb[j][1].x = t.z; bb[j][1].y = t.w;
      }
      if (FX & FX_LNF) {
        const float4 t = *reinterpret_cast<const float4*>(csum + n);
        cs[j][0].x = t.x; cs[j][0].y = t.y; cs[j][1].x = t.z; cs[j][1].y = t.w;
      }
    }
  }
  const float2* srow = reinterpret_cast<const float2*>(lds_x);
  char* Chb = reinterpret_cast<char*>(Cht);
  const unsigned ob = (unsigned)(m16 * 2 * N + 8 * q4) * 2u;        // row m16, piece q4 of the wave's first group (hi; lo 64 B on)
  const unsigned rstep = (unsigned)N * 64u;                          // 16 rows of the pair buffer
  float amax = 0.0f;   // range guard
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i < gl || i >= gh) continue;
    const int row = 16 * i + m16;
    if (CHECK && mt0 + row >= M) continue;
    f2 sx = splat2(1.f), sy = splat2(0.f);
    if (FX & FX_LNF) {
      const float2 st = srow[rbase + row];
      sx = splat2(st.x); sy = splat2(st.y);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (CHECK && nt0 + 32 * c >= N) continue;
      f2 v[4];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * c + jj;
        f2 a0, a1;
        a0.x = acc[i][j][0]; a0.y = acc[i][j][1]; a1.x = acc[i][j][2]; a1.y = acc[i][j][3];
        if (FX & FX_LNF) {   // LN(x) W^T + b = rstd (x W'^T) - rstd mean csum + b'
          v[2 * jj] = fma2(sx, a0, fma2(sy, cs[j][0], bb[j][0]));   // (sx = rstd * out_scale)
          v[2 * jj + 1] = fma2(sx, a1, fma2(sy, cs[j][1], bb[j][1]));
        } else {
          v[2 * jj] = fma2(a0, splat2(P_OUT_SCALE), bb[j][0]);
          v[2 * jj + 1] = fma2(a1, splat2(P_OUT_SCALE), bb[j][1]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_fast2(v[e]);
      h8 oh, ol;
      split8_x3(v, P_A_SCALE, oh, ol, amax);
      *reinterpret_cast<h8*>(Chb + (ob + (unsigned)i * rstep + (unsigned)c * 128u)) = oh;
      *reinterpret_cast<h8*>(Chb + (ob + (unsigned)i * rstep + (unsigned)c * 128u) + 64u) = ol;
    }
    if (i & 1) __builtin_amdgcn_sched_barrier(0);
  }
  range_note(rw, amax * P_A_SCALE);
}

