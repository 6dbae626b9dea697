// HBM-bound row kernels of the DDIM step (gfx950): input embedding, LayerNorm (pre/post/head), time-embedding table,
// regression head + DDIM update, seq2frame frame reduce, q_sample, flip-TTA merge + MPJPE.
// One 64-lane wave owns one token row (D floats as float4 per lane) so every reduction is a wave shuffle tree and
// every HBM access is a 16-byte-per-lane coalesced stream.
#include <cstdlib>
#include "d3d_kernels.h"

namespace d3d {

typedef _Float16 h4v __attribute__((ext_vector_type(4)));

// fp32 -> (hi, lo) fp16 pair of 8*x: the operand format of the F16X3 GEMM (kernels_gemm_x3p.hip)
__device__ __forceinline__ void split4_x3(const float4 v, h4v& hi, h4v& lo, float& amax) {
  const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    amax = fmaxf(amax, fabsf(f[j] * 8.0f));
    const float s = __builtin_amdgcn_fmed3f(f[j] * 8.0f, -65504.0f, 65504.0f);
    hi[j] = (_Float16)s;
    lo[j] = (_Float16)(s - (float)hi[j]);
  }
}

constexpr int WAVES_PER_BLOCK = 4;
constexpr int LN_MAXV = 4;  // float4 per lane -> D <= 1024

// Sum over the 64 lanes of a wave, the total in every lane.  In-row (16 lanes) butterfly by DPP -- quad xor 1, xor 2,
// half-row mirror, row mirror --, then the four row totals through v_readlane: no ds_bpermute (what __shfl_xor compiles to), no
// LDS traffic.  (The ds_bpermute butterfly was once suspected of the head kernel's deviation on a shared GPU and was EXCLUDED
// by experiment -- 0 wrong sums of 10^10 in experiments/bperm_probe.hip, and this DPP form deviated as well; see k_head.)
__device__ __forceinline__ float wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

// In-register LayerNorm of one row spread over a wave.  v[i] holds columns 4*(lane + 64 i) .. +3 (valid if < D).
template <int NV>
__device__ __forceinline__ void ln_row(float4 (&v)[NV], int D, int lane, const float* __restrict__ g,
                                       const float* __restrict__ b, float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (4 * (lane + 64 * i) < D) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (4 * (lane + 64 * i) < D) {
      const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (lane + 64 * i);
    if (c < D) {
      const float4 gg = *reinterpret_cast<const float4*>(g + c);
      const float4 bb = *reinterpret_cast<const float4*>(b + c);
      v[i].x = (v[i].x - mean) * rstd * gg.x + bb.x;
      v[i].y = (v[i].y - mean) * rstd * gg.y + bb.y;
      v[i].z = (v[i].z - mean) * rstd * gg.z + bb.z;
      v[i].w = (v[i].w - mean) * rstd * gg.w + bb.w;
    }
  }
}

// (sum, sum of squares) contribution of four values, every operation stated (no contraction left to the compiler): the row kernel
// and the embedding kernel that writes the same statistics form them identically
__device__ __forceinline__ void sums4(const float4 v, float& sm, float& sq) {
  sm = __fadd_rn(sm, __fadd_rn(__fadd_rn(v.x, v.y), __fadd_rn(v.z, v.w)));
  sq = __fadd_rn(sq, __fadd_rn(__fmaf_rn(v.x, v.x, __fmul_rn(v.y, v.y)), __fmaf_rn(v.z, v.z, __fmul_rn(v.w, v.w))));
}

__device__ __forceinline__ void add4(float4& a, const float4 b) {
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
}

// ------------------------------------------------------------------------------------------------ LayerNorm (fused)
template <int NV>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_layernorm(LnArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int D = a.D;
  float4 v[NV];
  float amax = 0.0f;   // range guard of the plane outputs
  const float* xr = a.x + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * (lane + 64 * i);
    v[i] = (c < D) ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0, 0, 0, 0);
  }
  if (!a.skip_ln1) ln_row<NV>(v, D, lane, a.g1, a.b1, a.eps1);
  if (a.pos) {  // S2S:238-242 (Temporal_pos_embed is added after Spatial_norm, before the block's time-emb add)
    const float* pr = a.pos + (size_t)((row / a.pos_div) % a.pos_mod) * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
      if (c < D) add4(v[i], *reinterpret_cast<const float4*>(pr + c));
    }
  }
  if (a.tvec) {  // S2S:113-116 of the NEXT block: x = x + time_mlp_i(t)[:,None,None,:]
    const float* tr = a.tvec + (size_t)(row / a.rows_per_batch) * a.tvec_stride;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
      if (c < D) add4(v[i], *reinterpret_cast<const float4*>(tr + c));
    }
  }
  if (a.y) {
    float* yr = a.y + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
      if (c < D) *reinterpret_cast<float4*>(yr + c) = v[i];
    }
  }
  if (a.y_x3) {   // the residual stream itself as GEMM operand planes (the consuming GEMM folds the LayerNorm, X3Fold)
    _Float16* yp = reinterpret_cast<_Float16*>(a.y_x3) + (size_t)row * 2 * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
      if (c < D) {
        h4v hi, lo;
        split4_x3(v[i], hi, lo, amax);
        *reinterpret_cast<h4v*>(yp + pair_col(c)) = hi;
        *reinterpret_cast<h4v*>(yp + pair_col(c) + PAIR_LO) = lo;
      }
    }
    if (amax > X3_HALF_MAX) range_raise(a.range, RANGE_BIT_ACT);
  }
  if (a.stats) {
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (4 * (lane + 64 * i) < D) sums4(v[i], sm, sq);
    sm = wave_sum(sm);
    sq = wave_sum(sq);
    if (lane == 0) *reinterpret_cast<float2*>(a.stats + 2 * (size_t)row) = make_float2(sm, sq);
  }
  if (a.h || a.h_x3 || a.h_bf16) {
    if (a.y || a.y_x3) ln_row<NV>(v, D, lane, a.g2, a.b2, a.eps2);
    if (a.h_bf16) {   // bf16 operand rows (round to nearest even)
      typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
      __bf16* hp = reinterpret_cast<__bf16*>(a.h_bf16) + (size_t)row * D;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = 4 * (lane + 64 * i);
        if (c < D) {
          bf4 o;
          o[0] = (__bf16)v[i].x; o[1] = (__bf16)v[i].y; o[2] = (__bf16)v[i].z; o[3] = (__bf16)v[i].w;
          *reinterpret_cast<bf4*>(hp + c) = o;
        }
      }
    } else if (a.h_x3) {   // F16X3 pair layout: 8 lanes fill one 128-byte line (64 B of hi, 64 B of lo) of the row
      _Float16* hp = reinterpret_cast<_Float16*>(a.h_x3) + (size_t)row * 2 * D;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = 4 * (lane + 64 * i);
        if (c < D) {
          h4v hi, lo;
          split4_x3(v[i], hi, lo, amax);
          *reinterpret_cast<h4v*>(hp + pair_col(c)) = hi;
          *reinterpret_cast<h4v*>(hp + pair_col(c) + PAIR_LO) = lo;
        }
      }
      if (amax > X3_HALF_MAX) range_raise(a.range, RANGE_BIT_ACT);
    } else {
      float* hr = a.h + (size_t)row * D;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = 4 * (lane + 64 * i);
        if (c < D) *reinterpret_cast<float4*>(hr + c) = v[i];
      }
    }
  }
}

hipError_t launch_layernorm(const LnArgs& a_, hipStream_t s) {
  LnArgs a = a_;
  a.range = launch_range_word();   // F16X3 range guard: the launching engine's word (d3d_kernels.h)
  if (a.rows <= 0 || a.D <= 0 || (a.D & 3) || a.D > 256 * LN_MAXV) return hipErrorInvalidValue;
  const int grid = (a.rows + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  if (a.D <= 256)
    hipLaunchKernelGGL(k_layernorm<1>, dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, s, a);
  else if (a.D <= 512)
    hipLaunchKernelGGL(k_layernorm<2>, dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, s, a);
  else
    hipLaunchKernelGGL(k_layernorm<4>, dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ input embedding
// X[m, c] = sum_k in[m,k] Wf[c,k] + bf[c] + spos[j,c] (+ tvec[b,c]);  in = cat(x2d[m], y[m or (b,j)]) (DIFF:255 / S2S:250).
// A thread owns 4 columns and keeps their CIN weights + bias in registers; the block walks a run of tokens, so the
// kernel is a pure streaming write of X (the 5 input scalars of a token are wave-uniform broadcast loads).
template <int CIN2>
__global__ __launch_bounds__(256) void k_embed(const float* __restrict__ x2d, const float* __restrict__ y,
                                               const float* __restrict__ Wf, const float* __restrict__ bf,
                                               const float* __restrict__ spos, const float* __restrict__ tvec,
                                               int64_t tvec_stride, float* __restrict__ X, int M, int T, int J, int D,
                                               int y_bcast_T, int tokens_per_block) {
  constexpr int CIN = CIN2 + 3;
  const int D4 = D >> 2;
  const int cgroups = (D4 + 255) / 256;                 // column groups of 256 threads x 4 columns
  const int cg = blockIdx.x % cgroups;
  const int c = (cg * 256 + threadIdx.x) * 4;
  if (c >= D) return;
  float w[4][CIN], bias[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int k = 0; k < CIN; ++k) w[q][k] = Wf[(size_t)(c + q) * CIN + k];
    bias[q] = bf[c + q];
  }
  const int m_begin = (blockIdx.x / cgroups) * tokens_per_block;
  const int m_end = min(M, m_begin + tokens_per_block);
  for (int m = m_begin; m < m_end; ++m) {
    const int j = m % J;
    const int b = m / (T * J);
    float in[CIN];
#pragma unroll
    for (int k = 0; k < CIN2; ++k) in[k] = x2d[(size_t)m * CIN2 + k];
    const size_t my = y_bcast_T ? ((size_t)b * J + j) : (size_t)m;  // DIFF-S2F:281: y.repeat(1, f, 1, 1)
#pragma unroll
    for (int k = 0; k < 3; ++k) in[CIN2 + k] = y[my * 3 + k];
    const float4 sp = *reinterpret_cast<const float4*>(spos + (size_t)j * D + c);
    float4 tv = make_float4(0, 0, 0, 0);
    if (tvec) tv = *reinterpret_cast<const float4*>(tvec + (size_t)b * tvec_stride + c);
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < CIN; ++k) acc = fmaf(in[k], w[q][k], acc);
      o[q] = acc + bias[q];
    }
    o[0] += sp.x; o[1] += sp.y; o[2] += sp.z; o[3] += sp.w;
    if (tvec) { o[0] += tv.x; o[1] += tv.y; o[2] += tv.z; o[3] += tv.w; }
    *reinterpret_cast<float4*>(X + (size_t)m * D + c) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

hipError_t launch_embed(const float* x2d, const float* y, const float* Wf, const float* bf, const float* spos,
                        const float* tvec, int64_t tvec_stride, float* X, int B, int T, int J, int D, int in_chans,
                        int y_bcast_T, hipStream_t s) {
  if ((D & 3) || in_chans < 1 || in_chans > 5) return hipErrorInvalidValue;
  const int M = B * T * J;
  const int cgroups = ((D >> 2) + 255) / 256;
  const int tpb = 32;
  const unsigned grid = (unsigned)(((M + tpb - 1) / tpb) * cgroups);
#define D3D_EMBED(C2)                                                                                                   \
  case C2:                                                                                                              \
    hipLaunchKernelGGL(k_embed<C2>, dim3(grid), dim3(256), 0, s, x2d, y, Wf, bf, spos, tvec, tvec_stride, X, M, T, J, D, \
                       y_bcast_T, tpb);                                                                                 \
    break;
  switch (in_chans) {
    D3D_EMBED(1) D3D_EMBED(2) D3D_EMBED(3) D3D_EMBED(4) D3D_EMBED(5)
  }
#undef D3D_EMBED
  return hipGetLastError();
}

// The same embedding for the plane-resident F16X3 flow (D = 512): one wave per token row, lane l holds columns 4 l .. + 3 and
// 256 + 4 l .. + 3 -- the layout of k_layernorm<2> --, and writes what embed + the stream-entry row kernel used to produce in two
// passes over HBM: the fp16 (hi, lo) planes of 8 x in the pair layout and the row's (sum, sum of squares) for the LayerNorm folded into
// the first qkv GEMM.  Same fma chain per element as k_embed, same split and the same stated summation (sums4, wave_sum) as k_layernorm.
template <int CIN2>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_embed_planes(const float* __restrict__ x2d, const float* __restrict__ y,
                                                                       const float* __restrict__ Wf, const float* __restrict__ bf,
                                                                       const float* __restrict__ spos, const float* __restrict__ tvec,
                                                                       int64_t tvec_stride, _Float16* __restrict__ XP,
                                                                       float* __restrict__ stats, int M, int T, int J, int y_bcast_T,
                                                                       int tokens_per_wave, unsigned* rw) {
  constexpr int CIN = CIN2 + 3, D = 512;
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int m_begin = gw * tokens_per_wave, m_end = min(M, m_begin + tokens_per_wave);
  if (m_begin >= M) return;
  float w[2][4][CIN], bias[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 4 * (lane + 64 * i) + q;
#pragma unroll
      for (int k = 0; k < CIN; ++k) w[i][q][k] = Wf[(size_t)c * CIN + k];
      bias[i][q] = bf[c];
    }
  float amax = 0.0f;
  for (int m = m_begin; m < m_end; ++m) {
    const int j = m % J;
    const int b = m / (T * J);
    float in[CIN];
#pragma unroll
    for (int k = 0; k < CIN2; ++k) in[k] = x2d[(size_t)m * CIN2 + k];
    const size_t my = y_bcast_T ? ((size_t)b * J + j) : (size_t)m;
#pragma unroll
    for (int k = 0; k < 3; ++k) in[CIN2 + k] = y[my * 3 + k];
    float4 v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = 4 * (lane + 64 * i);
      const float4 sp = *reinterpret_cast<const float4*>(spos + (size_t)j * D + c);
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < CIN; ++k) acc = fmaf(in[k], w[i][q][k], acc);
        o[q] = acc + bias[i][q];
      }
      o[0] += sp.x; o[1] += sp.y; o[2] += sp.z; o[3] += sp.w;
      if (tvec) {
        const float4 tv = *reinterpret_cast<const float4*>(tvec + (size_t)b * tvec_stride + c);
        o[0] += tv.x; o[1] += tv.y; o[2] += tv.z; o[3] += tv.w;
      }
      v[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
    _Float16* yp = XP + (size_t)m * 2 * D;
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = 4 * (lane + 64 * i);
      h4v hi, lo;
      split4_x3(v[i], hi, lo, amax);
      *reinterpret_cast<h4v*>(yp + pair_col(c)) = hi;
      *reinterpret_cast<h4v*>(yp + pair_col(c) + PAIR_LO) = lo;
      sums4(v[i], sm, sq);
    }
    sm = wave_sum(sm);
    sq = wave_sum(sq);
    if (lane == 0) *reinterpret_cast<float2*>(stats + 2 * (size_t)m) = make_float2(sm, sq);
  }
  if (amax > X3_HALF_MAX) range_raise(rw, RANGE_BIT_ACT);
}

bool embed_planes_ok(int D, int in_chans) { return D == 512 && in_chans >= 1 && in_chans <= 5; }

hipError_t launch_embed_planes(const float* x2d, const float* y, const float* Wf, const float* bf, const float* spos,
                               const float* tvec, int64_t tvec_stride, void* XP, float* stats, int B, int T, int J, int D,
                               int in_chans, int y_bcast_T, hipStream_t s) {
  if (!embed_planes_ok(D, in_chans) || !XP || !stats) return hipErrorInvalidValue;
  const int M = B * T * J;
  const int tpw = 16;
  const unsigned grid = (unsigned)((M + tpw * WAVES_PER_BLOCK - 1) / (tpw * WAVES_PER_BLOCK));
#define D3D_EMBEDP(C2)                                                                                                  \
  case C2:                                                                                                              \
    hipLaunchKernelGGL(k_embed_planes<C2>, dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, s, x2d, y, Wf, bf, spos, tvec,    \
                       tvec_stride, (_Float16*)XP, stats, M, T, J, y_bcast_T, tpw, launch_range_word());               \
    break;
  switch (in_chans) {
    D3D_EMBEDP(1) D3D_EMBEDP(2) D3D_EMBEDP(3) D3D_EMBEDP(4) D3D_EMBEDP(5)
  }
#undef D3D_EMBEDP
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ time embedding
// S2S:29-36: [sin(t f_k), cos(t f_k)], k < D/2.  t * f_k is the same fp32 product torch forms.
__global__ void k_sinusoid(const float* __restrict__ times, const float* __restrict__ freqs, float* __restrict__ out,
                           int n, int D) {
  const int half = D >> 1;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * half) return;
  const int i = gid / half, k = gid % half;
  const float arg = __fmul_rn(times[i], freqs[k]);
  out[(size_t)i * D + k] = sinf(arg);
  out[(size_t)i * D + half + k] = cosf(arg);
}

hipError_t launch_sinusoid(const float* times, const float* freqs, float* out, int n, int D, hipStream_t s) {
  const int total = n * (D >> 1);
  hipLaunchKernelGGL(k_sinusoid, dim3((total + 255) / 256), dim3(256), 0, s, times, freqs, out, n, D);
  return hipGetLastError();
}

// out[i, o] = post( sum_k pre(in[i,k]) W[o,k] + b[o] ); one wave per output element, K strided over lanes.
// act: 0 none, 1 GELU on the output (time_mlp.2, S2S:172), 2 SiLU on the input (Block.time_mlp.0, S2S:105)
__global__ __launch_bounds__(256) void k_small_linear(const float* __restrict__ in, const float* __restrict__ W,
                                                      const float* __restrict__ b, float* __restrict__ out, int n, int N,
                                                      int K, int act) {
  const int lane = threadIdx.x & 63;
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (size_t)n * N) return;
  const int i = (int)(w / N), o = (int)(w % N);
  const float* x = in + (size_t)i * K;
  const float* wr = W + (size_t)o * K;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) {
    float xv = x[k];
    if (act == 2) xv = silu(xv);
    acc = fmaf(xv, wr[k], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    acc += b[o];
    if (act == 1) acc = gelu_erf(acc);
    out[w] = acc;
  }
}

hipError_t launch_small_linear(const float* in, const float* W, const float* b, float* out, int n, int N, int K,
                               int act, hipStream_t s) {
  const size_t waves = (size_t)n * N;
  hipLaunchKernelGGL(k_small_linear, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, in, W, b, out, n, N, K, act);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ seq2frame reduce
// S2F:261-263: Conv1d(T -> 1, k = 1) over view(b, f, J*D): out[b, j, :] = sum_t w[t] X[b,t,j,:] + bias
__global__ __launch_bounds__(256) void k_frame_reduce(const float* __restrict__ X, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, int B,
                                                      int T, int JD4) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (size_t)B * JD4) return;
  const int b = (int)(gid / JD4), c = (int)(gid % JD4);
  const float4* xp = reinterpret_cast<const float4*>(X) + (size_t)b * T * JD4 + c;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int t = 0; t < T; ++t) {
    const float wt = w[t];
    const float4 v = xp[(size_t)t * JD4];
    acc.x = fmaf(wt, v.x, acc.x); acc.y = fmaf(wt, v.y, acc.y); acc.z = fmaf(wt, v.z, acc.z); acc.w = fmaf(wt, v.w, acc.w);
  }
  const float bb = bias[0];
  acc.x += bb; acc.y += bb; acc.z += bb; acc.w += bb;
  reinterpret_cast<float4*>(out)[gid] = acc;
}

hipError_t launch_frame_reduce(const float* X, const float* w, const float* bias, float* out, int B, int T, int J, int D,
                               hipStream_t s) {
  if (D & 3) return hipErrorInvalidValue;
  const int JD4 = J * D / 4;
  const size_t total = (size_t)B * JD4;
  hipLaunchKernelGGL(k_frame_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, w, bias, out, B, T, JD4);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ head + DDIM update
// head = LayerNorm(eps 1e-5) + Linear(D -> 3) (S2S:217-220); clamp (DIFF:252,256); DDIM update (DIFF:287-297) with the
// reference's `alpha * x_start` term (DIFF:296) and its fp32 operation order (no fma contraction: __f*_rn).
//
// A workgroup owns HEAD_ROWS = 32 consecutive rows: 32 rows x 3 floats = 384 bytes = three WHOLE 128-byte lines of every
// (rows, 3) output, written by one wave-instruction of lanes 0..95 after the row results have met in LDS.  With one row per
// wave and four rows per workgroup (the first form of this kernel) a line of y_next was shared by three or four
// workgroups on different XCDs, each storing 12-byte pieces of it.  That sharing was once suspected of the head's
// run-to-run deviation on a GPU used by two processes and was EXCLUDED by experiment (experiments/false_share_probe.hip: 0
// wrong words; this whole-line form deviated as well) -- what decides it is the instruction stream of the 3-row dot product
// below.  Whole-line ownership is kept as a performance rule: no two workgroups store into the same 128-byte line.
constexpr int HEAD_ROWS = 32;
template <int NV, bool FENCE>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k_head(HeadArgs a) {
  static_assert(HEAD_ROWS % WAVES_PER_BLOCK == 0 && (HEAD_ROWS * 3 * 4) % 128 == 0, "whole lines per workgroup");
  __shared__ float so[HEAD_ROWS * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * HEAD_ROWS;
  const int D = a.D;
#pragma unroll 1
  for (int r = 0; r < HEAD_ROWS / WAVES_PER_BLOCK; ++r) {
    const int lr = wave * (HEAD_ROWS / WAVES_PER_BLOCK) + r, row = row0 + lr;
    if (row >= a.rows) break;
    float4 v[NV];
    const float* xr = a.X + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
      v[i] = (c < D) ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0, 0, 0, 0);
    }
    ln_row<NV>(v, D, lane, a.g, a.b, a.eps);
    // The three dot products of the row, as a function of the weight pointer so that they can be evaluated more than once (FENCE below).
    // One weight fragment loaded and consumed at a time, and the three sums kept out of packed (v_pk_*) register pairs by an opaque
    // barrier after every update.  This is the instruction stream that never deviated on a GPU shared with a second process (0 of ~900
    // traced samplings); the compiler's free schedule -- three loads in flight behind counted waits, o[0] / o[1] in v_pk_fma_f32 pairs
    // with op_sel -- returned ONE wrong o[0] in about 1 launch of 60 there.  Round 6 identified the mechanism (experiments/NOTES.md
    // section 000, experiments/probes/pk_beside_mfma*.hip): a packed fp32 instruction whose SRC1 low half selects the high dword of its
    // register pair now and then reads 0 for that half in lanes 48-63 while the SIMD's other wave -- here: a wave of the other
    // process's GEMM -- STARTS issuing MFMAs on an idle matrix pipe.  tests/test_abi_host.py checks the built kernel's ISA for the two properties above and, since
    // round 6, EVERY kernel of the library for the absence of that instruction form.
    typedef const float __attribute__((address_space(1))) * gfp;   // (a laundered pointer stays a GLOBAL pointer: global_load, not flat_load)
    auto dot3 = [&](gfp Wh, float (&o)[3], float poke) {
      o[0] = poke; o[1] = 0.f; o[2] = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = 4 * (lane + 64 * i);
        if (c < D) {
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v w = *reinterpret_cast<const f4v __attribute__((address_space(1)))*>(Wh + (size_t)k * D + c);
            o[k] += (v[i].x * w.x + v[i].y * w.y) + (v[i].z * w.z + v[i].w * w.w);
            asm volatile("" : "+v"(o[k]));
            __builtin_amdgcn_sched_barrier(0);   // the next fragment's load stays behind this one's use (every NV)
          }
        }
      }
      // the three wave reductions one after the other, each sum's own adds only: the opaque statements tie the INPUT of the next
      // reduction to the RESULT of the previous one, so the vectoriser cannot pair adds of different sums into v_pk_* here either
      o[0] = wave_sum(o[0]);
      asm volatile("" : "+v"(o[0]), "+v"(o[1]));
      o[1] = wave_sum(o[1]);
      asm volatile("" : "+v"(o[1]), "+v"(o[2]));
      o[2] = wave_sum(o[2]);
      asm volatile("" : "+v"(o[2]));
    };
    // RUN-TIME FENCE ("head_fence" option; round 5's default, since round 6 off: the deviation it guarded against is identified and
    // pinned out of every kernel at build time): the sums are formed twice -- the second time from fragments loaded again through a
    // pointer the compiler cannot see through -- and compared bit for bit.  They have to agree: same instruction stream, same inputs.
    // If they do not, a third evaluation decides by majority and the engine's D3D_RANGE_RECOMPUTE bit is raised.  Cost: +0.08 ms per
    // launch of 0.21.  a.inject (tests only) perturbs the first evaluation of row 0.
    float o[3];
    if constexpr (!FENCE) {
      dot3((gfp)a.Wh, o, 0.0f);
    } else {
      float o2[3];
      dot3((gfp)a.Wh, o, (a.inject && row == 0 && lane == 0) ? 1.0f : 0.0f);
      gfp Wh2 = (gfp)a.Wh;
      asm volatile("" : "+s"(Wh2));
      dot3(Wh2, o2, 0.0f);
      const bool same = ((__float_as_uint(o[0]) ^ __float_as_uint(o2[0])) | (__float_as_uint(o[1]) ^ __float_as_uint(o2[1])) |
                         (__float_as_uint(o[2]) ^ __float_as_uint(o2[2]))) == 0u;
      if (!same) {                                      // wave-uniform: wave_sum hands every lane the same value
        float o3[3];
        gfp Wh3 = (gfp)a.Wh;
        asm volatile("" : "+s"(Wh3));
        dot3(Wh3, o3, 0.0f);
#pragma unroll
        for (int k = 0; k < 3; ++k)
          o[k] = (__float_as_uint(o3[k]) == __float_as_uint(o2[k])) ? o2[k] : ((__float_as_uint(o3[k]) == __float_as_uint(o[k])) ? o[k] : o3[k]);
        if (lane == 0) range_raise(a.range, RANGE_BIT_RECOMPUTE);
      }
    }
    if (lane < 3) so[lr * 3 + lane] = (lane == 0 ? o[0] : (lane == 1 ? o[1] : o[2]));
  }
  __syncthreads();
  const int t = threadIdx.x;                      // lanes 0..95: (local row, k) = (t / 3, t % 3), consecutive 4-byte words
  if (t >= HEAD_ROWS * 3 || row0 + t / 3 >= a.rows) return;
  const int k = t % 3;
  float x0 = so[t] + a.bh[k];
  const size_t idx = (size_t)row0 * 3 + t;
  if (a.x0_raw) a.x0_raw[idx] = x0;
  if (a.mode != 0) {
    if (a.clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    if (a.traj_x0) a.traj_x0[idx * a.traj_x0_stride + a.traj_idx] = x0;
    float yn;
    if (a.mode == 2) {
      yn = x0;  // DIFF:283-285: the last step returns the (clamped) x_start
    } else {
      const float al = a.alpha, an = a.alpha_next;
      // sigma = eta * sqrt((1 - a/an) * (1 - an) / (1 - a));  c = sqrt(1 - an - sigma^2)
      const float sigma = __fmul_rn(a.eta, __fsqrt_rn(__fdiv_rn(__fmul_rn(__fsub_rn(1.0f, __fdiv_rn(al, an)),
                                                                           __fsub_rn(1.0f, an)),
                                                                  __fsub_rn(1.0f, al))));
      const float cc = __fsqrt_rn(__fsub_rn(__fsub_rn(1.0f, an), __fmul_rn(sigma, sigma)));
      const float yc = a.y_cur[idx];
      const float t1 = __fmul_rn(x0, __fsqrt_rn(an));
      const float t4 = __fdiv_rn(__fsub_rn(yc, __fmul_rn(al, x0)), a.somac);
      yn = __fadd_rn(t1, __fmul_rn(cc, t4));
      const float nz = a.noise ? a.noise[idx] : 0.0f;
      yn = __fadd_rn(yn, __fmul_rn(sigma, nz));
    }
    a.y_next[idx] = yn;
    if (a.traj_rev) a.traj_rev[idx * a.traj_rev_stride + a.traj_idx] = yn;
  }
}

hipError_t launch_head(const HeadArgs& a_in, hipStream_t s) {
  HeadArgs a = a_in;
  a.range = launch_range_word();
  if (a.rows <= 0 || (a.D & 3) || a.D > 256 * LN_MAXV) return hipErrorInvalidValue;
  const int grid = (a.rows + HEAD_ROWS - 1) / HEAD_ROWS;
#define D3D_HEAD_LAUNCH(NV)                                                                                              \
  do {                                                                                                                   \
    if (a.fence || a.inject) hipLaunchKernelGGL((k_head<NV, true>), dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, s, a);    \
    else hipLaunchKernelGGL((k_head<NV, false>), dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, s, a);                       \
  } while (0)
  if (a.D <= 256) D3D_HEAD_LAUNCH(1);
  else if (a.D <= 512) D3D_HEAD_LAUNCH(2);
  else D3D_HEAD_LAUNCH(4);
#undef D3D_HEAD_LAUNCH
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ q_sample
// DIFF:360-366 with extract (DIFF:21-24): per-row gather of the two fp32 tables by the integer timestep.
__global__ __launch_bounds__(256) void k_q_sample(const float* __restrict__ x_start, const float* __restrict__ noise,
                                                  const int32_t* __restrict__ t, const float* __restrict__ sqrt_ac,
                                                  const float* __restrict__ somac, float* __restrict__ out, int B,
                                                  int64_t n, int nt, unsigned* __restrict__ rw) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)B * n) return;
  const int tb = t[gid / n];
  if (tb < 0 || tb >= nt) {   // the reference's table[t] raises IndexError (DIFF:21-24); here: no table read, a NaN row, a sticky bit
    out[gid] = __builtin_nanf("");
    if (gid % n == 0) range_raise(rw, RANGE_BIT_INDEX);
    return;
  }
  out[gid] = __fadd_rn(__fmul_rn(sqrt_ac[tb], x_start[gid]), __fmul_rn(somac[tb], noise[gid]));
}

hipError_t launch_q_sample(const float* x_start, const float* noise, const int32_t* t, const float* sqrt_ac,
                           const float* somac, float* out, int B, int64_t n, int nt, hipStream_t s) {
  const int64_t total = (int64_t)B * n;
  hipLaunchKernelGGL(k_q_sample, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x_start, noise, t, sqrt_ac,
                     somac, out, B, n, nt, launch_range_word());
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ p_losses tail (DIFF:411-418)
// out = loss_fn(model_out, target, reduction='none') * loss_coef,  loss_coef[b] = 1 + ac[t_b] / sqrt(1 - ac)[t_b]  [clamped to <= 3]
// -- the host framework's operation order: fp32 division, add, clamp; difference, square (l2) or absolute value (l1); one multiply.
__global__ __launch_bounds__(256) void k_weighted_loss(const float* __restrict__ model_out, const float* __restrict__ target,
                                                       const int32_t* __restrict__ t, const float* __restrict__ ac,
                                                       const float* __restrict__ somac, float* __restrict__ out, int B, int64_t n,
                                                       int l2, int clip, int nt, unsigned* __restrict__ rw) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)B * n) return;
  const int tb = t[gid / n];
  if (tb < 0 || tb >= nt) {   // (as in k_q_sample: the reference raises IndexError on alphas_cumprod[t], DIFF:411)
    out[gid] = __builtin_nanf("");
    if (gid % n == 0) range_raise(rw, RANGE_BIT_INDEX);
    return;
  }
  float coef = __fadd_rn(1.0f, __fdiv_rn(ac[tb], somac[tb]));
  if (clip) coef = fminf(coef, 3.0f);
  const float d = __fadd_rn(model_out[gid], -target[gid]);
  out[gid] = __fmul_rn(l2 ? __fmul_rn(d, d) : fabsf(d), coef);
}

hipError_t launch_weighted_loss(const float* model_out, const float* target, const int32_t* t, const float* ac, const float* somac,
                                float* out, int B, int64_t n, int l2, int clip, int nt, hipStream_t s) {
  const int64_t total = (int64_t)B * n;
  hipLaunchKernelGGL(k_weighted_loss, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, model_out, target, t, ac, somac, out,
                     B, n, l2, clip, nt, launch_range_word());
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ range-guard snapshot
// d3d_engine_range_post: ONE lane exchanges the engine's sticky word with zero and hands the value to the host through a pinned,
// device-mapped slot (bit 31 marks the slot as written); the event recorded behind this launch is what the host waits on or polls.
__global__ void k_range_snapshot(unsigned* __restrict__ word, unsigned* __restrict__ host_slot) {
  const unsigned w = atomicExch(word, 0u);
  __hip_atomic_store(host_slot, w | 0x80000000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_range_snapshot(unsigned* word, unsigned* host_slot_dev, hipStream_t s) {
  hipLaunchKernelGGL(k_range_snapshot, dim3(1), dim3(1), 0, s, word, host_slot_dev);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ repeat_n hypotheses (DIFF:433-448)
// tile: out[r * B + b, :] = x[b, :] (noisy_2d_pose.repeat(repeat_n, 1, 1, 1));  mean: out[b, :] = (sum_r pred[r * B + b, :]) / R in
// hypothesis order (torch.mean(pred.view(repeat_n, b, ...), dim=0)).  n = elements per batch row.
__global__ __launch_bounds__(256) void k_repeat_rows(const float* __restrict__ x, float* __restrict__ out, int64_t rowsn, int64_t total) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid < total) out[gid] = x[gid % rowsn];
}
__global__ __launch_bounds__(256) void k_hypothesis_mean(const float* __restrict__ pred, float* __restrict__ out, int64_t rowsn, int R) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= rowsn) return;
  float acc = pred[gid];
  for (int r = 1; r < R; ++r) acc = __fadd_rn(acc, pred[(int64_t)r * rowsn + gid]);
  out[gid] = __fdiv_rn(acc, (float)R);
}
hipError_t launch_repeat_rows(const float* x, float* out, int B, int64_t n, int R, hipStream_t s) {
  const int64_t rowsn = (int64_t)B * n, total = rowsn * R;
  hipLaunchKernelGGL(k_repeat_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, out, rowsn, total);
  return hipGetLastError();
}
hipError_t launch_hypothesis_mean(const float* pred, float* out, int B, int64_t n, int R, hipStream_t s) {
  const int64_t rowsn = (int64_t)B * n;
  hipLaunchKernelGGL(k_hypothesis_mean, dim3((unsigned)((rowsn + 255) / 256)), dim3(256), 0, s, pred, out, rowsn, R);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ flip-TTA + MPJPE
// RUN:583-590 + LOSS:15-22.  One thread per (b,t,j): merged = (pred + unflip(pred_flip)) / 2 * scale; err = ||.-gt||_2.
// perm[j] = index of the joint whose flipped prediction lands on j (identity when there is no TTA).
__global__ __launch_bounds__(256) void k_tta_mpjpe(const float* __restrict__ pred, const float* __restrict__ pred_flip,
                                                   const float* __restrict__ gt, const uint8_t* __restrict__ mask,
                                                   float scale, JointPerm perm,
                                                   float* __restrict__ merged, double* __restrict__ sums, int BT, int J) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  float err = 0.f, cnt = 0.f;
  if (gid < BT * J) {
    const int bt = gid / J, j = gid % J;
    float p[3];
    for (int k = 0; k < 3; ++k) p[k] = pred[(size_t)gid * 3 + k];
    if (pred_flip) {
      const size_t src = ((size_t)bt * J + perm.p[j]) * 3;
      p[0] = __fdiv_rn(__fadd_rn(p[0], -pred_flip[src + 0]), 2.0f);
      p[1] = __fdiv_rn(__fadd_rn(p[1], pred_flip[src + 1]), 2.0f);
      p[2] = __fdiv_rn(__fadd_rn(p[2], pred_flip[src + 2]), 2.0f);
    }
    for (int k = 0; k < 3; ++k) p[k] = __fmul_rn(p[k], scale);
    if (merged)
      for (int k = 0; k < 3; ++k) merged[(size_t)gid * 3 + k] = p[k];
    if (!mask || mask[bt]) {
      const float dx = p[0] - gt[(size_t)gid * 3 + 0], dy = p[1] - gt[(size_t)gid * 3 + 1],
                  dz = p[2] - gt[(size_t)gid * 3 + 2];
      err = sqrtf(dx * dx + dy * dy + dz * dz);
      cnt = 1.f;
    }
  }
  err = wave_sum(err);
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0 && cnt > 0.f) {
    atomicAdd(&sums[0], (double)err);
    atomicAdd(&sums[1], (double)cnt);
  }
}

hipError_t launch_tta_mpjpe(const float* pred, const float* pred_flip, const float* gt, const uint8_t* mask, float scale,
                            const JointPerm& perm, float* merged, double* sums, int B, int T, int J, hipStream_t s) {
  if (J > JointPerm::MAXJ) return hipErrorInvalidValue;
  const int total = B * T * J;
  hipLaunchKernelGGL(k_tta_mpjpe, dim3((total + 255) / 256), dim3(256), 0, s, pred, pred_flip, gt, mask, scale, perm,
                     merged, sums, B * T, J);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ evaluate()'s other protocols
// RUN:602-614 on the merged, de-normalised prediction: per KEPT frame (mask != 0) -- one thread each, fp64 inside --
//   N-MPJPE (LOSS:83-93): s = <g, p> / <p, p> over the frame's joints; sum_j |s p_j - g_j|
//   P-MPJPE (LOSS:43-81): centred, norm-scaled point sets X0 (target) / Y0 (prediction), H = X0^T Y0 = U S V^T, R = V U^T with the last
//           singular direction flipped when det < 0, a = tr normX / normY: aligned_j - x_j = normX (tr y0_j R - x0_j).  The SVD of the
//           3 x 3 H comes from the Jacobi eigen-decomposition of H^T H (V, S^2), u_i = H v_i / s_i for the two largest values, and
//           R = v1 u1^T + v2 u2^T + (v1 x v2)(u1 x u2)^T -- which IS the reflection-corrected rotation whatever the sign of det H (the
//           third pair of a proper SVD is +-(v1 x v2), +-(u1 x u2) with the product of the signs = sign det H, and the correction
//           multiplies that sign away); tr = s1 + s2 + sign(det H) s3.  No division by the smallest singular value: planar poses are fine.
//   MPJVE  (LOSS:132-142): first difference against the PREVIOUS KEPT frame of the flattened batch (np.diff after the mask), sum_j of
//           |(p_f - p_q) - (g_f - g_q)|; the first kept frame has no predecessor.
// sums[0..2] += the three joint-distance sums, sums[3] += kept frames, sums[4] += frames with a predecessor (caller zeroes sums).
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(128) void k_pose_metrics(const float* __restrict__ pred, const float* __restrict__ gt,
                                                      const uint8_t* __restrict__ mask, double* __restrict__ sums, int N, int J) {
  const int f = blockIdx.x * 128 + threadIdx.x;
  double en = 0.0, ep = 0.0, ev = 0.0, kept = 0.0, pairs = 0.0;
  if (f < N && (!mask || mask[f])) {
    kept = 1.0;
    const float* P = pred + (size_t)f * J * 3;
    const float* G = gt + (size_t)f * J * 3;
    double mp[3] = {0, 0, 0}, mg[3] = {0, 0, 0}, gp = 0.0, pp = 0.0;
    for (int j = 0; j < J; ++j)
      for (int k = 0; k < 3; ++k) {
        const double a = P[3 * j + k], b = G[3 * j + k];
        mp[k] += a; mg[k] += b; gp += a * b; pp += a * a;
      }
    const double sc = gp / pp;
    for (int k = 0; k < 3; ++k) { mp[k] /= J; mg[k] /= J; }
    // centred sets, their norms, H[a][b] = sum_j x0_j[a] y0_j[b]
    double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, nx = 0.0, ny = 0.0;
    for (int j = 0; j < J; ++j) {
      double x[3], y[3], d2 = 0.0;
      for (int k = 0; k < 3; ++k) {
        x[k] = (double)G[3 * j + k] - mg[k];
        y[k] = (double)P[3 * j + k] - mp[k];
        nx += x[k] * x[k]; ny += y[k] * y[k];
        const double d = sc * (double)P[3 * j + k] - (double)G[3 * j + k];
        d2 += d * d;
      }
      en += sqrt(d2);
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) H[a][b] += x[a] * y[b];
    }
    nx = sqrt(nx); ny = sqrt(ny);
    const double inv = 1.0 / (nx * ny);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) H[a][b] *= inv;
    const double detH = H[0][0] * (H[1][1] * H[2][2] - H[1][2] * H[2][1]) - H[0][1] * (H[1][0] * H[2][2] - H[1][2] * H[2][0]) +
                        H[0][2] * (H[1][0] * H[2][1] - H[1][1] * H[2][0]);
    // A = H^T H, cyclic Jacobi: A = V diag(lambda) V^T
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) A[a][b] = H[0][a] * H[0][b] + H[1][a] * H[1][b] + H[2][a] * H[2][b];
    for (int sweep = 0; sweep < 12; ++sweep) {
      const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
      if (off < 1e-300 || off <= 1e-17 * (fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]))) break;
      for (int pq = 0; pq < 3; ++pq) {
        const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
        if (A[p][q] == 0.0) continue;
        const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 3; ++k) {      // columns p, q of A, then rows p, q; columns p, q of V
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq;
        }
      }
    }
    int i1 = 0, i2 = 1, i3 = 2;            // eigenvalues in descending order
    if (A[i1][i1] < A[i2][i2]) { const int t_ = i1; i1 = i2; i2 = t_; }
    if (A[i2][i2] < A[i3][i3]) { const int t_ = i2; i2 = i3; i3 = t_; }
    if (A[i1][i1] < A[i2][i2]) { const int t_ = i1; i1 = i2; i2 = t_; }
    const double s1 = sqrt(fmax(A[i1][i1], 0.0)), s2 = sqrt(fmax(A[i2][i2], 0.0)), s3 = sqrt(fmax(A[i3][i3], 0.0));
    double v1[3] = {V[0][i1], V[1][i1], V[2][i1]}, v2[3] = {V[0][i2], V[1][i2], V[2][i2]}, u1[3], u2[3];
    for (int a = 0; a < 3; ++a) u1[a] = H[a][0] * v1[0] + H[a][1] * v1[1] + H[a][2] * v1[2];
    double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    for (int a = 0; a < 3; ++a) u1[a] /= n1;
    for (int a = 0; a < 3; ++a) u2[a] = H[a][0] * v2[0] + H[a][1] * v2[1] + H[a][2] * v2[2];
    const double dot = u2[0] * u1[0] + u2[1] * u1[1] + u2[2] * u1[2];
    for (int a = 0; a < 3; ++a) u2[a] -= dot * u1[a];
    double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    if (!(n2 > 1e-12 * s1)) {             // rank-one H (collinear points): any unit vector orthogonal to u1 serves (s2 = s3 = 0)
      const int m = fabs(u1[0]) <= fabs(u1[1]) ? (fabs(u1[0]) <= fabs(u1[2]) ? 0 : 2) : (fabs(u1[1]) <= fabs(u1[2]) ? 1 : 2);
      double e[3] = {0, 0, 0};
      e[m] = 1.0;
      const double d_ = e[0] * u1[0] + e[1] * u1[1] + e[2] * u1[2];
      for (int a = 0; a < 3; ++a) u2[a] = e[a] - d_ * u1[a];
      n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    }
    for (int a = 0; a < 3; ++a) u2[a] /= n2;
    const double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
    const double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
    double R[3][3];                         // R[b][a]: prediction space -> target space (row vector y R)
    for (int b = 0; b < 3; ++b)
      for (int a = 0; a < 3; ++a) R[b][a] = v1[b] * u1[a] + v2[b] * u2[a] + v3[b] * u3[a];
    const double tr = s1 + s2 + (detH > 0.0 ? s3 : (detH < 0.0 ? -s3 : 0.0));
    for (int j = 0; j < J; ++j) {
      double y0[3], d2 = 0.0;
      for (int k = 0; k < 3; ++k) y0[k] = ((double)P[3 * j + k] - mp[k]) / ny;
      for (int a = 0; a < 3; ++a) {
        const double x0 = ((double)G[3 * j + a] - mg[a]) / nx;
        const double d = tr * (y0[0] * R[0][a] + y0[1] * R[1][a] + y0[2] * R[2][a]) - x0;
        d2 += d * d;
      }
      ep += nx * sqrt(d2);
    }
    int q = f - 1;
    while (q >= 0 && mask && !mask[q]) --q;
    if (q >= 0) {
      pairs = 1.0;
      const float* Pq = pred + (size_t)q * J * 3;
      const float* Gq = gt + (size_t)q * J * 3;
      for (int j = 0; j < J; ++j) {
        double d2 = 0.0;
        for (int k = 0; k < 3; ++k) {     // np.diff of float32 arrays: the differences are rounded to fp32 there
          const float dp = P[3 * j + k] - Pq[3 * j + k], dg = G[3 * j + k] - Gq[3 * j + k];
          const double d = (double)(dp - dg);
          d2 += d * d;
        }
        ev += sqrt(d2);
      }
    }
  }
  en = wave_sum_d(en); ep = wave_sum_d(ep); ev = wave_sum_d(ev); kept = wave_sum_d(kept); pairs = wave_sum_d(pairs);
  if ((threadIdx.x & 63) == 0 && kept > 0.0) {
    atomicAdd(&sums[0], en); atomicAdd(&sums[1], ep); atomicAdd(&sums[2], ev); atomicAdd(&sums[3], kept); atomicAdd(&sums[4], pairs);
  }
}

hipError_t launch_pose_metrics(const float* pred, const float* gt, const uint8_t* mask, double* sums, int N, int J, hipStream_t s) {
  if (N < 1 || J < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_pose_metrics, dim3((N + 127) / 128), dim3(128), 0, s, pred, gt, mask, sums, N, J);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ eval windows
// ChunkedGenerator(out_all=True, pad=0) for one sequence (GEN:27-48, 247-276): non-overlapping T-frame windows, the last
// one shifted back to end at the last frame, frames it shares with its predecessor masked out; sequences shorter than T
// are edge-padded.  Optionally the horizontally flipped copy (x -> -x on channel 0, left/right joints swapped).
__global__ __launch_bounds__(256) void k_window_gather(const float* __restrict__ seq, float* __restrict__ out,
                                                       uint8_t* __restrict__ mask, JointPerm perm, int n,
                                                       int T, int J, int C, int nc, int flip) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)nc * T * J;
  if (gid >= total) return;
  const int j = (int)(gid % J);
  const int t = (int)((gid / J) % T);
  const int c = (int)(gid / ((long long)J * T));
  int f = (c < nc - 1 ? c * T : n - T) + t;
  f = f < 0 ? 0 : (f > n - 1 ? n - 1 : f);
  const int js = flip ? perm.p[j] : j;
  const float* sp = seq + ((size_t)f * J + js) * C;
  float* op = out + (size_t)gid * C;
  for (int k = 0; k < C; ++k) op[k] = (flip && k == 0) ? -sp[k] : sp[k];
  if (mask && j == 0) {
    const int n_unused = nc * T - n;
    mask[(size_t)c * T + t] = (n >= T && c == nc - 1 && t < n_unused) ? 0 : 1;
  }
}

hipError_t launch_window_gather(const float* seq, float* out, uint8_t* mask, const JointPerm& perm, int n, int T, int J, int C,
                                int flip, hipStream_t s) {
  if (n < 1 || T < 1 || J < 1 || C < 1 || J > JointPerm::MAXJ) return hipErrorInvalidValue;
  const int nc = (n + T - 1) / T;
  const long long total = (long long)nc * T * J;
  hipLaunchKernelGGL(k_window_gather, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, seq, out, mask, perm, n, T, J, C,
                     nc, flip);
  return hipGetLastError();
}

// Seq2frame windows (ChunkedGenerator / ChunkedGenerator_3dhp with out_all=False, chunk_length = stride = 1, GEN:402-420 pair table,
// :492-512 slicing): window w belongs to target frame f = first + w and holds frames f - pad .. f + pad (T = 2 pad + 1), edge-replicated
// at both ends of the sequence (np.pad(..., 'edge')).  Optionally the horizontally flipped copy.
__global__ __launch_bounds__(256) void k_window_gather_s2f(const float* __restrict__ seq, float* __restrict__ out, JointPerm perm,
                                                           int n, int T, int J, int C, int first, int count, int flip) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)count * T * J;
  if (gid >= total) return;
  const int j = (int)(gid % J);
  const int t = (int)((gid / J) % T);
  const int w = (int)(gid / ((long long)J * T));
  int f = first + w - (T - 1) / 2 + t;
  f = f < 0 ? 0 : (f > n - 1 ? n - 1 : f);
  const int js = flip ? perm.p[j] : j;
  const float* sp = seq + ((size_t)f * J + js) * C;
  float* op = out + (size_t)gid * C;
  for (int k = 0; k < C; ++k) op[k] = (flip && k == 0) ? -sp[k] : sp[k];
}

hipError_t launch_window_gather_s2f(const float* seq, float* out, const JointPerm& perm, int n, int T, int J, int C, int first, int count,
                                    int flip, hipStream_t s) {
  if (n < 1 || T < 1 || !(T & 1) || J < 1 || C < 1 || J > JointPerm::MAXJ || first < 0 || count < 1 || first + count > n)
    return hipErrorInvalidValue;
  const long long total = (long long)count * T * J;
  hipLaunchKernelGGL(k_window_gather_s2f, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, seq, out, perm, n, T, J, C, first,
                     count, flip);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ debug trace
// 64-bit position-weighted sum of the 32-bit words of a buffer (d3d_engine_set_trace): commutative, so the grid's
// atomic adds give the same value whatever their order; a changed, moved or swapped word changes it.
__global__ __launch_bounds__(256) void k_checksum(const uint32_t* __restrict__ p, size_t nwords, unsigned long long* out, int rot) {
  // rot: block b does the work of block (b + rot) % gridDim -- with workgroups dealt round-robin over the 8 XCDs, the
  // launches rot = 0..7 read every line once through each XCD's L2 ("views": they must all agree)
  const size_t vb = ((size_t)blockIdx.x + (size_t)rot) % gridDim.x;
  unsigned long long acc = 0;
  for (size_t i = vb * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256)
    acc += (unsigned long long)p[i] * ((i * 0x9E3779B97F4A7C15ull) | 1ull);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

hipError_t launch_checksum(const void* p, size_t bytes, unsigned long long* out, int rot, hipStream_t s) {
  const size_t nwords = bytes / 4;
  if (nwords == 0) return hipSuccess;
  size_t blocks = (nwords + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  blocks = (blocks + 7) / 8 * 8;
  hipLaunchKernelGGL(k_checksum, dim3((unsigned)blocks), dim3(256), 0, s, (const uint32_t*)p, nwords, out, rot);
  return hipGetLastError();
}

}  // namespace d3d
