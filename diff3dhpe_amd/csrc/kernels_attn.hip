// GRAND attention cores of MixSTE (S2S:75-83) for gfx950, exact fp32:   O = (softmax(q k^T / sqrt(dh)) - I) v
// Input is the packed qkv GEMM output (tokens, 3*D) with token m = (b*T + t)*J + j and columns [3][H][dh]; output is
// token-major (tokens, D) so the following proj GEMM needs no transpose -- the reference's rearrange copies
// (S2S:119-121,131-133; 13.8 % of its CPU time) do not exist here, the kernels index the strided groups directly.
//
//   spatial  (one group = the J=17 joints of a frame): 17x17 scores are far too small for a matrix core, so this is a
//            VALU kernel: ONE LANE OWNS ONE QUERY ROW (3 (frame,head) units = 51 rows per wave), K and V rows of the
//            unit are staged in LDS and broadcast-read, the 17-wide softmax is lane-local (no shuffles needed at all).
//   temporal (one group = the T<=256 frames of a joint): v_mfma_f32_32x32x2_f32 on swapped operands -- S^T = K Q^T so a
//            lane holds one query column of scores in registers; the softmax is over registers + one cross-half
//            shuffle, and P^T feeds O^T = V^T P^T straight from the accumulator registers (never touches LDS).
//   generic  any (N, dh): one thread per query row, used for the small test model (dh = 4) and as a cross-check.
#include "d3d_kernels.h"

#include <math.h>

namespace d3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h4a __attribute__((ext_vector_type(4)));
typedef _Float16 h8a __attribute__((ext_vector_type(8)));

// fp32 -> (hi, lo) fp16 pair of 8*x: the operand format of the F16X3 GEMM (kernels_gemm_x3p.hip)
__device__ __forceinline__ void split1_x3(float x, _Float16& hi, _Float16& lo, unsigned* rw) {
  if (fabsf(x) > X3_HALF_MAX * 0.125f) range_raise(rw, RANGE_BIT_ACT);   // (fp32-mode kernels feeding an F16X3 GEMM: not the hot path)
  const float s = __builtin_amdgcn_fmed3f(x * 8.0f, -65504.0f, 65504.0f);
  hi = (_Float16)s;
  lo = (_Float16)(s - (float)hi);
}
__device__ __forceinline__ void store4_x3(_Float16* hp, _Float16* lp, float a, float b, float c, float d, unsigned* rw) {
  const float f[4] = {a, b, c, d};
  h4a hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    _Float16 x, y;
    split1_x3(f[j], x, y, rw);
    hi[j] = x;
    lo[j] = y;
  }
  *reinterpret_cast<h4a*>(hp) = hi;
  *reinterpret_cast<h4a*>(lp) = lo;
}

// =============================================================================================== spatial (VALU)
constexpr int SP_DH = 64;

template <int NJ>
__global__ __launch_bounds__(256) void k_attn_spatial_f32(const float* __restrict__ qkv, float* __restrict__ out,
                                                          _Float16* __restrict__ out_x3,
                                                          int units, int H, int D, unsigned* rw) {
  constexpr int UPW = 64 / NJ;                   // (group, head) units per wave
  constexpr int UNIT_LD = NJ * SP_DH + 4;        // +16 B so the UPW broadcast addresses fall in different banks
  constexpr int ROWS4 = NJ * (SP_DH / 4);        // float4 per unit
  __shared__ __attribute__((aligned(16))) float kv_all[4 * UPW * UNIT_LD];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* kv = kv_all + wave * UPW * UNIT_LD;
  const int base = (blockIdx.x * 4 + wave) * UPW;
  const int slot_raw = lane / NJ, i = lane % NJ;
  const int slot = slot_raw < UPW ? slot_raw : 0;
  const int unit = base + slot;
  const bool valid = (slot_raw < UPW) && (unit < units);
  const int D3 = 3 * D;

  // Staging is split into "issue all global loads" / "write LDS" so the NIT loads of a unit group are in flight together
  // (a load->store loop serialises NIT HBM round trips per phase, which dominated this kernel), and V is fetched into
  // registers while the scores are being computed.
  constexpr int NIT = (UPW * ROWS4 + 63) / 64;
  auto gload = [&](int which, float4 (&tmp)[NIT]) {  // which: 1 = K, 2 = V
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = lane + 64 * it;
      const int s = idx / ROWS4, rem = idx % ROWS4;
      const int j = rem / (SP_DH / 4), c4 = rem % (SP_DH / 4);
      const int u = base + s;
      tmp[it] = make_float4(0, 0, 0, 0);
      if (idx < UPW * ROWS4 && u < units) {
        const int gg = u / H, hh = u % H;
        tmp[it] = *reinterpret_cast<const float4*>(qkv + ((size_t)gg * NJ + j) * D3 + which * D + hh * SP_DH + c4 * 4);
      }
    }
  };
  auto lstore = [&](const float4 (&tmp)[NIT]) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = lane + 64 * it;
      if (idx < UPW * ROWS4) {
        const int s = idx / ROWS4, rem = idx % ROWS4;
        *reinterpret_cast<float4*>(&kv[s * UNIT_LD + rem * 4]) = tmp[it];
      }
    }
  };

  float4 stg[NIT];
  gload(1, stg);
  float q[SP_DH];
  const int g = valid ? unit / H : 0, h = valid ? unit % H : 0;
  {
    const float* qp = qkv + ((size_t)g * NJ + i) * D3 + h * SP_DH;
#pragma unroll
    for (int c = 0; c < SP_DH; c += 4) {
      float4 v = valid ? *reinterpret_cast<const float4*>(qp + c) : make_float4(0, 0, 0, 0);
      q[c] = v.x * 0.125f; q[c + 1] = v.y * 0.125f; q[c + 2] = v.z * 0.125f; q[c + 3] = v.w * 0.125f;  // exact (2^-3)
    }
  }
  lstore(stg);
  __syncthreads();
  gload(2, stg);   // V in flight while the scores are computed from K

  float p[NJ];
  const float* ku = kv + slot * UNIT_LD;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < SP_DH; c += 4) {
      const float4 k4 = *reinterpret_cast<const float4*>(ku + j * SP_DH + c);
      acc = fmaf(q[c], k4.x, acc); acc = fmaf(q[c + 1], k4.y, acc);
      acc = fmaf(q[c + 2], k4.z, acc); acc = fmaf(q[c + 3], k4.w, acc);
    }
    p[j] = acc;
  }
  float m = p[0];
#pragma unroll
  for (int j = 1; j < NJ; ++j) m = fmaxf(m, p[j]);
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) { p[j] = expf(p[j] - m); l += p[j]; }
#pragma unroll
  for (int j = 0; j < NJ; ++j) { p[j] = p[j] / l; if (j == i) p[j] -= 1.0f; }  // attn - I (S2S:82-83)

  __syncthreads();
  lstore(stg);
  __syncthreads();

  float o[SP_DH];
#pragma unroll
  for (int c = 0; c < SP_DH; ++c) o[c] = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int c = 0; c < SP_DH; c += 4) {
      const float4 v4 = *reinterpret_cast<const float4*>(ku + j * SP_DH + c);
      o[c] = fmaf(p[j], v4.x, o[c]); o[c + 1] = fmaf(p[j], v4.y, o[c + 1]);
      o[c + 2] = fmaf(p[j], v4.z, o[c + 2]); o[c + 3] = fmaf(p[j], v4.w, o[c + 3]);
    }
  }
  if (valid) {
    const size_t oo = ((size_t)g * NJ + i) * D + h * SP_DH;
    if (out_x3) {   // pair layout, 16-byte stores: 8 hi at pair_col(c), their 8 lo 32 elements further
#pragma unroll
      for (int c = 0; c < SP_DH; c += 8) {
        h8a hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          _Float16 x, y;
          split1_x3(o[c + e], x, y, rw);
          hi[e] = x;
          lo[e] = y;
        }
        _Float16* op = out_x3 + 2 * oo + pair_col(c);   // oo % 32 == 0, so pair_col(oo + c) = 2 oo + pair_col(c)
        *reinterpret_cast<h8a*>(op) = hi;
        *reinterpret_cast<h8a*>(op + PAIR_LO) = lo;
      }
    } else {
#pragma unroll
      for (int c = 0; c < SP_DH; c += 4) *reinterpret_cast<float4*>(out + oo + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
    }
  }
}

bool attn_spatial_fast_ok(int J, int D, int H) { return J == 17 && H > 0 && D == H * SP_DH; }

hipError_t launch_attn_spatial_f32(const float* qkv, float* out, void* out_x3, int B, int T, int J, int D, int H,
                                   hipStream_t s) {
  if (!attn_spatial_fast_ok(J, D, H)) return hipErrorInvalidValue;
  const long long units = (long long)B * T * H;
  constexpr int UPB = 4 * (64 / 17);
  const long long grid = (units + UPB - 1) / UPB;
  if (units > 0x7fffffffLL || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_attn_spatial_f32<17>, dim3((unsigned)grid), dim3(256), 0, s, qkv, out, (_Float16*)out_x3, (int)units, H, D, launch_range_word());
  return hipGetLastError();
}

// =============================================================================================== temporal (MFMA f32)
constexpr int TP_DH = 64, K_LD = 68, V_LD = 64;

template <int NKT>
__global__ __launch_bounds__(64 * NKT) void k_attn_temporal_f32(const float* __restrict__ qkv, float* __restrict__ out,
                                                                _Float16* __restrict__ out_x3,
                                                                int T, int J, int H, int D, unsigned* rw) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int TP = 32 * NKT;
  float* Ks = lds;                 // [TP][K_LD]  (272-B rows: ds_read_b128 of 16 consecutive keys is conflict-free)
  float* Vs = lds + TP * K_LD;     // [TP][V_LD]

  const int unit = blockIdx.x;     // (b*J + j)*H + h
  const int h = unit % H;
  const int bj = unit / H;
  const int j = bj % J, b = bj / J;
  const int D3 = 3 * D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const size_t tok0 = (size_t)b * T * J + j;   // token(t) = tok0 + t*J

  for (int idx = tid; idx < TP * 16; idx += 64 * NKT) {
    const int row = idx >> 4, c4 = idx & 15;
    float4 kk = make_float4(0, 0, 0, 0), vv = kk;
    if (row < T) {
      const float* p = qkv + (tok0 + (size_t)row * J) * D3 + h * TP_DH + c4 * 4;
      kk = *reinterpret_cast<const float4*>(p + D);
      vv = *reinterpret_cast<const float4*>(p + 2 * D);
    }
    *reinterpret_cast<float4*>(&Ks[row * K_LD + c4 * 4]) = kk;
    *reinterpret_cast<float4*>(&Vs[row * V_LD + c4 * 4]) = vv;
  }

  // this lane's query row, pre-scaled by dh^-0.5 = 2^-3 (exact): lane half hh holds d in [32hh, 32hh+32)
  const int tq = 32 * wave + r;
  float qreg[32];
  {
    const float* qp = qkv + (tok0 + (size_t)(tq < T ? tq : 0) * J) * D3 + h * TP_DH + 32 * hh;
#pragma unroll
    for (int c = 0; c < 32; c += 4) {
      float4 v = (tq < T) ? *reinterpret_cast<const float4*>(qp + c) : make_float4(0, 0, 0, 0);
      qreg[c] = v.x * 0.125f; qreg[c + 1] = v.y * 0.125f; qreg[c + 2] = v.z * 0.125f; qreg[c + 3] = v.w * 0.125f;
    }
  }
  __syncthreads();

  // S^T tile kt: rows = keys kt*32 + (reg&3) + 8*(reg>>2) + 4*hh, column = query tq   (C/D map of the 32x32 MFMA)
  f32x16 sacc[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int q = 0; q < 16; ++q) sacc[kt][q] = 0.f;
    const float* kp = &Ks[(kt * 32 + r) * K_LD + 32 * hh];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 k4 = *reinterpret_cast<const float4*>(kp + 4 * u);
      sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, qreg[4 * u + 0], sacc[kt], 0, 0, 0);
      sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, qreg[4 * u + 1], sacc[kt], 0, 0, 0);
      sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, qreg[4 * u + 2], sacc[kt], 0, 0, 0);
      sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, qreg[4 * u + 3], sacc[kt], 0, 0, 0);
    }
  }

  // exact (max-subtracted, two-pass) softmax over the keys of this query column
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int key = kt * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh;
      if (key >= T) sacc[kt][q] = -INFINITY;
      m = fmaxf(m, sacc[kt][q]);
    }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float e = expf(sacc[kt][q] - m);
      sacc[kt][q] = e;
      l += e;
    }
  l += __shfl_xor(l, 32, 64);
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int key = kt * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh;
      float pv = sacc[kt][q] / l;
      if (key == tq) pv -= 1.0f;   // attn - I (S2S:82-83)
      sacc[kt][q] = pv;
    }

  // O^T[d][query] = sum_key V[key][d] * P^T[key][query]; B operand = the accumulator registers as they stand
  f32x16 oacc[2];
#pragma unroll
  for (int q = 0; q < 16; ++q) { oacc[0][q] = 0.f; oacc[1][q] = 0.f; }
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int key = kt * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh;
      const float v0 = Vs[key * V_LD + r];
      const float v1 = Vs[key * V_LD + 32 + r];
      oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sacc[kt][q], oacc[0], 0, 0, 0);
      oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sacc[kt][q], oacc[1], 0, 0, 0);
    }

  if (tq < T) {
    const size_t oo = (tok0 + (size_t)tq * J) * D + h * TP_DH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const size_t o = oo + dt * 32 + 8 * g4 + 4 * hh;
        if (out_x3)   // pair layout: oo % 32 == 0, the 32-column tile dt is one 128-byte line
          store4_x3(out_x3 + 2 * oo + dt * 64 + 8 * g4 + 4 * hh, out_x3 + 2 * oo + dt * 64 + 8 * g4 + 4 * hh + PAIR_LO,
                    oacc[dt][4 * g4], oacc[dt][4 * g4 + 1], oacc[dt][4 * g4 + 2], oacc[dt][4 * g4 + 3], rw);
        else
          *reinterpret_cast<float4*>(out + o) =
              make_float4(oacc[dt][4 * g4], oacc[dt][4 * g4 + 1], oacc[dt][4 * g4 + 2], oacc[dt][4 * g4 + 3]);
      }
  }
}

bool attn_temporal_fast_ok(int T, int D, int H) { return T >= 1 && T <= 256 && H > 0 && D == H * TP_DH; }

template <int NKT>
static hipError_t launch_temporal_nkt(const float* qkv, float* out, void* out_x3, int B, int T, int J, int D, int H,
                                      hipStream_t s) {
  const size_t lds_bytes = (size_t)32 * NKT * (K_LD + V_LD) * sizeof(float);
  static std::atomic<unsigned long long> attr_set{0};   // one bit per device
  if (hipError_t e = lds_optin(reinterpret_cast<const void*>(&k_attn_temporal_f32<NKT>), lds_bytes, attr_set)) return e;
  const long long grid = (long long)B * J * H;
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_attn_temporal_f32<NKT>, dim3((unsigned)grid), dim3(64 * NKT), lds_bytes, s, qkv, out,
                     (_Float16*)out_x3, T, J, H, D, launch_range_word());
  return hipGetLastError();
}

hipError_t launch_attn_temporal_f32(const float* qkv, float* out, void* out_x3, int B, int T, int J, int D, int H,
                                    hipStream_t s) {
  if (!attn_temporal_fast_ok(T, D, H)) return hipErrorInvalidValue;
  switch ((T + 31) / 32) {
    case 1: return launch_temporal_nkt<1>(qkv, out, out_x3, B, T, J, D, H, s);
    case 2: return launch_temporal_nkt<2>(qkv, out, out_x3, B, T, J, D, H, s);
    case 3: return launch_temporal_nkt<3>(qkv, out, out_x3, B, T, J, D, H, s);
    case 4: return launch_temporal_nkt<4>(qkv, out, out_x3, B, T, J, D, H, s);
    case 5: return launch_temporal_nkt<5>(qkv, out, out_x3, B, T, J, D, H, s);
    case 6: return launch_temporal_nkt<6>(qkv, out, out_x3, B, T, J, D, H, s);
    case 7: return launch_temporal_nkt<7>(qkv, out, out_x3, B, T, J, D, H, s);
    default: return launch_temporal_nkt<8>(qkv, out, out_x3, B, T, J, D, H, s);
  }
}

// =============================================================================================== generic (any N, dh)
template <int DH>
__global__ __launch_bounds__(256) void k_attn_generic(const float* __restrict__ qkv, float* __restrict__ out,
                                                      _Float16* __restrict__ out_x3,
                                                      long long rows, int N, int H, int D, int temporal, int T, int J, unsigned* rw) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= rows) return;
  const int i = (int)(gid % N);
  const long long u = gid / N;
  const int h = (int)(u % H);
  const long long g = u / H;
  const int D3 = 3 * D;
  const float scale = 1.0f / sqrtf((float)DH);
  auto token = [&](int n) -> size_t {
    if (!temporal) return (size_t)g * N + n;
    const long long b = g / J, j = g % J;
    return ((size_t)b * T + n) * J + j;
  };
  float q[DH];
  {
    const float* qp = qkv + token(i) * D3 + h * DH;
#pragma unroll
    for (int c = 0; c < DH; ++c) q[c] = qp[c];
  }
  auto score = [&](int n) {
    const float* kp = qkv + token(n) * D3 + D + h * DH;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) acc = fmaf(q[c], kp[c], acc);
    return acc * scale;
  };
  float m = -INFINITY;
  for (int n = 0; n < N; ++n) m = fmaxf(m, score(n));
  float l = 0.f;
  for (int n = 0; n < N; ++n) l += expf(score(n) - m);
  float o[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) o[c] = 0.f;
  for (int n = 0; n < N; ++n) {
    float p = expf(score(n) - m) / l;
    if (n == i) p -= 1.0f;
    const float* vp = qkv + token(n) * D3 + 2 * D + h * DH;
#pragma unroll
    for (int c = 0; c < DH; ++c) o[c] = fmaf(p, vp[c], o[c]);
  }
  const size_t oo = token(i) * D + h * DH;
#pragma unroll
  for (int c = 0; c < DH; ++c) {
    if (out_x3) {
      _Float16* op = out_x3 + token(i) * 2 * D + pair_col(h * DH + c);
      split1_x3(o[c], op[0], op[PAIR_LO], rw);
    }
    else out[oo + c] = o[c];
  }
}

hipError_t launch_attn_generic(const float* qkv, float* out, void* out_x3, int B, int T, int J, int D, int H,
                               int temporal, hipStream_t s) {
  if (H <= 0 || D % H) return hipErrorInvalidValue;
  const int dh = D / H;
  const int N = temporal ? T : J;
  const long long groups = temporal ? (long long)B * J : (long long)B * T;
  const long long rows = groups * H * N;
  const long long grid = (rows + 255) / 256;
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
#define D3D_GEN(DH)                                                                                                    \
  case DH:                                                                                                             \
    hipLaunchKernelGGL(k_attn_generic<DH>, dim3((unsigned)grid), dim3(256), 0, s, qkv, out, (_Float16*)out_x3, rows, N, \
                       H, D, temporal, T, J, launch_range_word());                                                   \
    break;
  switch (dh) {
    D3D_GEN(4) D3D_GEN(8) D3D_GEN(16) D3D_GEN(32) D3D_GEN(64)
    default: return hipErrorInvalidValue;
  }
#undef D3D_GEN
  return hipGetLastError();
}

}  // namespace d3d
