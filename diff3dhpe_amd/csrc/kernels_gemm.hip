// Token GEMMs of the MixSTE blocks (qkv / proj / fc1 / fc2; S2S:67,71,46,48) for gfx950, exact fp32.
//
//   C[M,N] = epi( A[M,K] . W[N,K]^T + bias[N] )            M = B*T*17 tokens (1e5..3e5), N,K in {512,1024,1536}
//
// MFMA: v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate; bit-for-bit a k-ordered fmaf chain, so the result is an
// honest fp32 GEMM).  Its rate is 64 FLOP/clk/SIMD = 157 TFLOP/s chip-wide, 1/16 of the bf16 rate, which makes this
// kernel MFMA-issue bound by a wide margin (a 128x128x32 tile needs 28 FLOP/B from L2; HBM sees each A tile once per
// XCD because the N-tiles of one M-tile are co-scheduled on one XCD).
//
// Tile: 128(M) x 128(N) x 32(K) per 256-thread workgroup; 2x2 waves, each wave 64x64 = 2x2 MFMA tiles (64 acc VGPRs).
// LDS: one A and one B tile, rows padded to 36 floats so the 16-lane groups of ds_read_b128 hit 16 distinct 4-bank
// slots (row stride 144 B = 36 banks: rows r..r+15 start at banks 36r mod 64 = distinct multiples of 4).
// The contraction index is permuted so a lane feeds the MFMA from ONE 16-byte LDS read per four k-steps:
//   MFMA k-slot of lane half h (= lane>>5) at sub-step (u,e) is  kk = 16h + 4u + e   (A and B use the same map).
// Global->LDS staging goes through registers (padded image, so no LDS-DMA) and is software-pipelined: the loads of
// k-tile t+1 are issued before the 64 MFMAs of k-tile t and written to LDS after the barrier that retires tile t.
#include "d3d_kernels.h"

namespace d3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = 36;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Interior tiles take a branch-free epilogue: a per-element bounds branch makes hipcc wait vmcnt(0) (= for the previous
// STORE, stores count in vmcnt on gfx950) before every store.
template <int EPI, bool CHECK>
__device__ __forceinline__ void f32_epilogue(f32x16 (&acc)[2][2], const float* __restrict__ bias, const float* R, float* C, int mw,
                                             int nw, int M, int N) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = nw + j * 32;
    if (CHECK && n >= N) continue;
    const float bn = bias ? bias[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float rv[16];
      if (EPI == EPI_RESIDUAL) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = mw + i * 32 + (q & 3) + 8 * (q >> 2);
          rv[q] = (!CHECK || m < M) ? R[(size_t)m * N + n] : 0.0f;
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = mw + i * 32 + (q & 3) + 8 * (q >> 2);
        if (CHECK && m >= M) continue;
        float v = acc[i][j][q] + bn;
        if (EPI == EPI_GELU) v = gelu_erf(v);
        if (EPI == EPI_RESIDUAL) v = rv[q] + v;
        C[(size_t)m * N + n] = v;
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void k_linear_f32(const float* __restrict__ A, const float* __restrict__ W,
                                                        const float* __restrict__ bias, const float* R, float* C, int M,
                                                        int N, int K, int mtiles, int ntiles) {
  __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_LD];

  // XCD-aware tile map: blocks b and b+8 share an XCD (round-robin dispatch); give one XCD all N-tiles of an M-tile
  // back to back so the A rows are fetched from HBM once and re-read from that XCD's L2.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int mt = (slot / ntiles) * 8 + xcd;
  const int nt = slot % ntiles;
  if (mt >= mtiles) return;
  const int m0 = mt * BM, n0 = nt * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // staging map: 4 float4 per thread per operand; idx = tid + 256 p -> row idx>>3 (0..127), 16-byte chunk idx&7
  float4 ra[4], rb[4];
  const int srow = tid >> 3, sc4 = tid & 7;

  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = srow + 32 * p;
      const int gm = m0 + row, gn = n0 + row;
      ra[p] = (gm < M) ? *reinterpret_cast<const float4*>(A + (size_t)gm * K + k0 + sc4 * 4) : make_float4(0, 0, 0, 0);
      rb[p] = (gn < N) ? *reinterpret_cast<const float4*>(W + (size_t)gn * K + k0 + sc4 * 4) : make_float4(0, 0, 0, 0);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = srow + 32 * p;
      *reinterpret_cast<float4*>(&As[row * LDS_LD + sc4 * 4]) = ra[p];
      *reinterpret_cast<float4*>(&Bs[row * LDS_LD + sc4 * 4]) = rb[p];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

  const int nk = K / BK;
  gload(0);
  const float* a_base = &As[(wm * 64 + r) * LDS_LD + 16 * h];
  const float* b_base = &Bs[(wn * 64 + r) * LDS_LD + 16 * h];

  for (int kt = 0; kt < nk; ++kt) {
    lstore();
    __syncthreads();
    if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 a0 = *reinterpret_cast<const float4*>(a_base + 4 * u);
      const float4 a1 = *reinterpret_cast<const float4*>(a_base + 32 * LDS_LD + 4 * u);
      const float4 b0 = *reinterpret_cast<const float4*>(b_base + 4 * u);
      const float4 b1 = *reinterpret_cast<const float4*>(b_base + 32 * LDS_LD + 4 * u);
      const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
      const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv0[e], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv1[e], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv0[e], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv1[e], acc[1][1], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // epilogue. C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int mw = m0 + wm * 64 + 4 * h, nw = n0 + wn * 64 + r;
  if (m0 + 128 <= M && n0 + 128 <= N) f32_epilogue<EPI, false>(acc, bias, R, C, mw, nw, M, N);
  else f32_epilogue<EPI, true>(acc, bias, R, C, mw, nw, M, N);
}

hipError_t launch_linear_f32(const float* A, const float* W, const float* bias, const float* R, float* C, int M, int N,
                             int K, int epi, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0) return hipErrorInvalidValue;
  if (epi == EPI_RESIDUAL && R == nullptr) return hipErrorInvalidValue;
  const int mtiles = (M + BM - 1) / BM, ntiles = (N + BN - 1) / BN;
  const int grid = ((mtiles + 7) / 8) * 8 * ntiles;
  switch (epi) {
    case EPI_NONE:
      hipLaunchKernelGGL(k_linear_f32<EPI_NONE>, dim3(grid), dim3(256), 0, s, A, W, bias, R, C, M, N, K, mtiles, ntiles);
      break;
    case EPI_GELU:
      hipLaunchKernelGGL(k_linear_f32<EPI_GELU>, dim3(grid), dim3(256), 0, s, A, W, bias, R, C, M, N, K, mtiles, ntiles);
      break;
    case EPI_RESIDUAL:
      hipLaunchKernelGGL(k_linear_f32<EPI_RESIDUAL>, dim3(grid), dim3(256), 0, s, A, W, bias, R, C, M, N, K, mtiles,
                         ntiles);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace d3d
