// The two plain GEMMs of the bf16 operand mode (D3D_PREC_BF16, second-class: DESIGN.md section 4.4) -- qkv = h Wqkv^T + b (q third
// pre-scaled by dh^-1/2) and hidden = gelu(h W1^T + b1), bf16 rows in, bf16 rows out (S2S:67, 46-48 on operands rounded to bf16) -- on the
// hand-specialised two-phase k-loop of the fused F16X3 kernels (qkv_fused_kloop.h) with a 256 x 256 stage of 64-deep bf16 k-tiles: eight
// waves (2 x 4) of 128 rows x 64 columns, the shape and the MFMA order of k_linear_x3q_persist<8,2,4, EPI, bf16-out, FX_BF16>, whose
// epilogue function (x3q_epilogue8) it calls.  Per element the same MFMAs in the same order and the same epilogue arithmetic: bit for bit
// the template's result (tests/test_gpu_round5.py).  What this kernel does not carry: tail slices, run-time group ranges, the one-barrier
// fallback; the ragged last M-tile goes through the checked instantiation of the same epilogue.
#include "d3d_kernels.h"
typedef __bf16 bq_bf8 __attribute__((ext_vector_type(8)));
#define QF_MMA(ACC, BH, BL, AH, AL)                                                                                                   \
  do {                                                                                                                                \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bq_bf8, BH), __builtin_bit_cast(bq_bf8, AH), ACC, 0, 0, 0);      \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bq_bf8, BL), __builtin_bit_cast(bq_bf8, AL), ACC, 0, 0, 0);      \
  } while (0)
#include "qkv_fused_kloop.h"

#include <math.h>
#include <stdio.h>

namespace d3d {
namespace {

#include "gemm_x3p_prelude.h"
#include "gemm_x3p_epilogue.h"

constexpr int BQ_TM = 8, BQ_NJ = 4;
constexpr int BQ_BM = 256, BQ_BN = 256;
constexpr int BQ_AREG = BQ_BM * 128, BQ_STAGE = (BQ_BM + BQ_BN) * 128;   // 65536
constexpr int BQ_AIT = 4, BQ_BIT = 4;                                    // 1-KiB DMA pieces per wave per k-tile
constexpr int BQ_PATCH = BQ_STAGE;                                       // eight 8 KiB transpose patches over stage 1
constexpr int BQ_LDS = 2 * BQ_STAGE;                                     // 131072
static_assert(BQ_PATCH + 65536 <= BQ_LDS, "LDS map");

__device__ __forceinline__ const char* sgpr_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

struct BqArgs {
  const _Float16* Ap;      // bf16 operand rows [>= 256 mtiles rows][K] (passed as 16-bit words; "pair columns" K2 = K / 2 per 4 bytes)
  const _Float16* Wp;      // bf16 weight rows [N padded to 256][K]
  const float* bias;
  _Float16* out;           // bf16 rows [M][N]
  int M, N, Kp, mtiles, ntiles, qcols;   // Kp = K / 2: a row is 4 Kp bytes = Kp / 32 staged 128-byte lines
  unsigned* range;
};

#define BQ_GLDS(SRC, DSTOFF)                                                                                            \
  __builtin_amdgcn_global_load_lds((SRC), (__attribute__((address_space(3))) void*)(uintptr_t)(lds + (DSTOFF)), 16, 0, 0)

__device__ __forceinline__ void wait_vm(int n) {   // s_waitcnt vmcnt(n), n wave-uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
  }
}

template <int EPI>
__global__ __launch_bounds__(512) void k_gemm_bf16q(BqArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  const int ntiles = a.ntiles, tiles = a.mtiles * ntiles;
  if (b >= tiles) return;
  const int nitems = (tiles - b + G - 1) / G;
  const int vfull = (a.mtiles / 8) * 8 * ntiles, mrem = a.mtiles % 8;
  auto tile_of = [&](int o, int& mt, int& nt) {   // all N-tiles of an M-tile on one XCD, the order of k_linear_x3q_persist
    if (o < vfull) {
      const int xcd = o & 7, slot = o >> 3;
      mt = (slot / ntiles) * 8 + xcd;
      nt = slot % ntiles;
    } else {
      const int o2 = o - vfull;
      mt = (a.mtiles / 8) * 8 + o2 % mrem;
      nt = o2 / mrem;
    }
  };
  const int K = a.Kp;
  const size_t K2 = 2 * (size_t)K;              // 16-bit words per operand row
  const int nk = K / 32;
  int mt = 0, nt = 0;
  tile_of(b, mt, nt);
  {   // first k-tile of the first tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lr = lane >> 3, csrc = (lane & 7) ^ (((wave & 1) << 2) | (lr >> 1));
    const unsigned lofs = (unsigned)(lr * (int)K2 + csrc * 8) * 2u;
    const char* ubA = reinterpret_cast<const char*>(a.Ap) + (size_t)(mt * BQ_BM + wave * 8) * K2 * 2;
    const char* ubB = reinterpret_cast<const char*>(a.Wp) + (size_t)(nt * BQ_BN + wave * 8) * K2 * 2;
    const size_t it_stride = (size_t)64 * K2 * 2;
#pragma unroll
    for (int it = 0; it < BQ_AIT; ++it) BQ_GLDS(sgpr_ptr(ubA + it * it_stride) + lofs, wave * 1024 + lane * 16 + it * 8192);
#pragma unroll
    for (int it = 0; it < BQ_BIT; ++it) BQ_GLDS(sgpr_ptr(ubB + it * it_stride) + lofs, BQ_AREG + wave * 1024 + lane * 16 + it * 8192);
  }
  int tid_o = (int)threadIdx.x;
  for (int item = 0; item < nitems; ++item) {
    asm volatile("" : "+v"(tid_o));   // per-lane offsets are re-derived in every tile instead of being hoisted (and spilled)
    const int tid = tid_o;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r16 = lane & 15, q = lane >> 4;
    const bool has_next = item + 1 < nitems;
    int mtn = 0, ntn = 0;
    if (has_next) tile_of((item + 1) * G + b, mtn, ntn);
    const int m0 = mt * BQ_BM, n0 = nt * BQ_BN;

    // ---- DMA plan (kernels_gemm_x3p.hip D3D_DMA_PLAN)
    const int lr_ = lane >> 3;
    const int csrc_ = (lane & 7) ^ (((wave & 1) << 2) | (lr_ >> 1));
    const char* ubA = reinterpret_cast<const char*>(a.Ap) + (size_t)(m0 + wave * 8) * K2 * 2;
    const char* ubB = reinterpret_cast<const char*>(a.Wp) + (size_t)(n0 + wave * 8) * K2 * 2;
    unsigned lofs_ = (unsigned)(lr_ * (int)K2 + csrc_ * 8) * 2u;
    const size_t it_stride = (size_t)64 * K2 * 2;
    const int dstA = wave * 1024 + lane * 16, dstB = BQ_AREG + wave * 1024 + lane * 16;
    const char* ubAn = reinterpret_cast<const char*>(a.Ap) + (size_t)(mtn * BQ_BM + wave * 8) * K2 * 2;
    const char* ubBn = reinterpret_cast<const char*>(a.Wp) + (size_t)(ntn * BQ_BN + wave * 8) * K2 * 2;
    // piece IT (A: 0..3, W: 4..7) of k-tile KTT of this tile, or (KTT == nk) of k-tile 0 of the next one
#define BQ_PIECE(KTT, IT)                                                                                               \
    do {                                                                                                                \
      const bool nxt_ = (KTT) >= nk;                                                                                    \
      const int st_ = ((KTT) & 1) * BQ_STAGE;                                                                           \
      if ((IT) < BQ_AIT) {                                                                                              \
        const char* b_ = nxt_ ? ubAn + (IT) * it_stride : ubA + ((size_t)(KTT) * 128 + (IT) * it_stride);               \
        BQ_GLDS(sgpr_ptr(b_) + lofs_, st_ + dstA + (IT) * 8192);                                                        \
      } else {                                                                                                          \
        const char* b_ = nxt_ ? ubBn + ((IT) - BQ_AIT) * it_stride : ubB + ((size_t)(KTT) * 128 + ((IT) - BQ_AIT) * it_stride); \
        BQ_GLDS(sgpr_ptr(b_) + lofs_, st_ + dstB + ((IT) - BQ_AIT) * 8192);                                             \
      }                                                                                                                 \
    } while (0)

    f32x4 acc[BQ_TM][BQ_NJ];
#pragma unroll
    for (int i = 0; i < BQ_TM; ++i)
#pragma unroll
      for (int j = 0; j < BQ_NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0f;

    const int foff = (q ^ (r16 >> 1)) << 4;
    const int aoff = (wm * 16 * BQ_TM + r16) * 128 + foff, boff = BQ_AREG + (wn * 64 + r16) * 128 + foff;
    h8 bh[BQ_NJ], bl[BQ_NJ], ah[2], al[2];
    int issued_prev = 0;
#define QF_STAGE BQ_STAGE
#define QF_NJ BQ_NJ
#define QF_TM BQ_TM
#define QF_AIT BQ_AIT
#define QF_BIT BQ_BIT
#define QF_PIECE(KTT, IT) BQ_PIECE(KTT, IT)
    QF_KLOOP_HEAD
    QF_KLOOP_TAIL
#undef QF_STAGE
#undef QF_NJ
#undef QF_TM
#undef QF_AIT
#undef QF_BIT
#undef QF_PIECE
#undef BQ_PIECE
    __builtin_amdgcn_s_setprio(0);
    {
      const int mt0 = m0 + wm * 16 * BQ_TM, nt0 = n0 + wn * 64;
      const size_t tbase = (size_t)mt0 * a.N + nt0;
      float* const patch = reinterpret_cast<float*>(lds + BQ_PATCH) + wave * (2 * 16 * 64);
      if (m0 + BQ_BM <= a.M)     // (wave-uniform: a whole tile)
        x3q_epilogue8<BQ_TM, 2, 4, EPI, 3, FX_BF16, false>(acc, patch, lds, a.bias, nullptr, a.out + tbase, nullptr, nullptr, nullptr, nullptr,
                                                            mt0, nt0, wm * 16 * BQ_TM, lane, a.M, a.N, a.qcols, 0, BQ_TM, 1.0f, a.range);
      else                       // the ragged last M-tile: rows >= M are computed from the operand buffer's pad rows and not stored
        x3q_epilogue8<BQ_TM, 2, 4, EPI, 3, FX_BF16, true>(acc, patch, lds, a.bias, nullptr, a.out + tbase, nullptr, nullptr, nullptr, nullptr,
                                                           mt0, nt0, wm * 16 * BQ_TM, lane, a.M, a.N, a.qcols, 0, BQ_TM, 1.0f, a.range);
    }
    mt = mtn; nt = ntn;
    __syncthreads();   // the patches (stage 1) are read before the next tile's second k-tile is staged there
  }
}

}  // namespace

// N a multiple of 256 (whole N-tiles: weight rows padded alike), K a multiple of 128 (an even number >= 2 of 64-deep k-tiles)
bool gemm_bf16q_ok(int N, int K) { return N % 256 == 0 && K % 128 == 0 && K >= 256; }

// Cb[M][N] (bf16) = epi(A[Mp][K] W[N][K]^T + bias), q scaling for columns < qcols; A's buffer spans whole 256-row tiles (pad rows finite or not:
// their results are not stored).  epi: EPI_NONE or EPI_GELU.
hipError_t launch_gemm_bf16q(const void* A, const void* W, const float* bias, void* Cb, int M, int N, int K, int epi, int qcols, hipStream_t s) {
  if (!gemm_bf16q_ok(N, K) || M < 1 || !A || !W || !bias || !Cb || (epi != EPI_NONE && epi != EPI_GELU)) return hipErrorInvalidValue;
  BqArgs a{};
  a.Ap = (const _Float16*)A; a.Wp = (const _Float16*)W; a.bias = bias; a.out = (_Float16*)Cb;
  a.M = M; a.N = N; a.Kp = K / 2; a.mtiles = (M + BQ_BM - 1) / BQ_BM; a.ntiles = N / BQ_BN; a.qcols = qcols;
  a.range = launch_range_word();
  int n_cu = device_cu_count();
  if (n_cu <= 0) return hipErrorUnknown;
  const int tiles = a.mtiles * a.ntiles;
  const int grid = tiles < n_cu ? tiles : n_cu;
  static std::atomic<unsigned long long> attr_done[2] = {{0}, {0}};   // one bit per device
  if (epi == EPI_GELU) {
    if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(k_gemm_bf16q<EPI_GELU>), BQ_LDS, attr_done[1])) return ae;
    hipLaunchKernelGGL(k_gemm_bf16q<EPI_GELU>, dim3(grid), dim3(512), BQ_LDS, s, a);
  } else {
    if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(k_gemm_bf16q<EPI_NONE>), BQ_LDS, attr_done[0])) return ae;
    hipLaunchKernelGGL(k_gemm_bf16q<EPI_NONE>, dim3(grid), dim3(512), BQ_LDS, s, a);
  }
  return hipGetLastError();
}

}  // namespace d3d
