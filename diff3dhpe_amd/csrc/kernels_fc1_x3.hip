// fc1 of the F16X3 block flow as its own kernel: hidden = gelu(norm2(x) W1^T + b1), LayerNorm folded (S2S:46-48 behind S2S:101), on the
// hand-specialised two-phase k-loop of the fused qkv + attention kernels (qkv_fused_kloop.h) with a 256 x 256 x 32 stage -- eight waves
// (2 x 4) of 128 rows x 64 columns, the shape and the MFMA order of k_linear_x3q_persist<8,2,4, EPI_GELU, pair-out, LN-folded>, whose
// epilogue (x3q_epilogue_acc: LayerNorm fold, erfc-series GELU, hi / lo split, accumulator-order pair output, no transpose) it calls:
// per element the same MFMAs in the same order and the same epilogue arithmetic, so the hidden activation is bit for bit the template's.
// What the template's persistent walk carries and this kernel does not: tail slices / ragged-tile instantiations and run-time group
// ranges (every tile is whole: the engine's buffers are padded to 256 rows and the pad rows of the stream are zeroed), the one-barrier
// fallback, the generic epilogue dispatch.  Measured against it: experiments/NOTES.md 0.11.
#include "d3d_kernels.h"
#include "qkv_fused_kloop.h"

#include <math.h>
#include <stdio.h>

namespace d3d {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr float P_A_SCALE = 8.0f;
__device__ __forceinline__ void range_note(unsigned* rw, float amax) {
  if (amax > X3_HALF_MAX) range_raise(rw, RANGE_BIT_ACT);
}
#define D3D_PATCH_FENCE() asm volatile("" ::: "memory")
#include "x3q_epilogue_acc.h"

constexpr int F1_BM = 256, F1_BN = 256, F1_TM = 8, F1_NJ = 4;
constexpr int F1_AREG = F1_BM * 128, F1_STAGE = (F1_BM + F1_BN) * 128;   // 65536
constexpr int F1_AIT = 4, F1_BIT = 4;                                    // 1-KiB DMA pieces per wave per k-tile
constexpr int F1_RAW = 2 * F1_STAGE;                                     // raw statistics partials while the k-loop runs (16 KiB)
constexpr int F1_RAW_MAX = 16384;
constexpr int F1_STX = F1_RAW + F1_RAW_MAX;                              // (rstd', -mean rstd) of the tile's 256 rows, 2 KiB
constexpr int F1_LDS = F1_STX + F1_BM * 8;                               // 149504
static_assert(F1_LDS <= 160 * 1024, "LDS map");

__device__ __forceinline__ const char* sgpr_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

struct F1Args {
  const _Float16* Ap;      // residual stream, pair layout [>= 256 mtiles rows][2 K] of 8 x
  const _Float16* Wp;      // folded fc1 weight W diag(gamma), pair layout, 2^k w, [N padded to 256][2 K]
  const float* bias;       // b + W beta
  const float* csum;       // sum_k W[n, k] gamma[k]
  const float* st_in;      // (sum, sum of squares) partials of the rows: [>= 256 mtiles rows][st_np][2]
  int st_np;
  float eps, out_scale;    // LayerNorm eps; 2^-(3 + k)
  _Float16* out;           // hidden activation, pair layout in accumulator order, [>= 256 mtiles rows][2 N]
  int M, N, K, mtiles, ntiles;
  unsigned* range;
};

#define F1_GLDS(SRC, DSTOFF)                                                                                            \
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


__global__ __launch_bounds__(512) void k_fc1_x3(F1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  const int ntiles = a.ntiles, tiles = a.mtiles * ntiles;
  if (b >= tiles) return;
  const int nitems = (tiles - b + G - 1) / G;
  const int vfull = (a.mtiles / 8) * 8 * ntiles, mrem = a.mtiles % 8;
  // tile ordinal -> (M-tile, N-tile): all N-tiles of an M-tile on one XCD, the order of k_linear_x3q_persist
  auto tile_of = [&](int o, int& mt, int& nt) {
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
  const int K = a.K;
  const size_t K2 = 2 * (size_t)K;
  const int nk = K / 32;
  int mt = 0, nt = 0;
  tile_of(b, mt, nt);
  {   // first k-tile of the first tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lr = lane >> 3, csrc = (lane & 7) ^ (((wave & 1) << 2) | (lr >> 1));
    const unsigned lofs = (unsigned)(lr * (int)K2 + csrc * 8) * 2u;
    const char* ubA = reinterpret_cast<const char*>(a.Ap) + (size_t)(mt * F1_BM + wave * 8) * K2 * 2;
    const char* ubB = reinterpret_cast<const char*>(a.Wp) + (size_t)(nt * F1_BN + wave * 8) * K2 * 2;
    const size_t it_stride = (size_t)64 * K2 * 2;
#pragma unroll
    for (int it = 0; it < F1_AIT; ++it) F1_GLDS(sgpr_ptr(ubA + it * it_stride) + lofs, wave * 1024 + lane * 16 + it * 8192);
#pragma unroll
    for (int it = 0; it < F1_BIT; ++it) F1_GLDS(sgpr_ptr(ubB + it * it_stride) + lofs, F1_AREG + wave * 1024 + lane * 16 + it * 8192);
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
    const int m0 = mt * F1_BM, n0 = nt * F1_BN;

    // ---- row statistics of the folded LayerNorm: raw partials by LDS-DMA under the k-loop, else read at the reduction
    const int st_bytes = F1_BM * a.st_np * 8;
    const bool st_dma = st_bytes <= F1_RAW_MAX;   // (uniform; the block of a 256-row tile is 16-byte aligned)
    int st_issued = 0;
    if (st_dma) {
      const char* src = reinterpret_cast<const char*>(a.st_in + (size_t)m0 * a.st_np * 2);
#pragma unroll
      for (int it = 0; it < F1_RAW_MAX / 1024 / 8; ++it) {
        const int pc = wave + it * 8;
        if (pc * 1024 < st_bytes) {
          F1_GLDS(sgpr_ptr(src + pc * 1024) + lane * 16, F1_RAW + pc * 1024);
          ++st_issued;
        }
      }
    }

    // ---- DMA plan (kernels_gemm_x3p.hip D3D_DMA_PLAN)
    const int lr_ = lane >> 3;
    const int csrc_ = (lane & 7) ^ (((wave & 1) << 2) | (lr_ >> 1));
    const char* ubA = reinterpret_cast<const char*>(a.Ap) + (size_t)(m0 + wave * 8) * K2 * 2;
    const char* ubB = reinterpret_cast<const char*>(a.Wp) + (size_t)(n0 + wave * 8) * K2 * 2;
    unsigned lofs_ = (unsigned)(lr_ * (int)K2 + csrc_ * 8) * 2u;
    const size_t it_stride = (size_t)64 * K2 * 2;
    const int dstA = wave * 1024 + lane * 16, dstB = F1_AREG + wave * 1024 + lane * 16;
    const char* ubAn = reinterpret_cast<const char*>(a.Ap) + (size_t)(mtn * F1_BM + wave * 8) * K2 * 2;
    const char* ubBn = reinterpret_cast<const char*>(a.Wp) + (size_t)(ntn * F1_BN + wave * 8) * K2 * 2;
    // piece IT (A: 0..3, W: 4..7) of k-tile KTT of this tile, or (KTT == nk) of k-tile 0 of the next one
#define F1_PIECE(KTT, IT)                                                                                               \
    do {                                                                                                                \
      const bool nxt_ = (KTT) >= nk;                                                                                    \
      const int st_ = ((KTT) & 1) * F1_STAGE;                                                                           \
      if ((IT) < F1_AIT) {                                                                                              \
        const char* b_ = nxt_ ? ubAn + (IT) * it_stride : ubA + ((size_t)(KTT) * 128 + (IT) * it_stride);               \
        F1_GLDS(sgpr_ptr(b_) + lofs_, st_ + dstA + (IT) * 8192);                                                        \
      } else {                                                                                                          \
        const char* b_ = nxt_ ? ubBn + ((IT) - F1_AIT) * it_stride : ubB + ((size_t)(KTT) * 128 + ((IT) - F1_AIT) * it_stride); \
        F1_GLDS(sgpr_ptr(b_) + lofs_, st_ + dstB + ((IT) - F1_AIT) * 8192);                                             \
      }                                                                                                                 \
    } while (0)

    f32x4 acc[F1_TM][F1_NJ];
#pragma unroll
    for (int i = 0; i < F1_TM; ++i)
#pragma unroll
      for (int j = 0; j < F1_NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0f;

    const int foff = (q ^ (r16 >> 1)) << 4;
    const int aoff = (wm * 128 + r16) * 128 + foff, boff = F1_AREG + (wn * 64 + r16) * 128 + foff;
    h8 bh[F1_NJ], bl[F1_NJ], ah[2], al[2];
    int issued_prev = st_issued;
#define QF_STAGE F1_STAGE
#define QF_NJ F1_NJ
#define QF_TM F1_TM
#define QF_AIT F1_AIT
#define QF_BIT F1_BIT
#define QF_PIECE(KTT, IT) F1_PIECE(KTT, IT)
    QF_KLOOP_HEAD
    // ---- row statistics -> (rstd * out_scale, -mean rstd) per tile row, in the shadow of the last k-tile (x3q_tile's x3_row_stats: the
    // partials added in column order)
    {
      float2* const srow = reinterpret_cast<float2*>(lds + F1_STX);
      if (lane < 32) {
        const int t = wave * 32 + lane, row = m0 + t;
        float sm = 0.f, sq = 0.f;
        if (row < a.M) {
          const float2* raw = st_dma ? reinterpret_cast<const float2*>(lds + F1_RAW) + t * a.st_np
                                     : reinterpret_cast<const float2*>(a.st_in) + (size_t)row * a.st_np;
          for (int p = 0; p < a.st_np; ++p) { sm += raw[p].x; sq += raw[p].y; }
        }
        if (sq >= (X3_HALF_MAX * 0.125f) * (X3_HALF_MAX * 0.125f)) range_raise(a.range, RANGE_BIT_ACT);
        const float mean = sm / (float)K;
        const float var = fmaxf(sq / (float)K - mean * mean, 0.0f);
        if (row < a.M && mean * mean > 256.0f * var) range_raise(a.range, RANGE_BIT_STATS);
        const float rstd = 1.0f / sqrtf(var + a.eps);
        srow[t] = make_float2(rstd * a.out_scale, -mean * rstd);
      }
    }
    QF_KLOOP_TAIL
#undef QF_STAGE
#undef QF_NJ
#undef QF_TM
#undef QF_AIT
#undef QF_BIT
#undef QF_PIECE
#undef F1_PIECE
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();   // statistics visible
    {
      const int mt0 = m0 + wm * 128, nt0 = n0 + wn * 64;
      _Float16* Cht = a.out + 2 * ((size_t)mt0 * a.N + nt0);
      x3q_epilogue_acc<F1_TM, 2, 4, FX_LNF, false>(acc, lds + F1_STX, a.bias, Cht, a.csum, mt0, nt0, wm * 128, lane, a.M, a.N, 0, F1_TM,
                                                    a.out_scale, a.range);
    }
    mt = mtn; nt = ntn;
    __syncthreads();   // the statistics rows are read before the next tile's reduction writes them
  }
}

}  // namespace

bool fc1_x3_ok(int N, int K) { return N % 256 == 0 && K % 64 == 0 && K >= 128; }

// hidden[M, N] = gelu(LN(x) W^T + b), operands and statistics as launch_linear_x3p's LN-folded GELU form; every buffer spans whole
// 256-row tiles and the pad rows of the stream hold finite values (the engine zeroes them).
hipError_t launch_fc1_x3(const void* Apair, const void* Wpair, const float* bias_f, const float* csum, const float* st_in, int st_np,
                         float eps, int w_exp, void* out_pair, int M, int N, int K, hipStream_t s) {
  if (!fc1_x3_ok(N, K) || M <= 0 || st_np < 1 || !Apair || !Wpair || !bias_f || !csum || !st_in || !out_pair) return hipErrorInvalidValue;
  if (w_exp < -14 || w_exp > 12) return hipErrorInvalidValue;
  F1Args a{};
  a.Ap = (const _Float16*)Apair; a.Wp = (const _Float16*)Wpair; a.bias = bias_f; a.csum = csum; a.st_in = st_in; a.st_np = st_np;
  a.eps = eps; a.out_scale = ldexpf(1.0f, -(3 + w_exp)); a.out = (_Float16*)out_pair;
  a.M = M; a.N = N; a.K = K; a.mtiles = (M + F1_BM - 1) / F1_BM; a.ntiles = N / F1_BN;
  a.range = launch_range_word();
  static std::atomic<unsigned long long> attr_done{0};   // one bit per device
  if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(k_fc1_x3), F1_LDS, attr_done)) return ae;
  int n_cu = device_cu_count();
  if (n_cu <= 0) return hipErrorUnknown;
  const int tiles = a.mtiles * a.ntiles;
  const int grid = tiles < n_cu ? tiles : n_cu;
  hipLaunchKernelGGL(k_fc1_x3, dim3(grid), dim3(512), F1_LDS, s, a);
  return hipGetLastError();
}

}  // namespace d3d
