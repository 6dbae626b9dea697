// F16X3 token GEMM, pre-split operands ("x3q"): the production GEMM of the F16X3 precision mode.
//
//   C[M,N] = epi( A[M,K] . W[N,K]^T + bias[N] )
//
// Both operands arrive pre-split into fp16 hi/lo (see kernels_gemm_f16x3.hip for the arithmetic and its measured
// accuracy) in the PAIR layout of d3d_kernels.h: a row of K values is 2K fp16, and the 32 hi and 32 lo values of
// k-tile t form ONE 128-byte line at [64 t, 64 t + 64):
//   A pair [M][2K]  = split(8 * a)      written by the PRODUCER of the activation (LayerNorm, attention, GELU epilogue)
//   W pair [N][2K]  = split(4096 * w)   made once at weight-commit time
// so the k-loop is a pure fp16 MFMA loop: per output tile and 32-deep k-tile the three products a_lo b_hi, a_hi b_lo,
// a_hi b_hi go into one fp32 accumulator, result scaled by 2^-15 in the epilogue.
//
// Why the pair layout: a k-tile of a row is exactly one cache line.  With separate hi and lo planes each row
// contributes two half-used 128-byte lines per k-tile, and the other halves (the next k-tile) are long evicted from
// the 32 KiB L1 when they are wanted (a k-tile's footprint is 128 KiB of lines), so the L2->L1 fill traffic doubles.
// Measured on MI355X, staging alone (no MFMA) for the qkv GEMM: 0.58 ms with planes, 0.33 ms with whole lines, against
// 0.63 ms of MFMA work to hide it under.
//
// Staging is LDS-DMA (global_load_lds_dwordx4): each wave-instruction drops 1 KiB = 8 rows x 128 B straight into LDS,
// no staging VGPRs.  The LDS image is lane-linear (a tile row is 128 B: chunks 0-3 hi, 4-7 lo), so the bank swizzle
// (16-byte chunk index XOR (row>>1)&7, which makes every ds_read_b128 16-lane group hit 16 distinct 4-bank slots) is
// applied to the per-lane SOURCE address -- it permutes chunks inside one line -- and again on the fragment read.
// Two LDS stages; the DMA of k-tile t+1 is issued after the barrier that retires k-tile t-1 and flies under the MFMAs
// of k-tile t (one barrier per k-tile; __syncthreads() drains the wave's own DMA with vmcnt(0) before the barrier).
//
// Products are issued as v_mfma_f32_16x16x32_f16 (one MFMA spans the 32-deep k-tile): measured on MI355X with random
// fp16 operands (experiments/mfma_ceiling.hip) the chip sustains 1.97 PFLOP/s on this shape against 1.54 PFLOP/s on
// 32x32x16 (it holds a higher clock).  The weight fragment is the first operand, so an accumulator tile holds C^T:
// lane -> token m = lane&15, registers -> n = 4*(lane>>4) + reg; fragment read: lane (r16 = lane&15, q = lane>>4) takes
// 16-byte chunk q (hi) and 4+q (lo) of row r16.
#include "d3d_kernels.h"

#include <algorithm>

namespace d3d {

#include "gemm_x3p_prelude.h"   // f32x4 / h8 / h4, PBK, P_A_SCALE, range_note*, store_split4, D3D_PATCH_FENCE

// Operand tile in LDS: rows of 128 B = 8 chunks of 16 B (0-3 hi, 4-7 lo of the k-tile); physical chunk = c ^ ((row>>1)&7).
// A 16-lane ds_read_b128 group reads 16 consecutive rows at one logical chunk: row parity picks the half of the 256-byte
// bank row, (row>>1)&7 permutes the 8 chunks of that half -> 16 distinct 4-bank slots.  The lo chunk of a fragment is
// the hi chunk's offset XOR 64.

// ---- DMA plan (branch-free): the k-tile of a BM x BN tile is (BM + BN)/8 pieces of 8 rows x
// 128 B; wave w moves pieces w, w + NW, ... of A, then of W.  A lane serves row (8 piece + lane/8), LDS slot lane%8,
// and fetches the source chunk the swizzle assigns to that slot (constant per lane: NW is even, so (row>>1)&7 =
// 4 (w&1) + lane/16).  Contract: the A buffer holds >= mtiles*BM rows and the W buffer >= ntiles*BN rows
// (padding rows are staged and multiplied but never stored).
// Source addresses are formed as (wave-uniform byte base: SGPR pair, advanced by scalar adds) + (one 32-bit per-lane byte
// offset, the same for every piece and k-tile), so that the DMA takes the saddr form and needs no per-piece 64-bit VALU
// address arithmetic.
#define D3D_DMA_PLAN(NW_, BM_)                                                                                          \
  const int lr_ = lane >> 3;                                                                                            \
  const int csrc_ = (lane & 7) ^ (((wave & 1) << 2) | (lr_ >> 1));                                                      \
  const size_t K2_ = 2 * (size_t)K;                                                                                     \
  const char* ubA = reinterpret_cast<const char*>(Ap) + (size_t)(m0 + wave * 8) * K2_ * 2;                             \
  const char* ubB = reinterpret_cast<const char*>(Wp) + (size_t)(n0 + wave * 8) * K2_ * 2;                             \
  unsigned lofs_ = (unsigned)(lr_ * (int)K2_ + csrc_ * 8) * 2u;                                                   \
  const size_t it_stride = (size_t)((NW_) * 8) * K2_ * 2;                 /* bytes */                                   \
  const int dstA = wave * 1024 + lane * 16, dstB = (BM_) * 128 + wave * 1024 + lane * 16

#define D3D_GLDS(SRC, DSTOFF)                                                                                           \
  __builtin_amdgcn_global_load_lds((SRC), (__attribute__((address_space(3))) void*)(uintptr_t)(lds + (DSTOFF)), 16, 0, 0)
// wave-uniform pointer pinned into an SGPR pair
__device__ __forceinline__ const char* sgpr_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}


// Launch-time dispatch over (epilogue, output form): the five combinations the engine and the op hooks use.
#define D3D_X3_DISPATCH(LAUNCH)                                                                                          \
  do {                                                                                                                   \
    if (outsplit == 0) {                                                                                                 \
      if (epi == EPI_NONE) LAUNCH(EPI_NONE, 0);                                                                          \
      else if (epi == EPI_GELU) LAUNCH(EPI_GELU, 0);                                                                     \
      else if (epi == EPI_RESIDUAL) LAUNCH(EPI_RESIDUAL, 0);                                                             \
      else return hipErrorInvalidValue;                                                                                  \
    } else if (outsplit == 1) {                                                                                          \
      if (epi == EPI_NONE) LAUNCH(EPI_NONE, 1);                                                                          \
      else return hipErrorInvalidValue;                                                                                  \
    } else {                                                                                                             \
      if (epi == EPI_GELU) LAUNCH(EPI_GELU, 2);                                                                          \
      else if (epi == EPI_NONE) LAUNCH(EPI_NONE, 2);                                                                     \
      else return hipErrorInvalidValue;                                                                                  \
    }                                                                                                                    \
  } while (0)

#include "gemm_x3p_epilogue.h"

// The products of one (m-tile, n-tile) pair for one staged 128-byte line of each operand row.  F16X3: the line holds the 32 hi
// and the 32 lo halves of a 32-deep k-tile -- a_lo b_hi + a_hi b_lo + a_hi b_hi, smallest terms first.  bf16 mode: the line holds
// 64 bf16 k values -- the same two 16-byte fragment reads per operand row are k 0..31 and k 32..63, one MFMA each.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int FX>
__device__ __forceinline__ void x3_mma(f32x4& acc, const h8& bh, const h8& bl, const h8& ah, const h8& al) {
  if constexpr ((FX & FX_BF16) != 0) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, bh), __builtin_bit_cast(bf8, ah), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, bl), __builtin_bit_cast(bf8, al), acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc, 0, 0, 0);
  }
}

// Tile shapes: BM = 16*TM*WM rows, BN = 64*WN columns, WM x WN waves, each wave (16 TM) x 64 = TM x 4 MFMA tiles.
//   <8,2,4> 256x256, 8 waves of 128x64, 128 KiB LDS  -- large problems
//   <4,4,2> 256x128, 8 waves of  64x64,  96 KiB LDS  -- problems too small to fill the chip with 256x256 tiles
// Both put ONE 8-wave workgroup on a CU.  Every shape adds the same MFMA results in the same order into an output element,
// so an element's value does not depend on the tile shape that produced it (results are batch-size independent, bitwise).
// (Shapes with 4-wave workgroups or two workgroups per CU measured 8-25 % slower and are not instantiated: experiments/NOTES.md.)
// One output tile (device function: the launch wrapper below maps blockIdx to tiles).
// PERSIST (k_linear_x3q_persist): the workgroup walks several tiles.  Then (i) the first k-tile of a tile has already been
// staged (by the caller for the first tile, by the previous tile otherwise), (ii) the LAST k-tile of this tile -- which
// reads stage 1 when K/32 is even -- stages the first k-tile of the NEXT tile (m0n, n0n) into stage 0, so that it lands
// under the last MFMAs and the epilogue, (iii) the epilogue's transpose patches live in stage 1.
// FULL: the caller guarantees a whole tile inside the matrix (m0 + BM <= M, n0 + BN <= N): only the unchecked epilogue is
// instantiated -- the persistent walk sends ragged tiles through the SUB instantiation, so that its whole-tile path carries ONE
// branch-free epilogue (the row-statistics form used to run the checked copy -- an exec-mask branch around every residual load
// and store -- for every tile, because two copies under a run-time branch spilled accumulators).
// NST (one tile per workgroup only): operand stages of the one-barrier loop.  2: k-tile t + 1 is requested while k-tile t is multiplied
// (a k-tile then lasts at least one DMA round trip -- what bounds the launches whose every tile has a CU of its own: B = 1); 3 / 4:
// k-tiles t + 1 .. t + NST - 1 are in flight, the wait in front of a k-tile's barrier is a counted vmcnt that leaves the younger ones out.
template <int TM, int WM, int WN, int EPI, int OUTSPLIT, int FX, bool PERSIST = false, bool SUB = false, bool FULL = false, int NST = 2>
__device__ __forceinline__ void x3q_tile(const _Float16* __restrict__ Ap, const _Float16* __restrict__ Wp,
                                         const float* __restrict__ bias, const float* R, float* C, _Float16* Ch, _Float16* Cl,
                                         int M, int N, int K, int m0, int n0, int nt, int ntiles, int qcols,
                                         unsigned long long* diag, const X3Tail& fx, bool has_next = false, int m0n = 0,
                                         int n0n = 0, int tid_in = -1, int sub_wm = -1, int g_lo = 0, int g_hi = TM) {
  // SUB -- split tail tile (k_linear_x3q_persist): only the m-tiles [g_lo, g_hi) (g_lo even) of the waves in wave-row sub_wm (-1:
  // every wave-row) are computed and stored; the other waves still stage W pieces and meet the barriers.  A pieces outside
  // the computed rows are not staged (after the first k-tile, which the previous tile staged in full).
  // PERSIST passes the thread index behind an opaque barrier so that the per-lane offsets are re-derived in every tile
  // instead of being hoisted out of the tile loop and held in (spilled) registers
  const int tidx = PERSIST ? tid_in : (int)threadIdx.x;
  constexpr int NW = WM * WN, BM = 16 * TM * WM, BN = 64 * WN;
  constexpr int A_REG = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int A_IT = BM / 8 / NW, B_IT = BN / 8 / NW, N_IT = A_IT + B_IT;   // 1-KiB DMA pieces per wave per k-tile
  constexpr int PPG = (N_IT + TM - 1) / TM;                                   // pieces issued per MFMA group
  static_assert(NW % 2 == 0 && (BM / 8) % NW == 0 && (BN / 8) % NW == 0, "pieces must split evenly over the waves");
  static_assert(NST * STAGE >= NW * 2 * 16 * 64 * 4, "epilogue patches must fit in the operand stages");
  static_assert(NST == 2 || (!PERSIST && !SUB && NST >= 2 && NST <= 4), "deeper staging: one tile per workgroup");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int bid = blockIdx.x;
  // diag (diagnostic launches only, experiments/gemm_bench.py): shader-clock and 100 MHz stamps around the k-loop and the
  // epilogue of every workgroup, into a buffer nothing else reads
  unsigned long long st_c0 = 0, st_r0 = 0;
  if (diag) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
  // beyond the operand stages and the persistent walk's epilogue patches, which start at stage 1 and take 64 KiB (allocated only for
  // the folded forms)
  constexpr int LDS_X = (PERSIST && STAGE + NW * 2 * 16 * 64 * 4 > 2 * STAGE) ? STAGE + NW * 2 * 16 * 64 * 4 : NST * STAGE;
  unsigned char* const lds_x = lds + LDS_X;
  const int tid = tidx;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r16 = lane & 15, q = lane >> 4;
  // Row statistics of the LayerNorm folded into this GEMM: (rstd, -mean rstd) per tile row in lds_x, from the producer's (sum, sum
  // of squares) partials.  Persistent walk: the raw partials of the tile's rows -- one contiguous block -- are requested by LDS-DMA
  // here, land under the k-loop and are reduced from LDS in front of the epilogue (x3_row_stats below); loaded into registers at
  // this point they cost every tile an exposed L2 round trip before its first k-tile (measured upper bound: qkv -2.0 %, fc1 -1.7 %
  // per launch with the loads removed).  One tile per workgroup: loaded and reduced here, visible after the first k-tile barrier.
  constexpr bool ST_DMA_FORM = PERSIST && (FX & FX_LNF) != 0;
  constexpr int ST_RAW_MAX = 16384;                                // bytes of lds_raw (BM rows x up to 8 partials x 8 B at BM = 256)
  unsigned char* const lds_raw = lds_x + BM * 8;
  const int st_bytes = (FX & FX_LNF) ? BM * fx.st_np * 8 : 0;
  const bool st_dma = ST_DMA_FORM && st_bytes <= ST_RAW_MAX;       // (uniform)
  int st_issued = 0;                                               // DMA pieces of the statistics this wave has in flight
  auto x3_row_stats = [&](auto&& partial) {                        // partial(row_in_tile, p) -> float2; same order of additions either way
    if (tidx < BM) {
      const int row = m0 + tidx;
      float sm = 0.f, sq = 0.f;
      if (row < M)
        for (int p = 0; p < fx.st_np; ++p) {
          const float2 t = partial(tidx, p);
          sm += t.x; sq += t.y;
        }
      // range guard for the producer of these rows (the proj / fc2 epilogues write the planes of x and these statistics of the
      // values themselves): an element |x| > 8188 implies sum x^2 > 8188^2 -- never missed; rows of ~512 values above ~360
      // would raise it falsely, a LayerNorm-ed stream is orders of magnitude below
      if (sq >= (X3_HALF_MAX * 0.125f) * (X3_HALF_MAX * 0.125f)) range_note(fx.range, 2.0f * X3_HALF_MAX);
      const float mean = sm / (float)K;
      const float var = fmaxf(sq / (float)K - mean * mean, 0.0f);
      if (row < M) range_note_stats(fx.range, mean, var);
      const float rstd = 1.0f / sqrtf(var + fx.eps);
      // (rstd carries the GEMM's power-of-two output scale: the epilogues form rstd' acc - rstd mean csum + b' in two fmas)
      reinterpret_cast<float2*>(lds_x)[tidx] = make_float2(rstd * fx.out_scale, -mean * rstd);
    }
  };
  if constexpr ((FX & FX_LNF) != 0) {
    if (st_dma) {   // pieces of 1 KiB: wave w takes pieces w, w + NW (the st_in buffer is padded to whole tiles of rows)
      const char* src = reinterpret_cast<const char*>(fx.st_in + (size_t)m0 * fx.st_np * 2);
#pragma unroll
      for (int it = 0; it < ST_RAW_MAX / 1024 / NW; ++it) {
        const int pc = wave + it * NW;
        if (pc * 1024 < st_bytes) {
          __builtin_amdgcn_global_load_lds(sgpr_ptr(src + pc * 1024) + lane * 16,
                                           (__attribute__((address_space(3))) void*)(uintptr_t)(lds_raw + pc * 1024), 16, 0, 0);
          ++st_issued;
        }
      }
    } else {
      x3_row_stats([&](int r, int p) { return *reinterpret_cast<const float2*>(fx.st_in + 2 * ((size_t)(m0 + r) * fx.st_np + p)); });
    }
  }
  // (a compile-time switch: with run-time ranges in the whole-tile path too, the whole GEMM ran 3.5 % slower)
  const bool w_act = !SUB || sub_wm < 0 || wm == sub_wm;          // wave-uniform
  const int gl = SUB ? (w_act ? g_lo : 0) : 0, gh = SUB ? (w_act ? g_hi : 0) : TM;
  unsigned amask = SUB ? 0u : ~0u;                                // A pieces of this wave that carry computed rows
  if (SUB) {
    const int a_lo = (sub_wm < 0 ? 0 : sub_wm) * 16 * TM + g_lo * 16, a_hi = (sub_wm < 0 ? WM - 1 : sub_wm) * 16 * TM + g_hi * 16;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int r0 = (it * NW + wave) * 8;
      if (r0 + 8 > a_lo && r0 < a_hi) amask |= 1u << it;
    }
  }

  D3D_DMA_PLAN(NW, BM);
#define D3D_QSTAGE_ONE(ST, KT, IT)                                                                                      \
  do {                                                                                                                  \
    if ((IT) < A_IT) {                                                                                                  \
      if ((amask >> (IT)) & 1u)                                                                                         \
        D3D_GLDS(sgpr_ptr(ubA + ((size_t)(KT) * 128 + (IT) * it_stride)) + lofs_, (ST) * STAGE + dstA + (IT) * NW * 1024); \
    } else                                                                                                              \
      D3D_GLDS(sgpr_ptr(ubB + ((size_t)(KT) * 128 + ((IT) - A_IT) * it_stride)) + lofs_,                               \
               (ST) * STAGE + dstB + ((IT) - A_IT) * NW * 1024);                                                        \
  } while (0)

  f32x4 acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0f;

  // fragment offsets: rows r16 + 16 i all share the swizzle key r16>>1
  const int foff = (q ^ (r16 >> 1)) << 4;
  const int aoff = (wm * 16 * TM + r16) * 128 + foff, boff = A_REG + (wn * 64 + r16) * 128 + foff;
  const int nk = K / PBK;
  if (!PERSIST) {
#pragma unroll
    for (int kt0 = 0; kt0 < NST - 1; ++kt0)
      if (kt0 == 0 || kt0 < nk) {
#pragma unroll
        for (int it = 0; it < N_IT; ++it) D3D_QSTAGE_ONE(kt0, kt0, it);
      }
  }
  // next tile's operand bases (PERSIST): same per-lane offset, other uniform bases
  const char* ubAn = reinterpret_cast<const char*>(Ap) + (size_t)(m0n + wave * 8) * K2_ * 2;
  const char* ubBn = reinterpret_cast<const char*>(Wp) + (size_t)(n0n + wave * 8) * K2_ * 2;
#define D3D_QSTAGE_NEXT(IT)                                                                                             \
  do {                                                                                                                  \
    if ((IT) < A_IT) D3D_GLDS(sgpr_ptr(ubAn + (IT) * it_stride) + lofs_, dstA + (IT) * NW * 1024);                       \
    else D3D_GLDS(sgpr_ptr(ubBn + ((IT) - A_IT) * it_stride) + lofs_, dstB + ((IT) - A_IT) * NW * 1024);                 \
  } while (0)

  [[maybe_unused]] auto qk_wait_vm = [](int n) {   // s_waitcnt vmcnt(n), n wave-uniform
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  // one k-tile = TM groups (one 16-row m-tile each): the A fragments of group g+1 are read, and PPG DMA pieces of the
  // next k-tile are issued, before the 12 MFMAs of group g; the 8 W fragments are read once at the top of the k-tile.
#define D3D_QKTILE(KT, PREFETCH, WAITN)                                                                                  \
  do {                                                                                                                   \
    /* this wave's DMA pieces of k-tile KT have landed before it meets the barrier: written out, not left to the fence */\
    /* of __syncthreads() -- in the SUB instantiation (uniform branches around the DMA issues) the compiler emitted no  */\
    /* vmcnt wait in the k-loop at all, and the slices read stale W rows (the last pieces issued)                        */\
    if constexpr (NST == 2) {                                                                                            \
      __builtin_amdgcn_s_waitcnt(0x0F70);   /* vmcnt(0) */                                                               \
      __syncthreads();                                                                                                   \
    } else {   /* the pieces of k-tiles KT + 1 .. stay in flight (vmcnt retires in order); LDS reads of k-tile KT - 1 are done */ \
      qk_wait_vm(WAITN);                                                                                                 \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* (row statistics written to LDS in front of the loop) */    \
      __builtin_amdgcn_s_barrier();                                                                                      \
    }                                                                                                                    \
    asm volatile("" : "+v"(lofs_)); /* keeps the lane offset out of the loop's pointer induction (saddr form) */         \
    const int nst = NST == 2 ? (((KT) + 1) & 1) : ((KT) + NST - 1) % NST;                                                \
    const unsigned char* sb = lds + (NST == 2 ? ((KT) & 1) : (KT) % NST) * STAGE;                                        \
    h8 bh[4], bl[4], ah[2], al[2];                                                                                       \
    ah[0] = *reinterpret_cast<const h8*>(sb + aoff + gl * 2048);                                                         \
    al[0] = *reinterpret_cast<const h8*>(sb + ((aoff + gl * 2048) ^ 64));                                                \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
      bh[j] = *reinterpret_cast<const h8*>(sb + boff + j * 2048);                                                        \
      bl[j] = *reinterpret_cast<const h8*>(sb + ((boff + j * 2048) ^ 64));                                               \
    }                                                                                                                    \
    _Pragma("unroll") for (int g = 0; g < TM; ++g) {                                                                     \
      const bool g_act = g >= gl && g < gh;                                                                              \
      if (g_act && g + 1 < gh) {                                                                                         \
        ah[(g + 1) & 1] = *reinterpret_cast<const h8*>(sb + aoff + (g + 1) * 2048);                                      \
        al[(g + 1) & 1] = *reinterpret_cast<const h8*>(sb + ((aoff + (g + 1) * 2048) ^ 64));                             \
      }                                                                                                                  \
      if (PREFETCH) {                                                                                                    \
        _Pragma("unroll") for (int pp = 0; pp < PPG; ++pp)                                                               \
          if (g * PPG + pp < N_IT) D3D_QSTAGE_ONE(nst, (KT) + NST - 1, g * PPG + pp);                                    \
      } else if (PERSIST) {                                                                                              \
        if (has_next) {                                                                                                  \
          _Pragma("unroll") for (int pp = 0; pp < PPG; ++pp)                                                             \
            if (g * PPG + pp < N_IT) D3D_QSTAGE_NEXT(g * PPG + pp);                                                      \
        }                                                                                                                \
      }                                                                                                                  \
      if (g_act) {                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                  \
          x3_mma<FX>(acc[g][j], bh[j], bl[j], ah[g & 1], al[g & 1]);                                                     \
        }                                                                                                                \
      }                                                                                                                  \
      if constexpr ((FX & FX_BF16) != 0) {   /* bf16: 8 MFMAs per group -- pairs where the F16X3 pattern has triples */         \
        if (!SUB && g == 0) {                                                                                            \
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                                                             \
        } else if (!SUB) {                                                                                               \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                             \
          __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);                                                             \
        }                                                                                                                \
      } else                                                                                                             \
      if (!SUB && g == 0) {   /* the k-tile opening group: A pair + first W pair, then a W pair ahead of each MFMA triple */    \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                                                               \
      } else if (!SUB) {   /* 2 MFMAs, a read, 2 MFMAs, a read, 2 MFMAs, a piece, 2 MFMAs, the other pieces, 4 MFMAs */       \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                               \
      }                                                                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                                 \
    }                                                                                                                    \
  } while (0)

  if constexpr (PERSIST && TM % 2 == 0 && WM == 2) {
    // ---- Two phases per k-tile, staging two phases ahead (persistent walk, 256 x 256 tiles).  A k-tile is split where its
    // buffers die: the W fragments go to registers at the top of the first phase, the A rows of m-tiles 0..TM/2-1 are read in the
    // first phase, those of TM/2..TM-1 in the second.  So the pieces of a LATER k-tile can be issued into a stage while its other
    // half is still read:
    //   phase 2t   (m-tiles 0..TM/2-1 of k-tile t) issues A(t+1)  [its stage last held A(t-1), read through phase 2t-1]
    //   phase 2t+1 (m-tiles TM/2..TM-1)           issues W(t+2)  [its stage half held W(t), in registers since phase 2t]
    // and every piece has at least one whole phase to land: the wait before the barrier of a phase is a COUNTED vmcnt that leaves
    // the pieces of the phase just finished in flight (the one-barrier form waits vmcnt(0) for pieces issued a sixth of a k-tile
    // earlier).  The stream runs on across tiles: W(0) / A(0) of the next tile are issued by phases 2nk-3 / 2nk-2 of this one, its
    // W(1) -- whose stage holds the epilogue's patches -- with A(1) in its own phase 0.  Same MFMAs in the same order per element
    // as the one-barrier form (values unchanged).  The whole-row form (WM == 1: 2 + 8 pieces per wave and k-tile) and the
    // non-persistent launches keep the one-barrier loop (measured: experiments/NOTES.md).
    auto wait_vm = [](int n) {   // s_waitcnt vmcnt(n), n wave-uniform (counts differ per wave only in tail slices)
      switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      }
    };
    constexpr bool XPF = !SUB;   // fragment reads that cross a phase barrier (whole tiles only: slices have run-time group ranges)
    // W fragments of the next k-tile read a phase early (below): qkv -1.3 % per launch; proj 0; fc1 +0.6 % (its GELU form sits at
    // 256 VGPRs and spills four more dwords) -- so only the form without an epilogue operation
    constexpr bool WPF = XPF && !(FX & FX_BF16) && EPI == EPI_NONE;
    const int nA = SUB ? __builtin_popcount(amask & ((1u << A_IT) - 1u)) : A_IT;   // A pieces this wave issues per k-tile
    int issued_prev = 0;                                                            // pieces this wave issued in the previous phase
    // piece `it` (A: 0..A_IT-1, W: A_IT..N_IT-1) of k-tile KTT of this tile (KTT < nk) or of k-tile 0 of the next one (KTT == nk)
#define D3D_PIECE(KTT, IT)                                                                                               \
    do {                                                                                                                 \
      const bool nxt_ = (KTT) >= nk;                                                                                     \
      const int st_ = ((KTT) & 1) * STAGE;                                                                               \
      if ((IT) < A_IT) {                                                                                                 \
        if ((amask >> (IT)) & 1u) {                                                                                      \
          const char* b_ = nxt_ ? ubAn + (IT) * it_stride : ubA + ((size_t)(KTT) * 128 + (IT) * it_stride);              \
          D3D_GLDS(sgpr_ptr(b_) + lofs_, st_ + dstA + (IT) * NW * 1024);                                                 \
        }                                                                                                                \
      } else {                                                                                                           \
        const char* b_ = nxt_ ? ubBn + ((IT) - A_IT) * it_stride : ubB + ((size_t)(KTT) * 128 + ((IT) - A_IT) * it_stride); \
        D3D_GLDS(sgpr_ptr(b_) + lofs_, st_ + dstB + ((IT) - A_IT) * NW * 1024);                                          \
      }                                                                                                                  \
    } while (0)
    // one phase: H = 0 / 1.  Even phase (H = 0) of k-tile KT: A(KT+1) if DO_A (and all of W(1) in a tile's first phase, W_FULL1);
    // odd phase: W(KT+2) if DO_W.  Instruction order inside an m-tile group (12 MFMAs, the next group's two A-fragment reads, 1-4
    // staging pieces) stated with sched_group_barrier: 2 MFMAs, a read, 2 MFMAs, a read, 2 MFMAs, a piece, 2 MFMAs, the other
    // pieces, 4 MFMAs; in the k-tile's opening group the A pair and first W pair, then a W pair ahead of each MFMA triple.  Left
    // alone the scheduler puts the reads and the pieces at the top of the group and the 12 MFMAs behind them (+2.4 % on qkv).
#define D3D_PHASE(KT, H, DO_A, W_FULL1, DO_W, W_AHEAD)                                                                    \
    do {                                                                                                                  \
      wait_vm(issued_prev);                                                                                               \
      __builtin_amdgcn_s_barrier();                                                                                       \
      if (!SUB) __builtin_amdgcn_s_setprio(3);               /* (see the end of the m-tile group) */                      \
      asm volatile("" : "+v"(lofs_) : : "memory");                                                                        \
      const unsigned char* sb = lds + ((KT) & 1) * STAGE;                                                                 \
      constexpr int G0 = (H) * (TM / 2), G1 = G0 + TM / 2;                                                                \
      const int gf = gl > G0 ? gl : G0;                     /* first active m-tile of this phase */                       \
      /* (whole tiles: the odd phase's first A pair was requested by the last group of the even phase -- those rows landed   \
         before the even phase's barrier --, so the odd phase opens on its MFMAs) */                                       \
      if (gf < gh && gf < G1 && !(XPF && (H) == 1)) {                                                                     \
        ah[gf & 1] = *reinterpret_cast<const h8*>(sb + aoff + gf * 2048);                                                 \
        al[gf & 1] = *reinterpret_cast<const h8*>(sb + ((aoff + gf * 2048) ^ 64));                                        \
      }                                                                                                                   \
      /* W_AHEAD (WPF): the W fragments of k-tile KT+1 -- in LDS since the barrier of this odd phase -- \
         replace those of KT pair by pair behind the last group's MFMA triples, so the next even phase opens on its A pair   \
         alone; an even phase reads W itself only in a tile's first k-tile */                                               \
      if ((H) == 0 && (!WPF || (W_FULL1))) {                                                                              \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                   \
          bh[j] = *reinterpret_cast<const h8*>(sb + boff + j * 2048);                                                     \
          bl[j] = *reinterpret_cast<const h8*>(sb + ((boff + j * 2048) ^ 64));                                            \
        }                                                                                                                 \
      }                                                                                                                   \
      _Pragma("unroll") for (int g = G0; g < G1; ++g) {                                                                   \
        const bool g_act = g >= gl && g < gh;                                                                             \
        if (g_act && g + 1 < gh && g + 1 < ((XPF && (H) == 0) ? TM : G1)) {                                               \
          ah[(g + 1) & 1] = *reinterpret_cast<const h8*>(sb + aoff + (g + 1) * 2048);                                     \
          al[(g + 1) & 1] = *reinterpret_cast<const h8*>(sb + ((aoff + (g + 1) * 2048) ^ 64));                            \
        }                                                                                                                 \
        constexpr int NP_ = ((H) == 0) ? A_IT + B_IT : B_IT;       /* piece slots of this phase kind, spread over TM/2 groups */ \
        constexpr int PPG_ = (NP_ + TM / 2 - 1) / (TM / 2);                                                               \
        auto pieces_ = [&]() {   /* (a lambda on purpose: written out in place, the kernel's register allocation changes) */     \
          _Pragma("unroll") for (int pp = 0; pp < PPG_; ++pp) {                                                           \
            const int sl = (g - G0) * PPG_ + pp;                                                                          \
            if ((H) == 0) {                                                                                               \
              if (WPF && (W_FULL1)) {   /* W(1) ahead of A(1): the next barrier's counted wait then retires W(1) */        \
                if (sl < B_IT) D3D_PIECE((KT) + 1, A_IT + sl);                                                            \
                else if (sl < A_IT + B_IT) { if (DO_A) D3D_PIECE((KT) + 1, sl - B_IT); }                                  \
              } else if (sl < A_IT) { if (DO_A) D3D_PIECE((KT) + 1, sl); }                                                \
              else if (sl < A_IT + B_IT) { if (W_FULL1) D3D_PIECE((KT) + 1, sl); }                                        \
            } else if (sl < B_IT) {                                                                                       \
              if (DO_W) D3D_PIECE((KT) + 2, A_IT + sl);                                                                   \
            }                                                                                                             \
          }                                                                                                               \
        };                                                                                                                \
        pieces_();                                                                                                        \
        const bool w_ahead_ = WPF && (H) == 1 && g == G1 - 1 && (W_AHEAD);                                                \
        if (g_act) {                                                                                                      \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                 \
            x3_mma<FX>(acc[g][j], bh[j], bl[j], ah[g & 1], al[g & 1]);                                                    \
            if (w_ahead_) {                                                                                               \
              const unsigned char* sbn = lds + (((KT) + 1) & 1) * STAGE;                                                  \
              bh[j] = *reinterpret_cast<const h8*>(sbn + boff + j * 2048);                                                \
              bl[j] = *reinterpret_cast<const h8*>(sbn + ((boff + j * 2048) ^ 64));                                       \
            }                                                                                                             \
          }                                                                                                               \
        }                                                                                                                 \
        if (WPF && !(FX & FX_BF16) && w_ahead_) {   /* MFMA triple, its W pair's successor, ...; the pieces in between */  \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
        } else                                                                                                            \
        if constexpr ((FX & FX_BF16) != 0) {   /* bf16: 8 MFMAs per group -- pairs where the F16X3 pattern has triples */       \
          if (!SUB && (H) == 0 && g == G0) {                                                                              \
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                                                            \
          } else if (!SUB) {                                                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                            \
            __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);                                                            \
          }                                                                                                               \
        } else                                                                                                            \
        if (!SUB && (H) == 0 && g == G0 && (!WPF || (W_FULL1))) {   /* the k-tile opening: fragments just ahead of their MFMAs */ \
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                                                              \
        } else if (!SUB) {                                                                                                \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);                                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                              \
        }                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        if (!SUB) {   /* a wave's priority falls with every group it has finished in this phase: the SIMD partner that is    \
                         behind wins the MFMA arbitration (otherwise always the older wave: it finishes its phase ~470        \
                         cycles early and leaves the younger one to run its last groups alone at 20 cycles per MFMA           \
                         instead of 16.5): qkv -0.6 %, fc1 -0.25 % per launch; the same in the one-barrier loop: +0.7 % */       \
          if (g - G0 == 0) __builtin_amdgcn_s_setprio(2);                                                                 \
          else if (g - G0 == 1) __builtin_amdgcn_s_setprio(1);                                                            \
          else __builtin_amdgcn_s_setprio(0);                                                                             \
          __builtin_amdgcn_sched_barrier(0);                                                                              \
        }                                                                                                                 \
      }                                                                                                                   \
      if ((H) == 0) issued_prev = ((DO_A) ? nA : 0) + ((W_FULL1) && !WPF ? B_IT : 0);                                     \
      else issued_prev = (DO_W) ? B_IT : 0;                                                                               \
    } while (0)
    h8 bh[4], bl[4], ah[2], al[2];
    // first k-tile: everything issued before this tile (stores of the previous epilogue included) has landed: vmcnt(0) -- but
    // for the statistics pieces just requested (the newest: vmcnt retires in order)
    issued_prev = st_issued;
    D3D_PHASE(0, 0, true, true, false, false);
    D3D_PHASE(0, 1, false, false, nk > 2 || has_next, nk > 1);
    int kt = 1;
    for (; kt + 2 < nk; ++kt) {
      D3D_PHASE(kt, 0, true, false, false, false);
      D3D_PHASE(kt, 1, false, false, true, true);
    }
    if (nk > 2) {   // k-tile nk-2: A(nk-1) of this tile, then W(0) of the next tile
      D3D_PHASE(kt, 0, true, false, false, false);
      D3D_PHASE(kt, 1, false, false, has_next, true);
      ++kt;
    }
    // k-tile nk-1: A(0) of the next tile; W(1) of the next tile waits for its own phase 0 (the epilogue's patches)
    D3D_PHASE(kt, 0, has_next, false, false, false);
    D3D_PHASE(kt, 1, false, false, false, false);
#undef D3D_PHASE
#undef D3D_PIECE
  } else if constexpr (NST == 2) {
    int kt = 0;
    for (; kt + 1 < nk; ++kt) D3D_QKTILE(kt, true, 0);
    D3D_QKTILE(kt, false, 0);
  } else {
    static_assert(N_IT == 2 || N_IT == 3 || N_IT == 4 || N_IT == 6 || N_IT == 8 || N_IT == 9, "counted waits listed above");
    int kt = 0;
    for (; kt + NST - 1 < nk; ++kt) D3D_QKTILE(kt, true, (NST - 2) * N_IT);        // k-tiles kt + 1 .. kt + NST - 2 still in flight
    for (; kt < nk; ++kt) D3D_QKTILE(kt, false, (nk - 1 - kt < NST - 2 ? nk - 1 - kt : NST - 2) * N_IT);
  }
#undef D3D_QKTILE
#undef D3D_QSTAGE_ONE
#undef D3D_QSTAGE_NEXT

  if constexpr (ST_DMA_FORM) {
    if (st_dma) {   // the raw partials landed under the k-loop (every wave's second-phase wait retired its pieces -- vmcnt is in
                    // order --, and the phase barriers since made them visible): reduce them from LDS, in the order of the direct form
      x3_row_stats([&](int r, int p) { return reinterpret_cast<const float2*>(lds_raw)[r * fx.st_np + p]; });
      __syncthreads();
    }
  }
  const int mt0 = m0 + wm * 16 * TM, nt0 = n0 + wn * 64;          // wave-uniform
  const size_t tbase = (size_t)mt0 * N + nt0;
  const float* Rt = R ? R + tbase : nullptr;
  float* Ct = C ? C + tbase : nullptr;
  _Float16* Cht = Ch ? Ch + (OUTSPLIT == 2 ? 2 * tbase : tbase) : nullptr;
  _Float16* Clt = Cl ? Cl + tbase : nullptr;
  unsigned long long st_c1 = 0, st_r1 = 0;
  if (diag) { st_c1 = __builtin_amdgcn_s_memtime(); st_r1 = __builtin_amdgcn_s_memrealtime(); }
  float* patch = reinterpret_cast<float*>(lds + (PERSIST ? STAGE : 0)) + wave * (2 * 16 * 64);
  // row-statistics form: a wave-private kilobyte behind the patches collects the wave's (sum, sum of squares) rows (x3q_epilogue8)
  float2* const statp = ((FX & FX_SO) && !(FX & FX_LNF)) ? reinterpret_cast<float2*>(lds_x) + wave * 128 : nullptr;
  const _Float16* Rpt = (FX & FX_RP) ? fx.Rp + 2 * tbase : nullptr;
  constexpr bool PLANES = (OUTSPLIT != 0 || (FX & FX_RP)) && !(EPI == EPI_RESIDUAL && !(FX & FX_RP));
  const bool full = FULL || (m0 + BM <= M && n0 + BN <= N);
  bool done = false;
  if constexpr ((FX & FX_PN) != 0) {   // whole rows in the tile: post-norm here (the launcher guarantees N == BN)
    static_assert(WM == 1 && EPI == EPI_RESIDUAL && ((FX & FX_RP) || (FX & FX_BF16)) && OUTSPLIT != 1, "post-norm form");
    static_assert(STAGE >= 65536 + ((FX & FX_BF16) ? 4 : 2) * BM * WN * 4, "row-sum exchange(s) beside the patches");
    float* xch = reinterpret_cast<float*>(lds + STAGE + 65536);
    // (one copy per instantiation: with a checked and an unchecked copy under a branch the accumulators spill)
    if constexpr ((FX & FX_BF16) != 0)   // bf16 mode: fp32 stream in place (C), bf16 operand rows of the following LayerNorm (Ch)
      x3q_epilogue_rows_bf16<TM, WN, !FULL>(acc, patch, xch, xch + 2 * BM * WN, bias, Ct, Cht, fx, mt0, nt0, wn, lane, M, N, gl, gh);
    else
      x3q_epilogue_pn<TM, WN, OUTSPLIT, !FULL>(acc, patch, xch, bias, Ct, Cht, Rpt, fx, mt0, nt0, wn, lane, M, N, gl, gh);
    done = true;
  } else if constexpr (EPI == EPI_GELU && OUTSPLIT == 2) {   // hidden activation, accumulator order: no transpose
    static_assert(!(FX & (FX_RP | FX_SO)), "fc1 form");
    if constexpr (FULL)
      x3q_epilogue_acc<TM, WM, WN, FX, false>(acc, lds_x, bias, Cht, fx.csum, mt0, nt0, mt0 - m0, lane, M, N, gl, gh, fx.out_scale, fx.range);
    else if (full)
      x3q_epilogue_acc<TM, WM, WN, FX, false>(acc, lds_x, bias, Cht, fx.csum, mt0, nt0, mt0 - m0, lane, M, N, gl, gh, fx.out_scale, fx.range);
    else
      x3q_epilogue_acc<TM, WM, WN, FX, true>(acc, lds_x, bias, Cht, fx.csum, mt0, nt0, mt0 - m0, lane, M, N, gl, gh, fx.out_scale, fx.range);
    done = true;
  } else if constexpr (PLANES) {   // 8 columns per lane: 16-byte plane accesses
    if ((N & 7) == 0) {
      // (the row-statistics form keeps one copy per instantiation: with two copies under the branch its accumulators spill)
      if constexpr (FULL)
        x3q_epilogue8<TM, WM, WN, EPI, OUTSPLIT, FX, false>(acc, patch, lds_x, bias, Ct, Cht, Clt, Rpt, fx.csum, fx.st_out, mt0, nt0,
                                                           mt0 - m0, lane, M, N, qcols, gl, gh, fx.out_scale, fx.range, statp);
      else if (full && !(FX & FX_SO))
        x3q_epilogue8<TM, WM, WN, EPI, OUTSPLIT, FX, false>(acc, patch, lds_x, bias, Ct, Cht, Clt, Rpt, fx.csum, fx.st_out, mt0, nt0,
                                                           mt0 - m0, lane, M, N, qcols, gl, gh, fx.out_scale, fx.range, statp);
      else
        x3q_epilogue8<TM, WM, WN, EPI, OUTSPLIT, FX, true>(acc, patch, lds_x, bias, Ct, Cht, Clt, Rpt, fx.csum, fx.st_out, mt0, nt0,
                                                          mt0 - m0, lane, M, N, qcols, gl, gh, fx.out_scale, fx.range, statp);
      done = true;
    }
  }
  if (!done) {
    if constexpr (FULL)
      x3q_epilogue<TM, WM, WN, EPI, OUTSPLIT, FX, false>(acc, patch, lds_x, bias, Rt, Ct, Cht, Clt, Rpt, fx.csum, fx.st_out, mt0,
                                                        nt0, mt0 - m0, lane, M, N, qcols, gl, gh, fx.out_scale, fx.range);
    else if (full)
      x3q_epilogue<TM, WM, WN, EPI, OUTSPLIT, FX, false>(acc, patch, lds_x, bias, Rt, Ct, Cht, Clt, Rpt, fx.csum, fx.st_out, mt0,
                                                        nt0, mt0 - m0, lane, M, N, qcols, gl, gh, fx.out_scale, fx.range);
    else
      x3q_epilogue<TM, WM, WN, EPI, OUTSPLIT, FX, true>(acc, patch, lds_x, bias, Rt, Ct, Cht, Clt, Rpt, fx.csum, fx.st_out, mt0,
                                                       nt0, mt0 - m0, lane, M, N, qcols, gl, gh, fx.out_scale, fx.range);
  }
  if (diag) {
    __builtin_amdgcn_s_waitcnt(0);   // the wave's own stores issued and acknowledged
    const unsigned long long c2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      unsigned long long* d = diag + ((size_t)bid * NW + wave) * 6;
      d[0] = st_c0; d[1] = st_r0; d[2] = st_c1; d[3] = st_r1; d[4] = c2; d[5] = r2;
    }
  }
}

// The SUB instantiation (tail slices, ragged edge tiles): it carries the checked epilogues, so that the whole-tile path of the
// persistent walk has ONE branch-free epilogue.
template <int TM, int WM, int WN, int EPI, int OUTSPLIT, int FX>
__device__ __forceinline__ void x3q_tile_sub(const _Float16* __restrict__ Ap, const _Float16* __restrict__ Wp,
                                                       const float* __restrict__ bias, const float* R, float* C, _Float16* Ch,
                                                       _Float16* Cl, int M, int N, int K, int m0, int n0, int nt, int ntiles,
                                                       int qcols, const X3Tail& fx, bool has_next, int m0n, int n0n, int tid_in,
                                                       int sub_wm, int g_lo, int g_hi) {
  x3q_tile<TM, WM, WN, EPI, OUTSPLIT, FX, true, true>(Ap, Wp, bias, R, C, Ch, Cl, M, N, K, m0, n0, nt, ntiles, qcols, nullptr, fx,
                                                      has_next, m0n, n0n, tid_in, sub_wm, g_lo, g_hi);
}

// Uniform launch: every workgroup one BM x BN tile; blockIdx -> tile keeps all N-tiles of an M-tile on one XCD.
template <int TM, int WM, int WN, int EPI, int OUTSPLIT, int FX, int NST = 2>
__global__ __launch_bounds__(64 * WM * WN) void k_linear_x3q(const _Float16* __restrict__ Ap, const _Float16* __restrict__ Wp,
                                                             const float* __restrict__ bias, const float* R, float* C,
                                                             _Float16* Ch, _Float16* Cl, int M, int N, int K, int mtiles,
                                                             int ntiles, int qcols, unsigned long long* diag, X3Tail fx) {
  constexpr int BM = 16 * TM * WM, BN = 64 * WN;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int mt = (slot / ntiles) * 8 + xcd;
  const int nt = slot % ntiles;
  if (mt >= mtiles) return;
  x3q_tile<TM, WM, WN, EPI, OUTSPLIT, FX, false, false, false, NST>(Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mt * BM, nt * BN, nt, ntiles,
                                                                    qcols, diag, fx);
}

// Persistent launch (one 8-wave workgroup per CU): the workgroup walks the tiles blockIdx, blockIdx + gridDim, ... of the
// uniform launch's order (so it stays on its XCD class, gridDim % 8 == 0).  Saves the per-tile workgroup relaunch and hides
// the first-k-tile staging latency of every tile but the first (x3q_tile, PERSIST).
// Tail: tiles = nfull * gridDim + rem.  The rem tiles of the last, partly filled round are cut into `split` (1, 2 or 4) row
// slices handled by split * rem <= gridDim workgroups (x3q_tile's sub_wm / g_lo / g_hi), so the round that would keep rem CUs
// busy for a whole tile time keeps split * rem CUs busy for a fraction of it.  A slice runs the same MFMAs in the same
// order for its rows as the whole tile would: values do not change.
struct X3Walk { int nfull, rem, split; unsigned long long* stamps; };   // stamps: diagnostic (100 MHz start / end per workgroup)

template <int TM, int WM, int WN, int EPI, int OUTSPLIT, int FX>
__global__ __launch_bounds__(512) void k_linear_x3q_persist(const _Float16* __restrict__ Ap, const _Float16* __restrict__ Wp,
                                                            const float* __restrict__ bias, const float* R, float* C,
                                                            _Float16* Ch, _Float16* Cl, int M, int N, int K, int mtiles,
                                                            int ntiles, int qcols, X3Walk wk, X3Tail fx) {
  constexpr int NW = WM * WN, BM = 16 * TM * WM, BN = 64 * WN;
  constexpr int A_IT = BM / 8 / NW, N_IT = (BM + BN) / 8 / NW;
  static_assert(NW == 8, "one 8-wave workgroup per CU");
  static_assert(TM % 2 == 0 && (WM == 1 || WM == 2), "tail slices");   // (four-way slices need TM % 4 == 0: x3q_walk)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  const int vfull = (mtiles / 8) * 8 * ntiles, mrem = mtiles % 8;
  // valid-tile ordinal -> tile: the uniform launch's blockIdx order without its padding slots
  auto tile_of = [&](int o, int& mt, int& nt) {
    if (o < vfull) {
      const int xcd = o & 7, slot = o >> 3;
      mt = (slot / ntiles) * 8 + xcd;
      nt = slot % ntiles;
    } else {
      const int o2 = o - vfull;
      mt = (mtiles / 8) * 8 + o2 % mrem;
      nt = o2 / mrem;
    }
  };
  const int nitems = wk.nfull + (b < wk.split * wk.rem ? 1 : 0);
  if (nitems == 0) return;
  const unsigned long long t_begin = wk.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
  auto stamp_end = [&]() {
    if (wk.stamps && threadIdx.x == 0) {
      __builtin_amdgcn_s_waitcnt(0);
      wk.stamps[2 * b] = t_begin;
      wk.stamps[2 * b + 1] = __builtin_amdgcn_s_memrealtime();
    }
  };
  // item k of this workgroup: a whole tile (k < nfull) or a slice of a tail tile
  auto item_of = [&](int k, int& mt, int& nt, int& sub_wm, int& g_lo, int& g_hi) {
    sub_wm = -1; g_lo = 0; g_hi = TM;
    if (k < wk.nfull) {
      tile_of(k * G + b, mt, nt);
      return;
    }
    tile_of(wk.nfull * G + b / wk.split, mt, nt);
    const int sub = b % wk.split;
    if (wk.split == 2) {
      if (WM == 2) sub_wm = sub;
      else { g_lo = sub * (TM / 2); g_hi = g_lo + TM / 2; }
    } else if (wk.split == 4) {
      if (WM == 2) { sub_wm = sub >> 1; g_lo = (sub & 1) * (TM / 2); g_hi = g_lo + TM / 2; }
      else { g_lo = sub * (TM / 4); g_hi = g_lo + TM / 4; }
    }
  };
  int k = 0, mt = 0, nt = 0, sub_wm = -1, g_lo = 0, g_hi = TM;
  item_of(0, mt, nt, sub_wm, g_lo, g_hi);
  {   // stage the first k-tile of the first item (what x3q_tile does for itself in the uniform launch)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int m0 = mt * BM, n0 = nt * BN;
    D3D_DMA_PLAN(NW, BM);
#pragma unroll
    for (int it = 0; it < N_IT; ++it) {
      if (it < A_IT) D3D_GLDS(sgpr_ptr(ubA + it * it_stride) + lofs_, dstA + it * NW * 1024);
      else D3D_GLDS(sgpr_ptr(ubB + (it - A_IT) * it_stride) + lofs_, dstB + (it - A_IT) * NW * 1024);
    }
  }
  int tid_o = (int)threadIdx.x;
  const int nwhole = (nitems > wk.nfull && wk.split > 1) ? wk.nfull : nitems;   // items run as whole tiles
  while (k < nwhole) {
    asm volatile("" : "+v"(tid_o));
    const bool has_next = k + 1 < nitems;
    int mtn = 0, ntn = 0, swn = -1, gln = 0, ghn = TM;
    if (has_next) item_of(k + 1, mtn, ntn, swn, gln, ghn);
    if ((mt + 1) * BM <= M && (nt + 1) * BN <= N)   // (wave-uniform) whole tile inside the matrix: the unchecked instantiation
      x3q_tile<TM, WM, WN, EPI, OUTSPLIT, FX, true, false, true>(Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mt * BM, nt * BN, nt, ntiles, qcols,
                                                                 nullptr, fx, has_next, mtn * BM, ntn * BN, tid_o);
    else                                            // ragged edge tile: the checked epilogue lives in the SUB instantiation
      x3q_tile_sub<TM, WM, WN, EPI, OUTSPLIT, FX>(Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mt * BM, nt * BN, nt, ntiles, qcols, fx, has_next,
                                                  mtn * BM, ntn * BN, tid_o, -1, 0, TM);
    if (!has_next) { stamp_end(); return; }
    ++k; mt = mtn; nt = ntn; sub_wm = swn; g_lo = gln; g_hi = ghn;
    __syncthreads();   // the epilogue's patches (stage 1) are read before the next tile's second k-tile is staged there
  }
  asm volatile("" : "+v"(tid_o));
  x3q_tile_sub<TM, WM, WN, EPI, OUTSPLIT, FX>(Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mt * BM, nt * BN, nt, ntiles, qcols, fx, false, 0, 0,
                                              tid_o, sub_wm, g_lo, g_hi);
  stamp_end();
}

// Diagnostic stamp buffer for the next launches (experiments/gemm_bench.py through d3d_op_linear_bench): variant 13 -> per-wave
// k-loop / epilogue stamps; the persistent walk -> start / end per workgroup.
static unsigned long long* g_x3_diag = nullptr;

thread_local LaunchCtx tl_launch_ctx;
static std::atomic<bool> g_x3q_deep{true};
void set_x3q_deep_stages(bool on) { g_x3q_deep = on; }

// walk of `tiles` tiles over `grid` persistent workgroups
static X3Walk x3q_walk(int tiles, int grid, bool four_way = true) {
  X3Walk w{tiles / grid, tiles % grid, 1, g_x3_diag};
  if (w.rem > 0 && tl_launch_ctx.tail_slices) w.split = (four_way && 4 * w.rem <= grid) ? 4 : ((2 * w.rem <= grid) ? 2 : 1);
  return w;
}

template <int TM, int WM, int WN, int NST = 2>
static hipError_t launch_x3q(const _Float16* Ap, const _Float16* Wp, const float* bias, const float* R, float* C, _Float16* Ch,
                             _Float16* Cl, int M, int N, int K, int epi, int outsplit, int qcols, hipStream_t s,
                             unsigned long long* diag = nullptr, const X3Fold* fold = nullptr, int w_exp = 12, bool bf16 = false) {
  constexpr int BM = 16 * TM * WM, BN = 64 * WN;
  const int mtiles = (M + BM - 1) / BM, ntiles = (N + BN - 1) / BN;
  const int grid = ((mtiles + 7) / 8) * 8 * ntiles;
  size_t lds_bytes = NST * (size_t)((BM + BN) * 128);
  X3Tail tail{};
  tail.out_scale = ldexpf(1.0f, -(3 + w_exp));
  tail.range = launch_range_word();
  int fx = 0;
  if (fold) {
    if (fold->st_in) fx |= FX_LNF;
    if (fold->Rp) fx |= FX_RP;
    if (fold->st_out) fx |= FX_SO;
    tail.st_in = fold->st_in; tail.st_np = fold->st_np; tail.csum = fold->csum; tail.eps = fold->eps;
    tail.Rp = (const _Float16*)fold->Rp; tail.st_out = fold->st_out;
    if (fx & FX_LNF) lds_bytes += (size_t)BM * 8;   // row statistics beyond the operand stages
    else if (fx & FX_SO) lds_bytes += 8 * 1024;     // a kilobyte per wave for the rows' statistics (x3q_epilogue8)
    if ((fx & FX_LNF) && (!fold->csum || fold->st_np < 1)) return hipErrorInvalidValue;
  }
#define D3D_X3Q_LAUNCH_FX(EPI_, OS_, FX_)                                                                                 \
  do {                                                                                                                    \
    auto kfn = k_linear_x3q<TM, WM, WN, EPI_, OS_, FX_, NST>;                                                             \
    static std::atomic<unsigned long long> attr_done{0};   /* one bit per device */                                       \
    if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(kfn), lds_bytes, attr_done)) return ae;                   \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * WM * WN), lds_bytes, s, Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mtiles,    \
                       ntiles, qcols, diag, tail);                                                                        \
  } while (0)
#define D3D_X3Q_LAUNCH(EPI_, OS_) D3D_X3Q_LAUNCH_FX(EPI_, OS_, 0)
  if (bf16) {   // bf16 operand mode: the three forms its block flow uses (launch_linear_bf16)
    tail.out_scale = 1.0f;
    if constexpr (WM * WN == 8) {
      if (fx == 0 && epi == EPI_NONE && outsplit == 3) D3D_X3Q_LAUNCH_FX(EPI_NONE, 3, FX_BF16);
      else if (fx == 0 && epi == EPI_GELU && outsplit == 3) D3D_X3Q_LAUNCH_FX(EPI_GELU, 3, FX_BF16);
      else if (fx == 0 && epi == EPI_RESIDUAL && outsplit == 0) D3D_X3Q_LAUNCH_FX(EPI_RESIDUAL, 0, FX_BF16);
      else return hipErrorInvalidValue;
    } else {
      return hipErrorInvalidValue;
    }
  } else if (fx == 0) {
    D3D_X3_DISPATCH(D3D_X3Q_LAUNCH);
  } else if constexpr (WM * WN == 8) {   // folded forms exist for the production (8-wave) shapes only
    // the four folded forms of the engine's plane-resident block (engine.hip run_blocks)
    if (fx == FX_LNF && epi == EPI_NONE && outsplit == 1) D3D_X3Q_LAUNCH_FX(EPI_NONE, 1, FX_LNF);                        // qkv
    else if (fx == (FX_RP | FX_SO) && epi == EPI_RESIDUAL && outsplit == 2) D3D_X3Q_LAUNCH_FX(EPI_RESIDUAL, 2, FX_RP | FX_SO);  // proj
    else if (fx == FX_LNF && epi == EPI_GELU && outsplit == 2) D3D_X3Q_LAUNCH_FX(EPI_GELU, 2, FX_LNF);                   // fc1
    else if (fx == FX_RP && epi == EPI_RESIDUAL && outsplit == 0) D3D_X3Q_LAUNCH_FX(EPI_RESIDUAL, 0, FX_RP);             // fc2
    else return hipErrorInvalidValue;
  } else {
    return hipErrorInvalidValue;
  }
#undef D3D_X3Q_LAUNCH
#undef D3D_X3Q_LAUNCH_FX
  return hipGetLastError();
}

// Tile choice: 256x256 -- as a persistent walk, one workgroup per CU (k_linear_x3q_persist: +1 % over one workgroup per tile:
// the next tile's first k-tile lands under the epilogue; the tiles of the partly filled last round are cut into row slices,
// +1 %) -- wherever it fills the chip for a few rounds, else 256x128.
// History of the tail: a second launch of 64x256 tiles for the remainder rows gained nothing (launch gap); a first single
// launch carrying two tile shapes computed wrong, run-to-run different values.  Its cause was found later with the slices:
// in an instantiation whose DMA issues sit behind uniform branches the compiler emits NO vmcnt wait before the k-tile
// barrier (the fence of __syncthreads() normally provides it), so fragments were read before the last-issued pieces had
// landed.  D3D_QKTILE now states the wait itself.
template <int TM = 8>
static hipError_t launch_x3q_persist(const _Float16* Ap, const _Float16* Wp, const float* bias, const float* R, float* C,
                                     _Float16* Ch, _Float16* Cl, int M, int N, int K, int epi, int outsplit, int qcols,
                                     hipStream_t s, const X3Fold* fold, int w_exp, bool bf16 = false) {
  constexpr int BM = 32 * TM;
  const int mtiles = (M + BM - 1) / BM, ntiles = (N + 255) / 256;
  const int tiles = mtiles * ntiles;
  int n_cu = device_cu_count() / 8 * 8;   // (per device)
  if (n_cu < 8) n_cu = 8;
  const int grid = tiles < n_cu ? tiles / 8 * 8 : n_cu;
  if (grid < 8) return hipErrorInvalidValue;
  const X3Walk wk = x3q_walk(tiles, grid, TM % 4 == 0);
  constexpr size_t STAGE = (size_t)(BM + 256) * 128;
  size_t lds_bytes = std::max(2 * STAGE, STAGE + 65536);     // two operand stages; the epilogue patches (64 KiB) start at stage 1
  X3Tail tail{};
  tail.out_scale = ldexpf(1.0f, -(3 + w_exp));
  tail.range = launch_range_word();
  int fx = 0;
  if (fold) {
    if (fold->st_in) fx |= FX_LNF;
    if (fold->Rp) fx |= FX_RP;
    if (fold->st_out) fx |= FX_SO;
    tail.st_in = fold->st_in; tail.st_np = fold->st_np; tail.csum = fold->csum; tail.eps = fold->eps;
    tail.Rp = (const _Float16*)fold->Rp; tail.st_out = fold->st_out;
    if (fx & FX_LNF) lds_bytes += (size_t)BM * 8 + 16384;   // (rstd, -mean rstd) per row + the raw partials staged by LDS-DMA
    else if (fx & FX_SO) lds_bytes += 8 * 1024;              // a kilobyte per wave for the rows' statistics (x3q_epilogue8)
    if ((fx & FX_LNF) && (!fold->csum || fold->st_np < 1)) return hipErrorInvalidValue;
  }
#define D3D_X3P_LAUNCH_FX(EPI_, OS_, FX_)                                                                                 \
  do {                                                                                                                    \
    auto kfn = k_linear_x3q_persist<TM, 2, 4, EPI_, OS_, FX_>;                                                                     \
    static std::atomic<unsigned long long> attr_done{0};   /* one bit per device */                                       \
    if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(kfn), lds_bytes, attr_done)) return ae;                   \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds_bytes, s, Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mtiles, ntiles,     \
                       qcols, wk, tail);                                                                                  \
  } while (0)
#define D3D_X3P_LAUNCH(EPI_, OS_) D3D_X3P_LAUNCH_FX(EPI_, OS_, 0)
  if constexpr (TM != 8) {   // the 192-row walk exists for the proj form only
    if (!bf16 && fx == (FX_RP | FX_SO) && epi == EPI_RESIDUAL && outsplit == 2) D3D_X3P_LAUNCH_FX(EPI_RESIDUAL, 2, FX_RP | FX_SO);
    else return hipErrorInvalidValue;
  } else if (bf16) {
    tail.out_scale = 1.0f;
    if (fx == 0 && epi == EPI_NONE && outsplit == 3) D3D_X3P_LAUNCH_FX(EPI_NONE, 3, FX_BF16);
    else if (fx == 0 && epi == EPI_GELU && outsplit == 3) D3D_X3P_LAUNCH_FX(EPI_GELU, 3, FX_BF16);
    else if (fx == 0 && epi == EPI_RESIDUAL && outsplit == 0) D3D_X3P_LAUNCH_FX(EPI_RESIDUAL, 0, FX_BF16);
    else return hipErrorInvalidValue;
  } else if (fx == 0) {
    D3D_X3_DISPATCH(D3D_X3P_LAUNCH);
  } else {
    if (fx == FX_LNF && epi == EPI_NONE && outsplit == 1) D3D_X3P_LAUNCH_FX(EPI_NONE, 1, FX_LNF);
    else if (fx == (FX_RP | FX_SO) && epi == EPI_RESIDUAL && outsplit == 2) D3D_X3P_LAUNCH_FX(EPI_RESIDUAL, 2, FX_RP | FX_SO);
    else if (fx == FX_LNF && epi == EPI_GELU && outsplit == 2) D3D_X3P_LAUNCH_FX(EPI_GELU, 2, FX_LNF);
    else if (fx == FX_RP && epi == EPI_RESIDUAL && outsplit == 0) D3D_X3P_LAUNCH_FX(EPI_RESIDUAL, 0, FX_RP);
    else return hipErrorInvalidValue;
  }
#undef D3D_X3P_LAUNCH
#undef D3D_X3P_LAUNCH_FX
  return hipGetLastError();
}

// Post-norm form (X3Fold::pn): 128 x 512 tiles (<8,1,8>: eight waves of 128 x 64 side by side, 2 x 80 KiB of LDS -- the whole
// CU), so that a workgroup owns whole rows.  Per k-tile it stages 80 KiB for the MFMA work a 256x256 tile does on 64 KiB;
// what it saves is the fp32 round trip of the stream through HBM and the row kernel behind the fc2 GEMM.
// The same tile function serves every M (persistent walk when the chip is filled for a few rounds, one workgroup per tile
// otherwise), so the result stays batch-size independent bitwise.
bool x3q_postnorm_ok(int N, int K) { return N == 512 && K % PBK == 0; }

static hipError_t launch_x3q_pn(const _Float16* Ap, const _Float16* Wp, const float* bias, float* C, _Float16* Ch, int M, int N,
                                int K, int outsplit, hipStream_t s, const X3Fold* fold, int w_exp) {
  if (!x3q_postnorm_ok(N, K) || !fold->Rp || !fold->pn.b || !bias || (outsplit == 2 ? (!Ch || !fold->st_out) : !C))
    return hipErrorInvalidValue;
  if (outsplit != 0 && outsplit != 2) return hipErrorInvalidValue;
  if (fold->pn.pos && (fold->pn.pos_div < 1 || fold->pn.pos_mod < 1)) return hipErrorInvalidValue;
  if (fold->pn.tvec && fold->pn.tvec_stride != 0 && fold->pn.rows_per_batch < 1) return hipErrorInvalidValue;
  const int mtiles = (M + 127) / 128, ntiles = 1;
  const int vtiles = ((mtiles + 7) / 8) * 8;
  int n_cu = device_cu_count() / 8 * 8;   // (per device)
  if (n_cu < 8) n_cu = 8;
  const bool persist = mtiles >= 4 * n_cu && (K / PBK) % 2 == 0;
  const X3Walk wk = x3q_walk(mtiles * ntiles, n_cu);
  const size_t lds_bytes = 2 * (size_t)((128 + 512) * 128);
  const bool small = mtiles < n_cu;
  const int mtiles64 = (M + 63) / 64, vtiles64 = ((mtiles64 + 7) / 8) * 8;
  const size_t lds_small = 2 * (size_t)((64 + 512) * 128);
  X3Tail tail{};
  tail.out_scale = ldexpf(1.0f, -(3 + w_exp));
  tail.range = launch_range_word();
  tail.Rp = (const _Float16*)fold->Rp; tail.st_out = fold->st_out; tail.pn = fold->pn;
  const int qcols = 0;
  unsigned long long* diag = nullptr;
  _Float16* Cl = nullptr;
  const float* R = nullptr;
#define D3D_X3PN_LAUNCH(OS_)                                                                                              \
  do {                                                                                                                    \
    if (persist) {                                                                                                        \
      auto kfn = k_linear_x3q_persist<8, 1, 8, EPI_RESIDUAL, OS_, FX_RP | FX_PN>;                                         \
      static std::atomic<unsigned long long> attr_done{0};   /* one bit per device */                                     \
      if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(kfn), lds_bytes, attr_done)) return ae;                 \
      hipLaunchKernelGGL(kfn, dim3(n_cu), dim3(512), lds_bytes, s, Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mtiles, ntiles,   \
                         qcols, wk, tail);                                                                                \
    } else if (small) {   /* fewer 128-row tiles than CUs: 64-row tiles (same values: rows are independent) */            \
      auto kfn = k_linear_x3q<4, 1, 8, EPI_RESIDUAL, OS_, FX_RP | FX_PN>;                                                 \
      static std::atomic<unsigned long long> attr_done{0};   /* one bit per device */                                     \
      if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(kfn), lds_small, attr_done)) return ae;                 \
      hipLaunchKernelGGL(kfn, dim3(vtiles64), dim3(512), lds_small, s, Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mtiles64, ntiles, \
                         qcols, diag, tail);                                                                              \
    } else {                                                                                                              \
      auto kfn = k_linear_x3q<8, 1, 8, EPI_RESIDUAL, OS_, FX_RP | FX_PN>;                                                 \
      static std::atomic<unsigned long long> attr_done{0};   /* one bit per device */                                     \
      if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(kfn), lds_bytes, attr_done)) return ae;                 \
      hipLaunchKernelGGL(kfn, dim3(vtiles), dim3(512), lds_bytes, s, Ap, Wp, bias, R, C, Ch, Cl, M, N, K, mtiles, ntiles, \
                         qcols, diag, tail);                                                                              \
    }                                                                                                                     \
  } while (0)
  if (outsplit == 2) D3D_X3PN_LAUNCH(2);
  else D3D_X3PN_LAUNCH(0);
#undef D3D_X3PN_LAUNCH
  return hipGetLastError();
}

// bf16 mode, whole-row form (d3d_kernels.h launch_linear_bf16_rows): the 128 x 512 tile shape of the post-norm form, bf16 MFMAs,
// x3q_epilogue_rows_bf16.  "K / 2 pair columns" as in launch_linear_bf16.
bool bf16_rows_ok(int N, int K) { return N == 512 && K % 64 == 0; }

hipError_t launch_linear_bf16_rows(const void* A, const void* W, const float* bias, float* X, void* Hb, int M, int N, int K,
                                   const X3PostNorm& pn, hipStream_t s) {
  if (!bf16_rows_ok(N, K) || M <= 0 || !A || !W || !bias || !X) return hipErrorInvalidValue;
  if (pn.g2 ? (!Hb || !pn.b2) : false) return hipErrorInvalidValue;
  if (pn.g && !pn.b) return hipErrorInvalidValue;
  if (pn.pos && (pn.pos_div < 1 || pn.pos_mod < 1)) return hipErrorInvalidValue;
  if (pn.tvec && pn.tvec_stride != 0 && pn.rows_per_batch < 1) return hipErrorInvalidValue;
  const _Float16 *Ap = (const _Float16*)A, *Wp = (const _Float16*)W;
  _Float16* Ch = (_Float16*)Hb;
  _Float16* Cl = nullptr;
  const float* R = nullptr;
  float* C = X;
  const int K2 = K / 2;
  const int mtiles = (M + 127) / 128, ntiles = 1;
  const int vtiles = ((mtiles + 7) / 8) * 8;
  int n_cu = device_cu_count() / 8 * 8;   // (per device)
  if (n_cu < 8) n_cu = 8;
  const bool persist = mtiles >= 4 * n_cu && (K2 / PBK) % 2 == 0;
  const X3Walk wk = x3q_walk(mtiles * ntiles, n_cu);
  const size_t lds_bytes = 2 * (size_t)((128 + 512) * 128);
  const bool small = mtiles < n_cu;
  const int mtiles64 = (M + 63) / 64, vtiles64 = ((mtiles64 + 7) / 8) * 8;
  const size_t lds_small = 2 * (size_t)((64 + 512) * 128);
  X3Tail tail{};
  tail.out_scale = 1.0f;
  tail.range = launch_range_word();
  tail.pn = pn;
  const int qcols = 0;
  unsigned long long* diag = nullptr;
  constexpr int FXB = FX_PN | FX_BF16;
  if (persist) {
    auto kfn = k_linear_x3q_persist<8, 1, 8, EPI_RESIDUAL, 0, FXB>;
    static std::atomic<unsigned long long> attr_done{0};   // one bit per device
    if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(kfn), lds_bytes, attr_done)) return ae;
    hipLaunchKernelGGL(kfn, dim3(n_cu), dim3(512), lds_bytes, s, Ap, Wp, bias, R, C, Ch, Cl, M, N, K2, mtiles, ntiles, qcols, wk, tail);
  } else if (small) {   // fewer 128-row tiles than CUs: 64-row tiles (same values: rows are independent)
    auto kfn = k_linear_x3q<4, 1, 8, EPI_RESIDUAL, 0, FXB>;
    static std::atomic<unsigned long long> attr_done{0};
    if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(kfn), lds_small, attr_done)) return ae;
    hipLaunchKernelGGL(kfn, dim3(vtiles64), dim3(512), lds_small, s, Ap, Wp, bias, R, C, Ch, Cl, M, N, K2, mtiles64, ntiles, qcols, diag, tail);
  } else {
    auto kfn = k_linear_x3q<8, 1, 8, EPI_RESIDUAL, 0, FXB>;
    static std::atomic<unsigned long long> attr_done{0};
    if (hipError_t ae = lds_optin(reinterpret_cast<const void*>(kfn), lds_bytes, attr_done)) return ae;
    hipLaunchKernelGGL(kfn, dim3(vtiles), dim3(512), lds_bytes, s, Ap, Wp, bias, R, C, Ch, Cl, M, N, K2, mtiles, ntiles, qcols, diag, tail);
  }
  return hipGetLastError();
}

static bool x3q_big(int M, int N) {
  const long long tiles = (long long)((M + 255) / 256) * ((N + 255) / 256);
  return N % 256 == 0 && tiles >= 4 * 256;
}
static bool x3q_small(int M, int N) {   // every 128 x 128 tile gets a CU of its own
  const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
  const int n_cu = device_cu_count();
  return N % 128 == 0 && n_cu > 0 && tiles <= (long long)n_cu;
}
int x3q_ntiles(int M, int N) { (void)M; return (N + 63) / 64; }   // statistics partials per row: one per 64 columns

static hipError_t launch_x3q_auto(const _Float16* ap, const _Float16* wp, const float* bias, const float* R, float* C,
                                  _Float16* ch, _Float16* cl, int M, int N, int K, int epi, int outsplit, int qcols,
                                  hipStream_t s, const X3Fold* fold, int w_exp) {
  // proj form (plane residual + row statistics; N = 512, the heaviest epilogue per MFMA): 192 x 256 tiles -- 232 VGPRs without a spill
  // where the 256-row tile sits at 256 with 8, 10.8 rounds instead of 8.07: 0.477 -> 0.455 ms per launch (same-box A/B, three
  // alternations; the same shape costs qkv +5.5 % and fc1 +2.3 %: they keep 256 rows).  Values do not depend on the tile shape.
  if (x3q_big(M, N) && (K / PBK) % 2 == 0 && fold && fold->Rp && fold->st_out && epi == EPI_RESIDUAL && outsplit == 2)
    return launch_x3q_persist<6>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, fold, w_exp);
  if (x3q_big(M, N) && (K / PBK) % 2 == 0)
    return launch_x3q_persist(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, fold, w_exp);
  if (x3q_big(M, N)) return launch_x3q<8, 2, 4>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, nullptr, fold, w_exp);
  // Smallest launches (a batch of one sequence: the visualisation scripts, the ragged last batch of evaluate()): where 128 x 128 tiles
  // (<2,4,2>: eight waves of 32 x 64) still find a CU each, the launch is as long as ONE tile takes and that tile is shorter -- proj at
  // B = 1, T = 243: 26 -> 20.7 us per launch.  Beyond that (two such workgroups sharing a CU) the shape loses: fc1 at B = 1 28.7 -> 38.2 us
  // (NOTES round 6).  Values do not depend on the tile shape.
  const bool deep = g_x3q_deep.load(std::memory_order_relaxed);
  if (x3q_small(M, N))
    return deep ? launch_x3q<2, 4, 2, 4>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, nullptr, fold, w_exp)
                : launch_x3q<2, 4, 2>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, nullptr, fold, w_exp);
  return deep ? launch_x3q<4, 4, 2, 3>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, nullptr, fold, w_exp)
              : launch_x3q<4, 4, 2>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, nullptr, fold, w_exp);
}

// ---- bf16 operand mode (D3D_PREC_BF16) ---------------------------------------------------------------------------------
// C = epi(A W^T + bias) with A [>= ceil(M/256)*256 rows][K] and W [>= ceil(N/256)*256 rows][K] as plain bf16 rows (K % 64 == 0;
// weights rounded once at commit, activations by their producer), ONE v_mfma_f32_16x16x32_bf16 per product, fp32 accumulate.
// The tile machinery is the F16X3 one: a 128-byte line of a row is now 64 bf16 k values instead of 32 (hi, lo) pairs, so the
// kernel is launched with "K / 2 pair columns" and differs only in x3_mma and in the output form:
//   EPI_NONE / EPI_GELU -> Cb, bf16 [M][N] (columns < qcols multiplied by 2^-3: the q third of a qkv GEMM, dh = 64)
//   EPI_RESIDUAL        -> C fp32 = R + ... (the fp32 residual stream; R may alias C)
// Same 256x256 persistent walk / 256x128 choice as launch_x3q_auto; values are tile-shape independent.
hipError_t launch_linear_bf16(const void* A, const void* W, const float* bias, const float* R, float* C, void* Cb, int M, int N,
                              int K, int epi, int qcols, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % 64) != 0 || (N % 8) != 0 || !A || !W) return hipErrorInvalidValue;
  if (epi == EPI_RESIDUAL ? (!R || !C) : !Cb) return hipErrorInvalidValue;
  const _Float16 *ap = (const _Float16*)A, *wp = (const _Float16*)W;
  _Float16* cb = (_Float16*)Cb;
  const int outsplit = epi == EPI_RESIDUAL ? 0 : 3;
  const int K2 = K / 2;                     // pair columns: 4 K2 bytes per row, K2 / 32 staged lines per row
  if (x3q_big(M, N) && (K2 / PBK) % 2 == 0)
    return launch_x3q_persist(ap, wp, bias, R, C, cb, nullptr, M, N, K2, epi, outsplit, qcols, s, nullptr, 12, true);
  if (x3q_big(M, N)) return launch_x3q<8, 2, 4>(ap, wp, bias, R, C, cb, nullptr, M, N, K2, epi, outsplit, qcols, s, nullptr, nullptr, 12, true);
  return launch_x3q<4, 4, 2>(ap, wp, bias, R, C, cb, nullptr, M, N, K2, epi, outsplit, qcols, s, nullptr, nullptr, 12, true);
}

// fp32 [rows, cols] -> bf16 (round to nearest even), and back: weight commit on the device side of the op hooks, tests
__global__ __launch_bounds__(256) void k_f32_to_bf16(const float* __restrict__ x, __bf16* __restrict__ y, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = (__bf16)x[i];
}
__global__ __launch_bounds__(256) void k_bf16_to_f32(const __bf16* __restrict__ x, float* __restrict__ y, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = (float)x[i];
}
// x[r][c] *= f for c < ncols_scaled (test hook: the q third of a packed qkv buffer)
__global__ __launch_bounds__(256) void k_scale_cols(float* __restrict__ x, size_t n, int cols, int ncols_scaled, float f) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && (int)(i % cols) < ncols_scaled) x[i] *= f;
}
hipError_t launch_scale_cols(float* x, size_t rows, int cols, int ncols_scaled, float f, hipStream_t s) {
  const size_t n = rows * cols;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_scale_cols, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, cols, ncols_scaled, f);
  return hipGetLastError();
}
hipError_t launch_f32_to_bf16(const float* x, void* y, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_f32_to_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, (__bf16*)y, n);
  return hipGetLastError();
}
hipError_t launch_bf16_to_f32(const void* x, float* y, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_bf16_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const __bf16*)x, y, n);
  return hipGetLastError();
}

void set_linear_x3_diag(unsigned long long* dev_buf) { g_x3_diag = dev_buf; }

// variant: 0 = auto (launch_x3q_auto); the two production shapes forced, one workgroup per tile (experiments/gemm_bench.py):
// 13 = 256x256 (with the per-wave diagnostic stamps when set_linear_x3_diag() armed them), 4 = 256x128
hipError_t launch_linear_x3p(const void* Ap_, const void* Wp_, const float* bias, const float* R, float* C, void* Ch, void* Cl,
                             int M, int N, int K, int epi, int outsplit, int qcols, int variant, hipStream_t s,
                             const X3Fold* fold, int w_exp) {
  if (w_exp < -14 || w_exp > 12) return hipErrorInvalidValue;
  if (M <= 0 || N <= 0 || K <= 0 || (K % PBK) != 0 || (N % 4) != 0) return hipErrorInvalidValue;
  if (epi == EPI_RESIDUAL && R == nullptr && !(fold && fold->Rp)) return hipErrorInvalidValue;
  if (fold && variant != 0) return hipErrorInvalidValue;
  if (outsplit == 0 ? !C : outsplit == 1 ? (!Ch || !Cl) : (!Ch || (N % 32) != 0)) return hipErrorInvalidValue;
  const _Float16 *ap = (const _Float16*)Ap_, *wp = (const _Float16*)Wp_;
  _Float16 *ch = (_Float16*)Ch, *cl = (_Float16*)Cl;
  if (fold && fold->pn.g) {
    if (epi != EPI_RESIDUAL) return hipErrorInvalidValue;
    return launch_x3q_pn(ap, wp, bias, C, ch, M, N, K, outsplit, s, fold, w_exp);
  }
  switch (variant) {
    case 0: return launch_x3q_auto(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, fold, w_exp);
    case 13: return launch_x3q<8, 2, 4>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, g_x3_diag, nullptr, w_exp);
    case 4: return launch_x3q<4, 4, 2>(ap, wp, bias, R, C, ch, cl, M, N, K, epi, outsplit, qcols, s, nullptr, nullptr, w_exp);
    default: return hipErrorInvalidValue;
  }
}

// fp32 [rows, cols] -> pair layout of 8*x (stand-alone converter: tests, and any activation whose producer is not ours)
__global__ __launch_bounds__(256) void k_split_x3(const float* __restrict__ x, _Float16* __restrict__ pair, size_t n4, int cols,
                                                  unsigned* rw) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(x)[i];
  const size_t row = (4 * i) / cols;
  const int c = (int)(4 * i - row * cols);
  const float f[4] = {v.x, v.y, v.z, v.w};
  h4 a, b;
  float amax = 0.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    amax = __builtin_fmaxf(amax, __builtin_fabsf(f[j] * P_A_SCALE));
    const float s = __builtin_amdgcn_fmed3f(f[j] * P_A_SCALE, -65504.0f, 65504.0f);
    a[j] = (_Float16)s;
    b[j] = (_Float16)(s - (float)a[j]);
  }
  range_note(rw, amax);
  _Float16* p = pair + row * 2 * cols + pair_col(c);
  *reinterpret_cast<h4*>(p) = a;
  *reinterpret_cast<h4*>(p + PAIR_LO) = b;
}

hipError_t launch_split_x3(const float* x, void* pair, size_t rows, int cols, hipStream_t s) {
  if (cols <= 0 || cols % 32) return hipErrorInvalidValue;
  const size_t n4 = rows * cols / 4;
  if (n4 == 0) return hipSuccess;
  hipLaunchKernelGGL(k_split_x3, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, (_Float16*)pair, n4, cols, launch_range_word());
  return hipGetLastError();
}

// pair layout of 8*x -> fp32 [rows, cols], and the row totals of (sum, sum of squares) partials: read-back side of the
// plane-form op hooks (d3d_op_linear_postnorm); not on the engine's path
__global__ __launch_bounds__(256) void k_unsplit_x3(const _Float16* __restrict__ pair, float* __restrict__ x, size_t n, int cols,
                                                    const float* __restrict__ part, int np, float* __restrict__ stats,
                                                    size_t rows) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const size_t row = i / cols;
    const int c = (int)(i - row * cols);
    const _Float16* p = pair + row * 2 * cols + pair_col(c);
    x[i] = ((float)p[0] + (float)p[PAIR_LO]) * 0.125f;
  }
  if (stats && i < rows) {
    float sm = 0.f, sq = 0.f;
    for (int k = 0; k < np; ++k) { sm += part[2 * (i * np + k)]; sq += part[2 * (i * np + k) + 1]; }
    stats[2 * i] = sm; stats[2 * i + 1] = sq;
  }
}

hipError_t launch_unsplit_x3(const void* pair, float* x, size_t rows, int cols, const float* part, int np, float* stats,
                             hipStream_t s) {
  if (cols <= 0 || cols % 32) return hipErrorInvalidValue;
  const size_t n = rows * cols;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_unsplit_x3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const _Float16*)pair, x, n, cols, part, np,
                     stats, rows);
  return hipGetLastError();
}

}  // namespace d3d
