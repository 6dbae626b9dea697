// Internal launcher interface between the engine (engine.hip) and the gfx950 kernels.
// Not part of the public ABI (that is include/d3d.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

namespace d3d {

// ---- per-device launch state ----------------------------------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device's function object, and the CU count is a
// property of a device: both are cached per device (one process may drive several GPUs -- nn.DataParallel, RUN:216-218 --
// from several threads).  `done` holds one bit per device ordinal; racing threads at worst repeat the idempotent call.
inline hipError_t lds_optin(const void* fn, size_t bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}
// compute units of the current device; 0 on error
inline int device_cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  int n = cache[dev & 63].load(std::memory_order_acquire);
  if (n > 0) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 0;
  cache[dev & 63].store(n, std::memory_order_release);
  return n;
}
// workgroups of `fn` (block threads, dyn_lds bytes of dynamic LDS) that one CU holds at a time -- registers AND LDS, as the runtime
// computes it; a persistent kernel sized beyond this runs its surplus workgroups as a second, thinner wave of work.  0 on error.
inline int resident_workgroups_per_cu(const void* fn, int block, size_t dyn_lds) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, block, dyn_lds) != hipSuccess) return 0;
  return nb;
}

// Launch context of the calling thread (launches are issued synchronously by the thread that calls the C ABI).  tail_slices:
// whether a persistent GEMM walk cuts the tiles of its partly filled last round into row slices (kernels_gemm_x3p.hip, X3Walk).
// On one stream that fills CUs which would idle for a tile time (+1.3 %); with two half-batches on two streams the other half's
// kernel takes those CUs, and slices -- 4x the workgroups for 0.7-1.0 of a tile time each -- only cost CU time (-0.6 %), so the
// two-stream loop turns them off.  Values do not depend on it.
// range_word: the F16X3 range-guard word (device memory, below) of the ENGINE whose call is being served -- the C ABI entry points of
// an engine set it around their launches; launches outside an engine (the single-op hooks) write to a per-device sink nobody reads.
struct LaunchCtx { bool tail_slices = true; unsigned* range_word = nullptr; };
extern thread_local LaunchCtx tl_launch_ctx;
// Process-wide switch (engine option "deep_stages", default on): the one-tile-per-workgroup GEMM launches (batches of a few sequences)
// stage three (256 x 128 tiles) / four (128 x 128) k-tiles deep instead of two (kernels_gemm_x3p.hip, NST).  Values do not depend on it.
void set_x3q_deep_stages(bool on);
unsigned* range_sink_word();   // engine.hip: 4 bytes of device memory per device, allocated on first use (nullptr if that failed)
inline unsigned* launch_range_word() { return tl_launch_ctx.range_word ? tl_launch_ctx.range_word : range_sink_word(); }

enum Epi { EPI_NONE = 0, EPI_GELU = 1, EPI_RESIDUAL = 2 };

// ---- F16X3 operand ("pair") layout ---------------------------------------------------------------------------------
// A [rows, K] matrix that feeds an F16X3 GEMM lives in HBM as rows of 2K fp16 (K % 32 == 0): for every 32-deep k-tile t
// the 32 hi values sit at [64 t, 64 t + 32) and the 32 lo values at [64 t + 32, 64 t + 64) -- one 128-byte cache line
// per row per k-tile, which is exactly what one GEMM k-tile stages (kernels_gemm_x3p.hip).  Element (r, c):
//   hi at r * 2K + pair_col(c),   lo at r * 2K + pair_col(c) + PAIR_LO.
constexpr int PAIR_LO = 32;
__host__ __device__ __forceinline__ size_t pair_col(int c) { return (size_t)((c >> 5) << 6) + (size_t)(c & 31); }
// "Accumulator order" variant of the pair layout (the hidden activation fc1 -> fc2 only): inside every 32-column group
// column 16 jj + 4 q + r sits in slot 8 q + 4 jj + r, so that the 8 values a lane of the C^T accumulator layout holds of a
// group (q = lane / 16) are one 16-byte piece.  Legal for a GEMM operand because the consumer's weight uses the same order.
__host__ __device__ __forceinline__ size_t pair_col_acc(int c) {
  const int w = c & 31;
  return (size_t)((c >> 5) << 6) + (size_t)(8 * ((w & 15) >> 2) + 4 * (w >> 4) + (w & 3));
}

// ---- F16X3 range guard -------------------------------------------------------------------------------------------------
// An fp16 plane holds 8*x (activations; the q third of a qkv output 1*x) or 2^k*w (weights; k per matrix, 12 unless a weight
// exceeds 15.99): |x| > 8188 does not fit (clamped to +-65504 by the row / attention kernels, inf behind the GEMM epilogues'
// v_fma_mix split).  Every device-side plane writer therefore tracks max |value| per lane and ORs a bit into the sticky word its
// launcher handed it (launch_range_word(): the launching ENGINE's word) when one left the range (one atomic per lane that saw one:
// none in a healthy run).  Bits of the word: 1 = a plane value left the range, 2 = a folded LayerNorm met a row with |mean| > 16 sigma,
// 4 = a timestep index of q_sample / the p_losses tail outside [0, num_timesteps) (not a precision matter: raised in every mode),
// 8 = the head kernel's duplicated dot product disagreed with itself (kernels_elem.hip k_head: repaired by a third evaluation).
constexpr float X3_HALF_MAX = 65504.0f;
constexpr unsigned RANGE_BIT_ACT = 1u, RANGE_BIT_STATS = 2u, RANGE_BIT_INDEX = 4u, RANGE_BIT_RECOMPUTE = 8u;
__device__ __forceinline__ void range_raise(unsigned* rw, unsigned bit) { if (rw) atomicOr(rw, bit); }

// ---- kernels_gemm.hip -------------------------------------------------------------------------------------------
// C[M,N] = epi(A[M,K] @ W[N,K]^T + bias[N]); all row-major fp32, K % 32 == 0.
hipError_t launch_linear_f32(const float* A, const float* W, const float* bias, const float* R, float* C, int M, int N,
                             int K, int epi, hipStream_t s);

// ---- kernels_gemm_f16x3.hip --------------------------------------------------------------------------------------
// Same contract, fp32-accurate product from 3 fp16 MFMAs; A split on the fly, W in the pair layout made by
// split_weight_f16x3() ([rows][2*cols] fp16 of 4096*w: this form takes k = 12 only).
hipError_t launch_linear_f16x3(const float* A, const void* Wpair, const float* bias, const float* R, float* C, int M, int N,
                               int K, int epi, hipStream_t s);
// Planes of 2^k * w, k chosen per matrix: 12 whenever |w| <= 15.99 everywhere, smaller for larger weights (LayerNorm-folded
// weights of checkpoints with big gains).  Returns k (the GEMM's w_exp argument); *clamped is set when an entry is not finite.
int split_weight_f16x3(const float* w, size_t rows, size_t cols, uint16_t* pair, bool acc_order = false, bool* clamped = nullptr);

// ---- kernels_gemm_x3p.hip ---------------------------------------------------------------------------------------
// F16X3 with pre-split operands in the pair layout: A (>= ceil(M/256)*256 rows allocated), W (>= ceil(N/256)*256 rows).
// outsplit: 0 = fp32 C; 1 = C as two [M][N] fp16 planes Ch/Cl of 8*c (read by the temporal attention kernel);
// 2 = C in the pair layout at Ch (operand of a following x3p GEMM; N % 32 == 0).  variant 0 = auto tile choice.
// qcols: with outsplit, columns < qcols carry 1*c instead of 8*c (q third of a temporal qkv GEMM).
//
// X3Fold (optional): the LayerNorm-folded / plane-resident forms used by the F16X3 engine path (engine.hip run_blocks):
//   st_in != null  : A holds the RAW rows x (pair layout of 8x), W holds W*diag(gamma), bias holds b + W beta, csum[n] =
//                    sum_k W[n,k] gamma[k]; the epilogue forms  LN(x) W^T + b = rstd (x W'^T) - rstd mu csum + b'  with the
//                    row's mean / rstd from st_in[(m * st_np + p)] = (sum, sum of squares) partials over the K columns
//                    (the buffer spans whole 256-row tiles: the persistent walk stages a tile's block of it by LDS-DMA)
//   Rp != null     : EPI_RESIDUAL takes the residual from a pair-layout plane buffer [M][2N] of 8r (instead of fp32 R);
//                    with outsplit == 2 and Ch == Rp the residual stream is updated in place, plane to plane
//   st_out != null : (with EPI_RESIDUAL) per row and 64-column block the (sum, sum of squares) of the new row values go to
//                    st_out[(m * (N/64) + block)] -- the st_in of the next folded GEMM (st_np = x3q_ntiles() = N/64)
//   pn.g != null   : (with EPI_RESIDUAL, Rp, N == 512) the tile spans whole rows, and the epilogue applies the block's post-norm
//                    to the new rows before they leave the chip:  y = LN(r + a W^T + b; pn.g, pn.b) [+ pos] [+ tvec]  (the
//                    operations of launch_layernorm); outsplit == 2: y -> pair planes Ch (may be Rp) and the (sum, sum of
//                    squares) partials of y -> st_out; outsplit == 0: y -> fp32 C
struct X3PostNorm {
  const float* g; const float* b; float eps;
  const float* pos; int pos_div, pos_mod;                          // nullable, as LnArgs
  const float* tvec; long long tvec_stride; int rows_per_batch;   // nullable, as LnArgs
  // bf16 mode only (launch_linear_bf16_rows): a SECOND LayerNorm of the rows just formed, written as bf16 operand rows -- the
  // norm2 behind proj (then g == nullptr: no post-norm, the rows are x + a W^T + b), the next block's norm1 behind fc2
  const float* g2; const float* b2; float eps2;
};
struct X3Fold {
  const float* st_in; int st_np; const float* csum; float eps;
  const void* Rp;
  float* st_out;
  X3PostNorm pn;
};
bool x3q_postnorm_ok(int N, int K);   // shapes the post-norm form exists for
// w_exp: the exponent k of the weight planes (split_weight_f16x3): the accumulators are un-scaled by 2^-(3 + w_exp)
hipError_t launch_linear_x3p(const void* Apair, const void* Wpair, const float* bias, const float* R, float* C, void* Ch,
                             void* Cl, int M, int N, int K, int epi, int outsplit, int qcols, int variant, hipStream_t s,
                             const X3Fold* fold = nullptr, int w_exp = 12);
int x3q_ntiles(int M, int N);   // statistics partials per row an st_out launch writes
hipError_t launch_split_x3(const float* x, void* pair, size_t rows, int cols, hipStream_t s);
hipError_t launch_unsplit_x3(const void* pair, float* x, size_t rows, int cols, const float* part, int np, float* stats,
                             hipStream_t s);   // op hooks only
// bf16 operand mode (D3D_PREC_BF16): A / W plain bf16 rows (K % 64 == 0, rows padded as above), one bf16 MFMA per product, fp32
// accumulate.  EPI_NONE / EPI_GELU write Cb = bf16 [M][N] (columns < qcols scaled by 2^-3), EPI_RESIDUAL writes fp32 C = R + ...
hipError_t launch_linear_bf16(const void* A, const void* W, const float* bias, const float* R, float* C, void* Cb, int M, int N,
                              int K, int epi, int qcols, hipStream_t s);
// Whole-row form (N == 512, 128-row tiles): X[M,N] fp32 is the residual stream, updated IN PLACE,
//   y = X + A W^T + bias;   if pn.g: y = LN(y; pn.g, pn.b, pn.eps) [+ pos] [+ tvec];   X = y;
//   if pn.g2: Hb = bf16(LN(y; pn.g2, pn.b2, pn.eps2))   (bf16 [M][N], the next GEMM's operand)
// -- proj + norm2 (pn.g == nullptr) and fc2 + post-norm + next norm1 of a bf16-mode block without any row kernel.
bool bf16_rows_ok(int N, int K);
hipError_t launch_linear_bf16_rows(const void* A, const void* W, const float* bias, float* X, void* Hb, int M, int N, int K,
                                   const X3PostNorm& pn, hipStream_t s);
// kernels_gemm_bf16q.hip: the plain bf16 GEMMs (qkv, fc1) on the hand-specialised two-phase k-loop; bit-identical to launch_linear_bf16's forms
bool gemm_bf16q_ok(int N, int K);
hipError_t launch_gemm_bf16q(const void* A, const void* W, const float* bias, void* Cb, int M, int N, int K, int epi, int qcols, hipStream_t s);
hipError_t launch_f32_to_bf16(const float* x, void* y, size_t n, hipStream_t s);
hipError_t launch_bf16_to_f32(const void* x, float* y, size_t n, hipStream_t s);
hipError_t launch_scale_cols(float* x, size_t rows, int cols, int ncols_scaled, float f, hipStream_t s);   // op hooks only
// diagnostic launches (variant 13): per (workgroup, wave) six u64 stamps {clk, 100 MHz} x {start, k-loop end, end}
void set_linear_x3_diag(unsigned long long* dev_buf);
void attn_x3_diag_report();   // -DD3D_ATTN_DIAG_BUILD builds only: prints the step stamps of the last staggered temporal-attention launch

// ---- kernels_qkv_sattn.hip: spatial blocks, qkv GEMM (LayerNorm-folded) + 17-key attention in one kernel --------------------
// Apair: the residual stream planes (rows up to 255 * ceil(frames / 15) + 1 are staged); W / bias / csum HEAD-MAJOR (row 192 h +
// 64 part + d); st_in / st_np / eps as X3Fold; out_x3: attention output in the pair layout.  M = frames * J tokens.
bool qkv_sattn_ok(int J, int D, int H, int K);
void set_qkv_sattn_diag(int on);   // "qs_diag" option: in-kernel stamp report of every 50th launch on stderr
hipError_t launch_qkv_sattn(const void* Apair, const void* Wpair_headmajor, const float* bias_hm, const float* csum_hm, const float* st_in,
                            int st_np, float eps, int w_exp, void* out_x3, int M, int K, int J, int D, int H, hipStream_t s);
// kernels_qkv_tattn.hip: the temporal counterpart -- the LayerNorm-folded qkv GEMM of one (batch, joint) group (T in 193..255 frames; T <= 127:
// of 255 / T joints of one batch element, per-joint key windows) x one head with the T-key attention of k_attn_temporal_x3s run from LDS; the same tile-ordered weight / bias / csum as launch_qkv_sattn.
// kernels_fc1_x3.hip: fc1 (LayerNorm-folded, GELU, accumulator-order pair output) on the hand-specialised k-loop of the fused kernels,
// whole 256 x 256 tiles only (buffers padded to 256 rows, finite pad rows); bit-identical to launch_linear_x3p's form.
// kernels_proj_x3.hip: proj (plane residual in place + row statistics) likewise, whole 192 x 256 tiles (M % 192 == 0: the caller runs the
// remaining rows through launch_linear_x3p); bit-identical to launch_linear_x3p's form.
bool proj_x3_ok(int N, int K);
hipError_t launch_proj_x3(const void* Apair, const void* Wpair, const float* bias, void* Xpair, float* st_out, int w_exp, int M, int N, int K,
                          hipStream_t s);
// kernels_fc2_ring.hip: fc2 + post-norm (launch_linear_x3p's X3Fold Rp + pn form, outsplit 2 / 0) on a k-loop without workgroup barriers
// (wave-private W slots, A through a four-slot ring with per-slot arrival counters in LDS), 128 x 512 whole-row tiles, the ragged last
// tile included; bit-identical to that form.  delay: waves 4-7 start each tile that many x 64 cycles late (default 24).
bool fc2_ring_ok(int N, int K);
void set_fc2_ring_delay(int d);
void set_fc2_ring_dbg(int d);
void set_fc2_ring_diag(unsigned long long* dev_buf);   // nullable; 8 u64 per workgroup
hipError_t launch_fc2_ring(const void* Apair, const void* Wpair, const float* bias, float* C, void* Ch, int M, int N, int K, int outsplit,
                           const X3Fold* fold, int w_exp, hipStream_t s);
bool fc1_x3_ok(int N, int K);
hipError_t launch_fc1_x3(const void* Apair, const void* Wpair, const float* bias_f, const float* csum, const float* st_in, int st_np,
                         float eps, int w_exp, void* out_pair, int M, int N, int K, hipStream_t s);
bool qkv_tattn_ok(int T, int J, int D, int H, int K);
void set_qkv_tattn_diag(int on);   // "qt_diag": in-kernel stamp report of every 50th launch
hipError_t launch_qkv_tattn(const void* Apair, const void* Wpair_tileorder, const float* bias_to, const float* csum_to, const float* st_in,
                            int st_np, float eps, int w_exp, void* out_x3, int B, int T, int J, int K, int D, int H, hipStream_t s);

// ---- kernels_elem.hip -------------------------------------------------------------------------------------------
// Row LayerNorm; optionally also writes a second LayerNorm of the first result (post-norm -> next block's norm1).
//   y  = LN(x; g1,b1,eps1) [+ pos[(row / pos_div) % pos_mod] ] [+ tvec[(row / rows_per_batch) * tvec_stride]]
//   h  = LN(y; g2,b2,eps2)                       (only when h != nullptr)
struct LnArgs {
  const float* x;
  float* y;          // may alias x; may be nullptr when only h is wanted (then h = LN(x; g1,b1))
  float* h;          // nullable
  void* h_x3;        // nullable: when set, h is written in the F16X3 pair layout (8*h, D % 32 == 0) instead of fp32
                     //           (h itself may then be nullptr)
  void* h_bf16;      // nullable: when set, h is written as plain bf16 rows (operand of a bf16-mode GEMM) instead of fp32
  const float* g1; const float* b1; float eps1;
  const float* g2; const float* b2; float eps2;
  const float* pos;  // nullable, (pos_mod, D)
  int pos_div, pos_mod;
  int skip_ln1;      // 1: y = x (+ pos + tvec) without the first normalisation (entry of block 0)
  void* y_x3;        // nullable: y also / instead goes out in the F16X3 pair layout (8*y): the plane-resident residual stream
  float* stats;      // nullable: (sum, sum of squares) of every y row, [rows][2] -- consumed by an LN-folded GEMM (X3Fold)
  const float* tvec; // nullable, per-batch vector (n, D) with row stride tvec_stride (0 = broadcast)
  int64_t tvec_stride;
  int rows_per_batch;
  int rows, D;
  unsigned* range;   // filled in by launch_layernorm: the range-guard word the plane outputs report to
};
hipError_t launch_layernorm(const LnArgs& a, hipStream_t s);

// X[m,:] = W_f [x2d[m], y[m']] + b_f + spos[j] + tvec[b]   (S2S:250, :229-233, :113-116 of block 0)
hipError_t launch_embed(const float* x2d, const float* y, const float* Wf, const float* bf, const float* spos,
                        const float* tvec, int64_t tvec_stride, float* X, int B, int T, int J, int D, int in_chans,
                        int y_bcast_T, hipStream_t s);

// the same, written as the F16X3 residual-stream planes (pair layout of 8 X) + per-row (sum, sum of squares): embed and the
// stream-entry row kernel in one pass (D == 512)
bool embed_planes_ok(int D, int in_chans);
hipError_t launch_embed_planes(const float* x2d, const float* y, const float* Wf, const float* bf, const float* spos,
                               const float* tvec, int64_t tvec_stride, void* XP, float* stats, int B, int T, int J, int D,
                               int in_chans, int y_bcast_T, hipStream_t s);

// sinusoid + trunk + per-block projections: out (n, nblk, D)
hipError_t launch_sinusoid(const float* times, const float* freqs, float* out, int n, int D, hipStream_t s);
// out[n,N] = post(act_pre(in[n,K]) @ W[N,K]^T + b); act: 0 none, 1 gelu(post), 2 silu(pre)
hipError_t launch_small_linear(const float* in, const float* W, const float* b, float* out, int n, int N, int K,
                               int act, hipStream_t s);

// seq2frame frame reduce (S2F:261-263): out[b,j,:] = sum_t w[t] X[b,t,j,:] + bias
hipError_t launch_frame_reduce(const float* X, const float* w, const float* bias, float* out, int B, int T, int J, int D,
                               hipStream_t s);

struct HeadArgs {
  const float* X;        // (rows, D)
  const float* g; const float* b; float eps;   // head.0
  const float* Wh; const float* bh;            // head.1 (3, D), (3,)
  int rows, D;
  // outputs
  float* x0_raw;         // nullable: raw network output (rows,3)
  // DDIM step (all nullable when mode == 0)
  int mode;              // 0: raw only; 1: clamp?/DDIM update; 2: final step (y_out = clamped x0)
  int clip;
  const float* y_cur;    // (rows,3)
  float* y_next;         // (rows,3) (may alias y_cur)
  const float* noise;    // nullable (rows,3)
  float alpha, alpha_next, somac, eta;
  float* traj_rev; float* traj_x0; int traj_rev_stride, traj_x0_stride, traj_idx;   // nullable
  unsigned* range;       // filled in by launch_head: the launching engine's guard word (RANGE_BIT_RECOMPUTE)
  int fence;             // "head_fence" option: every row's dot products evaluated twice and compared (RANGE_BIT_RECOMPUTE)
  int inject;            // tests only ("head_inject" option; implies the fence): perturb the first evaluation of row 0
};
hipError_t launch_head(const HeadArgs& a, hipStream_t s);

hipError_t launch_range_snapshot(unsigned* word, unsigned* host_slot_dev, hipStream_t s);   // d3d_engine_range_post
hipError_t launch_q_sample(const float* x_start, const float* noise, const int32_t* t, const float* sqrt_ac,
                           const float* somac, float* out, int B, int64_t n, int nt, hipStream_t s);

// p_losses tail (DIFF:411-418) and the repeat_n tiling / hypothesis mean of forward() (DIFF:433-448); n = elements per batch row
hipError_t launch_weighted_loss(const float* model_out, const float* target, const int32_t* t, const float* ac, const float* somac,
                                float* out, int B, int64_t n, int l2, int clip, int nt, hipStream_t s);
hipError_t launch_repeat_rows(const float* x, float* out, int B, int64_t n, int R, hipStream_t s);
hipError_t launch_hypothesis_mean(const float* pred, float* out, int B, int64_t n, int R, hipStream_t s);

// joint permutation of the horizontal flip, passed by value as a kernel argument (no device allocation, no copy, no sync)
struct JointPerm { static constexpr int MAXJ = 64; int32_t p[MAXJ]; };
hipError_t launch_tta_mpjpe(const float* pred, const float* pred_flip, const float* gt, const uint8_t* mask, float scale,
                            const JointPerm& perm, float* merged, double* sums, int B, int T, int J, hipStream_t s);
// evaluate()'s other three protocols on the merged prediction (kernels_elem.hip k_pose_metrics; RUN:602-614, LOSS:43-93, 132-142)
hipError_t launch_pose_metrics(const float* pred, const float* gt, const uint8_t* mask, double* sums, int N, int J, hipStream_t s);

hipError_t launch_window_gather(const float* seq, float* out, uint8_t* mask, const JointPerm& perm, int n, int T, int J, int C,
                                int flip, hipStream_t s);
hipError_t launch_window_gather_s2f(const float* seq, float* out, const JointPerm& perm, int n, int T, int J, int C, int first, int count,
                                    int flip, hipStream_t s);

// debug trace: *out += position-weighted 64-bit sum of the buffer's 32-bit words (order-independent)
hipError_t launch_checksum(const void* p, size_t bytes, unsigned long long* out, int rot, hipStream_t s);
// probes.hip: what the chip sustains (what 0: fp16 MFMA TFLOP/s in register loops; 1: L2 -> LDS staging GB/s chip-wide)
hipError_t launch_probe_machine(int what, float ms_target, float* result, hipStream_t s);

// ---- kernels_attn.hip -------------------------------------------------------------------------------------------
// qkv: (B*T*J, 3*D) -> out (B*T*J, D), GRAND core  O = softmax(q k^T * dh^-0.5) v - v
// When out_x3 is non-null the result is written in the F16X3 pair layout (8*o, operand of the proj GEMM; D % 32 == 0)
// and `out` is ignored.
hipError_t launch_attn_spatial_f32(const float* qkv, float* out, void* out_x3, int B, int T, int J, int D, int H, hipStream_t s);
hipError_t launch_attn_temporal_f32(const float* qkv, float* out, void* out_x3, int B, int T, int J, int D, int H, hipStream_t s);
hipError_t launch_attn_generic(const float* qkv, float* out, void* out_x3, int B, int T, int J, int D, int H, int temporal,
                               hipStream_t s);
// ---- kernels_attn_x3.hip: temporal attention on fp16 MFMA with F16X3 accuracy; qkv as hi/lo planes ([rows][3D] each,
// written by the qkv GEMM with outsplit = 1), output in the pair layout
hipError_t launch_attn_temporal_x3(const void* qkv_hi, const void* qkv_lo, void* out_x3, int B, int T, int J, int D, int H,
                                   hipStream_t s);
bool attn_temporal_x3_ok(int T, int D, int H);
hipError_t launch_split_qkv(const float* x, void* hi, void* lo, size_t rows, int D, hipStream_t s);
hipError_t launch_unsplit_pair(const void* pair, float* x, size_t rows, int cols, hipStream_t s);
bool attn_spatial_fast_ok(int J, int D, int H);
// ---- kernels_attn_bf16.hip: the same core on v_mfma_f32_32x32x16_bf16, one MFMA per product (D3D_PREC_BF16).  qkv: ONE bf16
// buffer [rows][3D] (q third pre-scaled by dh^-0.5, written by launch_linear_bf16 with qcols = D); out: bf16 [rows][D].
// Spatial blocks: call with (B * T, J, 1).  softmax - I is normalised and rounded to bf16 before the second product.
hipError_t launch_attn_bf16(const void* qkv_bf16, void* out_bf16, int B, int T, int J, int D, int H, hipStream_t s);
bool attn_bf16_ok(int T, int D, int H);
bool attn_temporal_fast_ok(int T, int D, int H);

}  // namespace d3d
