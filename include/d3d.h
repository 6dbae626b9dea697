/*
 * d3d.h -- C ABI of libd3d_hip.so, the MI355X (gfx950) engine for Diff3DHPE's DDIM sampling hot path.
 *
 * The reference (csiro-icvg/Diff3DHPE) is pure Python and has no FFI/plugin interface; its boundary for this
 * path is the Python object protocol of two classes.  Each entry point below names the reference code it
 * replaces (paths relative to the reference root):
 *   DIFF = common/conditional_diffusion_ddim_normal_directPredict_variableLoss_both_crossFrames.py
 *   DIFF-S2F = common/conditional_diffusion_s2f_ddim_normal_directPredict_variableLoss_both_crossFrames.py
 *   S2S  = common/nets/model_conditional_diffusion_mixste_s2s_grand_linLift.py
 *   S2F  = common/nets/model_conditional_diffusion_mixste_s2f_grand_linLift.py
 *   LOSS = common/loss.py,  RUN = run_conditionalDiffusionDDIM3dhpeNormalDirectPredictVariableLoss.py
 *
 * Conventions
 *   - plain C types only; every pointer marked "dev" is a device (HBM) pointer owned by the caller;
 *     "host" pointers are ordinary host memory.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - every function returns 0 on success, a negative D3D_E* code on failure; d3d_last_error() gives the message
 *     (thread-local).  No exception crosses the ABI.  Nothing here falls back to a CPU implementation: without a
 *     HIP device the compute entry points fail with D3D_EHIP.
 *   - one engine per device; an engine is not thread-safe, distinct engines are independent.
 *   - tensors are row-major fp32: x2d (B,T,J,in_chans), y / x0 / out (B,T,J,3) [(B,1,J,3) for seq2frame].
 */
#ifndef D3D_H_
#define D3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D3D_OK 0
#define D3D_EINVAL (-1)   /* bad argument / shape                        */
#define D3D_ESTATE (-2)   /* call order (weights not committed, ...)     */
#define D3D_EHIP (-3)     /* HIP runtime error or no device              */
#define D3D_ENOMEM (-4)   /* workspace too small                         */
#define D3D_EUNSUP (-5)   /* configuration outside what the kernels support */

#define D3D_PREC_FP32 0     /* exact fp32: v_mfma_f32_32x32x2_f32, fp32 everywhere (parity mode) */
#define D3D_PREC_F16X3 1    /* fp32-accurate GEMMs and attention from 3 fp16 MFMAs per product on hi/lo operand splits; the
                             * residual stream lives in the GEMM operand layout and norm1/norm2 are folded into the qkv/fc1
                             * GEMMs.  Same 1e-4 parity gate as FP32 INSIDE its operand range (range guard below); the engine the
                             * Python layer's default precision "auto" starts on -- it reads the guard after every call and repeats a flagged
                             * call on a D3D_PREC_FP32 engine. */
#define D3D_PREC_BF16 2     /* SECOND-CLASS precision (BASELINE configs[1]): bf16 operands for the block GEMMs
                             * and both attention products (one MFMA per product), everything else fp32 (residual stream, LayerNorm /
                             * softmax statistics, GELU, time vectors, embedding, head, DDIM update).  It CANNOT meet the 1e-4 parity
                             * gate (bf16 operands cost ~5e-2 max-abs at random init); it is gated against the CPU
                             * oracle's bf16-operand emulation instead (same rounding points).  Needs head_dim 64, num_frame <= 256,
                             * num_joints <= 32, widths % 64 == 0; never the default of the Python layer or of bench.py. */

typedef struct d3d_engine d3d_engine;

/* Shape-defining constructor arguments of the denoiser (S2S:140-142 / S2F ctor; runner passes them at RUN:178-180). */
typedef struct d3d_config {
  int32_t num_frame;     /* T  */
  int32_t num_joints;    /* J  */
  int32_t in_chans;      /* 2  */
  int32_t embed_dim;     /* D  */
  int32_t depth;         /* number of (spatial, temporal) block pairs */
  int32_t num_heads;     /* H  */
  int32_t mlp_hidden;    /* int(D * mlp_ratio) */
  int32_t with_time_emb; /* 0/1: 0 removes every time_mlp (S2S:163-177,104-107) */
  int32_t seq2frame;     /* 0 = ...S2S..., 1 = ...S2F... (LOADNET:5-10) */
  int32_t precision;     /* D3D_PREC_* */
} d3d_config;

const char* d3d_last_error(void);
int d3d_version(void);

/* ---- construction: replaces HPE_model(name)(**kw) + GaussianDiffusion(model=...) (RUN:176-189) ------------------- */
int d3d_engine_create(const d3d_config* cfg, d3d_engine** out);
void d3d_engine_destroy(d3d_engine* e);

/* Number of weight tensors the engine expects, and the i-th expected name / element count (reference state_dict keys
 * without the "model." prefix, e.g. "STEblocks.3.attn.qkv.weight"; S2S:160-220, S2F:216-218). */
int d3d_engine_num_weights(const d3d_engine* e);
int d3d_engine_weight_info(const d3d_engine* e, int i, const char** name, int64_t* numel);

/* Copy one fp32 tensor (host memory, torch layout) into the engine; replaces load_state_dict (RUN:226-235). */
int d3d_engine_set_weight(d3d_engine* e, const char* name, const float* host, int64_t numel);
/* Optional: the (D/2,) frequency table of SinusoidalPosEmb (S2S:29-36) as the host framework computes it; without it
 * the engine uses exp() in double rounded to fp32 (differs from torch's expf in ~1 of 256 entries by 1 ulp). */
int d3d_engine_set_time_freqs(d3d_engine* e, const float* host, int32_t n);
/* Verify every tensor was supplied, upload + repack to kernel layouts.  Must precede any compute call. */
int d3d_engine_commit_weights(d3d_engine* e);

/* Diffusion constants: fp32 buffers `alphas_cumprod` and `sqrt_one_minus_alphas_cumprod` (num_timesteps,) as registered
 * at DIFF:151-161, plus sampling_timesteps / ddim_sampling_eta / clip_denoised (DIFF:100-112).  Builds the integer DDIM
 * schedule (DIFF:270-273) and the per-step time-embedding table (S2S:169-174,104-107).  Weights must be committed. */
int d3d_engine_set_schedule(d3d_engine* e, int32_t num_timesteps, const float* alphas_cumprod_host,
                            const float* sqrt_one_minus_alphas_cumprod_host, int32_t sampling_timesteps, float eta,
                            int32_t clip_denoised, void* stream);

/* Optional fp32 buffer `sqrt_alphas_cumprod` (DIFF:155) -- needed only by d3d_q_sample.  Call after set_schedule. */
int d3d_engine_set_sqrt_alphas_cumprod(d3d_engine* e, const float* host, int32_t n);

/* Host-only, bit-exact restatement of `torch.linspace(-1, N-1, S+1).int()` reversed (DIFF:270-272): writes S+1 values. */
int d3d_ddim_times(int32_t num_timesteps, int32_t sampling_timesteps, int32_t* out);

/* Bytes of device scratch the compute calls need for a batch of B sequences. */
size_t d3d_workspace_bytes(const d3d_engine* e, int32_t B);

/* ---- compute: all asynchronous on `stream` ------------------------------------------------------------------------ */
/* B >= 1 (D3D_EINVAL otherwise): a rank whose shard of a batch is empty skips the call, as the host layer does
 * (diff3dhpe_amd/engine.py returns the empty tensor the reference's torch ops would, evaluate.py does not call). */

/* forward_denoise (S2S:249-257 / S2F:253-266) on cat([x2d, y], -1) (DIFF:255).  times_dev: n_times fp32 timesteps on
 * the device, n_times == 1 (broadcast, the sampling case DIFF:254) or == B (per-row, the p_losses case DIFF:392-408).
 * Ignored when with_time_emb == 0.  y is (B,y_frames,J,3) with y_frames == T, or == 1 to have the kernel broadcast one
 * frame over T (the seq2frame repeat of DIFF-S2F:281).  x0 receives the raw network output (no clamp):
 * (B,T,J,3), or (B,1,J,3) for a seq2frame engine. */
int d3d_denoise(d3d_engine* e, const float* x2d_dev, const float* y_dev, int32_t y_frames, const float* times_dev,
                int32_t n_times, float* x0_dev, int32_t B, void* ws_dev, size_t ws_bytes, void* stream);

/* ddim_sample_loop (DIFF:262-300; DIFF-S2F:263-300).  init_noise replaces torch.randn(target_shape) (DIFF:275);
 * step_noise_dev (nullable; required when eta != 0) is (S, B,T',J,3): the per-step randn_like draws (DIFF:293).
 * traj_rev_dev / traj_x0_dev (nullable) receive x_reverse_diffusion / x_start_est stacked on the last axis,
 * shape (B,T',J,3,S) (DIFF:303-347).  out: (B,T',J,3); T' = 1 for seq2frame else T. */
int d3d_ddim_sample(d3d_engine* e, const float* x2d_dev, const float* init_noise_dev, const float* step_noise_dev,
                    float* out_dev, float* traj_rev_dev, float* traj_x0_dev, int32_t B, void* ws_dev, size_t ws_bytes,
                    void* stream);

/* hipGraph replay of d3d_ddim_sample: when enabled (and eta == 0, no trajectory capture, profiling off) the whole S-step
 * launch sequence is captured once per (B, workspace pointer) and replayed with one hipGraphLaunch per call; inputs and
 * the result go through staging buffers inside the workspace.  Weights / schedule changes drop the captured graphs.  At most
 * FOUR captured graphs are kept per engine: a fifth (B, workspace) pair evicts the least recently used one (a caller that
 * re-allocates its workspace per batch re-captures every time, but does not accumulate graphs). */
int d3d_engine_set_graph_mode(d3d_engine* e, int32_t on);

/* Explicit switches (the library reads NO environment variables).  Engine options (results stay within the parity gate either way;
 * used by the A/B scripts under experiments/ and by tests that exercise the alternative flows):
 *   "fused_postnorm"  1 (default) / 0: F16X3 block flow with the block's post-norm inside the fc2 GEMM epilogue / as a row kernel
 *   "fold_layernorm"  1 (default) / 0: F16X3 flow with norm1 / norm2 folded into the qkv / fc1 GEMMs / as row kernels
 *   "fused_spatial"   1 (default) / 0: F16X3 flow, spatial blocks (17 joints of a frame, D = 512, 8 heads): the qkv GEMM of a group
 *                     of 15 frames keeps q / k / v in LDS and runs the frames' attention in the same kernel (S2S:67 + 73-83; the
 *                     q / k / v planes never go to HBM) / qkv GEMM and attention as two kernels.  Bit-identical either way.
 *   "fused_temporal"  1 (default) / 0: the same for the temporal blocks where the frames of a joint fit one 256-row tile (193 <= T <= 255,
 *                     D = 512, 8 heads): the qkv GEMM of one (batch, joint) group keeps K / V in LDS, exchanges the queries there and runs
 *                     the group's T-key attention in the same kernel (S2S:67 + 73-83 for the per-joint groups) / two kernels.  Bit-identical.
 *                     T <= 127: the frames of 255 / T joints of one batch element per tile, every query on its own joint's keys -- as
 *                     accurate as the two-kernel flow but not bit-identical to it (sums grouped by tile position); a batch element's
 *                     result does not depend on its position in the batch either way.
 *   "fc1_kernel"      1 (default) / 0: fc1 (LayerNorm-folded, GELU) on its own kernel -- the hand-specialised k-loop of the fused kernels
 *                     with the token GEMM's own epilogue function, from two rounds of 256 x 256 tiles on -- / as a form of the token GEMM.
 *                     Bit-identical.
 *   "proj_kernel"     1 (default) / 0: the same for proj (192 x 256 tiles; rows behind the last whole tile through the token GEMM).
 *   "bf16_gemm_kernel" 1 (default) / 0: BF16 mode, qkv and fc1 on their own kernel (the hand-specialised two-phase k-loop with one bf16
 *                     MFMA per fragment pair, from two rounds of 256 x 256 tiles on) / as forms of the token GEMM.  Bit-identical.
 *   "head_fence"      0 (default) / 1: the head kernel evaluates every row's three dot products twice from independently loaded weight
 *                     fragments, compares them bit for bit, repairs a disagreement by a third evaluation and raises D3D_RANGE_RECOMPUTE
 *                     (round 5's default against a deviation seen on a GPU shared by two processes; since round 6 that deviation is
 *                     identified -- one packed fp32 instruction form beside another wave's MFMAs -- and pinned out of every kernel at
 *                     build time, so the second evaluation is optional: +0.08 ms per launch of 0.21)
 *   "head_inject"     0 (default) / 1, tests only (implies "head_fence"): the head kernel perturbs the first of its two evaluations of
 *                     row 0 -- the fence must repair the row (result unchanged) and raise D3D_RANGE_RECOMPUTE
 *   "fc2_ring"        0 (default) / 1: fc2 + post-norm on the k-loop without workgroup barriers (kernels_fc2_ring.hip: wave-private W
 *                     slots, A through a ring with arrival counters in LDS); bit-identical; measured level with the token GEMM's form
 *   "norm_eps_bits"   the bit pattern of the float eps of the constructor's norm_layer (S2S:184; default 1e-6): norm1 / norm2 of every block
 *                     and the two post-norms; the head's LayerNorm keeps 1e-5 (S2S:218).  The Python classes set it from
 *                     norm_layer=partial(nn.LayerNorm, eps=...)
 *   "streams"         2 (default) / 1: d3d_ddim_sample runs a batch of B >= 2 as two half-batches on two HIP streams (the caller's and
 *                     one the engine owns, forked and joined by events: the caller sees ONE asynchronous operation on its stream;
 *                     bit-identical to one stream -- every output element is independent of the batch it is computed in; measured
 *                     +2.9 % at T=243 / B=64, neutral at T=81 / T=27).  Per-kernel profiling and the trace force one stream.
 * Process-wide switch (e may be NULL): "deep_stages" 1 (default) / 0: the one-tile-per-workgroup F16X3 GEMM launches (batches of a few
 * sequences: proj on 128 x 128 tiles, qkv / proj / fc1 on 256 x 128) keep three / four k-tiles of operands staged instead of two, the wait
 * in front of a k-tile's barrier a counted vmcnt -- a k-tile no longer lasts a DMA round trip (proj at B = 1, T = 243: 20.8 -> 17.6 us per
 * launch; a 9-step sampling 18.9 -> 18.3 ms at B = 1, 75.7 -> 72.2 at B = 8).  Bit-identical.
 * Process-wide diagnostics (e may be NULL): "gemm_diag", "attn_diag" 0 / 1: the op hooks print in-kernel stamp reports to stderr;
 * "fc2_ring_delay" (waves 4-7 of the ring kernel start each tile that many x 64 cycles late; default 24), "fc2_ring_op" 0 / 1
 * (d3d_op_linear_postnorm through the ring kernel), "fc2_ring_dbg" / "fc2_ring_diag" (builds with -DR2_DEBUG only: experiments/fc2_ring_op.py)
 * (attn_diag needs a -DD3D_ATTN_DIAG_BUILD library); "qs_diag" / "qt_diag" 0 / 1: every 50th launch of the fused spatial / fused temporal kernel
 * runs with per-step stamps and prints their summary to stderr (that launch synchronises its stream; launches inside a hipGraph capture
 * are never stamped).  The diagnostic switches are PROCESS-wide: they act on every engine.  Unknown key: D3D_EINVAL. */
int d3d_engine_set_option(d3d_engine* e, const char* key, int64_t value);

/* q_sample (DIFF:360-366, extract DIFF:21-24): out = sqrt_ac[t_b] * x_start + sqrt(1-ac)[t_b] * noise, per row b.
 * n = elements per batch row. */
int d3d_q_sample(d3d_engine* e, const float* x_start_dev, const float* noise_dev, const int32_t* t_dev, float* out_dev,
                 int32_t B, int64_t n, void* stream);

/* p_losses tail (DIFF:411-418): out = loss_fn(model_out, target, reduction='none') * loss_coef[b],
 * loss_coef[b] = 1 + alphas_cumprod[t_b] / sqrt_one_minus_alphas_cumprod[t_b], clamped to <= 3 when clip_loss (clipLoss, DIFF:111).
 * loss_type 1 = l1, 2 = l2 (DIFF:368-375).  model_out / target / out: B rows of n fp32 values; t_dev: B int32 timesteps. */
int d3d_weighted_loss(d3d_engine* e, const float* model_out_dev, const float* target_dev, const int32_t* t_dev, float* out_dev,
                      int32_t B, int64_t n, int32_t loss_type, int32_t clip_loss, void* stream);

/* repeat_n hypotheses of forward() (DIFF:433-448): d3d_repeat_batch writes out[r * B + b, :] = x[b, :] for r < repeat_n
 * (noisy_2d_pose.repeat(repeat_n, 1, 1, 1)); d3d_hypothesis_mean writes out[b, :] = (pred[b, :] + pred[B + b, :] + ...) / repeat_n
 * (torch.mean(pred.view(repeat_n, b, f, p, -1), dim=0)) -- in the operation order of the reference's CPU path: the hypotheses added in
 * order, ONE fp32 division by repeat_n.  (torch.mean on a GPU tensor multiplies the sum by a precomputed 1 / repeat_n instead: one ulp apart
 * for repeat_n not a power of two; parity is defined against the CPU path.)  n = fp32 values per batch row. */
int d3d_repeat_batch(const float* x_dev, float* out_dev, int32_t B, int64_t n, int32_t repeat_n, void* stream);
int d3d_hypothesis_mean(const float* pred_dev, float* out_dev, int32_t B, int64_t n, int32_t repeat_n, void* stream);

/* Read-only engine facts: "graphs_cached" (captured hipGraphs held now, <= 4), "graphs_captured" (captures since creation),
 * "streams", "device".  Unknown key: D3D_EINVAL. */
int d3d_engine_get_info(const d3d_engine* e, const char* key, int64_t* value);

/* evaluate() tail (RUN:583-590, LOSS:15-22): un-flip + average the TTA pair, multiply by scale, and reduce the masked
 * per-joint L2 error.  sums_dev[0] += sum of joint errors over frames with mask != 0, sums_dev[1] += number of such
 * joints (caller zeroes sums_dev).  merged_dev (nullable) receives the merged, de-normalised prediction (B,T,J,3).
 * pred_flip_dev may be NULL (no TTA).  joints_left/right: host index lists of equal length.
 * Asynchronous on `stream` (since ABI version 110; it used to synchronise): sums_dev / merged_dev are valid only after the stream
 * has been synchronised.  J <= 64 (the joint permutation travels as a kernel argument); more: D3D_EUNSUP. */
int d3d_tta_mpjpe(const float* pred_dev, const float* pred_flip_dev, const float* gt_dev, const uint8_t* mask_dev,
                  float scale, const int32_t* joints_left_host, const int32_t* joints_right_host, int32_t n_lr,
                  float* merged_dev, double* sums_dev, int32_t B, int32_t T, int32_t J, void* stream);

/* evaluate()'s other three protocols (RUN:602-614) on the merged, de-normalised prediction d3d_tta_mpjpe wrote (pred_dev, gt_dev:
 * N frames of J joints x 3 fp32; mask_dev: N bytes, nullable = every frame kept), one thread per kept frame, fp64 inside:
 *   sums_dev[0] += sum over kept frames and joints of |s p - g|, s = <g, p> / <p, p> per frame          (N-MPJPE, LOSS:83-93)
 *   sums_dev[1] += the same for the prediction after the best similarity transform onto its target      (P-MPJPE, LOSS:43-81: scale,
 *                  rotation without reflection, translation; the 3 x 3 SVD by a Jacobi eigen-decomposition)
 *   sums_dev[2] += sum over kept frames WITH a kept predecessor in the flattened batch, and joints, of
 *                  |(p_f - p_prev) - (g_f - g_prev)|                                                      (MPJVE, LOSS:132-142: np.diff
 *                  of the masked batch -- window and sequence boundaries included, as there)
 *   sums_dev[3] += kept frames, sums_dev[4] += frames with a predecessor (caller zeroes the five doubles).
 * A batch's protocol value is sums[k] / (J * sums[3]) (k = 0, 1) and sums[2] / (J * sums[4]) -- 0 / 0 = nan for a batch of one kept
 * frame, as numpy's mean of an empty difference there; evaluate() weights each by the batch's kept frames.  Asynchronous on `stream`.
 * J <= 64; more: D3D_EUNSUP. */
int d3d_pose_metrics(const float* pred_dev, const float* gt_dev, const uint8_t* mask_dev, double* sums_dev, int32_t N, int32_t J,
                     void* stream);

/* The path's one exchange step (RUN:216-218: nn.DataParallel's gather of the replicas' outputs) for hosts that do not go through
 * torch.distributed: all-gather of count_per_rank fp32 values (the rank's predicted sequences) over an RCCL communicator the
 * CALLER owns (nccl_comm: ncclComm_t), asynchronous on `stream`; recv_dev holds world_size * count_per_rank values in rank
 * order.  libd3d_hip.so does not link RCCL: ncclAllGather is resolved at run time from the RCCL library already loaded into the
 * process (D3D_EUNSUP if there is none).  The Python host uses torch.distributed instead (parallel.all_gather_pred). */
int d3d_allgather_pred(void* nccl_comm, const float* send_dev, float* recv_dev, int64_t count_per_rank, void* stream);

/* Evaluation windows of one whole sequence, on the device: ChunkedGenerator(out_all=True, pad=0) (common/nosiy_generators.py:27-48
 * window table, :247-276 slicing, edge padding, target_mask, horizontal flip).  seq (n_frames, J, C) -> out
 * (d3d_num_windows, T, J, C); mask (nullable) (windows, T) uint8, 0 for the frames of the shifted last window that its
 * predecessor already covers.  flip != 0 writes the flipped copy (channel 0 negated, left/right joints swapped).
 * Asynchronous on `stream`: out_dev / mask_dev are valid only after the stream has been synchronised.  J <= 64, else D3D_EUNSUP. */
int d3d_num_windows(int32_t n_frames, int32_t T);
int d3d_window_gather(const float* seq_dev, int32_t n_frames, int32_t T, int32_t J, int32_t C, int32_t flip,
                      const int32_t* joints_left_host, const int32_t* joints_right_host, int32_t n_lr, float* out_dev,
                      uint8_t* mask_dev, void* stream);

/* The seq2frame window table (BASELINE configs[4]: ...S2F... models): ChunkedGenerator / ChunkedGenerator_3dhp with out_all=False and
 * chunk_length = stride = 1 (common/nosiy_generators.py:402-420 pair table, :492-512 slicing; data/load_noisy_data.py:312-316
 * pad = (T - 1) // 2): ONE window per target frame f, holding frames f - pad .. f + pad, edge-replicated at both ends of the sequence.
 * Writes the windows of target frames first .. first + count - 1: out (count, T, J, C).  The window's 3D target is frame f of the
 * 3D sequence itself and its mask the frame's `valid` flag: neither needs a kernel.  T must be odd; flip as in d3d_window_gather.
 * Asynchronous on `stream`. */
int d3d_window_gather_s2f(const float* seq_dev, int32_t n_frames, int32_t T, int32_t J, int32_t C, int32_t flip,
                          const int32_t* joints_left_host, const int32_t* joints_right_host, int32_t n_lr, int32_t first, int32_t count,
                          float* out_dev, void* stream);

/* ---- per-kernel-class timing (HIP events recorded on the launch stream around every kernel of d3d_denoise /
 * d3d_ddim_sample while enabled; used by bench.py for the roofline figures).  flops / bytes are the ALGORITHMIC counts
 * of the launches timed, total_ms the sum of their event-pair durations. ------------------------ */
#define D3D_KC_LINEAR 0
#define D3D_KC_ATTN_SPATIAL 1
#define D3D_KC_ATTN_TEMPORAL 2
#define D3D_KC_LAYERNORM 3
#define D3D_KC_EMBED 4
#define D3D_KC_HEAD 5
#define D3D_KC_OTHER 6
/* the GEMM launches of the F16X3 block flow once more, by kind (each is ALSO counted under D3D_KC_LINEAR: do not add these to it) */
#define D3D_KC_LINEAR_QKV 7
#define D3D_KC_LINEAR_PROJ 8
#define D3D_KC_LINEAR_FC1 9
#define D3D_KC_LINEAR_FC2 10
/* spatial blocks of the F16X3 flow: the LayerNorm-folded qkv GEMM and the 17-key attention as ONE kernel ("fused_spatial"); its
 * launches are counted here only (neither under D3D_KC_LINEAR nor D3D_KC_ATTN_SPATIAL) */
#define D3D_KC_QKV_SATTN 11
/* temporal blocks likewise ("fused_temporal": the qkv GEMM of one (batch, joint) group and its T-key attention as ONE kernel); counted
 * here only (neither under D3D_KC_LINEAR nor D3D_KC_ATTN_TEMPORAL) */
#define D3D_KC_QKV_TATTN 12
#define D3D_KC_COUNT 13
int d3d_engine_set_profiling(d3d_engine* e, int32_t on);
int d3d_engine_profile_reset(d3d_engine* e);
int d3d_engine_profile_read(d3d_engine* e, int32_t kernel_class, double* total_ms, int64_t* launches, double* flops,
                            double* bytes);
const char* d3d_kernel_class_name(int32_t kernel_class);

/* ---- F16X3 range guard.  The F16X3 operand planes hold fp16 (hi, lo) pairs of 8*x (activations: residual stream, q/k/v,
 * attention output, MLP hidden) and 2^k*w (GEMM weights; k is chosen per matrix at commit: 12 unless an entry exceeds 15.99 --
 * LayerNorm-folded weights W diag(gamma) of checkpoints with large gains -- then smaller, nothing is clamped); activations
 * beyond the fp16 range, |x| > 8188, are not representable (the row and attention kernels clamp them to the range, the GEMM
 * epilogues let them become inf): results are then meaningless.
 * Every kernel that writes such planes raises a sticky flag in a word of device memory that belongs to the ENGINE it was launched
 * for when a value left the range (no cost in a healthy run), and d3d_engine_commit_weights notes non-finite weights.
 * d3d_engine_range_flags synchronises `stream` and returns
 *   D3D_RANGE_ACT    an activation of THIS engine left the range since its flag was last cleared
 *   D3D_RANGE_WEIGHT a GEMM weight of this engine was not finite at commit
 *   D3D_RANGE_STATS  a LayerNorm folded into a GEMM met a row with |mean| > 16 standard deviations: the folded form works from
 *                    one-pass row statistics (sum, sum of squares), whose variance loses accuracy like eps (1 + mean^2 / var)
 *                    -- beyond ~25 sigma the 1e-4 parity gate is no longer guaranteed (post-norm biases that dwarf the gains)
 * clear != 0 resets the activation and statistics flags of this engine.  The word is per ENGINE (ABI version 120; it used to be
 * per device): two engines on one device, or evaluate() beside a user's own engine, do not see or consume each other's flags; the
 * two internal streams of one d3d_ddim_sample call report to the same word, which is read behind their join.  The single-op hooks
 * below belong to no engine and report nowhere.  A set flag means the F16X3 result is NOT fp32-accurate for this checkpoint / input:
 * run the engine with D3D_PREC_FP32 (RUN:226-235 loads arbitrary checkpoints; random-init and LayerNorm-ed streams stay far
 * inside the range). */
#define D3D_RANGE_ACT 1u
#define D3D_RANGE_WEIGHT 2u
#define D3D_RANGE_STATS 4u
/* not a precision matter, raised in every mode: a timestep handed to d3d_q_sample / d3d_weighted_loss was outside [0, num_timesteps)
 * (the reference raises IndexError on table[t], DIFF:21-24, 411): the row's output is NaN, no table entry was read */
#define D3D_RANGE_INDEX 8u
/* not a precision matter either, raised in every mode: the head kernel (S2S:217-220 + the DDIM update) forms the three dot products of
 * every row TWICE, from independently loaded weight fragments, and the two evaluations disagreed bit for bit.  They cannot on a healthy
 * machine; this is the signature of the one run-to-run deviation ever seen in this library (one wrong o[0] in ~1 launch of 60, only
 * beside a second process on the GPU, only with an instruction schedule that is pinned out by an ISA test -- mechanism unidentified,
 * experiments/NOTES.md).  The kernel repairs the row by a third evaluation (majority) and raises this bit so that a toolchain or
 * driver change that brings the deviation back is reported by the default guard read instead of silently moving a pose. */
#define D3D_RANGE_RECOMPUTE 16u
int d3d_engine_range_flags(d3d_engine* e, uint32_t* flags, int32_t clear, void* stream);

/* The guard WITHOUT a blocking synchronisation (ABI version 130) -- what the Python layer does by default after every compute call
 * (RUN:226-235 loads arbitrary checkpoints, so the default precision has to notice when one is outside its range by itself).
 * d3d_engine_range_post enqueues on `stream`, behind everything already there, ONE one-lane kernel that exchanges this engine's
 * word with zero and stores the value into a pinned host slot the engine owns, then records an event; *ticket names that snapshot.
 * d3d_engine_range_take returns the flags of a ticket (same bits as d3d_engine_range_flags; D3D_RANGE_WEIGHT is sticky):
 *   block == 0  never waits: *ready = 0 while the snapshot kernel has not run yet (flags untouched), 1 once it has
 *   block != 0  waits for the ticket's event (NOT for the whole stream or device)
 * The flags of a ticket cover every kernel launched for this engine on `stream` since the previous post (or read with clear).
 * 256 tickets are kept; an older one: D3D_EINVAL.  Not legal while `stream` is capturing (D3D_ESTATE).  Measured cost of a post per
 * d3d_ddim_sample at T = 243 / B = 64: one ~2 us launch in 477 ms. */
int d3d_engine_range_post(d3d_engine* e, void* stream, int64_t* ticket);
int d3d_engine_range_take(d3d_engine* e, int64_t ticket, int32_t block, uint32_t* flags, int32_t* ready);

/* ---- debug trace: while on (capacity > 0 slots), every kernel of the F16X3 block flow and the head is followed on `stream`
 * by a checksum launch over the rows it has just written (64-bit position-weighted word sums, order-independent), so two runs
 * of the same call can be compared kernel by kernel (experiments/bisect_two_proc.py).  d3d_engine_trace_read synchronises
 * `stream`, copies out up to cap (sum, tag) pairs in launch order and clears the log;
 * views (1..8): every buffer is summed `views` times, launch v reading each line through a different XCD's L2 (workgroup ->
 * data mapping rotated by v): the views of one buffer must agree -- a cross-XCD coherence check that needs no reference run.
 * tag = view << 28 | forward << 16 | block << 8 | kernel << 4 | buffer (kernel: 0 embed, 1 stream entry, 2 qkv, 3 attention, 4 proj,
 * 5 fc1, 6 fc2 [+ post-norm], 7 post-norm row kernel, 8 head / DDIM update [buffer 1: its y input], 9 the head launched a second
 * time from the same inputs into scratch; buffer: 0 output, 1 row statistics).
 * Graph replay is bypassed while tracing.  capacity 0 turns it off. */
int d3d_engine_set_trace(d3d_engine* e, int32_t capacity, int32_t views);
int d3d_engine_trace_read(d3d_engine* e, uint64_t* sums_host, uint32_t* tags_host, int32_t cap, int32_t* n, void* stream);

/* ---- single-op hooks: the same kernels the engine launches, exposed for the parity tests -------------------------- */
/* Time-embedding table (S2S:29-36,169-174 trunk, then every Block.time_mlp S2S:104-107) for n fp32 timesteps on the
 * device: out (n, 2*depth, D) in execution order STE0, TTE0, STE1, ...  scratch: n*(D + 4*D) floats. */
int d3d_op_time_embedding(d3d_engine* e, const float* times_dev, int32_t n, float* out_dev, float* scratch_dev,
                          void* stream);
/* C[M,N] = epi(A[M,K] @ W[N,K]^T + bias[N]); epi 0 none, 1 erf-GELU (FP32: erff; F16X3: erfc series, |err| < 1e-7 |x|),
 * 2 add residual R[M,N] (R may alias C). */
int d3d_op_linear(const float* A_dev, const float* W_dev, const float* bias_dev, const float* R_dev, float* C_dev,
                  int32_t M, int32_t N, int32_t K, int32_t epi, int32_t precision, void* stream);
/* d3d_op_linear with a choice of tile variant (F16X3: 0 = the engine's choice, 13 = 256x256, 4 = 256x128, 9 = on-the-fly A
 * split)
 * and a timing leg: after one untimed call, `reps` back-to-back launches are timed with HIP events on `stream` and the
 * mean written to *avg_ms (nullable).  Operand conversion for F16X3 happens once, outside the timed launches. */
int d3d_op_linear_bench(const float* A_dev, const float* W_dev, const float* bias_dev, const float* R_dev, float* C_dev,
                        int32_t M, int32_t N, int32_t K, int32_t epi, int32_t precision, int32_t variant, int32_t reps,
                        float* avg_ms, void* stream);
/* Y[M,N] = LayerNorm(R + A[M,K] @ W[N,K]^T + bias; gamma, beta, eps) [+ pos[(m / pos_div) % pos_mod, :]]
 *           [+ tvec[(m / rows_per_batch) * tvec_stride + :]]
 * -- the fc2 GEMM of a block with the block's post-norm applied in its epilogue (S2S:131-135 followed by S2S:236 / 245, and
 * the additions of S2S:238-242 / 113-116), the form the F16X3 engine runs: whole-row tiles, so the sum never leaves the chip
 * un-normalised.  F16X3 only; N == 512, K % 32 == 0 (anything else: D3D_EUNSUP).  pos / tvec nullable.  stats nullable:
 * [M,2] = (sum, sum of squares) of every Y row -- when given, the plane-output form runs (the one inside the engine, which
 * hands these statistics to the next LayerNorm-folded GEMM) and Y is decoded from its planes; otherwise the fp32-output
 * form (last block).  reps / avg_ms: timing leg as d3d_op_linear_bench. */
int d3d_op_linear_postnorm(const float* A_dev, const float* W_dev, const float* bias_dev, const float* R_dev,
                           const float* gamma_dev, const float* beta_dev, float eps, const float* pos_dev, int32_t pos_div,
                           int32_t pos_mod, const float* tvec_dev, int64_t tvec_stride, int32_t rows_per_batch, float* Y_dev,
                           float* stats_dev, int32_t M, int32_t N, int32_t K, int32_t reps, float* avg_ms, void* stream);
/* Regression head of the engine's weights (S2S:217-220: LayerNorm eps 1e-5 + Linear D -> 3) on rows x D fp32 rows:
 * x0 (rows, 3), raw (no clamp, no DDIM update). */
int d3d_op_head(d3d_engine* e, const float* X_dev, float* x0_dev, int32_t rows, void* stream);
/* Row LayerNorm over the last axis (S2S:95,101,236,245 eps 1e-6; S2S:218 eps 1e-5). */
int d3d_op_layernorm(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* out_dev, int32_t rows,
                     int32_t D, float eps, void* stream);
/* GRAND attention core (S2S:75-83) on a packed qkv buffer (B*T*J, 3*D): out = (softmax(q k^T / sqrt(dh)) - I) v,
 * token-major (B*T*J, D).  temporal = 0: groups are frames (N = J keys); 1: groups are joints (N = T keys).
 * force_generic != 0 selects the slow any-shape kernel (cross-check). */
int d3d_op_attention(const float* qkv_dev, float* out_dev, int32_t B, int32_t T, int32_t J, int32_t D, int32_t H,
                     int32_t temporal, int32_t precision, int32_t force_generic, void* stream);

/* ---- machine probes (measurement support; no reference counterpart) ------------------------------------------------ */
/* What THIS device sustains for the two resources that co-limit the F16X3 GEMM k-loop, measured over about ms_target
 * milliseconds on `stream` (a bench line can then state its roofline fraction against measured ceilings beside the nominal
 * peaks of MI355X_MICROARCH.md; bench.py's "machine_probes" object).  what 0: *result = TFLOP/s of fp16 MFMA work
 * (v_mfma_f32_16x16x32_f16, two waves per SIMD on every CU, register operands with the statistics of real hi / lo halves --
 * the power-limited clock included); what 1: *result = GB/s, summed over the CUs, of the k-loop's staging stream alone
 * (global_load_lds_dwordx4 pieces of 256 x 256 x 32 stages from L2-resident operand rows, no MFMA).  Needs no engine. */
int d3d_probe_machine(int32_t what, float ms_target, float* result, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* D3D_H_ */
