/*
 * aurppo.h -- C ABI of libaurppo_hip.so: the MI355X (gfx950) hot path of the aur_ppo trainer.
 *
 * The reference (biirving/aur_ppo) is pure Python/PyTorch and has no FFI of its own; the seam this
 * library replaces is the block of torch-op chains between "rollout buffer filled" and
 * "loss.backward()" in src/ppo.py / src/robot_ppo.py.  Each entry point cites the reference lines
 * whose arithmetic it performs.  Binding stub a maintainer would add: INTEGRATION.md (ctypes).
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer unless its name ends in _h.  Tensors are fp32, contiguous.
 *     Rollout tensors are time-major (T, N): element (t, n) at t*N + n -- the layout of
 *     torch_buffer (src/ppo.py:24-29) and of buffer.flatten() (src/ppo.py:32-39).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls enqueue and
 *     return; they never allocate, never synchronise (except the *_get_state/_set_state helpers).
 *     All entry points are safe to capture into a hipGraph EXCEPT the aurppo_mt19937_* / aurppo_shuffle_* family: a
 *     generator handle carries host-side sequence state and its own side streams, so those calls must run eagerly
 *     (aurppo_shuffle_*_i32 return AURPPO_EINVAL on a capturing stream); the trainer runs them on a side stream, one
 *     update ahead of the captured graph that reads their output.
 *   - Return 0 on success, <0 on error; aurppo_last_error() gives a thread-local message.
 *   - The caller owns all tensors.  The library owns only RNG handles.
 *   - No CPU fallbacks exist in this library: without a gfx950 device every compute call fails.
 */
#ifndef AURPPO_H
#define AURPPO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AURPPO_OK 0
#define AURPPO_EINVAL (-1) /* null pointer, bad enum, misaligned workspace            */
#define AURPPO_ESHAPE (-2) /* non-positive / inconsistent sizes, workspace too small  */
#define AURPPO_EHIP (-3)   /* a HIP runtime call failed (launch error, no device ...) */

#define AURPPO_VERSION 1

/* gae `mode` */
#define AURPPO_GAE 0           /* ppo.run_gae                       src/ppo.py:125-142        */
#define AURPPO_NORMAL_ADV 1    /* ppo.normal_advantage              src/ppo.py:145-157        */
#define AURPPO_GAE_SKIP_LAST 2 /* robot_ppo.run_gae as written      src/robot_ppo.py:224-244  */

/* loss `vloss_mode` (a bool clip_vloss maps to 1; the two un-clipped flavours differ upstream) */
#define AURPPO_VLOSS_RETURNS 0   /* 0.5*mean((v-R)^2)               src/robot_ppo.py:390      */
#define AURPPO_VLOSS_CLIPPED 1   /* clipped value loss              src/ppo.py:250-259        */
#define AURPPO_VLOSS_OLDVALUES 2 /* 0.5*mean((v-V_old)^2)           src/ppo.py:261            */

/* layout of out_scalars written by aurppo_loss_fwd_bwd_f32 */
#define AURPPO_S_LOSS 0     /* pg - ent_coef*ent + vl*vf_coef       src/ppo.py:264 */
#define AURPPO_S_PG 1       /* policy_loss                          src/ppo.py:245 */
#define AURPPO_S_VL 2       /* value_loss (un-weighted)             src/ppo.py:259,261 */
#define AURPPO_S_ENT 3      /* entropy_loss = mean(entropy)         src/ppo.py:263 */
#define AURPPO_S_OLD_KL 4   /* mean(-log_ratio)                     src/ppo.py:232 */
#define AURPPO_S_KL 5       /* mean((ratio-1)-log_ratio)            src/ppo.py:233 */
#define AURPPO_S_CLIPFRAC 6 /* mean(|ratio-1| > clip)               src/ppo.py:234 */
#define AURPPO_S_ADV_MEAN 7 /* minibatch advantage mean             src/ppo.py:239 */
#define AURPPO_S_ADV_STD 8  /* minibatch advantage std (ddof=1)     src/ppo.py:239 */
#define AURPPO_N_SCALARS 9

#define AURPPO_MAX_STREAMS 8

int aurppo_version(void);
const char* aurppo_last_error(void);
/* Number of gfx950 devices visible; <0 on HIP error.  Does not create a context on any of them. */
int aurppo_device_count(void);
/* Which build of the fused MLP step (K7) aurppo_mlp_ppo_*_f32 launches: 2 = k_mlp_step2 (v_mfma_f32_32x32x2_f32, fp32
 * operands), 3 = k_mlp_step3 (v_mfma_f32_32x32x16_bf16 over three-way bf16 splits of the fp32 operands, six products,
 * fp32 accumulate).  Same arguments, same results to the tolerances of tests/test_mlp_fused.py; environment variable
 * AURPPO_K7_VARIANT overrides the built-in default.  (No reference counterpart: the policy nets are torch modules
 * there, src/models/actor_critic.py:8-51.) */
int aurppo_k7_variant(void);
/* The library reads its AURPPO_* environment knobs once per process; this re-reads them now (diagnostics: bench.py switches
 * AURPPO_K7_VARIANT after its timed region to time the fp32-MFMA build beside the default).  Returns 0. */
int aurppo_reload_knobs(void);
/* Which kernel the aurppo_mlp_wide_* entry points launch for a net shape (K7w): 1 = both nets per workgroup on fp32 MFMA
 * (hidden_dim and state_dim <= 64), 2 = one net per workgroup on fp32 MFMA, 3 = one net per workgroup on bf16 MFMA over
 * three-way splits (default for the wider shapes; AURPPO_K7W_VARIANT=2 selects 2).  Same results to the tolerances of
 * tests/test_mlp_wide.py. */
int aurppo_k7w_kernel(int hidden, int state_dim);

/* ---- K1: advantage estimation -------------------------------------------------------------
 * Replaces ppo.run_gae / ppo.normal_advantage (src/ppo.py:125-157; duplicates in
 * src/utils/advantages.py:4-37) and robot_ppo.run_gae (src/robot_ppo.py:224-244, mode 2).
 *   for t = T-1..0:  nnt = 1 - done[t+1] (next_done at T-1);  nv = V[t+1] (next_value at T-1)
 *     delta = (r[t] + ((g*nv)*nnt)) - V[t];  A[t] = delta + (((g*lam)*nnt) * A[t+1]);  R = A + V
 * fp32 with exactly that association and no fused multiply-add (bit-identical to the reference's
 * CPU result); gamma is rounded to fp32, gamma*lam is formed in fp64 then rounded, as Python does.
 * Mode 1:  R[t] = r[t] + ((g*nnt)*R[t+1]) (R[T] = next_value);  A = R - V.
 * Mode 2:  the loop starts at T-2, so A[T-1] = 0 and R[T-1] = V[T-1] (upstream behaviour).       */
int aurppo_gae_f32(const float* rewards, const float* values, const float* terminals, /* (T,N) */
                   const float* next_value, const float* next_done,                    /* (N,)  */
                   float* advantages, float* returns,                                  /* (T,N) out */
                   int T, int N, double gamma, double lam, int mode, void* stream);

/* K1 + pack: as aurppo_gae_f32, and additionally writes the per-sample record
 *   rec[(t*N+n)*4 + {0,1,2,3}] = { log_probs[t,n], A[t,n], R[t,n], V[t,n] }   ((T*N, 4), 16-B aligned)
 * i.e. the four per-sample scalars buffer.flatten() hands to the update (src/ppo.py:32-39) laid out
 * so that the minibatch gather fetches them with ONE 16-byte request per sample.                   */
int aurppo_gae_pack_f32(const float* rewards, const float* values, const float* terminals,
                        const float* next_value, const float* next_done, const float* log_probs,
                        float* advantages, float* returns, float* rec, int T, int N, double gamma,
                        double lam, int mode, void* stream);

/* ---- K2: numpy-legacy MT19937 + Fisher-Yates shuffle --------------------------------------
 * Replaces np.random.seed(seed) (src/ppo.py:182) and np.random.shuffle(b_inds) (src/ppo.py:217,
 * src/robot_ppo.py:338).  Bit-exact with numpy's RandomState: init_genrand seeding, one 32-bit
 * draw per masked-rejection trial, descending Fisher-Yates.  The handle is a device-resident
 * generator state; successive shuffles continue the same stream, as the global numpy stream does. */
typedef struct aurppo_rng aurppo_rng;
int aurppo_mt19937_create(aurppo_rng** out, uint32_t seed, int max_n, void* stream);
int aurppo_mt19937_destroy(aurppo_rng* rng);
int aurppo_mt19937_seed(aurppo_rng* rng, uint32_t seed, void* stream);
/* Host copies of (key[624], pos) == np.random.get_state()[1:3].  These two synchronise `stream`. */
int aurppo_mt19937_get_state(aurppo_rng* rng, uint32_t* key_h, int32_t* pos_h, void* stream);
int aurppo_mt19937_set_state(aurppo_rng* rng, const uint32_t* key_h, int32_t pos_h, void* stream);
/* out[0] (device) = 1.0f if any shuffle since the last (re)seed ran out of pre-generated draws (the inventory rule
 * covers the expectation + 12 sigma of numpy's rejection sampling; beyond it the permutation is invalid and the flag
 * sticks), else 0.0f.  Enqueued on `stream`, never synchronises: the trainer folds it into the one host read it makes
 * per update (what np.random.shuffle, src/ppo.py:217, cannot fail at). */
int aurppo_mt19937_status_f32(aurppo_rng* rng, float* out, void* stream);
/* idx[i] = i  (np.arange, src/ppo.py:213) */
int aurppo_arange_i32(int32_t* idx, int n, void* stream);
/* In-place shuffle of idx[0..n), n <= max_n given at create. */
int aurppo_shuffle_i32(aurppo_rng* rng, int32_t* idx, int n, void* stream);
/* `epochs` successive shuffles of one carried array, epoch e written to out[e*n .. (e+1)*n):
 * out[0] = shuffle(arange(n)), out[e] = shuffle(out[e-1])  (src/ppo.py:213-217).               */
int aurppo_shuffle_epochs_i32(aurppo_rng* rng, int32_t* out, int n, int epochs, void* stream);

/* ---- K3: fused minibatch gather -----------------------------------------------------------
 * Replaces the advanced-indexing gathers b_obs[mb_inds], b_actions[mb_inds], b_logprobs[mb_inds],
 * b_advantages[mb_inds], b_returns[mb_inds], b_values[mb_inds] (src/ppo.py:219-220,225,236,251-257;
 * src/robot_ppo.py:341-345) with ONE launch over up to 8 streams:
 *   dst_h[s][m*row_elems_h[s] + e] = src_h[s][idx[m]*row_elems_h[s] + e],  0<=m<M.
 * src_h/dst_h/row_elems_h are HOST arrays (of device pointers / ints) of length n_streams.       */
int aurppo_gather_f32(const int32_t* idx, int M, const float* const* src_h, float* const* dst_h,
                      const int* row_elems_h, int n_streams, void* stream);

/* ---- K4+K5: advantage normalisation + clipped-surrogate loss, forward and backward ---------
 * Replaces src/ppo.py:225-264 (src/robot_ppo.py:345-398): log-ratio, ratio, KL diagnostics,
 * clip fraction, minibatch advantage normalisation (mean, unbiased std, +1e-8), policy loss,
 * value loss (vloss_mode), entropy bonus, total loss -- and their autograd:
 *   g_newlogp[m] = d loss / d newlogp[m],  g_newv[m] = d loss / d newv[m],
 *   g_entropy[m] = -ent_coef / M     (torch conventions for max ties and clamp edges).
 * clip / ent_coef / vf_coef are the Python floats of the params dict (fp64); the clip bounds are
 * formed as 1-clip, 1+clip in fp64 and rounded to fp32, as torch does.
 * out_scalars: AURPPO_N_SCALARS floats (device).  workspace: aurppo_loss_workspace_bytes(M)
 * bytes, 16-byte aligned, contents need not be initialised.                                      */
size_t aurppo_loss_workspace_bytes(int M);
int aurppo_loss_fwd_bwd_f32(const float* newlogp, const float* oldlogp, const float* adv,
                            const float* newv, const float* oldv, const float* ret,
                            const float* entropy, int M, double clip, double ent_coef, double vf_coef,
                            int norm_adv, int vloss_mode, float* out_scalars, float* g_newlogp,
                            float* g_newv, float* g_entropy, void* workspace, void* stream);

/* Same computation with the old-side inputs as a gathered (M,4) record {old_logp, adv, ret, old_v}
 * (rows of aurppo_gae_pack_f32's rec, picked by aurppo_gather_f32 with row_elems = 4).              */
int aurppo_loss_fwd_bwd_packed_f32(const float* newlogp, const float* newv, const float* entropy,
                                   const float* rec, int M, double clip, double ent_coef,
                                   double vf_coef, int norm_adv, int vloss_mode, float* out_scalars,
                                   float* g_newlogp, float* g_newv, float* g_entropy, void* workspace,
                                   void* stream);

/* ---- K7: fused minibatch step for the MLP actor-critic -----------------------------------------
 * One launch sequence replaces, for the reference's MLP policy (actor_critic over two Tanh hidden
 * layers of `hidden` units; Gaussian head with state-independent log-std when `continuous`, Categorical
 * head over A logits otherwise; src/models/actor_critic.py:8-51, src/nets/nets.py:19-53), everything
 * between "mb_inds chosen" and "gradients ready":
 *   gathers (src/ppo.py:219-220,225,236,251-257) + evaluate() (src/ppo.py:220) + the loss block
 *   (src/ppo.py:225-264) + loss.backward() (src/ppo.py:267).
 * obs (B,D), actions ((B,A) floats, or (B,) action indices stored as floats for the Categorical head)
 * and rec (B,4) = {old_logp, adv, ret, old_v} are the flattened rollout buffers, idx (M,) the minibatch slice of the epoch permutation.  `params` is the flat parameter
 * bucket, layout_h 13 float offsets into it {w1,b1,w2,b2,w3,b3} for the actor, the same for the
 * critic, then actor_logstd (ignored for the Categorical head; nn.Linear layout: weight[out][in]).  `grads` (n_params floats) is
 * OVERWRITTEN with d loss / d params in the same layout; out_scalars as aurppo_loss_fwd_bwd_f32.
 * Built for hidden = 64, two layers, D <= 64, A <= 16; other shapes return AURPPO_ESHAPE (callers then use
 * aurppo_mlp_wide_ppo_step_f32 below, or the per-op path beyond its limits).  workspace: aurppo_mlp_workspace_bytes(n_params) bytes, 16-byte aligned.      */
size_t aurppo_mlp_workspace_bytes(int n_params);
int aurppo_mlp_ppo_step_f32(const float* obs, const float* actions, const float* rec,
                            const int32_t* idx, int M, int D, int A, int continuous, int hidden,
                            const float* params, const int* layout_h, int n_params, float* grads,
                            double clip, double ent_coef, double vf_coef, int norm_adv, int vloss_mode,
                            float* out_scalars, void* workspace, void* stream);

/* Measurement variant: identical work, and the two caller-owned hipEvent_t handles (either may be NULL)
 * are recorded on `stream` immediately before and after the main kernel (k_mlp_step) -- bench.py times
 * the kernel with them (hipEventElapsedTime) without a profiler attached.                           */
int aurppo_mlp_ppo_step_ev_f32(const float* obs, const float* actions, const float* rec,
                               const int32_t* idx, int M, int D, int A, int continuous, int hidden,
                               const float* params, const int* layout_h, int n_params, float* grads, double clip,
                               double ent_coef, double vf_coef, int norm_adv, int vloss_mode,
                               float* out_scalars, void* workspace, void* stream, void* ev_begin,
                               void* ev_end);

/* ---- K7 + K6b chained: one whole minibatch of the update loop ---------------------------------------
 * Replaces src/ppo.py:219-269 for one minibatch -- everything aurppo_mlp_ppo_step_f32 does, then
 * nn.utils.clip_grad_norm_(parameters, max_norm) and optimizer.step() (Adam, as aurppo_clip_adam_f32) over
 * the same flat bucket of n_params floats (params / grads / exp_avg / exp_avg_sq), in three launches instead
 * of five.  Single-process only: there is no place for a gradient all-reduce between the halves.
 * next_idx / next_M: the index slice of the minibatch that will be stepped next (NULL: none); its advantage
 * statistics and the operand copy of the updated first layer are prepared by this call's last launch.
 * chained != 0 promises that the previous call on this workspace was this function with next_idx equal to this
 * call's idx (same M); then nothing is prepared again.  lr_dev / step_dev / out_norm as aurppo_clip_adam_f32. */
int aurppo_mlp_ppo_minibatch_f32(const float* obs, const float* actions, const float* rec, const int32_t* idx, int M,
                                 int D, int A, int continuous, int hidden, float* params, const int* layout_h,
                                 int n_params, float* grads, double clip, double ent_coef, double vf_coef,
                                 int norm_adv, int vloss_mode, float* out_scalars, float* exp_avg,
                                 float* exp_avg_sq, double max_norm, const float* lr_dev, float* step_dev,
                                 double beta1, double beta2, double eps, float* out_norm, const int32_t* next_idx,
                                 int next_M, int chained, void* workspace, void* stream);

/* Packed records.  Every aurppo_mlp_ppo_* entry point accepts actions == NULL: rec is then (B, 16) floats per sample,
 * {old_logp, A, R, V, action row (at most 12 floats), 0 ...}, built by aurppo_pack_records_f32 from the (B, 4)
 * records of aurppo_gae_pack_f32 and the (B, action_floats) action buffer.  A sample's record and action row then
 * share one 64-B line instead of one 128-B line each: 3 lines per sample instead of 4 in K7's gather.            */
int aurppo_pack_records_f32(const float* rec4, const float* actions, int B, int action_floats, float* rec64,
                            void* stream);

/* The same minibatch in two halves, for one process per GPU (src/ppo.py:219-269 with a gradient all-reduce between
 * loss.backward() and clip_grad_norm_): aurppo_mlp_ppo_grad_f32 = aurppo_mlp_ppo_step_f32 that also advances the Adam
 * step count (step_dev) and, with chained != 0, skips the preparation the previous apply call already did;
 * aurppo_mlp_ppo_apply_f32 = clip_grad_norm_ + Adam over the bucket in ONE launch: the stored gradient is first
 * multiplied by grad_scale (1/world after a SUM all-reduce), its norm is formed inside the kernel (grads itself is
 * left as it was: other workgroups are still reading it), and -- as in
 * aurppo_mlp_ppo_minibatch_f32 -- the statistics of next_idx and the operand copy of the new first layer are
 * prepared for the next aurppo_mlp_ppo_grad_f32(chained = 1) on this workspace (rec_floats: 4, or 16 for packed
 * records, below).                                                                                           */
int aurppo_mlp_ppo_grad_f32(const float* obs, const float* actions, const float* rec, const int32_t* idx, int M, int D,
                            int A, int continuous, int hidden, const float* params, const int* layout_h, int n_params,
                            float* grads, double clip, double ent_coef, double vf_coef, int norm_adv, int vloss_mode,
                            float* out_scalars, float* step_dev, int chained, void* workspace, void* stream);
int aurppo_mlp_ppo_apply_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const int* layout_h,
                             int n_params, int D, double grad_scale, double max_norm, const float* lr_dev,
                             const float* step_dev, double beta1, double beta2, double eps, float* out_norm,
                             const float* rec, int rec_floats, const int32_t* next_idx, int next_M, void* workspace,
                             void* stream);

/* aurppo_mlp_ppo_apply_f32 for a gradient that is already the mean over ranks (aurppo_p2p_allreduce_mean_f32 below) and comes
 * with per-workgroup partial sums of squares: the clip's norm is their sum (n_part of them), nothing is rescaled. */
int aurppo_mlp_ppo_apply_parts_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const int* layout_h,
                                   int n_params, int D, const double* sq_part, int n_part, double max_norm,
                                   const float* lr_dev, const float* step_dev, double beta1, double beta2, double eps,
                                   float* out_norm, const float* rec, int rec_floats, const int32_t* next_idx, int next_M,
                                   void* workspace, void* stream);

/* ---- K11: 3x3 convolution of the robot policy's hidden encoder blocks on the bf16 matrix pipe -----------------
 * nn.Conv2d(Ci, Co, 3, padding = pad) WITHOUT bias (src/nets/base_cnns.py:32-45; the bias / ReLU / max-pool tail is
 * aurppo_bias_relu_pool2_*), stride 1, NCHW fp32, as an implicit GEMM whose fp32 products are formed from three-way bf16
 * splits (fp32-equivalent).  mode 0: z = conv2d(x, w, padding = pad), x (B, Ci, H, W) -> z (B, Co, H + 2 pad - 2, W + 2 pad - 2).
 * mode 1: the gradient with respect to the input: x is the OUTPUT gradient (B, Co, H, W), z the input gradient
 * (B, Ci, H + 2 - 2 pad, W + 2 - 2 pad), w the same (Co, Ci, 3, 3) filter, pad the forward padding.  The product's input
 * channel count (Ci in mode 0, Co in mode 1) must be a multiple of 16.  wop_ws: aurppo_conv3x3_wop_bytes(input channels,
 * output channels of the product) bytes of scratch for the filter in operand order.  (The weight gradient: aurppo_conv3x3_wgrad_f32.) */
size_t aurppo_conv3x3_wop_bytes(int cin_gemm, int cout_gemm);
int aurppo_conv3x3_f32(const float* x, const float* w, float* z, int B, int Ci_w, int Co_w, int H, int W, int pad, int mode,
                       void* wop_ws, void* stream);

/* nn.Linear without its bias on the same arithmetic (the MLP policies wider than the fused steps cover, hidden_dim > 128:
 * src/nets/nets.py:21-27,33-39,45-51 per layer).  mode 0: y (M, N_w) = x (M, K_w) . w (N_w, K_w)^T, K_w a multiple of 16;
 * mode 1 (gradient with respect to the input; x is dY): y (M, K_w) = x (M, N_w) . w, N_w a multiple of 16.  Row-major fp32.
 * wop_ws: aurppo_conv3x3_wop_bytes(inner dimension, columns) bytes. */
int aurppo_linear_f32(const float* x, const float* w, float* y, long long M, int K_w, int N_w, int mode, void* wop_ws,
                      void* stream);
/* The forward product with the layer's bias (may be NULL) and, act = 1, the nn.Tanh that follows every hidden layer
 * (src/nets/nets.py:21-27) in its epilogue: y = act(x . w^T + bias). */
int aurppo_linear_bias_act_f32(const float* x, const float* w, const float* bias, float* y, long long M, int K_w, int N_w,
                               int act, void* wop_ws, void* stream);
/* nn.Linear's weight gradient on the same arithmetic: dw (N, K) = dy (M, N)^T . x (M, K) (what loss.backward() leaves in
 * <layer>.weight.grad, src/ppo.py:266).  Both operands are split once per workgroup through LDS; the minibatch's rows are cut
 * into slices whose partial products are summed in slice order (deterministic).  N, K multiples of 4; ws:
 * aurppo_linear_wgrad_ws_bytes(M, N, K) bytes. */
size_t aurppo_linear_wgrad_ws_bytes(long long M, int N, int K);
int aurppo_linear_wgrad_f32(const float* dy, const float* x, float* dw, long long M, int N, int K, void* ws, void* stream);

/* K12 -- the weight gradient of the 3x3 convolution of aurppo_conv3x3_f32 (src/nets/base_cnns.py:32-45, src/nets/equiv.py:12-62;
 * what loss.backward() leaves in <conv>.weight.grad, src/robot_ppo.py:389): dw (Co, Ci, 3, 3) from x (B, Ci, H, W) and the
 * output gradient dy (B, Co, H + 2 pad - 2, W + 2 pad - 2), NCHW fp32, as a product over the batch's output pixels on bf16 MFMAs
 * over three-way splits (fp32-equivalent).  ws: aurppo_conv3x3_wgrad_ws_bytes(...) bytes (0 = shape not supported: pad outside
 * 0..2, an empty output, or one image of either tensor of 1 GB and more). */
size_t aurppo_conv3x3_wgrad_ws_bytes(int B, int Ci, int Co, int H, int W, int pad);
int aurppo_conv3x3_wgrad_f32(const float* dy, const float* x, float* dw, int B, int Ci, int Co, int H, int W, int pad, void* ws,
                             void* stream);

/* ---- one-shot gradient all-reduce over peer memory (one process per GPU; SURVEY 8e plan B) ------------------
 * Where it sits in the reference: between loss.backward() and clip_grad_norm_ (src/ppo.py:266-268); upstream is
 * single-process, so there is no call to cite -- the semantics are "mean of the ranks' gradients, identical bits on
 * every rank".  An aurppo_p2p handle owns one exchange buffer (two slots of max_floats + flags) on the CURRENT device.
 * Set-up, once, host side:  create on every rank -> get_handle (aurppo_p2p_handle_bytes() bytes) -> exchange the handles
 * between the processes by any means (the trainer uses torch.distributed.all_gather_object) -> open_peers(handles of all
 * ranks, in rank order; the own entry is ignored).  HSA_ENABLE_IPC_MODE_LEGACY=0 must be in every rank's environment.
 * aurppo_p2p_allreduce_mean_f32: grads[0..n) <- mean over ranks, summed in rank order (bit-identical everywhere), in ONE
 * launch and one hop over xGMI; sq_part (optional, aurppo_p2p_parts(n) doubles) receives per-workgroup partial sums of
 * squares of the result.  step_dev: a device float holding the exchange's sequence number -- Adam's step count -- which
 * every rank must advance identically by exactly one between calls (aurppo_mlp_ppo_grad_f32 does).  Capturable into a
 * hipGraph.  A peer that does not arrive within timeout_s (<= 0: 10 s) raises a sticky status instead of hanging.      */
typedef struct aurppo_p2p aurppo_p2p;
int aurppo_p2p_handle_bytes(void);
int aurppo_p2p_parts(int n);
int aurppo_p2p_create(aurppo_p2p** out, int rank, int world, int max_floats, void* stream);
int aurppo_p2p_get_handle(aurppo_p2p* x, void* handle_h);
int aurppo_p2p_open_peers(aurppo_p2p* x, const void* handles_h);
int aurppo_p2p_allreduce_mean_f32(aurppo_p2p* x, float* grads, int n, const float* step_dev, double* sq_part,
                                  double timeout_s, void* stream);
int aurppo_p2p_status(aurppo_p2p* x, int* status_h, void* stream); /* 0 ok; 1 + r: rank r's flag timed out (sticky) */
int aurppo_p2p_destroy(aurppo_p2p* x);

/* ---- K8: rollout step for the MLP actor-critic -----------------------------------------------------
 * Replaces `action, logprob, _, value = policy.evaluate(next_obs)` under no_grad and the three buffer row
 * stores that follow it (src/ppo.py:104-108), and with noise == NULL the bootstrap `policy.value(next_obs)`
 * (src/ppo.py:161).  obs (N,D); noise: (N,A) standard-normal draws for the Gaussian head (a = mu +
 * exp(logstd)*eps) or (N,) uniform [0,1) draws for the Categorical head (inverse CDF of softmax(logits));
 * outputs go wherever the caller points them -- normally rows of the rollout buffer: actions (N,A) | (N,),
 * logp (N,), value (N,).  params / layout_h / n_params / shape limits as aurppo_mlp_ppo_step_f32.       */
int aurppo_mlp_act_f32(const float* obs, const float* noise, int N, int D, int A, int continuous, int hidden,
                       const float* params, const int* layout_h, int n_params, float* actions, float* logp,
                       float* value, void* stream);

/* ---- K7w / K8w: the same two operators for the other MLP shapes of the reference's CLI --------------------------------
 * src/run_ppo.py:33,37 expose -d/--hidden_dim and -nl/--num_layers and src/nets/nets.py:19-53 builds any of them:
 * `num_layers` (1..3) Tanh layers of `hidden` (1..128) units over a state of D (1..128) floats, A <= 16.  Same
 * arguments and results as aurppo_mlp_ppo_step_ev_f32 / aurppo_mlp_act_f32 except
 *   layout_h: for the actor, then the critic: {w_0, b_0, ..., w_L, b_L} (L = num_layers; layer L is the head), then
 *             actor_logstd -- 4 * (num_layers + 1) + 1 float offsets into the bucket;
 *   workspace: aurppo_mlp_wide_workspace_bytes(n_params, hidden, D) bytes, 64-byte aligned (slab region sized from the
 *             shape); the act entry point only keeps the operand-order copy of the weights there and is content with
 *             aurppo_mlp_wide_workspace_bytes(0, hidden, D).
 * ev_begin / ev_end (either may be NULL): hipEvent_t handles recorded around the main kernel.  The optimizer step that
 * follows is aurppo_clip_adam_f32 (K6b).                                                                              */
size_t aurppo_mlp_wide_workspace_bytes(int n_params, int hidden, int state_dim);
int aurppo_mlp_wide_ppo_step_f32(const float* obs, const float* actions, const float* rec, const int32_t* idx, int M,
                                 int D, int A, int continuous, int hidden, int num_layers, const float* params,
                                 const int* layout_h, int n_params, float* grads, double clip, double ent_coef,
                                 double vf_coef, int norm_adv, int vloss_mode, float* out_scalars, void* workspace,
                                 void* stream, void* ev_begin, void* ev_end);
/* The step chained with clip_grad_norm_ + Adam.step over the same flat bucket (src/ppo.py:219-269 for one minibatch), as
 * aurppo_mlp_ppo_minibatch_f32 is for the default shape -- four launches (prepare, step, slab reduce that also leaves
 * the clip's partial sums and advances the step count, clip + Adam).  next_idx / next_M: the index slice the NEXT call
 * will step (NULL: none) -- the optimizer launch then also drops every updated weight into the operand-order copies and
 * forms that slice's advantage statistics; chained != 0: the previous call named this idx as its next_idx, so this call
 * has no prepare launch (three launches).  Single process only.                                                        */
int aurppo_mlp_wide_ppo_minibatch_f32(const float* obs, const float* actions, const float* rec, const int32_t* idx,
                                      int M, int D, int A, int continuous, int hidden, int num_layers, float* params,
                                      const int* layout_h, int n_params, float* grads, double clip, double ent_coef,
                                      double vf_coef, int norm_adv, int vloss_mode, float* out_scalars,
                                      float* exp_avg, float* exp_avg_sq, double max_norm, const float* lr_dev,
                                      float* step_dev, double beta1, double beta2, double eps, float* out_norm,
                                      const int32_t* next_idx, int next_M, int chained, void* workspace, void* stream);
int aurppo_mlp_wide_act_f32(const float* obs, const float* noise, int N, int D, int A, int continuous, int hidden,
                            int num_layers, const float* params, const int* layout_h, int n_params, float* actions,
                            float* logp, float* value, void* workspace, void* stream);

/* ---- K6: global-norm gradient clip over one flat bucket -------------------------------------
 * Replaces nn.utils.clip_grad_norm_(params, max_norm) (src/ppo.py:268; src/robot_ppo.py:401):
 * norm = ||g||_2, g *= min(1, max_norm / (norm + 1e-6)).  out_norm: 1 float (device), the
 * pre-clip norm.  workspace: aurppo_clip_workspace_bytes(n) bytes.                               */
size_t aurppo_clip_workspace_bytes(int64_t n);
int aurppo_grad_norm_clip_f32(float* flat_grads, int64_t n, double max_norm, float* out_norm,
                              void* workspace, void* stream);

/* K6b: clip + Adam fused over the flat bucket -- clip_grad_norm_ over the first clip_n elements
 * (src/ppo.py:268 clips everything, clip_n = n; src/robot_ppo.py:401 the actor's slice only) followed
 * by torch.optim.Adam's step (src/ppo.py:80,269; betas, eps as given) on all n elements.  lr_dev and
 * step_dev are 1-float DEVICE scalars: the learning rate (so an annealed rate works under hipGraph
 * replay) and the step counter, which this call increments.  exp_avg / exp_avg_sq are the moment
 * buffers (n floats each, zero-initialised by the caller).  workspace: aurppo_clip_workspace_bytes(n). */
int aurppo_clip_adam_f32(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                         int64_t clip_n, double max_norm, const float* lr_dev, float* step_dev,
                         double beta1, double beta2, double eps, float* out_norm, void* workspace,
                         void* stream);

/* ---- K9: bias + gripper-state plane + ReLU + 2x2 max-pool, fused (robot policy encoder) -----------------------
 * Replaces, behind each convolution of the plain-CNN encoder (src/nets/base_cnns.py:28-45), the convolution's bias
 * add, `nn.ReLU` and `nn.MaxPool2d(2)`, and for the first block the tiled gripper-state input channel of
 * src/models/robot_actor_critic.py:58-59,106-107 (convolution is linear in its input channels: the tiled plane
 * contributes scale[b] * plane[c,h,w], plane = conv(ones, W[:, state channel])).
 *   forward:  y[b,c,ho,wo] = max(0, max over the 2x2 window of  x + bias[c] + scale[b]*plane[c,h,w])
 *             mask = window position of the FIRST maximum (0..3, row-major, torch's tie rule), 4 if it is <= 0
 *   backward: dx = dy routed to the masked position (zero elsewhere, zero on an odd trailing row / column),
 *             dbias_part[b*C + c] = sum of dy over that plane's live pooled elements (sum over b gives d bias)
 * x, dx: (B,C,H,W) contiguous; y, dy, mask: (B,C,H/2,W/2) (floor).  bias (C), scale (B) + plane (C,H,W) may be NULL
 * (scale and plane together).  aurppo_weighted_batch_sum_f32: out[k] = sum_b w[b] * x[b,k] -- the gradient of `plane`
 * from dx and scale. */
int aurppo_bias_relu_pool2_fwd_f32(const float* x, const float* bias, const float* scale, const float* plane, float* y,
                                   uint8_t* mask, int B, int C, int H, int W, void* stream);
int aurppo_bias_relu_pool2_bwd_f32(const float* dy, const uint8_t* mask, float* dx, float* dbias_part, int B, int C,
                                   int H, int W, void* stream);
int aurppo_weighted_batch_sum_f32(const float* x, const float* w, float* out, int B, int64_t K, void* stream);

/* ---- K10: the encoder's first block in one kernel ----------------------------------------------------------------
 * conv2d(cat[obs, tiled state], W, bias, padding 1) -> ReLU -> MaxPool2d(2) of src/nets/base_cnns.py:28-31 on the input
 * of src/models/robot_actor_critic.py:58-59,106-107, without the full-resolution tensors: forward reads obs (B,Ci,H,W),
 * Ci in 1..3, W (Co, Ci+1, 3, 3) whose LAST input channel is the state plane, bias (Co, may be NULL), state (B) and
 * writes y, mask (B,Co,H/2,W/2) with K9's conventions.  Backward leaves, per (sample, group of 16 output channels),
 * partial sums dw_part (B*Co/16, 16, (Ci+1)*9) and db_part (B*Co/16, 16): summed over the first dimension and laid out
 * per channel they are the gradients of W (Co, Ci+1, 3, 3) and bias.  The input takes no gradient (it is data).  Co must
 * be a multiple of 16, W at least 3. */
int aurppo_first_block_fwd_f32(const float* obs, const float* w, const float* bias, const float* state, float* y,
                               uint8_t* mask, int B, int Ci, int Co, int H, int W, void* stream);
int aurppo_first_block_bwd_f32(const float* dy, const uint8_t* mask, const float* obs, const float* state,
                               float* dw_part, float* db_part, int B, int Ci, int Co, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AURPPO_H */
