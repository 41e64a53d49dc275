/*
 * swn.h - C ABI of libswn_hip.so: the MI355X (gfx950) kernels of the Switch-NeRF train hot path.
 *
 * The reference (MiZhenxing/Switch-NeRF) has no C/FFI boundary of its own: it is pure Python on torch plus the
 * third-party Tutel JIT kernels.  Every entry point below therefore names the reference *Python* interface it
 * replaces (paths relative to /root/reference/switch_nerf/).  The reference-side binding a maintainer would add
 * is a ctypes stub: see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's allocator in our host code); the library
 *     allocates nothing and keeps no mutable global state besides the last-error string (thread local);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *   - return value: 0 = ok, non-zero = error (message from swn_last_error()); nothing throws or aborts;
 *   - dtype codes: SWN_F32 = 0, SWN_BF16 = 1 (activations / compute copies of weights); master weights,
 *     gradients, gates, z-values and all reductions are fp32; indices are int32;
 *   - activations are row-major [rows, features]; "segments" are the reference's model chunks
 *     (rendering.py:354 `model_chunk_size`): routing, capacity and l_aux are computed per segment.
 */
#ifndef SWN_H
#define SWN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWN_F32 0
#define SWN_BF16 1
#define SWN_F16 2   /* IEEE half compute type (BASELINE configs[4]: fp16 MFMA) - libswn_hip_f16.so, see swn_half_dtype() */

const char* swn_last_error(void);
int swn_version(void);
/* The library is built twice from the same sources: libswn_hip.so computes in fp32 and bfloat16 (dtype codes SWN_F32, SWN_BF16),
 * libswn_hip_f16.so in fp32 and IEEE half (SWN_F32, SWN_F16; fp16 MFMA v_mfma_f32_32x32x16_f16) - identical entry points.  Returns
 * the 16-bit dtype code this build accepts.                                                                                   */
int swn_half_dtype(void);

/* Diagnostic: dumps the lane->element maps of v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32 as used by the
 * kernels (out: 3*64*16 int32 = for each of A(8 used),B(8 used),C(16) slot the (row<<16|col) it addresses). */
int swn_mfma_probe(int32_t* out, void* stream);

/* ---- ray sampling + positional encoding --------------------------------------------------------------------
 * replaces rendering.py:85-90 (z = near(1-t)+far t, xyz = o + d z), :573-584 (_expand_and_perturb_z_vals) and
 * models/nerf.py:21-26 (Embedding.forward) for xyz (L=12) and dir (L=4).
 *   rays[N,8] f32 (o,d,near,far); t_steps[S] f32 = linspace(0,1,S); perturb_rand[N,S] f32 U[0,1) or NULL
 *   z_out[N,S] f32; pe_xyz[N*S, pe_stride] (dtype), columns >= 3+6*Lxyz zero-filled;
 *   pe_dir[N, dir_stride] (dtype) or NULL.                                                                     */
int swn_sample_pe(const float* rays, const float* t_steps, const float* perturb_rand, float perturb,
                  int n_rays, int n_samples, int l_xyz, int l_dir, int dtype,
                  float* z_out, void* pe_xyz, int pe_stride, void* pe_dir, int dir_stride, void* stream);

/* same encoding for caller-supplied depths z[N,S] (fine pass: xyz_fine_fn, rendering.py:246) */
int swn_pe_from_z(const float* rays, const float* z, int n_rays, int n_samples, int l_xyz, int dtype, void* pe_xyz,
                  int pe_stride, void* stream);

/* ---- hierarchical sampling (rendering.py:587-637 _sample_pdf/_sample_cdf; :419-433 coarse+fine merge) ---------
 * z_coarse[N,S], weights[N,S] (coarse compositing weights; bins = mid points, weights[:,1:-1] as in :238-241);
 * u[N,F] uniform samples (NULL = torch.linspace(0,1,F), the deterministic eval branch); z_fine[N,F] out.         */
int swn_sample_pdf(const float* z_coarse, const float* weights, const float* u, int n_rays, int n_coarse, int n_fine,
                   float* z_fine, void* stream);
/* z_out[N,F+S] = sort(cat[z_fine, z_coarse]); order[N,F+S] = source index into the concatenation (ties keep cat order);
 * raw_out[N,F+S,4] = cat[raw_fine, raw_coarse] gathered by order.  F + S <= 1024.                                */
int swn_merge_samples(const float* z_fine, const float* z_coarse, const float* raw_fine, const float* raw_coarse,
                      int n_rays, int n_fine, int n_coarse, float* z_out, int32_t* order, float* raw_out, void* stream);
/* backward of the gather: d_fine[N,F,4], d_coarse[N,S,4] <- d_raw[N,F+S,4] */
int swn_unmerge_grad(const float* d_raw, const int32_t* order, int n_rays, int n_fine, int n_coarse, float* d_fine,
                     float* d_coarse, void* stream);

/* ---- gate: LayerNorm + fp32 router GEMV + softmax + top-1 ----------------------------------------------------
 * replaces nn.LayerNorm (models/nerf_moe.py:370-372), TopKGate logits/softmax
 * (modules/tutel_moe_ext/tutel_moe_layer_nobatch.py:105-126) and topk/gates_s (tutel_fast_dispatch.py:177-182).
 *   g[P,G] (dtype) gate-MLP output; ln_w, ln_b [G] f32 (NULL = no norm); wg[E,G] f32
 *   gates[P,E] f32; idx[P] i32 (first max); gmax[P] f32; stats[P,2] f32 (mean, rstd) for backward.            */
int swn_gate_fwd(const void* g, int dtype, const float* ln_w, const float* ln_b, const float* wg,
                 int n_tokens, int gate_dim, int n_experts,
                 float* gates, int32_t* idx, float* gmax, float* stats, void* stream);
/* The same with gate noise (a TRAINING forward under --gate_noise > 0: opts.py:208, tutel_moe_layer_nobatch.py:119-122,
 * `logits + gate_noise * randn_like(logits) / E`): logits += noise_scale * noise[token][expert] (fp32 [n_tokens][n_experts], drawn by the
 * caller; noise_scale = gate_noise / n_experts) before the softmax.  The backward (swn_gate_bwd) is unchanged - it starts from the
 * probabilities this call wrote.  Runs on the VALU kernel for every shape.                                                          */
int swn_gate_fwd_noise(const void* g, int dtype, const float* ln_w, const float* ln_b, const float* wg, const float* noise,
                       float noise_scale, int n_tokens, int gate_dim, int n_experts, float* gates, int32_t* idx, float* gmax,
                       float* stats, void* stream);

/* backward of the above + the l_aux gradient.  d_gates[s,e] = laux_coef[seg(s)] * counts[seg(s),e]
 * + (e==idx[s]) * d_gmax[s];  then softmax bwd, router bwd (d_wg accumulated with atomics into fp32 [E,G]),
 * LayerNorm bwd (d_ln_w/d_ln_b accumulated).  dg[P,G] (dtype) out.
 * counts: int32 [n_seg,E] (tokens routed to e, before capacity); laux_coef: f32 [n_seg] = dL/dl_aux_seg * E/P^2 */
int swn_gate_bwd(const void* g, int dtype, const float* ln_w, const float* ln_b, const float* wg,
                 const float* gates, const int32_t* idx, const float* d_gmax, const float* stats,
                 const int32_t* counts, const float* laux_coef, int seg_tokens,
                 int n_tokens, int gate_dim, int n_experts,
                 void* dg, float* dlogits /* scratch, written: swn_gate_bwd_scratch_floats() f32 = [P,E] logit gradients followed
                 by the per-block partial sums of the router weight gradient */, float* d_wg, float* d_ln_w, float* d_ln_b,
                 void* stream);
size_t swn_gate_bwd_scratch_floats(int n_tokens, int gate_dim, int n_experts);
/* ... with DENSE operands on top (either may be NULL, not both): d_gates[s,e] += d_probs[s,e] (fp32 [n_tokens, n_experts]; the normalised
 * gates of a top-k layer, swn_topk_gate_bwd - tutel_fast_dispatch.py:204-206 under autograd), and d_logits_add[s,e] added to the logit
 * gradient BEHIND the softmax backward (the load / importance loss, swn_load_importance_bwd).  d_gmax may be NULL.  VALU kernel. */
int swn_gate_bwd_dense(const void* g, int dtype, const float* ln_w, const float* ln_b, const float* wg,
                       const float* gates, const int32_t* idx, const float* d_gmax, const float* d_probs,
                       const float* d_logits_add, const float* stats,
                       const int32_t* counts, const float* laux_coef, int seg_tokens,
                       int n_tokens, int gate_dim, int n_experts,
                       void* dg, float* dlogits, float* d_wg, float* d_ln_w, float* d_ln_b, void* stream);

/* ---- routing: batch-prioritised top-1 capacity assignment ----------------------------------------------------
 * replaces extract_critical / compute_sorted_location / load_balance (tutel_fast_dispatch.py:136-217) and the
 * Tutel `fast_cumsum_sub_one` kernel for top_k = 1.  Bit-exact contract: loc[i] = number of tokens of the same
 * segment and expert ranked before i; rank = stable descending gmax (bpr=1) or token order (bpr=0).
 *   idx[P] i32, gmax[P] f32, gates[P,E] f32 (for l_aux; may be NULL -> l_aux not computed)
 *   P = n_seg * seg_tokens (last segment may be short: n_tokens gives the true total)
 *   loc[P] i32; counts[n_seg,E] i32; perm[n_seg, E*capacity] i32 (row -> token, -1 = empty) or NULL;
 *   tok2row[P] i32 (token -> row seg*E*C + idx*C + loc, -1 = dropped) or NULL;
 *   l_aux[n_seg] f32 = sum_e(me*ce) * E / P_seg^2.
 *   workspace: swn_route_workspace_bytes(n_tokens) bytes.                                                      */
size_t swn_route_workspace_bytes(int n_tokens, int n_seg, int n_experts);
int swn_route_top1(const int32_t* idx, const float* gmax, const float* gates,
                   int n_tokens, int seg_tokens, int n_experts, int capacity, int bpr,
                   int32_t* loc, int32_t* counts, int32_t* perm, int32_t* tok2row, float* l_aux,
                   void* workspace, size_t workspace_bytes, void* stream);

/* The routing plus the list of dropped tokens (swn_route_dropped's outputs: both NULL or both given) from ONE call.  Replaces the same
 * reference code as swn_route_top1 (tutel_fast_dispatch.py:136-217) - every output is identical to swn_route_top1 + swn_route_dropped.
 *   mode: 0 = the per-phase kernels of swn_route_top1 (+ swn_route_dropped): 20 launches - the only mode of the product library.
 *         Modes 1 / 2 (fused phases / ONE launch with grid barriers: bit-identical, measured no faster / slower on this part - the XCDs'
 *         L2s are not coherent inside a kernel, profiles/r05_experiments.md 3) exist in the experiment build only
 *         (scripts/experiments/route_one.inc, build_route_one.sh); the product library rejects them.
 *   sync: int32 [swn_route_sync_bytes() / 4] of the experimental modes (zero before the first call, left zero); ignored in mode 0, may be NULL.
 *   workspace: swn_route_workspace_bytes() as for swn_route_top1.                                                                    */
size_t swn_route_sync_bytes(void);
int swn_route_top1x(const int32_t* idx, const float* gmax, const float* gates,
                    int n_tokens, int seg_tokens, int n_experts, int capacity, int bpr,
                    int32_t* loc, int32_t* counts, int32_t* perm, int32_t* tok2row, float* l_aux,
                    int32_t* drop_begin, int32_t* dropped, int32_t* sync, int mode,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ---- top-k routing, k > 1 (extract_critical, tutel_fast_dispatch.py:176-217 with top_k > 1; no shipped config sets `k` above 1) ----
 * swn_topk_select: torch.topk(gates, k, dim=1) per token (:177; descending, the lower expert on exact ties) -> idx int32 [k, n_tokens]
 *   (choice-major: row j = indices_s[j]), gsel fp32 [k, n_tokens] = gates_s before, gnorm = gates_s after the normalisation
 *   g_j / clamp(sum_j g_j, min=eps) (:204-206; k = 1: gnorm = gsel, `if top_k > 1`).
 * swn_route_topk: locations of every choice (:192-202).  Choice j ranks its tokens like swn_route_top1 (batch-prioritised: by the TOKEN's
 *   max gate, `importance_scores = -gates.max(dim=1)`, :187 - gmax is the top-1 gate for every choice; else token order) and adds
 *   acc_base = the counts of the choices before it (:199-201), so every (token, choice) owns one row of the [n_seg * E, capacity] row
 *   space, kept iff loc < capacity.  capacity = k * int(cf * ceil(P / E)) (:211) is the caller's.  loc int32 [k, n_tokens], counts int32
 *   [k, n_seg * E], perm [n_seg * E * capacity] row -> token (all choices; -1 = empty), tok2row [k, n_tokens] or NULL, group_rows
 *   [n_seg * E] = sum over the choices of counts (the chains' group_rows with group_rows_clamp = capacity), l_aux [n_seg] from choice
 *   0's mask (load_balance(gates, masks_se[0]), :184) or NULL.  workspace: swn_route_workspace_bytes().
 * swn_topk_gate_bwd: backward of the normalisation, d_gnorm [k, n_tokens] -> d_probs [n_tokens, E] (zero outside the token's k experts):
 *   the dense operand of swn_gate_bwd_dense.                                                                                        */
/* ---- load / importance loss (--use_load_importance_loss: load_importance_loss, tutel_fast_dispatch.py:152-174; needs gate_noise > 0) ----
 * swn_gate_logits: logits [n_tokens, E] = g @ wg^T (+ noise_scale * noise; fp32 router without LayerNorm, E <= 16): the loss compares
 *   the noise-free probabilities with the token's k-th largest NOISY LOGIT (`threshold = topk_logits[:, -1]`, :159-160).
 * swn_load_importance_fwd: l_loss [1] = (cv2(Imp) + cv2(Load)) / 2, Imp_e = sum_t scores[t,e], Load_e = sum_t Normal(0, sigma).cdf(
 *   scores[t,e] - logits_w_noise[t, idx_last[t]]), cv2(v) = v.var() / (v.mean()^2 + 1e-10), sigma = gate_noise / E; coef [2 E] receives
 *   dl/dImp_e, dl/dLoad_e for the backward.  workspace: swn_load_importance_workspace_floats() floats.  Sums in a fixed order.
 * swn_load_importance_bwd: d_logits [n_tokens, E] = dL/d logits through the probabilities and through the threshold entry, times the
 *   device scalar d_l_loss [1] - swn_gate_bwd_dense's d_logits_add.                                                                  */
int swn_gate_logits(const void* g, int dtype, const float* wg, const float* noise, float noise_scale, int n_tokens, int gate_dim,
                    int n_experts, float* logits, void* stream);
size_t swn_load_importance_workspace_floats(int n_tokens, int n_experts);
int swn_load_importance_fwd(const float* scores_wo_noise, const float* logits_w_noise, const int32_t* idx_last, float sigma,
                            int n_tokens, int n_experts, float* l_loss, float* coef, float* workspace, void* stream);
int swn_load_importance_bwd(const float* scores_wo_noise, const float* logits_w_noise, const int32_t* idx_last, const float* coef,
                            const float* d_l_loss, float sigma, int n_tokens, int n_experts, float* d_logits, void* stream);
int swn_topk_select(const float* gates, int n_tokens, int n_experts, int top_k, int32_t* idx, float* gsel, float* gnorm, void* stream);
int swn_route_topk(const int32_t* idx, const float* gmax, const float* gates, int n_tokens, int seg_tokens, int n_experts,
                   int capacity, int bpr, int top_k, int32_t* loc, int32_t* counts, int32_t* perm, int32_t* tok2row,
                   int32_t* group_rows, float* l_aux, void* workspace, size_t workspace_bytes, void* stream);
int swn_topk_gate_bwd(const float* gates, const int32_t* idx, const float* d_gnorm, int n_tokens, int n_experts, int top_k,
                      float* d_probs, void* stream);

/* ---- Tutel sparse kernel ABI (batched, capacity padded) ------------------------------------------------------
 * replaces tutel.jit_kernels.sparse func_fwd / func_bwd_data / func_bwd_gate as called from
 * tutel_fast_dispatch.py:27,36,43,61,70,76 - same argument order (gates, indices, locations, reshaped_input,
 * dispatched_input, extra=[samples, hidden, capacity]); gates may be NULL (= the reference's ones_helper).
 * Unlike the Tutel forward (atomicAdd into a zero-filled buffer) the forward here also writes the zeros of the
 * unused capacity slots when `counts` is given, so the caller does not have to memset.                          */
int swn_dispatch_fwd(const float* gates, const int32_t* indices, const int32_t* locations,
                     const void* reshaped_input, void* dispatched, int dtype,
                     int samples, int hidden, int capacity, int n_experts, void* stream);
int swn_dispatch_bwd_data(const float* gates, const int32_t* indices, const int32_t* locations,
                          void* grad_reshaped_input, const void* dispatched, int dtype,
                          int samples, int hidden, int capacity, void* stream);
int swn_dispatch_bwd_gate(float* grad_gates, const int32_t* indices, const int32_t* locations,
                          const void* reshaped_input, const void* dispatched, int dtype,
                          int samples, int hidden, int capacity, void* stream);

/* Top-k layers (k > 1): the reference calls the three kernels once per choice inside `for g, i, l in zip(gates_, indices_, locations_)`
 * (tutel_fast_dispatch.py:26-27, 34-37, 59-62, 69-70).  The first iteration is the entry point above; these are the later ones:
 * swn_dispatch_fwd_more writes a further choice's rows into the SAME dispatched buffer (no zero fill: the choices' rows are disjoint,
 * swn_route_topk), swn_dispatch_bwd_data_more ADDS g * dispatched[row] to grad_reshaped_input (`last_result + grad_data` /
 * `last_result + single_output`; dropped tokens add nothing).  swn_dispatch_bwd_gate is per choice as it is.                     */
int swn_dispatch_fwd_more(const float* gates, const int32_t* indices, const int32_t* locations,
                          const void* reshaped_input, void* dispatched, int dtype,
                          int samples, int hidden, int capacity, int n_experts, void* stream);
int swn_dispatch_bwd_data_more(const float* gates, const int32_t* indices, const int32_t* locations,
                               void* grad_reshaped_input, const void* dispatched, int dtype,
                               int samples, int hidden, int capacity, void* stream);

/* ---- the no-batch (evaluation) variants: tutel_sparse_nobatch.py:24-133 as called from tutel_fast_dispatch_nobatch.py:36, :47,
 * :53, :73, :87, :93 - the same argument order with `expert_locations_begin` (int32 [n_experts], exclusive prefix sum of
 * expert_input_nums) as 4th argument: row(i) = expert_locations_begin[indices[i]] + locations[i], rows packed contiguously per
 * expert, no capacity test (a token is skipped iff indices[i] < 0).  dispatched_rows = sum(expert_input_nums) (zero-filled first,
 * like the reference's torch.zeros).                                                                                        */
int swn_dispatch_nobatch_fwd(const float* gates, const int32_t* indices, const int32_t* locations,
                             const int32_t* expert_locations_begin, const void* reshaped_input, void* dispatched, int dtype,
                             int samples, int hidden, int capacity, int n_experts, long dispatched_rows, void* stream);
int swn_dispatch_nobatch_bwd_data(const float* gates, const int32_t* indices, const int32_t* locations,
                                  const int32_t* expert_locations_begin, void* grad_reshaped_input, const void* dispatched,
                                  int dtype, int samples, int hidden, int capacity, int n_experts, void* stream);
int swn_dispatch_nobatch_bwd_gate(float* grad_gates, const int32_t* indices, const int32_t* locations,
                                  const int32_t* expert_locations_begin, const void* reshaped_input, const void* dispatched,
                                  int dtype, int samples, int hidden, int capacity, int n_experts, void* stream);
/* The tokens no expert kept (locations >= capacity), listed per (segment, expert) in location order: drop_begin [n_groups + 1] receives
 * the exclusive prefix of max(counts - capacity, 0) over the groups (drop_begin[n_groups] = the number of dropped tokens), dropped
 * [n_tokens] the tokens (entries past the count are not written).  Input of the fused tail of swn_mlp_chain (tail_dropped).  */
int swn_route_dropped(const int32_t* idx, const int32_t* loc, const int32_t* counts, int n_tokens, int seg_tokens, int n_experts,
                      int capacity, int32_t* drop_begin, int32_t* dropped, void* stream);

/* packed (no-batch) row space from a routing: begin[n_seg * E] = exclusive prefix sum of counts (= expert_locations_begin,
 * per segment and expert), perm[n_tokens] row -> token, tok2row[n_tokens] token -> row (either may be NULL)                  */
int swn_route_pack(const int32_t* idx, const int32_t* loc, const int32_t* counts, int n_tokens, int seg_tokens, int n_experts,
                   int32_t* begin, int32_t* perm, int32_t* tok2row, void* stream);

/* Fast-path combine (the reference's decode + the MoE layer's `act: relu`, models/nerf_moe.py:385-386):
 * y[i] = relu?(gate[i] * expert_out[seg(i)*E*C + idx*C + loc]), zero rows for dropped tokens.                 */
int swn_combine_fwd(const float* gates, const int32_t* indices, const int32_t* locations, void* y,
                    const void* expert_out, int dtype, int samples, int hidden, int capacity, int seg_tokens,
                    int n_experts, int relu, void* stream);
/* Backward of the above w.r.t. expert_out and gate, given dy_in = dL/dy (+ optional rank-1 term dsig[i]*wsig[:],
 * the sigma head's contribution): dy = (dy_in + dsig*wsig) * (y > 0); dgate[i] = <y_i, dy_i> / gate[i];
 * dout[i] = dy_i * gate[i] (token order; the expert backward gathers it through `perm`).                      */
int swn_combine_bwd(const void* dy_in, const void* y, const float* dsig, const float* wsig, const float* gate,
                    int dtype, int samples, int hidden, void* dout, float* dgate, void* stream);

/* ---- sigma / colour heads ----------------------------------------------------------------------------------------
 * replaces models/nerf_moe.py:393-416 (sigma Linear + noise + ShiftedSoftplus) and :431-441 (colour Linear + sigmoid).
 * raw[P,4] f32 = (rgb, sigma).  w_sigma[M], b_sigma[1], w_color[3,H2], b_color[3] f32.                         */
int swn_heads_fwd(const void* y, const void* h2, int dtype, const float* w_sigma, const float* b_sigma,
                  const float* w_color, const float* b_color, const float* sigma_noise, int n_points,
                  int model_dim, int h2_dim, float* raw, void* stream);
/* Backward: dh2, dsig (per point) and the four parameter gradients ACCUMULATED (+=) into d_w_sigma[M], d_b_sigma[1], d_w_color[3,H2],
 * d_b_color[3].  The parameter gradients are block partial sums in `workspace` (swn_heads_bwd_workspace_bytes) added in a fixed order:
 * the same bits on every run.  y == NULL: the sigma weight gradient is not formed here (d_w_sigma receives + 0; a fused backward
 * chain forms it where y is read anyway: swn_chain_desc.comb_dwsig) - the launch then reads 512 bytes per point less.           */
size_t swn_heads_bwd_workspace_bytes(int n_points, int model_dim, int h2_dim);
int swn_heads_bwd(const void* y, const void* h2, int dtype, const float* w_color, const float* raw,
                  const float* d_raw, int n_points, int model_dim, int h2_dim, void* dh2, float* dsig,
                  float* d_w_sigma, float* d_b_sigma, float* d_w_color, float* d_b_color, int rows_per_group,
                  float* group_colsum, void* workspace, size_t workspace_bytes, void* stream);
/* rows_per_group > 0 (the samples per ray; must divide n_points): also group_colsum[n_points / rows_per_group][h2_dim] f32 = the column
 * sums of each group's dh2 rows as stored - what swn_group_colsum(dh2, ...) returns (the per-ray bias gradient, nerf_moe.py:419-429)
 * without reading dh2 back.  0: group_colsum is not touched.                                                          */
/* out[g][c] = sum_r in[g*rows_per_group + r][c]  (per-ray bias gradient) */
int swn_group_colsum(const void* in, int dtype, int n_groups, int rows_per_group, int cols, float* out, void* stream);

/* ---- volumetric compositing ------------------------------------------------------------------------------------
 * replaces rendering.py:435-494 (and rendering_mip.py:380-425).  raw[N,S,4] f32 (rgb, sigma); z[N,S] f32; last_delta
 * scalar (1e10); rgb_padding: the colours are widened to rgb * (1 + 2 pad) - pad first (rendering_mip.py:383-384; 0 = off).
 * rgb[N,3], depth[N], depth_var[N], weights[N,S] (any may be NULL).                                             */
int swn_composite_fwd(const float* raw, const float* z, float last_delta, float rgb_padding, int n_rays, int n_samples,
                      float* rgb, float* depth, float* depth_var, float* weights, void* stream);
/* d_raw[N,S,4] = gradient of sum(d_rgb * rgb) w.r.t. raw. */
int swn_composite_bwd(const float* raw, const float* z, float last_delta, float rgb_padding, const float* d_rgb,
                      int n_rays, int n_samples, float* d_raw, void* stream);

/* ---- mip path (rendering_mip.py, MipNeRFMoE) -------------------------------------------------------------------
 * swn_sample_z: the n_samples interval edges of a level, z = near (1 - t) + far t with the stratified perturbation of
 *   rendering.py:573-584 (perturb_rand [N,S] U[0,1) supplied by the caller, NULL / perturb = 0: none).
 * swn_mip_encode: replaces mip_cast_rays (rendering_mip.py:15-25) + MipEmbedder (models/nerf.py:28-56): for every ray and
 *   every pair of consecutive edges z[i], z[i+1] the conical frustum's mean / diagonal covariance and from them the
 *   integrated positional encoding [mean, sin(2^k mean) exp(-4^k var / 2), cos(2^k mean) exp(-4^k var / 2)]_k<l_xyz,
 *   zero-padded to pe_stride: pe [n_rays * (n_edges - 1), pe_stride] dtype.  radii [n_rays] f32.
 * swn_mip_resample: replaces the level hand-over rendering_mip.py:206-223 (weights blur + weights_resample_padding) and
 *   sorted_piecewise_constant_pdf1 (:75-131): weights [N, n_edges - 1] of the level -> n_fine new edges per ray (sorted).
 *   u_rand [N, n_fine] U[0,1) = the tensor of the randomized branch (:103), NULL = deterministic linspace (:110).        */
int swn_sample_z(const float* rays, const float* t_steps, const float* perturb_rand, float perturb, int n_rays, int n_samples,
                 float* z_out, void* stream);
int swn_mip_encode(const float* rays, const float* radii, const float* z, int n_rays, int n_edges, int l_xyz, int dtype,
                   void* pe, int pe_stride, void* stream);
int swn_mip_resample(const float* z, const float* weights, const float* u_rand, float padding, int n_rays, int n_edges,
                     int n_fine, float* z_out, void* stream);

/* ---- multiresolution hash-grid input encoding (BASELINE.json configs[4]; NOT in the reference: own definition) ----------
 * Algorithm: Mueller et al., "Instant Neural Graphics Primitives with a Multiresolution Hash Encoding" (2022), section 3.
 * Conventions of this library (restated on the CPU in oracle/switchnerf_oracle.py hash_encode):
 *   x' = clamp((x - aabb_lo) / (aabb_hi - aabb_lo), 0, 1) with x = o + d * z;
 *   level l < n_levels: scale_l = base_res * per_level_scale^l - 1 (evaluated in double, rounded to float),
 *     R_l = ceil(scale_l) + 2 grid points per axis; pos = x' * scale_l + 0.5; cell = floor(pos); w = pos - cell;
 *     corner c = cell + {0,1}^3 -> entry = cx + R_l * (cy + R_l * cz)                      if R_l^3 <= T = 2^log2_table
 *                                  entry = (cx ^ cy * 2654435761 ^ cz * 805459861) mod T   otherwise (uint32 arithmetic);
 *   feature pair l = sum over the 8 corners of the trilinear weight * table[l][entry][0..1].
 * table: fp32 [n_levels, T, 2].  out: [n_rays * n_samples, out_stride] of `dtype`, columns [0, 2 n_levels) = the features,
 * the rest zero (out_stride <= 128, a multiple of 16 bytes).  The backward adds dL/d table (fp32 atomics) given
 * d_out [n_rays * n_samples, d_stride]; positions carry no gradient (the rays are data).                              */
typedef struct {
  int n_levels;          /* <= 16 */
  int log2_table;        /* T = 2^log2_table entries of 2 features per level */
  int base_res;
  float per_level_scale;
  float aabb_lo[3], aabb_hi[3];
} swn_hash_cfg;
int swn_hash_encode_fwd(const float* rays, const float* z, int n_rays, int n_samples, const swn_hash_cfg* cfg,
                        const float* table, int dtype, void* out, int out_stride, void* stream);
int swn_hash_encode_bwd(const float* rays, const float* z, int n_rays, int n_samples, const swn_hash_cfg* cfg,
                        const void* d_out, int dtype, int d_stride, float* d_table, void* stream);
/* The same with one private copy of the gradient table per XCD (xcd_tables: fp32 [8][n_levels][T][2], ZERO on entry and left zero): a
 * workgroup's atomics go to the copy of the XCD it runs on (HW_REG_XCC_ID) and a second kernel adds the copies into d_table.        */
int swn_hash_encode_bwd_xcd(const float* rays, const float* z, int n_rays, int n_samples, const swn_hash_cfg* cfg,
                            const void* d_out, int dtype, int d_stride, float* d_table, float* xcd_tables, void* stream);
/* The same gradient WITHOUT global float atomics, bit-deterministic (round 6): every (corner, level) contribution is an item binned by
 * table tile (2^13 entries) into `workspace`, then one workgroup per tile adds its items in fixed point (int64, scaled by the power of
 * two of max |d_out|: an item is exact to 2^-38 of the largest gradient - tighter than an fp32 running sum) in LDS and adds the tile into
 * d_table as its only writer.  Four launches (count, scan, scatter, tiles) + two small fills; 12 bytes of workspace traffic per item each
 * way.  Tables of more than 2^22 entries per level are rejected (use swn_hash_encode_bwd).  A non-finite d_out makes the touched entries NaN.
 *   workspace: swn_hash_bwd_workspace_bytes(n_rays * n_samples, cfg) bytes (3.2 GB for 2M points x 16 levels), any contents.          */
size_t swn_hash_bwd_workspace_bytes(long n_points, const swn_hash_cfg* cfg);
int swn_hash_encode_bwd_binned(const float* rays, const float* z, int n_rays, int n_samples, const swn_hash_cfg* cfg,
                               const void* d_out, int dtype, int d_stride, float* d_table, void* workspace, size_t workspace_bytes,
                               void* stream);

/* ---- background model + foreground bound (render_rays' bg_nerf branch, /root/reference/switch_nerf/rendering.py:32-159) ---
 * swn_fg_bounds: _intersect_sphere (:497-518) per ray against the ellipsoid (center, radius: 3 floats each in HOST memory,
 *   both NULL = the unit sphere) and the bookkeeping of :34-44:
 *     fg_far[r]     = max(intersection depth, near)
 *     has_bg[r]     = far > fg_far          (the rays `rays_with_bg` that continue into the background model)
 *     last_delta[r] = has_bg ? fg_far : 1e10 (before :216-217 subtracts the ray's last sample depth)
 *     rays_fg[r]    = the ray with far = min(far, fg_far)
 *   *n_outside (int32, zeroed by the caller) counts rays whose closest approach lies outside the unit sphere - the
 *   reference raises "Not all your cameras are bounded by the unit sphere" (:515-517); the host mirror does the same.
 * swn_bg_sample_pe: the background samples of `n_rays` (already gathered) rays:
 *   z_in == NULL: n_samples stratified inverse-distance depths in [0,1] (t_steps = linspace(0,1,n_samples) on the device,
 *     perturb_rand [n_rays, n_samples] in ascending-sample order or NULL, _expand_and_perturb_z_vals :573-584), evaluated in
 *     the flipped (descending) order of :302-304: z_out[r, j] and PE row r * n_samples + j belong to ascending sample
 *     n_samples - 1 - j, while depth_real[r, a] stays in ascending order a exactly like the reference's un-flipped tensor;
 *   z_in != NULL: the caller's depths (hierarchical pass :246), everything in the order of z_in.
 *   Points: _depth2pts_outside (:521-570, include_xyz_real False): unit-sphere point rotated by Rodrigues' formula + the
 *   inverse distance -> 4-D, encoded with l_xyz octaves (4 + 8 l_xyz columns, zero-padded to pe_stride elements of `dtype`).
 * swn_composite_bounded_fwd / _bwd: volumetric compositing (:435-494) with a per-ray last delta (last_delta [n_rays] or
 *   NULL = 1e10), flip != 0 for descending depths (:436-437), depth_real [n_rays, n_samples] or NULL as the depth map's
 *   source (:483-484) and bg_lambda [n_rays] or NULL = the transmittance behind the last sample (:456-457); the backward
 *   takes dL/d bg_lambda [n_rays] or NULL in addition to dL/d rgb.                                                        */
int swn_fg_bounds(const float* rays, const float* center_host, const float* radius_host, int n_rays, float* rays_fg,
                  float* fg_far, float* last_delta, int32_t* has_bg, int32_t* n_outside, void* stream);
int swn_bg_sample_pe(const float* rays, const float* center_host, const float* radius_host, const float* t_steps,
                     const float* perturb_rand, float perturb, int n_rays, int n_samples, int l_xyz, int dtype,
                     const float* z_in, float* z_out, float* depth_real, void* pe, int pe_stride, void* stream);
int swn_composite_bounded_fwd(const float* raw, const float* z, const float* last_delta, int flip, const float* depth_real,
                              int n_rays, int n_samples, float* rgb, float* depth, float* depth_var, float* weights,
                              float* bg_lambda, void* stream);
int swn_composite_bounded_bwd(const float* raw, const float* z, const float* last_delta, int flip, const float* d_rgb,
                              const float* d_bg_lambda, int n_rays, int n_samples, float* d_raw, void* stream);

/* dst[r] = src[index[r]] for r < n_rows (rows of row_bytes bytes, a multiple of 16), zero rows where index[r] < 0.
 * Builds the send buffer of the expert-parallel token exchange (the reference's all-to-all payload,
 * tutel_moe_layer_nobatch.py:157, 172) from the routing permutation.                                            */
int swn_gather_rows(const void* src, const int32_t* index, long n_rows, int row_bytes, void* dst, void* stream);
/* Sign bits of a 16-bit activation matrix [rows, features] (features a multiple of 32): bit j of word q of a row = (h[row][32 q + j] > 0),
 * and back into a 0 / 1 matrix of the library's 16-bit type.  Expert parallelism with the tail on the expert's rank (ep_owner.py; replaces
 * what tutel_moe_layer_nobatch.py:172-185 sends home): the per-ray bias gradient needs of layer "2"'s output only its ReLU mask - 16 bytes
 * per token at 128 features travel instead of the 256-byte row.                                                                        */
int swn_sign_bits_pack(const void* h, long rows, int features, uint32_t* bits, void* stream);
int swn_sign_bits_unpack(const uint32_t* bits, long rows, int features, void* h, void* stream);
/* dst[index[r]] = src[r] for rows of row_bytes (a multiple of 4; index < 0: the row goes nowhere; no index twice): the mirror of
 * swn_gather_rows - what an exchange brought home in send order goes back to token order (the combine's scatter of
 * tutel_moe_layer_nobatch.py:220-225 for the 16-byte results of the owner-tail mode).                                             */
int swn_scatter_rows(const void* src, const int32_t* index, long n_rows, int row_bytes, void* dst, void* stream);
/* The 16-byte record that travels with a kept token's row to its expert's rank (ep_owner.py):
 *   aux[r] = (gate[t], int bits of (t / rows_per_ray + ray_base), noise[t] (0 if NULL), 0),  t = index[r]     (zero_gate != 0: gate 0)
 * and its split into the fused launch's planar per-token operands on the receiving side (noise may be NULL).                      */
int swn_owner_aux(const float* gate, const float* noise, const int32_t* index, long n, int rows_per_ray, int ray_base, int zero_gate,
                  float* aux, void* stream);
int swn_owner_aux_split(const float* aux, long n, float* gate, int32_t* ray, float* noise, void* stream);
/* The per-ray bias gradient of layer "2" (nerf_moe.py:419-429: the PE(dir) / embedding_a half of its input is per ray) from what the
 * owner-tail mode brings home: dc_ray[ray][f] = sum over the ray's rows_per_ray tokens of (bit f of bits[token]) * round16(dc0 wc[0][f] +
 * dc1 wc[1][f] + dc2 wc[2][f]), dc_c = d_raw[token][c] raw[token][c] (1 - raw[token][c]) - swn_heads_bwd's dh2 and its per-ray sums without h2.
 * bits: swn_sign_bits_pack's words [tokens][features / 32]; w_color fp32 [3][features]; dc_ray fp32 [n_rays][features].               */
int swn_ray_bias_grad_bits(const uint32_t* bits, const float* raw, const float* d_raw, const float* w_color, int n_rays, int rows_per_ray,
                           int features, float* dc_ray, void* stream);

/* ---- dense / grouped MLP chains on MFMA ------------------------------------------------------------------------
 * One launch runs `n_layers` (<= SWN_MAX_CHAIN_LAYERS) Linear layers back to back with the activations of a 128-row tile resident in
 * LDS.  Layer l: h <- act_l( h @ W_l^T + b_l [+ rowbias[row/rows_per_bias]] [+ skip] ).  W_l (logically [N_l][K_l],
 * torch.nn.Linear.weight orientation) must be given in the MFMA-fragment-major layout produced by swn_pack_weights
 * (compute dtype); b_l fp32.  Ragged groups: rows of group g are
 * [g*group_stride, g*group_stride + group_rows[g]) and group g uses weight set g % n_wsets
 * (expert MLP: group = (segment, expert), n_wsets = E).
 * replaces ExpertMLP.forward (tutel_moe_layer_nobatch.py:887-924: baddbmm chain, skip, ReLU) and Mlp.forward
 * (models/nerf_moe.py:30-49).  Details of the descriptor: see swn_chain_desc below.                             */
#define SWN_MAX_CHAIN_LAYERS 12
typedef struct swn_chain_layer {
  const void* w;        /* [n_wsets] x packed(N, K) in dtype: output of swn_pack_weights */
  const float* b;       /* [n_wsets][N] f32 or NULL                                */
  void* save;           /* row-major [rows_total][N] dtype: output of this layer (post activation) or NULL */
  uint32_t* mask;       /* packed ReLU mask: written when relu==1, read when relu==2; size =
                           n_workgroups * 8 waves * MI * 64 uint32 (MI = 2 bf16 / 1 fp32), or NULL */
  const float* rowbias; /* f32 [n_rows / rows_per_bias][N] extra bias shared by runs of rows (per-ray terms) or NULL */
  int32_t rows_per_bias;
  int32_t n, k;         /* output / input features: n a multiple of 64 up to 512, k in {64,128,256,512}; any layer wider than
                           256 selects the 512-feature kernels for the whole chain                                */
  int32_t relu;         /* 0 none; 1 ReLU (and record the mask if given); 2 multiply by the recorded mask (backward) */
  int32_t skip;         /* 1: add the chain input x before the activation (residual; needs n == layers[0].k).
                         * 2: first half of a concat-skip Linear(cat([x, h])) = h W_h + x W_x (models/nerf.py:155-156): this
                         *    entry is h W_h - no bias / activation / save; the NEXT entry (k = layers[0].k, same n) multiplies
                         *    the re-staged chain input and carries the bias, activation, mask and save of the layer */
} swn_chain_layer;

typedef struct swn_chain_desc {
  int32_t dtype, n_layers, n_groups, n_wsets;
  int32_t group_stride;         /* rows reserved per group in the row space                      */
  const int32_t* group_rows;    /* device [n_groups] valid rows per group, or NULL = group_stride (then clamp) */
  int32_t group_rows_clamp;     /* rows valid = min(group_rows[g], clamp)                         */
  const int32_t* group_begin;   /* device [n_groups] first row of every group (packed / no-batch layout: exclusive prefix sum of
                                   the group sizes, tutel_fast_dispatch_nobatch.py:24-36) or NULL = g * group_stride           */
  const void* x;                /* chain input, row-major [*, k0] dtype                           */
  const int32_t* x_gather;      /* device [n_groups*group_stride] row -> source row of x (-1 = zero row), or NULL */
  void* x_save;                 /* optional copy of the (gathered, scaled) input rows (row-major) or NULL */
  const float* x_scale;         /* per destination row scale applied to the gathered input (the gate value: fused
                                   GatingDecoder, tutel_fast_dispatch.py:50-63), or NULL             */
  int32_t x_relu;               /* ReLU after scaling (the MoE layer's `act: relu`, nerf_moe.py:385)  */
  void* y;                      /* output rows, row-major [*, n_last] dtype                       */
  const void* y_add;            /* row-major [*, n_last] tensor added to the output rows (skip gradient) or NULL */
  const int32_t* y_add_gather;  /* row -> row of y_add (-1 = nothing to add), or NULL (identity)  */
  int32_t geometry;             /* 0 / 1 = the 64-row tile kernels (chain.hip); 2 = one 256-row workgroup per CU, 3 = two 96-row
                                   workgroups per CU, 4 = the 256-row workgroup with its two row groups half a layer apart
                                   (one group's epilogue beside the other's K loop; the accumulators start at the bias), 5 = the
                                   same with the bias added in the epilogue like every other geometry (bit-identical to them)
                                   6 / 7 = geometry 5 / 4 as a PERSISTENT launch: one resident workgroup per CU walks a queue of
                                   256-row tiles, a row group stages its next tile and writes its last one out while its partner
                                   computes (no per-tile prologue / tail / fill phase; `sched` below)
                                   (chain_big.hip: chains of 256 x 256 layers, bf16 / fp16, no rowbias / x_scale / x_save /
                                   y_add_gather; swn_chain_big_ok).
                                   The ReLU masks of the 64-row, 96-row and 256-row tiles are laid out differently (2, 4, 5, 6 and
                                   7 share one layout): run a backward chain (relu = 2) on the tile geometry of the forward chain
                                   that recorded its masks.                                                                   */
  int32_t tag;                  /* profiling only: selects an identical kernel instantiation with its own symbol so that
                                   rocprofv3 reports the roles separately (0 generic, 1 expert fwd, 2 expert bwd,
                                   3 front fwd, 4 tail fwd, 5 tail bwd, 6 front bwd; 7 = the expert forward chain WITH the fused tail, tail_first below; 8 = the expert backward chain behind the tail's backward layers, head_layers below)                 */
  /* Combine backward fused into the write-out of the LAST layer (comb_y != NULL; the tail backward chain): with z = the layer's
     output row (the gradient of the decoded, gate-scaled, ReLU'd expert output y, tutel_fast_dispatch.py:50-63 + nerf_moe.py:385)
       t = (z + comb_dsig[row] * comb_wsig) * (comb_y[row] > 0);   y[row] = t * comb_gate[row];   comb_dgate[row] = <comb_y[row], t> / comb_gate[row]
     i.e. swn_combine_bwd without the round trip of z through memory.  comb_y: row-major [*, n_last] dtype; comb_dsig (NULL = 0),
     comb_gate, comb_dgate: fp32 per row; comb_wsig: fp32 [n_last] (NULL = 0).  n_last must be 128, 256 or 512; tag must be 5 (only that kernel
     instantiation carries the code).                                                                                              */
  const void* comb_y;
  const float* comb_dsig;
  const float* comb_wsig;
  const float* comb_gate;
  float* comb_dgate;
  /* (head_layers > 0 only) The sigma head's WEIGHT gradient from the same pass - the combine backward holds comb_y[row] and
     comb_dsig[row] of every kept token in registers:  comb_dwsig[f] += sum over the tokens of comb_dsig[token] * comb_y[token][f]
     (fp32 [256]; nerf_moe.py:393-400 sigma = Linear(256, 1): its weight.grad; the dropped tokens' y is zero).  Summed in a FIXED order
     whatever workgroup ran a tile: a wave leaves the sum of its 32 rows in comb_dwsig_ws[(tile * 8 + wave)][256], the launch then adds
     the rows up in order (two small kernels behind the chain kernel).  comb_dwsig_ws: fp32 workspace of
     swn_chain_dwsig_workspace_bytes(n_groups, group_rows_clamp) bytes (zeroed by the launch); both NULL = not computed (swn_heads_bwd
     forms the gradient from y then).                                                                                              */
  float* comb_dwsig;
  float* comb_dwsig_ws;
  /* Sigma / colour heads fused into the tail FORWARD chain (heads_raw != NULL; tag must be 4, geometry 0 / 1): with y = the staged
     chain input row (gathered, gate-scaled, ReLU'd: what x_save would hold) and h2 = the last layer's output row,
       heads_raw[row] = (sigmoid(<h2, heads_wc[c]> + heads_bc[c]) c < 3, softplus(<y, heads_ws> + heads_bs[0] + heads_noise[row] - 1))
     i.e. swn_heads_fwd (models/nerf_moe.py:393-441) without reading y and h2 back from memory.  heads_ws fp32 [k0], heads_wc fp32
     [3][n_last], heads_noise fp32 per row or NULL; chain input of 256 / 512 features, last layer of 128 / 256.  With the heads fused
     `y` (and x_save) may be NULL: an inference forward then writes nothing but raw.                                             */
  const float* heads_ws;
  const float* heads_bs;
  const float* heads_wc;
  const float* heads_bc;
  const float* heads_noise;
  float* heads_raw;
  int32_t* sched;               /* geometry 6 / 7: device int32 [16] tile-queue counters, ZERO before the first launch that uses them
                                   (the kernel leaves them zero); launches that may run concurrently need their own.  NULL: the tiles
                                   are dealt round-robin over the resident workgroups (no balancing of ragged groups)          */
  int32_t x_features;           /* geometry 6 / 7: features per row of x when fewer than layers[0].k: 128 under a first layer whose packed
                                   weights are zero-padded from k = 128 to k = 256 (the rows of x are 128 features wide; the kernel zeroes
                                   the other half of its input tile).  0 = layers[0].k                                         */
  /* The dense TAIL of the network folded into the expert FORWARD chain (tail_first > 0; geometry 7, tag 7, x_gather required): the
     expert output never travels to memory and back between the MoE layer and the layers behind it (models/nerf_moe.py:385-441:
     GatingDecoder -> act relu -> sigma head, Linear "1", Linear "2" over cat([h, PE(dir), embedding_a]), colour head).
       * x_gather maps a row to its TOKEN (= its source row of x, swn_route_top1's perm); tokens are the row index of everything below;
       * behind layer tail_first - 1 (the last expert layer) a row becomes z = relu(tail_gate[token] * z) - rounded to dtype before and
         after the scaling, the arithmetic of x_scale / x_relu on the 64-row tail chain - and that is what layers[tail_first - 1].save
         receives;
       * layers[tail_first ..] are SHARED layers (weight set 0 for every group); the LAST layer may carry a rowbias (fp32
         [tail_tokens / rows_per_bias][y_features], indexed by token / rows_per_bias) and be narrower than 256 features: y_features = 128
         with its packed weights zero-padded to n = 256 (swn_pack_item.out_cols);
       * layers[l].save for l >= tail_first - 1, y ([tail_tokens][y_features]), heads_raw and heads_noise are in TOKEN order;
       * tail_dropped[0 .. *tail_n_dropped) (swn_route_dropped) lists the tokens no expert kept: they enter at layer tail_first as zero
         rows (and are saved as zero rows of layers[tail_first - 1].save);
       * heads_* as for tag 4 (optional), y / the saves may be NULL (an inference forward writes nothing but heads_raw);
       * tail_tokens * 512 bytes must stay below 4 GiB (the scattered stores carry 32-bit offsets).                                 */
  /* The mirror image for the BACKWARD-DATA pass (head_layers > 0; geometry 7, tag 8, x_gather required, x_features = 128): the tail's
     backward layers in FRONT of the expert backward chain.  x = dh2 in TOKEN order (128 features under a first layer zero-padded in K),
     layers[0 .. head_layers) are shared layers (weight set 0), layers[l].save for l < head_layers - 1 is in TOKEN order (dh1); behind
     layer head_layers - 1 the combine backward runs on the row (comb_* above, all indexed by TOKEN here: comb_y [tail_tokens][256],
     comb_dsig, comb_gate, comb_dgate per token - the tokens tail_dropped lists get comb_dgate = 0), whose result is what
     layers[head_layers - 1].save receives in the ROW space (the last expert layer's dZ for the weight gradients) and what the expert
     layers behind it consume; y / y_add / the other saves are the expert backward chain's (row space).  The dropped tokens run the
     head layers only.                                                                                                              */
  int32_t head_layers;
  int32_t tail_first;
  int32_t y_features;           /* real width of the last layer (tail mode): 128 or 256; 0 = layers[n_layers - 1].n              */
  const float* tail_gate;       /* fp32 [tail_tokens]                                                                              */
  const int32_t* tail_dropped;  /* device int32 [tail_dropped_max]                                                                  */
  const int32_t* tail_n_dropped;/* device int32 scalar                                                                              */
  int32_t tail_dropped_max;
  int32_t tail_tokens;
  const int32_t* tail_bias_row; /* (tail_first > 0) device int32 [tail_tokens] or NULL: the row of the last layer's rowbias that token t adds,
                                   instead of t / rows_per_bias - a token space that is not in ray order (expert parallelism with the tail on
                                   the expert's rank, ep_owner.py: the tokens an owner received from every rank, rowbias = the all-gathered
                                   per-ray terms)                                                                                       */
  swn_chain_layer layers[SWN_MAX_CHAIN_LAYERS];
} swn_chain_desc;

int swn_mlp_chain(const swn_chain_desc* desc, void* stream);
/* bytes of swn_chain_desc.comb_dwsig_ws for a fused backward chain over n_groups groups of at most group_rows_clamp rows */
size_t swn_chain_dwsig_workspace_bytes(int n_groups, int group_rows_clamp);
/* 1 if the chain can run on the 256-row geometry (geometry = 2) */
int swn_chain_big_ok(const swn_chain_desc* desc);
/* rows per workgroup tile for dtype (sizes the ReLU mask buffers: ceil(group_stride / rows) * n_groups * rows * 8 words) */
int swn_chain_tile_rows(int dtype);
/* uint32 words of one ReLU mask buffer of a chain over n_groups x group_stride rows whose widest layer has max_width features
 * (layers wider than 256 features run on the 512-feature kernels, whose tiles carry twice the bits per row).           */
long swn_chain_mask_words(int dtype, int n_groups, int group_stride, int max_width);

/* Pack fp32 master weights [n_wsets][in_dim][out_dim] (the reference's ExpertMLP layout, tutel_moe_layer_nobatch.py:853)
 * into the compute copy swn_mlp_chain consumes.  transpose = 1: forward weights (N = out, K = in);
 * transpose = 0: backward-data weights (N = in, K = out).  Output: n_wsets * in_dim * out_dim elements of dtype,
 * ordered [wset][N/32][K/16][64 lanes][8] (bf16) or [wset][N/32][K/8][64 lanes][4] (fp32) so that one MFMA operand
 * fragment is one contiguous 1 KiB wave load.                                                                     */
int swn_pack_weights(const float* master, void* out, int dtype, int n_wsets, int in_dim, int out_dim, int transpose,
                     void* stream);

/* The same for up to SWN_MAX_PACK_ITEMS weights in one launch (the per-step refresh of all compute copies after the optimizer). */
#define SWN_MAX_PACK_ITEMS 32
typedef struct swn_pack_item {
  const float* master;  /* [n_wsets][in_dim][out_dim] f32 */
  void* out;            /* packed compute copy (dtype of the call) */
  int32_t n_wsets, in_dim, out_dim, transpose;
  int32_t in_rows;      /* rows / columns of `master` actually stored when fewer than in_dim / out_dim: master is                     */
  int32_t out_cols;     /* [n_wsets][in_rows][out_cols] and the packed copy is zero-padded to (in_dim, out_dim) - a 128-feature first
                           layer (forward: in_rows = 128 under in_dim = 256; backward-data of a 128-output layer: out_cols = 128 under
                           out_dim = 256) for the K = 256 kernels of chain geometries 6 / 7.  0 = in_dim / out_dim                   */
} swn_pack_item;
int swn_pack_weights_batched(const swn_pack_item* items, int n_items, int dtype, void* stream);

/* Grouped weight gradient: for every group g, dW[g % n_wsets] += A_g^T @ B_g (fp32 atomics), and optionally
 * db += column sums of B_g.  A[rows, m_dim], B[rows, n_dim] row-major dtype; dW [n_wsets][m_dim][n_dim] f32.
 * With A = layer input, B = dZ this yields the reference's ExpertMLP weight layout [E, in, out]
 * (tutel_moe_layer_nobatch.py:853); with A = dZ, B = input it yields torch.nn.Linear's [out, in].
 * a_gather / b_gather (device int32 [n_groups * group_stride], or NULL): row r of a group reads source row
 * gather[g * group_stride + r] of A / B instead of row g * group_stride + r - the operand is taken through the routing
 * permutation (swn_route_top1's perm) rather than from a dispatched copy, which then never has to be written.          */
int swn_wgrad(const void* a, const void* b, const int32_t* a_gather, const int32_t* b_gather, int dtype, int m_dim, int n_dim,
              int n_groups, int n_wsets, int group_stride, const int32_t* group_rows, int group_rows_clamp,
              float* dw, float* db, int n_splits, int tag /* profiling only: 0 generic, 1 expert */,
              void* workspace, size_t workspace_bytes, void* stream);
/* Up to 8 such GEMMs of identical shape and grouping in ONE launch (grid z): the weight gradients of all layers of the
 * ragged expert MLP - a group with few rows finishes early, and only a launch that holds every layer's workgroups lets the
 * hardware fill the freed CUs (per-layer launches run as long as their fullest group).  workspace: n_items x the size below. */
typedef struct swn_wgrad_item {
  const void* a;
  const void* b;
  const int32_t* a_gather;
  const int32_t* b_gather;
  float* dw;
  float* db;
} swn_wgrad_item;
int swn_wgrad_batched(const swn_wgrad_item* items, int n_items, int dtype, int m_dim, int n_dim,
                      int n_groups, int n_wsets, int group_stride, const int32_t* group_rows, int group_rows_clamp,
                      int n_splits, int tag, void* workspace, size_t workspace_bytes, void* stream);
/* The same with the items being column BLOCKS of wider operands (layers of more than 256 features): item.a / item.b / item.dw point
 * at the block's first column, lda / ldb / ldw are the row strides (elements) of the full matrices, dw_set_stride / db_set_stride
 * the elements between consecutive weight sets of the full dW / db; give db only to the items of one row block of dW.            */
int swn_wgrad_blocks(const swn_wgrad_item* items, int n_items, int dtype, int m_dim, int n_dim, int lda, int ldb, int ldw,
                     size_t dw_set_stride, size_t db_set_stride, int n_groups, int n_wsets, int group_stride,
                     const int32_t* group_rows, int group_rows_clamp, int n_splits, int tag, void* workspace,
                     size_t workspace_bytes, void* stream);
/* Balanced form (round 3): up to 8 JOBS - one GEMM each, with its own operands, widths (multiples of 32 up to 256) and strides - over
 * ONE common row grouping, in one launch whose work follows the valid rows: the rows of all jobs are cut into equal shares, one per
 * CU, whatever the groups' fill (a full (segment, expert) group no longer is the long pole of the launch), a workgroup hands over one
 * partial tile per (job, weight set) it touches and a second kernel adds them in a fixed order (deterministic, no atomics).
 * Requires n_groups % n_wsets == 0 (group g uses weight set g % n_wsets), n_groups <= 2048, and a workspace of
 * swn_wgrad_multi_workspace_bytes(n_jobs, n_wsets).  swn_wgrad / swn_wgrad_batched / swn_wgrad_blocks take this path themselves when
 * their workspace is large enough (n_splits is ignored then).  dw / db: 16-byte aligned, ldw and the set strides multiples of 4.
 * Replaces the backward of torch.baddbmm w.r.t. the expert weights for ALL layers of ExpertMLP (tutel_moe_layer_nobatch.py:887-924)
 * and of F.linear for the dense Mlp layers (models/nerf_moe.py:30-49).                                                            */
typedef struct swn_wgrad_job {
  const void* a;             /* [rows, lda] dtype: layer input  */
  const void* b;             /* [rows, ldb] dtype: dZ           */
  const int32_t* a_gather;   /* or NULL (see swn_wgrad)         */
  const int32_t* b_gather;
  float* dw;                 /* [n_wsets][m_dim][ldw] f32, accumulated into */
  float* db;                 /* [n_wsets][n_dim] f32 or NULL                */
  size_t dw_set_stride, db_set_stride;   /* elements between weight sets    */
  int32_t m_dim, n_dim, lda, ldb, ldw;
} swn_wgrad_job;
size_t swn_wgrad_multi_workspace_bytes(int n_jobs, int n_wsets);
/* group_begin (device int32 [n_groups], or NULL): first row of every group - packed row layouts (the rows an expert-parallel rank
 * received: swn_route_pack's layout, as in swn_chain_desc.group_begin); NULL = group g starts at row g * group_stride.          */
int swn_wgrad_multi(const swn_wgrad_job* jobs, int n_jobs, int dtype, int n_groups, int n_wsets, int group_stride,
                    const int32_t* group_rows, int group_rows_clamp, const int32_t* group_begin, int tag, void* workspace,
                    size_t workspace_bytes, void* stream);
/* workspace (optional, recommended): n_groups * n_splits * (m_dim*n_dim + n_dim) * 4 bytes.  With it every workgroup
 * stores its partial tile and a second kernel reduces them into dw/db (deterministic, no atomics); without it
 * (NULL) partial tiles are added with fp32 atomics.                                                               */

/* ---- per-ray work of the step as single launches (round 3) ----------------------------------------------------------
 * The direction / appearance half of layer "2" is constant along a ray (models/nerf_moe.py:419-429: cat([h, embedding_dir(d),
 * embedding_a(idx)]) -> Linear "2"):  feat[n] = [pe_dir[n][0..in_dir), emb[idx[n]]] (fp32, [N, in_dir + app_dim]),
 * c_ray[n] = feat[n] @ w2r + b2 ([N, h2] fp32, the per-ray bias of the tail chain's last layer).  w2r [in_dir + app_dim][h2] f32.
 * image_indices: int32 or int64 (indices_are_int64).  Replaces torch cat / embedding / addmm.                             */
int swn_ray_feat_fwd(const void* pe_dir, int dtype, int dir_stride, int in_dir, const float* emb, int app_dim,
                     const void* image_indices, int indices_are_int64, const float* w2r, const float* b2, int n_rays, int h2,
                     float* feat, float* c_ray, void* stream);
/* The loss of the training step (runner.py:1099-1111, 646-658) and its gradient seeds in one launch:
 *   photo = mean((rgb - target)^2) over n_values = 3 N_rays;  gate_loss = mean(l_aux_a)  or, with l_aux_b (hierarchical: fine /
 *   coarse), (mean(a) + mean(b)) / 2;  loss = photo + l_aux_weight * gate_loss;  psnr = -10 log10(photo)   -> out4 (device, 4 floats)
 *   d_rgb = 2 (rgb - target) / n_values * s,  d_l_aux_x = l_aux_weight * share / n_x * s,  s = *loss_scale_dev (fp16 training) or 1. */
int swn_step_loss(const float* rgb, const float* target, int n_values, const float* l_aux_a, int n_a, const float* l_aux_b, int n_b,
                  float l_aux_weight, const float* loss_scale_dev, float* d_rgb, float* d_l_aux_a, float* d_l_aux_b, float* out4,
                  void* stream);
/* d_emb[n_images, app_dim] f32 += the rows d_feat[n] (n-th ray, leading dimension ld) of the rays with image_indices[n] == row, added in
 * ascending ray order with a fixed association: nn.Embedding's backward (models/nerf_moe.py:215-222) with run-to-run identical bits. */
int swn_emb_grad(const float* d_feat, int ld, const void* image_indices, int indices_are_int64, int n_rays, int app_dim,
                 int n_images, float* d_emb, void* stream);
/* Parameter gradients of the per-ray half of layer "2" (models/nerf_moe.py:419-429, the backward of cat([.., PE(dir), emb]) @ W):
 * d_w2r[n_feat, h2] += feat[n_rays, n_feat]^T dc_ray[n_rays, h2], d_b2[h2] += column sums of dc_ray (all fp32, row-major, dense).
 * Block partial sums in `workspace` added in a fixed order: the same bits on every run.  n_feat <= 256, h2 in {64, 128, 256}.       */
size_t swn_ray_feat_wgrad_workspace_bytes(int n_rays, int n_feat, int h2);
int swn_ray_feat_wgrad(const float* feat, const float* dc_ray, int n_rays, int n_feat, int h2, float* d_w2r, float* d_b2,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---- optimiser -------------------------------------------------------------------------------------------------
 * torch.optim.Adam (runner.py:486) over one flat fp32 parameter buffer; grad_scale multiplies the gradient
 * (1/world_size after a sum all-reduce).  Also refreshes the compute copies: shadow (dtype) same layout.         */
int swn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow, int dtype,
                  long n, float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream);
int swn_cast(const float* in, void* out, int dtype, long n, void* stream);
/* out[b][c][r] = in[b][r][c]  (fp32 master -> dtype compute copy, transposed) */
int swn_cast_transpose(const float* in, void* out, int dtype, int batch, int rows, int cols, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SWN_H */
