/*
 * dgs_train_ops.h -- C ABI of the per-step helper kernels next to the rasterizer on the training path
 * (SURVEY.md section 8 rows f1 and f3).  They are NOT part of the reference's rasterizer FFI; each replaces a
 * PyTorch / third-party call of the reference's train step:
 *
 *   dgs_ssim_forward/backward  <-  utils/loss_utils.py:45-76  ssim() : five 11x11 grouped conv2d + their autograd
 *   dgs_knn_points             <-  pytorch3d.ops.knn_points (utils/time_utils.py:950), K nearest control nodes
 *
 * All pointers are device pointers, fp32 contiguous unless noted; every call is asynchronous on `stream`.
 * Return value: 0 or a negative status, message through dgs_train_ops_last_error().
 */
#ifndef DGS_TRAIN_OPS_H
#define DGS_TRAIN_OPS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGS_TRAIN_OPS_ABI_VERSION 3   /* 2: the round-4 additions (dgs_adam_step_origin, dgs_select_row, dgs_loss_forward_merged,
                                         dgs_mlp_forward_select, dgs_mlp_backward_reduce) are required exports; 3: + dgs_adam_step_sum2 */

int dgs_train_ops_abi_version(void);
const char* dgs_train_ops_last_error(void);

/* SSIM with the 11-tap sigma=1.5 Gaussian window, zero padding (loss_utils.py:33-76), separable.
 * img1, img2: [C,H,W].  ssim_sum: device float, ACCUMULATED into (caller zeroes it): sum over all C*H*W of the
 * SSIM map, so mean = ssim_sum / (C*H*W).  If dm_dmu1 / dm_dsigma1_sq / dm_dsigma12 ([C,H,W] each) are non-NULL the
 * per-pixel partial derivatives of the map w.r.t. the three img1-dependent window statistics are stored for
 * dgs_ssim_backward. */
int dgs_ssim_forward(int C, int H, int W, const float* img1, const float* img2, float* ssim_sum, float* dm_dmu1,
                     float* dm_dsigma1_sq, float* dm_dsigma12, void* stream);

/* dL/dimg1 [C,H,W] (overwritten) for L = mean(SSIM map) * (*dL_dmean): dL_dmean is a DEVICE scalar. */
int dgs_ssim_backward(int C, int H, int W, const float* img1, const float* img2, const float* dm_dmu1,
                      const float* dm_dsigma1_sq, const float* dm_dsigma12, const float* dL_dmean, float* dL_dimg1,
                      void* stream);

/* Brute-force K nearest neighbours under squared L2, ascending, ties to the lower index.
 * x: [N,D], nodes: [M,D], 1 <= D <= 16, 1 <= K <= 4, K <= M.  idx: [N,K] int64.  dist2 (may be NULL): [N,K]. */
int dgs_knn_points(int N, int M, int D, int K, const float* x, const float* nodes, long long* idx, float* dist2,
                   void* stream);

/* Control-node linear blend skinning: ControlNodeWarp.forward with local frames (utils/time_utils.py:1139-1194) and
 * cal_nn_weight (:956-962) fused into one kernel per direction.  K = 3 neighbours, H = hyper dims (<= 13).
 *   x[N,3] surfel centres (no gradient), feature[N,feature_stride] (first H columns used, gradient), idx[N,3] int64,
 *   ntab[M, 3+H+2]  = per node [xyz(3, no gradient) | hyper(H) | radius | weight]      (radius = exp(_node_radius),
 *                                                                                       weight = sigmoid(_node_weight))
 *   attrs[M,13]     = per node [local rotation quaternion r,i,j,k (bias already added) | d_xyz(3) | d_rotation(4) | d_scaling(2)]
 *   mask[N]         motion mask (no gradient)
 * forward : d_xyz[N,3], d_rot[N,4], d_scale[N,2]
 * backward: g_feature[N,H] (overwritten), g_ntab[M,3+H+2], g_attrs[M,13] (overwritten; xyz columns of g_ntab are 0).
 *           scratch: at least dgs_lbs_scratch_bytes(M, H) bytes. */
size_t dgs_lbs_scratch_bytes(int M, int H);
int dgs_lbs_supported(int M, int H);   /* 1 when the backward's tables for M nodes / H hyper dims fit the 160 KB of LDS */
int dgs_lbs_forward(int N, int M, int H, const float* x, const float* feature, int feature_stride, const long long* idx,
                    const float* ntab, const float* attrs, const float* mask, float* d_xyz, float* d_rot, float* d_scale,
                    void* stream);
int dgs_lbs_backward(int N, int M, int H, const float* x, const float* feature, int feature_stride, const long long* idx,
                     const float* ntab, const float* attrs, const float* mask, const float* g_xyz, const float* g_rot,
                     const float* g_scale, float* g_feature, float* g_ntab, float* g_attrs, void* scratch, void* stream);

/* Normal-consistency + depth-distortion regularisers of the train step (train_gui.py:298-301) fused with the allmap
 * post-processing of render() they need (gaussian_renderer/__init__.py:172-207, utils/point_utils.py:9-38):
 *   rend_normal = allmap[2:5] rotated to world space, surf_depth = nan_to_num(allmap[5]) (depth_ratio = 1),
 *   surf_normal = normalize(cross(d/dy, d/dx of the back-projected depth)) * alpha.detach(), 0 on the border,
 *   loss = lambda_normal * mean(1 - <rend_normal, surf_normal>) + lambda_dist * mean(allmap[6]).
 * allmap[8,H,W]; rays_d[H*W,3], rays_o[3] as in depths_to_points; wvt = world_view_transform [4,4] (device).
 * forward ACCUMULATES the scalar into *loss (caller zeroes it).  backward: d_allmap[8,H,W] must be zeroed by the
 * caller; channels 2-4, 5 (atomically) and 6 receive the gradient scaled by the DEVICE scalar *g. */
int dgs_regloss_forward(int H, int W, const float* allmap, const float* rays_d, const float* rays_o, const float* wvt,
                        float lambda_normal, float lambda_dist, float* loss, void* stream);
int dgs_regloss_backward(int H, int W, const float* allmap, const float* rays_d, const float* rays_o, const float* wvt,
                         float lambda_normal, float lambda_dist, const float* g, float* d_allmap, void* stream);

/* One-launch Adam over a flat gradient bucket (torch.optim.Adam semantics: no weight decay, no amsgrad), replacing
 * the per-group multi_tensor_apply launches of gaussians.optimizer.step() + deform.optimizer.step()
 * (train_gui.py:427-431).  Parameters stay separate allocations: segment s owns elements
 * [offsets[s], offsets[s+1]) of grad / exp_avg / exp_avg_sq and updates params[s][0 .. len) with learning rate lrs[s].
 * `plan` is an opaque device buffer of dgs_adam_plan_bytes(total) bytes that dgs_adam_plan() fills once (block -> segment
 * map); step_count is a DEVICE float holding t (1 for the first step), so the call is stream-capture safe. nseg <= 64. */
size_t dgs_adam_plan_bytes(long long total);
int dgs_adam_plan(int nseg, const long long* offsets /*host, nseg+1*/, void* plan, void* stream);
int dgs_adam_step(int nseg, float* const* params /*host array of device pointers*/, const long long* offsets /*host*/,
                  const float* lrs /*host*/, const float* grad, float* exp_avg, float* exp_avg_sq, const float* step_count,
                  float beta1, float beta2, float eps, const void* plan, void* stream);
/* Same with an optional periodic learning-rate pattern per segment (host arrays of nseg entries, or all three NULL):
 * element i of segment s uses lrs2[s] when periods[s] > 0 and (i % periods[s]) >= splits[s], else lrs[s].  For the SH
 * coefficients kept as ONE [P,16,3] parameter: period 48, split 3 gives the DC term feature_lr and the higher bands
 * feature_lr / 20 (scene/gaussian_model.py:181-203 keeps them as two parameters and concatenates them every render). */
int dgs_adam_step_pattern(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                          const int* periods, const int* splits, const float* grad, float* exp_avg, float* exp_avg_sq,
                          const float* step_count, float beta1, float beta2, float eps, const void* plan, void* stream);
/* Same with an optional exponential learning-rate schedule per segment, evaluated ON THE DEVICE from step_count (a captured
 * step needs no host-side rate update): where sched_steps[s] > 0, step t of segment s runs at
 *   exp(log(lrs[s]) (1 - tau) + log(lrs_final[s]) tau),  tau = clip((t - 1 + sched_t0) / sched_steps[s], 0, 1)
 * i.e. get_expon_lr_func(lr_init, lr_final, max_steps) of utils/general_utils.py:49-83 (lr_delay_steps = 0, the only way
 * the reference calls it) at the iteration the reference uses: update_learning_rate runs after optimizer.step
 * (train_gui.py:427-432), so step t sees schedule(t - 1).  The pattern rate lrs2 is not scheduled (f_rest is constant in
 * the reference).  lrs_final / sched_steps: host arrays of nseg entries, or both NULL.
 * grad_scale: every gradient is read as grad * grad_scale (data parallel: the bucket holds the sum over the ranks and
 * grad_scale = 1 / world replaces a separate averaging pass over the bucket; 1 otherwise). */
int dgs_adam_step_sched(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                        const int* periods, const int* splits, const float* lrs_final, const float* sched_steps, float sched_t0,
                        float grad_scale, const float* grad, float* exp_avg, float* exp_avg_sq, const float* step_count, float beta1, float beta2,
                        float eps, const void* plan, void* stream);

/* Guarded step.  In capacity mode (dgs_surfel_rasterizer.h) a view whose tile lists do not fit renders as background and
 * raises an int32 device flag; nothing on the host knows yet.  The kernels below read such a flag ON THE DEVICE (`skip`:
 * the rasterizer's flag itself, or -- data parallel -- a copy that went through a MAX all-reduce so that every rank sees
 * "some rank overflowed"):
 *   dgs_step_guard(skip, step_count, status, host_ring, ring_len, loss): one thread; advances the Adam step count unless
 *     skip[0] != 0; status[3] = {skip flag of this step, skipped steps so far, guarded steps so far}; if host_ring (pinned,
 *     device-accessible host memory, 4 floats per entry) is given, entry (steps % ring_len) <- (steps, flag, skipped, loss[0]
 *     or 0) so the host can poll the outcome -- and the loss -- of step k a few steps later without synchronising,
 *   dgs_adam_step_guarded / dgs_densify_accumulate_guarded: return without changing anything when skip[0] != 0.
 * A frame that overflowed therefore trains nothing -- not the parameters, not the moments, not the statistics. */
int dgs_step_guard(const int* skip, float* step_count, float* status, float* host_ring, int ring_len, const float* loss, void* stream);

/* View selection on the device (first node of a captured step): row_out[0 .. row_floats) <- table[v], v = override[0] if >= 0 (then
 * reset to -1) else (counter[0] * stride + offset) mod nrows; counter[0] += 1.  A replayed step that walks its views in the default
 * order (rank r of `stride` ranks renders view (i * stride + r) mod nrows in step i, dgs_amd.train.Trainer.view_for) then needs no
 * host-issued copy per step; any other order writes `override` before the replay. */
int dgs_select_row(const float* table, int nrows, int row_floats, int* counter, int* override_, int stride, int offset, float* row_out, void* stream);
int dgs_adam_step_guarded(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                          const int* periods, const int* splits, const float* lrs_final, const float* sched_steps, float sched_t0,
                          float grad_scale, const float* grad, float* exp_avg, float* exp_avg_sq, const float* step_count, float beta1,
                          float beta2, float eps, const void* plan, const int* skip, void* stream);
/* optimizer.step() and zero_grad() in one pass (train_gui.py:427-431 calls them back to back): with zero_grad != 0 every gradient
 * element the launch reads is cleared behind the read -- on a skipped step too, or the next step would add to a stale buffer --
 * so the trainer needs no 57 MB fill in front of the next backward.  zero_grad = 0: dgs_adam_step_guarded. */
int dgs_adam_step_zero(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                       const int* periods, const int* splits, const float* lrs_final, const float* sched_steps, float sched_t0,
                       float grad_scale, float* grad, int zero_grad, float* exp_avg, float* exp_avg_sq, const float* step_count,
                       float beta1, float beta2, float eps, const void* plan, const int* skip, void* stream);
/* The same with a step ORIGIN per segment (host array, may be NULL = all 0): the bias corrections of segment s use
 * max(step_count - step_origins[s], 1).  torch.optim.Adam keeps a step count per parameter and skips parameters whose .grad is
 * None, so a parameter that joins the optimisation late -- the reference's deformation network, control nodes and `feature`
 * after the warm-up (train_gui.py:281-285, 427-432) -- starts at step 1; pass the run's step count at that moment as its origin.
 * A negative origin is a parameter that arrives with steps already taken (the deformation model's optimiser runs on from the node
 * pre-training stage); NaN is rejected.  Learning-rate schedules keep using the run's counter. */
int dgs_adam_step_origin(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                         const int* periods, const int* splits, const float* lrs_final, const float* sched_steps, float sched_t0,
                         const float* step_origins, float grad_scale, float* grad, int zero_grad, float* exp_avg, float* exp_avg_sq,
                         const float* step_count, float beta1, float beta2, float eps, const void* plan, const int* skip, void* stream);
/* The same reading the gradient as grad + grad2 (grad2: a second buffer of the same layout, or NULL): two views of one step that were
 * rendered CONCURRENTLY keep a gradient buffer each (they cannot add into one without racing) and the update takes their sum on the
 * fly -- one extra read instead of an adding pass over both.  zero_grad must be 0 when grad2 is given.  (Round 6; not in the
 * reference, which renders one view per optimizer step: train_gui.py:258,426-432.) */
int dgs_adam_step_sum2(int nseg, float* const* params, const long long* offsets, const float* lrs, const float* lrs2,
                       const int* periods, const int* splits, const float* lrs_final, const float* sched_steps, float sched_t0,
                       const float* step_origins, float grad_scale, float* grad, const float* grad2, int zero_grad, float* exp_avg,
                       float* exp_avg_sq, const float* step_count, float beta1, float beta2, float eps, const void* plan, const int* skip,
                       void* stream);

/* Control-node deformation MLP (DeformNetwork, utils/time_utils.py:311-453, is_blender + local_frame configuration:
 * posenc(xyz,10) | timenet(posenc(t,6)): 13->256->30, 8 x 256 ReLU layers, skip concat after layer 4, heads
 * local_rotation 4 / warp 3 / rotation 4 / scaling 2) forward + backward as four kernels on fp32 MFMA.
 *   params / grads: HOST arrays of 28 device pointers, (weight, bias) of: timenet.0, timenet.2, linear.0 .. linear.7,
 *                   local_rotation, gaussian_warp, gaussian_rotation, gaussian_scaling  (torch Linear layout [out][in])
 *   x[M, x_stride] node positions (first 3 columns), t[M * t_stride] time per node (t_stride 0 = one shared value)
 *   attrs[M,13] = [local_rotation + rot_bias(4, host) | d_xyz 3 | d_rotation 4 | d_scaling 2]  (dgs_lbs_forward's table)
 *   packed: dgs_mlp_packed_floats() floats, written by forward, read by backward (weights re-laid as [k/4][column] float4
 *           operands of the two chains: csrc/node_mlp.h)
 *   saved : dgs_mlp_saved_floats(M) floats of activations, written by forward, read by backward
 *   scratch: dgs_mlp_scratch_floats(M) floats
 * backward writes (accumulate = 0) or adds to (accumulate = 1) every gradient tensor.  M must be a multiple of 64. */
size_t dgs_mlp_packed_floats(void);
size_t dgs_mlp_saved_floats(int M);
size_t dgs_mlp_scratch_floats(int M);
int dgs_mlp_forward(int M, const float* x, int x_stride, const float* t, int t_stride, const float* const* params,
                    const float* rot_bias, float* packed, float* saved, float* attrs, void* stream);
/* dgs_mlp_forward whose weight-packing launch also performs dgs_select_row(table .. row_out) (table may be NULL: then exactly
 * dgs_mlp_forward): for a captured train step whose node MLP is the first consumer of the selected view (its time `t` points
 * into row_out) -- the selection then costs no node of its own in front of the step. */
int dgs_mlp_forward_select(int M, const float* x, int x_stride, const float* t, int t_stride, const float* const* params,
                           const float* rot_bias, float* packed, float* saved, float* attrs, const float* table, int nrows, int row_floats,
                           int* counter, int* override_, int stride, int offset, float* row_out, void* stream);
int dgs_mlp_backward(int M, const float* g_attrs, const float* packed, const float* saved, float* scratch, float* const* grads,
                     int accumulate, void* stream);
/* dgs_mlp_backward with dgs_deform_reduce folded into its first kernel (lbs_table may be NULL: then exactly dgs_mlp_backward):
 * the skinning backward left its node table [M][13 + H + 2] unreduced (dgs_deform_backward, accumulate bit 3); every workgroup of
 * the MLP's backward chain reduces the rows of its own nodes -- g_attrs[M][13] is WRITTEN here (and read by the weight-gradient
 * kernel), g_nodes / g_radius_raw / g_weight_raw as dgs_deform_reduce writes them; reduce_flags as its `accumulate`. */
int dgs_mlp_backward_reduce(int M, float* g_attrs, const float* packed, const float* saved, float* scratch, float* const* grads,
                            int accumulate, int H, const float* node_radius_raw, const float* node_weight_raw, float* g_nodes,
                            float* g_radius_raw, float* g_weight_raw, int reduce_flags, void* lbs_table, void* stream);

/* dgs_knn_points with the query coordinates split over two arrays: [0,D1) from x1[N,D1], [D1,D1+D2) from
 * x2[N, x2_stride] (avoids materialising cat([xyz, feature[:, :hyper]]) every step). */
int dgs_knn_points2(int N, int M, int D1, int D2, int K, const float* x1, const float* x2, int x2_stride, const float* nodes,
                    long long* idx, float* dist2, void* stream);

/* Control-node skinning of dgs_lbs_* fused with (a) the node-table activations node_radius = exp(_node_radius),
 * node_weight = sigmoid(_node_weight) (utils/time_utils.py:812-818) and (b) the surfel activations render() applies
 * around the deformation (gaussian_renderer/__init__.py:60-75, scene/gaussian_model.py:60-78):
 *   means3D = xyz + d_xyz, scales = exp(_scaling) + d_scaling, rotations = normalize(_rotation + d_rotation),
 *   opacity = sigmoid(_opacity).
 * nodes[M,3+H] = [xyz (detached) | hyper]; mask may be NULL (= 1).  backward: g_attrs[M,13] is always overwritten; all
 * other gradient arrays are overwritten (accumulate = 0; the xyz columns of g_nodes are zeroed) or added to
 * (accumulate = 1, e.g. the .grad views of a flat gradient bucket).  g_feature has the row stride of feature.
 * accumulate bit 1 (value 2, 3) selects the COHERENT backward for point sets stored in the order of their nearest node
 * (dgs_amd.train.Trainer.sort_surfels): the 64 points of a wave then share one or two nodes per neighbour slot, their
 * contributions are summed across the wave and one 23-lane global atomic per (wave, node) goes into a single [M][13+H+2]
 * table -- instead of 256 per-workgroup LDS tables of 94 KB.  Same results for any order (float summation order aside);
 * an unsorted set makes it slow, not wrong.  accumulate bit 2 (value 4, coherent variant only): `scratch` is a persistent
 * table that is all zero on entry and is left all zero (no memset launch).  accumulate bit 4 (value 16, coherent variant only):
 * the table holds 64-bit FIXED-POINT sums (units of 2^-44) added with integer atomics -- order-free, so the node gradients are
 * bit-identical from run to run (the float atomics of the default are not); pass the same bit to dgs_deform_reduce /
 * dgs_mlp_backward_reduce, which convert.  scratch: dgs_lbs_scratch_bytes(M, H) bytes (enough for either table). */
int dgs_deform_forward(int N, int M, int H, const float* xyz, const float* feature, int feature_stride, const long long* idx,
                       const float* nodes, const float* node_radius_raw, const float* node_weight_raw, const float* attrs,
                       const float* mask, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                       float* means3D, float* scales, float* rotations, float* opacity, void* stream);
int dgs_deform_backward(int N, int M, int H, const float* xyz, const float* feature, int feature_stride, const long long* idx,
                        const float* nodes, const float* node_radius_raw, const float* node_weight_raw, const float* attrs,
                        const float* mask, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                        const float* g_means3D, const float* g_scales, const float* g_rotations, const float* g_opacity,
                        float* g_xyz, float* g_scaling_raw, float* g_rotation_raw, float* g_opacity_raw, float* g_feature,
                        float* g_nodes, float* g_radius_raw, float* g_weight_raw, float* g_attrs, int accumulate, void* scratch,
                        void* stream);
/* accumulate bit 3 (value 8, coherent variant only) makes dgs_deform_backward leave its [M][13+H+2] table unreduced: g_nodes,
 * g_radius_raw, g_weight_raw and g_attrs are then produced by this call (same accumulate bits 0, 2 and 4), which the caller may
 * launch on another stream -- the train step runs it in front of the node-MLP backward on that backward's side stream, so that
 * the surfels' Adam update starts right behind the skinning backward. */
int dgs_deform_reduce(int M, int H, const float* node_radius_raw, const float* node_weight_raw, float* g_nodes, float* g_radius_raw,
                      float* g_weight_raw, float* g_attrs, int accumulate, void* scratch, void* stream);

/* Photometric loss of the train step, (1 - lambda) * mean|img - gt| + lambda * (1 - SSIM(img, gt)) (train_gui.py:292-296),
 * on the SSIM kernels, with the regularisers of dgs_regloss_*.  Reductions go through per-workgroup partial sums (no
 * atomics: thousands of atomics on one address serialise in a single L2 channel and dominated these kernels):
 *   dgs_photo_forward            partials[0 .. B)  = SSIM-map sums, partials[B .. 2B) = |img - gt| sums, B = dgs_photo_blocks()
 *   dgs_regloss_forward_partials partials[0 .. R)  = regulariser sums (already scaled and divided by H*W), R = dgs_regloss_blocks()
 *   dgs_loss_combine             out[0] = (1 - lambda) * sum(l1) / n + lambda * (1 - sum(ssim) / n) + sum(reg),  n = C*H*W
 *   dgs_photo_backward           dL/dimg for the DEVICE scalar *g_loss.
 * gt_slot / rays_slot (may be NULL): DEVICE locations holding the pointer to use instead of gt / rays_d -- a captured HIP
 * graph then switches target image and ray table per replay by rewriting 8 bytes instead of copying 7.7 MB each.
 * dgs_regloss_backward_slot with write_all = 1 stores every element of d_allmap except plane 5 (the depth gradient, added
 * atomically): the caller zero-fills that plane only instead of all eight. */
size_t dgs_photo_blocks(int C, int H, int W);
size_t dgs_regloss_blocks(int H, int W);
int dgs_photo_forward(int C, int H, int W, const float* img, const float* gt, float* partials, float* dm_dmu1, float* dm_dsigma1_sq,
                      float* dm_dsigma12, const float* const* gt_slot, void* stream);
int dgs_regloss_forward_partials(int H, int W, const float* allmap, const float* rays_d, const float* rays_o, const float* wvt,
                                 float lambda_normal, float lambda_dist, float* partials, const float* const* rays_slot, void* stream);
int dgs_regloss_backward_slot(int H, int W, const float* allmap, const float* rays_d, const float* rays_o, const float* wvt,
                              float lambda_normal, float lambda_dist, const float* g, float* d_allmap, const float* const* rays_slot,
                              int write_all, void* stream);
/* dgs_regloss_forward_partials that also clears zero_plane[H,W] (may be NULL): the plane of the backward's gradient image that
 * collects atomics, cleared here so that the backward needs no fill launch */
int dgs_regloss_forward_partials_z(int H, int W, const float* allmap, const float* rays_d, const float* rays_o, const float* wvt,
                                   float lambda_normal, float lambda_dist, float* partials, const float* const* rays_slot,
                                   float* zero_plane, void* stream);
/* Value AND gradient of the regularisers in one kernel, for an upstream gradient dL/dloss == 1 (what the train step uses):
 * partials[0 .. dgs_regloss_fused_blocks()) as dgs_regloss_forward_partials, and EVERY element of d_allmap[8,H,W] is stored -- the
 * depth gradient (plane 5) is gathered from the four neighbouring normals instead of added atomically, so nothing needs clearing.
 * Same values as dgs_regloss_forward_partials + dgs_regloss_backward_slot(g = 1) up to the summation order of plane 5. */
size_t dgs_regloss_fused_blocks(int H, int W);
int dgs_regloss_fused(int H, int W, const float* allmap, const float* rays_d, const float* rays_o, const float* wvt, float lambda_normal,
                      float lambda_dist, float* partials, float* d_allmap, const float* const* rays_slot, void* stream);
/* dgs_photo_forward and dgs_regloss_fused in ONE launch: the two kinds of workgroup alternate in one grid (they read different
 * rasterizer outputs and write different buffers), so neither leaves the chip half idle in its last round of workgroups and one
 * launch gap goes away.  Same outputs, bit for bit, as the two calls. */
int dgs_loss_forward_merged(int C, int H, int W, const float* img, const float* gt, float* photo_partials, float* dm_dmu1,
                            float* dm_dsigma1_sq, float* dm_dsigma12, const float* const* gt_slot, const float* allmap, const float* rays_d,
                            const float* rays_o, const float* wvt, float lambda_normal, float lambda_dist, float* reg_partials,
                            float* d_allmap, const float* const* rays_slot, void* stream);
int dgs_loss_combine(const float* photo_partials, long long nphoto, const float* reg_partials, long long nreg, long long n,
                     float lambda_dssim, float* out, void* stream);
int dgs_photo_backward(int C, int H, int W, const float* img, const float* gt, const float* dm_dmu1, const float* dm_dsigma1_sq,
                       const float* dm_dsigma12, float lambda_dssim, const float* g_loss, float* dL_dimg, const float* const* gt_slot,
                       void* stream);
/* dgs_photo_backward whose last workgroup also performs dgs_loss_combine (loss_out may be NULL: then exactly dgs_photo_backward).
 * For callers that launch it AFTER the forward kernels that fill the partials (the unit-gradient train step). */
int dgs_photo_backward_combine(int C, int H, int W, const float* img, const float* gt, const float* dm_dmu1, const float* dm_dsigma1_sq,
                               const float* dm_dsigma12, float lambda_dssim, const float* g_loss, float* dL_dimg, const float* const* gt_slot,
                               const float* photo_partials, long long nphoto, const float* reg_partials, long long nreg, float* loss_out,
                               void* stream);
/* ... whose same thread also performs dgs_step_guard (guard_step_count != NULL) with the loss it has just written: the train
 * step's guard kernel (one thread, ~5 us of a replayed step) needs nothing that does not exist at that point. */
int dgs_photo_backward_combine_guard(int C, int H, int W, const float* img, const float* gt, const float* dm_dmu1, const float* dm_dsigma1_sq,
                                     const float* dm_dsigma12, float lambda_dssim, const float* g_loss, float* dL_dimg,
                                     const float* const* gt_slot, const float* photo_partials, long long nphoto, const float* reg_partials,
                                     long long nreg, float* loss_out, const int* guard_skip, float* guard_step_count, float* guard_status,
                                     float* guard_ring, int guard_ring_len, void* stream);

/* Densification statistics (train_gui.py:411, scene/gaussian_model.py:484-486).  dgs_densify_view, per rendered view:
 * visible = radii > 0, grad_norm = |dL/dmeans2D[:, :2]| where visible (else 0), radii_vis = radii where visible.
 * dgs_densify_accumulate: xyz_gradient_accum += grad_norm, denom += visible, max_radii2D = max(max_radii2D, radii_vis). */
int dgs_densify_view(int P, const int* radii, const float* g_means2D, float* grad_norm, float* visible, int* radii_vis, void* stream);
int dgs_densify_accumulate(int P, const float* grad_norm, const float* visible, const int* radii_vis, float* accum, float* denom,
                           int* max_radii, void* stream);
/* ... with the step guard (see dgs_step_guard) */
int dgs_densify_accumulate_guarded(int P, const float* grad_norm, const float* visible, const int* radii_vis, float* accum, float* denom,
                                   int* max_radii, const int* skip, void* stream);

/* Exact K nearest neighbours seeded with a previous answer: idx[N,K] holds any earlier result on entry (typically last
 * step's; stale, random or invalid entries only cost time) and the exact answer of dgs_knn_points2 on exit.  The scan over
 * the M <= 2048 nodes uses only coordinates 0..2 (D1 >= 3: a lower bound of the full squared distance) against the bound
 * the seed gives; full distances are evaluated for the few survivors.  Whole 32-node blocks are skipped when the bounding box
 * of a wave's search spheres misses the block's: this pays when consecutive points are neighbours in space and consecutive
 * nodes are too (Trainer.sort_surfels / sort_nodes); any order gives the same, exact result.  D2 may be 0 (x2 unused). */
int dgs_knn_refine(int N, int M, int D1, int D2, int K, const float* x1, const float* x2, int x2_stride, const float* nodes,
                   long long* idx, void* stream);
/* dgs_knn_refine with the filter chosen by the caller.  mode 0: the 3-D culling above.  mode 1 (D1 + D2 <= 11, M <= 1024): every
 * (node, point) score |n|^2 - 2 x.n on the matrix cores (operands split into two bf16 each, three v_mfma_f32_32x32x16_bf16 per
 * 32 x 32 tile; a conservative filter, candidates are then evaluated exactly in f32) against the seed's bound in the FULL D1 + D2
 * dimensions -- dense, independent of the data; for scenes whose extra coordinates (x2: the hyper features) have drifted so far
 * from the nodes' that the K-th neighbour distance is no longer a spatial radius and the 3-D culling stops culling (trained scenes:
 * 172 us -> 19 us at 125 k points x 512 nodes; its cost grows with M, so on an untrained, spatially sorted scene with 1024 nodes
 * mode 0 is the faster one, 43 against 71 us at 200 k).  Same exact result either way. */
int dgs_knn_refine_mode(int N, int M, int D1, int D2, int K, const float* x1, const float* x2, int x2_stride, const float* nodes,
                        long long* idx, int mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DGS_TRAIN_OPS_H */
