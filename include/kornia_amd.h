/*
 * kornia_amd - C ABI of the MI355X (gfx950) native warp + filter path.
 *
 * One shared library, `kornia_amd/lib/libkornia_amd.so`, built by `python -m kornia_amd.build`
 * (hipcc --offload-arch=gfx950).  The reference (kornia, pure Python on PyTorch) has no FFI for
 * this path: the interface each entry point replaces is the Python call site cited next to it
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding a Kornia
 * maintainer would add.
 *
 * Conventions
 *   - every pointer is a raw DEVICE pointer into memory owned by the caller (PyTorch's caching
 *     allocator in the shipped binding) unless the argument is named *_host;
 *   - tensors are dense, contiguous, NCHW; sizes are element counts;
 *   - `dtype`: 0 = float32, 1 = float64, 2 = bfloat16, 3 = float16.  Matrices, filter taps and
 *     gradient accumulators use the COMPUTE dtype: float32 for dtype 0/2/3, float64 for dtype 1;
 *   - `stream` is a hipStream_t; launches are asynchronous on it; nothing synchronises the device,
 *     allocates, frees, or retains pointers after returning;
 *   - return value: 0 = ok, < 0 = invalid argument, > 0 = hipError_t; `km_last_error()` returns a
 *     thread-local message.  No C++ exception crosses the boundary.
 *
 * Versioning (`km_abi_version()`, currently 3)
 *   The version counts SYMBOL SETS: a library of version N exports every entry point of the sets 1 .. N with unchanged
 *   signatures, so a binding checks `km_abi_version() >= V` with V the set of the newest entry point it calls (no per-symbol
 *   probing).  Adding an entry point, or changing what an existing one computes for some input, opens a new set.
 *     set 1  the round-1 .. round-3 entry points;
 *     set 2  km_warp2d_bwd_ws on any channel count / border / reflection, zeroing `gmat` itself (round 4);
 *     set 3  + km_color_params_ws_fwd, km_gaussian_taps_dtype_fwd, km_warp_masked_loss_finish, km_scale_f64 (added in round 5 while the
 *            version still read 2: a version-2 library may lack them), km_stream_copy; and the sampler follows ATen's CPU rule for
 *            NaN / inf sampling coordinates (taps outside the image are zeros that are still multiplied: NaN out, NaN matrix gradient)
 *            where sets 1-2 returned the padding value.
 */
#ifndef KORNIA_AMD_H
#define KORNIA_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

enum { KM_DTYPE_F32 = 0, KM_DTYPE_F64 = 1, KM_DTYPE_BF16 = 2, KM_DTYPE_F16 = 3 };
/* warp coordinate generators */
enum { KM_WARP_PERSPECTIVE = 0, KM_WARP_AFFINE = 1, KM_WARP_HOMOGRAPHY = 2 };
/* interpolation / padding of the sampler (F.grid_sample modes; 3 = kornia's 'fill') */
enum { KM_NEAREST = 0, KM_BILINEAR = 1, KM_BICUBIC = 2 };
enum { KM_ZEROS = 0, KM_BORDER = 1, KM_REFLECTION = 2, KM_FILL = 3 };
/* filter border modes (F.pad modes) */
enum { KM_CONSTANT = 0, KM_REFLECT = 1, KM_REPLICATE = 2, KM_CIRCULAR = 3 };

int km_abi_version(void);
const char* km_last_error(void);
/* gcnArchName of the current device into name[n]; returns the CU count or < 0 */
int km_device_info(char* name, int n);
/* Batch traversal of the streaming kernels (warp forward / backward, separable blur): mode 0 (default) alternates the direction
 * from one launch to the next, so that a consumer starts on what its producer left in the 256 MB Infinity Cache; mode 1 keeps
 * every launch forward (what a per-kernel timing loop over one input needs: otherwise launch k+1 re-reads the tail launch k just
 * read).  Results do not depend on it.  Returns the previous mode.  (No reference counterpart: the reference has no launch policy.) */
int km_set_traversal(int mode);
/* Launch policy.  The library reads its A/B switches (profiles/README.md: KM_WARP_FWD_ALGO, KM_BLUR_ROWS, ...) from the environment ONCE,
 * when it is first used; no launcher calls getenv.  km_config_set changes one entry explicitly ("traversal_fixed", "warp_fwd_algo",
 * "warp_gm_algo", "warp_bwd_generic", "warp_bwd_fused", "sep_lds", "sg_generic", "pyrdown_separable", "blur_rows", "warp_bwd_no_scan") and returns its
 * previous value (-1: unknown key).  For tests and A/B timing - call it between launches, not concurrently with them.  The alternating
 * traversal itself keeps its parity per (device, stream): what one stream launches never changes the order another stream's kernels
 * walk the batch.  (No reference counterpart.) */
int km_config_set(const char* key, int value);
int km_config_get(const char* key);
/* Diagnostic: a plain streaming copy of `bytes` bytes (a multiple of 16, both pointers 16-byte aligned), 16 bytes per lane, nontemporal != 0:
 * non-temporal loads and stores.  bench.py times it beside the hot path, so that every line carries the copy bandwidth of THE BOX IT RAN ON
 * (leases differ by several per cent) next to the 8 TB/s of the data sheet.  (No reference counterpart.) */
int km_stream_copy(const void* src, void* dst, long long bytes, int nontemporal, void* stream);

/* ---- batched 3x3 homography chain -----------------------------------------------------------
 * Replaces normalize_homography (kornia/geometry/conversions.py:1691-1726),
 * normal_transform_pixel (:1729-1763), convert_affinematrix_to_homography (:342-378) and
 * _inverse_3x3_closed_form (kornia/core/utils.py:137-166) as called from
 * kornia/geometry/transform/imgwarp.py:146-153 and :249-254.
 *   M      (B, rows, 3)  pixel src->dst matrix, rows = 3 (homography) or 2 (affine)
 *   A_out  (B, 9)        normalised src->dst  = N_dst @ (M @ inv(N_src))          (nullable)
 *   m_out  (B, 9)        normalised dst->src  = inv(A)                            (nullable)
 * dtype: 0 | 1. */
int km_homography_chain_fwd(const void* M, int rows, void* A_out, void* m_out, int B, int src_h, int src_w, int dst_h,
                            int dst_w, int dtype, void* stream);
/* gm (B,9) float64 gradient wrt m_out  ->  gM (B, rows, 3) in dtype. */
int km_homography_chain_bwd(const void* M, int rows, const double* gm, void* gM, int B, int src_h, int src_w, int dst_h,
                            int dst_w, int dtype, void* stream);

/* ---- warps ---------------------------------------------------------------------------------
 * Replaces the eager grid construction + F.grid_sample of warp_perspective
 * (kornia/geometry/transform/imgwarp.py:157-174), warp_affine (:271-290), homography_warp with a
 * normalised homography (:1539-1546 -> warp_grid :323-353 -> transform_points,
 * kornia/geometry/linalg.py:219-239) and _fill_and_warp (:293-320).  The (B,h,w,2) grid is never
 * materialised.
 *   src   (B,C,H,W) dtype      mat (B_M,9) compute dtype, B_M in {1,B}: the matrix applied to the
 *   dst   (B,C,h,w) dtype          base grid (m_out of the chain, or the user's normalised H)
 *   coord_mode  KM_WARP_*      norm_coords: homography mode only (normalized_coordinates flag)
 *   fill  (C) compute dtype, only read when pad == KM_FILL */
int km_warp2d_fwd(const void* src, const void* mat, void* dst, int B, int C, int H, int W, int h, int w, int B_M,
                  int coord_mode, int norm_coords, int interp, int pad, int align_corners, const void* fill, int dtype,
                  void* stream);
/* Replaces autograd's aten::grid_sampler_2d_backward + the reverse of the grid chain.
 *   gout  (B,C,h,w) dtype
 *   gsrc  (B,C,H,W) COMPUTE dtype, nullable; zeroed by the caller iff km_warp2d_bwd_needs_zero_init()
 *   gmat  (B_M,9) float64 accumulators, zeroed by the caller, nullable */
int km_warp2d_bwd(const void* gout, const void* src, const void* mat, void* gsrc, double* gmat, int B, int C, int H,
                  int W, int h, int w, int B_M, int coord_mode, int norm_coords, int interp, int pad, int align_corners,
                  const void* fill, int dtype, void* stream);
/* The same with a caller-owned workspace.  When both gsrc and gmat are wanted and workspace_bytes >= km_warp2d_bwd_workspace_bytes(...)
 * (16-byte aligned device memory, contents irrelevant, not used after the launches it is passed to), both gradients come from ONE read
 * of grad_out (a persistent tile-owner kernel that keeps its source tile in LDS: 3e bytes per element instead of 4e).  With a null /
 * short workspace it is km_warp2d_bwd.  Same results to the rounding of the fixed-point scale (both within the tolerances of tests/).
 * On that path (both gradients wanted, km_warp2d_bwd_workspace_bytes(...) > 0 and a workspace of at least that size) gmat need NOT be
 * zeroed by the caller: the first launch of the sequence does it (ABI version 2; zeroing it anyway is harmless).
 * Border / reflection padding (fp32 storage): with such a workspace the same tile-owner kernel also serves gsrc alone (gmat == NULL) and
 * writes every element of gsrc - no zeroing by the caller then, whatever km_warp2d_bwd_needs_zero_init() says about the path without one. */
int km_warp2d_bwd_ws(const void* gout, const void* src, const void* mat, void* gsrc, double* gmat, int B, int C, int H,
                     int W, int h, int w, int B_M, int coord_mode, int norm_coords, int interp, int pad, int align_corners,
                     const void* fill, int dtype, void* workspace, long long workspace_bytes, void* stream);
/* bytes of workspace the one-read backward uses for these sizes and modes; 0: it does not apply (pass no workspace) */
long long km_warp2d_bwd_workspace_bytes(int B, int C, int H, int W, int h, int w, int interp, int pad, int dtype);

/* ---- warp + separable blur, fused (forward) ------------------------------------------------------
 * y = filter2d_separable(warp(src, mat), kx, ky, border): kornia/geometry/transform/imgwarp.py:143-174 / :246-290 / :1539-1546 followed by
 * kornia/filters/filter.py:155-207 (GaussianBlur2d: kornia/filters/gaussian.py:32-120) in ONE launch that never writes the warped image
 * (5e instead of 9e bytes per element over forward + backward).  Bit-identical to the two calls.  Bilinear + zeros, C in {1, 3},
 * square odd K in {3, 5, 7}, 'same' output; km_warp2d_blur_supported() tells (1 / 0), anything else: the two calls.
 *   src (B,C,H,W) dtype   mat (B_M,9) float32 as for km_warp2d_fwd   kx, ky (Bk,K) float32, Bk in {1,B}   dst (B,C,h,w) dtype
 *   border: KM_BORDER_* of the blur (F.pad's modes) */
int km_warp2d_blur_supported(int C, int H, int W, int h, int w, int interp, int pad, int K, int border, int dtype);
int km_warp2d_blur_fwd(const void* src, const void* mat, const void* kx, const void* ky, void* dst, int B, int C, int H, int W, int h,
                       int w, int B_M, int Bk, int coord_mode, int norm_coords, int align_corners, int K, int border, int dtype, void* stream);

/* 1 if km_warp2d_bwd with these modes accumulates with atomics and needs gsrc zeroed by the caller,
 * 0 if it overwrites gsrc completely (tile-owner path: bilinear, zeros/fill padding, dtype != f64). */
int km_warp2d_bwd_needs_zero_init(int interp, int pad, int dtype);

/* ---- augmentation parameters -> matrix ---------------------------------------------------------
 * Replaces get_affine_matrix2d (kornia/geometry/transform/imgwarp.py:746-787; get_rotation_matrix2d :529-622,
 * get_shear_matrix2d :815-869, angle_to_rotation_matrix kornia/geometry/conversions.py:1652-1688) as called by
 * RandomAffine.compute_transformation (kornia/augmentation/_2d/geometric/affine.py:125-141): one launch, one
 * thread per matrix, instead of ~45 elementwise / bmm launches.
 *   translations, center, scale (B,2); angle (B) degrees; sx, sy (B) shear angles in radians or NULL;
 *   out (B,3,3) pixel matrix; dtype KM_F32 or KM_F64 (all operands in that dtype). */
int km_affine_matrix2d_fwd(const void* translations, const void* center, const void* scale, const void* angle, const void* sx,
                           const void* sy, void* out, int B, int dtype, void* stream);

/* RandomAffine.compute_transformation + the normalise / invert chain of warp_affine in one launch (float32):
 * kornia/augmentation/_2d/geometric/affine.py:125-141 (-> imgwarp.py:746-787), then imgwarp.py:271-284.
 * translations / center / scale (B,2), angle (B) degrees, shear_x / shear_y (B) DEGREES or NULL ->
 * M_out (B,9) pixel src->dst matrix (nullable), m_out (B,9) the normalised dst->src matrix km_warp2d_fwd reads with
 * coord_mode = affine for a (Hs,Ws) source and (hd,wd) destination.  batch_prob (B) fp32 or NULL -> apply (B) uint8 =
 * batch_prob > 0.5 (kornia/augmentation/base.py:380), the per-sample switch km_warp2d_fwd_masked takes. */
int km_affine_params_chain_fwd(const void* translations, const void* center, const void* scale, const void* angle, const void* shear_x,
                               const void* shear_y, const void* batch_prob, void* M_out, void* m_out, void* apply, int B, int Hs, int Ws, int hd,
                               int wd, void* stream);

/* ColorJitter's sampled factors -> the inputs of km_color_jitter_fwd(_masked) in one launch
 * (kornia/augmentation/_2d/intensity/color_jitter.py:137-148, base.py:380): brightness / contrast / saturation / hue (B) fp32
 * (hue in turns), batch_prob (B) fp32 or NULL -> params (B,4) fp32 (hue in radians), enable (4) uint8 = the module's
 * `(factor != neutral).any()` guards, apply (B) uint8 = batch_prob > 0.5 (iff batch_prob is given). */
int km_color_params_fwd(const void* brightness, const void* contrast, const void* saturation, const void* hue, const void* batch_prob, void* params,
                        void* enable, void* apply, int B, void* stream);
/* The same, and gray_sum (B) fp64 - the workspace km_color_jitter_fwd's contrast stage accumulates into - is zeroed by the launch (the
 * caller's fill launch folded in). */
int km_color_params_ws_fwd(const void* brightness, const void* contrast, const void* saturation, const void* hue, const void* batch_prob,
                           void* params, void* enable, void* apply, void* gray_sum, int B, void* stream);

/* Replaces get_perspective_transform (kornia/geometry/transform/imgwarp.py:397-525, Heckbert closed form) as called by
 * RandomPerspective.compute_transformation (kornia/augmentation/_2d/geometric/perspective.py): one launch instead of ~40.
 *   points_src, points_dst (B,4,2) -> out (B,3,3) with out[2][2] == 1; dtype KM_F32 or KM_F64. */
int km_perspective_transform_fwd(const void* points_src, const void* points_dst, void* out, int B, int dtype, void* stream);

/* ---- explicit sampling grid ------------------------------------------------------------------
 * Replaces F.grid_sample(input, grid, mode, padding_mode, align_corners) as called by remap
 * (kornia/geometry/transform/imgwarp.py:702) and by HomographyWarper's cached-grid forward
 * (kornia/geometry/transform/homography_warper.py:182).
 *   grid (B_G,h,w,2): normalised (x, y) pairs in the IMAGE dtype, B_G in {1, B} (1 = shared by the batch);
 *   interp 0 nearest / 1 bilinear / 2 bicubic; pad 0 zeros / 1 border / 2 reflection.
 *   bwd: gsrc (B,C,H,W) compute dtype, zeroed by the caller, nullable; ggrid (B,h,w,2) compute dtype,
 *   overwritten, nullable (for B_G == 1 the caller sums it over the batch). */
int km_grid_sample2d_fwd(const void* src, const void* grid, void* dst, int B, int C, int H, int W, int h, int w, int B_G,
                         int interp, int pad, int align, int dtype, void* stream);
int km_grid_sample2d_bwd(const void* gout, const void* src, const void* grid, void* gsrc, void* ggrid, int B, int C, int H,
                         int W, int h, int w, int B_G, int interp, int pad, int align, int dtype, void* stream);

/* ---- fused ColorJitter ---------------------------------------------------------------------------
 * Replaces the per-stage elementwise chains of ColorJitter.apply_transform
 * (kornia/augmentation/_2d/intensity/color_jitter.py:126-159): adjust_brightness_accumulative
 * (kornia/enhance/adjust.py:542-593), adjust_contrast_with_mean_subtraction (:414-469),
 * adjust_saturation_with_gray_subtraction (:80-134), adjust_hue (:212-254; rgb_to_hsv / hsv_to_rgb
 * kornia/color/hsv.py:27-131).
 *   x, y (B,3,H,W) RGB, dtype f32 / bf16 / f16; params (B,4) fp32 DEVICE: brightness, contrast, saturation
 *   factors and the hue shift in radians; stages: HOST int array, n_stages <= 4 ids in application order
 *   (0 brightness, 1 contrast, 2 saturation, 3 hue; contrast at most once); gray_sum (B) fp64 DEVICE
 *   workspace zeroed by the caller, required iff a contrast stage is present; enable (4) uint8 DEVICE, indexed by
 *   stage id, 0 = skip that kind of stage (the reference's `(factor != neutral).any()` guards, evaluated on the
 *   device so that no host sync is needed), nullable = all stages on. */
int km_color_jitter_fwd(const void* x, void* y, const void* params, double* gray_sum, const void* enable, const int* stages,
                        int n_stages, int B, int H, int W, int dtype, void* stream);

/* ---- filters -------------------------------------------------------------------------------
 * Replaces F.pad + F.conv2d(groups = Bk*C) of filter2d (kornia/filters/filter.py:131-150).
 *   x (B,C,H,W) dtype; k (Bk,kH,kW) prepared taps (flipped for 'conv', normalised, rounded to the
 *   input dtype) stored in the compute dtype; sample b uses kernel b % Bk (filter.py:141-142);
 *   same = 1: output (B,C,H,W) with `border`; same = 0 ('valid'): (B,C,H-kH+1,W-kW+1). */
int km_filter2d_fwd(const void* x, const void* k, void* y, int B, int C, int H, int W, int Bk, int kH, int kW, int border,
                    int same, int dtype, void* stream);
int km_filter2d_bwd_input(const void* gy, const void* k, void* gx, int B, int C, int H, int W, int Bk, int kH, int kW,
                          int border, int same, int dtype, void* stream);
/* gk (Bk,kH,kW) float64 accumulators, zeroed by the caller */
int km_filter2d_bwd_kernel(const void* gy, const void* x, void* gk, int B, int C, int H, int W, int Bk, int kH, int kW,
                           int border, int same, int dtype, void* stream);
/* Replaces filter2d_separable (filter.py:155-207) = GaussianBlur2d's path (gaussian.py:109-115):
 * both passes in one launch, intermediate kept in LDS.  kx (Bk,kW), ky (Bk,kH) compute dtype. */
int km_filter2d_sep_fwd(const void* x, const void* kx, const void* ky, void* y, int B, int C, int H, int W, int Bk, int kH,
                        int kW, int border, int same, int dtype, void* stream);
int km_filter2d_sep_bwd_input(const void* gy, const void* kx, const void* ky, void* gx, int B, int C, int H, int W, int Bk,
                              int kH, int kW, int border, int same, int dtype, void* stream);
/* which fused separable kernels accept this kernel size (LDS budget): bit 0 = km_filter2d_sep_fwd, bit 1 =
 * km_filter2d_sep_bwd_input (without it the caller runs the adjoint as two km_filter2d_bwd_input passes) */
int km_filter2d_sep_supported(int kH, int kW, int same, int dtype);

/* ---- spatial gradient / sobel --------------------------------------------------------------
 * Replaces spatial_gradient (kornia/filters/sobel.py:59-72) and the magnitude of sobel (:164-171).
 *   kern_host: HOST pointer, (n_out,kS,kS) derivative stack in the compute dtype (by-value kernel arg)
 *   out (B,C,n_out,H,W) nullable; mag (B,C,H,W) nullable = sqrt(gx*gx + gy*gy + eps), n_out == 2 */
int km_spatial_gradient_fwd(const void* x, const void* kern_host, void* out, void* mag, int B, int C, int H, int W,
                            int n_out, int kS, double eps, int dtype, void* stream);
int km_spatial_gradient_bwd(const void* gout, const void* kern_host, void* gx, int B, int C, int H, int W, int n_out,
                            int kS, int dtype, void* stream);

/* ---- Canny: back half -----------------------------------------------------------------------------
 * Replaces the elementwise / fixed-kernel-convolution tail of canny (kornia/filters/canny.py:119-159) after the Gaussian blur and
 * the Sobel derivatives (km_filter2d_sep_fwd, km_spatial_gradient_fwd).  fp32.
 * km_canny_nms_fwd: magnitude sqrt(gx^2 + gy^2 + eps) (:120), direction binning (:123-127), non-maximum suppression against the two
 *   neighbours along the gradient (:129-146, the 8 one-hot difference kernels of kornia/filters/kernels.py:943-976) and the two
 *   thresholds (:149-153).   grads (B,2,H,W) = spatial_gradient's (B,1,2,H,W);  mag, edges (B,H,W) written (edges: 0 / 0.5 / 1).
 * km_canny_hysteresis_sweep: the `while` loop of :156-176 (weak pixels touching strong ones are promoted until nothing changes) as
 *   block-local fixed points: state (B,H,W) in place, out (B,H,W) = 1 where state is strong else 0 (rewritten by every sweep),
 *   *changed (device int) OR-ed with 1 if a pixel was promoted - zero it, sweep, read it back, repeat while set. */
int km_canny_nms_fwd(const void* grads, void* mag, void* edges, int B, int H, int W, double low, double high, double eps, void* stream);
int km_canny_hysteresis_sweep(void* state, void* out, int* changed, int B, int H, int W, void* stream);

/* ---- image pyramid ---------------------------------------------------------------------------
 * km_pyrdown_fwd replaces pyrdown (kornia/geometry/transform/pyramid.py:409-453): filter2d with the fixed 5x5
 * binomial kernel / 256 (:32-47) and border mode `border` (codes as km_filter2d_fwd), then
 * F.interpolate(size=(oh, ow), mode='bilinear', align_corners=align) - fused, the blurred image is never stored.
 * The caller passes oh = int(H / factor), ow = int(W // factor) (:449).
 * km_resize_bilinear_fwd replaces the F.interpolate(mode='bilinear') call of pyrup (:494-496).
 *   x (B,C,H,W), y (B,C,oh,ow), same dtype. */
int km_pyrdown_fwd(const void* x, void* y, int B, int C, int H, int W, int oh, int ow, int border, int align,
                   int dtype, void* stream);
int km_resize_bilinear_fwd(const void* x, void* y, int B, int C, int H, int W, int oh, int ow, int align, int dtype,
                           void* stream);
/* adjoint of the resize (aten::upsample_bilinear2d_backward): gy (B,C,oh,ow) -> gx (B,C,H,W), written completely;
 * gather form, deterministic */
int km_resize_bilinear_bwd(const void* gy, void* gx, int B, int C, int H, int W, int oh, int ow, int align, int dtype,
                           void* stream);

/* ---- masked photometric loss of a warp + its matrix gradient, one launch ------------------------
 * Replaces ImageRegistrator.get_single_level_loss (kornia/geometry/transform/image_registrator.py:225-245) and its
 * autograd backward wrt the model: warp of src and of a ones image (HomographyWarper = homography_warp, bilinear,
 * zeros padding), loss_fn(warped, dst, reduction='none'), masked_select(ones > threshold), mean.
 *   src (B,C,H,W), dst (B,C,h,w): same dtype (f32 / bf16 / f16); mat (B_M,9) fp32, B_M in {1, B};
 *   coord_mode / norm_coords / align as km_warp2d_fwd (HomographyWarper: KM_COORD_HOMOGRAPHY, 1, 0);
 *   loss_kind 0: |w - d| (F.l1_loss), 1: (w - d)^2 (F.mse_loss);
 *   acc: (B, 11) fp64, zeroed by the caller, per image b:  acc[b][0] = sum of the selected elementwise losses,
 *   acc[b][1] = number of selected elements, acc[b][2 + k] = d acc[b][0] / d mat[b][k] (for B_M == 1: its contribution to
 *   d / d mat[0][k]).   loss = sum_b acc[b][0] / sum_b acc[b][1];  d loss / d mat = acc[.][2..] / sum_b acc[b][1]. */
int km_warp_masked_loss(const void* src, const void* dst, const void* mat, double* acc, int B, int C, int H, int W,
                        int h, int w, int B_M, int coord_mode, int norm_coords, int align, int loss_kind,
                        double threshold, int dtype, void* stream);

/* The rest of get_single_level_loss and of its backward (image_registrator.py:240-245: `.masked_select(...).mean()`, and autograd's
 * product with the upstream gradient), as two small launches instead of ~nine torch ones:
 * km_warp_masked_loss_finish: acc (B,11) as written above -> loss[0] = sum_b acc[b][0] / sum_b acc[b][1] (fp64; also as fp32 in loss_f32
 *   when non-null; 0 / 0 = NaN like the mean of an empty selection) and gm_unit (B_M,9) fp64 = d loss / d mat (summed over the batch
 *   for B_M == 1).  Deterministic (fixed summation order).
 * km_scale_f64: out[k] = (out dtype)(in[k] * scale[0]), the product in fp64, rounded once; scale: ONE value on the device
 *   (the upstream gradient of the scalar loss); scale_dtype / out_dtype 0 (f32) | 1 (f64). */
int km_warp_masked_loss_finish(const double* acc, int B, int B_M, double* loss, float* loss_f32, double* gm_unit, void* stream);
int km_scale_f64(const double* in, const void* scale, int scale_dtype, void* out, int out_dtype, long long n, void* stream);

/* ---- transform_points ----------------------------------------------------------------------
 * Replaces kornia/geometry/linalg.py:183-239 (+ conversions.py:247-339).  T (B_T,D+1,D+1),
 * pts/out (B,N,D), D in {2,3}, B_T in {1,B}; dtype 0 | 1. */
int km_transform_points_fwd(const void* T, const void* pts, void* out, int B, int N, int D, int B_T, int dtype, void* stream);
/* gpts (B,N,D) nullable; gT (B_T,(D+1)^2) float64 accumulators, zeroed by the caller, nullable */
int km_transform_points_bwd(const void* gout, const void* T, const void* pts, void* gpts, void* gT, int B, int N, int D,
                            int B_T, int dtype, void* stream);

/* ---- augmentation layer ---------------------------------------------------
 * km_gaussian_taps_fwd replaces get_gaussian_kernel1d x 2 inside gaussian_blur2d for a per-sample sigma
 * (kornia/filters/kernels.py:77-120, kornia/filters/gaussian.py:111-114; RandomGaussianBlur.apply_transform,
 * kornia/augmentation/_2d/intensity/gaussian_blur.py:95-114):  sigma (B,2) fp32 = (sigma_y, sigma_x) per sample,
 * taps_x (B,kx) from sigma[:,1], taps_y (B,ky) from sigma[:,0], fp32, normalised to sum 1; 1 <= kx, ky <= 64.
 * km_select_samples_fwd replaces the per-sample probability blend of _AugmentationBase.transform_inputs
 * (kornia/augmentation/base.py:348-393, torch.where(to_apply, transformed, input)):
 * out[b] = apply[b] ? transformed[b] : original[b], apply (B) uint8 on the device, n_per_sample elements of dtype per sample.
 * The same switch folded INTO the kernels, so that the blend costs no pass of its own: km_gaussian_taps_fwd's `apply` (nullable; a
 * sample whose entry is 0 gets the identity kernel - odd sizes), km_warp2d_fwd_masked and km_color_jitter_fwd_masked (the arguments of
 * km_warp2d_fwd / km_color_jitter_fwd plus `apply`; a sample whose entry is 0 is copied; the warp needs h == H, w == W). */
int km_gaussian_taps_fwd(const void* sigma, const void* apply, void* taps_x, void* taps_y, int B, int kx, int ky, void* stream);
/* The same for RandomGaussianBlur.apply_transform's own call (kornia/augmentation/_2d/intensity/gaussian_blur.py:95-114: ONE sigma per sample,
 * images of any supported dtype) with what the host layer would otherwise spend ATen launches on folded in: sigma (B,2) [per_axis != 0] or
 * (B) [per_axis == 0: the same sigma for both axes - no expanded copy]; batch_prob (B) fp32, nullable: the layer's probability draw,
 * thresholded here (`> 0.5` blurred, else the identity kernel; odd sizes) - no comparison launch, no uint8 copy; round_dtype (KM_F32 /
 * KM_BF16 / KM_F16): the taps rounded to the image's dtype and kept as fp32 values - filter2d's cast of its kernel to the input dtype
 * (kornia/filters/filter.py:126) without the two cast round trips. */
int km_gaussian_taps_dtype_fwd(const void* sigma, int per_axis, const void* batch_prob, void* taps_x, void* taps_y, int B, int kx, int ky,
                               int round_dtype, void* stream);
int km_warp2d_fwd_masked(const void* src, const void* mat, void* dst, const void* apply, int B, int C, int H, int W, int h,
                         int w, int B_M, int coord_mode, int norm_coords, int interp, int pad, int align, const void* fill,
                         int dtype, void* stream);
int km_color_jitter_fwd_masked(const void* x, void* y, const void* params, double* gray_sum, const void* enable,
                               const void* apply, const int* stages, int n_stages, int B, int H, int W, int dtype, void* stream);

/* Backward of km_color_jitter_fwd(_masked) - the reference's adjustments are differentiable through autograd
 * (kornia/enhance/adjust.py:80-593, kornia/color/hsv.py:27-131).  gx = d loss / d x given gy, same stage list / params / enable /
 * apply as the forward.  gray_sum: (B) fp64, the forward's workspace as the forward left it; gsum: (B) fp64 workspace zeroed by
 * the caller - both required iff a contrast stage is present.  gparams: (B,4) fp64 accumulators zeroed by the caller
 * (d loss / d params, hue in radians), or NULL. */
int km_color_jitter_bwd(const void* x, const void* gy, void* gx, const void* params, const double* gray_sum, double* gsum, double* gparams,
                        const void* enable, const void* apply, const int* stages, int n_stages, int B, int H, int W, int dtype, void* stream);
int km_select_samples_fwd(const void* transformed, const void* original, const void* apply, void* out, int B,
                          long long n_per_sample, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KORNIA_AMD_H */
