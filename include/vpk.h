/*
 * vpk.h -- C-ABI of the MI355X-native vanishing-point hot path (libvpk.so).
 *
 * Drop-in boundary for the hot path of fkluger/vanishing_points_2017.  The reference is pure
 * Python and has no FFI of its own; its boundary for this path is the call surface of
 * evaluation.py (run_cnn :254, caffe_forward :34, run_em :295, run_em_single :332) plus
 * vp_localisation.expectation_maximisation (:168-172).  Each entry point below names the
 * reference interface it replaces (file:line under /root/reference).  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns 0 on success or a negative vpk_status code; vpk_last_error()
 *     returns a human-readable message for the last failure on that handle.
 *   - unless a parameter is marked [host], buffers are DEVICE pointers (HBM-resident, e.g. a
 *     torch tensor's data_ptr()); all work is enqueued on the handle's HIP stream
 *     (vpk_set_stream) and is asynchronous with respect to the host unless stated.
 *   - one handle per process/GPU; a handle is thread-compatible (not thread-safe).
 *   - there is NO CPU fallback: on a machine without a gfx950 device vpk_create fails.
 */
#ifndef VPK_H_
#define VPK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPK_VERSION 110   /* 110 (round 6): + vpk_cnn_calibrate, vpk_cnn_get/set_activation_scales, vpk_cnn_range_flags, VPK_ERR_RANGE */

typedef struct vpk_handle vpk_handle;

enum vpk_status {
    VPK_OK = 0,
    VPK_ERR_ARG = -1,       /* bad argument (null pointer, size out of range)   */
    VPK_ERR_HIP = -2,       /* a HIP runtime call failed                        */
    VPK_ERR_NO_DEVICE = -3, /* no usable gfx950 device                          */
    VPK_ERR_STATE = -4,     /* call order (e.g. cnn_forward before cnn_load)    */
    VPK_ERR_LIMIT = -5,     /* problem exceeds a compiled-in limit (max VPs...) */
    VPK_ERR_RANGE = -6      /* the CNN's fp16-pair arithmetic left its calibrated value range (vpk_cnn_range_flags) */
};

/* per-image EM result status (status_out of vpk_em_batch) */
enum vpk_em_status {
    VPK_EM_OK = 0,            /* VPs returned                                                   */
    VPK_EM_NO_VP = 1,         /* reference returns the all-None result (vp_localisation.py:258-260,
                                 :402-404) -- also used where the reference would raise on an empty
                                 argmax (M == 0 at :349)                                         */
    VPK_EM_NO_INITIAL_VP = 2, /* reference raises ValueError from np.vstack([]) (:165)           */
    VPK_EM_NO_SLOT = 3        /* time-sliced launches only: no working-set slot became free within the launch's
                                 bounded wait (cannot happen while the documented slot invariant holds); the image
                                 was NOT refined and its other outputs are unset -- an error, never a result */
};
/* bit flags OR-ed into flags_out: situations where third-party tie-breaking is implementation
 * defined (Python heapq order inside sklearn's AgglomerativeClustering, vp_localisation.py:574) */
#define VPK_EM_FLAG_SPLIT_TIE 1u         /* exact tie between cluster distances during a split   */
#define VPK_EM_FLAG_SPLIT_DISCONNECTED 2u /* all-parallel line set: sklearn would complete graph */
#define VPK_EM_FLAG_VP_OVERFLOW 4u       /* more VPs than max_vp: result truncated               */

/* Mirrors the keyword defaults of expectation_maximisation (vp_localisation.py:168-172). */
typedef struct vpk_em_params {
    int32_t num_iter;          /* 100   */
    int32_t do_merge;          /* 1     */
    int32_t do_split;          /* 1     */
    int32_t do_iterations;     /* 1     */
    int32_t use_weights;       /* 1     */
    int32_t num_init_vp;       /* 25    */
    int32_t split_merge_freq;  /* 10    */
    int32_t num_min_lines;     /* 3     */
    double wbias;              /* 1.0   */
    double merge_thresh;       /* 1e-3  */
    double outlier_thresh;     /* 1.96^2 */
    double final_convergence;  /* 5e-3  */
    double s_thresh;           /* 1e-200 */
} vpk_em_params;

/* ---- lifetime ------------------------------------------------------------------------------ */
/* replaces: caffe.set_mode_gpu(); caffe.set_device(gpu_id)  (evaluation.py:20-21) */
int vpk_create(int device, vpk_handle** out);
int vpk_destroy(vpk_handle* h);
/* use an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream */
int vpk_set_stream(vpk_handle* h, void* hip_stream);
/* the hipStream_t the handle enqueues on */
void* vpk_get_stream(const vpk_handle* h);
int vpk_synchronize(vpk_handle* h);
const char* vpk_last_error(const vpk_handle* h);
int vpk_version(void);
void vpk_em_default_params(vpk_em_params* p);
/* device properties as seen by the library: [0]=CU count, [1]=LDS bytes per block, [2]=gfx arch number */
int vpk_device_info(const vpk_handle* h, int32_t info[4]);
/* Upper bound on the persistent workgroups (= CUs held) of one vpk_em_batch launch; 0 = one per image up
 * to the CU count (lowest latency for a single call).  A pipeline that runs other kernels beside the EM
 * (the reference's run_cnn of the next batch, evaluation.py:254-292) sets about batch/3: the images
 * queue inside the launch, which then lasts about as long as its slowest image anyway and leaves the
 * other CUs to the CNN. */
int vpk_em_set_workgroups(vpk_handle* h, int max_workgroups);
/* Which kernel evaluates weight_matrix (vp_localisation.py:515-524) inside the EM.  0 (default): the row-sliced kernel
 * (partial sums never leave the wave, operands by DPP row broadcast) wherever its LDS panel fits, the round-1/2 kernels
 * elsewhere; 1: always the round-1/2 kernels; 2: for images of up to 448 lines the sparse kernel (the terms whose operand
 * p_vl * lweight is exactly zero -- more than four in five -- are left out; a wave per group of VPs, lsim staged through
 * LDS by DMA once per call; measured slower than the dense kernel, kept as an option).  All of them sum every (column,
 * VP) in the same order, so every output of vpk_em_batch / vpk_weight_matrix is bit-identical under the three settings
 * (tests/test_gpu_em.py): the switch exists for that test and for A/B timing.  Mode 1 also selects the earlier forms of the
 * other phases that have been rebuilt with the same arithmetic since (round 6: the E-step's one thread per line, the serial
 * VP compaction, the row-by-row pair pass of calc_lsim for images of 512 lines and more, the M-step's four loads in flight
 * there; vpk_pairwise honours it too) -- the same test therefore pins those rebuilds bit for bit. */
int vpk_em_set_smoother(vpk_handle* h, int mode);
/* LDS the EM workgroup may plan with for its weight_matrix operand panel and the split's cluster matrix, in doubles;
 * 0 (default) = everything a CU has beside the workgroup's state (~18 800).  A smaller budget sends images down the
 * paths that larger images take by necessity -- the chunked smoother (operands staged through LDS piece by piece), the
 * split's distance matrix and direction vectors in HBM.  Results agree to rounding, not to the bit: the chunked
 * smoother sums a column's rows in ONE chain where the in-LDS kernels sum eight row slices; the parity bar
 * (assignments exact, VPs 1e-4) holds either way, and the GPU tests use the switch to run those paths on the
 * reference's small goldens.  Applies to vpk_em_batch and vpk_weight_matrix. */
int vpk_em_set_lds_panel(vpk_handle* h, int doubles);

/* Time-sliced EM launches for pipelines (run_cnn of batch k+1 while the EM of batch k is unfinished,
 * evaluation.py:254-329).  The EM of a never-converging image takes 99 iterations (vp_localisation.py:256)
 * where the average image takes ~25, so a launch that runs every image to completion holds its CUs for the
 * slowest one.  With slice_ms > 0 every vpk_em_batch launch on this handle gets a time budget instead: an
 * image still iterating when it expires is suspended at the top of its next iteration (its state is parked in
 * HBM) and resumed -- by any workgroup, before fresh images -- in the next launch on the handle; images that
 * had not started yet are parked likewise.  Results are bit-identical to an uninterrupted run.  Consequences
 * for the caller: the outputs of a call are complete only after vpk_em_flush (or once later launches have
 * finished its images), and the INPUT buffers of a call (l, lp, cnn, sphere, init_vp) as well as its output
 * buffers must stay alive and untouched until then.  n_max: slots are sized for images of up to n_max lines
 * (a later batch with more lines, or with other EM parameters, needs a vpk_em_flush first).
 * slice_ms = 0 switches back (flushing first).  replaces: nothing in the reference (its run_em is one
 * sequential loop, evaluation.py:309-329); this is what lets CNN and EM share one GPU without a tail. */
int vpk_em_set_time_slice(vpk_handle* h, double slice_ms, int n_max);
/* enqueue one launch that runs every parked image to completion (asynchronous on the handle's stream) */
int vpk_em_flush(vpk_handle* h);

/* ---- CNN (AlexNet-500, cnn/deploy.prototxt:1-304) -------------------------------------------- */
/* replaces: caffe.Net(model_def, model_weights, caffe.TEST) + read_mean_blob
 * (evaluation.py:17-31).  blobs [host]: 16 host pointers to fp32 arrays in Caffe layout,
 * order conv1.w, conv1.b, conv2.w, conv2.b, ..., conv5.b, fc6.w, fc6.b, fc7.w, fc7.b, fc8.w,
 * fc8.b (OIHW / (out,in)); mean [host]: 500*500 fp32 (mean.binaryproto, 1x1x500x500).
 * Copies to HBM (synchronous), packs the weights for every arithmetic mode, and CALIBRATES the default mode's activation
 * scales: six forwards of three built-in rasters (batch 3) on the f32 direct kernels, tapped at the inputs of conv2..5, fc6 and
 * fc7 -- ~20 ms of GPU time; see vpk_cnn_calibrate.  The response maps of the default mode depend on those scales (to rounding,
 * not beyond: tests/test_gpu_cnn.py), which are a function of the weights and the mean alone -- two loads of one model give
 * the same bits.  If calibration fails the model stays unloaded. */
int vpk_cnn_load(vpk_handle* h, const float* const blobs[16], const float* mean);
/* The value range of the default arithmetic (vpk_cnn_set_algorithm(4): scaled fp16 pairs) and how it is kept.
 * A layer's input x reaches the matrix cores as the fp16 pair of s x, s a power of two per consuming layer (conv2..5, fc6, fc7).
 * s is chosen so that the LARGEST |x| of that blob over a calibration set lands in [32, 64): rasters whose activations are up to
 * 2^10 x those of the calibration set stay finite, and smaller values keep full precision (22 bits) down to 2^-13 of the
 * calibration maximum and an absolute error below 2^-31 of it beyond that.  The built-in set is a sparse raster, a raster of
 * 1000 blended strokes (the density of a 1000-line sphere image, evaluation.py:12-14) and the all-255 raster.
 *   vpk_cnn_calibrate      rasters: n x 500 x 500 uint8 (DEVICE, 4-byte aligned) of the caller's choice -- the scales are set
 *                          from THESE rasters' blob maxima alone; n = 0 (rasters ignored): the built-in set again.  Runs
 *                          6 x ceil(n / 8) forwards on the f32 direct kernels and waits for them.
 *   vpk_cnn_get/set_activation_scales   the six powers of two (inputs of conv2, conv3, conv4, conv5, fc6, fc7); set: each
 *                          must be a power of two (VPK_ERR_ARG otherwise).  Restoring a saved vector restores the bits.
 *   vpk_cnn_range_flags    fp16's largest finite number is 65 504.  Every kernel that writes a scaled pair clamps to it and ORs
 *                          bit li (1 = conv2 ... 4 = conv5, 5 = fc6, 6 = fc7: the CONSUMING layer) into a device word when a value
 *                          reached it; nothing non-finite is ever produced by the scaling.  This call waits for the handle's
 *                          stream, returns the word in *flags_out (may be NULL), clears it, and returns VPK_ERR_RANGE when it
 *                          was non-zero (vpk_last_error names the layers), VPK_OK otherwise: it covers every forward on the
 *                          handle since the previous call.  A flagged response map was computed from clamped activations -- it
 *                          is finite but NOT the net's: recalibrate, or switch to vpk_cnn_set_algorithm(2) (exact operands, no
 *                          range to leave).  The reference's f32 Caffe forward (evaluation.py:34-38) has no such limit; this
 *                          is how the limit of the faster arithmetic is made impossible to miss. */
int vpk_cnn_calibrate(vpk_handle* h, const uint8_t* rasters, int n);
int vpk_cnn_get_activation_scales(vpk_handle* h, float scales[6]);
int vpk_cnn_set_activation_scales(vpk_handle* h, const float scales[6]);
int vpk_cnn_range_flags(vpk_handle* h, uint32_t* flags_out);
/* replaces: caffe_forward (evaluation.py:34-38) for a batch: sphere B x 500 x 500 uint8 ->
 * out B x 20 x 20 fp32 (sigout).  max batch per call is unbounded (internally chunked).  `sphere` must be 4-byte
 * aligned (device allocations are). */
int vpk_cnn_forward(vpk_handle* h, const uint8_t* sphere, int batch, float* out);
/* debugging/parity: run the net and also return an intermediate blob by name index
 * (0=conv1 relu, 1=pool1, 2=conv2, 3=pool2, 4=conv3, 5=conv4, 6=conv5, 7=pool5, 8=fc6, 9=fc7,
 * 10=fc8 pre-sigmoid); tap_out must hold batch * blob_size fp32. */
int vpk_cnn_forward_tap(vpk_handle* h, const uint8_t* sphere, int batch, float* out, int tap,
                        float* tap_out);

/* How the net's layers are computed.  The reference runs Caffe in fp32 (deploy.prototxt, evaluation.py:20: cuDNN's pick of
 * algorithm per layer); every setting below keeps f32 operands, f32 accumulation and f32 results -- what changes is which matrix
 * instruction multiplies, in how many pieces the operands reach it, and where the sums are rounded.  Each setting's error against the SAME net evaluated in float64 is
 * measured by tests/test_gpu_cnn.py; the defaults are, at every tap, no further from it than the f32-input direct kernels.
 *
 * conv1 + relu1 + norm1 + pool1 (deploy.prototxt:9-55):
 *   3 (default)  ONE kernel on the bf16 matrix cores with EXACT operands: the uint8 raster is one bf16 piece, each weight the sum
 *                of three, `- mean` (evaluation.py:35) a constant map made in float64 at load -- three bf16 products per f32
 *                product, none of them rounded (csrc/cnn_conv1_pieces.hpp).  The 96 x 123 x 123 conv1 blob is never written.
 *   4            the same kernel with the weights as scaled fp16 PAIRS (two products: the raster's integers are exact fp16 numbers too);
 *                no faster -- the kernel is bound by its LDS epilogue, not by the matrix pipe (0.366 against 0.375 ms) -- and not the default
 *   1            ONE kernel on the f32-input matrix instructions (v_mfma_f32_16x16x4_f32: an f32 FMA chain over the 121 taps)
 *   2            the implicit-GEMM kernel with the fused LRN / pooling epilogue
 *   0            separate conv1 and LRN / pooling kernels (also used whenever tap 0 is requested) */
int vpk_cnn_set_fusion(vpk_handle* h, int mode);
/* Arithmetic of conv2..conv5 (deploy.prototxt:56-174) when vpk_cnn_set_algorithm is 0:
 *   0 (default)  f32-input matrix instructions (v_mfma_f32_32x32x2_f32): bit-for-bit an f32 FMA chain per output
 *   1            every f32 operand as the exact sum of three bf16 pieces, six bf16 matrix products per f32 product
 *                (everything above 2^-24 of the product), f32 accumulation, implicit GEMM (csrc/cnn_split_gemm.hpp) */
int vpk_cnn_set_precision(vpk_handle* h, int mode);
/* Algorithm of conv2..conv5 and fc6 (precision 0):
 *   4  (default) DIRECT convolutions (and fc6's weight stream) on the fp16 matrix cores, every f32 operand as a SCALED PAIR of fp16
 *      numbers h0 = fp16(s x), h1 = fp16(s x - h0) -- 22 of its 24 significand bits, the remainder below 2^-23 |x| -- and THREE exact
 *      products per f32 product (h0 h0', h0 h1', h1 h0'; the fourth, h1 h1', is at most 2^-22 and typically 2^-24 of the product).  s is a power of two: per layer
 *      for the weights (the largest lands in [2^13, 2^14)), per consuming layer for the activations (calibrated at load: vpk_cnn_calibrate
 *      above, with the range check that goes with it); the epilogue multiplies by the exact reciprocal.
 *      Sums as in 2: the products of a kernel row x 16 channels (all taps of a 3 x 3 layer) accumulate from zero and join the f32
 *      accumulator with ONE rounding.  Half the matrix instructions of 2 -- which matters because the matrix cores are
 *      power-limited with real operands (1.7 PFLOP/s sustained against the 2.5 dense peak, scripts/ubench/mfma_f16_pairs.hip).
 *      Error against the float64 net (B = 3, scale of each blob): conv2 0.25e-6, conv3..5 0.4-0.5e-6, fc6 0.4e-6 -- at every tap
 *      below 2's and 2-4 x below the f32 direct kernels' (csrc/cnn_conv_pieces.hpp, csrc/cnn_dense_pieces.hpp)
 *   2  conv2 (5 x 5, 2 x 48 -> 128 channels) and fc6 on the bf16 matrix cores with EXACT operands -- three bf16 pieces per operand,
 *      six products per f32 product, block sums as in 4 (error 0.2e-6; the f32 direct kernel's is 1.0e-6) --; conv3..conv5 as in 1
 *   1  Winograd's minimal filtering on the f32-input matrix instructions: conv2 by F(2 x 2, 5 x 5) (36 instead of 100 products per
 *      2 x 2 outputs and input channel), conv3..conv5 by F(2 x 2, 3 x 3) (16 instead of 36); input / output transforms in f32.
 *      Same arithmetic type and accumulation width as the direct form; the rounding differs (products of transformed
 *      operands, shorter sums): error 0.6-0.8e-6 against 1.0-2.0e-6 for the direct kernels (csrc/cnn_winograd.hpp)
 *   0  direct: implicit GEMM over the taps -- every output is one f32 FMA chain over (channel, tap)
 *   3  (measurements) conv2, conv3 and conv5 as in 2's conv2, conv4 as in 1 */
int vpk_cnn_set_algorithm(vpk_handle* h, int mode);

/* per-layer device time of the last vpk_cnn_forward (single chunk), from HIP events recorded on
 * the handle's stream between the layers: ms[13] = conv1, norm1, pool1, conv2, norm2, pool2, conv3,
 * conv4, conv5, pool5, fc6, fc7, fc8 (with the fused first stage: conv1 = the whole fused kernel, norm1 = pool1 = 0).
 * vpk_cnn_last_layer_ms waits for the pass to finish. */
int vpk_cnn_set_profiling(vpk_handle* h, int on);
int vpk_cnn_last_layer_ms(vpk_handle* h, float ms[13]);
/* the same, averaged over the profiled passes since the last vpk_cnn_set_profiling call (at most the last 64); *passes
 * (may be NULL) = how many.  Waits for them to finish. */
int vpk_cnn_mean_layer_ms(vpk_handle* h, float ms[13], int* passes);

/* ---- sphere rasteriser (sphere_mapping.py:36-72) ----------------------------------------------- */
/* replaces: get_sphere_image / sphere_line_plot (evaluation.py:12-14).  l: sum(N) x 3 fp64
 * homogeneous lines, offsets [host]: B+1 int64 prefix of line counts; out: B x size x size uint8,
 * image row 0 = beta = +pi/2.  alpha = per-line blend weight (0.1 in the reference).  size 8..1024.
 * The reference's matplotlib / Agg pipeline stage by stage (10 000 samples per line, PathSimplifier, 1 pt stroke,
 * anti-aliased scanline coverage, plain 8-bit "over" in line order, black axes spines last): pixel-exact.
 * Asynchronous on the handle's stream, except that `offsets` (caller-owned host memory) is uploaded and waited for when
 * it differs from the previous call's on this handle -- a pipeline that rasterises the same batch structure again does
 * not wait for anything.  An image without lines gets the frame-only canvas, like the reference's; when the whole
 * batch has no line, l may be NULL.  Workspace (kept on the handle, grown on demand), per line of the largest chunk of
 * <= 49 152 lines: ~38 KB of outline scratch + 64 x size bytes of row table + 16 KB of coverage pool (~86 KB per line at
 * size 500: 4.2 GB for a full chunk; the pool is never smaller than 8 x size x (size + 2) bytes and must stay below
 * 4 GiB -- an image of more than ~262 000 lines is refused with VPK_ERR_LIMIT). */
int vpk_sphere_raster(vpk_handle* h, const double* l, const int64_t* offsets, int batch, int size,
                      double alpha, uint8_t* out);
/* replaces: the `alternative` keyword of sphere_line_plot (sphere_mapping.py:58-59; no caller in the reference passes it):
 * on != 0 makes the following vpk_sphere_raster calls on this handle draw beta = atan(-c / (a cos alpha + b sin alpha))
 * instead of atan((-a sin alpha - c cos alpha) / b). */
int vpk_sphere_raster_set_alternative(vpk_handle* h, int on);
/* per-image flags of the LAST vpk_sphere_raster call on this handle (waits for it): bit 0 = a line produced more outline
 * vertices / coverage than the kernel's buffers hold and was truncated or dropped (never seen on real line sets: a
 * simplified curve keeps 30-100 of its 10 000 samples).  `batch` must be that call's batch (VPK_ERR_ARG otherwise). */
int vpk_sphere_raster_flags(vpk_handle* h, int batch, uint32_t* flags_out);

/* ---- front end: line segment detection (HOST code, host pointers) -------------------------------------------- */
/* replaces: lsdpython.lsd.detect_line_segments(image) as called by detect_lsd_lines (evaluation.py:227-251; the
 * detector itself is an un-vendored submodule of the reference, .gitmodules:1-3 -- parity unpinned, see
 * csrc/vpk_lsd.cpp).  image [host]: height x width fp64 grey levels 0..255, row-major; scale: Gaussian sub-sampling
 * factor (0.8 = LSD's default).  out [host]: up to max_segments rows of 7 doubles (x1, y1, x2, y2, width, p,
 * -log10(NFA)) in pixel coordinates of the input image; *n_out = number of segments found (if it exceeds
 * max_segments only the first max_segments were written: call again with a larger buffer).  No GPU involved. */
int vpk_lsd_detect(const double* image, int width, int height, double scale, double* out, int max_segments, int* n_out);

/* ---- EM refinement (vp_localisation.py:168-450) ------------------------------------------------ */
/* replaces: run_em / run_em_single -> expectation_maximisation (evaluation.py:295-354) for a
 * batch of images.  One workgroup runs the whole EM of one image; images are independent.
 *   offsets  [host] B+1 int64 prefix sums of per-image line counts N_b
 *   l        sum(N) x 3 fp64, normalised IN PLACE (reference :185-186,:226)
 *   lp       sum(N) x 4 fp64 segment end points (x1,y1,x2,y2)
 *   cnn      B x 400 fp32 (20x20 sigout, row = beta bin)
 *   sphere   B x size x size uint8
 *   init_vp  NULL or B x n_init x 3 fp64 (reference keyword init_vp)
 *   max_vp   row capacity of the per-image outputs below
 * outputs (device): vp_out B x max_vp x 3, sigma_out / counts_out / counts_w_out B x max_vp,
 *   num_vp_out B, assoc_out sum(N) int64 (-1 = outlier), iterations_out B, status_out B
 *   (vpk_em_status), flags_out B (VPK_EM_FLAG_*), metric_out NULL or sum(N) x max_vp fp64
 *   (decision_metric, [line][vp]), trace_out NULL or B x (num_iter+1) x 12 fp64
 *   (row i: M after the M-step, max_err, M at the end of the iteration, event bits, then device
 *   microseconds spent in E-step / smoothing / M-step / whole iteration; row num_iter: microseconds
 *   of pairwise setup, remaining setup, whole image, M after the final merge / hard M-step / winner
 *   selection, then microseconds inside the single-pass smoother: staging + reduction, main loop). */
int vpk_em_batch(vpk_handle* h, int batch, const int64_t* offsets, double* l, const double* lp,
                 const float* cnn, const uint8_t* sphere, int sphere_size, const double* init_vp,
                 int n_init, const vpk_em_params* p, int max_vp, double* vp_out, double* sigma_out,
                 double* counts_out, double* counts_w_out, int32_t* num_vp_out, int64_t* assoc_out,
                 int32_t* iterations_out, int32_t* status_out, uint32_t* flags_out,
                 double* metric_out, double* trace_out);
/* EM_result['distribution'] (vp_localisation.py:441: the probability_functions.PDF of the LAST calc_probabilities call,
 * probability_functions.py:99-120) of every image of the NEXT vpk_em_batch call (the setting is consumed by that call; a
 * NULL argument clears it).  Device buffers, rows beyond an image's VP count are zero:
 *   p_v B x max_vp (PDF.v), angles B x max_vp x 2 (PDF.angles: alpha, beta), p_l sum(N) (PDF.l),
 *   p_lv sum(N) x max_vp (PDF.lv, [line][vp]), p_vl sum(N) x max_vp (PDF.vl transposed to [line][vp]),
 *   lvsq sum(N) x max_vp (PDF.lvsq).  Not available together with time-sliced launches. */
typedef struct vpk_em_dist_out {
    double* p_v;
    double* angles;
    double* p_l;
    double* p_lv;
    double* p_vl;
    double* lvsq;
} vpk_em_dist_out;
int vpk_em_set_distribution_out(vpk_handle* h, const vpk_em_dist_out* d);
/* bytes of device workspace the next vpk_em_batch with these sizes will hold (informational) */
size_t vpk_em_workspace_bytes(const vpk_handle* h, int batch, int n_max, const vpk_em_params* p,
                              int n_init);

/* ---- one pipeline step as one host call ------------------------------------------------------- */
/* replaces: one batch's worth of run_cnn followed by run_em (evaluation.py:254-329) in a pipeline that overlaps
 * consecutive batches: the CNN forward on `cnn`'s stream, then -- ordered behind it by an event, without a host wait
 * -- a copy of the resident lines into l_work, the EM on `em`'s stream with the CNN's response maps as its prior, and
 * optionally the fixed-size result records a multi-GPU run gathers (layout of vpk_build_records).  Everything is
 * enqueued from C++: the host spends tens of microseconds per step.  Buffers as for vpk_cnn_forward / vpk_em_batch;
 * `events`: NULL or four hipEvent_t of the caller (any may be NULL) recorded before / after the CNN on its stream and
 * before / after the EM on its stream.  `cnn` and `em` may be the same handle (one stream: the stages run in turn).
 * The buffers of a step must not be reused before its EM has finished: `reuse_event` expresses that on the device. */
typedef struct vpk_step_args {
    const uint8_t* sphere;          /* B x sphere_size x sphere_size */
    int32_t batch, sphere_size;
    float* cnn_out;                 /* B x 400: response maps (kept: they are the EM's input) */
    const int64_t* offsets;         /* [host] B + 1 */
    const double* l_in;             /* sum(N) x 3 resident lines (not modified) */
    double* l_work;                 /* sum(N) x 3 working copy, normalised in place by the EM */
    const double* lp;               /* sum(N) x 4 */
    const double* init_vp;          /* NULL or B x n_init x 3 */
    int32_t n_init, max_vp;
    const vpk_em_params* params;
    double* vp_out; double* sigma_out; double* counts_out; double* counts_w_out;
    int32_t* num_vp_out; int64_t* assoc_out; int32_t* iterations_out; int32_t* status_out; uint32_t* flags_out;
    double* records;                /* NULL or B x vpk_record_width() */
    const int64_t* image_ids;       /* device, B (with records) */
    void* events;                   /* NULL or hipEvent_t[4] */
    void* reuse_event;              /* NULL or a hipEvent_t of the caller that guards these buffers: the CNN stream waits
                                       for it before it overwrites cnn_out, and it is recorded behind this step's EM --
                                       so a ring of vpk_step_args can be re-enqueued without host synchronisation */
    const float* em_prior;          /* NULL: the EM's prior is this step's cnn_out (the reference's flow, run_cnn then run_em);
                                       else B x 400 response maps to use instead -- the CNN still runs and still writes
                                       cnn_out (a pipeline whose priors were computed earlier, evaluation.py:285 stores them
                                       in the datum; bench.py's value_fixture_prior) */
} vpk_step_args;
int vpk_pipeline_step(vpk_handle* cnn, vpk_handle* em, const vpk_step_args* a);
/* Fixed-size result records, one row of vpk_record_width() doubles per image: [image id, status, m, (x, y, z) x 20,
 * line count x 20, NaN] with the m <= 20 best-supported VPs in descending order of their counts (calc_horizon.py:34-36:
 * what the horizon selection reads); the records every rank contributes to the final all_gather. */
int vpk_record_width(void);
int vpk_build_records(vpk_handle* h, int batch, int max_vp, const int64_t* image_ids, const double* vp,
                      const double* counts, const int32_t* num_vp, const int32_t* status, double* records);

/* ---- fine-grained entry points (single image; unit parity against the reference functions) ---- */
/* calc_lsim (vp_localisation.py:87-108) + line_rating_knn (:34-84) in one pass over the pairs:
 * lsim_out n x n (row stride n), lscore_out n (before the clip), langle_out n (lines_angles). */
int vpk_pairwise(vpk_handle* h, int n, const double* lp, double* lsim_out, double* lscore_out,
                 double* langle_out);
/* find_initial_vps (:111-165) + pdf_params (probability_functions.py:62-96):
 * v0_out num_max x 3, m0_out 1 int32, weights_out 400 fp32. */
int vpk_init_vps(vpk_handle* h, const float* cnn, const uint8_t* sphere, int sphere_size,
                 int num_max, double* v0_out, int32_t* m0_out, float* weights_out);
/* calc_probabilities (probability_functions.py:99-120): v m x 3, s m (floored in place),
 * outputs p_v m, lvsq [m][n], p_vl [m][n], p_l n. */
int vpk_estep(vpk_handle* h, int n, int m, const double* lp, const float* cnn, const double* v,
              double* s, double* p_v_out, double* lvsq_out, double* p_vl_out, double* p_l_out);
/* weight_matrix (vp_localisation.py:515-524): p_vl [m][n], lsim n x n -> w_out [m][n]. */
int vpk_weight_matrix(vpk_handle* h, int n, int m, const double* p_vl, const double* lweight,
                      const double* lsim, double bias, double* w_out);
/* calc_new_vanishing_point (:453-479) for every row of w [m][n]: vp_out m x 3, valid_out m. */
int vpk_mstep(vpk_handle* h, int n, int m, const double* l, const double* w, double* vp_out,
              int32_t* valid_out);
/* calc_vp_line_counts (:482-512, thresh = 1.96^2 at :248,:419): v m x 3, s m, w [m][n] (the decision metric),
 * lweight n -> counts_out m, counts_w_out m, assoc_out n int64 (-1 = outlier). */
int vpk_line_counts(vpk_handle* h, int n, int m, const double* lp, const double* v, const double* s, const double* w,
                    const double* lweight, double thresh, double* counts_out, double* counts_w_out, int64_t* assoc_out);
/* the clustering inside split_best_vp (:568-578): ldist n x n -> labels n (0/1), flags 1. */
int vpk_cluster2(vpk_handle* h, int n, const double* ldist, int32_t* labels_out,
                 uint32_t* flags_out);

/* diagnostics: y[i] = f(x[i]) for the device's double-precision exp / acos / asin / atan / sqrt / sin / cos / log (fn =
 * 0..7) as the EM kernels call them (same translation unit, same flags).  replaces: nothing -- it measures the premise
 * of the parity bar: the reference's probability_functions.py:99-176 evaluates these through NumPy / libm, and results
 * can only agree to the last bit where these functions do (tests/test_gpu_math.py reports max ulp and mismatch rate). */
int vpk_math_probe(vpk_handle* h, int fn, long long n, const double* x, double* y);

/* ---- horizon from the best orthogonal VP triplet, batched ------------------------------------------ */
/* replaces: calc_horizon.calculate_horizon_and_ortho_vp -- calc_horizon.py:19-225 -- called per image from the
 * scoring loop of benchmark.py:229-243 (maxbest = 20, theta_vmin = pi/10; theta_z = pi/4 at :19).
 *   vp / counts / num_vp  the outputs of vpk_em_batch (B x max_vp x 3, B x max_vp, B; device)
 *   order   B x maxbest int32 (device): np.argsort(counts[:M])[::-1][:maxbest] per image (:34-36); the caller
 *           supplies it because the order of equal counts is a property of its NumPy sort
 *   out     B x 15 fp64 (device): hP1 | hP2 | zVP | hVP1 | hVP2 (3 each), the first five returned values
 *   combo_out  B x 3 int32: best_combo (VP indices; [0,1,-1] / [0,0,-1] in the < 3 VP fallbacks, :200-217) */
int vpk_horizon_batch(vpk_handle* h, int batch, int max_vp, const double* vp, const double* counts,
                      const int32_t* num_vp, const int32_t* order, int maxbest, double theta_vmin, double theta_z,
                      double* out, int32_t* combo_out);

#ifdef __cplusplus
}
#endif
#endif /* VPK_H_ */
