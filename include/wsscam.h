/*
 * wsscam.h -- C ABI of libwsscam, the MI355X (gfx950) implementation of the
 * CAM pseudo-label hot path of lyndonchan/wsss-analysis.
 *
 * The reference has no FFI for this path: it is Python calling torch / Keras /
 * cv2 / pydensecrf.  Each entry point below names the reference code it
 * replaces (file:line under the reference tree).  The Python shim in
 * wsss-analysis_amd/wsscam binds exactly these symbols with ctypes and keeps
 * the reference's Python signatures and on-disk formats on top of them.
 *
 * Conventions
 *   - every function returns 0 (WSC_OK) or a negative wsc_status; nothing
 *     throws, nothing calls exit().  wsc_last_error() returns a thread-local
 *     message for the last failing call on this thread.
 *   - all data pointers named *_dev are DEVICE pointers (hipMalloc'ed memory,
 *     e.g. torch.Tensor.data_ptr() or wsc_malloc); pointers named *_host are
 *     host pointers.  The caller owns every buffer it passes in; the library
 *     owns only what a wsc_*_create returned.
 *   - a wsc_ctx is bound to one device and one HIP stream.  All work of a ctx
 *     is enqueued on that stream and is asynchronous unless stated; wsc_sync
 *     waits for it.  Distinct ctxs may be used from distinct threads.
 *   - there is no CPU fallback: if no gfx950 device is present
 *     wsc_ctx_create fails with WSC_ERR_NO_DEVICE.
 */
#ifndef WSSCAM_H
#define WSSCAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WSC_VERSION 100 /* 0.1.0 */

typedef enum wsc_status {
    WSC_OK = 0,
    WSC_ERR_INVALID = -1,     /* bad argument (shape, null pointer, enum) */
    WSC_ERR_NO_DEVICE = -2,   /* no HIP device / wrong architecture */
    WSC_ERR_HIP = -3,         /* a HIP runtime call failed */
    WSC_ERR_MISSING_KEY = -4, /* state-dict key missing (strict load) */
    WSC_ERR_SHAPE = -5,       /* state-dict tensor has the wrong shape */
    WSC_ERR_NOMEM = -6,
    WSC_ERR_KEY_RANGE = -7,   /* CRF lattice coordinate outside packed-key range */
    WSC_ERR_CAPACITY = -8,    /* CRF hash table / vertex capacity exceeded */
    WSC_ERR_RANGE = -9        /* an activation of an IEEE-half mode (f16, f16x3) left half's finite range (|v| >= 65504) and was
                                 saturated: the reference's fp32 (03b_irn/net/resnet50.py:11-14) has no such ceiling, so the maps
                                 of this ctx are NOT the reference's.  Sticky: returned by wsc_sync / wsc_memcpy_d2h until
                                 wsc_ctx_range_status(ctx, &f, 1) clears it.  Run the model in WSC_PREC_BF16X3 instead */
} wsc_status;

/* architectures: 03b_irn/net/{resnet50_cam,vgg16_cam,m7_cam}.py */
typedef enum wsc_arch {
    WSC_ARCH_RESNET50_CAM = 0, /* net/resnet50.py:57-108 + resnet50_cam.py:12-20,55-70 */
    WSC_ARCH_VGG16_CAM = 1,    /* net/vgg16.py:44 + common_cnn.py:128-141 + vgg16_cam.py:24-60 */
    WSC_ARCH_M7_CAM = 2,       /* net/m7.py:41 + m7_cam.py:22-57 */
    WSC_ARCH_RESNET50_IRN = 3, /* net/resnet50_irn.py:8-132,210-232 (EdgeDisplacement) */
    WSC_ARCH_VGG16_IRN = 4,    /* net/vgg16_irn.py:8-212,301-321 (EdgeDisplacement, ds_fac = 0.25) */
    WSC_ARCH_M7_IRN = 5        /* net/m7_irn.py:8-118,195-213 (EdgeDisplacement; edge map at 1/2 resolution) */
} wsc_arch;

/* arithmetic of the conv stack */
typedef enum wsc_precision {
    WSC_PREC_BF16 = 0,  /* bf16 operands, fp32 MFMA accumulation, bf16 activations in HBM */
    WSC_PREC_BF16X3 = 1, /* split-bf16 (hi+lo) operands, 3 MFMA products: fp32-class accuracy */
    WSC_PREC_F16 = 2,    /* IEEE half operands (11-bit significand, saturating), fp32 accumulation */
    WSC_PREC_F16X3 = 3   /* split-half (hi+lo, 22-bit significand) operands and activations, 3 MFMA products per K-slice with
                            both planes staged once: the fp32-class mode (reference arithmetic is fp32, SURVEY 8 header) */
} wsc_precision;

typedef struct wsc_ctx wsc_ctx; /* device + stream + workspace arena */
typedef struct wsc_net wsc_net; /* immutable packed weights of one CNN */
/* Path selectors of a context (wsc_ctx_set_option).  Each one chooses between two code paths that BOTH serve some inputs
 * in the default configuration (e.g. the Gaussian message is formed on chip only while a tile's vertex set fits the LDS) and
 * are held to identical bits by tests/test_gpu_crf.py and tests/test_gpu_irn.py; the selector forces the fallback for every
 * input so that a test (or a debugging session) can compare.  There are no environment switches in the library; tuning
 * knobs and timing-only ablations exist only in builds with -DWSC_AB_KNOBS (python __graft_entry__.py --ab). */
typedef enum {
    WSC_OPT_CRF_GAUSS_ON_CHIP = 0, /* default 1; 0: blur kernels + value-row gathers instead of gauss_msg_kernel */
    WSC_OPT_CRF_FUSED_BLUR = 1,    /* default 1; 0: three blur4 passes instead of blur3_tile_kernel (Gaussian lattice) */
    WSC_OPT_CRF_BLUR_ON_CHIP = 2,  /* default 1; 0: one blur4 launch per pass instead of blur_lds_kernel (bilateral lattice) */
    WSC_OPT_CRF_RANK_BALLOT = 3,   /* default 0; 1: ballot-matching rank walk for every tile of the lattice build */
    WSC_OPT_CRF_EMBED_FULL = 4,    /* default 0; 1: the 2048-slot LDS table for every tile of the lattice build */
    WSC_OPT_RW_TILED = 5,          /* default -1 (by batch size); 0: flat random-walk step; 1: tiled step */
    WSC_OPT_STEM_POOL_FUSED = 6,   /* default 1; 0: the f16x3 ResNet stem as conv_igemm + max-pool launches instead of stem_pool_kernel */
    WSC_OPT_CONV_WINDOW = 7,       /* default 1; 0: per-tap A tiles instead of the LDS input window of the f16x3 3x3 / stride 1 layers (same bits) */
    WSC_OPT_CAM_HEAD_STREAM = 8,   /* default 1; 0: the 1x1 CAM / Grad-CAM head through the tiled conv kernel instead of cam_head_kernel
                                      (the one selector whose two paths are equal to fp32 round-off, not bit-identical: K is summed in four quarters) */
    WSC_OPT_CRF_MSG_IN_UPDATE = 9, /* default 1; 0: gauss_msg_kernel + update_splat_kernel as two launches with E = -U + Gaussian message in HBM between
                                      them, instead of the message formed inside the update kernel (same bits) */
    WSC_OPT_COUNT = 10
} wsc_option;

typedef struct wsc_crf wsc_crf; /* lattices (Gaussian + bilateral) of a batch of images */
typedef struct wsc_crf_v wsc_crf_v; /* the same for a RAGGED batch: every image its own (H, W) and class count */

/* A named host tensor of a torch state_dict (float32, C-contiguous). */
typedef struct wsc_tensor_desc {
    const char *name;  /* e.g. "resnet50.layer1.0.conv1.weight" */
    const float *data; /* host pointer */
    int32_t ndim;
    int64_t shape[4];
} wsc_tensor_desc;

/* ---- library / context ------------------------------------------------ */

int wsc_version(void);
const char *wsc_last_error(void);

/* device: HIP device ordinal.  stream: a hipStream_t to enqueue on (e.g.
 * torch.cuda.current_stream().cuda_stream), or NULL to let the ctx create and
 * own a stream.  Replaces `model.cuda()` / `cuda.device(process_id)`
 * (03b_irn/step/make_cam.py:31-33). */
int wsc_ctx_create(int device, void *stream, wsc_ctx **out);
void wsc_ctx_destroy(wsc_ctx *ctx);
/* Sets a path selector (wsc_option) of the context; WSC_ERR_INVALID for an unknown option. */
int wsc_ctx_set_option(wsc_ctx *ctx, int option, int value);
int wsc_sync(wsc_ctx *ctx);
/* Range guard of the IEEE-half conv modes.  Every conv epilogue that writes half activations raises a sticky per-ctx flag when
 * a value reaches the half ceiling (it is stored saturated at +-65504; fp32 in the reference would have kept it).  wsc_sync and
 * wsc_memcpy_d2h report the raised flag as WSC_ERR_RANGE AFTER completing their work (the copy is done, the stream is idle).
 * *flag_out (may be NULL): 0 = clean, otherwise the output-channel count of a layer that saturated.  Synchronises the ctx
 * stream.  clear != 0 resets the flag. */
int wsc_ctx_range_status(wsc_ctx *ctx, int *flag_out, int clear);
/* Make all work enqueued on `ctx` after this call wait (on the device, without blocking the host) for
 * everything enqueued so far on `other`: lets one process overlap independent stages on two contexts
 * of the same device (e.g. the lattice build of a batch, which needs only the RGB images, with its
 * CNN forward pass) and join them before the stage that needs both. */
int wsc_ctx_wait(wsc_ctx *ctx, wsc_ctx *other);
/* Markers: wsc_ctx_mark records marker `slot` (0 .. 7) at the current end of the ctx's stream; wsc_ctx_wait_mark blocks the
 * HOST until everything enqueued before that record has completed (and returns at once for a slot never recorded).  A caller
 * that keeps several steps in flight on one stream waits for "the step two back" this way -- without a wsc_sync that would
 * also wait for the newest one, and without an extra stream whose wait packets can end up in front of another stream's
 * kernels on a shared hardware queue (replaces the `torch.cuda.synchronize()` granularity of 03b_irn/step/make_cam.py:81-85). */
int wsc_ctx_mark(wsc_ctx *ctx, int slot);
int wsc_ctx_wait_mark(wsc_ctx *ctx, int slot);
/* Device-side: work enqueued on `ctx` after this call waits for marker `slot` of `other` (its last record) -- and for nothing
 * `other` enqueued after that record, unlike wsc_ctx_wait.  No-op for a slot never recorded.  Same device. */
int wsc_ctx_wait_for_mark(wsc_ctx *ctx, wsc_ctx *other, int slot);
/* name of the device's gcnArchName ("gfx950...") and CU count */
int wsc_device_info(wsc_ctx *ctx, char *arch_name, size_t arch_name_len, int *num_cus);

/* device memory helpers so a host without torch can drive the library
 * (replace tensor.cuda() / .cpu(): make_cam.py:48,81-82) */
int wsc_malloc(wsc_ctx *ctx, size_t bytes, void **dptr_out);
int wsc_free(wsc_ctx *ctx, void *dptr);
int wsc_memcpy_h2d(wsc_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes); /* async on ctx stream */
int wsc_memcpy_d2h(wsc_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes); /* synchronises */
int wsc_memset(wsc_ctx *ctx, void *dst_dev, int value, size_t bytes);
/* Page-locked host staging for the batch pipeline (replaces DataLoader(pin_memory=...) + .cuda(non_blocking=True),
 * make_cam.py:29,45-48, and the .cpu() copies of :81-85): copies between pinned memory and the device are truly
 * asynchronous on the ctx stream; the host buffer must stay untouched until a later wsc_sync(ctx) returns. */
int wsc_host_alloc(wsc_ctx *ctx, size_t bytes, void **host_out);
int wsc_host_free(wsc_ctx *ctx, void *host);
/* Host I/O helper of the writer threads (03b_irn/step/make_cam.py:80-88 np.save per image): creates / truncates `path` and
 * writes the n byte segments one after the other (open + writev + close; no GPU involved, any thread).  A caller that holds
 * the bytes of an .npy container as [header | metadata | array | metadata | array ...] writes the arrays straight from where
 * they are (e.g. the page-locked staging buffer of a D2H copy) without the interpreter lock: wsscam.step.make_cam.save_npy_object. */
int wsc_host_write_segments(const char *path, int n, const void *const *ptrs, const size_t *sizes);
/* Asynchronous copies on the context's stream (page-locked host memory).  Scheduling note for callers with several contexts in
 * flight: the runtime hands a queued copy to a DMA engine at once and turns "after the kernels before it on the stream" into a
 * poll command on that engine's in-order queue -- every copy submitted later to the same engine, from ANY context, waits behind
 * it.  Issue a copy when its producers have (nearly) finished: wsc_ctx_mark after the producer, wsc_ctx_wait_mark on a helper
 * thread, then the copy (bench.py's end-to-end leg, wsscam.step.pipeline._copy_out); wsc_memcpy_d2h does that wait itself. */
int wsc_memcpy_h2d_async(wsc_ctx *ctx, void *dst_dev, const void *src_pinned_host, size_t bytes);
int wsc_memcpy_d2h_async(wsc_ctx *ctx, void *dst_pinned_host, const void *src_dev, size_t bytes);

/* timing on the ctx stream with HIP events (bench.py's roofline leg):
 * wsc_timer_begin/_end bracket a region; _end synchronises and returns ms. */
int wsc_timer_begin(wsc_ctx *ctx);
int wsc_timer_end(wsc_ctx *ctx, float *ms_out);

/* Per-kernel-class timing with HIP events on the ctx stream (bench.py's roofline object): between
 * _begin and _end every kernel launch of the ctx is bracketed by an event pair; _end synchronises and
 * returns, per class, the number of launches, their summed duration and their summed algorithmic
 * work (FLOPs for the conv classes, bytes otherwise).  Arrays hold max_classes entries. */
int wsc_profile_begin(wsc_ctx *ctx);
int wsc_profile_end(wsc_ctx *ctx, int max_classes, int32_t *calls_out, float *total_ms_out, double *work_out,
                    int *n_classes_out);
const char *wsc_profile_class_name(int cls);

/* ---- CNN + CAM head --------------------------------------------------- */

/* Build a network from a state_dict.  Replaces model construction +
 * load_state_dict(strict=True) + .eval() + .cuda()
 * (03b_irn/step/make_cam.py:96-100, 33).  Inference BatchNorm
 * (net/resnet50.py:11-14, eps from "<bn>.eps" if given else 1e-5) is folded
 * into per-channel scale/shift applied in the conv epilogue.
 * Missing keys -> WSC_ERR_MISSING_KEY, wrong shapes -> WSC_ERR_SHAPE. */
int wsc_net_create(wsc_ctx *ctx, int arch, const wsc_tensor_desc *weights, int n_weights,
                   int num_classes, int precision, wsc_net **out);
void wsc_net_destroy(wsc_net *net);
/* spatial size of the CAM for an S x S input (21 for resnet50 @321, 40 vgg16 @321, 56 m7 @224) */
int wsc_net_cam_size(const wsc_net *net, int S, int *h_out);
/* channels of the last conv feature map (2048 / 1024 / 256) */
int wsc_net_feat_channels(const wsc_net *net, int *f_out);

/* CAM.forward for a batch of B images (resnet50_cam.py:55-70, vgg16_cam.py:24-50):
 *   x_dev   float32 [B][2][3][S][S]  -- per image: original and h-flipped sample,
 *           exactly the "img" tensor of VOC12ClassificationDatasetMSF
 *           (voc12/dataloader.py:240)
 *   cam_dev float32 [B][C][h][w]     -- relu(conv1x1(feat)) of sample 0 plus the
 *           w-flipped map of sample 1
 *   score_dev float32 [B][C] or NULL -- sigmoid(Linear(GAP(feat of sample 0)))
 *           (vgg16_cam.py:34-36); NULL for resnet50 which has no classifier branch
 * Asynchronous on the ctx stream. */
int wsc_net_forward_cam(wsc_ctx *ctx, const wsc_net *net, const float *x_dev, int B, int S,
                        float *cam_dev, float *score_dev);
/* The same for NON-SQUARE network inputs x_dev float32 [B][2][3][H][W] -- the reference's resnet50 configuration
 * outsize = None (03b_irn/func_sample.py:143-148): every image goes through the network at its own size, so a device batch
 * holds images of one (H, W).  cam_dev float32 [B][C][h][w] with (h, w) from wsc_net_cam_size_hw. */
int wsc_net_cam_size_hw(const wsc_net *net, int H, int W, int *h_out, int *w_out);
int wsc_net_forward_cam_hw(wsc_ctx *ctx, const wsc_net *net, const float *x_dev, int B, int H, int W, float *cam_dev,
                           float *score_dev);

/* Last-conv feature map for Grad-CAM style heads (02_cues/utilities.py:129-132,
 * K.function([input],[conv_output])): feat_dev float32 [N][h][w][F] (NHWC as Keras). */
int wsc_net_forward_features(wsc_ctx *ctx, const wsc_net *net, const float *x_dev /*[N][3][S][S]*/,
                             int N, int S, float *feat_dev);

/* IRNet EdgeDisplacement.forward (03b_irn/net/resnet50_irn.py:210-232, vgg16_irn.py:301-321) for B images,
 * net created with WSC_ARCH_*_IRN from the EdgeDisplacement state dict (backbone keys as for the CAM nets,
 * `fc_edge<k>.0.weight`, `fc_edge<k>.1.{weight,bias}` (GroupNorm), `fc_edge6.{weight,bias}`, `fc_dp<k>...`,
 * `fc_dp7.3.weight`, `mean_shift.running_mean`):
 *   x_dev    float32 [B][2][3][S][S] -- [orig, h-flip] pairs already zero-padded on the right / bottom to the
 *            crop size S (F.pad(x, [0, S - w, 0, S - h]), step/make_sem_seg_labels.py:46 feeds pack['img'][0])
 *   edge_dev float32 [B][feat_h][feat_w] = sigmoid(edge[0]/2 + edge[1].flip(-1)/2), cropped to the feature size
 *            ((h-1)/stride+1, (w-1)/stride+1) of the un-padded image
 *   dp_dev   float32 [B][2][feat_h][feat_w] = displacement field of sample 0 (minus MeanShift.running_mean) */
int wsc_net_forward_edge(wsc_ctx *ctx, const wsc_net *net, const float *x_dev, int B, int S, int feat_h, int feat_w,
                         float *edge_dev, float *dp_dev);

/* misc.indexing.propagate_to_edge(x, edge, radius, beta, exp_times) -- the IRNet random walk called by
 * 03b_irn/step/make_sem_seg_labels.py:59,76,93 (the `misc` package is missing from the reference tree; the
 * algorithm is upstream IRNet's misc/indexing.py, its affinity step is in-tree as to_affinity,
 * net/vgg16_irn.py:247-261):  rw = (x * (1 - edge)) @ T^(n_steps), T = column-normalised affinity^beta.
 *   x_dev float32 [K][h][w], edge_dev float32 [h][w] (device); rw_dev float32 [K][h][w]
 *   dirs_host int32 [D][2] search directions (dy, dx) of PathIndex (dy > 0, or dy == 0 and dx > 0),
 *   path_start_host int32 [D+1], path_yx_host int32 [path_start[D]][2]: the pixels (relative to the source)
 *   of the straight path of every direction, end points included
 *   n_steps = 2^exp_times applications of T (the reference squares the dense matrix exp_times times) */
int wsc_rw_propagate(wsc_ctx *ctx, const float *x_dev, const float *edge_dev, int K, int h, int w,
                     const int32_t *dirs_host, const int32_t *path_start_host, const int32_t *path_yx_host, int D,
                     float beta, int n_steps, float *rw_dev);
/* The same for n_img images of different sizes in one pass (every stencil step is one launch over all of them --
 * a single 94 x 125 image is launch-bound at ~15 us per step): image b has K_host[b] maps on an h_host[b] x
 * w_host[b] grid; x_dev / rw_dev hold the images' [K][h][w] blocks back to back, edge_dev their [h][w] maps. */
int wsc_rw_propagate_batch(wsc_ctx *ctx, int n_img, const int32_t *K_host, const int32_t *h_host,
                           const int32_t *w_host, const float *x_dev, const float *edge_dev,
                           const int32_t *dirs_host, const int32_t *path_start_host, const int32_t *path_yx_host,
                           int D, float beta, int n_steps, float *rw_dev);

/* Grad-CAM for a plain batch of N samples (02_cues/utilities.py:128-133, 03c_hsn/utilities.py:258-263):
 *   cams[n][y][x][c] = [relu]( sum_f feat[n][y][x][f] * alpha[f][c] )      ('ijkl,lm->ijkm')
 * with alpha the `gradcam_weights` tensor given to wsc_net_create (vgg16 / m7).  One CNN pass
 * yields both the maps and, if score_dev != NULL, sigmoid classifier scores [N][C] -- the
 * reference runs the network twice (model.predict + K.function).
 *   x_dev float32 [N][3][S][S];  cams_dev float32 [N][h][w][C] (NHWC as numpy's einsum result). */
int wsc_net_forward_gradcam(wsc_ctx *ctx, const wsc_net *net, const float *x_dev, int N, int S, int relu,
                            float *cams_dev, float *score_dev);

/* One nn.Conv2d (+ per-channel scale/shift, residual add, ReLU) through the production
 * implicit-GEMM kernel with NCHW float32 tensors on the device -- the unit the per-layer
 * numerics tests drive (conv shape classes of SURVEY.md section 8 a4/a6).  w_host is OIHW.
 * Cin must be <= 4 (stem-style, small-Cin path) or a multiple of 64; Cout a multiple of 8.
 * precision may be OR-ed with WSC_CONV_GENERIC: the layer then runs on the kernel's generic variants instead of the
 * specialised ones the host picks for common layers (same arithmetic; a test holds the two to identical bits). */
#define WSC_CONV_GENERIC 0x100
int wsc_conv2d_nchw(wsc_ctx *ctx, const float *x_dev, int N, int Cin, int H, int W, const float *w_host,
                    int Cout, int kh, int kw, int stride, int pad, const float *scale_host,
                    const float *shift_host, const float *residual_dev, int relu, int precision,
                    float *y_dev);

/* ---- CAM tail ---------------------------------------------------------- */

/* make_cam._work tail for a batch (03b_irn/step/make_cam.py:41-42,62-76 with
 * misc.imutils.get_strided_size / get_strided_up_size):
 * for image b with original size (H0,W0)=size_hw[b], valid classes
 * keys[key_off[b] .. key_off[b+1]) :
 *   strided  = bilinear(cam[b], ((H0-1)/4+1, (W0-1)/4+1), align_corners=False)[keys]
 *   high_res = bilinear(cam[b], (((H0-1)/16+1)*16, ...))[keys][:, :H0, :W0]
 *   each channel divided by (its spatial max + 1e-5)
 * Outputs are packed back to back: image b's strided block starts at float
 * offset strided_off[b] (K_b*h4*w4 floats), high_res at highres_off[b]
 * (K_b*H0*W0 floats); the caller computes the offsets (prefix sums).
 * All descriptor arrays are HOST pointers (copied by the call). */
int wsc_cam_postprocess(wsc_ctx *ctx, const float *cam_dev, int B, int C, int h, int w,
                        const int32_t *size_hw_host /*[B][2]*/, const int32_t *keys_host,
                        const int32_t *key_off_host /*[B+1]*/, const int64_t *strided_off_host /*[B]*/,
                        const int64_t *highres_off_host /*[B]*/, float *strided_dev, float *highres_dev);

/* eval_cam for a batch straight from the packed high_res maps of wsc_cam_postprocess
 * (03b_irn/step/eval_cam.py:48-62 and chainercv.evaluations.calc_semantic_segmentation_confusion):
 *   cams = np.pad(high_res, ((1,0),...), constant_values=bg_thres); keys' = np.pad(keys + 1, (1,0))
 *   pred = keys'[np.argmax(cams, axis=0)];  confusion[gt][pred] += 1 where gt != ignore_label
 * gt_dev / pred_dev: uint8, images packed back to back (H0*W0 each, in batch order); pred_dev may be
 * NULL; gt_dev may be NULL (labels only).  confusion_dev int64 [n_class][n_class] is ACCUMULATED into, so
 * one matrix can be carried over a whole dataset (zero it first with wsc_memset). */
int wsc_cam_eval_confusion(wsc_ctx *ctx, const float *highres_dev, int B, const int32_t *size_hw_host,
                           const int32_t *keys_host, const int32_t *key_off_host,
                           const int64_t *highres_off_host, float bg_thres, const uint8_t *gt_dev, int n_class,
                           int ignore_label, uint8_t *pred_dev, int64_t *confusion_dev);

/* Probabilities -> CRF unaries for a [background | class maps] stack:
 *   v_0 = bg_value, v_{c+1} = maps[b][c][p];  U[b][m][p] = -log(clip(v_m / sum_m v_m, 1e-5, 1))
 * i.e. eval_cam.py:49-51 (np.pad(high_res, constant_values=cam_eval_thres)) followed by
 * pydensecrf.utils.unary_from_softmax (03c_hsn/utilities.py:431).
 *   maps_dev float32 [B][C][N]  ->  unary_dev float32 [B][C+1][N] */
int wsc_unary_from_maps(wsc_ctx *ctx, const float *maps_dev, int B, int C, int N, float bg_value,
                        float *unary_dev);

/* The ADP / DeepGlobe branch of eval_cam (03b_irn/step/eval_cam.py:53-63): no background padding, the keys are class ids,
 *   pred = keys[np.argmax(maps, axis=0)];  pred = cv2.resize(pred, outsize, interpolation=cv2.INTER_NEAREST)
 *   confusion[gt][pred] += 1 where gt != ignore_label
 * maps_dev: image b's [K_b][h_b * w_b] block at float offset maps_off[b] (high_res for ADP, the strided cam for DeepGlobe);
 * gt_dev / pred_dev: uint8 at the OUTPUT size (out_h * out_w per image, packed in batch order; 1088 x 1088 / 2448 x 2448 in
 * the reference).  The nearest-neighbour source index follows cv2: min(floor(x * (1. / (out / src))), src - 1) in double.
 * confusion_dev int64 [n_class][n_class] is accumulated into, as in wsc_cam_eval_confusion. */
int wsc_cam_eval_confusion_nn(wsc_ctx *ctx, const float *maps_dev, int B, const int32_t *src_hw_host /*[B][2]*/,
                              const int32_t *out_hw_host /*[B][2]*/, const int32_t *keys_host, const int32_t *key_off_host /*[B+1]*/,
                              const int64_t *maps_off_host /*[B]*/, const uint8_t *gt_dev, int n_class, int ignore_label,
                              uint8_t *pred_dev, int64_t *confusion_dev);

/* The evaluation tail of the HistoSegNet drivers (03c_hsn/demo.py:386-408): the CRF's label maps (int32, image b's h_b x w_b
 * map at int32 offset labels_off[b]) are resized to the ground truth's size with cv2's INTER_NEAREST rule (as above) and
 * counted:  confusion[gt][pred] += 1 where gt != ignore_label.  The reference's per-class intersections / unions / ground-
 * truth counts are sums over this matrix when every pixel whose colour matches no class is given the extra ground-truth
 * index n_class - 1 (wsscam.hsn.demo.evaluate_labels).  gt_dev / pred_dev uint8 at the output size, packed in batch order. */
int wsc_label_confusion_nn(wsc_ctx *ctx, const int32_t *labels_dev, int B, const int32_t *src_hw_host /*[B][2]*/,
                           const int32_t *out_hw_host /*[B][2]*/, const int64_t *labels_off_host /*[B]*/, const uint8_t *gt_dev,
                           int n_class, int ignore_label, uint8_t *pred_dev, int64_t *confusion_dev);

/* The tail of make_sem_seg_labels for a batch of images (03b_irn/step/make_sem_seg_labels.py:74-79 voc12, :91-96 ADP,
 * :113-118 DeepGlobe), straight from the random-walk maps wsc_rw_propagate_batch left in HBM:
 *   rw_up = F.interpolate(rw, size=up_hw, mode='bilinear', align_corners=False)[..., 0, :H0, :W0]
 *   rw_up = rw_up / torch.max(rw_up)                      (one maximum over all of the image's maps and pixels)
 *   has_bg: rw_up = F.pad(rw_up, (0,0,0,0,1,0), value=bg_thres)          (voc12; keys then hold K + 1 entries)
 *   label = keys[torch.argmax(rw_up, dim=0)]              (first maximum; an all-zero image gives NaN maps: the first of them)
 * rw_dev: image b's [K_b][h_b * w_b] maps at float offset rw_off[b]; khw = (K, h, w); label_dev uint8, H0 * W0 per image,
 * packed in batch order.  Keys must fit a uint8 label. */
int wsc_sem_seg_finish(wsc_ctx *ctx, const float *rw_dev, int B, const int64_t *rw_off_host /*[B]*/, const int32_t *khw_host /*[B][3]*/,
                       const int32_t *up_hw_host /*[B][2]*/, const int32_t *out_hw_host /*[B][2]*/, const int32_t *keys_host,
                       const int32_t *key_off_host /*[B+1]*/, int has_bg, float bg_thres, uint8_t *label_dev);

/* wsc_cam_postprocess (all C classes, every image at H0 x W0) followed by wsc_unary_from_maps, fused: the
 * max-normalised high-resolution maps are never written to HBM, only the unaries are (bit-identical to the
 * two-step path).  Replaces, for a whole batch, make_cam.py:64-76 (upsample to the strided-up size, crop,
 * x /= max + 1e-5) + eval_cam.py:49-51 (background channel) + pydensecrf.utils.unary_from_softmax
 * (03c_hsn/utilities.py:431).
 *   cam_dev float32 [B][C][h][w] (wsc_net_forward_cam)  ->  unary_dev float32 [B][C+1][H0*W0] */
int wsc_cam_unary(wsc_ctx *ctx, const float *cam_dev, int B, int C, int h, int w, int H0, int W0, float bg_value,
                  float *unary_dev);
/* The same with the unaries written PIXEL-major, rows padded to Mp = 4 * ceil((C+1)/4) floats (padding = 0): the layout
 * the mean-field loop reads, for wsc_crf_inference_pm (saves the class-major -> pixel-major pass of wsc_crf_inference).
 *   unary_pm_dev float32 [B][H0*W0][Mp] */
int wsc_cam_unary_pm(wsc_ctx *ctx, const float *cam_dev, int B, int C, int h, int w, int H0, int W0, float bg_value,
                     float *unary_pm_dev);

/* Multi-scale inference (args.cam_scales, 03b_irn/step/make_cam.py:62-69: `torch.sum(torch.stack([F.interpolate(o, size) for
 * o in outputs]), 0)` for the strided and the high-resolution maps): every scale of the MSF dataset is resized to the same
 * network input (voc12/dataloader.py:232-240), so the per-scale CAMs have one size and, bilinear interpolation being linear,
 * the sum of the interpolated maps is the interpolation of the summed map (up to fp32 rounding, ~1e-7 relative).  The scales
 * of an image are consecutive "images" of one wsc_net_forward_cam batch; this adds them in scale order:
 *   cam_dev float32 [n_images * n_scales][map_elems]  ->  out_dev float32 [n_images][map_elems]   (out_dev != cam_dev) */
int wsc_cam_sum_scales(wsc_ctx *ctx, const float *cam_dev, int n_images, int n_scales, long long map_elems, float *out_dev);

/* The MSF dataset transform of a batch of DECODED images (03b_irn/voc12/dataloader.py:68-106, 225-246; constants of
 * adp/dataloader.py:64-80, deepglobe/dataloader.py:60-66): float64 bilinear resize to S x S (cv2.resize INTER_LINEAR
 * on the float64 image, skipped when the image already has that size), float32 normalisation
 * (x - mean[c]) / std[c] (pre_div255: (x / 255 - mean[c]) / std[c], norm_mode 'float'), HWC -> CHW, and with
 * pair = 1 the [x, flip(x, -1)] stack make_cam feeds the network; pair = 0 gives the plain batch of
 * 02_cues/utilities.py:146-181.  Bit-identical to the numpy path of wsscam.voc12.dataloader (same float64 expression).
 *   images_dev uint8: image b is the HWC block [H0_b][W0_b][3] at byte offset_host[b];  size_hw_host int32 [B][2];
 *   x_dev float32 [B][2][3][S][S] (pair) or [B][3][S][S]. */
int wsc_msf_input_u8(wsc_ctx *ctx, const uint8_t *images_dev, int B, const int32_t *size_hw_host,
                     const int64_t *offset_host, int S, const float *mean3_host, const float *std3_host,
                     int pre_div255, int pair, float *x_dev);

/* cv2.resize(uint8 HWC image, (OW, OH)) with the default INTER_LINEAR, for a batch of images of different sizes:
 * read_batch of 02_cues/utilities.py:172-176 and 03c_hsn/utilities.py:170-181 keeps the resized batch as uint8, so the network
 * input AND the CRF image are OpenCV's 8-bit result.  OpenCV's 8U path is fixed point (11-bit coefficients, two passes, the
 * vertical pass (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2; exact 2 x 2 decimation = rounded box mean);
 * restated from the published algorithm, cv2 itself is absent offline (parity vs cv2 unpinned).
 *   images_dev uint8: image b is the HWC block [H0_b][W0_b][3] at byte offset_host[b];  size_hw_host int32 [B][2];
 *   out_dev uint8 [B][OH][OW][3]. */
int wsc_resize_u8(wsc_ctx *ctx, const uint8_t *images_dev, int B, const int32_t *size_hw_host, const int64_t *offset_host,
                  int OH, int OW, uint8_t *out_dev);

/* F.interpolate(mode='bilinear', align_corners=False) on float32 [C][h][w] -> [C][H][W]
 * (make_cam.py:64-69; also resize_stack 02_cues/utilities.py:20-40 up to the
 * cv2/torch border convention). */
int wsc_bilinear_resize(wsc_ctx *ctx, const float *src_dev, int C, int h, int w, float *dst_dev, int H,
                        int W);

/* ---- cam_to_ir_label (03b_irn/step/cam_to_ir_label.py:26-75) ------------------------------- */

/* labels = np.argmax(np.pad(high_res, ((1,0),(0,0),(0,0)), constant_values=thres), axis=0) and the unary energy of
 * imutils.crf_inference_label: pydensecrf.utils.unary_from_labels(labels, n_labels = K + 1, gt_prob, zero_unsure=False).
 *   highres_dev float32 [B][K][N];  unary_dev float32 [B][K+1][N];  labels_dev int32 [B][N] or NULL. */
int wsc_label_unary_from_cam(wsc_ctx *ctx, const float *highres_dev, int B, int K, int N, float thres, float gt_prob,
                             float *unary_dev, int32_t *labels_dev);
/* conf = keys[fg_pred]; with bg_pred_dev (VOC, :54-57): conf[fg_conf == 0] = 255, conf[bg_conf + fg_conf == 0] = 0;
 * without (ADP / DeepGlobe, keys[0] = -1, :38-40 / :71-73): conf[fg_conf == -1] = 255.
 *   fg_pred_dev / bg_pred_dev int32 [B][N] (CRF arg-max);  keys_host int32 [B][M];  conf_dev uint8 [B][N]. */
int wsc_ir_label_combine(wsc_ctx *ctx, const int32_t *fg_pred_dev, const int32_t *bg_pred_dev, const int32_t *keys_host,
                         int B, int M, int N, uint8_t *conf_dev);

/* ---- HistoSegNet post-processing (03c_hsn/utilities.py:231-397), device resident ---------- */

/* HSN grad_cam after the einsum (utilities.py:262-277), for the NHWC maps of wsc_net_forward_gradcam(relu = 0):
 *   per (image, class) cv2.resize(map, (S, S)) (bilinear) then max(., 0); cams /= max(max_{h,w,c} cams, 1e-7) per
 *   image; cams *= gate[b][c]  (gate = conf_scores * is_pass_threshold).
 *   cams_nhwc_dev float32 [B][h][w][C];  gate_dev float32 [B][C];  out_dev float32 [B][C][S*S]  (class-major). */
int wsc_hsn_gradcam_post(wsc_ctx *ctx, const float *cams_nhwc_dev, int B, int h, int w, int C, int S,
                         const float *gate_dev, float *out_dev, int out_channels, int out_first);
/* (out_channels > C: the maps go to channels [out_first, out_first + C) of a wider [B][out_channels][S*S] stack, e.g. behind
 * the background channel of the VOC branch; out_channels <= 0 means a plain [B][C][S*S] output.) */
/* VOC background channel (03c_hsn/demo.py:145-147): X_bg = sum over the bg model's class maps; channel 0 of the stack =
 * 0.15 * expit(max over the WHOLE batch of X_bg - X_bg)  (SURVEY Q6).
 *   Hbg_dev float32 [B][Cb][N];  y_dev float32 [B][Ctot][N] (channel 0 is written). */
int wsc_hsn_voc_background(wsc_ctx *ctx, const float *Hbg_dev, int B, int Cb, int N, float *y_dev, int Ctot);
/* mass[i] = 1 if map i has a positive entry (dcrf_process keeps the classes whose sum is > 0, utilities.py:425; the maps
 * are >= 0).  maps_dev float32 [n_maps][N];  mass_dev uint32 [n_maps]. */
int wsc_hsn_class_mass(wsc_ctx *ctx, const float *maps_dev, int n_maps, int N, uint32_t *mass_dev);
/* modify_by_htt's background activation (utilities.py:341-347; twins 02_cues/adp_cues.py:279-285,
 * 03b_irn/net/common_cam.py:36-44): 0.75 * expit(4 * (mean_rgb - 240)) smoothed by scipy.ndimage.gaussian_filter(
 * sigma = 2) (truncate 4 sigma, mode 'reflect', axis 0 then axis 1), then cv2.resize (bilinear) to Ho x Wo when the
 * CAM grid differs from the image (:345-347).
 *   rgb_dev uint8 [B][H][W][3];  bg_dev FLOAT64 [B][Ho*Wo] (on tissue the activation is far below the fp32 range and
 *   still decides whether the Background class has mass). */
int wsc_hsn_background(wsc_ctx *ctx, const uint8_t *rgb_dev, int B, int H, int W, int Ho, int Wo, double *bg_dev);
/* ADP background / 'other' channels of the 03b_irn CAM networks (03b_irn/net/common_cam.py:31-92, vgg16_cam.py:51-58),
 * joined with the use_cls CAM channels on the device and summed over the scales of an image (make_cam.py:62-69):
 *   mode 0 (adp_morph, :31-55): out[0] = relu(bg - max_k cam[adipose[k]]);  out[1 + i] = cam[use[i]]
 *   mode 1 (adp_func,  :57-92): out[0] = bg - max_k cam[exc[k]];
 *                               out[1] = max(0.05 (1 - max(out[0], max_i cam[use[i]])), max_k cam[adipose[k]]);  out[2 + i] = cam[use[i]]
 *   cam_dev float32 [B][n_sc][C][hw] (wsc_net_forward_cam);  bg_dev float64 [B][n_sc][hw] (wsc_hsn_background of the ORIGINAL,
 *   un-flipped image of each scale, resized to the CAM grid);  use / adipose / exc: channel indices into the C CAM channels
 *   (the caller composes the X1.7 class filter of common_cam.py:26-29 into them);  out_dev float32 [B][n_use + 1 + mode][hw]. */
int wsc_cam_adp_modify(wsc_ctx *ctx, const float *cam_dev, int B, int n_sc, int C, int hw, const double *bg_dev, int mode,
                       const int32_t *use_host, int n_use, const int32_t *adipose_host, int n_adip, const int32_t *exc_host,
                       int n_exc, float *out_dev);
/* The valid-class stack of one HTT type, modify_by_htt (utilities.py:348-363) and get_cs_gradcam (:367-397) in one pass:
 *   Y[v] = H[src_of_valid[v]] (0 where src_of_valid[v] < 0);  Y[bg_ind] = bg - max_{v in exceptions} Y[v];
 *   functional types (other_ind >= 0): Y[other_ind] = max(0.05 * (1 - max_v Y[v]), max_k H[adipose_src[k]]);
 *   cs[v] = (top1 - top2)(Y) where argmax(Y) == v else 0;  cs[other_ind] = Y[other_ind];
 *   mass[b][v] = 1 if cs[b][v] has a positive entry (dcrf_process keeps the classes whose sum is > 0, :425; every
 *   entry of cs is >= 0).
 *   bg_dev == NULL skips the modify_by_htt step (get_cs_gradcam alone on an already modified stack).
 *   H_dev float32 [B][C_all][N] (wsc_hsn_gradcam_post);  bg_dev float64 [B][N];  cs_dev, y_dev (may be NULL) [B][Cv][N];
 *   mass_dev uint32 [B][Cv];  Cv <= 32, at most 4 exception / adipose classes. */
int wsc_hsn_cs_gradcam(wsc_ctx *ctx, const float *H_dev, int B, int C_all, int N, const double *bg_dev,
                       const int32_t *src_of_valid_host, int Cv, int bg_ind, int other_ind,
                       const int32_t *exception_inds_host, int n_exc, const int32_t *adipose_src_host, int n_adip,
                       float *cs_dev, float *y_dev, uint32_t *mass_dev);
/* unary_from_softmax of a gathered channel list (utilities.py:431): unary[i][p] = -log(clip(maps[chan_off[i] + p],
 * 1e-5, 1)), i < n_chan; chan_off_host = float offsets of the channels inside maps_dev. */
int wsc_hsn_gather_unary(wsc_ctx *ctx, const float *maps_dev, const int64_t *chan_off_host, int n_chan, int N,
                         float *unary_dev);

/* ---- dense CRF (pydensecrf replacement) -------------------------------- */

/* DenseCRF2D(W,H,M) + addPairwiseGaussian(sxy=g_sxy) + addPairwiseBilateral(
 * sxy=bi_sxy, srgb=bi_srgb, rgbim) for a batch of B images of one size
 * (03c_hsn/utilities.py:427-440; misc.imutils.crf_inference_label call sites
 * 03b_irn/step/cam_to_ir_label.py:35-67): builds both permutohedral lattices
 * and the symmetric normalisation vectors, which depend on the image only and
 * are reused by every mean-field iteration.
 *   rgb_dev uint8 [B][H][W][3] (HWC, as np.uint8(images[i])). */
int wsc_crf_create(wsc_ctx *ctx, const uint8_t *rgb_dev, int B, int H, int W, float g_sxy,
                   float bi_sxy, float bi_srgb, wsc_crf **out);
/* Ordering rule: the lattice memory goes back to the BUILD ctx's stream-ordered cache.  wsc_crf_inference may run on
 * another ctx (stream); it records the end of its loop in the wsc_crf, and wsc_crf_destroy makes the build ctx's stream
 * wait for that point before any later work of the build ctx can reuse the memory.  The caller only has to keep the
 * wsc_crf alive until wsc_crf_inference has RETURNED (not until it has finished on the device), and must destroy it
 * before the ctx it was created on. */
void wsc_crf_destroy(wsc_crf *crf);
/* number of occupied lattice vertices per image: v_gauss/v_bilat int32[B] host arrays (may be NULL) */
int wsc_crf_lattice_sizes(wsc_ctx *ctx, const wsc_crf *crf, int32_t *v_gauss_host,
                          int32_t *v_bilat_host);
/* 1 when wsc_crf_inference with M classes sums, blurs and slices the Gaussian (addPairwiseGaussian) lattice inside its
 * update kernel -- per pixel tile, in LDS, no value rows in HBM -- and 0 when the separate blur kernel runs (vertex sets of
 * a pixel tile too large for the kernel's LDS at this M: very narrow g_sxy; the first use of an image size on a ctx; or WSC_OPT_CRF_GAUSS_ON_CHIP = 0).  Same Q bits either
 * way; a diagnostic for tests and benchmarks (03c_hsn/utilities.py:435 is the call this concerns). */
int wsc_crf_gaussian_on_chip(const wsc_crf *crf, int M);

/* d.setUnaryEnergy(U); Q = d.inference(n_iters) with Potts compatibilities
 * g_compat / bi_compat (03c_hsn/utilities.py:431-442):
 *   unary_dev float32 [B][M][H*W]  (= -log p, as unary_from_softmax returns it)
 *   q_dev     float32 [B][M][H*W]  or NULL   (np.array(Q))
 *   argmax_dev int32  [B][H*W]     or NULL   (np.argmax(Q, axis=0))
 * M <= 32. */
int wsc_crf_inference(wsc_ctx *ctx, wsc_crf *crf, const float *unary_dev, int M, float g_compat,
                      float bi_compat, int n_iters, float *q_dev, int32_t *argmax_dev);
/* wsc_crf_inference on pixel-major unaries: unary_pm_dev float32 [B][H*W][Mp], Mp = 4 * ceil(M/4), padding classes 0
 * (wsc_cam_unary_pm).  The buffer is read in place during the whole loop and never written. */
int wsc_crf_inference_pm(wsc_ctx *ctx, wsc_crf *crf, const float *unary_pm_dev, int M, float g_compat,
                         float bi_compat, int n_iters, float *q_dev, int32_t *argmax_dev);

/* ---- ragged batch ----------------------------------------------------------------------------------------------------
 * The reference runs its CRF one image at a time, every image with its own size and class count
 * (03b_irn/step/cam_to_ir_label.py:25-58: crf_inference_label per image, K_b + 1 labels; 03c_hsn/utilities.py:420-445:
 * DenseCRF2D(W, H, len(pass_inds[i])) per image).  A wsc_crf_v takes such a list as it comes:
 *   rgb_dev[b]   device pointer of image b, uint8 [H_b][W_b][3];  hw_host[2b], hw_host[2b+1] = H_b, W_b
 * Images of equal size share a lattice build and a mean-field loop; the groups are issued back to back on the ctx's stream
 * (no host synchronisation between them).  wsc_crf_v_inference:
 *   unary_dev[b] device pointer of image b's unaries, float32 [M_b][H_b*W_b] (class-major, as wsc_crf_inference takes them)
 *   M_host[b]    its class count (1..32);  q_dev[b] (float32 [M_b][H_b*W_b]) and / or argmax_dev[b] (int32 [H_b*W_b]);
 *                either pointer ARRAY may be NULL, single entries too
 * Inside a group the loop runs at the largest M_b; a smaller image's missing classes enter with probability zero, which
 * leaves every bit of its result what a call with its own M_b gives (tests/test_gpu_crf.py::test_crf_ragged_batch). */
int wsc_crf_v_create(wsc_ctx *ctx, const uint8_t *const *rgb_dev, const int32_t *hw_host /*[B][2]*/, int B, float g_sxy,
                     float bi_sxy, float bi_srgb, wsc_crf_v **out);
void wsc_crf_v_destroy(wsc_crf_v *crf);
int wsc_crf_v_num_groups(const wsc_crf_v *crf); /* lattice builds / loops the batch needs (distinct image sizes) */
int wsc_crf_v_inference(wsc_ctx *ctx, wsc_crf_v *crf, const float *const *unary_dev, const int32_t *M_host, float g_compat,
                        float bi_compat, int n_iters, float *const *q_dev, int32_t *const *argmax_dev);

#ifdef __cplusplus
}
#endif
#endif /* WSSCAM_H */
