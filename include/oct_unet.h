/*
 * oct_unet.h -- C ABI of the MI355X-native OCT U-Net engine (liboct_unet_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path of
 * NIH-NEI/oct-image-segmentation-models: the U-Net forward/backward + softmax/
 * Dice head that the reference delegates to tf.keras (Model.fit / Model.predict).
 * Plain C, plain pointers and sizes, no torch types.  Every device buffer is
 * owned by the CALLER (any allocator: torch, hipMalloc); the handle owns only
 * host-side plans.  No entry point allocates device memory, so every call after
 * oct_unet_create() is hipGraph-capturable.  All launches are asynchronous on
 * the caller-supplied stream.  One handle per GPU rank, driven by one host
 * thread.
 *
 * Reference interface each entry point replaces (paths relative to
 * /root/reference/oct_image_segmentation_models/):
 *
 *   oct_unet_create / _layer_info      UNet.__init__/build_model      models/unet.py:61-153
 *   params/state buffer layout         Model.get_weights/set_weights  training.py:319-342 (checkpoint, early stop)
 *   oct_unet_forward (training=0)      loaded_model.predict(...)      evaluation/evaluation.py:129-135,
 *                                                                     prediction/prediction.py:75-81
 *   oct_unet_forward (training=1)      Model.fit train step (fwd)     training/training.py:401-407
 *     x_is_u8 preprocessing            get_preprocess_input_fn x/255  models/unet.py:87-91
 *     io.argmax                        perform_argmax                 common/utils.py:80-112
 *   oct_unet_loss_dice                 dice_loss_micro/_macro         common/custom_losses.py:47-81
 *                                      dice_coef_micro/_macro         common/custom_metrics.py:18-77
 *   oct_unet_backward                  Keras autodiff of the above    training/training.py:262-266,401-407
 *   oct_adam_step / oct_sgd_step       optimizer.apply_gradients      training/training.py:190-193
 *   gradient buffer (caller-owned)     MirroredStrategy all-reduce    training/training.py:185-188,243
 *   oct_unet_graph_capture/_launch     (none: replaces per-call Keras dispatch overhead, evaluation.py:108-135)
 */
#ifndef OCT_UNET_H
#define OCT_UNET_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oct_unet oct_unet;   /* opaque */
typedef void* oct_stream_t;         /* a hipStream_t (NULL = default stream) */

typedef struct oct_unet_cfg {
    int in_ch;            /* input_channels                       (unet.py:65)          */
    int n_cls;            /* num_classes, 2..8                    (training.py:176)     */
    int H, W;             /* image_height, image_width; multiples of 2^pool_layers     */
    int max_batch;        /* largest per-rank batch any call will pass                  */
    int start_neurons;    /* default 8; multiple of 4 in 4..32    (unet.py:69)          */
    int pool_layers;      /* default 4                            (unet.py:70)          */
    int conv_layers;      /* default 2                            (unet.py:71)          */
    int enc_k;            /* 3  (enc_kernel (3,3))                (unet.py:72)          */
    int dec_k;            /* 2  (dec_kernel (2,2))                (unet.py:73)          */
    int dtype;            /* 0 = f32; 1 = bf16 activations / activation gradients (storage AND, */
                          /*     with mfma_mode 1, MFMA operands), f32 accumulation, f32 BN     */
                          /*     statistics, f32 parameters + gradients (BASELINE configs[2])   */
    int training;         /* 1: workspace also holds saved activations + gradients      */
    float bn_eps;         /* 1e-3  keras BatchNormalization default                     */
    float bn_momentum;    /* 0.99                                                       */
    float dropout_rate;   /* 0.5                                  (unet.py:130)         */
    int bn_unbiased_moving_var; /* 1: moving_var fed with Bessel-corrected batch var (TF fused BN) */
    unsigned long long seed;    /* dropout stream seed (differs per DP rank)            */
} oct_unet_cfg;

/* One Conv2D(+BN) node in Keras creation order; offsets are in floats. */
typedef struct oct_layer_info {
    char name[32];
    int kh, kw, cin, cout, has_bn;
    int out_h, out_w;
    size_t kernel_off, bias_off, gamma_off, beta_off;     /* into params / grads buffers */
    size_t moving_mean_off, moving_var_off;               /* into the state buffer        */
} oct_layer_info;

typedef struct oct_unet_io {
    float* probs;                 /* (B,H,W,n_cls) f32 softmax output, or NULL           */
    unsigned char* argmax;        /* (B,H,W) u8 class map, or NULL                       */
    const unsigned char* labels;  /* (B,H,W) u8 sparse labels, or NULL; enables Dice sums */
} oct_unet_io;

/* ---- sizing (host only, no GPU needed) ---- */
void   oct_unet_cfg_default(oct_unet_cfg* cfg);
int    oct_unet_cfg_check(const oct_unet_cfg* cfg);              /* 0 ok, <0 + oct_last_error() */
size_t oct_unet_param_count(const oct_unet_cfg* cfg);            /* trainable floats (487403 default, C=3) */
size_t oct_unet_state_count(const oct_unet_cfg* cfg);            /* BN moving stats (1712 default)         */
size_t oct_unet_workspace_bytes(const oct_unet_cfg* cfg);        /* activations, gradients, scratch        */
int    oct_unet_layer_count(const oct_unet_cfg* cfg);
int    oct_unet_layer_info(const oct_unet_cfg* cfg, int index, oct_layer_info* out);

/* ---- lifetime ---- */
/* params/grads: param_count floats; state: state_count floats; grads may be NULL when cfg.training==0. */
int  oct_unet_create(const oct_unet_cfg* cfg, float* params_dev, float* grads_dev, float* state_dev,
                     void* workspace_dev, size_t workspace_bytes, oct_unet** out);
void oct_unet_destroy(oct_unet* h);

/* ---- hot path ---- */
/* x: (B,H,W,in_ch) u8 (x_is_u8: /255 table applied on load) or f32 already in [0,1].
 * training=1: batch-statistic BN (+moving update), dropout, activations saved for backward. */
int oct_unet_forward(oct_unet* h, const void* x_dev, int x_is_u8, int B, int training,
                     const oct_unet_io* io, oct_stream_t stream);
/* After a forward with io.labels: out4_dev = {dice_loss_macro, dice_loss_micro,
 * dice_coef_macro, dice_coef_micro} (device floats). */
int oct_unet_loss_dice(oct_unet* h, float smooth, float* out4_dev, oct_stream_t stream);
/* focal_dice_loss (reference common/custom_losses.py:98-178, `SparseCategoricalFocalDiceLoss`):
 *   L = w * mean_px[ cw[y] * (1 - p_y)^gamma * (-log p_y) ] + (1 - w) * dice_loss_{macro|micro}
 * oct_unet_set_focal_dice selects it for the following forward / loss / backward calls (w = 0 restores the plain Dice
 * losses; class_weight_dev = n_cls device floats or NULL, caller-owned and kept alive).  oct_unet_loss_focal_dice is
 * oct_unet_loss_dice with 8 outputs: out8_dev = out4 + {focal term, w*focal + (1-w)*dice_macro,
 * w*focal + (1-w)*dice_micro, 0}; oct_unet_backward then differentiates the combination chosen by its `macro` flag. */
/* Unverifiable detail, isolated as a switch (like cfg.bn_unbiased_moving_var): third-party focal-loss==0.0.7 clips the
 * probabilities to [1e-7, 1-1e-7] for the logarithm; whether its (1 - p_y)^gamma modulation also sees the clipped value
 * cannot be checked here (package absent).  oct_set_option("focal_clip_modulation", 0) [default]: only the logarithm is
 * clipped; 1: both.  The two differ only where p_y is outside the clip range, by < 1e-7 relative in the loss and -- after
 * the softmax Jacobian, which multiplies by p_y -- by < 1e-6 of the gradient scale (tests/test_gpu_parity.py::
 * test_focal_clip_modulation_switch pins both against the oracle on a saturated head). */
int oct_unet_set_focal_dice(oct_unet* h, float focal_loss_weight, float gamma, const float* class_weight_dev);
int oct_unet_loss_focal_dice(oct_unet* h, float smooth, float* out8_dev, oct_stream_t stream);
/* After training forward + loss_dice: fills the grads buffer with d(loss_scale*loss)/dparams. */
int oct_unet_backward(oct_unet* h, const unsigned char* labels_dev, int macro, float loss_scale,
                      oct_stream_t stream);

/* ---- data-parallel overlap (reference: tf.distribute.MirroredStrategy, training/training.py:185-188,243) ----
 * The gradient buffer has the parameter layout (Keras creation order: encoder, bottleneck, decoder, head) and backward
 * runs head -> decoder -> bottleneck -> encoder.  With a tail event set, oct_unet_backward sums the partial slabs of
 * every layer from the first bottleneck conv on as soon as that conv's gradients are queued and records the event behind
 * that sum (on the handle's internal side stream when "dw_side_stream" is on -- `stream` does not wait for it -- else on
 * `stream`): floats [oct_unet_grad_tail_offset(cfg), param_count) of grads are final from then on, so the launcher can
 * all-reduce that TAIL segment (bottleneck + decoder + head: 97 % of the floats) on a side stream (after
 * hipStreamWaitEvent) while the encoder backward still runs, and the remaining ENCODER segment [0, offset) after
 * backward.  hip_event: a hipEvent_t owned by the caller, NULL disables.  If oct_unet_backward returns an error the
 * event may not have been recorded and the handle's internal side stream has been joined to `stream` before returning. */
int    oct_unet_set_tail_event(oct_unet* h, void* hip_event);
size_t oct_unet_grad_tail_offset(const oct_unet_cfg* cfg);

/* ---- optimizers on flat buffers (Keras formulations) ---- */
int oct_adam_step(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, size_t n,
                  float lr, float beta1, float beta2, float eps, long step /*1-based*/, oct_stream_t stream);
int oct_sgd_step(float* params_dev, const float* grads_dev, float* momentum_buf_dev /*or NULL*/, size_t n,
                 float lr, float momentum, oct_stream_t stream);

/* ---- dropout stream control (parity tests replay the mask) ---- */
int oct_unet_set_dropout_step(oct_unet* h, unsigned long long step);
/* keep-mask (B, H/2^P, W/2^P, start_neurons*2^P) u8 {0,1} that a training forward at the current step uses */
int oct_unet_dropout_mask(oct_unet* h, int B, unsigned char* mask_dev, oct_stream_t stream);

/* ---- inference hipGraph: capture one forward (training=0) with fixed buffers, then replay ---- */
int oct_unet_graph_capture(oct_unet* h, const void* x_dev, int x_is_u8, int B, const oct_unet_io* io,
                           oct_stream_t stream);
int oct_unet_graph_launch(oct_unet* h, oct_stream_t stream);

/* ---- per-launch profiler: HIP events recorded around every kernel launch, on the launch stream ----
 * begin() arms it; every forward/backward call of this handle made on the calling thread is then recorded;
 * end() synchronises the device and returns one entry per (kernel instantiation, layer) with the summed
 * duration and the ALGORITHMIC flops/bytes of those launches (cost model). */
typedef struct oct_profile_entry {
    char kernel[80];
    char layer[32];
    int launches;
    double total_ms, flops, bytes;
} oct_profile_entry;
int oct_unet_profile_begin(oct_unet* h);
int oct_unet_profile_end(oct_unet* h, oct_profile_entry* out, int max_entries, int* n_out);

/* ---- post-step on device: (B,H,W) u8 class maps -> (B, n_cls-1, H, W) u8 boundary maps, exactly
 * convert_predictions_to_maps_semantic(one_hot(argmax)) of the reference (common/utils.py:115-168) ---- */
int oct_boundary_maps(const unsigned char* labels_dev, int B, int H, int W, int n_cls, int bg_ilm, int bg_csi,
                      unsigned char* maps_dev, oct_stream_t stream);

/* ---- options ----
 * oct_set_option edits the PROCESS-WIDE DEFAULTS; a handle copies them when it is created (oct_unet_create) and every
 * launch of that handle reads its own copy: changing an option never affects a live handle, and two handles created
 * under different settings coexist in one process.  oct_unet_workspace_bytes uses the defaults current at the call, so
 * size and create a handle under the same settings.  oct_unet_get_option reads a handle's copy.
 * Two select ARITHMETIC (documented alternatives, each tested against the oracle):
 *   "mfma_mode" (default 1): 1 = convolutions on the bf16 matrix pipe -- in fp32 mode (cfg.dtype 0) every fp32 operand is
 *   split exactly into three bf16 terms and a product is six bf16 MFMAs accumulated in fp32 (fp32-equivalent results,
 *   DESIGN.md section 4); in bf16 mode (cfg.dtype 1) activations and weights are rounded once and multiplied directly.
 *   0 = the fp32-pipe kernels (v_mfma_f32_*_f32), fp32 math on whatever the storage type is.
 *   "focal_clip_modulation" (default 0): see oct_unet_set_focal_dice.
 * The rest are tuning knobs (results do not depend on them beyond fp32 rounding, only which kernel variant runs):
 *   "bx_min_blocks" (256): a wide bf16-pipe launch takes the taller pixel tile only if that still yields this many blocks.
 *   "bx_two_blocks" (1): wide bf16-pipe launches (3x3 and 2x2-over-upsample) run as 4-wave blocks with ONE input image in LDS,
 *   two blocks per CU -- each block's prologue, barriers and epilogue hide behind the other's MFMAs; 0 = the 8-wave blocks
 *   with a double-buffered image, one per CU (then "bx_min_blocks" / "bx_waves" pick their tile height and wave count).
 *   "fuse_dw_thin" (1): 3x3 layers with 8 output channels (the full-resolution convs): the backward-data launches also
 *   reduce the layer's backward-weights (conv_bt_k FDW); 0 = a separate backward-weights kernel.
 *   "bx_waves" (8 | 4): waves per block of conv_bx_k where the tile has >= 8 rows (two / one per SIMD).
 *   "dwbx_blocks" (256): grid target of the wide bf16-pipe backward-weights kernel.
 *   "bt_blocks_per_cu" (0 = as many as the LDS images allow): persistent blocks of the thin bf16-pipe kernel.
 *   "bt_m2" (1): thin bf16-pipe launches with exactly 8 output channels use the two-pixel form (16 MFMA rows = 2 adjacent
 *   pixels x 8 channels); read when the handle is created (the weights are prepared in that form).  0 = one pixel per column.
 *   "fuse_first_apply" (1): the first conv's BN-backward transform is applied inside its backward-weights kernel (the
 *   only consumer of that dz) instead of by a bn_bwd_apply pass; bit-identical gradients; oct_unet_debug_activation(0, 1)
 *   then returns the masked gradient g' of block 0, not dz.  0 = separate pass.
 *   "fuse_bn_apply" (1): every other block's BN-backward transform dz = ga g' + gb z + gd is applied by the consumers of
 *   dz -- its backward-weights kernel and its backward-data launches -- while they stage g' and z, wherever all of them
 *   can (the bf16-pipe conv kernels; every MFMA backward-weights kernel); the bn_bwd_apply pass (3 tensor passes per
 *   block) disappears.  Same two fmas per element as the stand-alone pass: bit-identical gradients.  The block's g buffer
 *   then keeps g' (oct_unet_debug_layer_fused tells which blocks).  0 = separate pass everywhere.
 *   "fuse_bn_apply16" (1): ... including the thin backward-data launches with 16 K and 16 output channels (their transform
 *   coefficients live in LDS: the registers are taken by two raw tiles, the mask input and the weights); 0 = those blocks
 *   keep the separate pass (same step time at B = 32, 256x512; 0.2 GB more HBM traffic per step).
 *   "fuse_bn_finalize" (0): 1 = the BN records of the thin layers (<= 32 channels) are written by the LAST block of the
 *   launch that produces the partial rows (arrival counter; write-through rows; csrc/kernels_fin.hpp) instead of by a
 *   bn_*_finalize launch.  Same results to fp32 rounding; measured 0.5-1 % slower per step than the launches it removes,
 *   hence off by default.
 *   "dwbt_f32_all" (0): 1 = fp32 mode takes conv_dwbt_k for every thin backward-weights shape (default: where it wins).
 *   "dw_side_stream" (1): backward-weights kernels and the per-step weight preparation run on a low-priority stream
 *   owned by the handle, beside the backward-data chain.  0 = everything on the caller's stream.
 *   "fork_on_launch" (1): the event such a backward-weights kernel waits for is the completion signal of the preceding
 *   launch itself (hipExtLaunchKernelGGL stopEvent) rather than a marker recorded behind it.  "dw_fork_group" (1): blocks
 *   whose backward-weights launches share one fork (2-4: fewer waits on the caller's stream, measured slower overall).
 *   "event_sysfence" (0; read at oct_unet_create): 1 = the handle's internal fork/join events carry a system-scope fence.
 *   Experiment switches, not for production use: "dwbx_enable" (1; 0 routes the wide backward-weights layers to the
 *   fp32-pipe kernel) and "timing_skip" (0; bit 0 / 1 SKIP the forward / backward BN finalize launches after the second
 *   step -- the results are then WRONG; used once to measure what those launches cost).
 *   "igemm_persistent_min_tiles" (default 2048): number of pixel tiles from which thin single-chunk convs use the
 *   persistent software-pipelined kernel instead of the one-tile-per-block kernel.
 *   "thin8_min_tiles" (default 2048): number of pixel tiles from which 8-output-channel convs run on the VALU
 *   thin-layer kernel instead of the (half-padded) 16x16 MFMA kernel.
 *   "pair8_min_tiles" (default 2048): number of pixel tiles from which 3x3 convs with 8 output channels run on the
 *   pixel-pair MFMA kernel (takes precedence over the VALU kernel).
 *   "dw32_blocks" (512) / "dw16_blocks" (768): target block count of a backward-weights launch (wide / thin kernels; one
 *   partial slab per block -- read when the handle is created, later launches never exceed the slabs allocated then).
 *   "igemm_persistent_blocks" (1280): grid of the persistent igemm kernel.  "igemm_min_blocks" (512): a layer takes the
 *   taller pixel tile only if that still yields this many blocks.
 *   "dwpair8_enable" (default 1): backward-weights of 3x3 layers with 8 output channels on the pixel-pair kernel
 *   (0 = the padded 16-column kernel).
 *   "pair8_geometry" (NWY*100 + NWX*10 + RPW in {221, 111}, default 221): waves per block (rows x columns)
 *   and 4-row groups per wave of that kernel; its tile is (4*RPW*NWY) x (32*NWX) pixels. */
int oct_set_option(const char* name, int value);
int oct_get_option(const char* name, int* value);
int oct_unet_get_option(const oct_unet* h, const char* name, int* value);
/* edit a live handle's copy (between steps).  Everything but "bt_m2" (it shapes the prepared weights); launches never use
 * more partial-slab rows than were allocated at creation, whatever the block-count knobs say later. */
int oct_unet_set_option(oct_unet* h, const char* name, int value);

/* ---- introspection for tests: device pointer of a layer's saved pre-BN output / gradient ---- */
/* which: 0 = z, 1 = g (f32 or bf16 per cfg.dtype; max_batch x out_h x out_w x cout);
 *        2 = the layer's BN record, f32 [9][cout]: a, b (y = relu(a*z + b)), batch/moving mean, rstd, c1, c2 and the
 *            BN-backward transform dz = ga g' + gb z + gd as rows ga, gb, gd */
const void* oct_unet_debug_activation(oct_unet* h, int layer, int which);
/* 1 if, in the last oct_unet_backward, the layer's BN-backward transform was applied on load by its consumers (its g
 * buffer still holds the masked gradient g'), 0 if the stand-alone pass turned g into dz in place, -1 on a bad index */
int oct_unet_debug_layer_fused(const oct_unet* h, int layer);

const char* oct_last_error(void);
/* "oct_unet_hip <ver> (gfx950) src:<12 hex digits>": the digits are the SHA-256 of the sources the library was built from
 * (csrc build.sh); the Python binding refuses a library whose stamp differs from the sources beside it */
const char* oct_version(void);

#ifdef __cplusplus
}
#endif
#endif /* OCT_UNET_H */
