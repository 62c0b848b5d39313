/*
 * pcc_geo.h -- C ABI of libpcc_geo_hip.so, the MI355X (gfx950) implementation of the
 * 64^3-voxel-block encode+decode hot path of mauriceqch/pcc_geo_cnn_v2.
 *
 * The reference has NO native/FFI interface for this path: its operators are TensorFlow-1.15 /
 * tensorflow-compression-1.3 ops invoked from Python.  Each entry point below
 * therefore names the reference call site (file:line under /root/reference) whose third-party op it
 * replaces; INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no torch types; every function returns 0 on success, <0 on error
 *     (pcc_last_error() returns the message of the calling thread's last failure);
 *   - device buffers are owned by the caller (PyTorch-ROCm tensors' data_ptr()); the library
 *     allocates nothing persistent except the context it frees in pcc_ctx_destroy;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are stream-ordered,
 *     asynchronous, and not re-entrant on one context;
 *   - activations: NDHWC float32, (D,H,W) = (x,y,z) as in src/model_types.py:108-114;
 *     forward kernels (kd,kh,kw,Cin,Cout), transposed kernels (kd,kh,kw,Cout,Cin)  [Keras layouts];
 *   - all convolutions are TF padding='same' (asymmetric, extra element on the high side).
 */
#ifndef PCC_GEO_H
#define PCC_GEO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCC_ABI_VERSION 4

/* ---- errors / context ------------------------------------------------------------------ */
#define PCC_OK 0
#define PCC_ERR_ARG (-1)     /* bad argument / unsupported shape                                   */
#define PCC_ERR_HIP (-2)     /* a HIP runtime call failed                                          */
#define PCC_ERR_NOGPU (-3)   /* no gfx950 device visible                                           */
#define PCC_ERR_SPACE (-4)   /* output buffer too small                                            */
#define PCC_ERR_CORRUPT (-5) /* range decoder ran past a corrupt stream                            */

typedef struct pcc_ctx pcc_ctx;

int pcc_abi_version(void);
const char* pcc_last_error(void);
/* Replaces tf.Session creation (src/compress_octree.py:84-92).  One context per (process, GPU). */
int pcc_ctx_create(int device, pcc_ctx** out);
int pcc_ctx_destroy(pcc_ctx* ctx);
/* Number of compute units of the context's device (256 on MI355X); <0 on error. */
int pcc_ctx_num_cu(pcc_ctx* ctx);

/* ---- codec numerics: which kernel family computes a layer ---------------------------------
 * Encoder and decoder must produce the SAME bits for sigma-hat (it selects the entropy coder's rows: a single flipped scale index
 * desynchronises the range decoder -- the reason for the reference's --debug retries, src/decompress_octree.py:84-101).  Every
 * kernel here is bit-deterministic, but two kernel families (exact-fp32 MFMA, split-bf16 MFMA, Winograd, direct ...) give different
 * round-off.  What selects the family is therefore STATE OF THE CONTEXT, read ONCE from the environment in pcc_ctx_create (the
 * PCC_* variables named below) and changeable only through pcc_ctx_set_numerics -- never per call -- and it is recorded beside
 * every stream the CLIs write (gzip header comment, model_syntax.write_tagged_gzip) together with PCC_KERNEL_FAMILY, which is bumped whenever
 * a default kernel's summation order changes.  The decoder refuses a stream written under another tag. */
#define PCC_KERNEL_FAMILY 6
#define PCC_NUM_NO_SPLIT 0x1         /* PCC_NO_SPLIT=1: every split-bf16 kernel off (exact-fp32 MFMA everywhere)              */
#define PCC_NUM_NO_SPLIT_DIRECT 0x2  /* PCC_NO_SPLIT_DIRECT=1: the direct 32- / 64-channel split kernels off                  */
#define PCC_NUM_NO_SPLIT_TR2 0x4     /* PCC_NO_SPLIT_TR2=1: the stride-2 transposed split kernels off                         */
#define PCC_NUM_NO_WINOGRAD 0x8      /* PCC_NO_WINOGRAD=1: direct kernels instead of Winograd                                 */
#define PCC_NUM_NO_WINOGRAD32 0x10   /* PCC_NO_WINOGRAD32=1                                                                   */
#define PCC_NUM_NO_WINOGRAD64 0x20   /* PCC_NO_WINOGRAD64=1                                                                   */
#define PCC_NUM_WINO_PER_GROUP 0x40  /* PCC_WINO_PER_GROUP=1: one Winograd launch per cin group                               */
#define PCC_NUM_NO_TR2M 0x80         /* PCC_NO_TR2M=1: tiled stride-2 transposed kernels instead of the z march               */
#define PCC_NUM_TR2M 0x100           /* PCC_TR2M=1: the z march wherever it is eligible                                       */
#define PCC_NUM_TR2_OLD 0x200        /* PCC_TR2_OLD=1: per-tile conv_tr2_kernel                                               */
#define PCC_NUM_SPLIT_MFMA16 0x400   /* PCC_SPLIT_MFMA=16: 16x16x32 formulation of the direct split kernel                    */
#define PCC_NUM_SPLIT_MFMA32 0x800   /* PCC_SPLIT_MFMA=32: 32x32x16 formulation                                               */
#define PCC_NUM_SPLIT_TILE8 0x1000   /* PCC_SPLIT_TILE=8                                                                      */
#define PCC_NUM_P16 0x2000           /* PCC_P16=1: one 8-wave workgroup per CU in the direct 16 -> 16 kernel                  */
#define PCC_NUM_NO_F16S 0x4000       /* PCC_NO_F16S=1: the Winograd layers keep three bf16 pieces / exact fp32 (no two-piece fp16 split) */
#define PCC_NUM_COUT1_T16 0x8000     /* PCC_COUT1_T16=1: 16 x 16 columns in the 16 -> 1 last layer (same bits as 32 x 32: tested)  */
/* family = PCC_KERNEL_FAMILY of this build, switches = OR of PCC_NUM_* in effect on this context */
int pcc_ctx_get_numerics(pcc_ctx* ctx, uint32_t* family, uint32_t* switches);
/* Replace the switches (tests, A/B runs).  Not to be called between an encode and the decode of its stream. */
int pcc_ctx_set_numerics(pcc_ctx* ctx, uint32_t switches);

/* ---- 3-D convolution / transposed convolution -------------------------------------------
 * Replaces the TF ops behind keras Conv3D / Conv3DTranspose (+BiasAdd, Relu, AddV2) at
 * src/model_transforms.py:45-47,56-58,67-69,78-80,93,107,121,135,144-146,155-157 and the residual
 * add of ResidualLayer.call (src/model_transforms.py:30-36).
 * out = [clip01]( relu?(conv(in) + bias?) + residual? )                                        */
#define PCC_CONV_BIAS 1
#define PCC_CONV_RELU 2
#define PCC_CONV_ADD 4     /* add `residual` AFTER the activation (ResidualLayer 'add' mode)      */
#define PCC_CONV_CLIP01 8  /* np.clip(x_hat,0,1) fused (src/model_types.py:202), encoder flavour  */
#define PCC_CONV_F16 16    /* fp16 matrix instructions (operands rounded RTN, fp32 accumulate and storage) on the
                              direct MFMA kernels; BASELINE.json configs[4].  Not the default: the reference is fp32 */

/* fp16 STORAGE inside the fp16 mode (mid-network tensors of the c3 / c3p blocks; used by pcc_network_forward with
 * PCC_CONV_F16, accepted here for callers that chain layers themselves).  Buffers are passed through the same pointers.   */
#define PCC_CONV_IN16 32   /* `in` (and `residual`, if any: see RES16) are fp16 NDHWC: k3 stride-1 layers with Cin = Cout in
                              {16, 32, 64} and H, W multiples of 16 (conv_f16.hip, v_mfma_f32_16x16x32_f16), and the 16 -> 1
                              k3 stride-1 transposed layer.  The 64-channel layers keep fp16 partial sums of the first input
                              half in a scratch tensor owned by the context (allocated on first use, N*D*H*W*128 bytes):
                              calls that use it must be ordered on one stream per context                                 */
#define PCC_CONV_OUT16 64  /* `out` is fp16 NDHWC: the IN16 layers and the k3 stride-2 transposed layers                  */
#define PCC_CONV_RES16 128 /* `residual` is fp16 (always together with IN16)                                              */

#define PCC_IMPL_AUTO 0    /* MFMA implicit-GEMM when the shape is covered, else generic          */
#define PCC_IMPL_GENERIC 1 /* direct convolution, any shape (reference-order fp32 FMA chain)      */
#define PCC_IMPL_MFMA 2    /* force the direct MFMA implicit-GEMM path; PCC_ERR_ARG if not covered */
#define PCC_IMPL_WINOGRAD 3 /* force Winograd F(2x2,3x3)+z on MFMA (Cin = Cout in {16,32,64}, k3 s1, W,H % 16 == 0); AUTO
                              picks it when eligible (env PCC_NO_WINOGRAD=1 disables); 16-channel layers take the split-bf16
                              kernel (three bf16 pieces per fp32 operand, fp32-equivalent; env PCC_NO_SPLIT=1: exact-fp32 MFMA)  */
#define PCC_IMPL_SPLIT 4    /* force the direct k3 stride-1 kernel with split-bf16 operands (Cin = Cout in {32,64}, W % 16 == 0);
                              AUTO picks it for launches that fill the CUs (env PCC_NO_SPLIT_DIRECT=1 / PCC_NO_SPLIT=1 disable)    */

typedef struct {
    int32_t N, D, H, W;    /* input batch and spatial size                                       */
    int32_t Cin, Cout;
    int32_t k;             /* cubic kernel size: 3, 5 or 9 on the fast path, any odd k generic    */
    int32_t stride;        /* 1 or 2                                                              */
    int32_t transposed;    /* 0 = Conv3D, 1 = Conv3DTranspose                                     */
    int32_t flags;         /* PCC_CONV_*                                                          */
    int32_t impl;          /* PCC_IMPL_*                                                          */
    int32_t out_cstride;   /* channel stride of `out` (>= Cout); 0 means Cout.  With out_coffset  */
    int32_t out_coffset;   /* it implements ResidualLayer 'concat' mode (model_transforms.py:38). */
} pcc_conv_desc;

/* Output spatial size for a descriptor (ceil(n/s) forward, n*s transposed). */
int pcc_conv_out_dims(const pcc_conv_desc* d, int32_t* OD, int32_t* OH, int32_t* OW);
/* Returns 1 if the MFMA path covers the descriptor, 0 otherwise. */
int pcc_conv_mfma_supported(const pcc_conv_desc* d);
/* Size in floats of the MFMA-fragment-ordered weight image (0 if the MFMA path does not apply). */
size_t pcc_conv_packed_floats(const pcc_conv_desc* d);
/* HOST-side repack of a Keras-layout kernel into MFMA fragment order (done once at model load,
 * replaces saver.restore's variable placement, src/compress_octree.py:90-92). */
int pcc_conv_pack_weights(const pcc_conv_desc* d, const float* w_keras_host, float* packed_host);
/* Name of the kernel family pcc_conv3d takes for this descriptor on this context (w_packed given), e.g. "conv16_wino_f16s (...)":
 * written NUL-terminated into buf (truncated to cap).  Diagnostic: bench.py prints it beside every layer's time.             */
int pcc_conv_kernel_family(pcc_ctx* ctx, const pcc_conv_desc* d, char* buf, int32_t cap);
/* `w` (device, Keras layout) is used by the generic path, `w_packed` (device, may be NULL) by the
 * MFMA path; `bias`/`residual` may be NULL when the matching flag is clear.  `residual` has the
 * shape of `out` with channel stride Cout. */
int pcc_conv3d(pcc_ctx* ctx, const pcc_conv_desc* d, const float* in, const float* w,
               const float* w_packed, const float* bias, const float* residual, float* out,
               void* stream);

/* ---- batched graph: whole transforms and whole graph phases in ONE call ------------------------------------------
 * The layer stacks of src/model_transforms.py:41-158 (one id per member of its TransformType enum, :161-169) run as a
 * sequence of pcc_conv3d launches enqueued by the library on `stream`; the reference runs them as one Keras
 * `layer(tensor)` call inside sess.run (src/model_types.py:289-293,379-388,405-408).  ResidualLayer mode 'add' only
 * (the mode every config uses); 'concat' stays available through pcc_conv3d's out_cstride/out_coffset.               */
#define PCC_NET_ANALYSIS_V1 0               /* model_transforms.py:41-48   */
#define PCC_NET_SYNTHESIS_V1 1              /* :51-59                      */
#define PCC_NET_ANALYSIS_V2 2               /* :84-95                      */
#define PCC_NET_SYNTHESIS_V2 3              /* :98-109                     */
#define PCC_NET_ANALYSIS_PROGRESSIVE_V2 4   /* :112-123                    */
#define PCC_NET_SYNTHESIS_PROGRESSIVE_V2 5  /* :126-137                    */
#define PCC_NET_HYPER_ANALYSIS 6            /* :140-147                    */
#define PCC_NET_HYPER_SYNTHESIS 7           /* :150-158                    */

/* Number of conv layers of a transform (conv_layers() order = Keras construction order), or <0. */
int32_t pcc_network_num_layers(int32_t transform, int32_t filters);
/* Geometry of layer `layer` (Cin, Cout, k, stride, transposed, flags; N/D/H/W left 0) and its role in a residual block:
 * 0 plain, 1 its output is the block's `tensor1`, 2 `tensor1` is added after its activation (model_transforms.py:30-36). */
int pcc_network_layer(int32_t transform, int32_t filters, int32_t layer, pcc_conv_desc* d, int32_t* residual_role);
/* Weight upload -- replaces saver.restore's variable placement (src/compress_octree.py:90-92).  The blob holds, per layer,
 * the Keras-layout kernel, its MFMA/Winograd fragment image and the bias.  It lives in CALLER-owned device memory of
 * pcc_weights_blob_floats() floats; pcc_weights_pack builds the same image on the host.  kernels[i] / biases[i]: host
 * pointers in conv_layers() order (biases[i] ignored for layers without bias).  Synchronous (model-load time).          */
size_t pcc_weights_blob_floats(int32_t transform, int32_t filters);
int pcc_weights_pack(int32_t transform, int32_t filters, const float* const* kernels, const float* const* biases,
                     float* blob_host);
int pcc_weights_upload(pcc_ctx* ctx, int32_t transform, int32_t filters, const float* const* kernels,
                       const float* const* biases, float* blob_device, void* stream);
/* Activations workspace (caller-owned device memory) for an input of N x D x H x W voxels, and the output size.          */
size_t pcc_network_workspace_bytes(int32_t transform, int32_t filters, int32_t N, int32_t D, int32_t H, int32_t W);
int pcc_network_out_dims(int32_t transform, int32_t filters, int32_t D, int32_t H, int32_t W, int32_t* OD, int32_t* OH,
                         int32_t* OW, int32_t* OC);
/* y = transform(x).  x: (N,D,H,W,Cin) with Cin = 1 for the analysis transforms, `filters` otherwise; y: NDHWC of
 * pcc_network_out_dims.  layer_flags: 0 or PCC_CONV_F16 (every layer); final_flags: 0 or PCC_CONV_CLIP01 (last layer).
 * Results are bit-identical to the same layers issued one by one through pcc_conv3d (with layer_flags = PCC_CONV_F16 the
 * library additionally keeps the mid-block tensors of the residual blocks in fp16: PCC_CONV_IN16 / OUT16 / RES16 above).  */
int pcc_network_forward(pcc_ctx* ctx, int32_t transform, int32_t filters, const float* blob, const float* x, int32_t N,
                        int32_t D, int32_t H, int32_t W, float* y, void* workspace, size_t workspace_bytes,
                        int32_t layer_flags, int32_t final_flags, void* stream);
/* The same call restricted to one family: PCC_ERR_ARG when `transform` is of another family.               */
int pcc_network_forward_analysis(pcc_ctx* ctx, int32_t transform, int32_t filters, const float* blob, const float* x,
                                 int32_t N, int32_t D, int32_t H, int32_t W, float* y, void* workspace,
                                 size_t workspace_bytes, int32_t layer_flags, int32_t final_flags, void* stream);
int pcc_network_forward_synthesis(pcc_ctx* ctx, int32_t transform, int32_t filters, const float* blob, const float* x,
                                  int32_t N, int32_t D, int32_t H, int32_t W, float* y, void* workspace,
                                  size_t workspace_bytes, int32_t layer_flags, int32_t final_flags, void* stream);
int pcc_network_forward_hyper_a(pcc_ctx* ctx, int32_t transform, int32_t filters, const float* blob, const float* x,
                                int32_t N, int32_t D, int32_t H, int32_t W, float* y, void* workspace,
                                size_t workspace_bytes, int32_t layer_flags, int32_t final_flags, void* stream);
int pcc_network_forward_hyper_s(pcc_ctx* ctx, int32_t transform, int32_t filters, const float* blob, const float* x,
                                int32_t N, int32_t D, int32_t H, int32_t W, float* y, void* workspace,
                                size_t workspace_bytes, int32_t layer_flags, int32_t final_flags, void* stream);

/* Graph phases of CompressionModelV1 / V2 (src/model_types.py:283-309, :371-411): the GPU part of compress() and of
 * decompress() in one call each; the range coder (host) sits between them.  Device pointers, NDHWC, caller-owned.        */
typedef struct {
    int32_t version;          /* 1 = CompressionModelV1 (factorized prior on y), 2 = V2 (hyperprior)                     */
    int32_t filters;
    int32_t analysis, synthesis;            /* PCC_NET_* ids; analysis < 0 for a decoder-only model                      */
    const float* w_analysis;                /* weight blobs (pcc_weights_upload); NULL where the transform is absent     */
    const float* w_synthesis;
    const float* w_hyper_analysis;
    const float* w_hyper_synthesis;
    const float* medians;                   /* (filters,) EntropyBottleneck medians                                      */
    const float* scale_table;               /* (scale_levels,) GaussianConditional scale table (V2)                      */
    int32_t scale_levels;
    int32_t round_mode;                     /* PCC_ROUND_*                                                               */
} pcc_codec_desc;
size_t pcc_codec_workspace_bytes(const pcc_codec_desc* c, int32_t N, int32_t D, int32_t H, int32_t W);
/* compress graph on N blocks: x (N,D,H,W) -> y, [z, zsym, z_hat, sigma, idx,] ysym, y_hat, x_hat (N,D,H,W; final_flags =
 * PCC_CONV_CLIP01 applies np.clip(x_hat,0,1), model_types.py:202).  The V2-only tensors may be NULL for version 1.
 * thr != NULL (fixed-threshold policy, model_opt.py:27-31) also extracts the encoder-side point lists in the same call:
 * thr, xyz, counts, cap, scratch as in pcc_threshold_compact with clip = 1.  symbols_ready: NULL or a hipEvent_t the
 * library records on `stream` as soon as zsym / idx / ysym are final, i.e. BEFORE the synthesis transform is enqueued, so
 * that the device->host copy and the host range coder overlap the synthesis.                                               */
/* io (may be NULL): the symbols on their way to / from the host coder, in the coder's stream order and integer width.
 * Encoder: before `symbols_ready` is recorded the library packs zsym, ysym and idx into the caller's device staging buffers
 * (pcc_symbols_pack), so that ONE device->host copy on the caller's side stream is all that runs beside the synthesis
 * transform (round 2 ran the permutation, the narrowing and a max-reduction there as separate kernels, which compete with the
 * one-workgroup-per-CU convolution kernels for CUs).  Decoder: io->zsym / io->ysym are the INPUT of pcc_codec_decode_hyper /
 * _main (what the host->device copy delivered; the library unpacks it into the int32 `zsym` / `ysym` argument, which is then
 * an output), io->idx receives the packed indexes of pcc_codec_decode_hyper.  Pointers of tensors a call does not touch may be
 * NULL.                                                                                                                    */
typedef struct {
    void* zsym;               /* device, stream order, sym_bytes per element (V2)                                       */
    void* ysym;
    void* idx;                /* device, stream order, idx_bytes per element (V2)                                       */
    int32_t* zsym_tile_max;   /* device int32[pcc_symbols_tiles(N, vox_z, F)]: max|symbol| per packed tile; may be NULL  */
    int32_t* ysym_tile_max;   /* device int32[pcc_symbols_tiles(N, vox_y, F)]                                           */
    int32_t sym_bytes;        /* 2 or 4                                                                                 */
    int32_t idx_bytes;        /* 1 or 4                                                                                 */
    int32_t channels_first;   /* stream order: 1 = (C, D,H,W) per block (the reference's default), 0 = (D,H,W, C)       */
} pcc_symbol_io;
int pcc_codec_encode(pcc_ctx* ctx, const pcc_codec_desc* c, const float* x, int32_t N, int32_t D, int32_t H, int32_t W,
                     float* y, float* z, int32_t* zsym, float* z_hat, float* sigma, int32_t* idx, int32_t* ysym,
                     float* y_hat, float* x_hat, const float* thr, float* xyz, int32_t* counts, int64_t cap,
                     int32_t* scratch, void* workspace, size_t workspace_bytes, int32_t layer_flags, int32_t final_flags,
                     const pcc_symbol_io* sink, void* symbols_ready, void* stream);
/* decompress graph, V2 first phase (model_types.py:403-406): zsym -> z_hat -> sigma -> idx.                               */
int pcc_codec_decode_hyper(pcc_ctx* ctx, const pcc_codec_desc* c, int32_t* zsym, int32_t N, int32_t D, int32_t H,
                           int32_t W, float* z_hat, float* sigma, int32_t* idx, void* workspace, size_t workspace_bytes,
                           int32_t layer_flags, const pcc_symbol_io* io, void* stream);
/* decompress graph, main phase (:305-307 / :407-408): ysym -> y_hat -> x_hat and, when thr != NULL, the unclipped
 * thresholding + compaction of :232-234 (arguments as pcc_threshold_compact).                                              */
int pcc_codec_decode_main(pcc_ctx* ctx, const pcc_codec_desc* c, int32_t* ysym, int32_t N, int32_t D, int32_t H,
                          int32_t W, float* y_hat, float* x_hat, const float* thr, float* xyz, int32_t* counts, int64_t cap,
                          int32_t* scratch, void* workspace, size_t workspace_bytes, int32_t layer_flags,
                          const pcc_symbol_io* io, void* stream);

/* Live kernel timing: HIP events recorded on the launch stream around layer `layer` of transform `transform` in every
 * pcc_network_forward / pcc_codec_* call (transform < 0 switches it off); pcc_profile_read waits for the recorded events,
 * returns their durations in milliseconds (at most `cap`) and clears the list.  The events belong to the context.
 * `layer` = index | (stride << 16): with stride > 1 only every stride-th call of the layer is timed (an event record costs the
 * queue a few microseconds: sampling keeps the measurement from slowing what it measures).                               */
int pcc_profile_select(pcc_ctx* ctx, int32_t transform, int32_t layer);
int pcc_profile_read(pcc_ctx* ctx, float* ms, int32_t cap, int32_t* n);

/* ---- entropy-model element-wise kernels -------------------------------------------------
 * Quantisation of tfc.EntropyBottleneck / tfc.GaussianConditional `_quantize`
 * (call sites src/model_types.py:291,382,386): mode 0 = floor(v + (0.5 - median_c)) [tfc 1.3],
 * mode 1 = round-half-even(v - median_c).  Channel of element i is i % C (channels-last).
 * medians/sym/deq may be NULL.  deq = float(sym) + median.                                      */
#define PCC_ROUND_FLOOR_HALF 0
#define PCC_ROUND_HALF_EVEN 1
int pcc_quantize(pcc_ctx* ctx, const float* v, const float* medians, int32_t* sym, float* deq,
                 size_t n, int32_t C, int32_t mode, void* stream);
/* deq[i] = float(sym[i]) + medians[i % C]  (tfc `_dequantize`, decoder side). */
int pcc_dequantize(pcc_ctx* ctx, const int32_t* sym, const float* medians, float* deq, size_t n,
                   int32_t C, void* stream);
/* Scale -> CDF-table index, src/utils/patch_gaussian_conditional.py:57-58,104-116: sigma is
 * lower-bounded at table[0]; idx = (L-1) - #{j < L-1 : sigma <= table[j]}.  Deterministic. */
int pcc_scale_to_index(pcc_ctx* ctx, const float* sigma, const float* table, int32_t L,
                       int32_t* idx, size_t n, void* stream);

/* ---- symbols / CDF-row indexes between the device tensors and the host coder -------------
 * tfc 1.3 codes each block's tensor flattened in ITS memory order (src/model_types.py:180,254,377: data_format
 * 'channels_first' -> (C, D,H,W), channel-major streams); the device tensors are NDHWC int32.  pcc_symbols_pack writes N
 * blocks of (vox, C) int32 in stream order as dst_bytes-wide integers (1: uint8, 2: int16, 4: int32; values are truncated --
 * tile_max, when given, receives max|value| of every 64 x 64 tile (pcc_symbols_tiles entries) so that the host can tell
 * whether the narrow type was enough); pcc_symbols_unpack is the inverse (decoder side).  HBM-bound byte work.            */
size_t pcc_symbols_tiles(int32_t N, int64_t vox, int32_t C);
int pcc_symbols_pack(pcc_ctx* ctx, const int32_t* src, int32_t N, int64_t vox, int32_t C, int32_t channels_first,
                     void* dst, int32_t dst_bytes, int32_t* tile_max, void* stream);
int pcc_symbols_unpack(pcc_ctx* ctx, const void* src, int32_t src_bytes, int32_t N, int64_t vox, int32_t C,
                       int32_t channels_first, int32_t* dst, void* stream);

/* ---- occupancy thresholding + order-preserving compaction ------------------------------
 * Replaces `np.argwhere(x_hat > thresholds[t]).astype(float32)` (src/model_types.py:209,233-234,
 * src/model_opt.py:12,29).  x: B blocks of D*H*W float32; thr[b] is the float32 threshold of
 * block b (device array); clip!=0 applies np.clip(x,0,1) first (encoder, model_types.py:202).
 * xyz: B * cap * 3 float32, block b's points at xyz + b*cap*3 in C order (x slowest, z fastest),
 * counts[b] = number of points of block b (may exceed cap: then only the first cap are written).
 * `scratch` must hold B * ceil(D*H*W/4096) int32.  Bit-exact with the numpy expression.         */
int pcc_threshold_compact(pcc_ctx* ctx, const float* x, int32_t B, int32_t D, int32_t H,
                          int32_t W, const float* thr, int32_t clip, float* xyz, int32_t* counts,
                          int64_t cap, int32_t* scratch, void* stream);
size_t pcc_threshold_scratch_ints(int32_t B, int32_t D, int32_t H, int32_t W);

/* sparse_to_dense (src/model_types.py:108-114): scatter ones.  pts: int32 (npts,3) local block
 * coordinates, block_of[i] = destination block; dense must be zero-filled B*D*H*W float32.       */
int pcc_voxelize(pcc_ctx* ctx, const int32_t* pts, const int32_t* block_of, int64_t npts,
                 int32_t B, int32_t D, int32_t H, int32_t W, float* dense, void* stream);

/* ---- adaptive threshold search statistics (src/model_opt.py:21-77, src/utils/pc_metric.py:76-138) ------------
 * Exact D1 sums for EVERY threshold of EVERY block in one call (the reference builds up to 255 KD-trees per block).
 * With level k(v) = #{t : x_hat[v] > thr[t]} the decoded set at threshold t is B_t = {v : k(v) > t}.  Outputs
 * (device, (B,256) uint64 unless noted, zero-filled by the call):
 *   s_ab[b][t]  = sum over the original points a of block b of min_{v in B_t} |a - v|^2        (d1_sum_AB)
 *   hsum[b][k]  = sum over voxels of level k of min_a |v - a|^2;  d1_sum_BA(t) = sum_{k>t} hsum[b][k]
 *   hcnt[b][k]  = number of voxels of level k;                    |B_t|        = sum_{k>t} hcnt[b][k]
 *   tcount[b]   (int32, (B,)) = number of thresholds whose decoded set is non-empty.
 * All sums are integers, so the host reproduces the reference's float64 metrics bit for bit.
 * pts: (npts,3) int32 block-local coordinates, block_of: (npts,) int32; thr: nthr <= 256 increasing float32
 * thresholds (device).  clip != 0 applies np.clip(x_hat,0,1) first (the encoder does, model_types.py:202).
 * Blocks up to 128^3.  `workspace`: pcc_d1_search_workspace_bytes(B,D,H,W) bytes of device memory.             */
size_t pcc_d1_search_workspace_bytes(int32_t B, int32_t D, int32_t H, int32_t W);
int pcc_d1_threshold_stats(pcc_ctx* ctx, const float* x_hat, int32_t B, int32_t D, int32_t H, int32_t W,
                           const float* thr, int32_t nthr, int32_t clip, const int32_t* pts,
                           const int32_t* block_of, int64_t npts, void* workspace, uint64_t* s_ab,
                           uint64_t* hsum, uint64_t* hcnt, int32_t* tcount, void* stream);

/* The same search with the point-to-plane statistics (D2; src/utils/pc_metric.py:109-131: the reference's experiment optimises
 * d1_mse AND d2_mse, src/ev_experiment.yml:47).  D2 needs WHICH point is nearest; where several are equally near the reference
 * takes whatever scipy's KD-tree returns (pc_metric.py:114) -- here the rule is fixed: the candidate with the lowest (x, y, z)
 * in lexicographic order (= lowest row-major voxel index).  A decoded point takes the mean normal of the original points that
 * chose it, summed in ascending point order (pc_metric.py:16-18).  Additional arguments:
 *   normals     : (npts,3) float32, normal of original point i (device)
 *   block_start : (B+1,) int32 offsets of the blocks' points inside pts (points are grouped by block, ascending) (device)
 *   workspace2  : pcc_d12_search_workspace_bytes(B,D,H,W,npts) bytes of device memory (beside `workspace` of the D1 call)
 *   d2_ab, d2_ba: (B,256) float64 (device): d2_sum_AB / d2_sum_BA of threshold t, valid for t < tcount[b]
 * No floating-point atomics: every sum has a fixed order, results are bit-reproducible.                                    */
size_t pcc_d12_search_workspace_bytes(int32_t B, int32_t D, int32_t H, int32_t W, int64_t npts);
int pcc_d12_threshold_stats(pcc_ctx* ctx, const float* x_hat, int32_t B, int32_t D, int32_t H, int32_t W, const float* thr,
                            int32_t nthr, int32_t clip, const int32_t* pts, const int32_t* block_of, const int32_t* block_start,
                            int64_t npts, const float* normals, void* workspace, void* workspace2, uint64_t* s_ab, uint64_t* hsum,
                            uint64_t* hcnt, int32_t* tcount, double* d2_ab, double* d2_ba, void* stream);

/* ---- focal loss (src/utils/focal_loss.py:5-12) ------------------------------------------
 * Deterministic two-stage reduction (wavefront DPP/shuffle tree, fixed block order); result is a
 * single float32 written to out[0] (device).  `scratch` must hold pcc_focal_scratch_floats().   */
int pcc_focal_loss(pcc_ctx* ctx, const float* y_true, const float* y_pred, size_t n, float gamma,
                   float alpha, float* out, float* scratch, void* stream);
size_t pcc_focal_scratch_floats(void);

/* ---- range coder (HOST) ----------------------------------------------------------------
 * Replaces tfc's C++ ops range_coding_ops.unbounded_index_range_encode/decode
 * (src/utils/patch_gaussian_conditional.py:27-31; src/model_types.py:291-292,382-387,404-407),
 * which the reference also runs on the CPU (patch_gaussian_conditional.py:105-106).
 * Streams are independent (one per block and per string); they are coded on `n_threads` host
 * threads (0 = hardware concurrency).  The workers belong to the CALLING thread (one persistent pool per calling
 * thread, joined when that thread exits): a host that codes from two threads at once -- the encoder of one chunk beside the
 * decoder of another -- owns 2 x n_threads workers and has to size n_threads for that; call from long-lived threads
 * (a short-lived caller pays thread creation and teardown per call sequence).
 *   data[s], index[s] : n[s] int32 symbols / CDF-row indices of stream s (host pointers);
 *   index[s] == NULL  : row = i % index_mod (EntropyBottleneck: per-channel tables);
 *   cdf               : (rows, cdf_stride) int32, row r valid for cdf_size[r] entries;
 *   offset[r]         : smallest in-table value of row r.                                        */
typedef struct {
    const int32_t* cdf;
    const int32_t* cdf_size;
    const int32_t* offset;
    int32_t rows, cdf_stride, precision, overflow_width;
} pcc_cdf_table;

int pcc_range_encode_batch(const pcc_cdf_table* t, int32_t n_streams, const int32_t* const* data,
                           const int32_t* const* index, int32_t index_mod, const size_t* n,
                           uint8_t* const* out, const size_t* cap, size_t* out_len,
                           int32_t n_threads);
int pcc_range_decode_batch(const pcc_cdf_table* t, int32_t n_streams, const uint8_t* const* str,
                           const size_t* str_len, const int32_t* const* index, int32_t index_mod,
                           const size_t* n, int32_t* const* out, int32_t n_threads);
/* The same coders on narrow host arrays (round 3): data_bytes / out_bytes 2 (int16) or 4 (int32) symbols, index_bytes 1 (uint8)
 * or 4 (int32) CDF rows -- the codec moves y symbols as int16 and the 64 Gaussian scale rows as uint8 across PCIe.  Decoding
 * into 16 bits returns PCC_ERR_SPACE when a symbol does not fit (decode again with out_bytes 4).                              */
int pcc_range_encode_batch_n(const pcc_cdf_table* t, int32_t n_streams, const void* const* data, int32_t data_bytes,
                             const void* const* index, int32_t index_bytes, int32_t index_mod, const size_t* n,
                             uint8_t* const* out, const size_t* cap, size_t* out_len, int32_t n_threads);
int pcc_range_decode_batch_n(const pcc_cdf_table* t, int32_t n_streams, const uint8_t* const* str, const size_t* str_len,
                             const void* const* index, int32_t index_bytes, int32_t index_mod, const size_t* n,
                             void* const* out, int32_t out_bytes, int32_t n_threads);
/* tfc `pmf_to_quantized_cdf` (src/utils/patch_gaussian_conditional.py:87-89): pmf[n] -> cdf[n+1]. */
int pcc_pmf_to_quantized_cdf(const float* pmf, int32_t n, int32_t precision, int32_t* cdf);

/* ---- octree blocking, host (replaces the per-point loop of src/utils/octree_coding.py:82-108) ----------------------
 * Buckets `n` points (row-major doubles, `ncols` >= 3 columns, x y z first) into blocks of edge `block_size`:
 * bucket = Morton code of the block id over `level` bits per axis, x least significant.  order[n] receives the point
 * indices sorted by bucket, input order kept inside a bucket; bucket_count[8^level] the points per bucket.
 * Returns the number of occupied buckets (>= 0) or a negative status.  level <= 7.                                   */
int64_t pcc_octree_bucket(const double* points, int64_t n, int32_t ncols, int32_t block_size, int32_t level,
                                     int64_t* order, int64_t* bucket_count);

#ifdef __cplusplus
}
#endif
#endif /* PCC_GEO_H */
