/*
 * semseg_hip.h -- C ABI of libsemseg_hip.so, the gfx950 (MI355X / CDNA4) kernel
 * library underneath the HRNet-OCR-MScale hot path.
 *
 * The reference (NVIDIA/semantic-segmentation) has no FFI of its own: every
 * arithmetic op on the hot path is a PyTorch/cuDNN/apex call made from Python
 * (SURVEY.md section 2b, rows K1..K15).  Each entry point below replaces one of
 * those call sites; the reference file:line is cited per function.  The Python
 * binding is ctypes (semseg_amd/_lib.py): plain pointers, ints and a
 * hipStream_t passed as void*.  No torch types cross this boundary.
 *
 * Conventions
 *  - every function returns 0 on success, SSA_EINVAL(-1)/SSA_EUNSUPPORTED(-2)
 *    for bad arguments, or a positive hipError_t.
 *  - all pointers are DEVICE pointers borrowed for the duration of the call;
 *    nothing is allocated, freed, retained or synchronised inside the library
 *    (hipGraph-capture safe, callable from the autograd thread).
 *  - activations are NHWC.  "ld" arguments are the pixel stride in elements so
 *    a tensor may be a channel slice of a wider buffer (concat-free writes).
 *  - bf16 tensors are 16-bit words, 16-byte aligned, channel count % 8 == 0.
 */
#ifndef SEMSEG_HIP_H
#define SEMSEG_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSA_OK 0
#define SSA_EINVAL (-1)
#define SSA_EUNSUPPORTED (-2)

int ssa_version(void);
/* Storage format of the 16-bit activations this build was compiled for (csrc/common.h): 0 = bf16
 * (libsemseg_hip.so), 1 = fp16 (libsemseg_hip_f16.so, -DSSA_ELEM_F16; the reference's --fp16 / apex O1 format,
 * train.py:381).  Every `bf16` in the names and comments below reads "the build's 16-bit element".            */
int ssa_elem_type(void);
/* First 16 hex digits of sha256 over the kernel sources the library was built from (csrc/Makefile): the Python
 * loader recomputes it over the sources next to the binary and refuses a stale build.                          */
const char* ssa_source_sha(void);

/* ------------------------------------------------------- grouped launches --
 * The 2-4 resolution branches of a HighResolutionModule (network/hrnetv2.py:181-254)
 * times the scale passes of MscaleOCR (network/ocrnet.py:264-327) are independent
 * problems that the reference issues as separate cuDNN/ATen calls.  Between
 * ssa_group_begin() and ssa_group_end(stream) the group-aware entry points (convolutions,
 * BatchNorm passes, sums, bilinear resampling, weight gradients) queue their launch on
 * the calling thread instead of issuing it; ssa_group_end issues ONE launch per kernel
 * instantiation covering all queued problems (job table in the kernel arguments).
 * Contract: launches queued in one bracket are mutually independent.  Other entry
 * points launch immediately.  Brackets nest (the outermost end flushes);
 * ssa_group_abort drops a bracket after a host-side error.
 * ssa_launch_count: kernel launches issued by this library so far (reset != 0 clears). */
int ssa_group_begin(void);
int ssa_group_end(void* stream);
int ssa_group_abort(void);
long ssa_launch_count(int reset);

/* Per-launch timing for the roofline measurement (bench.py): between ssa_profile_begin and
 * ssa_profile_end every launch of a group-aware kernel (all convolution-class kernels,
 * BatchNorm, sums, resampling) is bracketed by HIP events on the stream it is launched on.
 * ssa_profile_note attaches algorithmic flops / HBM bytes to the NEXT job submitted by the
 * calling thread.  ssa_profile_end waits for the device and returns one aggregate record
 * per kernel instantiation (name as rocprofv3 prints the template arguments).            */
typedef struct ssa_profile_rec {
  char kernel[120];
  long launches, jobs;
  double total_us, flops, bytes;
} ssa_profile_rec;
int ssa_profile_begin(void);
int ssa_profile_note(double flops, double bytes);
int ssa_profile_end(ssa_profile_rec* out, int max_recs);

/* ------------------------------------------------------------------ conv --
 * Implicit-GEMM convolution on MFMA (v_mfma_f32_32x32x16_bf16), NHWC bf16 in,
 * fp32 accumulate, bf16 or fp32 out.  Replaces nn.Conv2d forward / dgrad /
 * wgrad at network/hrnetv2.py:31-34,42-45,76-77,270-275 (K1,K2),
 * network/ocrnet.py:54-58 (K3), network/utils.py:348-357 (K4), every 1x1 in
 * network/ocr_utils.py:68-93,142-147 (K5) and the dilated ASPP convs at
 * network/utils.py:192-198 (K6). */
typedef struct ssa_conv_desc {
  int B, H, W, Cin;   /* input  [B,H,W,Cin], Cin % 8 == 0                      */
  int ldx;            /* input pixel stride (elements)                         */
  int Ho, Wo, Cout;   /* output [B,Ho,Wo,Cout]                                 */
  int ldy;            /* output pixel stride (elements)                        */
  int KH, KW;         /* filter taps                                           */
  int stride, pad, dil;
  int transposed;     /* 0: iy = oy*stride - pad + kh*dil   (forward)
                         1: iy = (oy - pad + kh*dil)/stride when divisible
                            (data-gradient of a strided forward conv)          */
  int Kpad;           /* row length of the packed filter matrix (multiple of 32) */
  int out_f32;        /* 0: bf16 output, 1: fp32 output                        */
  int cfg;            /* tile configuration id, -1 = choose automatically       */
} ssa_conv_desc;

/* y[m, n] = bias[n] + sum_k A[m, k] * Wp[n, k];  m=(b,oy,ox), k=(kh,kw,ci). */
int ssa_conv2d_igemm(const ssa_conv_desc* d, const void* x, const void* w_packed,
                     const float* bias, void* y, void* stream);

/* Same, with the BatchNorm batch statistics of the bf16-rounded outputs accumulated in
 * the epilogue (stats: [ssa_bn_stat_replicas()][2][Cout] fp64, caller clears; bf16
 * output, forward form only) -- replaces the ssa_bn_stats pass after the conv.        */
int ssa_conv2d_igemm_stats(const ssa_conv_desc* d, const void* x, const void* w_packed,
                           const float* bias, void* y, double* stats, void* stream);

/* Halo-tile convolution for the small-channel 3x3 stride-1 "same" convs of the
 * HRNet branches (Cin in {48, 64, 96}): the input halo tile is staged in LDS
 * once, all taps are computed from it, the filter streams from L2 in fragment
 * order (ssa_pack_filter mode 2/3).  stats (optional): BatchNorm partial sums
 * [ssa_bn_stat_replicas()][2][Cout] fp64, ACCUMULATED (caller clears), of the
 * bf16-rounded outputs -- the conv epilogue replaces the ssa_bn_stats pass.
 * ssa_conv2d_tile_supported: 1 if this descriptor can run on this kernel.     */
int ssa_conv2d_tile_supported(const ssa_conv_desc* d);
int ssa_conv2d_tile(const ssa_conv_desc* d, const void* x, const void* w_frag,
                    const float* bias, void* y, double* stats, void* stream);
int ssa_bn_stat_replicas(void);
/* The same convolution as the DATA GRADIENT of a residual block's conv (flipped, transposed
 * filter: ssa_pack_filter mode 3), with a second tile `aux` ([B,H,W,>=Cout] bf16, pixel stride
 * ldaux), congruent with the output, folded into the epilogue:
 *   aux_mode 1: y = bf16(bf16(conv) + aux) -- the gradient of the identity branch
 *               (network/hrnetv2.py:58-64 `out += residual`) added where autograd would launch
 *               a separate add; stats must be NULL.
 *   aux_mode 2: aux is the input x of the BatchNorm+ReLU layer whose output this conv consumed
 *               (conv1 -> bn1 -> relu -> conv2, network/hrnetv2.py:44-56); besides y = dz the
 *               epilogue ACCUMULATES that layer's backward sums into stats
 *               [ssa_bn_stat_replicas()][2][Cout] fp64:  sum(m*dz), sum(m*dz*xhat),
 *               m = [coef[0][c]*x + coef[1][c] > 0], xhat = (x - coef[2][c]) * coef[3][c]
 *               (coef = the [4][Cout] table ssa_bn_apply_train wrote) -- replaces the
 *               ssa_bn_bwd_reduce pass over (x, dz) of that layer.
 * Same shape support as ssa_conv2d_tile.                                                  */
int ssa_conv2d_tile_aux(const ssa_conv_desc* d, const void* x, const void* w_frag,
                        const float* bias, void* y, double* stats, const void* aux, int ldaux,
                        const float* coef, int aux_mode, void* stream);

/* Persistent, software-pipelined form of ssa_conv2d_tile / ssa_conv2d_tile_aux for the trunk's 48/96/192/384-channel
 * 3x3 convs (csrc/conv_tile_p.hip): a workgroup walks a strip of tiles of one (problem, 32-channel n-block), the input
 * in chunks of 48 channels, the filter streamed as one continuous LDS-DMA pipeline through a ring of three buffers,
 * the next halo in flight during the MFMAs, a register-only epilogue (swapped MFMA operands + v_permlane32_swap) and
 * the BatchNorm statistics in registers over the strip; three workgroups per CU.  No bias (SSA_EUNSUPPORTED).
 * stats / aux / aux_mode / coef as ssa_conv2d_tile_aux (aux_mode 0: none).
 * aux_mode 3 / 4 (inference; stats NULL): the conv's BatchNorm in evaluation mode as the epilogue -- z = scale * y + shift
 * (+ aux, the residual tile, when given), 4: followed by the ReLU; coef = the layer's [4][Cout] table (rows 0, 1 = scale,
 * shift: ssa_bn_finalize / ssa_bn_finalize_eval_batched); y is rounded to 16 bits first, so z equals ssa_bn_apply on the
 * stored conv output bit for bit.  conv -> bn -> relu of network/hrnetv2.py:37-66 in eval() as one launch.
 * ssa_conv_tile_strip(units): work (in units of one 128-pixel tile x one 48-channel chunk x one n-block = 27 MFMAs
 * per wave) a workgroup of the calling thread's NEXT launches should carry -- the caller of a grouped level knows the
 * level's total; 0 = derive from each problem alone.                                                              */
int ssa_conv2d_tile_p_supported(const ssa_conv_desc* d);
int ssa_conv_tile_strip(int units);
int ssa_conv2d_tile_p(const ssa_conv_desc* d, const void* x, const void* w_frag, const float* bias, void* y,
                      double* stats, const void* aux, int ldaux, const float* coef, int aux_mode, void* stream);

/* Halo-chunk implicit GEMM for the large-channel 3x3 / 1x1 stride-1 "same" convs
 * of the OCR and attention heads (Cin >= 192, Cin % 48 == 0 or % 64 == 0):
 * 256-pixel x 128-channel workgroup tiles, the input halo tile of each 48/64
 * channel chunk staged in LDS once for all 9 taps, the filter (fragment order,
 * ssa_pack_filter mode 2/3) streamed by global_load_lds.  bf16 or fp32 output;
 * stats as for ssa_conv2d_tile (bf16 output only).                             */
int ssa_conv2d_halo_supported(const ssa_conv_desc* d);
int ssa_conv2d_halo(const ssa_conv_desc* d, const void* x, const void* w_frag,
                    const float* bias, void* y, double* stats, void* stream);

/* Wide-tile GEMM for the large 1x1 stride-1 convs of the OCR / attention heads and their data gradients (aux_head
 * 720->720, SpatialOCR 1024->512, f_up 256->512; network/ocrnet.py:61-76, network/ocr_utils.py:68-93,142-156):
 * 256 pixels x 256 output channels per workgroup, 128 x 64 per wave (4 x 2 MFMA 32x32x16 tiles), both operands by
 * global_load_lds in 32-channel stages through a ring of four LDS buffers (csrc/conv_gemm_wide.hip).  Same operands
 * as ssa_conv2d_halo (w_frag: ssa_pack_filter mode 2 / 3 with cin_pad = d->Cin, a multiple of 16); 16-bit output
 * only; stats as for ssa_conv2d_tile.  ssa_conv2d_halo forwards the problems for which
 * ssa_conv2d_gemm_wide_supported() returns 1 (1x1, more than 256 output channels, >= 16384 pixels; SSA_GEMM_WIDE=0
 * keeps them on the 256 x 128 kernel); the entry point itself takes any 1x1 stride-1 problem of >= 256 pixels.   */
int ssa_conv2d_gemm_wide_supported(const ssa_conv_desc* d);
int ssa_conv2d_gemm_wide(const ssa_conv_desc* d, const void* x, const void* w_frag, const float* bias,
                         void* y, double* stats, void* stream);

/* Second geometry for the 3x3 stride-1 "same" convs of the OCR / attention heads and their data gradients
 * (conv3x3_ocr 720->512, attn 512->256 / 256->256; network/ocrnet.py:54-58, network/utils.py:348-357): 128 pixels x
 * 256 channels per 4-wave workgroup, 128 x 64 per wave (4 x 2 MFMA 32x32x16 tiles); the filter fragments go straight
 * from global memory into the MFMA operand registers through a ring six k-steps deep (they never touch LDS), the input
 * halo tile by LDS DMA, double buffered per 48 / 64-channel chunk: one workgroup barrier per chunk
 * (csrc/conv_halo_reg.hip).  Same operands as ssa_conv2d_halo; 16-bit output only; stats as for ssa_conv2d_tile.
 * ssa_conv2d_halo forwards the 3x3 problems this entry point supports (SSA_HALO3_REG=0 keeps them on the 256 x 128
 * LDS-ring kernel).                                                                                              */
int ssa_conv2d_halo_reg_supported(const ssa_conv_desc* d);
int ssa_conv2d_halo_reg(const ssa_conv_desc* d, const void* x, const void* w_frag, const float* bias,
                        void* y, double* stats, void* stream);

/* Tile configuration ssa_conv2d_igemm would use for this problem
 * (0: 128x128, 1: 256x64, 2: 128x96, 3: 256x32, 4: 64x64, 5: 128x64 tiles). */
int ssa_conv2d_igemm_tile(const ssa_conv_desc* d);
/* ssa_conv2d_igemm with the conv's BatchNorm in evaluation mode as the epilogue (inference): z = scale * y + shift
 * (+ residual [B,Ho,Wo,Cout], pixel stride ldres) and, relu != 0, the ReLU; coef = the layer's [4][Cout] table (rows 0, 1);
 * y = the conv output (+ bias) rounded to 16 bits first, so z equals ssa_bn_apply on the stored output bit for bit.
 * 16-bit output, not transposed.  conv -> bn (-> relu) of the stem, layer1 and the fuse layers
 * (network/hrnetv2.py:278-300, 192-222) in eval() as one launch.                                                  */
int ssa_conv2d_igemm_affine(const ssa_conv_desc* d, const void* x, const void* w_packed, const float* bias, void* y,
                            const float* coef, const void* residual, int ldres, int relu, void* stream);


/* Data gradient of a 3x3, stride-2, pad-1 convolution (the fuse / transition / stem
 * down-convs of network/hrnetv2.py:218-250, 357-371; cuDNN's backward-data behind
 * nn.Conv2d) decomposed by OUTPUT PARITY: the pixels (2m+py, 2n+px) of dx depend on
 * (1+py)*(1+px) of the 9 taps only, so the four classes are four dense stride-1
 * correlations of dy with 1, 2, 2 and 4 taps -- no multiplications by the zeros of the
 * zero-inserted form (ssa_conv2d_igemm with transposed = 1 does 4x the MFMA work).
 * dy: [B,Ho,Wo,cout_pad] bf16 (row stride lddy), dx: [B,H,W,Cin] bf16 (lddx), every pixel
 * written.  w_cls[c], kpad_cls[c]: operand of class c = 2*py+px packed by
 * ssa_pack_filter(mode 4+c) -- [Cin][Kpad], k = (jy, jx, co).  Inside a group bracket the
 * four problems share one launch.                                                        */
int ssa_conv2d_dgrad_s2(int B, int H, int W, int Cin, int lddx, int Ho, int Wo, int cout_pad,
                        int lddy, const void* dy, const void* const* w_cls, const int* kpad_cls,
                        void* dx, void* stream);

/* Filter packing: OIHW fp32 parameter -> bf16 [rows][Kpad] GEMM operand.
 * mode 0 (forward): rows = Cout, k = (kh,kw,ci) with ci < cin_pad.
 * mode 1 (dgrad)  : rows = Cin,  k = (kh',kw',co) with co < cout_pad, taps
 *                   flipped (kh' = KH-1-kh) -- the transposed filter.
 * mode 2 / 3      : the same two operands in MFMA-fragment order for
 *                   ssa_conv2d_tile: [n-block][k-step][lane][8], rows padded to
 *                   a multiple of 32, Kpad = KH*KW*c_pad (a multiple of 16).
 * mode 4 + 2*py + px (3x3 only): data-gradient operand of parity class (py, px) of a
 *                   stride-2 conv (ssa_conv2d_dgrad_s2): rows = Cin,
 *                   k = (jy, jx, co), jy <= py, jx <= px, co < cout_pad; tap j of
 *                   class 0 is forward tap 1, of class 1 forward tap 2 - 2j.       */
int ssa_pack_filter(const float* w_oihw, void* w_packed, int Cout, int Cin,
                    int KH, int KW, int cin_pad, int cout_pad, int Kpad,
                    int mode, void* stream);

/* Weight gradient, split over the pixel axis.  partial is
 * [nsplit][cout_pad][KH*KW*Cin] fp32; ssa_conv2d_wgrad_reduce sums the splits
 * and writes dW in the parameter's OIHW fp32 layout (cin_real <= d->Cin);
 * accumulate != 0: ADDS to dW instead (dW = the layer's slice of the step's gradient
 * arena, cleared once per step; the passes of a multi-scale step write their splits
 * behind one another in `partial` and ONE reduce sums them all).
 * The weight-gradient kernels and the reduce are group-aware: the host defers a
 * module's weight gradients and issues them inside one ssa_group bracket.
 * d->cfg > 0 in the *_plan calls: bound a workgroup's pixel strip to cfg 128-pixel
 * stages instead of splitting for parallelism (grouped launches get their
 * parallelism from the number of layers).                                     */
int ssa_conv2d_wgrad_plan(const ssa_conv_desc* d, int cout_pad, int* nsplit,
                          size_t* ws_bytes);
int ssa_conv2d_wgrad(const ssa_conv_desc* d, const void* x, const void* dy,
                     int lddy, int cout_pad, int nsplit, float* partial,
                     void* stream);
int ssa_conv2d_wgrad_reduce(const float* partial, int nsplit, int cout_pad,
                            int Cout, int Cin_pad, int Cin, int KH, int KW,
                            float* dw_oihw, int accumulate, void* stream);

/* All filters of a network repacked in ONE launch (the parameters change every
 * optimizer step; 1,276 separate pack launches cost more than the packing).
 * jobs_dev: device array of ssa_pack_job; same semantics as ssa_pack_filter.  */
typedef struct ssa_pack_job {
  const float* w_oihw;
  void* w_packed;
  long elem_begin;   /* ssa_pack_filters_tiled: 1 + index of the next job packed from the same OIHW tensor (its tiles
                      * are then NOT listed: the owner's workgroups write every chained form from one read); 0: none.
                      * ssa_pack_filters_batched ignores it                                                   */
  int Cout, Cin, KH, KW, cin_pad, cout_pad, Kpad, mode, rows;
  int layout;        /* reserved, 0: mode 2 / 3 write the one fragment order of conv_tile(_p).hip / conv_halo_gemm.hip */
} ssa_pack_job;
int ssa_pack_filters_batched(const void* jobs_dev, int njobs, int blocks_per_job,
                             void* stream);
/* The same repack balanced by TILE: tiles_dev = device array of int4 {job index, first output
 * channel (multiple of 32), first input channel (multiple of ct), ct} covering every job's
 * [Cout] x [Cin] plane with 32 x ct tiles, ct = ssa_pack_tile_channels(KH, KW) (0: filter
 * too large for this path -> use ssa_pack_filters_batched).  Writes the data elements only:
 * the destinations' zero padding must be in place (cleared buffers packed once by
 * ssa_pack_filter).  max_ct_taps = the largest ct * KH * KW over the tiles (sizes the LDS
 * tile).  One coalesced read of each OIHW tile serves either operand form.                 */
int ssa_pack_tile_channels(int KH, int KW);
int ssa_pack_filters_tiled(const void* jobs_dev, const void* tiles_dev, int ntiles,
                           int max_ct_taps, void* stream);

/* Halo-staged weight gradient for the 3x3 stride-1 trunk convs with
 * Cin == cout_pad in {48, 64, 96, 192, 384}: persistent workgroups keep their
 * block of dW in MFMA accumulators while they walk 128-pixel tiles whose x halo
 * image and dy tile are staged in LDS once (transposing ds_read_b64_tr_b16
 * fragment reads).  Same partial layout as ssa_conv2d_wgrad
 * ([nsplit][cout_pad][9*Cin] fp32) -> finish with ssa_conv2d_wgrad_reduce.
 * _plan returns SSA_EUNSUPPORTED for shapes this kernel does not take.          */
int ssa_conv2d_wgrad_tile_plan(const ssa_conv_desc* d, int cout_pad, int* nsplit,
                               size_t* ws_bytes);
int ssa_conv2d_wgrad_tile(const ssa_conv_desc* d, const void* x, const void* dy,
                          int lddy, int cout_pad, int nsplit, float* partial,
                          void* stream);
/* Launch geometry of ssa_conv2d_wgrad_tile for a Cin = cout_pad layer, for the host's strip fitting
 * (hip_backend._fit_tile_strips): parts = workgroups per strip of 128-pixel tiles, slots = workgroups of that
 * instantiation resident on the chip at once, kind = which instantiation (launches of one kind share a grouped launch:
 * 0 = the 8-wave all-taps form of the 96-channel blocks, csrc/conv_wgrad_tile.hip ConvWgradTileA). */
int ssa_conv2d_wgrad_tile_geometry(int Cin, int cout_pad, int* parts, int* slots, int* kind);
/* Weight gradient of the large-channel 3x3 / 1x1 stride-1 head convs (Cin >= 128,
 * >= 16 K pixels): 8-wave workgroups persistent over 128-pixel tiles hold a
 * 128(co) x 128(ci) x 3(kw) [3x3] or 128 x 256 [1x1] block of dW in MFMA
 * accumulators; LDS images pixel-major, transposing fragment reads.  Partial
 * layout as ssa_conv2d_wgrad -> finish with ssa_conv2d_wgrad_reduce.            */
int ssa_conv2d_wgrad_head_plan(const ssa_conv_desc* d, int cout_pad, int* nsplit,
                               size_t* ws_bytes);
int ssa_conv2d_wgrad_head(const ssa_conv_desc* d, const void* x, const void* dy,
                          int lddy, int cout_pad, int nsplit, float* partial,
                          void* stream);

/* Column sum over pixels: out[c] = sum_p x[p, c]  (bias gradient). x bf16. */
int ssa_colsum_bf16(const void* x, long P, int C, int ld, float* out,
                    double* scratch2c, void* stream);

/* fp32 [P,C] -> bf16 [P,Cpad] zero padded (gradient of the fp32 heads). */
int ssa_pad_cast_f32_bf16(const float* x, long P, int C, int ldx, void* y,
                          int Cpad, void* stream);

/* --------------------------------------------------------------- batchnorm --
 * Replaces cfg.MODEL.BNFUNC (nn.BatchNorm2d / apex SyncBatchNorm) selected at
 * config.py:216-225 and instantiated by network/mynn.py:18-24 (K7, C3).      */
/* sums[0:C] += sum_p x, sums[C:2C] += sum_p x^2 (fp64).  zero_sums=1 clears sums
 * first (one memset per call); callers that carve sums out of an arena they
 * clear once per step pass 0.                                                  */
int ssa_bn_stats(const void* x, long P, int C, int ld, double* sums, int zero_sums,
                 void* stream);
/* Training-mode normalisation with the finalize step fused in (one launch):
 * z = post*act(bn(x) + residual) with batch statistics from `sums`/`count`
 * (possibly all-reduced: SyncBN).  sums is [nrep][2][C]: nrep = 1 after
 * ssa_bn_stats, ssa_bn_stat_replicas() after a conv epilogue
 * (ssa_conv2d_tile) -- the replicas are summed here.  Writes coef = [scale|shift|mean|invstd] (4*C
 * fp32) for the backward pass.  Running statistics: either updated here
 * (running_mean/var + *num_batches_tracked given; momentum, unbiased variance)
 * or deferred: pass_stats (2*C+1 fp32: mean, biased var, count) is filled and
 * ssa_bn_update_running_batched applies every layer's passes in order later
 * -- required when passes over the same layer run on concurrent streams.      */
int ssa_bn_apply_train(const void* x, int ldx, const void* residual, int ldr, void* z,
                       int ldz, long P, int C, const double* sums, int nrep, double count,
                       const float* gamma, const float* beta, float* running_mean,
                       float* running_var, long* num_batches_tracked, float momentum,
                       float eps, float* coef, float* pass_stats, int relu,
                       const float* post, long pix_per_img, void* sign_mask, void* stream);
/* sign_mask (optional, [P][C/8] bytes): bit j of byte (p, g) = z[p][8g + j] > 0 as stored.  Handed to the two
 * backward passes below in place of z: behind a residual add the ReLU mask cannot be recomputed from x alone
 * (mask_scale / mask_shift), and reading 1 byte instead of 16 per piece takes 28 MB off the 118 MB a trunk
 * level's bn2 backward moves.                                                                               */
typedef struct ssa_bn_update_job {
  float* running_mean;
  float* running_var;
  long* num_batches_tracked;
  const float* pass_stats[8];
  int C, npass;
  float momentum;
  int pad_;
} ssa_bn_update_job;
/* jobs_dev: device array of ssa_bn_update_job; one launch for all layers.     */
int ssa_bn_update_running_batched(const void* jobs_dev, int njobs, int max_channels,
                                  void* stream);
/* Evaluation-mode coefficients of many layers in one launch (use_running = 1 of ssa_bn_finalize, the same arithmetic):
 * coef = [4][C] scale, shift, mean, invstd.  nn.BatchNorm2d.eval() of the reference (network/mynn.py:18-24) computes them
 * inside every cuDNN call; here they were one launch per layer and scale pass.  jobs_dev: device array.            */
typedef struct ssa_bn_eval_job {
  const float* gamma;          /* NULL: 1 */
  const float* beta;           /* NULL: 0 */
  const float* running_mean;
  const float* running_var;
  float* coef;
  int C;
  float eps;
} ssa_bn_eval_job;
int ssa_bn_finalize_eval_batched(const void* jobs_dev, int njobs, int max_channels, void* stream);
/* From (possibly all-reduced) sums and total count: scale/shift for the apply
 * pass, mean/invstd for backward, running-stat update (momentum, unbiased var).
 * use_running=1 (eval): scale/shift from running stats, sums ignored.          */
int ssa_bn_finalize(const double* sums, double count, int C, const float* gamma,
                    const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, int use_running, float* scale,
                    float* shift, float* mean, float* invstd, void* stream);
/* z = post[b,c] * act(scale[c]*x + shift[c] + residual)                       */
int ssa_bn_apply(const void* x, int ldx, const void* residual, int ldr, void* z,
                 int ldz, long P, int C, const float* scale, const float* shift,
                 int relu, const float* post, long pix_per_img, void* stream);
/* mask_scale/mask_shift (both backward passes, optional): the ReLU mask is recomputed
 * as scale[c]*x + shift[c] > 0 (the forward's own coefficients) instead of read from
 * z -- z may then be NULL: one tensor less to read, and to keep, per BN+ReLU layer.
 * backward pass 1: sum g and sum g*xhat, g = dz*post*(z>0), accumulated into
 * sums[nrep][2][C] (workgroup b adds into replica b % nrep: fewer same-address
 * fp64 atomics); ssa_bn_bwd_apply sums the replicas.                           */
int ssa_bn_bwd_reduce(const void* x, int ldx, const void* dz, int lddz,
                      const void* z, int ldz, long P, int C, const float* mean,
                      const float* invstd, int relu, const float* post,
                      long pix_per_img, double* sums, int nrep, int zero_sums,
                      const float* mask_scale, const float* mask_shift, const void* sign_mask, void* stream);
/* backward pass 2: dx = gamma*invstd*(g - sum_g/N - xhat*sum_gxhat/N);
 * dres (optional) = g.  sums may have been all-reduced; count is global.
 * dgamma/dbeta (optional): = param_grad_scale * sums[C:2C] / sums[0:C]
 * (1/world under SyncBN, so that DDP's mean over ranks is unchanged);
 * accumulate_param_grads: ADD them (fp32 atomics) into buffers the caller cleared --
 * the gradient arena of the step, shared by every pass over the layer.        */
int ssa_bn_bwd_apply(const void* x, int ldx, const void* dz, int lddz,
                     const void* z, int ldz, void* dx, int lddx, void* dres,
                     int lddres, long P, int C, const float* gamma,
                     const float* mean, const float* invstd, const double* sums,
                     int nrep, double count, int relu, const float* post,
                     long pix_per_img, float* dgamma, float* dbeta,
                     float param_grad_scale, const float* mask_scale,
                     const float* mask_shift, int accumulate_param_grads, const void* sign_mask, void* stream);
/* Backward reduce + apply as ONE launch for the BatchNorm layers whose sums no conv epilogue has formed (bn2 of a
 * BasicBlock, network/hrnetv2.py:53-64): phase 1 forms sum g / sum g xhat as ssa_bn_bwd_reduce does, a grid-wide
 * rendezvous (one atomic ticket per workgroup) follows, phase 2 applies them to the chunk the workgroup still holds in
 * registers -- (x, dz, mask) are read once instead of twice.  Same arguments as ssa_bn_bwd_apply (sums: zeroed
 * [nrep][2][C], accumulated here) plus `ticket`, one zeroed 32-bit word per call.  Every workgroup of the launch must be
 * able to be resident at once: ssa_bn_bwd_fused_blocks(P, C) = workgroups the problem takes (0: unsupported -- more
 * than one chunk per workgroup, emulation build), ssa_bn_bwd_fused_capacity() = workgroups the device holds; the caller
 * keeps a bracket's total within it and takes the two-launch form otherwise, and whenever a SyncBN exchange has to happen
 * between the halves.  A workgroup that waits ~1 s gives up and counts itself: ssa_bn_bwd_fused_timeouts (0 = never). */
int ssa_bn_bwd_fused_blocks(long P, int C);
int ssa_bn_bwd_fused_capacity(void);
int ssa_bn_bwd_fused_timeouts(unsigned* out);
int ssa_bn_bwd_fused(const void* x, int ldx, const void* dz, int lddz, const void* z, int ldz,
                     void* dx, int lddx, void* dres, int lddres, long P, int C, const float* gamma,
                     const float* mean, const float* invstd, double* sums, int nrep,
                     double count, int relu, const float* post, long pix_per_img, float* dgamma,
                     float* dbeta, float param_grad_scale, const float* mask_scale,
                     const float* mask_shift, int accumulate_param_grads, const void* sign_mask, void* ticket,
                     void* stream);
/* dgamma[c] = sums[C+c], dbeta[c] = sums[c] (fp64 -> fp32)                     */
int ssa_bn_param_grads(const double* sums, int C, float* dgamma, float* dbeta,
                       void* stream);

/* ------------------------------------------------- SyncBN exchange, peer to peer ----
 * One-shot all-reduce (SUM, fp64, in place) of the SyncBN partial sums over peer-mapped device memory -- what
 * apex.parallel.SyncBatchNorm's all-reduce is in the reference (config.py:216-222, network/__init__.py:37-39) and
 * ncclAllReduce is on this path by default.  Every rank writes its n values into its slot of every peer's exchange
 * buffer, publishes a sequence number (system-scope release), waits for all peers' numbers in its own buffer and sums
 * the slots in rank order: one single-workgroup kernel on `stream`, capturable (the sequence number lives in *seq_dev and
 * is advanced by the kernel).  peers_dev: device array of `world` base addresses -- every rank's buffer as mapped into
 * THIS process (hipIpcOpenMemHandle; the rank's own allocation at [rank]); each buffer *bytes of ssa_p2p_buffer_bytes(world,
 * slot_doubles, &bytes) of zeroed fine-grained device memory.  n <= slot_doubles, else SSA_EUNSUPPORTED (the caller keeps
 * the collective library for those).  ssa_p2p_timeouts: ranks that gave up waiting (~2 s) -- 0 unless a peer died.
 * Host side: semseg_amd/p2p.py (opt-in, SSA_SYNCBN_P2P=1).                                                        */
int ssa_p2p_buffer_bytes(int world, long slot_doubles, size_t* bytes);
int ssa_p2p_allreduce_f64(double* data, long n, void* const* peers_dev, int rank, int world,
                          unsigned long long* seq_dev, long slot_doubles, void* stream);
int ssa_p2p_timeouts(unsigned* out);
/* Exchange buffers without hipIpc: *ptr = `bytes` (rounded up to the allocation granularity -> *mapped_bytes) of zeroed
 * uncached device memory on the current device, created through the virtual-memory API and exported as the POSIX file
 * descriptor *fd (the caller hands it to the peers over a unix socket, SCM_RIGHTS, and closes it).  A peer maps the
 * allocation with ssa_p2p_vmm_import(fd, mapped_bytes, &ptr) on ITS current device.  Needs no ptrace rights, which
 * hipIpcOpenMemHandle in dmabuf mode (pidfd_getfd) does.  SSA_EUNSUPPORTED when the runtime refuses every variant.   */
int ssa_p2p_vmm_alloc(size_t bytes, void** ptr, int* fd, size_t* mapped_bytes);
int ssa_p2p_vmm_import(int fd, size_t mapped_bytes, void** ptr);
int ssa_p2p_vmm_unmap(void* ptr, size_t mapped_bytes);

/* ----------------------------------------------------------- elementwise ---- */
/* z = relu?(a + b + c + d); b,c,d optional.  HRNet fuse sum,
 * network/hrnetv2.py:236-252 (K12). All bf16, dense [n] with n % 8 == 0.      */
int ssa_sum_act(const void* a, const void* b, const void* c, const void* d,
                void* z, long n, int relu, void* stream);
/* g = dz * (z > 0) */
int ssa_relu_bwd(const void* dz, const void* z, void* g, long n, void* stream);
/* NCHW fp32 image -> NHWC bf16 with channels zero-padded to cpad.
 * (train.py:487 hands the module an NCHW fp32 batch.)                         */
int ssa_nchw_f32_to_nhwc_bf16(const float* x, void* y, int B, int C, int H, int W,
                              int cpad, void* stream);

/* ResizeX(images, s) (network/mynn.py:101-114, called at network/ocrnet.py:276)
 * fused with the layout change: NCHW fp32 [B,C,Hi,Wi] -> NHWC bf16
 * [B,Ho,Wo,cpad]; Ho==Hi && Wo==Wi is an exact copy.                          */
int ssa_image_resize_to_nhwc_bf16(const float* x, int B, int C, int Hi, int Wi,
                                  void* y, int Ho, int Wo, int cpad, void* stream);

/* -------------------------------------------------------------- bilinear ----
 * F.interpolate(mode='bilinear', align_corners=False): network/mynn.py:42-114,
 * network/hrnetv2.py:246-249,440-445 (K8).  in_dtype/out_dtype: 0 bf16, 1 f32.
 * Backward is a deterministic gather (no atomics).                             */
int ssa_bilinear_fwd(const void* x, int in_dtype, int B, int Hi, int Wi, int C,
                     int ldx, void* y, int out_dtype, int Ho, int Wo, int ldy,
                     void* stream);
int ssa_bilinear_bwd(const void* dy, int dy_dtype, int B, int Ho, int Wo, int C,
                     int lddy, void* dx, int dx_dtype, int Hi, int Wi, int lddx,
                     void* stream);
/* The same backward for an UPSAMPLING resize (Ho >= 2 Hi), separable (bilinear weights factor): pass X sums over the
 * output columns into tmp ([B, Ho, Wi, C] fp32, dense, 16-byte aligned), pass Y over the output rows -- window_x +
 * window_y taps per element instead of their product, dy read once.  Two dependent launches: never both in one
 * ssa_group bracket.  network/mynn.py:42-114 (Upsample / scale_as), network/hrnetv2.py:246-249,440-445.          */
int ssa_bilinear_bwd_x(const void* dy, int dy_dtype, int B, int Ho, int Wo, int C, int lddy, float* tmp, int Wi,
                       void* stream);
int ssa_bilinear_bwd_y(const float* tmp, int B, int Ho, int Wi, int C, void* dx, int dx_dtype, int Hi, int lddx,
                       void* stream);

/* ----------------------------------------------------------------- pooling ----
 * DeepLabV3+/ResNet-50 (BASELINE configs[0]): nn.MaxPool2d(3, 2, 1) of the ResNet
 * stem (network/Resnet.py:147) and nn.AdaptiveAvgPool2d(1) of the ASPP image
 * branch (network/utils.py:201).  NHWC bf16, C % 8 == 0.  idx: winning tap per
 * output element (uint8, PyTorch's first-maximum rule), consumed by the backward. */
int ssa_maxpool3x3s2_fwd(const void* x, int ldx, int B, int H, int W, int C, void* y,
                         unsigned char* idx, int Ho, int Wo, void* stream);
int ssa_maxpool3x3s2_bwd(const void* dy, const unsigned char* idx, int B, int Ho, int Wo,
                         int C, void* dx, int H, int W, void* stream);
/* out[b,c] = mean_p x[b,p,c];  dx[b,p,c] = dout[b,c] / HW                        */
int ssa_global_avg_pool_fwd(const void* x, int ldx, int B, long HW, int C, void* out,
                            void* stream);
int ssa_global_avg_pool_bwd(const void* dout, int B, long HW, int C, void* dx,
                            void* stream);

/* Label-map resize, mask.resize(size, Image.NEAREST) of
 * transforms/joint_transforms.py:193,267,290,319,339,364,466:
 * uint8 [B,Hs,Ws] -> [B,Hd,Wd], dst[y,x] = src[iy_table[y], ix_table[x]].  The
 * index tables carry Pillow's exact rule (a running double-precision sum, see
 * semseg_amd/datasets/transforms.py); the result is bit-identical to PIL.      */
int ssa_resize_nearest_u8(const unsigned char* src, int B, int Hs, int Ws,
                          unsigned char* dst, int Hd, int Wd, const int* iy_table,
                          const int* ix_table, void* stream);

/* ------------------------------------------------------------------- OCR ----
 * SpatialGather_Module.forward, network/ocr_utils.py:34-46 (K9) and
 * ObjectAttentionBlock.forward, network/ocr_utils.py:95-119 (K10).
 * The matrix products (probs^T @ feats, q @ k^T, sim @ v and their gradients)
 * run on ssa_conv2d_igemm / ssa_conv2d_wgrad as 1x1 GEMMs whose "filters" are
 * activations packed by ssa_pack_matrix; the functions below are the two
 * softmaxes and their backward.
 *
 * softmax over HW, per (image, class): logits fp32 [HW, K] (pixel stride ld);
 * rowstat fp32 [K][2] = (max, sum exp).                                       */
int ssa_softmax_hw_stats(const float* logits, int ld, long HW, int K,
                         float* rowstat, void* stream);
/* probs bf16 [HW, Kpad] (zero padded columns K..Kpad)                         */
int ssa_softmax_hw_probs(const float* logits, int ld, long HW, int K,
                         const float* rowstat, void* probs, int Kpad, void* stream);
/* out[k] = sum_c a[k,c]*b[k,c]                                                */
int ssa_rowdot_f32(const float* a, const float* b, int K, int C, float* out,
                   void* stream);
/* dlogits[p,k] (+)= probs[p,k] * (dprobs[p,k] - dot[k])                        */
int ssa_softmax_hw_bwd(const float* logits, int ld, long HW, int K,
                       const float* rowstat, const float* dprobs, int lddp,
                       const float* dot, float* dlogits, int lddl, int accumulate,
                       void* stream);
/* per-pixel softmax over K object regions of scale*sim: sim fp32 [P, K];
 * probs bf16 [P, Kpad]; backward returns dsim as bf16 [P, Kpad].              */
int ssa_softmax_lastdim_fwd(const float* sim, int ld, long P, int K, float scale,
                            void* probs, int Kpad, void* stream);
int ssa_softmax_lastdim_bwd(const float* sim, int ld, long P, int K, float scale,
                            const float* dprobs, int lddp, void* dsim, int Kpad,
                            void* stream);
/* GEMM operand from an activation matrix: dst bf16 [rows_out][Kpad] = src
 * (or src^T), zero padded.  src_dtype 0 bf16, 1 fp32; src is [R][C], row
 * stride ld.                                                                  */
int ssa_pack_matrix(const void* src, int src_dtype, int R, int C, int ld,
                    int transpose, void* dst, int rows_out, int Kpad, void* stream);

/* Fused object attention (network/ocr_utils.py:100-113; csrc/ocr_attn.hip): out[p, :] = softmax_K(scale * q[p, :] k^T) v
 * per image, one launch, sim / probs never leave the registers.  q: [P] pixel rows of Dch = 256 16-bit channels with
 * pixel stride ldq, k / v: dense [K][256] 16-bit, out: [P] rows with stride ldo.  K <= 96 object regions
 * (ssa_ocr_attn_supported; otherwise the caller runs the three-launch form on ssa_conv2d_igemm +
 * ssa_softmax_lastdim_*).  Group-aware (the scale passes of a step share one launch).
 * Backward: dq, and the two [P][Kpad] 16-bit matrices (Kpad = K rounded up to 32, padding = 0) the pixel reductions
 * dv = probs^T dout and dk = dsim^T q are computed from (ssa_conv2d_wgrad with KH = KW = 1):
 * probs = the forward's probabilities, dsim = scale * probs * (dprobs - <probs, dprobs>), dprobs = dout v^T.          */
int ssa_ocr_attn_supported(int K, int Dch);
int ssa_ocr_attn_fwd(const void* q, int ldq, const void* k, const void* v, long P, int K, int Dch, float scale,
                     void* out, int ldo, void* stream);
int ssa_ocr_attn_bwd(const void* q, int ldq, const void* k, const void* v, const void* dout, int lddo, long P, int K,
                     int Dch, float scale, void* dq, int lddq, void* probs, void* dsim, void* stream);

/* ----------------------------------------------------- scale attention -----
 * sigmoid of the attention logit (network/utils.py:363) and the two-scale
 * fusion of MscaleOCR.two_scale_forward, network/ocrnet.py:289-298 (K11):
 *   joint = up(attn*p_lo) + (1 - up(attn)) * p_hi   (all fp32 NHWC)           */
int ssa_sigmoid_fwd(const float* x, float* y, long n, void* stream);
int ssa_sigmoid_bwd(const float* y, const float* dy, float* dx, long n, void* stream);
/* out[p,c] = a[p]*x[p,c]  (attn broadcast over classes), and its backward.   */
int ssa_bcast_mul_fwd(const float* a, const float* x, float* out, long P, int C,
                      void* stream);
int ssa_bcast_mul_bwd(const float* a, const float* x, const float* dout, float* da,
                      float* dx, long P, int C, void* stream);
/* joint[p,c] = lo[p,c] + (1-a[p])*hi[p,c], and its backward.                  */
int ssa_attn_blend_fwd(const float* lo, const float* a, const float* hi,
                       float* joint, long P, int C, void* stream);
int ssa_attn_blend_bwd(const float* a, const float* hi, const float* djoint,
                       float* da, float* dhi, long P, int C, int accumulate_da,
                       void* stream);

/* ---------------------------------------------------------------- losses ----
 * CrossEntropyLoss2d, loss/utils.py:121-134 (K13): mean over valid pixels of
 * -log_softmax(x)[t], ignore_index skipped.  logits fp32 NHWC [P,C], labels
 * int64 [P].  acc = {sum nll, valid count} fp64; dlogits (optional) receives
 * softmax - onehot for valid pixels (un-normalised; scale in ssa_scale_grad).  */
int ssa_ce_fwd(const float* logits, int ld, const int64_t* labels, long P, int C,
               int ignore_index, double* acc, float* dlogits, void* stream);
/* RMILoss.forward_sigmoid part I, loss/rmi.py:91-116 (K14): masked BCE with
 * logits, sum over pixels and classes; acc = {sum bce, valid count}.
 * dlogits = (sigmoid(x) - onehot) * mask (un-normalised).                     */
int ssa_bce_fwd(const float* logits, int ld, const int64_t* labels, long P, int C,
                double* acc, float* dlogits, void* stream);
/* loss = acc[0] / (acc[1] + denom_add) ; writes fp32 scalar                    */
int ssa_loss_finalize(const double* acc, double denom_add, float* loss, void* stream);
/* g[i] *= upstream[0] * coef / (acc[1] + denom_add)                            */
/* Backward of ssa_bce_fwd WITHOUT its saved gradient: dlogits = (sigmoid - onehot) * mask * upstream * coef /
 * (acc[1] + denom_add), recomputed from the logits (dense [P, C], ld == C, P * C % 4 == 0, 16-byte aligned; otherwise
 * SSA_EUNSUPPORTED and the caller scales the gradient ssa_bce_fwd saved).  loss/rmi.py:103-112's autograd backward. */
int ssa_bce_bwd(const float* logits, int ld, const int64_t* labels, long P, int C, const float* upstream,
                double coef, const double* acc, double denom_add, float* dlogits, void* stream);
int ssa_scale_grad(float* g, long n, const float* upstream, double coef,
                   const double* acc, double denom_add, void* stream);
/* dst[i] = src[i] * upstream[0] * coef / (acc[1] + denom_add): the same without touching src (the un-normalised
 * gradient the forward saved stays valid; the backward needs no copy of it)       */
int ssa_scale_grad_to(const float* src, float* dst, long n, const float* upstream, double coef,
                      const double* acc, double denom_add, void* stream);

/* RMILoss.rmi_lower_bound, loss/rmi.py:139-215 + loss/rmi_utils.py:15-56,
 * 95-107 (K15).  Fused: sigmoid*mask+1e-6 -> 4x4/4 avg pool (pad 2) ->
 * 3x3-neighbourhood Gram matrices in fp64 (never materialising the
 * [B,C,9,65025] stack) -> 9x9 inverse / Cholesky per (b,c) in one wavefront. */
int ssa_rmi_pool(const float* logits, int ld, const int64_t* labels, int B, int H,
                 int W, int C, float* pooled_pr, float* pooled_la, int Hp, int Wp,
                 void* stream);
int ssa_rmi_gram(const float* pooled_pr, const float* pooled_la, int BC, int Hp,
                 int Wp, double* gram /*[BC][189]*/, void* stream);
/* per (b,c): loss value + gradient matrices G_lp, G_pp (9x9) + means.         */
int ssa_rmi_solve(const double* gram, int BC, int Hp, int Wp, double* loss_bc,
                  double* gmat /*[BC][2*81+18]*/, void* stream);
/* rmi = sum_c mean_b loss_bc / 9  -> fp32 scalar                              */
int ssa_rmi_finalize(const double* loss_bc, int B, int C, float* out, void* stream);
/* d rmi / d pooled_pr  [BC][Hp][Wp] fp32                                      */
int ssa_rmi_bwd_pooled(const float* pooled_pr, const float* pooled_la,
                       const double* gmat, int BC, int Hp, int Wp, float* dpooled,
                       void* stream);
/* dlogits[p,c] += coef*upstream * dpooled[cell(p)]/16 * mask * s*(1-s)        */
int ssa_rmi_bwd_logits(const float* logits, int ld, const int64_t* labels, int B,
                       int H, int W, int C, const float* dpooled, int Hp, int Wp,
                       const float* upstream, double coef, float* dlogits,
                       int accumulate, void* stream);
/* The same with the BCE half of RMILoss.forward_sigmoid's gradient (loss/rmi.py:103-134: 0.5 * bce + 0.5 * rmi) folded in:
 * dlogits = bce_grad * upstream * bce_coef / (bce_acc[1] + bce_denom_add) + the RMI term -- what ssa_scale_grad_to
 * followed by the accumulating form computes, in one pass over the logits' gradient.  bce_grad NULL: the BCE half is
 * recomputed from the logits (the forward then saves no gradient).                                                */
int ssa_rmi_bwd_logits_bce(const float* logits, int ld, const int64_t* labels, int B, int H, int W, int C,
                           const float* dpooled, int Hp, int Wp, const float* upstream, double coef,
                           const float* bce_grad, double bce_coef, const double* bce_acc, double bce_denom_add,
                           float* dlogits, void* stream);

/* Tail of the input pipeline on the device: joint crop window + optional
 * horizontal flip of the cropped pair (transforms/joint_transforms.py:276-281
 * RandomHorizontallyFlip after the crop transforms), then for the image ToTensor + Normalize
 * (datasets/base_loader.py:141-142; mean/std of config.py:96-97) written as the NHWC bf16
 * [ch][cw][cpad] tensor the trunk reads (channels 3.. zero), value
 * bf16(((float)u8/255 - mean[c]) / std[c]) in IEEE fp32 -- bit-identical to the CPU transforms
 * followed by .to(bfloat16); for the labels MaskToTensor (uint8 -> int64).
 * img_hwc: uint8 [H][W][3] RGB on the device; lab_hw: uint8 [H][W].  mean3/std3: HOST floats.  */
int ssa_image_u8_crop_flip_normalize(const unsigned char* img_hwc, int H, int W, int x0, int y0,
                                     int cw, int ch, int flip, const float* mean3,
                                     const float* std3, void* out_nhwc_bf16, int cpad,
                                     void* stream);
int ssa_label_u8_crop_flip(const unsigned char* lab_hw, int H, int W, int x0, int y0, int cw,
                           int ch, int flip, int64_t* out, void* stream);

/* The scale step of the input pipeline on the device:
 * `img.resize((w, h), Image.BICUBIC)` of transforms/joint_transforms.py:433-471 (RandomSizeAndCrop ->
 * scale_and_crop; also the Scale / ResizeHeight transforms) is Pillow's two-pass 8-bit resampling
 * (libImaging/Resample.c): a horizontal pass into an 8-bit image, then a vertical pass over it.  One
 * call = one pass over an interleaved uint8 [Hs][Ws][C] device image along axis 1 (x: n_out output
 * columns) or 0 (y: n_out output rows); bounds [n_out][2] = (first source index, tap count) and coefs
 * [n_out][ksize] = 22-bit fixed-point taps are DEVICE arrays the host derives exactly as
 * precompute_coeffs / normalize_coeffs_8bpc do (semseg_amd/datasets/transforms.py).  Integer
 * arithmetic: bit-identical to Pillow.                                                          */
int ssa_resample_u8(const unsigned char* src, int Hs, int Ws, int C, int axis, unsigned char* dst,
                    int n_out, const int* bounds, const int* coefs, int ksize, void* stream);

/* Evaluation tail on the device (utils/trnval_utils.py:173-196 + utils/misc.py:50-67
 * fast_hist): pred[p] = first argmax_c logits[p,c] (uint8, optional) and
 * hist[gt*C + pred] += 1 for 0 <= gt < C (int64 [C*C], ACCUMULATED: clear it once per
 * evaluation).  logits fp32 NHWC [P,C] (pixel stride ld), labels int64 [P].          */
int ssa_confusion_matrix(const float* logits, int ld, const int64_t* labels, long P, int C,
                         unsigned char* pred_out, int64_t* hist, void* stream);

/* Optimizer step (train.py:509 on the torch.optim.SGD of loss/optimizer.py:47-53;
 * SURVEY.md 8f rank 3): for every tensor i, elementwise in fp32
 *     d = g + weight_decay*p;  buf = momentum*buf + d;  p -= lr * (nesterov ? d + momentum*buf : buf)
 * (torch's SGD with dampening 0; a momentum buffer starting at zero reproduces its
 * first step).  params/grads/bufs/numel are HOST arrays of n_tensors entries holding
 * device pointers to dense fp32 tensors; bufs may be NULL (momentum 0) and so may
 * single entries of it.  Up to 96 tensors go into one launch (pointers travel as
 * kernel arguments, so a captured graph owns them).  lr_dev, when not NULL, is a
 * device float read at run time instead of `lr` -- a captured step then follows the
 * LR schedule without re-capture.
 * amp_state (optional): the loss-scaling record of fp16 training, 4 device floats {scale, found_inf, clean steps,
 * 1 / scale}: every gradient is multiplied by amp_state[3] on the way in, and when amp_state[1] != 0 (an inf / nan
 * was found by ssa_amp_check_grads) the call changes NOTHING -- apex.amp's skipped step (train.py:503-505).          */
int ssa_sgd_momentum_step(void* const* params, const void* const* grads, void* const* bufs,
                          const int64_t* numel, int n_tensors, float lr, const float* lr_dev,
                          float momentum, float weight_decay, int nesterov, const float* amp_state,
                          void* stream);
/* Dynamic loss scaling (apex.amp's LossScaler, the reference's --fp16 path: train.py:380-381,503-505), capturable:
 * ssa_amp_check_grads sets amp_state[1] = 1 if any element of the n_tensors dense fp32 gradient tensors (HOST arrays
 * of device pointers / element counts, as above) is inf or nan;  ssa_amp_update then closes the step:
 * found_inf: scale = max(scale * backoff, min_scale), clean steps = 0; else clean steps += 1 and, on reaching
 * growth_interval, scale = min(scale * growth, max_scale), clean steps = 0; found_inf = 0; amp_state[3] = 1 / scale. */
int ssa_amp_check_grads(const void* const* grads, const int64_t* numel, int n_tensors, float* amp_state,
                        void* stream);
int ssa_amp_update(float* amp_state, int growth_interval, float growth, float backoff, float min_scale,
                   float max_scale, void* stream);
/* The same update, additionally counting what apex LOGS ("Gradient overflow.  Skipping step", apex/amp/scaler.py):
 * counters[0] += 1 and counters[1] += 1 on a skipped step, counters[1] = 0 on a clean one -- skipped steps in total and in
 * a row (2 floats of device memory, or NULL).  A run whose gradients are genuinely nan pins the scale at min_scale and
 * skips every step: the host reads this record (semseg_amd.amp.LossScaler.state_dict / .health) to say so. */
int ssa_amp_update_counted(float* amp_state, float* counters, int growth_interval, float growth, float backoff,
                           float min_scale, float max_scale, void* stream);

/* fp32 elementwise out = a (+ | * | /) b (op 0 | 1 | 2) and its backward (da, db optional): the
 * attention normalisation and the attention-weighted sum over scales of the attention-to-scale
 * heads, network/attnscale.py:153-166,330-352.                                   */
int ssa_ewise_f32(int op, const float* a, const float* b, float* out, long n, void* stream);
int ssa_ewise_bwd_f32(int op, const float* a, const float* b, const float* dout, float* da,
                      float* db, long n, void* stream);

/* axpy on fp32: y = alpha*x + (accumulate? y : 0) */
int ssa_axpy_f32(const float* x, float alpha, float* y, long n, int accumulate,
                 void* stream);

/* ---------------------------------------------------------------- probes ----
 * Hardware-layout probes used by tests/test_probe_gpu.py (MFMA fragment and
 * ds_read_b64_tr_b16 lane maps are verified on the device, not assumed).      */
int ssa_probe_mfma32(const void* a, const void* b, float* c, void* stream);
int ssa_probe_tr16(unsigned short* out, int mode, void* stream);
/* c[16][16] = a[16][32] * b[32][16] (b given transposed: [16 n][32 k]) through v_mfma_f32_16x16x32; out[64][2] =
 * (a, b) of every lane after v_permlane16_swap of a = lane, b = lane + 100.                                  */
int ssa_probe_mfma16(const void* a, const void* bt, float* c, void* stream);
int ssa_probe_swap16(unsigned* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEMSEG_HIP_H */
