/*
 * vcvits_hip.h -- C ABI of libvcvits_hip.so, the MI355X (gfx950) kernel library behind the
 * vcvits hot path.
 *
 * The reference (vtuber-plan/vcvits) has no FFI: its hot path is plain PyTorch calls
 * (F.conv1d / F.conv2d / F.conv_transpose1d / torch.stft / torch.matmul / softmax / layer_norm).
 * Each entry point below replaces one of those call sites; the reference file:line it stands
 * for is cited next to it.  The Python mirror of the reference modules (the vcvits_amd package) binds
 * these symbols through ctypes (vcvits_amd/_lib.py) -- see INTEGRATION.md for the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's allocator); the library
 *     allocates nothing persistent and never synchronises;
 *   - `stream` is the caller's HIP stream (hipStream_t passed as void*); all launches are
 *     asynchronous on it;
 *   - return value: 0 on success, negative VCV_E* otherwise (never aborts);
 *   - tensors are dense fp32, layout [B, C, T] (or [B, C, H, P] for the period discriminators,
 *     which the kernels address as rows of P contiguous columns).
 */
#ifndef VCVITS_HIP_H
#define VCVITS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VCV_OK 0
#define VCV_EINVAL (-1)  /* bad shape / unsupported configuration */
#define VCV_EHIP (-2)    /* a HIP runtime error was raised by the launch */
#define VCV_ELDS (-3)    /* tile does not fit the 160 KiB LDS */

/* activation / transform selectors */
#define VCV_ACT_NONE 0
#define VCV_ACT_LEAKY 1    /* x > 0 ? x : slope*x */
#define VCV_ACT_RELU 2
#define VCV_ACT_TANH 3
#define VCV_ACT_LOGCLAMP 4 /* log(max(x, clamp)) with clamp passed in `slope` */

#define VCV_TF_NONE 0
#define VCV_TF_LEAKY 1     /* operand := leaky(operand) on load */
#define VCV_TF_DLEAKY 2    /* operand := operand * leaky'(aux) on load (aux > 0 ? 1 : slope) */
#define VCV_TF_DRELU 3     /* operand := operand * (aux > 0) */
#define VCV_TF_DTANH 4     /* operand := operand * (1 - aux^2)   (aux = tanh output) */
#define VCV_TF_DLOGCLAMP 5 /* operand := aux > log(clamp) ? operand * exp(-aux) : 0  (aux = logclamp output,
                              clamp passed in `slope`) */

/*
 * Generic implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
 *
 *   y[b, g*Mg + m, q*os + oo, p]  (+)=  act( alpha * sum_{c<Cg, j<J} A(m, c, kw0 + j*kws) *
 *                                   tf(x[b, g*Cg + c, q*s + j*dj + off, p]) + bias[g*Mg+m] ) ...
 *
 * with rows outside [0, Tin) read as zero and rows outside [0, Tout) not written.  One
 * parametrisation covers (reference call sites):
 *   - Conv1d forward, any stride / dilation / groups   (modules.py:126-143,190-201;
 *     discriminator.py:53-61; relative_attention_transformer.py:121-124,283-284)
 *   - Conv2d (k,1)/(s,1) of the period discriminators, P = period  (discriminator.py:18-25)
 *   - ConvTranspose1d forward, one launch phase per output residue (HiFi-GAN `ups`,
 *     synthesizer_svc.py:59 / synthesizer_tts.py:71-78)
 *   - the data gradients of all of the above (torch autograd in the reference).
 * Weight addressing: a_mode 0: A(m,c,k) = w[((g*Mg + m)*Cg + c)*K + k]   (Conv weight, forward)
 *                    a_mode 1: A(m,c,k) = w[((g*Cg + c)*Mg + m)*K + k]   (transposed use)
 * Epilogue order: v = alpha*acc + bias[m]; v = act(v); v *= dact(oaux) (out_tf); v += res;
 *                 v *= mask[b, row]; if (accumulate) v += y;  y = v.
 */
typedef struct VcvConvArgs {
  const float* x;     /* [B, G*Cg, Tin, P] */
  const float* w;     /* see a_mode */
  const float* bias;  /* [G*Mg] or NULL */
  const float* res;   /* like y, or NULL */
  const float* mask;  /* [B, Tout] or NULL */
  const float* xaux;  /* like x, for in_tf == DLEAKY/DRELU, else NULL */
  const float* oaux;  /* like y, for out_tf != NONE, else NULL */
  float* y;           /* [B, G*Mg, Tout, P] */
  int32_t B, G, Cg, Mg;
  int32_t Tin, Tout, P;
  int32_t K;          /* taps stored per (m,c) pair in w */
  int32_t s, dj, off; /* input row = q*s + j*dj + off */
  int32_t os, oo;     /* output row = q*os + oo (phase r adds r to oo when phases > 1) */
  int32_t phases;     /* 1: J = K, kw0 = 0, kws = 1.  >1: phase r uses taps kw = r + j*phases */
  int32_t Q;          /* number of q positions per phase */
  int32_t a_mode, in_tf, out_act, out_tf, accumulate;
  float alpha, slope;
  int32_t io;         /* storage type of the activations in HBM: 0 = all fp32 (every entry point); else VCV_IO_* bits: `x`,
                         `y` and `res` are 16-bit tensors (same [B, C, T(, P)] layout, rows of an even number of elements):
                         vcv_conv_bf16io_* only -- every other entry point returns VCV_EINVAL for io != 0 */
  int32_t ms;         /* 0 / 1: none.  ms > 1 (vcv_conv_bf16io_* only): a ConvTranspose1d of stride ms with ALL its output phases as
                         rows of ONE launch -- Mg = Cout * ms rows ordered (cout, phase), K = taps per phase (kernel / ms),
                         a_mode 1, phases 1, s 1, dj -1; output row (cout, r), column q goes to y[b, cout, q * ms + r + oo]
                         (oo = -padding): one staged input span feeds every phase, and a lane's four accumulator rows of
                         one channel are four consecutive samples -- 8-byte stores, contiguous per wave -- instead of the
                         phased launch's 2-byte stores `ms` elements apart.  `w` is the ConvTranspose weight [Cin, Cout, ms*K] */
  float post_scale;   /* 0: none.  Else the epilogue becomes v = (act(alpha*acc + bias) * dact + res) * mask * post_scale
                         (+ y if accumulate): the mean over the three ResBlocks of a generator stage is accumulated by the
                         blocks' last convs (post_scale 1/3) instead of by a pass over three stored outputs
                         (vcv_conv_bf16io_* only) */
} VcvConvArgs;
#define VCV_IO_BF16 3

int vcv_conv_gemm(const VcvConvArgs* args, void* stream);

/*
 * LDS-DMA variant of vcv_conv_gemm for forward-type launches (a_mode 0, G == 1, phases <= 1, in_tf in
 * {NONE, LEAKY}): weights are packed per launch into LDS-image order in `workspace`
 * (vcv_conv_dma_workspace floats, 0 = launch not eligible -> use vcv_conv_gemm) and both operands are
 * staged by global/buffer loads that write LDS directly (double-buffered).  flip = 1: `w` is the
 * forward weight [C, M, K] of a conv whose stride-1 DATA GRADIENT this launch computes (the pack
 * flips / transposes it).  Same epilogue semantics as vcv_conv_gemm.
 */
int64_t vcv_conv_dma_workspace(const VcvConvArgs* args);
int vcv_conv_dma(const VcvConvArgs* args, float* workspace, int flip, void* stream);
/* Split form for callers that keep packed weights across launches (the discriminator weights are the same
 * in the generator step and the discriminator step of one batch).  vcv_conv_dma_plan: out[0] = floats of the
 * packed-weight buffer, out[1] = floats of per-launch scratch, out[2] = signature of the pack layout; a packed
 * buffer may be reused by a later launch over the same unchanged weights with the same out[0] and out[2].
 * vcv_conv_dma_run: as vcv_conv_dma with the two buffers separate; pack_valid != 0 skips the pack pass. */
int vcv_conv_dma_plan(const VcvConvArgs* args, int flip, int64_t* out);
int vcv_conv_dma_run(const VcvConvArgs* args, float* pack_ws, float* scratch_ws, int flip, int pack_valid,
                     void* stream);

/*
 * bf16-operand variant of vcv_conv_dma_plan / vcv_conv_dma_run (same launch family, same argument meaning, same
 * epilogue): operands are rounded to bf16 on their way into v_mfma_f32_32x32x16_bf16, sums are fp32, `x` / `y` and
 * every epilogue operand stay fp32 in HBM -- the reference's AMP recipe (configs/base.json:18 `fp16_run`,
 * train.py:104-106) with bf16 in place of fp16.  The packed-weight buffer holds bf16 slabs; out[0] of the plan is
 * its size in 4-byte words.  A pack made by one family is never valid for the other (different out[2]).
 */
int vcv_conv_bf16_plan(const VcvConvArgs* args, int flip, int64_t* out);
/* the same kernel on fp32 elements (4-channel 16-byte LDS groups, four v_mfma_f32_32x32x2_f32 per fragment pair: exact
 * fp32 like vcv_conv_dma_*, with a quarter of its LDS read instructions per MFMA) */
int vcv_conv_pk_plan(const VcvConvArgs* args, int flip, int64_t* out);
int vcv_conv_pk_run(const VcvConvArgs* args, float* pack_ws, float* scratch_ws, int flip, int pack_valid, void* stream);
int vcv_conv_bf16_run(const VcvConvArgs* args, float* pack_ws, float* scratch_ws, int flip, int pack_valid,
                      void* stream);
/*
 * The bf16 kernel with 16-bit ACTIVATIONS in HBM (args->io != 0; conv_pk_io*.hip): the conv <-> conv tensors of the
 * HiFi-GAN decoder in inference (synthesizer_svc.py:108 under the reference's fp16 autocast, train.py:104-106: AMP stores
 * conv activations in half precision).  io = VCV_IO_X16 | VCV_IO_Y16 [| VCV_IO_XF16] [| VCV_IO_YF16]: `x`, and `y` with
 * `res` and the accumulate target, are 16-bit tensors -- bf16, or IEEE fp16 with the *F16 bit (values clamped to the
 * finite fp16 range).  `x` is read with 16-byte loads of eight positions (rows of an even number of elements) and reaches
 * the matrix cores as bf16 (a bf16 `x` without input transform: as stored); the epilogue runs in fp32 and rounds once
 * (nearest even); bias / mask stay fp32; out_tf must be NONE.  Combinations built: 3 (all bf16), 7 (fp16 x -> bf16 y), 11
 * (bf16 x -> fp16 y / res), 15 (all fp16).  Packs are interchangeable with vcv_conv_bf16_* wherever the plans agree
 * (out[2]); out[1] is always 0 (no split reduction).
 * vcv_cast_f32_x16 / vcv_cast_x16_f32: the conversions at the two ends of such a chain (kind 1 = bf16, 2 = fp16).
 * vcv_conv_m1_x16_fwd: vcv_conv_m1_fwd (one output channel, stride 1, dilation 1, K in {3, 5, 7}) over a 16-bit `x`.
 */
#define VCV_IO_X16 1
#define VCV_IO_Y16 2
#define VCV_IO_XF16 4
#define VCV_IO_YF16 8
int vcv_conv_bf16io_plan(const VcvConvArgs* args, int flip, int64_t* out);
int vcv_conv_bf16io_run(const VcvConvArgs* args, float* pack_ws, float* scratch_ws, int flip, int pack_valid,
                        void* stream);
int vcv_cast_f32_x16(const float* x, void* y, int64_t n, int kind, void* stream);
int vcv_cast_x16_f32(const void* x, float* y, int64_t n, int kind, void* stream);
int vcv_conv_m1_x16_fwd(const void* x, int kind, const float* w, const float* bias, float* y, int B, int C, int Tin,
                        int Tout, int K, int dil, int pad, int in_leaky, int out_act, float slope, void* stream);
/*
 * fp32 convolutions on the bf16 matrix pipe by exact operand splitting (conv_x3.hip): every fp32 operand is the exact sum
 * of three bf16 terms, so an fp32 product is the sum of nine exact bf16 products accumulated in fp32 -- the arithmetic
 * of the fp32 kernels above (torch.nn.functional.conv1d / conv2d / conv_transpose1d in fp32: vits/model/modules.py,
 * discriminators/discriminator.py, the hub generator) up to summation order, at the bf16 MFMA rate.  Same launch family,
 * argument meaning and epilogue as vcv_conv_pk_plan / vcv_conv_pk_run; `x`, `y` and all epilogue operands are fp32.
 * vcv_conv_x3_set_terms(6 | 9): 6 (default) = the three smallest products (each < 2^-24 of the fp32 product) left out,
 * 9 = all products.
 * vcv_conv_x3_set_all(1): take every eligible launch; 0 (default): only the shapes where the split kernel is the faster one.
 */
/* Batched weight packs (conv_pack.hip): the layout of the pack a launch (args, flip) reads, as a job; vcv_pack_many packs
 * any number of jobs (w = the weights, wp = a buffer of the size the matching *_plan returned) in ONE launch.  The host
 * records the jobs of a module tree once and replays them whenever the tree's weights were re-normalised. */
typedef struct VcvPackJob {
  const float* w;   /* weights ([M, C, K], or [C, M, K] for mode 1 / 2) */
  void* wp;         /* packed buffer */
  int32_t kind;     /* 0: vcv_conv_x3_*, 1: vcv_conv_pk_*, 2: vcv_conv_bf16_* */
  int32_t M, C, K, BM, BKC, JA, nch, nmt, phases, mode, reserved;
  int64_t total;    /* pack threads */
  int64_t block0;   /* (filled by vcv_pack_many) */
} VcvPackJob;
int vcv_conv_x3_pack_job(const VcvConvArgs* args, int flip, VcvPackJob* out);
int vcv_conv_pk_pack_job(const VcvConvArgs* args, int flip, VcvPackJob* out);
int vcv_conv_bf16_pack_job(const VcvConvArgs* args, int flip, VcvPackJob* out);
int vcv_pack_many(VcvPackJob* jobs, int n, void* table_dev, void* stream);
/* The same launches without the host -> device copy of the job table: `jobs` is finalised in place (order, block0) and the
 * caller copies it into `table_dev` itself before the launches first execute (launch sequences recorded into a HIP graph:
 * no copy node that re-reads host memory at every replay). */
int vcv_pack_many_prepared(VcvPackJob* jobs, int n, void* table_dev, void* stream);
/* Host table -> device on `stream`; inside a stream capture the copy is replayed from `src`, which the caller keeps alive
 * and unchanged for the life of the graph (the python layer's device tables of a captured pass). */
int vcv_upload_table(void* dst, const void* src, int64_t bytes, void* stream);
int vcv_conv_x3_plan(const VcvConvArgs* args, int flip, int64_t* out);
int vcv_conv_x3_run(const VcvConvArgs* args, float* pack_ws, float* scratch_ws, int flip, int pack_valid, void* stream);
int vcv_conv_x3_set_terms(int n);
int vcv_conv_x3_get_terms(void);
int vcv_conv_x3_set_all(int all);
/* tuning probe: fix the tile variant (0..6; -1 = the library's choice), taps per stage (1 | 2) and channel-group split (>= 2 | -1) */
int vcv_conv_x3_set_variant(int variant, int js, int ks);

/*
 * Weight gradient of the same family (torch autograd of the call sites above):
 *   dw[(g*Mg + m), c, k] (+)= alpha * sum_{b, q, p} tfa(dy[b, g*Mg+m, q, p]) *
 *                                               tfb(x[b, g*Cg+c, q*s + k*dj + off, p])
 * `a` is the un-shifted operand (dy for Conv, x for ConvTranspose), `b` the shifted one.
 * dw is [G*Mg, Cg, K] in the `a`-major order; the reduction over (b, q) is split over
 * workgroups and combined with fp32 atomics, so dw must be zeroed (or hold the value to
 * accumulate onto) before the call.
 */
typedef struct VcvWgradArgs {
  const float* a;     /* [B, G*Mg, Ta, P] */
  const float* b;     /* [B, G*Cg, Tb, P] */
  const float* aaux;  /* like a, for a_tf, or NULL */
  const float* baux;  /* like b, for b_tf, or NULL */
  float* dw;          /* [G*Mg, Cg, K] */
  int32_t B, G, Cg, Mg;
  int32_t Ta, Tb, P;
  int32_t K;
  int32_t s, dj, off;
  int32_t a_tf, b_tf;
  int32_t transpose_out; /* 1: write dw[(g*Cg + c), m, k] instead (Conv weight from ConvT roles) */
  float alpha, slope;
  float* dbias;       /* [G*Mg] or NULL: db[m] += sum_{b, q, p} a[b, m, q, p] -- the bias gradient of a Conv, whose
                         un-shifted operand IS dy: the row sums are collected while the kernel stages `a` (needs a_tf ==
                         NONE); replaces a separate vcv_bias_grad pass over dy */
  float* slab;        /* NULL: the workgroups that split the (batch, position) reduction combine with fp32 atomics
                         (summation order varies from run to run).  Non-NULL: each writes its partial tile to its own
                         slab of Mg*Cg*K floats and a finishing pass adds the slabs in a fixed order -- bit-reproducible */
  int64_t slab_floats; /* capacity of `slab`; the split is limited to slab_floats / (Mg*Cg*K) ways */
} VcvWgradArgs;

int vcv_conv_wgrad(const VcvWgradArgs* args, void* stream);
/* 1: vcv_conv_wgrad runs this launch on the LDS-DMA weight-gradient kernel (wgrad_dma.hip); 0: on the register-staged one
 * (grouped launches, derivative-masked operands, rows under 64 positions, (C, K) blocks under 96 columns). */
int vcv_conv_wgrad_takes_dma(const VcvWgradArgs* args);

/*
 * bf16-operand weight gradient (same VcvWgradArgs, G == 1, transforms NONE / LEAKY): operands rounded to bf16 on their
 * way into v_mfma_f32_32x32x16_bf16, fp32 sums.  The reduction over (batch, positions) is split over workgroups that
 * each write a partial tile to `scratch`; a finishing pass adds the partials in a fixed order onto dw -- no atomics,
 * bit-reproducible.  vcv_wgrad_bf16_scratch: floats of scratch the launch wants (0 = not eligible: use
 * vcv_conv_wgrad); any scratch of at least Mg*Cg*K floats is accepted (less scratch = coarser split).
 */
int64_t vcv_wgrad_bf16_scratch(const VcvWgradArgs* args);
int vcv_wgrad_bf16(const VcvWgradArgs* args, float* scratch, int64_t scratch_floats, void* stream);
/* weight gradient in the split-operand fp32 arithmetic of vcv_conv_x3_* (wgrad_bf16.hip with three term planes per
 * operand): calling convention of vcv_wgrad_bf16_scratch / vcv_wgrad_bf16; deterministic (per-workgroup slabs added in
 * a fixed order) */
int64_t vcv_wgrad_x3_scratch(const VcvWgradArgs* args);
int vcv_wgrad_x3(const VcvWgradArgs* args, float* scratch, int64_t scratch_floats, void* stream);
/* tuning probe: fix the tile candidate (0..5, -1 = first that fits) and the reduction split (> 0, -1 = cost model) */
int vcv_wgrad_bf16_set_force(int cand, int z);

/* Thin convolutions (HBM-bound, no MFMA): a conv with ONE output channel (discriminator conv_post
 * 1024->1, discriminator.py:25,61; generator conv_post 32->1 + tanh) and the weight gradient of a
 * conv with one output or one input channel (conv_post / first layers, discriminator.py:18,53).
 * x [B,C,Tin,P], w [1,C,K], y [B,1,Tout,P]; in_leaky / out_act as in vcv_conv_gemm.
 * vcv_thin_wgrad: dw[m,c,k] += alpha * sum tf(a[b,m,q,p]) * tf(bsh[b,c,q*s+k*d+off,p]), K <= 16. */
int vcv_conv_m1_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int Tin,
                    int Tout, int P, int K, int stride, int dil, int pad, int in_leaky, int out_act,
                    float slope, void* stream);
/* One INPUT channel (first discriminator layers, discriminator.py:18,53): x [B,1,Tin,P], w [M,1,K] (M <= 64,
 * K <= 16), y / dy [B,M,Tout,P].  Forward fuses bias + out_act; dgrad gives dx [B,1,Tin,P] (overwrites). */
int vcv_conv_c1_fwd(const float* x, const float* w, const float* bias, float* y, int B, int M, int Tin, int Tout,
                    int P, int K, int stride, int dil, int pad, int out_act, float slope, void* stream);
/* The same with y[b,m,t,p] *= leaky'(oaux[b,m,t,p]) in the epilogue (oaux may be NULL): the data gradient of a one-OUTPUT-
 * channel conv is this launch with the flipped taps, and oaux the leaky-ReLU output the gradient flows back into
 * (the pass the reference's autograd spends on LeakyReluBackward, discriminator.py:38,69; generator conv_post). */
int vcv_conv_c1_fwd_masked(const float* x, const float* w, const float* bias, float* y, const float* oaux, int B, int M,
                           int Tin, int Tout, int P, int K, int stride, int dil, int pad, int out_act, float slope,
                           void* stream);
/* ... with the taps of every weight row read in reverse order (flip_taps != 0): the data gradient of a one-OUTPUT-channel conv
 * (the discriminators' conv_post, discriminator.py:39,61) is a one-input-channel convolution of dy with the flipped taps */
int vcv_conv_c1_fwd_flip(const float* x, const float* w, const float* bias, float* y, const float* oaux, int B, int M, int Tin,
                         int Tout, int P, int K, int stride, int dil, int pad, int out_act, float slope, int flip_taps,
                         void* stream);
int vcv_conv_c1_dgrad(const float* dy, const float* w, float* dx, int B, int M, int Tin, int Tout, int P, int K,
                      int stride, int dil, int pad, void* stream);
/* One-frame pointwise layers (speaker conditioning `cond_layer` / `cond`: Conv1d(gin, M, 1) applied to g [B, gin, 1],
 * modules.py:126-131 and the hub generator): x [B, C], w [M, C], y / dy [B, M], B <= 32.  Forward fuses the bias;
 * dgrad overwrites dx [B, C]; wgrad ADDS onto dw [M, C]. */
int vcv_linear_t1_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int M, void* stream);
int vcv_linear_t1_dgrad(const float* dy, const float* w, float* dx, int B, int C, int M, void* stream);
int vcv_linear_t1_wgrad(const float* dy, const float* x, float* dw, int B, int C, int M, void* stream);
int vcv_thin_wgrad(const float* a, const float* bsh, const float* aaux, const float* baux, float* dw, int B,
                   int M, int C, int Ta, int Tb, int P, int K, int s, int d, int off, int a_tf, int b_tf,
                   float slope, float alpha, void* stream);

/* Grouped k=41, stride 4, padding 20 convolutions with 4 input channels per group (DiscriminatorS
 * layers 2-5, discriminator.py:55-58): direct fp32 FMA kernels (an MFMA tile would be mostly padding).
 * x [B, G*4, Tin], w [G*Mg, 4, 41], y/dy [B, G*Mg, Tout], Mg in {4, 16}.  dtf / yaux: activation
 * derivative applied to dy while staging: VCV_TF_NONE or VCV_TF_DLEAKY with yaux = y (any other dtf: VCV_EINVAL).
 * wgrad accumulates into dw. */
int vcv_grouped41_fwd(const float* x, const float* w, const float* bias, float* y, int B, int G, int Mg, int Tin,
                      int Tout, int out_act, float slope, void* stream);
/* The forward with bf16 operands (v_mfma_f32_16x16x16_bf16: K = 4 taps x the group's 4 input channels; fp32 accumulate) for the
 * 16-channel groups (Mg == 16, else VCV_EINVAL): the library's bf16 mode. */
int vcv_grouped41_fwd_bf16(const float* x, const float* w, const float* bias, float* y, int B, int G, int Mg, int Tin, int Tout,
                           int out_act, float slope, void* stream);
int vcv_grouped41_dgrad(const float* dy, const float* yaux, const float* w, float* dx, int B, int G, int Mg, int Tin,
                        int Tout, int dtf, float slope, void* stream);
/* bf16-operand form for the 16-channel groups (Mg == 16, else VCV_EINVAL): K of the bf16 MFMA = the group's 16 output channels. */
int vcv_grouped41_dgrad_bf16(const float* dy, const float* yaux, const float* w, float* dx, int B, int G, int Mg, int Tin, int Tout,
                             int dtf, float slope, void* stream);
int vcv_grouped41_wgrad(const float* dy, const float* yaux, const float* x, float* dw, int B, int G, int Mg, int Tin,
                        int Tout, int dtf, float slope, void* stream);
/* bf16-operand form for the 16-channel groups (Mg == 16, else VCV_EINVAL): K of the bf16 MFMA = 16 consecutive output times. */
int vcv_grouped41_wgrad_bf16(const float* dy, const float* yaux, const float* x, float* dw, int B, int G, int Mg, int Tin, int Tout,
                             int dtf, float slope, void* stream);

/* sum over (b, t) of tf(dy) per channel -> dbias[C] (overwrites, or adds onto dbias when `accumulate`).
 * dy: [B, C, T] (T = Tout*P) */
int vcv_bias_grad(const float* dy, const float* aux, float* dbias, int B, int C, int T, int tf, float slope,
                  int accumulate, void* stream);

/* ---- weight norm: torch.nn.utils.weight_norm(dim=0) at modules.py:126,132,143,190-201 and
 * discriminator.py:16-25,52-61.  v,w: [R, C] rows; g, norm: [R] ---- */
int vcv_weight_norm_fwd(const float* v, const float* g, float* w, float* norm, int R, int C, void* stream);
int vcv_weight_norm_bwd(const float* dw, const float* v, const float* g, const float* norm, float* dv,
                        float* dg, int R, int C, void* stream);
/* Batched form: one launch for all weight-normed layers of a module.  `items_dev` is a DEVICE array of
 * n_items records of ten int64 {v ptr, g ptr, w offset (floats into wbuf), first row (index into the
 * row-indexed norm buffer), R, C, dw ptr, dv ptr, dg ptr, accumulate}, sorted by first row; total_rows =
 * sum of R.  Backward writes dv / dg of every record through its own pointers, adding onto the existing
 * values where `accumulate` is set (gradient buffers of an optimizer). */
int vcv_weight_norm_many_fwd(const void* items_dev, int n_items, int total_rows, float* wbuf, float* norm,
                             void* stream);
int vcv_weight_norm_many_bwd(const void* items_dev, int n_items, int total_rows, const float* norm, void* stream);

/* ---- spectral norm: torch.nn.utils.spectral_norm (dim 0, n_power_iterations 1, eps 1e-12), which the reference's
 * discriminators use in place of weight norm under use_spectral_norm=True (discriminator.py:17,52;
 * multi_scale_discriminator.py:13-19).  w, w_sn: [R, N] rows (R = output channels); u [R], v [N]: the layer's persistent
 * vectors, UPDATED IN PLACE when `power_iteration` (training forward); sigma [1] = u . (W v); w_sn = w / sigma.
 * work: R + N floats (fwd), 256 floats (bwd).  Backward: dw = (dw_sn - <dw_sn, w_sn> u v^T) / sigma with the u, v,
 * sigma of that forward (copies: the next forward overwrites the vectors). ---- */
int vcv_spectral_norm_fwd(const float* w, float* u, float* v, float* w_sn, float* sigma, float* work, int R, int N,
                          int power_iteration, float eps, void* stream);
int vcv_spectral_norm_bwd(const float* dw_sn, const float* w_sn, const float* u, const float* v, const float* sigma,
                          float* dw, float* work, int R, int N, void* stream);

/* wt[c, m, K-1-k] = w[m, c, k]: lets the stride-1 data gradient of a conv run as a forward conv */
int vcv_weight_flip_transpose(const float* w, float* wt, int M, int C, int K, void* stream);

/* out = dy * act'(y) for tf in {DLEAKY, DRELU, DTANH, DLOGCLAMP}: one pass that the data-, weight- and
 * bias-gradient kernels of a fused conv+activation then share */
int vcv_act_grad(const float* dy, const float* y, float* out, int tf, float slope, int64_t n, void* stream);
/* out = (dy + add) * act'(y); add may be NULL (= vcv_act_grad).  `add`: a second gradient of the same tensor -- the
 * feature-matching loss's gradient of a recorded discriminator feature map (vcvits.py:119 loss_fm over fmaps of
 * discriminator.py:38-45) -- summed in this pass instead of in one of its own */
int vcv_act_grad_add(const float* dy, const float* add, const float* y, float* out, int tf, float slope, int64_t n, void* stream);
/* the same pass over [B, C, T] tensors that also collects the bias gradient: dbias[c] += sum_{b, t} out[b, c, t] */
int vcv_act_grad_bias(const float* dy, const float* y, float* out, float* dbias, int B, int C, int T, int tf, float slope,
                      void* stream);

/* ---- streaming helpers ---- */
/* y = (a + b + c) / 3 : mean of the three ResBlock1 branches of a HiFi-GAN stage */
int vcv_avg3(const float* a, const float* b, const float* c, float* y, int64_t n, void* stream);
int vcv_scale(const float* x, float* y, float alpha, int64_t n, void* stream);
/* y[b,c,t] = x[b,c,t] * mask[b,t]   (the `* x_mask` of modules.py / posterior_encoder.py) */
int vcv_mask_mul(const float* x, const float* mask, float* y, int B, int C, int T, void* stream);
/* F.pad(x, (0, Tp-T), "reflect") of discriminator.py:33-36 on rows [R, T] -> [R, Tp], and its adjoint */
int vcv_reflect_pad_fwd(const float* x, float* y, int R, int T, int Tp, void* stream);
int vcv_reflect_pad_bwd(const float* dy, float* dx, int R, int T, int Tp, void* stream);
/* AvgPool1d(4, 2, padding=2) of multi_scale_discriminator.py:20-25: [R, T] -> [R, T/2+1], and adjoint */
int vcv_avgpool4_fwd(const float* x, float* y, int R, int T, void* stream);
int vcv_avgpool4_bwd(const float* dy, float* dx, int R, int T, void* stream);

/* ---- loss reductions (losses.py:4-38, vcvits.py:115): out[0] += scale * sum f(a,b)
 * mode 0: |a-b| ; mode 1: (a-target)^2.  grad: da (+)= scale * gout[0] * f'(a,b) ---- */
int vcv_loss_sum(const float* a, const float* b, float target, int mode, float scale, float* out,
                 int64_t n, void* stream);
int vcv_loss_grad(const float* a, const float* b, float target, int mode, float scale, const float* gout,
                  float* da, int accumulate, int64_t n, void* stream);
/* Batched form of the two calls above over a list of tensors.  `items_dev`: DEVICE array of n_items records
 * of six int64 {a ptr, b ptr (mode 0), n, first workgroup, float bits of scale, offset of da_i (floats) in
 * dabuf}, sorted by first workgroup; total_blocks = workgroups of the launch.  sum: out[i] += scale_i *
 * sum f(a_i, b_i) (out zeroed by the caller); grad: da_i = scale_i * gout[i] * f'(a_i, b_i). */
int vcv_loss_many_sum(const void* items_dev, int n_items, int total_blocks, float target, int mode, float* out,
                      void* stream);
int vcv_loss_many_grad(const void* items_dev, int n_items, int total_blocks, float target, int mode,
                       const float* gout, float* dabuf, void* stream);

/* ---- one conv PAIR of a ResBlock1 over 16-bit activations as ONE launch (inference; modules.py:186-222 behind
 * synthesizer_svc.py:108 under fp16 autocast):  xt = leaky(conv1(leaky(x); w1, dilation dil) + b1);  out = conv2(xt; w2) + b2 + x;
 * y = out, or y += post_scale * out with `accumulate` (a block's last pair: the stage mean).  x, y: fp16 [B, C, T], T a multiple
 * of 8, 16-byte aligned; both convs [C, C, K] "same"-padded; the intermediate xt (rounded to bf16 as the two-launch path stores
 * it) never leaves the CU.  C in {32, 64}, K in {3, 7, 11}, dil <= 5 where the LDS images fit (vcv_resblock_pair_supported
 * returns the bytes of the packed weight buffer, 0 = run the pair as two vcv_conv_bf16io_* launches). */
typedef struct VcvResPairArgs {
  const void* x;    /* fp16 [B, C, T] */
  const void* wp;   /* packed weights of both convs (vcv_resblock_pair_pack) */
  const float* b1;  /* [C] */
  const float* b2;  /* [C] */
  void* y;          /* fp16 [B, C, T] */
  int32_t B, C, T, K, dil, accumulate;
  float post_scale; /* 0 = none */
  float slope;      /* leaky-ReLU slope of both convs' inputs */
} VcvResPairArgs;
int64_t vcv_resblock_pair_supported(int C, int K, int dil, int T);
int vcv_resblock_pair_pack(const float* w1, const float* w2, void* wp, int C, int K, void* stream);
int vcv_resblock_pair_x16(const VcvResPairArgs* args, void* stream);

/* ---- embedding rows laid out [B, C, T] (content_encoder.py:58-60: emb_pitch(pitch).transpose(1, -1); synthesizer_svc.py:77:
 * emb_g(sid).unsqueeze(-1) with T = 1) and the table gradient.  idx: int64 [B, T]; W: [rows, C]; y / dy: [B, C, T].
 * An index outside [0, rows) reads as a zero row.  The gradient is one workgroup per table row, positions summed in ascending
 * order: no atomics, no sort, no host read-back (torch's embedding_dense_backward sorts with thrust and reads the segment
 * count back -- not capturable into a HIP graph on this stack); accumulate: dW += instead of dW =. */
int vcv_embedding_t_fwd(const void* idx, const float* W, float* y, int B, int T, int C, int rows, void* stream);
/* ... with an error word: the number of positions whose index fell outside [0, rows) is ADDED onto *err (device int32, caller-
 * owned; NULL: not counted).  The reference's nn.Embedding raises on such an index (content_encoder.py:40: emb_pitch;
 * synthesizer_svc.py:68: emb_g); the host mirror reads the word at its check points and raises there. */
int vcv_embedding_t_fwd_checked(const void* idx, const float* W, float* y, int B, int T, int C, int rows, int* err, void* stream);
int vcv_embedding_t_bwd(const void* idx, const float* dy, float* dW, int B, int T, int C, int rows, int accumulate, void* stream);

/* ---- torch.optim.AdamW step over a flat buffer (vcvits.py:247-257) ---- */
int vcv_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
              float eps, float wd, int step, void* stream);
/* The same step with the per-step scalars in device memory: hyper_dev -> {fp32 learning rate, int32 step delta}; the step
 * count of the bias corrections is step_base + delta.  For an optimizer step recorded into a HIP graph (arguments are
 * baked at capture; the host refreshes the 8-byte record before each replay). */
/* dst[0..n) = the first n <= 4 of (w0..w3), passed as kernel arguments (the record above: fp32 bits of lr, step delta). */
int vcv_set_words(void* dst, int n, int w0, int w1, int w2, int w3, void* stream);
int vcv_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, float b1, float b2, float eps, float wd,
                  const void* hyper_dev, int step_base, void* stream);

/* ---- STFT magnitude sqrt(re^2+im^2+eps) (mel_processing.py:54-96): n_fft = 2048 (both reference configs; the tuned
 * kernels), any other power of two in [64, 4096] (generic radix-2 kernels) or any other even size in [16, 4096] (direct DFT;
 * stft_generic.hip).
 * y: [B, T]; window: [n_fft] -- the analysis window already zero-padded (centred) to n_fft when win_length < n_fft, as
 * torch.stft does; twiddle: [n_fft/2] complex (cos, -sin)(2*pi*k/n_fft) interleaved;
 * mag/dmag: [B, n_fft/2+1, F], F = (T + 2*pad - n_fft)/hop + 1; reflect: 0 zero pad (torchaudio
 * spectrogram, :76-96), 1 reflect pad (:54-74).  bwd overwrites dy [B, T]. ---- */
int vcv_stft_mag_fwd(const float* y, const float* window, const float* twiddle, float* mag, int B, int T,
                     int n_fft, int hop, int pad, int reflect, float eps, void* stream);
int vcv_stft_mag_bwd(const float* y, const float* window, const float* twiddle, const float* dmag,
                     float* dy, int B, int T, int n_fft, int hop, int pad, int reflect, float eps,
                     void* stream);

/* ---- WaveNet block glue (modules.py:147-175; gate = commons.py:99-106) ----
 * xin [B,2H,T]; g = cond_layer output viewed [B, gstride] (NULL when no speaker conditioning),
 * layer slice at goff; acts [B,H,T] = tanh(xin[:, :H]+g[:H]) * sigmoid(xin[:, H:]+g[H:]). */
int vcv_wn_gate_fwd(const float* xin, const float* g, int gstride, int goff, float* acts, int B, int H, int T,
                    void* stream);
int vcv_wn_gate_bwd(const float* xin, const float* g, int gstride, int goff, const float* dacts, float* dxin,
                    int B, int H, int T, void* stream);
/* out[(r/inner)*ostride + ooff + r%inner] = sum_t x[r, t]  (dg of the gate: rows = B*2H, inner = 2H) */
int vcv_row_sum(const float* x, float* out, int R, int T, int inner, int ostride, int ooff, void* stream);
/* x_new = (x + rs[:, :H])*mask, out_new = out + rs[:, H:]  (last layer: out_new = out + rs); out may be NULL */
int vcv_wn_res_skip_fwd(const float* x, const float* out, const float* rs, const float* mask, float* xn,
                        float* on, int B, int H, int T, int last, void* stream);
int vcv_wn_res_skip_bwd(const float* dxn, const float* don, const float* mask, float* drs, float* dx, int B,
                        int H, int T, void* stream);

/* ---- (m, logs) = split(stats * mask); z = (m + eps*exp(logs))*mask  (posterior_encoder.py:36-38,
 * content_encoder.py:70-72).  eps/z may be NULL (split only). ---- */
int vcv_split_sample_fwd(const float* stats, const float* eps, const float* mask, float* m, float* logs,
                         float* z, int B, int C, int T, void* stream);
int vcv_split_sample_bwd(const float* dm, const float* dlogs, const float* dz, const float* eps,
                         const float* logs, const float* mask, float* dstats, int B, int C, int T, void* stream);

/* ---- mean-only coupling (modules.py:327-336): y = m + x1*mask, or (x1 - m)*mask when reverse ---- */
int vcv_coupling(const float* x1, const float* m, const float* mask, float* y, int B, int C, int T, int reverse,
                 void* stream);

/* ---- prior sample of SynthesizerSVC.infer (synthesizer_svc.py:104): z = m + noise * exp(logs) * noise_scale ---- */
int vcv_prior_sample(const float* m, const float* logs, const float* noise, float* z, int64_t n, float noise_scale,
                     void* stream);

/* ---- LayerNorm over the channel dim of [B,C,T] applied to (x + y) (modules.py:19-31 with the
 * residual add of relative_attention_transformer.py:41,45 fused); y may be NULL ---- */
int vcv_layernorm_c_fwd(const float* x, const float* y, const float* gamma, const float* beta, float* out,
                        float* mean, float* rstd, int B, int C, int T, float eps, void* stream);
int vcv_layernorm_c_bwd(const float* x, const float* y, const float* gamma, const float* mean,
                        const float* rstd, const float* dout, float* dx, float* dgamma, float* dbeta, int B,
                        int C, int T, void* stream);
/* The same with a caller-owned workspace for the partial sums of the register-resident form (C = 128 / 256):
 * vcv_layernorm_c_bwd_scratch floats (0: none needed).  vcv_layernorm_c_bwd keeps a library-owned scratch per device. */
int64_t vcv_layernorm_c_bwd_scratch(int B, int C, int T);
int vcv_layernorm_c_bwd_ws(const float* x, const float* y, const float* gamma, const float* mean, const float* rstd,
                           const float* dout, float* dx, float* dgamma, float* dbeta, int B, int C, int T,
                           float* scratch, int64_t scratch_floats, void* stream);

/* ---- banded relative-position softmax (relative_attention_transformer.py:157-180).
 * S,P,Pt,dP,dSt: [B*H, T, T]; q,dO,dqband: [B, H*dk, T]; embk/embv: [2w+1, dk]; mask [B,T]. ---- */
/* P = softmax probabilities; Pd = P after dropout(pdrop) (NULL / unused when pdrop == 0: Pd == P);
 * Pt = transpose of Pd (the layout the P.V contraction consumes). */
int vcv_rel_softmax_fwd(const float* S, const float* q, const float* embk, const float* mask, float* P,
                        float* Pd, float* Pt, int B, int H, int dk, int T, int w, float qscale, float pdrop,
                        uint64_t seed, void* stream);
int vcv_rel_value_fwd(const float* P, const float* embv, float* out, int B, int H, int dk, int T, int w,
                      void* stream);
int vcv_rel_softmax_bwd(const float* P, const float* Pd, float* dP, const float* dO, const float* q,
                        const float* embk, const float* embv, const float* mask, float* dSt, float* dqband,
                        float* dembk, float* dembv, int B, int H, int dk, int T, int w, float qscale,
                        void* stream);

/*
 * Fused relative-position self-attention (attention.hip), replacing the four / six launches above per layer
 * (relative_attention_transformer.py:150-182 and its autograd): QK^T, banded relative-key logits, masked_fill(-1e4),
 * softmax, dropout, P.V + banded relative values in ONE launch; the backward pass in two (row pass: dP, dS, dQ, table
 * gradients; column pass: dV, dK).  Both contractions on the matrix cores (bf16 != 0: operands rounded to bf16, fp32
 * accumulate).  q / k / v / out / dO / dq / dk / dv: [B, H*dk, T]; embk / embv: [2w+1, dk]; mask: [B, T]; P / Pd / dS:
 * [B*H, T, T] (forward: P and Pd may each be NULL = not wanted; the backward pass regenerates the dropout mask from
 * `seed`).  vcv_rel_attn_supported() == 0 for the shapes these kernels take (dk <= 64 and even, T <= 896, 2w+1 <= 16).
 */
int vcv_rel_attn_supported(int B, int H, int dk, int T, int w);
int vcv_rel_attn_fwd(const float* q, const float* k, const float* v, const float* embk, const float* embv,
                     const float* mask, float* out, float* P, float* Pd, int B, int H, int dk, int T, int w, float qscale,
                     float pdrop, uint64_t seed, int bf16, void* stream);
int vcv_rel_attn_bwd(const float* q, const float* k, const float* v, const float* embk, const float* embv,
                     const float* mask, const float* P, const float* dO, float* dS, float* dq, float* dk_out, float* dv,
                     float* dembk, float* dembv, int B, int H, int dk, int T, int w, float qscale, float pdrop,
                     uint64_t seed, int bf16, void* stream);
/* The same with the forward's output `out` [B, H*dk, T] (null: as vcv_rel_attn_bwd); with `out`, dS must hold
 * B*H*T*T + B*H*ceil(T/32)*2*(2w+1)*dk floats (per-tile partial tables of dembk / dembv follow dS; summed in a fixed order).  Given it, shapes with T <= 256 and
 * 32 or 64 channels per head run the wave-per-tile backward (dS formed in one pass: sum_j dPd Pd = sum_d dO out). */
int vcv_rel_attn_bwd2(const float* q, const float* k, const float* v, const float* embk, const float* embv,
                      const float* mask, const float* P, const float* out, const float* dO, float* dS, float* dq,
                      float* dk_out, float* dv, float* dembk, float* dembv, int B, int H, int dk, int T, int w, float qscale,
                      float pdrop, uint64_t seed, int bf16, void* stream);
/* nn.Dropout (relative_attention_transformer.py:40,44,292): y = x * mask(seed, index) / (1-p); the
 * backward is the same call on dy with the same seed */
int vcv_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream);

/* ---- KL loss (losses.py:40-55): out2[0] = sum(kl*mask), out2[1] = sum(mask) ---- */
int vcv_kl_fwd(const float* zp, const float* lq, const float* mp, const float* lp, const float* mask,
               float* out2, int B, int C, int T, void* stream);
int vcv_kl_bwd(const float* zp, const float* mp, const float* lp, const float* mask, const float* gout,
               const float* den, float* dzp, float* dlq, float* dmp, float* dlp, int B, int C, int T,
               void* stream);

/* ---- F.interpolate(mode="nearest") along T (synthesizer_svc.py:83-84) and slice_segments
 * (commons.py:48-54): y[b,c,s] = x[b,c,ids[b]*mul + s] ---- */
int vcv_nearest_fwd(const float* x, float* y, int R, int Tin, int Tout, void* stream);
int vcv_nearest_bwd(const float* dy, float* dx, int R, int Tin, int Tout, void* stream);
/* ... with the index map of the batch's RAW padded sizes (raw: device int64[2] = {Tin_raw <= Tin or <= 0 for Tin itself,
 * Tout_raw <= Tout}):
 * y[r, to] = x[r, floor(to * Tin_raw / Tout_raw)] for to < Tout_raw, 0 beyond.  synthesizer_svc.py:82-83 interpolates padded
 * content frames onto padded spectrogram frames, so the alignment depends on the batch's own maxima; a batch padded further
 * to bucket multiples (vcvits_amd/data/collate.py) keeps that alignment by passing them here.  Read from device memory: a
 * recorded batch replays with each batch's own pair. */
int vcv_nearest_raw_fwd(const float* x, float* y, int R, int Tin, int Tout, const void* raw, void* stream);
int vcv_nearest_raw_bwd(const float* dy, float* dx, int R, int Tin, int Tout, const void* raw, void* stream);
int vcv_slice_fwd(const float* x, const int64_t* ids, int mul, float* y, int B, int C, int T, int S,
                  void* stream);
int vcv_slice_bwd(const float* dy, const int64_t* ids, int mul, float* dx, int B, int C, int T, int S,
                  void* stream);

/* ---- optional per-launch HIP-event timing of the MFMA kernel families (bench.py roofline).
 * vcv_prof_begin arms a pool of `max_launches` event pairs; every vcv_conv_gemm / vcv_conv_dma /
 * vcv_conv_wgrad launch then records one pair around its MFMA kernel on its own stream.  vcv_prof_end
 * synchronises and fills out[cls*3 + {0,1,2}] = {launches, total ms, total algorithmic flops}, cls 0 =
 * conv_gemm_kernel, 1 = conv_wgrad_kernel, 2 = packed-weight conv kernels, 3 = weight-gradient kernels, 4 = fused attention (ncls >= 5). ---- */
int vcv_prof_begin(int max_launches);
int vcv_prof_end(double* out, int ncls);
/* out[cls] = algorithmic HBM bytes (operands read once + result written once) summed over the class's launches of the
 * window vcv_prof_end just closed */
int vcv_prof_bytes(double* out, int ncls);
/* out[cls] = sum over the class's launches in that window of their time at the DENSE PEAK of the matrix pipe each one
 * runs on (seconds): MFMA flops the launch executes / peak of its pipe (fp32-input MFMA 157.3 TFLOP/s; bf16 MFMA 2.5
 * PFLOP/s; a split-operand fp32 launch executes 6 or 9 bf16 products per fp32 product).  sum / measured time = the
 * roofline fraction of a class that mixes pipes. */
int vcv_prof_roof(double* out, int ncls);
/* sampling inside an open window: while paused != 0 launches carry no events (they run as plain launches) */
int vcv_prof_pause(int paused);
int vcv_prof_active(void); /* 1 while launches get events attached (window open, not paused) */
/* per-launch CSV of the last profiling window: cls, ms, gflop, 12 shape tags */
int vcv_prof_dump(const char* path);

/* ---- source-audio front end (vits/model/pipeline.py:24-28,59-70):
 * complex STFT (torchaudio Spectrogram(power=None, pad, center=False)) and inverse STFT
 * (torchaudio InverseSpectrogram = torch.istft, Hann window, center=True trims n_fft/2 per side).
 * n_fft = 2048 (tuned kernels) or any even size in [16, 4096]; window / twiddle tables as for vcv_stft_mag_fwd.
 * spec: complex64 [B, n_fft/2+1, F] interleaved (re, im); ola: workspace [B, n_fft + hop*(F-1)] ---- */
int vcv_stft_complex_fwd(const float* y, const float* window, const float* twiddle, float* out, int B, int T,
                         int n_fft, int hop, int pad, int reflect, void* stream);
int vcv_istft(const float* spec, const float* window, const float* twiddle, float* ola, float* out, int B, int F,
              int n_fft, int hop, int center, void* stream);

/* returns a static string describing the build (arch, kernel variants) */
const char* vcv_version(void);
/* Deterministic mode (also VCVITS_DETERMINISTIC=1): every launcher that splits a reduction over workgroups and combines
 * with fp32 atomics runs it unsplit, one writer per output element (slower; bit-reproducible run to run).  The MFMA
 * weight-gradient kernels combine through VcvWgradArgs.slab instead; the caller passes it (and no dbias). */
int vcv_set_seed_offset_ptr(const void* dev_u64); /* non-NULL: the FORWARD dropout / attention launchers make their kernels add
                                                      *dev_u64 to the host seed -- a launch sequence replayed from a HIP
                                                      graph (vcvits_amd/light/graphed.py) gets fresh masks per replay */
const void* vcv_get_seed_offset_ptr(void);
int vcv_set_deterministic(int on);
int vcv_get_deterministic(void);

/* ---- tuning table (csrc/tuning.h): the library's A/B switches and tuning probes, one int per key, defaults = the measured
 * choices.  Initialised once from the environment variable VCVITS_TUNING="key=value,key=value" (plus VCVITS_DETERMINISTIC=1);
 * readable / writable at run time.  Not part of the reference's interface: the reference has no kernels to tune.
 * Returns VCV_EINVAL for an unknown key. */
int vcv_tuning_set(const char* key, int value);
int vcv_tuning_get(const char* key, int* value);

#ifdef __cplusplus
}
#endif
#endif /* VCVITS_HIP_H */
