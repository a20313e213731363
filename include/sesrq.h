/*
 * sesrq -- MI355X-native INT8 SESR / NRDM integer inference (C ABI).
 *
 * Drop-in boundary for ONE hot path of gui-yupeng/sesr-pytorch-quantize: the per-conv chain
 *   quantize_asymmetrical_by_tensor -> reshape_input_for_hardware_pe -> nn.Conv2d ->
 *   PEs_and_bias_adder -> requan_conv2d_output -> ReLU  (x5)  -> PixelShuffle
 * that the reference's sim.py executes as a torch.fx graph (sim.py:82-114, :205).  The
 * reference has no FFI: its boundary is five Python callables that hand state to each other
 * through files under ./output_pt/.  Each entry point below names the reference code it
 * replaces (paths relative to the reference root).
 *
 * Plain C types only: pointers, sizes, ints.  Device pointers are HIP device pointers of
 * the current device; `stream` is a hipStream_t passed as void* (NULL = default stream).
 * All functions return 0 on success, non-zero on error; sesrq_last_error() returns a
 * thread-local message for the last failure on the calling thread.
 */
#ifndef SESRQ_H
#define SESRQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SESRQ_VERSION 4
#define SESRQ_MAX_LAYERS 16
#define SESRQ_MAX_CH 16

/* data types of the frame buffers handed over the boundary (always NCHW, like the reference) */
enum { SESRQ_F32 = 0, SESRQ_I8 = 1 };

/* kernel families (sesrq_options.engine): AUTO = MFMA kernels where a layer shape has one, else dot4 */
enum { SESRQ_ENGINE_AUTO = 0, SESRQ_ENGINE_DOT4 = 1, SESRQ_ENGINE_MFMA = 2 };

/* Options of a net, fixed at sesrq_create (there is no setter: a created net is immutable).
 * sesrq_default_options() fills the defaults; a NULL options pointer means the defaults too. */
typedef struct sesrq_options {
    int32_t engine;          /* SESRQ_ENGINE_*                                                        (default AUTO) */
    int32_t force_general;   /* 1: always run the per-PE clamp path, even where a load-time proof says it is a no-op */
    int32_t exact_div;       /* how the input quantiser forms x / s0 (myQL/quan_func.py:225).  0 (default): the true fp32 quotient --
                              * what torch evaluates on a CPU and what every golden vector pins -- by a 3-instruction form
                              * where sesrq_create proved it bit-identical, else by the division instruction (the first layer
                              * then runs on the dot4 kernel: the MFMA first-layer kernels only carry the 3-instruction form);
                              * 1: always the division instruction; 2: x * fl(1/s0), what torch evaluates for tensor / scalar on a GPU (the
                              * reference's scripts call .cuda()): differs from the quotient by an ulp at most, i.e. q0 by one
                              * LSB at rounding ties only -- parity UNPINNED (no fixture of a GPU-run reference exists) */
    int32_t anchor_add;      /* 1: add the nearest-upsampled fp32 input frame to the fp32 output (the x2 "anchor" of the
                              * reference's eval loop, test.py:148-155: gfake + inps_x2); needs Cin*r*r == Cout and an
                              * fp32 input; the int8 output is unaffected */
    int32_t fuse_hidden;     /* 1 (default): every eligible run of three hidden 3x3 layers is ONE launch (the residual-merging trio
                              * first); 0: one launch per layer.  (Rounds 2-3 also had 2 = the first layer inside the trio's launch,
                              * bit-identical and measured slower: retired in round 4, DESIGN.md 4.2b keeps the measurement.) */
    int32_t wg_budget;       /* workgroup slots one launch may fill; 0 (default) = one full round of the chip (occupancy x CUs).
                              * The persistent kernels cut every 64-column strip into as many vertical runs as fit the budget:
                              * a small budget leaves compute units to a concurrent stream (and makes a workgroup walk many
                              * tiles, which the tests use to cover the steady-state walk on small frames) */
    float i8_in_scale;       /* 0 (default): an SESRQ_I8 input frame already is q0 (input.0.pt).  > 0: it is the int8 OUTPUT of an
                              * upstream net in the domain (i8_in_scale, i8_in_zero) = that net's (scale_out, zero[L]); the first
                              * layer re-quantises it while staging, x = (q - zero) * f32(scale) then clamp8(rint(x/s0 + z0)) --
                              * bit-identical to handing the upstream net's fp32 output over (the float hand-off between chained
                              * nets) at a quarter of the bytes */
    int32_t i8_in_zero;
    int32_t reduced_forms;   /* bit mask of the PROVEN reduced epilogue forms the kernels may select (round 5; no reference counterpart: every form
                              * gives the reference's bits, the mask only chooses which kernel instantiation computes them -- it exists so that
                              * tests can run EVERY instantiation on reference-made data).  -1 (default) = all; the environment variable
                              * SESRQ_DIRECT=0 turns the default into 1 | 8.
                              *   1: fused trio: cvt_pk_u8 epilogues where every zero point involved is -128
                              *   2: one-fma requant (sesrq_layer_one_fma == 1) in the first layer and in layers a, b of a fused trio
                              *   4: ... in the third layer of a fused trio
                              *   8: fused trio: the residual operand out of the LDS input window where it is the trio's input tensor
                              *  16: output layer: one-fma form 1        32: output layer: single-rounding form 2 (tried when form 1 is
                              *      not proven or not allowed) */
} sesrq_options;
void sesrq_default_options(sesrq_options *opts);

/* One collapsed convolution with its integer epilogue.
 *   w         : conv.weight.K.pt   (myQL/quan_func.py:71,78)  [oc][ic][k][k] int8
 *   add_const : conv.bias.quanK.pt (myQL/quan_func.py:481-489) clamp16(bq - zero*sum(W)), [oc]
 *   M, n      : requan_K_K+1 / n_K_K+1 (myQL/quan_func.py:495-515,527-528) t = acc*M*2^-n
 *   relu      : activation after the requant node (models/model_utils_pt.py:15-18) */
typedef struct sesrq_layer_desc {
    int32_t k;                 /* 3 or 5 (odd, stride 1, same padding) */
    int32_t ic, oc;            /* 1..16 */
    const int8_t *w;
    const int32_t *add_const;
    uint32_t M;                /* < 2^16  (define.py REQUAN_BIT)   */
    uint32_t n;                /* <= 32   (define.py REQUAN_N_MAX) */
    int32_t relu;
    /* Per-OUTPUT-CHANNEL requant constants [oc], or both NULL (the reference: one (M, n) per layer).  No reference counterpart -- its
     * weight quantiser is per tensor (myQL/quan_func.py:58-71); BASELINE's north star names per-channel weight scales: with a weight scale
     * per output channel the layer's requant multiplier s_in / s_next * s_w[oc] becomes per channel, t = f32(f32(acc) * f32(M_oc[oc])) *
     * 2^-n_oc[oc], and add_const[oc] is formed with s_w[oc].  PARITY UNPINNED (oracle <-> HIP self-consistency only).  A layer with these
     * set runs on the dot4 kernels (round 5; the MFMA kernels keep the per-tensor form and pay nothing for the option); M and n are ignored. */
    const uint32_t *M_oc;
    const uint32_t *n_oc;
} sesrq_layer_desc;

/* A whole net ("parameter bundle"; replaces the CWD-relative output_pt/ tree).
 * Roles follow myQL/quan_func.py:220-280,523-609 by position: layer 0 quantises the fp32
 * frame and publishes the long-residual operand; layer L-2 requantises into domain 1 and its
 * epilogue merges the residual (M_res, n_res; quan_func.py:249-270); layer L-1 requantises into
 * the output domain zero[L]/scale_out and is followed by PixelShuffle(pixel_shuffle). */
typedef struct sesrq_net_desc {
    int32_t n_layers;                  /* L >= 3 */
    const sesrq_layer_desc *layers;    /* [L] */
    const int32_t *zero;               /* [L+1] input.K.zero.pt, K = 0..L (test.py:185-217) */
    float scale_in;                    /* f32(input.0.scale) */
    float scale_out;                   /* f32(input.L.scale) */
    uint32_t M_res, n_res;             /* requan_res / n_res (quan_func.py:259-267) */
    int32_t pixel_shuffle;             /* 1 (none), 2, 4      (models/sesr_sim.py:31) */
    int32_t pe_num;                    /* define.py PE          (must be 4) */
    int32_t pe_acc_bits;               /* define.py PE_ACC_BIT  (18) */
    int32_t pe_add_bits;               /* define.py PE_ADD_BIT  (20) */
} sesrq_net_desc;

typedef struct sesrq_net sesrq_net;

/* ---- device path ------------------------------------------------------------------ */

/* Validates the bundle, repacks the weights for the kernels and uploads them to the current
 * HIP device.  The net is immutable afterwards (options included): forward calls are thread-safe
 * and stream-ordered, on the device the net was created on (a forward issued while another device
 * is current is refused).  Replaces quantize_model_weight's file output + every torch.load of
 * the output_pt/ tree inside the five callables. */
int sesrq_create(const sesrq_net_desc *desc, const sesrq_options *opts, sesrq_net **out);
void sesrq_destroy(sesrq_net *net);
/* 1 if sesrq_create proved (exhaustively, on the device) that the 3-instruction reciprocal form of
 * the input quantiser's x / scale_in is bit-identical for this net; 0 = IEEE division is used. */
int sesrq_fast_division_proven(const sesrq_net *net);
/* The load-time verdict on layer k's requant into its -128 domain, clamp8(rint(fl(fl(s * M) * 2^-n - 128))) (myQL/quan_func.py:280;
 * the output layer: :601): 1 = sesrq_create proved (all reachable sums enumerated on the host) that ONE fused multiply-add followed by
 * the saturating byte convert gives the same bits; 2 (output layer only) = that form failed, but one fma that also subtracts the 128 (a
 * single rounding of s * M * 2^-n - 128) followed by the add of 128 is proven identical; 0 = neither, or the layer does not requantise
 * into a -128 domain, or SESRQ_DIRECT=0.  For the residual-merging layer L-2 the flag speaks of its first requant (into the fixed -128
 * domain of ic, quan_func.py:250).  It is a PROOF, not a launch record: the reduced form runs in the first-layer MFMA kernel, in the
 * fused trio when all three of its layers carry the flag and every zero point its epilogues add is -128, and in the last-layer MFMA
 * kernel (int8 output, zero[L] == -128); the per-layer hidden kernels (fuse_hidden = 0), the dot4 engine and the debug forward always
 * run the two-step form -- the bits are the same either way.  No reference counterpart. */
int sesrq_layer_one_fma(const sesrq_net *net, int k);
/* The proof behind it as a host function of the requant constants alone (no device, no net): 1 = the one-fma form is
 * bit-identical to clamp8(rint(fl(fl(s * M) * 2^-n - 128))) for every accumulator value s, 2 = (output_layer != 0 only) it is
 * not, but the single-rounding form is, 0 = neither.  No reference counterpart. */
int sesrq_requant_form(uint32_t M, uint32_t n, int output_layer);

/* Channel geometry of a created net: input channels, output channels of the last conv (before PixelShuffle), PixelShuffle factor -- what a
 * caller needs to size the output of sesrq_forward, (N, cout / r^2, H * r, W * r) (csrc/torch_op/sesrq_torch_op.cpp does). */
int sesrq_net_shape(const sesrq_net *net, int *cin, int *cout, int *pixel_shuffle);

/* Bytes of device workspace sesrq_forward needs for N frames of H x W (caller-owned). */
size_t sesrq_workspace_bytes(const sesrq_net *net, int N, int H, int W);

/* The hot path: replaces `gfake = model(inps)` (sim.py:205).
 *   in     : (N, Cin, H, W)  fp32 frame (SESRQ_F32)  or already-quantised q0 int8 (SESRQ_I8)
 *   out_q  : (N, Cout, H*r, W*r) int8  -- input.L.pt after PixelShuffle; may be NULL
 *   out_f  : same shape, fp32 (q - zero_L) * scale_out -- what the reference returns
 *            (quan_func.py:594); may be NULL
 * No allocation, no synchronisation; everything is enqueued on `stream`. */
int sesrq_forward(const sesrq_net *net, const void *in, int in_dtype, void *out_q, void *out_f,
                  int N, int H, int W, void *workspace, size_t workspace_bytes, void *stream);

/* Many independent forwards enqueued by ONE call (round 4): frame k of `count` runs exactly what sesrq_forward would run for it, on
 * streams[k % n_streams] with workspaces[k % n_streams] (frames that share a stream are ordered on it, so they may share a workspace).
 * For callers whose frames are small enough that the HOST's per-call cost bounds the rate (540p: three launches of ~5 us each): one
 * crossing of the language boundary per batch instead of one per frame, the launch arguments of the net built once per call.  The
 * reference has no counterpart (it is batch-1, quan_func.py:349, 373); per frame the bytes are sesrq_forward's.  Every frames[k].in is an
 * (N, Cin, H, W) buffer of in_dtype, out_q / out_f as in sesrq_forward (either may be NULL, not both).  Caller-owned buffers, no
 * allocation, no synchronisation.  Returns non-zero at the first frame that fails (earlier frames stay enqueued).
 * Grouping (`group` = G, explicit since ABI v4 -- v3 inferred it from the workspace size, which turned a generously sized workspace into a
 * silent write race for callers that re-use one output buffer per stream): G = 1 keeps one launch sequence per frame.  G in 2..8 needs
 * N == 1, the MFMA first- and last-layer kernels (every reference net on the default engine) and workspaces of
 * sesrq_workspace_bytes(net, G, H, W) bytes; up to G consecutive frames of a stream then become the G images of ONE launch sequence
 * (their buffers stay where they are: a pointer table in the kernel arguments of the first and the last layer) -- the same kernels and
 * bytes, the launches' fixed cost once per group.  The frames of one group are written concurrently: two of them sharing an out_q or
 * out_f buffer is refused (error return, nothing of that group enqueued).
 * Threads: every stream's frames are enqueued by a persistent library thread of its own, in order (SESRQ_SUBMIT_THREADS=0: by the
 * caller's thread); a second sesrq_forward_many that arrives while one is running enqueues on its caller's thread.  A worker spins on
 * its job slot for SESRQ_SPIN_US microseconds (default 2000; 0 = sleep at once) after a batch before it sleeps on a condition variable:
 * with S streams that is up to S - 1 busy host cores between closely spaced batches -- set it to 0 where ranks outnumber a quarter of
 * the cores (bench.py does).  The pool is per process: a fork()ed child gets a fresh one on its first call (pthread_atfork), the
 * parent's threads are never touched from the child.  A job a worker does not pick up within 200 ms is taken back and run by the
 * caller; a worker that holds a job for more than 60 s makes the call return an error. */
typedef struct sesrq_frame_io {
    const void *in;
    void *out_q;
    void *out_f;
} sesrq_frame_io;
int sesrq_forward_many(const sesrq_net *net, const sesrq_frame_io *frames, int count, int in_dtype, int N, int H, int W,
                       void *const *workspaces, size_t workspace_bytes, void *const *streams, int n_streams, int group);

/* Self-test of the submission pool without a net or a device (tests/test_host_abi.py: a fork()ed child must get a working pool of its
 * own): `rounds` times, n_streams trivial jobs go through exactly the hand-off sesrq_forward_many uses (job 0 on the caller's thread, the
 * others on the pool's threads).  Returns the number of jobs that ran (n_streams * rounds) or -1. */
int sesrq_submit_selftest(int n_streams, int rounds);

/* Debug taps mirroring the reference's dump flags (define.py:23-31).  After a forward run
 * with sesrq_forward_debug, stage tensors are written to caller buffers (device pointers, any
 * may be NULL):
 *   act[k]    : (N, C_k, H, W) int8   input.k.pt, k = 0..L-1      (INPUT_W_FLG)
 *   pe_out[k] : (N, 4, OC_k, H, W) int32 pe_outputK_P.pt           (OUTPUT_PE_W_FLG)
 *   pe_add[k] : (N, OC_k, H, W) int32 pe_add_outputK.pt            (OUTPUT_PE_ADD_W_FLG)
 * A layer with taps runs its per-PE (general) kernel: on the MFMA engine the PE taps are written by the MFMA kernels
 * themselves; act[0] (the quantised input), the overflow counters and the pe-split last layer (OC <= 4) take their layer
 * to the dot4 kernels, which carry those taps.  The fused trio never runs in a debug forward.
 * There is no sesrq_forward_cpu (SURVEY 8b proposed one): the CPU restatement of the arithmetic is test infrastructure
 * and lives under oracle/, outside the product library. */
typedef struct sesrq_taps {
    void *act[SESRQ_MAX_LAYERS];
    void *pe_out[SESRQ_MAX_LAYERS];
    void *pe_add[SESRQ_MAX_LAYERS];
    /* (L, 2) int32, zeroed by the call: [k][0] = number of PE sums of layer k above the PE_ACC_BIT range,
     * [k][1] = below it, BEFORE saturation -- the events the reference reports as 'max_overflow' /
     * 'min_overflow' (myQL/quan_func.py:358-361) and then saturates silently, as this library does. */
    void *overflow;
    /* round 5: the two tensors the reference writes besides (both optional; either takes its layer to the dot4 kernels)
     *   shortcut : (N, OC_0, H, W) fp32  residual/shortcut_tensor.pt = relu(acc * M * 2^-n) of layer 0, un-rounded (myQL/quan_func.py:529-549)
     *   ic       : (N, OC_{L-2}, H, W) int8  input.4.spcial.pt = clamp8(rint(relu(t) - 128)) of layer L-2 (myQL/quan_func.py:250,254) */
    void *shortcut;
    void *ic;
} sesrq_taps;
int sesrq_forward_debug(const sesrq_net *net, const void *in, int in_dtype, void *out_q, void *out_f,
                        int N, int H, int W, void *workspace, size_t workspace_bytes, void *stream,
                        const sesrq_taps *taps);

/* The launch sequence of sesrq_forward for this net: launch j runs layers first[j] .. first[j]+count[j]-1
 * (count 3 = fused hidden trio).  Returns the number of launches (<= n_layers); first/count may be NULL. */
int sesrq_launch_plan(const sesrq_net *net, int *first, int *count);

/* Measurement hook: runs `iters` forwards back to back on `stream`, every kernel launched with its own begin / end
 * HIP events (hipExtLaunchKernelGGL: the timestamps of the dispatch itself -- the duration a rocprofv3 kernel trace
 * reports for it, without the dispatch latency a hipEventRecord pair around the launch would add), synchronises once
 * at the end and returns the AVERAGE device time per launch in launch_ms[0..sesrq_launch_plan()-1] (ms) and the
 * average time from the begin of the first to the end of the last kernel of a forward in *forward_ms.  Not part of
 * the hot path. */
int sesrq_forward_timed(const sesrq_net *net, const void *in, int in_dtype, void *out_q, void *out_f,
                        int N, int H, int W, void *workspace, size_t workspace_bytes, void *stream,
                        int iters, float *launch_ms, float *forward_ms);

/* Every kernel instantiation the library can launch, by construction (csrc/sesrq_common.h: launch_kernel<KERN> registers KERN when the
 * library is loaded), with the name a rocprofv3 kernel trace prints for it ("mfma_h5_kernel<1, 2, 22, 3>") and the number of launches
 * of it by this process so far -- so that a test can prove that it has run every instantiation (tests/test_gpu_parity.py:
 * test_every_kernel_instance_runs_on_reference_data).  No reference counterpart. */
int sesrq_instance_count(void);
const char *sesrq_instance_name(int i);
long long sesrq_instance_launches(int i);

/* Name of the kernel the net resolved to for layer k ("dot4-general", "mfma-h3-merged", "mfma-trio-merged", ...). */
const char *sesrq_layer_engine(const sesrq_net *net, int k);

/* ---- calibration pass (the reference's exe_mode 0) ------------------------ */

/* One conv of the calibration forward (reference: quantize_asymmetrical_by_tensor mode 0, quan_func.py:175-215
 * -> reshape_input_for_hardware_pe -> Conv2d with fake-quantised weights -> PEs_and_bias_adder mode 0,
 * quan_func.py:330-333,431-434,457-459 -> activation).  All pointers are device pointers. */
typedef struct sesrq_calib_conv_desc {
    int32_t k, ic, oc;
    const int32_t *w;          /* [oc][ic][k][k] quantised weights as int32 */
    const float *qbias;        /* [oc] clamp16(rint(b/(s*sw))) * f32(s*sw) */
    float in_scale;            /* f32 of this batch's input scale */
    int32_t in_zero;           /* this batch's input zero point (may be < -128) */
    float ss;                  /* f32(in_scale * weight_scale) */
    float acc_lo, acc_hi;      /* (-2^17 - zero)*s*sw , (2^17-1 - zero)*s*sw */
    float add_lo, add_hi;      /* same with 2^19 */
    int32_t relu;
} sesrq_calib_conv_desc;
/* out = act(conv) (+ skip, the float long residual added after the activation; may be NULL) */
int sesrq_calib_conv(const sesrq_calib_conv_desc *d, const float *in, const float *skip, float *out,
                     int N, int H, int W, void *stream);
/* min and max of a device fp32 tensor -> out_min_max[0..1] (device); scratch8: 8 bytes of device scratch */
int sesrq_calib_minmax(const float *x, size_t n, float *out_min_max, void *scratch8, void *stream);
/* Entropy (KL) calibration variant -- no reference counterpart: the reference's test.py keeps min/max only; BASELINE's north
 * star asks for KL-entropy activation ranges.  Adds the histogram of a device fp32 tensor over [lo, hi) in `bins` (2..4096)
 * equal bins to hist[bins] (device, uint32; the caller zeroes it): bin = floor((x - lo) * f32(bins / (hi - lo))) clamped to
 * [0, bins-1], NaNs skipped.  The range search over the histogram is host code (sesrq/calibrate.py: entropy_range). */
int sesrq_calib_histogram(const float *x, size_t n, float lo, float hi, int bins, uint32_t *hist, void *stream);
/* (clamp8(rint(x/scale + zero)) - zero) * scale  (quan_func.py:207,215) */
int sesrq_calib_fakequant(const float *in, float *out, size_t n, float scale, int zero, void *stream);

/* ---- host scalar code of the path (load time) -------------------------------------- */

/* quan_layer_between_const (myQL/quan_func.py:495-515): r -> (M, n), truncating. */
int sesrq_requant_const(double r, int data_bit, int shift_max, uint32_t *M, uint32_t *n);
/* quantize_symmetrical_by_tensor (myQL/quan_func.py:58-71): per-tensor symmetric INT8. */
int sesrq_quantize_weight(const float *w, size_t count, int width, int8_t *wq, double *scale);
/* The same per OUTPUT CHANNEL (w: [oc][per_oc]): scale_oc[o] = 2 * absmax(w[o]) / (2^width - 1), wq = clamp(rint(w / f32(scale_oc[o]))).  No
 * reference counterpart (see sesrq_layer_desc.M_oc); an all-zero channel is an error like the reference's all-zero tensor. */
int sesrq_quantize_weight_per_channel(const float *w, int oc, size_t per_oc, int width, int8_t *wq, double *scale_oc);
/* quantize_bias_with_scale + add-constant (myQL/quan_func.py:402,448-449,481-486). */
int sesrq_add_const(const float *bias, const int8_t *wq, int oc, int per_oc, double s_in, int z_in,
                    double s_w, int bias_width, int32_t *out);
/* min/max -> scale/zero (test.py:185-217). */
int sesrq_calib_scale_zero(double min_val, double max_val, int width, double *scale, int *zero);

const char *sesrq_last_error(void);
int sesrq_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SESRQ_H */
