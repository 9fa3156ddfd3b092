/* mi355attn.h -- C ABI of libmi355attn.so: MI355X (gfx950 / CDNA4) forward kernels for the
 * changzy00/pytorch-attention block zoo.
 *
 * The reference has no FFI of its own: its boundary is the Python nn.Module surface.  Each
 * entry point below therefore replaces the *body of one reference forward()* (file:line cited per
 * function, paths relative to the reference checkout); the host-side nn.Module mirror in
 * pytorch-attention_amd/mi355attn/modules/ keeps the class name / ctor / state_dict / forward signature
 * and calls these through ctypes (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer on the calling thread's current
 *     HIP device; tensors are dense, row-major in the layout named per function, fp32 unless noted;
 *   - the caller owns every buffer (inputs, parameters, output, workspace).  The library allocates
 *     nothing, frees nothing and keeps no pointer after return;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream), with
 *     no implicit device synchronisation;
 *   - `ws` must hold at least mi355_<op>_workspace_bytes(...) bytes and be 16-byte aligned; it is scratch
 *     (contents undefined after return) and may be shared between calls on the same stream;
 *   - return 0 on success, <0 on failure: MI355_EINVAL (bad argument), MI355_EUNSUPPORTED (shape outside the
 *     kernel's envelope), MI355_EHIP (HIP runtime error), MI355_ESYNC (see mi355_sync_status).  mi355_last_error() returns a thread-local,
 *     NUL-terminated description of the last failure on this thread.  Nothing throws or aborts across the ABI.
 *   - precision: 0 = strict (3-way split-bf16 MFMA, fp32-class accuracy), 1 = fp16 MFMA operands with fp32
 *     accumulate (default of the modules; within the 1e-3 parity tolerance), 2 = bf16 MFMA operands (fast,
 *     outside the tolerance; reported separately).  Vector (non-MFMA) math is always fp32.
 */
#ifndef MI355ATTN_H
#define MI355ATTN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK            0
#define MI355_EINVAL       -1
#define MI355_EUNSUPPORTED -2
#define MI355_EHIP         -3
#define MI355_ESYNC        -4   /* an inter-workgroup exchange of an earlier launch timed out: see mi355_sync_status */
#define MI355_ERANGE (-5)  /* a finite value saturated to inf in an fp16 operand tensor: mi355_range_status */

#define MI355_PREC_STRICT 0
#define MI355_PREC_FP16   1
#define MI355_PREC_BF16   2

#define MI355_ACT_NONE 0
#define MI355_ACT_GELU 1
#define MI355_ACT_RELU 2   /* fp32-in engine only (mi355_linear_fwd, mi355_conv2d_tokens_fwd): relu with torch's NaN behaviour */

typedef void* mi355_stream_t; /* hipStream_t */

/* ---- library ------------------------------------------------------------------------------------- */
int         mi355_version(void);            /* ABI version, bumped on any signature change */
const char* mi355_last_error(void);         /* thread-local message of the last failing call */
/* Tuning knobs (read at launch; results never depend on them unless a key says so).  Scope: mi355_set_option / mi355_get_option act
 * on the calling thread's CURRENT HIP device -- every device has its own option block, and every launch reads the block of the device
 * it launches on, so a host that drives several GPUs from one process can tune (or, in a test, sabotage) one of them without the
 * others seeing it.  A device whose key was never set follows the process default, which mi355_set_default_option changes (what a
 * binding uses for promises it makes for every device, e.g. "ws_persistent").  The failure word of mi355_sync_status and the range
 * word of mi355_range_status are per device as well: a time-out or an fp16 overflow on one GPU never fails another GPU's next call.
 *   "chunk_images"  images per pool->scale chunk (0 = auto: ~200 MB of x per chunk so that a chunk's re-read is served
 *                   by the 256 MiB Infinity Cache -- default; a value >= B disables chunking);
 *   "nt"            bit0 = non-temporal loads, bit1 = non-temporal stores in the final streaming pass (default 3);
 *   "reverse"       1 = the final pass walks the batch backwards (most recently touched rows first), default 0;
 *   "se_single"     1 (default) = SE reads x once: 8 channel rows per workgroup stay in registers, the image's channel means are
 *                   exchanged between its workgroups as 8-byte {mean, tag} granules (one write-through store each, polled with
 *                   bounded sweeps); 0 = two passes (pool, then gate + scale).
 *   "cbam_single"   1 (default) = full CBAM (stage 0) reads x once: a workgroup keeps a band of rows of all channels in registers;
 *                   per-channel pools, image-level (avg, max) and the halo rows of the per-pixel statistics travel between the
 *                   bands of an image as 16-byte self-validating granules; 0 = three passes (pool, statistics, apply).
 *   "eca_single"    1 (default) = ECA reads x once: a workgroup keeps 8 channel rows in registers and re-sums the k-1 halo
 *                   rows next to them (served by the same XCD's L2), no exchange between workgroups; 0 = two passes.
 *   "ws_persistent" 0 (default) = the single-read kernels zero their exchange area in the workspace on every call; 1 = the caller
 *                   promises that a workspace handed to mi355_se_fwd / mi355_cbam_fwd is DEDICATED to that op and shape: nobody
 *                   else writes it between calls and it is not freed and re-allocated behind the library's back (call
 *                   mi355_workspace_forget before freeing it).  The library then remembers that the buffer has been zeroed for the
 *                   shape and skips the zeroing (43 MB per call at the C2 shape of CBAM); the launch tag and the ticket of the
 *                   SE / CBAM exchange live in the workspace itself (epoch + 1; reset by the last ticket draw of a launch).
 *   "zoo_single"    1 (default) = SimAM / SRM / GCT / LCT read x once when the shape allows; 0 = always two passes.
 *   "stem_direct"   1 (default) = mi355_conv2d_tokens_fwd runs the image layer of a narrow stem (NCHW input, Cin <= 4, Cout <= 64,
 *                   Cin*kh*kw*Cout <= 2048, e.g. 3 -> 16 3x3) on a direct fp32 kernel; 0 = implicit GEMM for every shape.
 *                   hipGraph stream capture: mi355_se_fwd / mi355_cbam_fwd record their single-read exchange kernels (nothing about
 *                   a launch lives on the host, so replays and eager launches may share a workspace; an unknown workspace gets its
 *                   zeroing kernel recorded with the launch).  The GCT and LCT entry points do the same since round 4 (their launch tag
 *                   moved into the workspace as well).  A kernel's first launch loads its code object, which HIP forbids inside a capture:
 *                   run every entry once eagerly before capturing it.
 *   "spin_limit"    poll budget (sweeps) of the exchange kernels before they give up and report through mi355_sync_status:
 *                   1024 .. 2^30 (default 2^22 ~ a second); 0 is accepted to force the time-out path in tests (every exchange then
 *                   fails on its first unsuccessful poll).
 *   "gemm_splitk"   1 (default) = mi355_linear16_ws_fwd may cut the tiles of the persistent kernel's last partial round along K
 *                   (K >= 1536 only); the fp32 summation order of those tiles then differs from the unsplit order, so a row's
 *                   result can depend (at rounding level) on where its tile falls in the launch; 0 = never split: every output row
 *                   is bit-identical whatever the batch around it.  Never split under hipGraph stream capture (a recorded launch
 *                   replays its per-launch flag tag).
 *   "gemm_pa"       1 (default) = fp32 (+ residual) outputs with M % 128 == 0, N % 256 == 0, K >= 640 run the two-accumulator
 *                   persistent kernel (the epilogue of tile i rides in the main loop of tile i + 1; no inter-workgroup exchange, so it
 *                   is capture-safe and every row is bit-identical whatever the batch); 0 = never.
 *   "gemm_variant"  0 (default) = kernel chosen by shape; 7 / 15 / 16 / 17 force the round-1 tile kernel / the persistent 256 x 256 kernel / the
 *                   two-accumulator persistent kernel / the one-wave-per-SIMD persistent kernel (bit-identical results; A/B partners of the
 *                   tests).  Values above 17 are refused.
 *   "gemm_pa16"     16-bit outputs with a long reduction (K >= 576) on the two-accumulator kernel: 1 (default) = those with a GELU epilogue
 *                   (its pieces hide most of the GELU that the persistent 256 x 256 kernel exposes), 2 = all, 0 = none.  Shorter
 *                   reductions (K = 256 .. 512) always take it ("gemm_pa" = 1).  Bit-identical results either way.
 *   "gemm_pa_block" 1 (default) = the two-accumulator kernel walks wide outputs (>= 8 column tiles) in blocks of 8 row x 4 column tiles, so
 *                   that the 32 workgroups of an XCD share 1.5 MB of X rows + 1.5 MB of W instead of all of W; 0 = column tile fastest.
 *   "gemm_pa_tail"  fp32 outputs on the two-accumulator kernel with K >= 1024 whose last round of tiles would be at most this many percent
 *                   full (default 10; 0 = off): the kernel runs whole rounds only and the left-over rows go to a small-tile kernel
 *                   spread over all CUs (a second launch on the same stream; same K order, bit-identical results).
 *   "mixer_fused"   1 (default) = the host mirror's MixerLayer runs its token-mixing half through mi355_mixer_token_fwd where the
 *                   geometry allows (N = 196, T % 32 == 0, C % 256 == 0); 0 = three launches (transposing LayerNorm, fc1, transposed fc2).
 *   "gemm_small"    1 (default) = mi355_linear_fwd without activation / LayerScale / residual whose output is less than an eighth of a round of
 *                   the engine's 128 x 128 tiles (a classifier head: 256 x 1000) runs on one-wave 16 x 32 tiles spread over every CU
 *                   (gemm_small.hip; same K order, bit-identical results); 0 = always the 128 x 128 engine.
 *   "gemm_w4"       1 (default) = 16-bit outputs without LayerScale / residual (a GELU epilogue is taken whenever "gemm_pa16" does not claim
 *                   it first), M % 256 == 0, N % 256 == 0, 576 <= K < 1536 (the qkv product of a ViT) run the one-wave-per-SIMD persistent kernel (gemm16_w4.hip: four waves x 128 x 128 outputs, 256
 *                   accumulation registers per lane, a hand-placed MFMA / ds_read / LDS-DMA stream; same K order, bit-identical results);
 *                   0 = the eight-wave persistent kernel (gemm16_p8.hip) as before.  "gemm_variant" 17 forces the kernel on any shape it takes.
 *   "attn_nw"       waves per workgroup of the ViT attention core at 193 .. 208 tokens (13 query tiles): 8 (default; five waves carry two
 *                   tiles, three carry one) or 7 (six carry two, one carries one).  Bit-identical results (a tile's arithmetic does not depend on
 *                   its wave); measured equal in round 5.  Range 7 .. 8.
 *   "mlp_tt4"       mi355_mlp_fused_fwd / mi355_proj_mlp_fused_fwd at C = 64: 1 = eight waves with four 16-token tiles each (256 registers per
 *                   lane, a weight fragment read from LDS feeds four MFMAs); 0 (default) = sixteen waves with two tiles (128 registers).
 *                   Bit-identical results.
 *   "mixer_stats"   mi355_mixer_token_fwd at C = 512: 1 = the LayerNorm row statistics are computed inside the token kernel (every workgroup
 *                   of an image reads the image's rows once more, the second request served by L2: the two workgroups of an image run
 *                   on one XCD); 0 (default) = by a separate statistics pass over x.  Bit-identical results; measured in round 5: the two
 *                   cost the same (MixerLayer 0.428-0.433 vs 0.432-0.437 ms on one box), so the simpler pre-pass stays the default.
 *   "mixer_early"   mi355_mixer_token_fwd: 1 = the kernel issues the residual loads of its epilogue in two batches ahead of their stores (two
 *                   exposed round trips instead of thirteen; bit-identical results); 0 (default) = a token tile's loads right before its stores.
 *   "lpi_patch"     1 (default) = mi355_lpi_fwd / mi355_ln_lpi_fwd at 14 x 14 tokens with C % 32 == 0 run the patch kernel (a lane owns a
 *                   2 x 2 token patch of one channel quad, taps in scalar registers, fused multiply-adds); 0 = the general kernel
 *                   (separately rounded products and sums: the two agree to ~1e-7 relative, not bit for bit).
 *   "ln_fold"       1 = the ViT encoder chain of the host mirror folds its LayerNorms into the neighbouring GEMMs
 *                   (mi355_linear16_emit_fwd / mi355_ln_finalize_fwd / mi355_linear16_lnfold_fwd); 0 (default) = one LayerNorm launch
 *                   each.  Measured in round 4: the fold costs more in the GEMM epilogues than the 36 us launches it removes.
 *   "gemm_wreg"     1 (default) = fp32 (+ residual) outputs of square short products, N = K = 256 or 384, M >= 32, no activation / LayerScale
 *                   (XCiT's proj, CSWin stage-3 proj) run the weight-stationary-in-registers streaming kernel (gemm16_wreg.hip: eight waves, each
 *                   holds its N / 8 columns of W as MFMA fragments for the whole kernel; X, residual and Y cross HBM once).  Bit-identical
 *                   results; 0 = the tile kernels as before.
 *   "xca_tr"        1 (default) = mi355_xca16_fwd with 16-bit qkv and N <= 224 tokens forms the d x d covariance on the 16-bit matrix pipe
 *                   (fp32 accumulation; the products of 16-bit operands are exact in fp32) from one transposed LDS image of q and k
 *                   (xca_tr_kernel); 0 = the token-streaming kernel with exact-fp32 MFMAs on fp32 copies (any N; differs at the 1e-7 level:
 *                   the summation order inside the matrix instruction).
 *   "mlp_wide"      1 = mi355_mlp_fused_fwd also takes C = 256 and C = 384 with hidden = 4 C (CSWin stage 3, XCiT-S: mlp_wide.hip -- producer waves
 *                   form gelu(W1' LN(x) + b1') 128 hidden units at a time, consumer waves the second product one slice behind, weight fragments go
 *                   global -> VGPR, the token rows are LayerNorm'ed into LDS once per 98-row step); 0 (default) = those shapes are
 *                   MI355_EUNSUPPORTED and the host mirror composes LayerNorm + two GEMMs.  Built, parity-tested and measured in round 6: 257 us
 *                   against 224 us for the three launches at C = 384, 157 against 137 at C = 256 (profiles/r06_mlp_wide.md: with 98 rows per
 *                   step every weight byte is used for 98 rows, the 256 x 128 GEMM tiles use it for 256 -- the fused kernel pulls as many bytes
 *                   from L2 as the two GEMMs and cannot hold more rows' accumulators), hence opt-in.
 *   "gemm_wst"      1 / 2 = 16-bit outputs with K = 768 and N % 192 == 0 (the qkv / fc1 products of ViT-Base) keep a 192-column slab of W in the
 *                   registers of a persistent workgroup and stream X through LDS once per slab (gemm16_wst.hip; 1 = products without activation,
 *                   2 = GELU epilogues too).  A row's K halves are added as two chains: results differ from the tile kernels by at most one unit
 *                   of the 16-bit output.  3 / 4 = the same products (N % 256 == 0, M % 32 == 0) on the one-wave-per-SIMD kernel that keeps W in
 *                   AGPRs as MFMA operands: one chain per row, bit-identical to the tile kernels (3 = without activation, 4 = GELU too).
 *                   0 (default): both measured slower than the tile kernels (profiles/r06_gemm_wst.md: qkv 200-219 vs 174-180 us, fc1 281-337
 *                   vs 275-288).
 *   "gemm_wslab"    1 (default) = 16-bit outputs with K = 256 / 384 / 512 and N a multiple of the slab width (256 / 192 / 256 columns) keep a column slab
 *                   of W in the registers of a persistent workgroup and stream 32-row tiles of X (gemm16_wslab.hip) when the epilogue is a GELU or M is
 *                   not a multiple of 256 -- the products it measured faster on (5-12 % / 88-96 -> 65-69 us; profiles/r06_gemm_wslab.md); 2 = every
 *                   product it takes (plain epilogues: a tie); 0 = the tile kernels.  Bit-identical results either way.
 *   "range_fallback" 1 (default) = the host mirror's modules re-run a forward whose fp16 operands saturated in precision 0 (one warning;
 *                   mi355_range_arm / mi355_range_wait below: no device synchronisation unless it fires); 0 = they do not wait and the NEXT call
 *                   reports MI355_ERANGE (the round-3 contract).  Host policy: the C entries themselves never re-run anything.
 * Unknown key or a value outside the key's range -> MI355_EINVAL. */
int         mi355_set_option(const char* key, long value);          /* current device */
int         mi355_set_default_option(const char* key, long value);  /* process default: devices without an own setting */
long        mi355_get_option(const char* key);                      /* what a launch on the current device would read */
/* In-process kernel tally: between mi355_trace_begin() and mi355_trace_end() every launch of the instrumented kernels (the GEMM engine,
 * the attention cores, LayerNorm / cast / im2col passes, the XCiT kernels) on the calling thread's current device is bracketed by a
 * pair of HIP events on its launch stream.  mi355_trace_end waits for those launches and writes one line per kernel tag,
 *   "count\ttotal_us\tmin_us\tmax_us\ttag\n"   (largest total first; the tag = kernel name + the shape parameters that tell its
 * launches apart), NUL-terminated, truncated to report_bytes; it returns the size of the full report (without the NUL).  A report that did
 * not fit -- or report == NULL -- is kept: size a buffer from the return value (+ 1) and call again; it is dropped once handed out whole
 * or at the next mi355_trace_begin.  One trace at a time per process; launches under hipGraph stream capture are not recorded.  What
 * bench.py fills roofline.dominant_kernel from. */
int         mi355_trace_begin(void);
long        mi355_trace_end(char* report, size_t report_bytes);
/* Drop what the library remembers about workspaces inside [ws, ws + ws_bytes) ("ws_persistent"): call before freeing or
 * repurposing such a buffer.  The next call that uses the memory zeroes its exchange area again. */
int         mi355_workspace_forget(const void* ws, size_t ws_bytes);
/* Failure report of the single-read exchange kernels (SE, CBAM, GCT / LCT gates).  Their inter-workgroup polls are bounded
 * (option "spin_limit", sweeps; default 1 << 22 ~ a second); a poll that runs out stores a code into a pinned host word that
 * the library reads WITHOUT a device synchronisation.  MI355_OK = nothing pending.  MI355_ESYNC = some launch that has already
 * executed produced invalid output (text in mi355_last_error); the condition is cleared by the report.  The same check runs at
 * the start of every later mi355_se_fwd / mi355_se_ex_fwd / mi355_cbam_fwd / gate call, which then fails instead of launching.
 * Launches also refuse shapes whose per-image workgroup set cannot be resident at once (they take the multi-pass kernels). */
int         mi355_sync_status(void);
/* fp16 range guard.  The default operand format (precision 1) is IEEE half: a finite fp32 value of magnitude >= 65520 becomes inf where
 * the fp32 reference stays finite.  The producers of fp16 operand tensors -- mi355_cast16_fwd, mi355_layernorm16_fwd and every GEMM
 * epilogue with a 16-bit output (mi355_linear16*_fwd, mi355_mhsa_fwd's qkv, the patch embedding) -- watch the magnitudes they convert
 * and, on the first saturation, store a code into a pinned host word.  mi355_range_status() reads that word WITHOUT a device
 * synchronisation: MI355_OK = nothing pending; MI355_ERANGE = some launch that has already executed produced inf from finite values
 * (cleared by the report).  A host that wants certainty for a forward synchronises the stream first.  Remedy: run the module in
 * precision 0 (strict: bf16 hi/lo split, fp32 range) or precision 2 (bf16).  bf16 operands are never flagged. */
int         mi355_range_status(void);
/* "Is the range word final for what I have launched?" without draining the device (round 6).  mi355_range_arm(on) arms the calling thread's
 * current device (on >= 1) or disarms it (0): while armed, the entries that launch fp16 producers are counted.  mi355_range_wait() records
 * (if it has not happened yet) ONE event behind the last producer, synchronises on it -- launches queued behind that event keep running --
 * and returns the range status (MI355_OK / MI355_ERANGE, cleared by the report); with no producer since the arm it returns the status
 * without waiting.  on = 1 + k (k >= 1) carries a prediction, "the k-th producer entry since this call is the last one" (what
 * mi355_range_launches() returned after the caller's previous forward of the same module): the event is then recorded in front of the
 * first launch that follows that producer, so the wait does not cover the non-reporting tail of the forward (attention core, fp32-output
 * projections).  A wrong prediction costs that slack, never correctness.  Not for use under hipGraph capture.  This is what the host
 * mirror's modules use to give the reference's behaviour on large activations by default: a forward whose fp16 operands saturated is run
 * again in precision 0 (option "range_fallback" = 1, per device; 0 = no wait, the next call reports MI355_ERANGE). */
int         mi355_range_arm(int on);
int         mi355_range_wait(void);
long        mi355_range_launches(void);   /* producer entries since the last mi355_range_arm(on >= 1) on this device */

/* ---- channel / spatial attention family: NCHW fp32, HBM-bound ------------------------------------ */

/* SELayer.forward  (attention_mechanisms/se_module.py:29-33)
 *   y = x * sigmoid(W2 relu(W1 mean_hw(x)));  x,y (B,C,H,W);  w1 (Cr,C) = fc.0.weight;  w2 (C,Cr) = fc.2.weight. */
size_t mi355_se_workspace_bytes(int B, int C, int H, int W);
int    mi355_se_fwd(const float* x, const float* w1, const float* w2, float* y,
                    int B, int C, int Cr, int H, int W, void* ws, size_t ws_bytes, mi355_stream_t stream);
/* SE with the options its copies inside the reference's CNNs use (cnns/efficientnet.py:13-28 and mnasnet.py:11-26 Linear
 * layers WITH bias; mobilenetv3.py:15-30 / moat.py:18-33 without; ghostnet.py:48-65 1x1 convs with bias + hard-sigmoid gate):
 *   y = x * gate(w2 relu(w1 mean_hw(x) + b1) + b2),  gate = 0: sigmoid, 1: hard sigmoid relu6(z + 3) / 6;  b1 (Cr), b2 (C) may be NULL.
 * Same kernels, workspace (mi355_se_workspace_bytes) and single-read / two-pass selection as mi355_se_fwd. */
int         mi355_se_ex_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y, int B, int C,
                            int Cr, int H, int W, int gate, void* ws, size_t ws_bytes, mi355_stream_t stream);

/* ECALayer.forward  (attention_mechanisms/eca.py:26-30; kernel-size rule :21-22 stays on the host)
 *   y = x * sigmoid(conv1d_k(mean_hw(x)) across the channel axis, zero pad (k-1)/2, no bias); wconv (k,) = conv.weight. */
size_t mi355_eca_workspace_bytes(int B, int C, int H, int W);
int    mi355_eca_fwd(const float* x, const float* wconv, float* y,
                     int B, int C, int k, int H, int W, void* ws, size_t ws_bytes, mi355_stream_t stream);

/* CBAM.forward  (attention_mechanisms/cbam.py:56-59 = SpatialAttention :43-48 o ChannelAttention :31-35)
 *   w1 (Cr,C) = ca.fc.0.weight, w2 (C,Cr) = ca.fc.2.weight (1x1 convs, no bias), wconv (2,ks,ks) = sa.conv.weight.
 *   stage: 0 = full CBAM, 1 = ChannelAttention only (wconv ignored), 2 = SpatialAttention only (w1,w2 ignored). */
size_t mi355_cbam_workspace_bytes(int B, int C, int H, int W);
int    mi355_cbam_fwd(const float* x, const float* w1, const float* w2, const float* wconv, float* y,
                      int B, int C, int Cr, int ks, int H, int W, int stage,
                      void* ws, size_t ws_bytes, mi355_stream_t stream);

/* DoubleAttention.forward  (attention_mechanisms/double_attention.py:32-48)
 *   wA (cm,C) bA (cm) | wB (cn,C) bB (cn) | wV (cn,C) bV (cn) | wP (C,cm) bP (C);  x,y (B,C,H,W). */
size_t mi355_double_attn_workspace_bytes(int B, int C, int cm, int cn, int H, int W);
/* The same for one precision mode (any mode fits in the size above): the 16-bit modes at c_m = c_n = 128, C in {128, 256} run in two
 * passes over the image (x read once, y written once, a 16-bit softmax(V) in between) and need a fraction of it.  Option "da_fused" = 0
 * selects the seven-launch pipeline for every shape. */
size_t mi355_double_attn_ws_bytes(int B, int C, int cm, int cn, int H, int W, int precision);
int    mi355_double_attn_fwd(const float* x, const float* wA, const float* bA, const float* wB, const float* bB,
                             const float* wV, const float* bV, const float* wP, const float* bP, float* y,
                             int B, int C, int cm, int cn, int H, int W, int precision,
                             void* ws, size_t ws_bytes, mi355_stream_t stream);

/* ---- the rest of the channel-attention zoo: per-channel statistic -> tiny transform -> broadcast scale ------------
 * NCHW fp32, HBM-bound, x read once / y written once when HW % 4 == 0, HW <= 4096 and C % 8 == 0 (8 channel rows per workgroup stay
 * in registers; GCT / LCT exchange one number per channel between the workgroups of an image exactly like mi355_se_fwd), two plain
 * passes otherwise.  One workspace size serves all five: mi355_chan_stat_workspace_bytes(B, C).  "zoo_single" = 0 forces two passes.
 *   simam      simam.py:32-41      y = x * sigmoid(d / (4 * (sum_hw d / (HW-1) + e_lambda)) + 0.5),  d = (x - mean_hw x)^2
 *   srm        srm.py:23-34        g = sigmoid(BatchNorm1d_eval(cfc[c,0] * mean_hw x + cfc[c,1] * std_hw x)), std unbiased; cfc (C,2)
 *   gct_gauss  gct.py:23-30        g = exp(-c/2 * yn^2), yn = (m - mean_c m) / sqrt(mean_c m^2 - (mean_c m)^2 + eps), m = mean_hw x
 *   lct        lct.py:29-39        g = sigmoid(w[c] * yn + b[c]), yn as above but over the C/groups channels of the channel's group
 *   gct        gate_channel_module.py:32-50   l2: e = sqrt(sum_hw x^2 + eps) * alpha, g = 1 + tanh(e * gamma / sqrt(mean_c e^2 + eps) + beta)
 *                                             l1: e = sum_hw |x| * alpha (x itself when after_relu), g = 1 + tanh(e * gamma / (mean_c |e| + eps) + beta) */
size_t mi355_chan_stat_workspace_bytes(int B, int C);
int mi355_simam_fwd(const float* x, float* y, int B, int C, int H, int W, float e_lambda, void* ws, size_t ws_bytes, mi355_stream_t stream);
int mi355_srm_fwd(const float* x, const float* cfc, const float* bn_weight, const float* bn_bias, const float* bn_mean,
                  const float* bn_var, float bn_eps, float* y, int B, int C, int H, int W, void* ws, size_t ws_bytes,
                  mi355_stream_t stream);
int mi355_gct_gauss_fwd(const float* x, float* y, int B, int C, int H, int W, float c, float eps, void* ws, size_t ws_bytes,
                        mi355_stream_t stream);
int mi355_lct_fwd(const float* x, const float* w, const float* b, float* y, int B, int C, int groups, int H, int W, float eps,
                  void* ws, size_t ws_bytes, mi355_stream_t stream);
int mi355_gct_fwd(const float* x, const float* alpha, const float* gamma, const float* beta, float* y, int B, int C, int H, int W,
                  float epsilon, int mode_l1, int after_relu, void* ws, size_t ws_bytes, mi355_stream_t stream);

/* ---- gates built from axis reductions (second group) ---------------------------------------------------------
 * x, y (B,C,H,W) fp32 contiguous; caller workspace of mi355_axis_attn_workspace_bytes (mi355_bam_workspace_bytes for BAM), no state
 * kept between calls.  BatchNorm layers run in eval mode and are passed FOLDED: scale = weight / sqrt(running_var + eps),
 * shift = bias - running_mean * scale (+ scale * the preceding convolution's bias where noted).
 *   gc        gc_module.py:30-43   ctx[b,c] = sum_hw x[b,c,hw] * (conv_w . x[b,:,hw] + conv_b)   (the declared softmax is never applied),
 *                                  y = x + W2 relu(LayerNorm_Cr(W1 ctx + b1)) + b2;  w1 (Cr,C), w2 (C,Cr), ln_w / ln_b (Cr)
 *   coordatt  coordatten.py:30-44  p = [mean_w x ; mean_h x] (B,C,H+W); h = relu(bn(W1 p + b1)) (hidden rows);
 *                                  y = x * (Wh h[:, :H] + bh)[b,c,i] * (Ww h[:, H:] + bw)[b,c,j]   (no sigmoid in the reference);
 *                                  w1 (hidden,C), wh / ww (C,hidden)
 *   triplet   triplet_attention.py:58-63   three AttentionGates s = sigmoid(relu(bn(conv_kxk([mean, max])))) over the pooled axis:
 *                                  s_ch[b,c,i] from pooling over w, s_cw[b,c,j] over h, s_hw[b,i,j] over c;
 *                                  y = (x s_ch + x s_cw + x s_hw) / 3.  w_* (2,k,k); affine = 6 floats {scale, shift} x {ch, cw, hw},
 *                                  shift including scale * conv bias.
 *   bam       bam.py:63-71         y = x + x * sigmoid(cg[b,c] + sg[b,hw]);  cg = bn1d(W2 relu(W1 mean_hw x + b1) + b2),
 *                                  sg = bn(conv3(relu(bn(dconv2(relu(bn(dconv1(conv1 x)))))))), dconv = 3x3, dilation = padding.
 *                                  Parameters as an array of MI355_BAM_NPARAMS device pointers (host array), indexed by the enum. */
size_t mi355_axis_attn_workspace_bytes(int B, int C, int H, int W);
int mi355_gc_fwd(const float* x, const float* conv_w, const float* conv_b, const float* w1, const float* b1, const float* ln_w,
                 const float* ln_b, const float* w2, const float* b2, float* y, int B, int C, int Cr, int H, int W, float ln_eps,
                 void* workspace, size_t workspace_bytes, mi355_stream_t stream);
int mi355_coordatt_fwd(const float* x, const float* w1, const float* b1, const float* bn_scale, const float* bn_shift, const float* wh,
                       const float* bh, const float* ww, const float* bw, float* y, int B, int C, int hidden, int H, int W,
                       void* workspace, size_t workspace_bytes, mi355_stream_t stream);
int mi355_triplet_fwd(const float* x, const float* w_ch, const float* w_cw, const float* w_hw, const float* affine, float* y, int B, int C,
                      int H, int W, int ksize, void* workspace, size_t workspace_bytes, mi355_stream_t stream);
enum {
    MI355_BAM_FC1_W = 0,       /* (Cr,C)      channel_attn.mlp[0].weight */
    MI355_BAM_FC1_B,           /* (Cr)        channel_attn.mlp[0].bias */
    MI355_BAM_FC2_W,           /* (C,Cr)      channel_attn.mlp[2].weight */
    MI355_BAM_FC2_B,           /* (C)         channel_attn.mlp[2].bias */
    MI355_BAM_BN1D_SCALE,      /* (C)         channel_attn.bn folded */
    MI355_BAM_BN1D_SHIFT,      /* (C) */
    MI355_BAM_CONV1_W,         /* (Cr,C)      spatial_attn.conv1.weight */
    MI355_BAM_CONV1_B,         /* (Cr) */
    MI355_BAM_DCONV1_W,        /* (Cr,Cr,3,3) spatial_attn.conv2[0].weight */
    MI355_BAM_DCONV1_SCALE,    /* (Cr)        conv2[1] folded */
    MI355_BAM_DCONV1_SHIFT,    /* (Cr)        conv2[1] folded + scale * conv2[0].bias */
    MI355_BAM_DCONV2_W,        /* (Cr,Cr,3,3) spatial_attn.conv2[3].weight */
    MI355_BAM_DCONV2_SCALE,    /* (Cr)        conv2[4] folded */
    MI355_BAM_DCONV2_SHIFT,    /* (Cr) */
    MI355_BAM_CONV3_W,         /* (Cr)        spatial_attn.conv3.weight * bn scale */
    MI355_BAM_CONV3_B,         /* (1)         spatial_attn.bn folded + scale * conv3.bias */
    MI355_BAM_NPARAMS
};
size_t mi355_bam_workspace_bytes(int B, int C, int Cr, int H, int W);
int mi355_bam_fwd(const float* x, const float* const* params, float* y, int B, int C, int Cr, int H, int W, int dilation,
                  void* workspace, size_t workspace_bytes, mi355_stream_t stream);
/* The helper classes of these modules on their own (the reference exposes them as importable nn.Modules):
 *   mi355_bam_gates_fwd        ChannelGate.forward bam.py:28-33 -> cg (B,C) and / or SpatialGate.forward :53-59 -> sg (B,H*W), both
 *                              BEFORE their expand_as (a broadcast view is the caller's); either output may be NULL.  Parameters and
 *                              workspace as for mi355_bam_fwd.
 *   mi355_zpool_fwd            ZPool.forward triplet_attention.py:31-36: y (B,2,H,W) = [mean over channels, max over channels].
 *   mi355_attention_gate_fwd   AttentionGate.forward :38-49: y = x * sigmoid(relu(bn(conv_kxk(ZPool(x))))); w (2,k,k), affine[0] / [1] =
 *                              eval-BatchNorm scale / shift with the conv bias folded in; workspace of
 *                              mi355_attention_gate_workspace_bytes.
 * BasicConv2d.forward (:19-29, conv -> bn -> relu) is mi355_conv2d_tokens_fwd with act = MI355_ACT_RELU and the BatchNorm folded,
 * followed by mi355_tokens_to_nchw_axpy_fwd. */
int mi355_bam_gates_fwd(const float* x, const float* const* params, float* cg, float* sg, int B, int C, int Cr, int H, int W, int dilation,
                        void* workspace, size_t workspace_bytes, mi355_stream_t stream);
int mi355_zpool_fwd(const float* x, float* y, int B, int C, int H, int W, mi355_stream_t stream);
size_t mi355_attention_gate_workspace_bytes(int B, int H, int W);
int mi355_attention_gate_fwd(const float* x, const float* w, const float* affine, float* y, int B, int C, int H, int W, int ksize,
                             void* workspace, size_t workspace_bytes, mi355_stream_t stream);

/*   sk        sk_module.py:41-56   u1 = relu(bn(conv3x3_grouped(x))), u2 = relu(bn(conv3x3_grouped_dilation2(x))), s = mean_hw(u1 + u2),
 *                                  z = relu(bn1d(fc s)), [a, b] = softmax over the two branches of [fc1 z, fc2 z], y = u1 a + u2 b.
 *                                  x (B,Cin,H,W), y (B,planes,H,W); planes / groups in {1,2,4,8,16}; parameters as an array of
 *                                  MI355_SK_NPARAMS device pointers. */
enum {
    MI355_SK_CONV3_W = 0,      /* (planes, Cin/groups, 3, 3)  split_3x3[0].weight */
    MI355_SK_CONV3_SCALE,      /* (planes)   split_3x3[1] folded */
    MI355_SK_CONV3_SHIFT,      /* (planes)   split_3x3[1] folded + scale * split_3x3[0].bias */
    MI355_SK_CONV5_W,          /* (planes, Cin/groups, 3, 3)  split_5x5[0].weight (3x3, dilation 2) */
    MI355_SK_CONV5_SCALE,
    MI355_SK_CONV5_SHIFT,
    MI355_SK_FC_W,             /* (d, planes) fc[0].weight */
    MI355_SK_FC_B,             /* (d) */
    MI355_SK_FC_BN_SCALE,      /* (d)        fc[1] folded */
    MI355_SK_FC_BN_SHIFT,      /* (d) */
    MI355_SK_FC1_W,            /* (planes, d) */
    MI355_SK_FC1_B,            /* (planes) */
    MI355_SK_FC2_W,            /* (planes, d) */
    MI355_SK_FC2_B,            /* (planes) */
    MI355_SK_NPARAMS
};
size_t mi355_sk_workspace_bytes(int B, int planes, int H, int W);
int mi355_sk_fwd(const float* x, const float* const* params, float* y, int B, int Cin, int planes, int groups, int d, int H, int W,
                 void* workspace, size_t workspace_bytes, mi355_stream_t stream);

/* DANet dual attention (dual_attention.py).  CAM :35-42: y = beta * softmax(X X^T) X + x with X = x viewed as (C, HW) per image; beta is
 * a 1-element device array; H*W and C multiples of 4; workspace of mi355_cam_workspace_bytes (the Gram matrices).  The logits are
 * unscaled sums over HW, so X X^T always runs in the fp32-class split-bf16 mode; `precision` selects the operand format of attn . X.
 * PAM :20-28 is composed by the caller from mi355_conv2d_tokens_fwd (the three 1x1 convs as one token-major GEMM),
 * mi355_sdpa_general_fwd (one head of width C, scale 1) and mi355_tokens_to_nchw_axpy_fwd:
 *   y[b,c,p] = alpha[0] * tokens[b,p,c] + x[b,c,p]      tokens (B,HW,C), x / y (B,C,HW), alpha a 1-element device array;
 *   alpha == NULL means 1, x == NULL means 0 (a plain token-major -> NCHW transpose). */
size_t mi355_cam_workspace_bytes(int B, int C);
int mi355_cam_fwd(const float* x, const float* beta, float* y, int B, int C, int H, int W, int precision, void* workspace,
                  size_t workspace_bytes, mi355_stream_t stream);
int mi355_tokens_to_nchw_axpy_fwd(const float* tokens, const float* x, const float* alpha, float* y, int B, int HW, int C,
                                  mi355_stream_t stream);

/* ---- glue of the remaining multi-head-attention copies ------------------------------------------------------------
 * mi355_dwconv_nchw_tokens_fwd: depth-wise ks x ks convolution (stride 1, zero padding (ks-1)/2) of an NCHW map, written token-major:
 *   y[b, i*W+j, c] = bias[c] + sum_uv weight[c,u,v] * x[b,c,i+u-pad,j+v-pad]     (cvt.py:48-50 with the BatchNorm folded by the caller)
 * mi355_qk_logits_fwd: unscaled attention logits, logits[b,h,i,j] = q[b,i,h*d:(h+1)*d] . k[b,j,h*d:(h+1)*d]; q (B,Nq,*) / k (B,Nkv,*) fp32
 *   with row strides ldq / ldk (views into a fused projection), logits (B,heads,Nq,Nkv) fp32.
 * mi355_topk_mask_fwd: in place on `rows` rows of length N: the k largest entries of a row become 0, all others -1e30 -- the additive
 *   bias that makes mi355_sdpa_general_fwd the k-NN attention of kvt.py:83-89 (top-k of the scaled logits == top-k of the unscaled
 *   ones).  N <= 4096. */
/* P2T's pooled key/value source (p2t.py:76-83) on the token layout.  mi355_adaptive_pool_tokens_fwd: F.adaptive_avg_pool2d of the
 * (H x W) token grid x (B, H*W, C) to y (B, OH*OW, C).  mi355_dwconv3x3_tokens_residual_fwd: y = x + dwconv3x3(x) + bias on a (H x W)
 * token grid (weight (C,3,3)); y points into a longer token sequence whose images are y_batch_stride floats apart (the concatenation
 * of the pyramid levels, p2t.py:82). */
int mi355_adaptive_pool_tokens_fwd(const float* x, float* y, int B, int H, int W, int C, int OH, int OW, mi355_stream_t stream);
int mi355_dwconv3x3_tokens_residual_fwd(const float* x, const float* weight, const float* bias, float* y, int B, int H, int W, int C,
                                        long y_batch_stride, mi355_stream_t stream);
int mi355_dwconv_nchw_tokens_fwd(const float* x, const float* weight, const float* bias, float* y, int B, int C, int H, int W, int ks,
                                 mi355_stream_t stream);
int mi355_qk_logits_fwd(const float* q, const float* k, float* logits, int B, int heads, int Nq, int Nkv, int head_dim, int ldq, int ldk,
                        int precision, mi355_stream_t stream);
int mi355_topk_mask_fwd(float* logits, long rows, int N, int k, mi355_stream_t stream);

/* ---- dense building blocks used by the transformer blocks ---------------------------------------- */

/* nn.Linear (+ optional GELU, LayerScale, residual):  Y = resid + gamma * act(X W^T + bias)
 *   X (M,K) row-major with row stride ldx, W (N,K) row-major (nn.Linear layout), Y (M,N) row stride ldy.
 *   bias (N) / gamma (N) / resid (M,N; row stride ldy) may be NULL.  Covers ViT.py:58-65,81,87;
 *   cswin.py:185,192,40-48; xcit.py:32-38,248,263; mlp_mixer.py:26-33. */
int mi355_linear_fwd(const float* X, const float* W, const float* bias, const float* gamma, const float* resid,
                     float* Y, int M, int N, int K, int ldx, int ldy, int act, int precision,
                     mi355_stream_t stream);

/* Batched left-multiplication used by MLP-Mixer token mixing (mlp_mixer.py:47):
 *   Y_b = resid_b + act(W X_b + bias 1^T)   with W (T,N), X_b (N,C), Y_b (T,C), bias (T), b = 0..B-1. */
int mi355_token_mix_fwd(const float* W, const float* X, const float* bias, const float* resid, float* Y,
                        int B, int T, int N, int C, int act, int precision, mi355_stream_t stream);

/* nn.LayerNorm over the last axis, eps inside the sqrt (ViT.py:111-114, cswin.py:139, xcit.py:271). */
int mi355_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y,
                        int rows, int cols, float eps, mi355_stream_t stream);

/* ---- 16-bit activation dataflow (precision 1 = IEEE half, 2 = bfloat16) ----------------------------------------
 * The composite blocks keep activations that only ever feed an MFMA in the MFMA operand format IN HBM: the rounding
 * point is the same as in the fp32-in entry points (operands are rounded to 16 bit exactly once, accumulation and
 * epilogues stay fp32), so results are bit-identical, but GEMM operand traffic halves and tiles go HBM -> LDS by DMA.
 * `void*` 16-bit buffers hold _Float16 (precision 1) or bfloat16 (precision 2) elements. */
int mi355_cast16_fwd(const float* src, void* dst16, size_t n, int precision, mi355_stream_t stream);
int mi355_layernorm16_fwd(const float* x, const float* weight, const float* bias, void* y16, int rows, int cols, float eps,
                          int precision, mi355_stream_t stream);
/* Y = resid + gamma * act(X16 W16^T + bias); Y is fp32 (out16 = 0) or 16-bit (out16 = 1); bias/gamma/resid fp32 or NULL.
 * Needs K % 64 == 0, N % 4 == 0, ldx % 8 == 0 (other shapes: cast back and use mi355_linear_fwd). */
int mi355_linear16_fwd(const void* X16, const void* W16, const float* bias, const float* gamma, const float* resid, void* Y,
                       int M, int N, int K, int ldx, int ldy, int act, int out16, int precision, mi355_stream_t stream);
/* The same with a scratch buffer of mi355_linear16_workspace_bytes(M, N, K) bytes (0 = this shape needs none): the persistent
 * 256 x 256 kernel then cuts the tiles of its last, partially filled round along K across the otherwise idle CUs (partial
 * accumulators travel through `ws`; N = 768 outputs of ViT-Base take 2.33 rounds instead of 3).  Results are bit-identical to
 * mi355_linear16_fwd only where the split does not apply; with it the fp32 summation order over K changes (covered by the same
 * tolerance).  ws may be NULL / too small: identical to mi355_linear16_fwd. */
size_t mi355_linear16_workspace_bytes(int M, int N, int K);
/* Y16 = act(T(X32) W16^T + bias) with the fp32 -> 16-bit cast of the activation inside the GEMM's X staging (round 6, gemm16_wslab.hip): one
 * launch and no 16-bit copy of X in HBM where a caller would otherwise run mi355_cast16_fwd + mi355_linear16_fwd (XCA.forward on an fp32
 * input, xcit.py:251; Mlp.forward, ViT.py:59); the same bits, the same fp16 range report for X (code 1) and for Y (code 3).
 * Built for K = 256 / 384 / 512, N a multiple of 256 / 384 / 256, ldx % 4 == 0, ldy % 8 == 0 and at least 4 x 32 rows per resident workgroup;
 * anything else, or option "gemm_wslab" = 0: MI355_EUNSUPPORTED with nothing launched and no error text -- cast and call mi355_linear16_fwd. */
int mi355_linear16_x32_fwd(const float* X32, const void* W16, const float* bias, void* Y16, int M, int N, int K, int ldx, int ldy, int act,
                           int precision, mi355_stream_t stream);
/* Y = resid + X16 W16^T + bias (fp32) AND the LayerNorm statistics of every row of Y: row_stats[2 m] = mean, row_stats[2 m + 1] =
 * 1 / sqrt(var + eps) (two-pass, biased variance -- what mi355_ln_lpi_fwd computes with a pass of its own).  Built where a workgroup owns
 * whole output rows: N = K = 256 / 384, M >= 32 (the weight-stationary kernel, option "gemm_wreg"); other shapes MI355_EUNSUPPORTED.
 * XCABlock (xcit.py:290-293): the proj GEMM writes x1 = x + gamma1 * XCA(..) and the statistics norm3 needs in front of LPI. */
int mi355_linear16_stats_fwd(const void* X16, const void* W16, const float* bias, const float* resid, float* Y, int M, int N, int K, int ldx,
                             int ldy, int precision, float* row_stats, float eps, mi355_stream_t stream);
/* Y = resid + X16 W16^T + bias (fp32) AND the next LayerNorm of every row of Y in the 16-bit operand format:
 * U16[m][n] = T((Y[m][n] - mean_m) / sqrt(var_m + eps) * ln_w[n] + ln_b[n]) (row stride ldu elements) -- what mi355_layernorm16_fwd would
 * compute from Y with a launch and a pass of its own.  Built at N = K = 256, M >= 32 (other shapes MI355_EUNSUPPORTED).  CSWinBlock at stage 3 (cswin.py:192-194):
 * x = x + proj(att) and norm2(x) in one launch.  The fp16 conversions report into the range word (code 2). */
int mi355_linear16_ln16_fwd(const void* X16, const void* W16, const float* bias, const float* resid, float* Y, const float* ln_w,
                            const float* ln_b, float eps, void* U16, int M, int N, int K, int ldx, int ldy, int ldu, int precision,
                            mi355_stream_t stream);
int mi355_linear16_ws_fwd(const void* X16, const void* W16, const float* bias, const float* gamma, const float* resid, void* Y,
                          int M, int N, int K, int ldx, int ldy, int act, int out16, int precision, void* ws, size_t ws_bytes,
                          mi355_stream_t stream);
/* Channel-major token mixing (mlp_mixer.py:45-47: norm1 -> transpose(1,2) -> token_mlp -> transpose(1,2) -> + x) without transposes
 * in HBM.  mi355_layernorm16_t_fwd writes LayerNorm(x (B,N,C)) TRANSPOSED per image in 16 bit: ut (B, C, NP), zero-filled for
 * N <= n < NP (NP % 32 == 0; pick NP % 64 == 0 so that it is a valid K for mi355_linear16_fwd against a weight zero-padded to
 * (T, NP)).  mi355_linear16_tr_fwd is mi355_linear16_fwd with the result written transposed per image and no gamma/act:
 *   Y[(img * N + n) * rows_per_image + c] = resid[same] + (X16[img * rows_per_image + c, :] . W16[n, :]) + bias[n]
 * for M = B * rows_per_image rows of X16; Y / resid fp32 (B, N, rows_per_image).  Needs K % 64 == 0, rows_per_image % 4 == 0. */
int mi355_layernorm16_t_fwd(const float* x, const float* weight, const float* bias, void* ut16, int B, int N, int C, int NP, float eps,
                            int precision, mi355_stream_t stream);
int mi355_linear16_tr_fwd(const void* X16, const void* W16, const float* bias, const float* resid, float* Y, int M, int N, int K,
                          int ldx, int rows_per_image, int precision, mi355_stream_t stream);
/* The whole token-mixing half of a MixerLayer (mlp_mixer.py:47 with norm1 and Mlp.forward :27-33) as a row-statistics pass + ONE kernel:
 *   y (B,N,C) = x + ( gelu( LayerNorm(x)^T W1^T + b1 ) W2^T + b2 )^T
 * for N = 196 tokens, T % 32 == 0 hidden token units (T <= 512), C % 256 == 0 (C <= 1024): neither the transposed LayerNorm output nor the hidden
 * tensor exists in HBM (mixer_fused.hip).  w1p16: (T, 224) 16-bit = fc1.weight (T, 196) zero-padded along in_features;
 * w2s16: (T / 32, 208, 32) 16-bit slice-major = fc2.weight (196, T) with w2s[kb][n][j] = W2[n][kb * 32 + j], rows n >= 196 zero.
 * ws: mi355_mixer_token_workspace_bytes (the per-token (mean, rstd) of the pre-pass).  y may alias x.  Other geometries:
 * MI355_EUNSUPPORTED (compose mi355_layernorm16_t_fwd / mi355_linear16_fwd / mi355_linear16_tr_fwd above). */
size_t mi355_mixer_token_workspace_bytes(int B, int N, int C);
int mi355_mixer_token_fwd(const float* x, const float* ln_w, const float* ln_b, float ln_eps, const void* w1p16, const float* b1,
                          const void* w2s16, const float* b2, float* y, int B, int N, int C, int T, int precision, void* ws,
                          size_t ws_bytes, mi355_stream_t stream);
/* LayerNorm + Linear in one pass for narrow rows (K = 64 or 128 = the normalised width; CSWin stage 1 / 2 and XCiT-nano qkv):
 *   Y = act( ((x - mean) / sqrt(var + eps)) . W16^T + bias ),  x (M,K) fp32 with row stride ldx floats, Y fp32 or 16-bit (out16).
 * The LayerNorm affine part must be folded into W16 / bias by the caller (W' = W diag(ln_weight), b' = b + W ln_bias).  The rows are
 * normalised on their way into LDS, so the 16-bit LayerNorm output never exists in HBM.  N % 8 == 0. */
int mi355_ln_linear16_fwd(const float* X, const void* W16, const float* bias, void* Y, int M, int N, int K, int ldx, int ldy, float eps,
                          int act, int out16, int precision, mi355_stream_t stream);
/* mi355_sdpa_fwd / mi355_cswin_lepe_attn_fwd with 16-bit qkv and out buffers (same layouts). */
int mi355_sdpa16_fwd(const void* qkv16, void* out16, int B, int N, int heads, int d, float scale, int precision,
                     mi355_stream_t stream);
int mi355_cswin_lepe_attn16_fwd(const void* qkv16, const float* getv_w, const float* getv_b, void* out16,
                                int B, int reso, int Ctot, int c0, int Cb, int heads, int Hsp, int Wsp,
                                float scale, int precision, mi355_stream_t stream);
/* Both stripe branches of a CSWinBlock in ONE launch (cswin.py:155-165, 186-192): branch 0 = vertical stripes (H_sp = reso, W_sp =
 * split) on channels [0, Ctot/2) with get_v weights w0/b0, branch 1 = horizontal stripes on [Ctot/2, Ctot) with w1/b1; `heads` per branch.
 * Same arithmetic as two mi355_cswin_lepe_attn16_fwd calls. */
int mi355_cswin_lepe_attn16_pair_fwd(const void* qkv16, const float* getv_w0, const float* getv_b0, const float* getv_w1,
                                     const float* getv_b1, void* out16, int B, int reso, int Ctot, int heads, int split, float scale,
                                     int precision, mi355_stream_t stream);

/* Second half of a CSWinBlock for the narrow stages in one kernel (cswin.py:191-196): x1 = x + ctx16 Wp^T + bp (the proj Linear and
 * its residual), y = x1 + gamma * (W2 gelu(W1' LN(x1) + b1') + b2).  ctx16 (M, C) and wp16 (C, C) 16-bit, bp (C) fp32, the rest as
 * mi355_mlp_fused_fwd (at C = 128 w2_16 slice-major).  Built for C = 64 / hidden 256 (x1 never reaches HBM) and C = 128 / hidden 512
 * (x1 is parked in the y rows and read back by the same lanes); other shapes: MI355_EUNSUPPORTED (callers use mi355_linear16_fwd +
 * mi355_mlp_fused_fwd). */
int mi355_proj_mlp_fused_fwd(const float* x, const void* ctx16, const void* wp16, const float* bp, const void* w1_16, const float* b1,
                             const void* w2_16, const float* b2, const float* gamma, float* y, long M, int C, int hidden, int layernorm,
                             float eps, int precision, mi355_stream_t stream);

/* First half of a CSWinBlock for the narrow stages in one kernel (cswin.py:180-190, LePEAttention.forward :101-127):
 * ctx16 = concat over branches / heads of  softmax((q * scale) k^T) v + LePE(v)  with  [q | k | v] = LayerNorm(x) Wqkv^T + bqkv
 * on the two stripe branches (branch 0: stripes reso x split on channels [0, C/2), branch 1: split x reso on [C/2, C)).  x (B, L, C)
 * fp32; wqkv16 (3C, C) 16-bit and bqkv (3C) fp32 with the LayerNorm affine part folded in by the caller (W' = W diag(ln_w),
 * b' = b + W ln_b); getv_w* (C/2, 1, 3, 3) / getv_b* (C/2) of the two branches; ctx16 (B, L, C) 16-bit.  Built for C = 64 / 128,
 * head width 32 (heads_per_branch = C / 64), reso * split <= 64 tokens per stripe; other shapes: MI355_EUNSUPPORTED (callers use
 * mi355_ln_linear16_fwd + mi355_cswin_lepe_attn16_pair_fwd).  qkv never exists in HBM. */
int mi355_cswin_stripe_attn_fwd(const float* x, const void* wqkv16, const float* bqkv, const float* getv_w0, const float* getv_b0,
                                const float* getv_w1, const float* getv_b1, void* ctx16, int B, int reso, int C, int heads_per_branch,
                                int split, float scale, float eps, int precision, mi355_stream_t stream);

/* Fused LayerNorm + MLP + residual for narrow token streams (C = 64 / hidden 256: CSWin stage 1; C = 128 / hidden 512: CSWin stage 2,
 * XCiT-nano; cswin.py:194-196 with Mlp :29-44, xcit.py:293 with Mlp :21-38):
 *   y = x + gamma * (W2 gelu(W1' xn + b1') + b2),   xn = (x - mean) / sqrt(var + eps) over the C channels (layernorm != 0) or x.
 * The hidden activations stay in registers (no (M x hidden) tensor in HBM).  The LayerNorm affine part must be folded into the first
 * Linear by the caller: W1' = W1 diag(ln_weight), b1' = b1 + W1 ln_bias.  w1_16 (hidden, C) and w2_16 (C, hidden) are 16-bit in the
 * operand type of `precision` (1 fp16, 2 bf16); for C = 128, w2_16 is arranged slice-major: (hidden/32, C, 32) with
 * w2_16[s][c][j] = W2[c][32 s + j] (the kernel streams 32-unit slices of both matrices through LDS).  x, y (M, C), b1', b2, gamma
 * fp32; b2 / gamma may be NULL.  Other shapes:
 * MI355_EUNSUPPORTED (callers use mi355_layernorm16_fwd + mi355_linear16_fwd x 2).
 * `layernorm` is a flag word: bit 0 = normalise x; bit 1 (round 6) = the caller has PROVEN from the folded weights that the one unbounded
 * 16-bit intermediate, gelu(W1' xn + b1'), stays below 65504 (|xn| <= sqrt(C - 1), so max_i (sum_j |W1'[i][j]| sqrt(C - 1) + |b1'[i]|) bounds
 * it): the kernel then does not report into the fp16 range word and the launch does not count as a producer for mi355_range_wait.  Without
 * bit 1 (and in precision 1) a saturating hidden activation is reported with code 4. */
int mi355_mlp_fused_fwd(const float* x, const void* w1_16, const float* b1, const void* w2_16, const float* b2, const float* gamma,
                        float* y, long M, int C, int hidden, int layernorm, float eps, int precision, mi355_stream_t stream);

/* ---- attention cores ------------------------------------------------------------------------------ */

/* ViT Attention core (ViT.py:82-86): qkv (B,N,3,h,d) fp32 as produced by the qkv Linear; out (B,N,h*d).
 *   out[b,n,i*d+j] = (softmax((Q_i K_i^T) * scale) V_i)[n,j].  d in {32,64}. */
int mi355_sdpa_fwd(const float* qkv, float* out, int B, int N, int heads, int d, float scale, int precision,
                   mi355_stream_t stream);

/* CSWin LePEAttention.forward (cswin.py:101-127, im2cswin :78-84, get_lepe :86-99, windows :199-216).
 *   qkv: token-major (B,L,3,Ctot) buffer of the block's qkv Linear; this call handles the channel slice
 *   [c0, c0+Cb) with `heads` heads of width Cb/heads, stripe window (Hsp x Wsp) on a (reso x reso) token
 *   grid; writes out[b,l,c0+...] of a (B,L,Ctot) buffer.  getv_w (Cb,3,3), getv_b (Cb): depth-wise 3x3
 *   LePE conv, zero padded at the WINDOW border. */
int mi355_cswin_lepe_attn_fwd(const float* qkv, const float* getv_w, const float* getv_b, float* out,
                              int B, int reso, int Ctot, int c0, int Cb, int heads, int Hsp, int Wsp,
                              float scale, int precision, mi355_stream_t stream);

/* XCA core (xcit.py:249-262): qkv (B,N,3,h,d); temperature (h); out (B,N,h*d).
 *   per head: L2-normalise q,k over N; A = softmax((q^ k^T) * temp_h) (d x d); out = (A v)^T. */
int mi355_xca_fwd(const float* qkv, const float* temperature, float* out, int B, int N, int heads, int d,
                  int precision, mi355_stream_t stream);
/* The same core with out in the 16-bit operand format of `precision` (1 / 2), ready for the proj GEMM (no cast pass over ctx);
 * qkv_is16 != 0: qkv is in that format too (16-bit output of the qkv GEMM).  Arithmetic is fp32 in every case. */
int mi355_xca16_fwd(const void* qkv, int qkv_is16, const float* temperature, void* out16, int B, int N, int heads, int d,
                    int precision, mi355_stream_t stream);

/* XCiT LPI.forward (xcit.py:149-157) with BatchNorm2d in eval mode (running statistics):
 *   tokens (B,N=H*W,C) -> dw3x3(w1,b1) -> GELU -> (v - bn_mean)/sqrt(bn_var + bn_eps)*bn_w + bn_b -> dw3x3(w2,b2) -> tokens.
 *   w1,w2 (C,3,3); y = resid + gamma * LPI(x) when gamma / resid are non-NULL (XCABlock :292). */
size_t mi355_lpi_workspace_bytes(int B, int H, int W, int C);
int    mi355_lpi_fwd(const float* x, const float* w1, const float* b1, const float* bn_w, const float* bn_b,
                     const float* bn_mean, const float* bn_var, float bn_eps, const float* w2, const float* b2,
                     const float* gamma, const float* resid, float* y, int B, int H, int W, int C,
                     void* ws, size_t ws_bytes, mi355_stream_t stream);
/* The same block with the LayerNorm in front of it fused in (XCABlock.forward xcit.py:292: x + gamma3 * LPI(norm3(x))):
 *   y = resid + gamma * LPI( LayerNorm(x; ln_w, ln_b, ln_eps) ).
 * A row-statistics pass writes (mean, rstd) per token into the workspace (mi355_lpi_workspace_bytes), the stencil kernel normalises
 * the rows as it parks them in LDS: the normalised tensor never exists in HBM; pass resid = x for the block's residual (the rows the
 * kernel has just read).  Bit-identical to mi355_layernorm_fwd followed by mi355_lpi_fwd. */
int    mi355_ln_lpi_fwd(const float* x, const float* ln_w, const float* ln_b, float ln_eps, const float* w1, const float* b1,
                        const float* bn_w, const float* bn_b, const float* bn_mean, const float* bn_var, float bn_eps, const float* w2,
                        const float* b2, const float* gamma, const float* resid, float* y, int B, int H, int W, int C, void* ws,
                        size_t ws_bytes, mi355_stream_t stream);

/* ViT PatchEmbedding + token assembly (ViT.py:101-105,183-185):
 *   tokens[b, p, :] = patch_p(img_b) . Wp^T + bp + pos[p]   for p < P = (H/ps)*(W/ps);
 *   tokens[b, P, :] = cls + pos[P]                         (cls token LAST).
 *   img (B,Cin,H,W), Wp (E, Cin*ps*ps), bp (E), cls (E), pos (P+1,E), tokens (B,P+1,E).
 *   cls == pos == NULL: plain patch embedding, tokens (B,P,E)  (mlp_mixer.py:60-63). */
int mi355_patch_embed_fwd(const float* img, const float* Wp, const float* bp, const float* cls, const float* pos,
                          float* tokens, int B, int Cin, int H, int W, int ps, int E, int precision,
                          mi355_stream_t stream);
/* The same with a scratch buffer: in the 16-bit operand modes the ViT form (cls + pos) at sizes that fill the chip runs as one im2col
 * pass into the 16-bit operand format + the persistent GEMM with the position rows as a periodic residual (0.25 -> 0.15 ms at
 * ViT-Base/16, B = 256).  workspace_bytes() returns 0 where that path does not apply; with ws == NULL or too small a buffer the call
 * is mi355_patch_embed_fwd. */
size_t mi355_patch_embed_workspace_bytes(int B, int Cin, int H, int W, int ps, int E, int precision);
int mi355_patch_embed_ws_fwd(const float* img, const float* Wp, const float* bp, const float* cls, const float* pos,
                             float* tokens, int B, int Cin, int H, int W, int ps, int E, int precision,
                             void* ws, size_t ws_bytes, mi355_stream_t stream);

/* ---- LayerNorm folded into the GEMMs around it (csrc/ln_fold.hip; the pre-LN chain of ViT.py:116-119) -----------------------------
 * Replaces "GEMM -> LayerNorm launch -> GEMM" by three pieces that never re-read the fp32 stream:
 *   mi355_linear16_emit_fwd    producer: Y (M,N) fp32 = resid + act(X16 W16^T + bias) exactly as mi355_linear16_fwd, and beside it
 *                              a16_out[m][n] = T(Y[m][n] - cvec[m]) (16-bit operand type of `precision`, row stride N) and
 *                              stats[(n / 32) * M + m] = {mean, sum of squared deviations} of Y[m][32 g .. 32 g + 31] (fp32 pairs,
 *                              mi355_ln_fold_stats_bytes(M, N) bytes).  Two-accumulator persistent kernel only: M % 128 == 0,
 *                              N % 256 == 0, K % 64 == 0, K >= 640, else MI355_EUNSUPPORTED (nothing touched).
 *   mi355_ln_finalize_fwd      per row: exact mean mu / rstd r of Y from the group pairs; rowtau[m] = {r, r (cvec[m] - mu)};
 *                              cvec[m] := mu.  A row with |cvec - mu| > tol * std, or std outside [2^-7, 2^10] (fp16 band of
 *                              Y - cvec), is REWRITTEN from x (= Y, row stride cols): a16 = T((x - mu) r), rowtau = {1, 0} -- the
 *                              plain LayerNorm operand -- so the result never depends on how well cvec predicted the mean.
 *                              slow_rows (device word or NULL) counts such rows (diagnostics).
 *   mi355_linear16_lnfold_fwd  consumer: Y16 = act(rowtau[m].x * (A16 W16'^T)[m][n] + rowtau[m].y * colsum[n] + bias[n]) with
 *                              W16'[n][k] = T(gamma[k] W[n][k]), colsum[n] = sum_k W16'[n][k] (fp32 sum of the ROUNDED weights),
 *                              bias = b + W beta: = act(LayerNorm(Y) W^T + b) of the reference.  Persistent 256 x 256 kernel.
 *   mi355_ln_center16_fwd      first LayerNorm of a chain (no producer): a16 = T((x - mu) r), rowtau = {1, 0}, cvec = mu for every row;
 *                              cols % 4 == 0, cols <= 2048.
 * precision 1 (fp16) or 2 (bf16).  Results: within operand rounding of LayerNorm -> 16-bit -> GEMM (the rounding point moves from
 * LayerNorm(x) to x - c; tests/test_ln_fold_gpu.py bounds the difference). */
size_t mi355_ln_fold_stats_bytes(int rows, int cols);
int mi355_ln_center16_fwd(const float* x, void* a16, float* rowtau, float* cvec, int rows, int cols, float eps, int precision,
                          mi355_stream_t stream);
int mi355_ln_finalize_fwd(const float* stats, const float* x, void* a16, float* cvec, float* rowtau, int rows, int cols, float eps,
                          float tol, int precision, unsigned* slow_rows, mi355_stream_t stream);
int mi355_linear16_emit_fwd(const void* X16, const void* W16, const float* bias, const float* resid, float* Y, int M, int N, int K, int ldx,
                            int act, int precision, const float* cvec, void* a16_out, float* stats, mi355_stream_t stream);
int mi355_linear16_lnfold_fwd(const void* A16, const void* W16, const float* bias, const float* rowtau, const float* colsum, void* Y16,
                              int M, int N, int K, int lda, int ldy, int act, int precision, mi355_stream_t stream);

/* ViT Attention.forward as ONE call (ViT.py:79-89; SURVEY 8b lists `mhsa` among the exported ops):
 *   y = resid + proj( concat_heads( softmax(q k^T scale) v ) ) + b_proj,   [q | k | v] = x Wqkv^T + b_qkv  viewed (B,N,3,heads,d)
 * x (B,N,C): fp32 (x_is16 = 0: converted to the operand format first) or already in the 16-bit operand format of `precision`
 * (x_is16 = 1: e.g. the output of mi355_layernorm16_fwd); Wqkv16 (3C,C) / Wproj16 (C,C): 16-bit copies of the weights in the same
 * format (mi355_cast16_fwd makes them); b_qkv (3C) / b_proj (C) / resid (B,N,C) fp32 or NULL; y (B,N,C) fp32.  The same kernels the
 * host composition launches (16-bit GEMM engine, K/V-resident core for d in {32,64} and N <= 224, the streaming core otherwise);
 * q / k / v / context live in the workspace in 16 bit.  precision 1 (fp16) or 2 (bf16); C % 64 == 0, head_dim in {32,64,128,192,256}. */
size_t mi355_mhsa_workspace_bytes(int B, int N, int C, int x_is16);
int mi355_mhsa_fwd(const void* x, int x_is16, const void* Wqkv16, const float* b_qkv, const void* Wproj16, const float* b_proj,
                   const float* resid, float* y, int B, int N, int C, int heads, float scale, int precision, void* workspace,
                   size_t workspace_bytes, mi355_stream_t stream);

/* General multi-head attention core (the plain softmax(QK^T*s)V pattern of setr.py:62-72, pvt.py:73-91,
 * segformer.py:33-50, cmt.py:93-111, moat.py:74-84, bvit.py:66-76 ...): any N_q / N_kv, online softmax over 64-key tiles.
 *   out[b, n, i*d + j] = sum_m softmax_m(scale * <q[b,n,i,:], k[b,m,i,:]> + bias[b?, i, n, m]) v[b,m,i,j]
 * q (B,Nq,.) / k,v (B,Nkv,.) / out (B,Nq,.) are addressed as  ptr + (b*N + n)*ld + i*head_dim + j  (ld = row stride in elements,
 * >= heads*head_dim), so they may be slices of one fused projection or separate tensors.  bias (heads, Nq, Nkv) fp32 or NULL;
 * image b uses bias + b*bias_batch_stride (0 = shared).  io16 = 0: fp32 tensors; 1: tensors in the 16-bit operand type of
 * `precision` (1 fp16, 2 bf16).  head_dim in {32, 64}. */
int mi355_sdpa_general_fwd(const void* q, const void* k, const void* v, const float* bias, void* out, int B, int num_heads, int Nq,
                           int Nkv, int head_dim, long ldq, long ldk, long ldv, long ldo, long bias_batch_stride, float scale,
                           int io16, int precision, mi355_stream_t stream);

/* ---- model-level glue (callers of the blocks) ------------------------------------------------------------ */

/* Conv2d as implicit GEMM, token-major output: y (B, OH*OW, Cout) = act(conv(x) + bias + pos); no im2col buffer.
 *   in_layout 0: x NCHW (B,Cin,H,W), weight rows ordered (c,ky,kx)        -- CSWin stem conv 7x7 s4 p2 (cswin.py:247-251),
 *                                                                            first conv of XCiT ConvPatchEmbed (xcit.py:88-126)
 *   in_layout 1: x token-major (B,H*W,Cin), weight rows ordered (ky,kx,c)  -- CSWin Merge_Block conv 3x3 s2 p1 (cswin.py:218-233),
 *                                                                            the later ConvPatchEmbed convs
 * weight is (Cout, ldw), ldw >= Cin*KH*KW, ldw % 4 == 0, zero padded beyond Cin*KH*KW.  bias (Cout) and pos (OH*OW, Cout: a
 * per-output-token addend, e.g. the XCiT Fourier position encoding xcit.py:398-400) may be NULL; act = MI355_ACT_NONE|GELU.
 * An eval-mode BatchNorm after the conv is folded into weight / bias by the caller. */
int mi355_conv2d_tokens_fwd(const float* x, const float* weight, const float* bias, const float* pos, float* y, int B, int Cin,
                            int H, int W, int Cout, int KH, int KW, int stride, int pad, int ldw, int in_layout, int act,
                            int precision, mi355_stream_t stream);

/* y[b,c] = mean over n of x[b*batch_stride + n*C + c]  (cswin.py:341, mlp_mixer.py:77, ViT.py:189-190). */
int mi355_token_mean_fwd(const float* x, float* y, int B, int N, int C, long batch_stride, mi355_stream_t stream);

/* Class attention core (CaiT / XCiT ClassAttention.forward xcit.py:174-188): one query per image and head attends over N keys,
 *   out[b, i*d + j] = sum_n softmax_n(scale * sum_j' q[b, i*d + j'] k[b,n,i*d + j']) v[b,n,i*d + j],   exact fp32.
 * q row b starts at q + b*ldq; key / value token (b,n) starts at k|v + (b*N + n)*ldkv (so the fused (B,N,3C) qkv tensor can be
 * passed in place with k = qkv + C, v = qkv + 2C, ldkv = 3C).  out (B, h*d) dense.  d <= 64, N <= 4096. */
int mi355_class_attn_fwd(const float* q, const float* k, const float* v, float* out, int B, int N, int num_heads, int head_dim,
                         long ldq, long ldkv, float scale, mi355_stream_t stream);

/* Depth-wise patch convolution on a token grid: x (B, H*W, C) -> y (B, (H/sr)*(W/sr), C), kernel == stride == sr, groups == C,
 *   y[b, oy*OW + ox, c] = bias[c] + sum_{ky,kx} weight[c, ky*sr + kx] * x[b, (oy*sr + ky)*W + ox*sr + kx, c]
 * (the spatial reduction in front of K / V in PVT / CMT, pvt.py:66-70; an eval-mode BatchNorm is folded into weight / bias by the
 * caller).  bias may be NULL.  C % 4 == 0, H % sr == 0, W % sr == 0. */
int mi355_dwconv_patch_tokens_fwd(const float* x, const float* weight, const float* bias, float* y, int B, int H, int W, int C,
                                  int sr, mi355_stream_t stream);

/* y[r, c] = alpha * x[r, c] + gamma[c] * u[r, c]  over rows x cols with row strides ldx / ldu / ldy (floats; ldx or ldu may be 0
 * to broadcast one row).  u == NULL drops the second term, gamma == NULL means 1.  The token-axis glue of the XCiT class-attention
 * stage (xcit.py:219-231, 402-403): residual with LayerScale, cls-row scatter / gather, token concatenation. */
int mi355_axpby_fwd(const float* x, const float* u, const float* gamma, float* y, long rows, int cols, long ldx, long ldu,
                    long ldy, float alpha, mi355_stream_t stream);

/* ---- measurement helpers --------------------------------------------------------------------------- */
/* ---- multi-GPU: the end-of-forward all-gather (the reference has no distributed code, this is the collective the
 * batch-sharded ViT forward of BASELINE configs[4] ends with) --------------------------------------------------------------------
 * One process per GPU.  Rank 0 calls mi355_comm_unique_id and hands the MI355_COMM_ID_BYTES bytes to every rank through the host's
 * own bootstrap (the Python mirror uses the torch.distributed store); every rank then calls mi355_comm_init with the SAME id on its
 * current HIP device (collective: returns once all `world` ranks have joined) and owns the opaque handle until mi355_comm_destroy.
 * mi355_allgather_f32: every rank contributes `count` floats from `send`; `recv` (world * count floats) receives the contributions
 * in rank order on every rank -- one ncclAllGather over xGMI, asynchronous on `stream`.  RCCL is bound at run time (the copy already
 * loaded in the process wins); without a loadable librccl these return MI355_EUNSUPPORTED. */
#define MI355_COMM_ID_BYTES 128
/* ---- SURVEY.md 8(b) names ---------------------------------------------------------------------------------------------------------
 * The survey's contract lists one `mi355_<op>_workspace_bytes` / `mi355_<op>_fwd` pair per op and three op names this header spells
 * differently.  Both spellings are exported; the aliases forward to the entries above with the same arguments:
 *     mi355_sdpa_core_fwd          = mi355_sdpa_fwd            (ViT.py:82-86)
 *     mi355_gemm_bias_act_fwd      = mi355_linear_fwd          (epilogue none / gelu / residual: `act`, `resid`)
 *     mi355_mixer_token_mlp_fwd    = mi355_mixer_token_fwd     (+ mi355_mixer_token_mlp_workspace_bytes)
 * and the ops that need no scratch get the workspace query the contract promises (it returns 0): sdpa_core, gemm_bias_act,
 * cswin_lepe_attn, xca, layernorm. */
int    mi355_sdpa_core_fwd(const float* qkv, float* out, int B, int N, int heads, int d, float scale, int precision, mi355_stream_t stream);
size_t mi355_sdpa_core_workspace_bytes(int B, int N, int heads, int d);
int    mi355_gemm_bias_act_fwd(const float* X, const float* W, const float* bias, const float* gamma, const float* resid, float* Y, int M, int N,
                               int K, int ldx, int ldy, int act, int precision, mi355_stream_t stream);
size_t mi355_gemm_bias_act_workspace_bytes(int M, int N, int K);
int    mi355_mixer_token_mlp_fwd(const float* x, const float* ln_w, const float* ln_b, float ln_eps, const void* w1p16, const float* b1,
                                 const void* w2s16, const float* b2, float* y, int B, int N, int C, int T, int precision, void* ws,
                                 size_t ws_bytes, mi355_stream_t stream);
size_t mi355_mixer_token_mlp_workspace_bytes(int B, int N, int C);
size_t mi355_cswin_lepe_attn_workspace_bytes(int B, int reso, int Ctot);
size_t mi355_xca_workspace_bytes(int B, int N, int heads, int d);
size_t mi355_layernorm_workspace_bytes(int rows, int cols);

/* mi355_ln_lpi_fwd with the LayerNorm statistics given (stats (B*H*W, 2) from mi355_linear16_stats_fwd): no statistics pass, no workspace. */
int mi355_ln_lpi_stats_fwd(const float* x, const float* stats, const float* ln_w, const float* ln_b, const float* w1, const float* b1,
                           const float* bn_w, const float* bn_b, const float* bn_mean, const float* bn_var, float bn_eps, const float* w2,
                           const float* b2, const float* gamma, const float* resid, float* y, int B, int H, int W, int C,
                           mi355_stream_t stream);

int mi355_comm_unique_id(void* id_out, size_t id_bytes);
int mi355_comm_init(const void* id, size_t id_bytes, int rank, int world, void** comm_out);
int mi355_allgather_f32(void* comm, const float* send, float* recv, size_t count, mi355_stream_t stream);
int mi355_comm_destroy(void* comm);

/* Bicubic resize of a token-major table (n0h*n0w, dim) -> (oh*ow, dim), the arithmetic of F.interpolate(mode="bicubic",
 * align_corners=False, scale_factor=(scale_h, scale_w)) on the (1, dim, n0h, n0w) view: the position-embedding interpolation of
 * vision_transformers/ViT.py:160-178 (run once per resolution by the host mirror, cached). */
int mi355_bicubic_rows_fwd(const float* table, float* out, int n0h, int n0w, int oh, int ow, int dim, float scale_h, float scale_w,
                           mi355_stream_t stream);
/* float4 streaming copy of `bytes` (multiple of 16): the achievable-HBM-bandwidth yardstick for bench.py. */
int mi355_stream_copy(const void* src, void* dst, size_t bytes, mi355_stream_t stream);
/* read-only float4 sweep of `bytes` (sum-reduced, result discarded; `sink` is a 4-byte device scratch). */
int mi355_stream_read(const void* src, size_t bytes, float* sink, mi355_stream_t stream);
/* Box calibration for bench.py (SURVEY.md 8d "a measured MFMA microbench"): a register-operand MFMA loop, two 4-wave workgroups per CU,
 * every wave `iters` x 8 independent v_mfma_f32_16x16x32_f16 (shape 0: a 32 x 64 wave tile, the engine's instruction) or `iters` x 4
 * v_mfma_f32_32x32x16_f16 (shape 1: a 64 x 64 wave tile), operands random in [-1, 1) -- nothing but the matrix pipe, i.e. what THIS box
 * sustains under dense MFMA load.  report (device, 3 x 8 bytes, written by workgroup 0): {shader-clock ticks, ticks of the constant 100 MHz
 * counter} over the first wave's loop and the number of workgroups launched.  MFMA instructions of the launch = workgroups x 4 x iters x
 * (8 | 4) at 16 384 | 32 768 FLOP each; the caller times the launch (mi355_event_time_*).  sink: 4 bytes of device scratch. */
int mi355_mfma_yardstick(int shape, int iters, float* sink, unsigned long long* report, mi355_stream_t stream);
/* HIP-event stopwatch ON `stream` (torch.cuda.Event only sees torch's current stream): begin records an event and
 * returns an opaque handle; end records the closing event, waits for it and returns elapsed milliseconds. */
int mi355_event_time_begin(mi355_stream_t stream, void** handle);
int mi355_event_time_end(mi355_stream_t stream, void* handle, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* MI355ATTN_H */
