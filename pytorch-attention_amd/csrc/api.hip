// api.hip -- library-level entry points: version, thread-local error text, tuning options, HIP-event stopwatch.
#include "common.h"
#include <atomic>
#include <cstring>
#include <mutex>
#include <algorithm>
#include <unordered_map>
#include <map>
#include <string>
#include <vector>

namespace mi355 {
static thread_local char g_err[512] = "";

// ---- tuning options: ONE BLOCK PER DEVICE --------------------------------------------------------------------------------------
// mi355_set_option / mi355_get_option act on the block of the calling thread's CURRENT device (hipGetDevice), and every launch reads
// the block of the device it launches on: a host that drives eight GPUs from one process (one thread per GPU) can tune --
// or, in a test, sabotage -- one device without the others seeing it.  A block starts from the defaults below.
constexpr int MAX_DEV = 64;
enum Opt { O_CHUNK_IMAGES, O_NT, O_REVERSE, O_GEMM_VARIANT, O_ECA_SINGLE, O_SE_SINGLE, O_CBAM_SINGLE, O_WS_PERSISTENT, O_STEM_DIRECT,
           O_ZOO_SINGLE, O_SPIN_LIMIT, O_GEMM_PA, O_GEMM_SPLITK, O_DA_FUSED, O_DA_RANGES, O_SE_OCC, O_LN_FOLD, O_GEMM_PA16, O_GEMM_PA_BLOCK, O_GEMM_PA_TAIL, O_LPI_PATCH, O_MIXER_FUSED, O_MIXER_EARLY, O_GEMM_SMALL, O_MLP_TT4, O_MIXER_STATS, O_ATTN_NW, O_GEMM_W4, O_RANGE_FALLBACK, O_GEMM_WREG, O_XCA_TR, O_MLP_WIDE, O_GEMM_WST, O_GEMM_WSLAB, O_COUNT };
struct OptDesc { const char* key; long def, lo, hi; };
// key, default, accepted range.  spin_limit additionally accepts 0 (forces the time-out path in tests: every exchange then fails on
// its first unsuccessful poll; real budgets start at 1024 sweeps)
static const OptDesc kOpts[O_COUNT] = {
    {"chunk_images", 0, 0, 1L << 40},    // 0 = auto (about 200 MB of x per chunk)
    {"nt", 3, 0, 3},                       // bit0: non-temporal loads, bit1: non-temporal stores in the final pass
    {"reverse", 0, 0, 1},
    {"gemm_variant", 0, 0, 17},            // tile/schedule variant of the 16-bit GEMM (gemm16.hip); 0 = dispatch by shape
    {"eca_single", 1, 0, 1},               // ECA: one read + one write of x, halo channel rows re-summed per workgroup (chan_fused.hip)
    {"se_single", 1, 0, 1},                // SE: x read once, channel means exchanged as 8-byte {mean, tag} granules (chan_fused.hip)
    {"cbam_single", 1, 0, 1},              // CBAM: x read once, row bands in registers, granule hops per band (cbam_single.hip)
    {"ws_persistent", 0, 0, 1},            // 1 = caller keeps workspace contents between calls: granule exchanges skip their zeroing
    {"stem_direct", 1, 0, 1},              // narrow conv stems: direct fp32 kernel (stem_conv.hip) vs implicit GEMM
    {"zoo_single", 1, 0, 1},               // SimAM / SRM / GCT / LCT: single-read register-resident path (chan_stat.hip) vs two passes
    {"spin_limit", 1L << 22, 1024, 1L << 30},   // poll budget of the exchange kernels (sweeps) before they give up with an error code
    {"gemm_pa", 1, 0, 1},                  // fp32-output GEMMs: two-accumulator persistent kernel where it applies (gemm16_pa.hip)
    {"gemm_splitk", 1, 0, 1},              // persistent GEMM: cut the tiles of the last partial round along K (gemm16_p8.hip)
    {"da_fused", 1, 0, 1},                 // DoubleAttention: fused kernels where they apply (double_attn_fused.hip, double_attn_small.hip)
    {"da_ranges", 0, 0, 32},               // ... pixel ranges per image in pass 1: 0 = from the batch size, 1..32 = fixed
    {"se_occ", 3, 2, 3},                   // single-read SE: workgroups per CU (2: <= 128 VGPRs, 3: <= 80 VGPRs)
    {"ln_fold", 0, 0, 1},                  // ViT encoder chain: 1 = LayerNorm folded into the neighbouring GEMMs (ln_fold.hip); measured slower
                                           // than the LayerNorm launches it removes, so it is opt-in
    {"gemm_pa16", 1, 0, 2},                // 16-bit outputs with K >= 576 on the two-accumulator kernel: 0 never, 1 GELU epilogues, 2 all
    {"gemm_pa_block", 1, 0, 1},            // two-accumulator kernel: blocked tile order (8 row x 4 column tiles per XCD round) for wide outputs
    {"gemm_pa_tail", 10, 0, 100},          // two-accumulator kernel, K >= 1024: a last round at most this many percent full goes to the small-tile ring kernel (0 = off)
    {"lpi_patch", 1, 0, 1},                // LPI at 14 x 14 tokens, C % 32 == 0: 2 x 2 patches per lane on channel-quad-major LDS planes (xcit.hip)
    {"mixer_fused", 1, 0, 1},              // MixerLayer token mixing (host mirror): one kernel where the geometry allows (mixer_fused.hip)
    {"mixer_early", 0, 0, 1},              // mixer_token_kernel: all residual loads of the epilogue before its first store (A/B switch)
    {"gemm_small", 1, 0, 1},               // mi355_linear_fwd: outputs under an eighth of a round of 128 x 128 tiles on one-wave 16 x 32 tiles (gemm_small.hip)
    {"mlp_tt4", 0, 0, 1},                  // fused MLP at C = 64 (CSWin stage 1): 8 waves x 4 token tiles at 256 VGPRs instead of 16 x 2 at 128 (A/B switch)
    {"mixer_stats", 0, 0, 1},              // mixer_token_kernel at C = 512: LayerNorm row statistics inside the kernel (1) or by the row_stats_kernel pre-pass (0, default: measured equal)
    {"attn_nw", 8, 7, 8},                  // ViT attention core at 193 .. 208 tokens (13 query tiles): waves per workgroup, 8 (13 / 16 balance) or 7 (13 / 14)
    {"gemm_w4", 1, 0, 1},                  // 16-bit outputs, 576 <= K < 1536, whole 256 x 256 tiles: the one-wave-per-SIMD persistent kernel (gemm16_w4.hip) instead of gemm16_p8
    {"range_fallback", 1, 0, 1},           // host policy of the drop-in modules (read by the binding): 1 = a forward whose fp16 operands saturated is re-run in strict mode, 0 = raise on the next call
    {"gemm_wreg", 1, 0, 1},                // fp32 (+ residual) outputs with N = K = 256 / 384: weight-stationary-in-registers streaming kernel (gemm16_wreg.hip)
    {"xca_tr", 1, 0, 1},                   // XCA core with 16-bit q / k / v and N <= 224: covariance on the 16-bit matrix pipe from one transposed LDS image (xcit.hip xca_tr_kernel)
    {"mlp_wide", 0, 0, 1},                 // 1 = mi355_mlp_fused_fwd takes C = 256 / 384 (hidden 4C) on the weight-split kernel (mlp_wide.hip); measured SLOWER than LayerNorm + two GEMMs
                                           // (profiles/r06_mlp_wide.md), hence opt-in; 0 (default) = those shapes are MI355_EUNSUPPORTED
    {"gemm_wst", 0, 0, 4},                 // 16-bit outputs with K = 768, N % 192 == 0 (ViT qkv / fc1): weights stationary in registers (gemm16_wst.hip); 1 = products without
                                           // activation, 2 = GELU epilogues too; 3 / 4 = the same on the one-wave-per-SIMD kernel with W in AGPRs.  Measured slower than the tile kernels (profiles/r06_gemm_wst.md): opt-in
    {"gemm_wslab", 1, 0, 2},               // 16-bit outputs with K = 256 / 384 / 512 (XCiT / CSWin stage 3-4 / Mixer qkv and fc1): a column slab of W stationary in
                                           // registers (gemm16_wslab.hip); 1 = GELU epilogues and M % 256 != 0 (where it measured faster), 2 = every product it takes
};
static_assert(sizeof(kOpts) / sizeof(kOpts[0]) == O_COUNT, "one table row per option, in enum order");
namespace {
constexpr long OPT_UNSET = (long)0x8000000000000000ull;              // a device block entry that follows the process default
struct OptBlock { std::atomic<long> v[O_COUNT]; };
OptBlock g_opt[MAX_DEV];                 // per-device overrides (mi355_set_option on the current device)
std::atomic<long> g_def[O_COUNT];        // process defaults (mi355_set_default_option): what a device without an override reads
std::once_flag g_opt_once;
int cur_dev() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return dev >= 0 && dev < MAX_DEV ? dev : 0;
}
void opt_init() {
    std::call_once(g_opt_once, [] {
        for (int i = 0; i < O_COUNT; ++i) g_def[i].store(kOpts[i].def, std::memory_order_relaxed);
        for (auto& b : g_opt)
            for (int i = 0; i < O_COUNT; ++i) b.v[i].store(OPT_UNSET, std::memory_order_relaxed);
    });
}
inline long opt(Opt o) {
    opt_init();
    const long v = g_opt[cur_dev()].v[o].load(std::memory_order_relaxed);
    return v != OPT_UNSET ? v : g_def[o].load(std::memory_order_relaxed);
}
}  // namespace

char* err_buf() { return g_err; }

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
long opt_chunk_images() { return opt(O_CHUNK_IMAGES); }
long opt_nt() { return opt(O_NT); }
long opt_reverse() { return opt(O_REVERSE); }
long opt_eca_single() { return opt(O_ECA_SINGLE); }
long opt_se_single() { return opt(O_SE_SINGLE); }
long opt_cbam_single() { return opt(O_CBAM_SINGLE); }
long opt_ws_persistent() { return opt(O_WS_PERSISTENT); }
long opt_ln_fold() { return opt(O_LN_FOLD); }
long opt_gemm_pa16() { return opt(O_GEMM_PA16); }
long opt_gemm_pa_block() { return opt(O_GEMM_PA_BLOCK); }
long opt_gemm_pa_tail() { return opt(O_GEMM_PA_TAIL); }
long opt_lpi_patch() { return opt(O_LPI_PATCH); }
long opt_mixer_early() { return opt(O_MIXER_EARLY); }
long opt_gemm_small() { return opt(O_GEMM_SMALL); }
long opt_mlp_tt4() { return opt(O_MLP_TT4); }
long opt_mixer_stats() { return opt(O_MIXER_STATS); }
long opt_attn_nw() { return opt(O_ATTN_NW); }
long opt_gemm_w4() { return opt(O_GEMM_W4); }
long opt_range_fallback() { return opt(O_RANGE_FALLBACK); }
long opt_gemm_wreg() { return opt(O_GEMM_WREG); }
long opt_xca_tr() { return opt(O_XCA_TR); }
long opt_mlp_wide() { return opt(O_MLP_WIDE); }
long opt_gemm_wst() { return opt(O_GEMM_WST); }
long opt_gemm_wslab() { return opt(O_GEMM_WSLAB); }

// ---- workspaces of the granule-exchange kernels (chan_fused.hip, cbam_single.hip, chan_stat.hip) ---------------------------------
// A granule is valid when it carries the tag of the CURRENT launch = the workspace's epoch word + 1 (advanced on the device by the
// launch's last ticket draw), so a slot written by any earlier launch can never look valid and the region only has to be zeroed
// when its history is unknown: first use of the pointer, a different layout key, or "ws_persistent" off (the default: a C caller
// that frees / reuses workspace memory between calls must not opt in).  Nothing about a launch lives on the host, which is what
// lets these kernels be recorded by hipGraph capture.
namespace {
struct WsEntry { unsigned long long key; };
std::mutex g_ws_mu;
std::unordered_map<const void*, WsEntry> g_ws;
}  // namespace

bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return cs != hipStreamCaptureStatusNone;
}

// Workspaces of the kernels that keep their launch tag in device memory (se_single_kernel, cbam_single_kernel): the host only has to know whether the
// region was zeroed for this shape.  Under stream capture an unknown region stays unknown (the memset the caller records runs at
// replay time, not now); a known one needs nothing.
bool ws_known(const void* region, unsigned long long key, hipStream_t st) {
    if (!opt_ws_persistent()) return false;
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto it = g_ws.find(region);
    if (it != g_ws.end() && it->second.key == key) return true;
    if (!stream_is_capturing(st)) g_ws[region] = WsEntry{key};
    return false;
}
// Zero an exchange area with a KERNEL.  Under stream capture a recorded hipMemsetAsync did not reliably take effect before the kernel
// node behind it on this runtime (ROCm 7.2: the second of two back-to-back memset nodes -- a replayed SE launch saw the granules of
// the previous replay); a kernel node is ordered like any other launch, so the exchange kernels zero their areas this way everywhere.
namespace {
__global__ __launch_bounds__(256) void ws_zero_kernel(unsigned* p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
}  // namespace
hipError_t ws_zero_async(void* p, size_t bytes, hipStream_t st) {
    const size_t words = bytes / 4;                                    // callers pass multiples of 4 bytes, 4-byte aligned
    if (!words) return hipSuccess;
    size_t blocks = (words + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    ws_zero_kernel<<<(int)blocks, 256, 0, st>>>(static_cast<unsigned*>(p), words);
    return hipGetLastError();
}
void ws_forget(const void* region) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    g_ws.erase(region);
}
void ws_forget_range(const void* base, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    const char* lo = static_cast<const char*>(base);
    for (auto it = g_ws.begin(); it != g_ws.end();) {
        const char* p = static_cast<const char*>(it->first);
        if (p >= lo && p < lo + bytes) it = g_ws.erase(it);
        else ++it;
    }
}

// ---- exchange-kernel failure word + fp16 range word (common.h): ONE PAIR PER DEVICE ---------------------------------------------
// One pinned, device-visible 4 KB block, 64 bytes per device ordinal: word 0 = exchange failure code, word 4 = range code.  A kernel
// reports into the words of the device it runs on and the host checks the words of the calling thread's current device, so one
// device's time-out (or fp16 overflow) never fails another device's next call.
namespace {
std::once_flag g_sync_once;
unsigned* g_sync_block = nullptr;
}  // namespace
unsigned* sync_err_word() {
    std::call_once(g_sync_once, [] {
        void* p = nullptr;
        if (hipHostMalloc(&p, 64 * MAX_DEV, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) == hipSuccess && p) {
            std::memset(p, 0, 64 * MAX_DEV);
            g_sync_block = static_cast<unsigned*>(p);
        } else {
            (void)hipGetLastError();
        }
    });
    return g_sync_block ? g_sync_block + 16 * cur_dev() : nullptr;
}
unsigned* sync_err_word_on(hipStream_t st) {
    if (!g_sync_block && stream_is_capturing(st)) return nullptr;         // never allocate pinned memory inside a capture
    return sync_err_word();
}
// ---- "which launch do I have to wait for before the range word is final?" (round 6: mi355_range_arm / mi355_range_wait) ---------------
// While a device is ARMED, every launcher that hands a kernel the range word (range_word() below: the fp16 producers) counts itself and
// marks the device dirty.  mi355_range_wait() must synchronise on an event that lies BEHIND the last producer; recording one behind every
// producer would put ~50 marker packets into a ViT-Base forward (measured: +0.13 ms of 12.1), so the caller PREDICTS the last producer --
// mi355_range_arm(1 + k): "the k-th producer since this call is the last one", the count its previous forward of the same module reported
// (mi355_range_launches) -- and the ONE event of the forward is recorded in front of the first launch that follows the k-th producer
// (TraceScope's constructor sits in front of every instrumented launch).  The non-reporting launches queued behind it (attention core,
// fp32-output projections, fc2) then keep the GPU busy while the host already returns.  A wrong or missing prediction costs slack, never
// correctness: a producer launched after the event marks the device dirty again and mi355_range_wait() records a second event at the tail.
namespace {
struct RangeMark {
    std::atomic<int> armed{0}, dirty{0}, fresh{0}, have{0}, count{0}, expect{0};
    hipEvent_t ev = nullptr;
    hipStream_t st = nullptr;
};
RangeMark g_rmark[MAX_DEV];
void range_mark_record(RangeMark& m) {
    m.dirty.store(0, std::memory_order_relaxed);
    if (stream_is_capturing(m.st)) return;                                // never record the shared event into a graph
    if (!m.ev && hipEventCreateWithFlags(&m.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); m.ev = nullptr; return; }
    if (hipEventRecord(m.ev, m.st) == hipSuccess) m.have.store(1, std::memory_order_relaxed);
    else (void)hipGetLastError();
}
}  // namespace
// A producer ENTRY may issue several launches (split rounds, helper passes): range_word() only opens it (pending), the entry's closing
// MI355_LAUNCH_CHECK marks the device dirty, and no event is ever recorded while an entry is open -- so the event always lies behind every
// launch of the producer it covers.
void range_mark_before_launch() {                                          // TraceScope: in front of every instrumented launch
    RangeMark& m = g_rmark[cur_dev()];
    if (!m.dirty.load(std::memory_order_relaxed) || m.fresh.load(std::memory_order_relaxed)) return;
    const int e = m.expect.load(std::memory_order_relaxed);
    if (e > 0 && m.count.load(std::memory_order_relaxed) >= e) range_mark_record(m);
}
void range_mark_entry_done() {                                             // MI355_LAUNCH_CHECK: the launches of the current entry are all enqueued
    RangeMark& m = g_rmark[cur_dev()];
    if (!m.fresh.load(std::memory_order_relaxed)) return;
    m.fresh.store(0, std::memory_order_relaxed);
    m.dirty.store(1, std::memory_order_relaxed);
}
unsigned* range_word(hipStream_t st) {
    if (!g_sync_block && stream_is_capturing(st)) return nullptr;         // never allocate pinned memory inside a capture
    unsigned* w = sync_err_word();
    if (w) {
        RangeMark& m = g_rmark[cur_dev()];
        if (m.armed.load(std::memory_order_relaxed)) {
            m.st = st;
            m.count.fetch_add(1, std::memory_order_relaxed);
            m.fresh.store(1, std::memory_order_relaxed);                   // an open producer entry ("pending")
        }
    }
    return w ? w + 4 : nullptr;                                            // second quarter of the device's 64 bytes
}
int range_arm(int on) {
    RangeMark& m = g_rmark[cur_dev()];
    m.dirty.store(0, std::memory_order_relaxed);
    m.fresh.store(0, std::memory_order_relaxed);
    m.have.store(0, std::memory_order_relaxed);
    if (on) m.count.store(0, std::memory_order_relaxed);                   // a disarm keeps the count for mi355_range_launches()
    m.expect.store(on > 1 ? on - 1 : 0, std::memory_order_relaxed);
    m.armed.store(on ? 1 : 0, std::memory_order_relaxed);
    return MI355_OK;
}
long range_launches() { return g_rmark[cur_dev()].count.load(std::memory_order_relaxed); }
int range_wait() {
    RangeMark& m = g_rmark[cur_dev()];
    if (m.fresh.load(std::memory_order_relaxed)) { m.fresh.store(0, std::memory_order_relaxed); m.dirty.store(1, std::memory_order_relaxed); }
    if (m.dirty.load(std::memory_order_relaxed)) range_mark_record(m);     // no event behind the last producer yet: at the tail of the stream
    if (m.have.load(std::memory_order_relaxed) && m.ev) {
        if (hipEventSynchronize(m.ev) != hipSuccess) return fail(MI355_EHIP, "mi355_range_wait: hipEventSynchronize -> %s", hipGetErrorString(hipGetLastError()));
        m.have.store(0, std::memory_order_relaxed);
    }
    return range_pending("mi355_range_wait");
}
int range_pending(const char* who) {
    unsigned* w = g_sync_block ? g_sync_block + 16 * cur_dev() + 4 : nullptr;
    if (!w || !__atomic_load_n(w, __ATOMIC_ACQUIRE)) return MI355_OK;
    const unsigned code = __atomic_exchange_n(w, 0u, __ATOMIC_ACQ_REL);
    if (!code) return MI355_OK;
    static const char* const names[] = {"?", "mi355_cast16_fwd", "mi355_layernorm16_fwd", "a 16-bit-output GEMM epilogue (mi355_linear16_fwd family)",
                                        "a fused block kernel (mi355_mlp_fused_fwd / mi355_proj_mlp_fused_fwd / mi355_cswin_stripe_attn_fwd / "
                                        "mi355_ln_linear16_fwd / mi355_layernorm16_t_fwd / mi355_mixer_token_fwd)",
                                        "the LayerNorm-folding GEMM epilogue (mi355_linear16_lnc_fwd / mi355_ln_center16_fwd)"};
    return fail(MI355_ERANGE, "%s: an EARLIER launch of %s converted a finite value of magnitude >= 65520 to fp16: that tensor holds inf "
                "where the fp32 reference is finite.  Run the module in precision 0 (strict) or 2 (bf16)", who, names[code < 6 ? code : 0]);
}
unsigned spin_limit() { return (unsigned)opt(O_SPIN_LIMIT); }
int sync_pending(const char* who) {
    unsigned* w = sync_err_word();
    if (!w) return MI355_OK;
    if (!__atomic_load_n(w, __ATOMIC_ACQUIRE)) return MI355_OK;
    const unsigned code = __atomic_exchange_n(w, 0u, __ATOMIC_ACQ_REL);       // read AND clear: a code stored in between is not lost
    if (!code) return MI355_OK;
    static const char* const names[] = {"?", "SE (se_single_kernel)", "CBAM (cbam_single_kernel)", "channel-statistics gate (stat_single_kernel)",
                                        "the persistent GEMM's split last round (gemm16_p8_kernel)"};
    return fail(MI355_ESYNC, "%s: an inter-workgroup exchange of an EARLIER launch of %s ran out of its poll budget (%u sweeps): that launch's "
                "output is invalid.  Typical cause: fewer workgroups resident than one image needs (partitioned / masked device)",
                who, names[code < 5 ? code : 0], spin_limit());
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device and is a LIMIT: remember the largest byte count set per
// (kernel, device ordinal) and raise it when a later launch asks for more (a kernel whose dynamic LDS depends on the shape --
// double_attn_small: HW * 64 + 43008 -- first run on a small image would otherwise fail on a larger one).
int func_dynamic_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, std::unordered_map<int, int>> done;      // kernel -> device ordinal -> bytes configured
    const int dev = cur_dev();
    std::lock_guard<std::mutex> lk(mu);
    int& have = done[fn][dev];
    if (have >= bytes) return MI355_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(MI355_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize, %d) -> %s", bytes, hipGetErrorString(e));
    have = bytes;
    return MI355_OK;
}
// ---- in-process kernel tally (common.h TraceScope) --------------------------------------------------------------------------------
namespace {
struct TraceRec { std::string tag; hipEvent_t a, b; };
std::atomic<int> g_trace_dev{-1};             // device ordinal with an open trace, -1 = none
std::mutex g_trace_mu;
std::vector<TraceRec> g_trace;
std::string g_trace_report;                   // finished report of the last closed trace, kept until it has been handed out whole
bool g_trace_report_pending = false;
}  // namespace
bool trace_on() { return g_trace_dev.load(std::memory_order_relaxed) >= 0; }
TraceScope::TraceScope(hipStream_t st_, const char* fmt, ...) : idx(-1), st(st_) {
    range_mark_before_launch();
    const int dev = g_trace_dev.load(std::memory_order_relaxed);
    if (dev < 0 || dev != cur_dev() || stream_is_capturing(st_)) return;
    char tag[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tag, sizeof(tag), fmt, ap);
    va_end(ap);
    TraceRec r{tag, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess || hipEventRecord(r.a, st_) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    std::lock_guard<std::mutex> lk(g_trace_mu);
    idx = (int)g_trace.size();
    g_trace.push_back(r);
}
TraceScope::~TraceScope() {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_trace_mu);
    if (idx < (int)g_trace.size()) (void)hipEventRecord(g_trace[idx].b, st);
}
int trace_begin() {
    std::lock_guard<std::mutex> lk(g_trace_mu);
    for (auto& r : g_trace) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_trace.clear();
    g_trace_report.clear();
    g_trace_report_pending = false;
    g_trace_dev.store(cur_dev(), std::memory_order_relaxed);
    return MI355_OK;
}
// closes the trace, waits for the recorded launches and writes one line per tag: "count\ttotal_us\tmin_us\tmax_us\ttag\n", largest
// total first; returns the number of bytes the full report needs (like snprintf).  A report that did not fit (or buf == NULL) is KEPT:
// the caller sizes a buffer from the return value and calls again; it is dropped once it has been handed out whole, or by the next
// mi355_trace_begin.
long trace_end(char* buf, size_t n) {
    g_trace_dev.store(-1, std::memory_order_relaxed);
    std::lock_guard<std::mutex> lk(g_trace_mu);
    if (!g_trace.empty() || !g_trace_report_pending) {
        struct Acc { long cnt; double tot, mn, mx; };
        std::map<std::string, Acc> acc;
        for (auto& r : g_trace) {
            float ms = 0.f;
            if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
                auto it = acc.find(r.tag);
                const double us = ms * 1e3;
                if (it == acc.end()) acc[r.tag] = Acc{1, us, us, us};
                else { it->second.cnt++; it->second.tot += us; if (us < it->second.mn) it->second.mn = us; if (us > it->second.mx) it->second.mx = us; }
            } else {
                (void)hipGetLastError();
            }
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
        g_trace.clear();
        std::vector<std::pair<std::string, Acc>> rows(acc.begin(), acc.end());
        std::sort(rows.begin(), rows.end(), [](const auto& x, const auto& y) { return x.second.tot > y.second.tot; });
        g_trace_report.clear();
        char line[256];
        for (auto& kv : rows) {
            snprintf(line, sizeof(line), "%ld\t%.1f\t%.1f\t%.1f\t%s\n", kv.second.cnt, kv.second.tot, kv.second.mn, kv.second.mx, kv.first.c_str());
            g_trace_report += line;
        }
        g_trace_report_pending = true;
    }
    const std::string& out = g_trace_report;
    const long need = (long)out.size();
    if (buf && n) {
        const size_t m = out.size() < n - 1 ? out.size() : n - 1;
        std::memcpy(buf, out.data(), m);
        buf[m] = 0;
        if (m == out.size()) {                       // handed out whole: nothing to keep
            g_trace_report.clear();
            g_trace_report_pending = false;
        }
    }
    return need;
}

int resident_slots(int per_cu) {
    int dev = 0, ncu = 256;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); ncu = 256; }
    return ncu * per_cu;
}

long opt_zoo_single() { return opt(O_ZOO_SINGLE); }
long opt_stem_direct() { return opt(O_STEM_DIRECT); }
long opt_se_occ() { return opt(O_SE_OCC); }
long opt_gemm_variant() { return opt(O_GEMM_VARIANT); }
long opt_gemm_splitk() { return opt(O_GEMM_SPLITK); }
long opt_gemm_pa() { return opt(O_GEMM_PA); }
long opt_da_fused() { return opt(O_DA_FUSED); }
long opt_da_ranges() { return opt(O_DA_RANGES); }
static int opt_index(const char* key) {
    for (int i = 0; i < O_COUNT; ++i)
        if (std::strcmp(key, kOpts[i].key) == 0) return i;
    return -1;
}
int opt_set(const char* key, long value, bool as_default) {
    const int i = opt_index(key);
    if (i < 0) return fail(MI355_EINVAL, "mi355_set_option: unknown key '%s'", key);
    const bool ok = (value >= kOpts[i].lo && value <= kOpts[i].hi) || (i == O_SPIN_LIMIT && value == 0);
    if (!ok) return fail(MI355_EINVAL, "mi355_set_option: '%s' accepts %ld .. %ld, got %ld", key, kOpts[i].lo, kOpts[i].hi, value);
    opt_init();
    if (as_default) g_def[i].store(value, std::memory_order_relaxed);
    else g_opt[cur_dev()].v[i].store(value, std::memory_order_relaxed);
    return MI355_OK;
}
long opt_get(const char* key) {
    const int i = key ? opt_index(key) : -1;
    if (i < 0) {
        fail(MI355_EINVAL, "mi355_get_option: unknown key '%s'", key ? key : "(null)");
        return -1;
    }
    return opt((Opt)i);
}
}  // namespace mi355

struct mi355_timer {
    hipEvent_t a, b;
};

extern "C" {

int mi355_version(void) { return MI355_ABI_VERSION; }
const char* mi355_last_error(void) { return mi355::err_buf(); }

int mi355_set_option(const char* key, long value) {
    MI355_CHECK_ARG(key != nullptr);
    return mi355::opt_set(key, value, false);
}

int mi355_set_default_option(const char* key, long value) {
    MI355_CHECK_ARG(key != nullptr);
    return mi355::opt_set(key, value, true);
}

int mi355_workspace_forget(const void* ws, size_t ws_bytes) {
    MI355_CHECK_ARG(ws != nullptr);
    mi355::ws_forget_range(ws, ws_bytes);
    return MI355_OK;
}

long mi355_get_option(const char* key) { return mi355::opt_get(key); }

int mi355_trace_begin(void) { return mi355::trace_begin(); }
long mi355_trace_end(char* report, size_t report_bytes) { return mi355::trace_end(report, report_bytes); }

int mi355_sync_status(void) { return mi355::sync_pending("mi355_sync_status"); }
int mi355_range_status(void) { return mi355::range_pending("mi355_range_status"); }
// SURVEY.md 8(b) spellings (include/mi355attn.h, last section): same arguments, same code
int mi355_sdpa_core_fwd(const float* qkv, float* out, int B, int N, int heads, int d, float scale, int precision, mi355_stream_t stream) {
    return mi355_sdpa_fwd(qkv, out, B, N, heads, d, scale, precision, stream);
}
size_t mi355_sdpa_core_workspace_bytes(int, int, int, int) { return 0; }
int mi355_gemm_bias_act_fwd(const float* X, const float* W, const float* bias, const float* gamma, const float* resid, float* Y, int M, int N,
                            int K, int ldx, int ldy, int act, int precision, mi355_stream_t stream) {
    return mi355_linear_fwd(X, W, bias, gamma, resid, Y, M, N, K, ldx, ldy, act, precision, stream);
}
size_t mi355_gemm_bias_act_workspace_bytes(int, int, int) { return 0; }
int mi355_mixer_token_mlp_fwd(const float* x, const float* ln_w, const float* ln_b, float ln_eps, const void* w1p16, const float* b1,
                              const void* w2s16, const float* b2, float* y, int B, int N, int C, int T, int precision, void* ws,
                              size_t ws_bytes, mi355_stream_t stream) {
    return mi355_mixer_token_fwd(x, ln_w, ln_b, ln_eps, w1p16, b1, w2s16, b2, y, B, N, C, T, precision, ws, ws_bytes, stream);
}
size_t mi355_mixer_token_mlp_workspace_bytes(int B, int N, int C) { return mi355_mixer_token_workspace_bytes(B, N, C); }
size_t mi355_cswin_lepe_attn_workspace_bytes(int, int, int) { return 0; }
size_t mi355_xca_workspace_bytes(int, int, int, int) { return 0; }
size_t mi355_layernorm_workspace_bytes(int, int) { return 0; }
int mi355_range_arm(int on) { return mi355::range_arm(on); }
int mi355_range_wait(void) { return mi355::range_wait(); }
long mi355_range_launches(void) { return mi355::range_launches(); }

int mi355_event_time_begin(mi355_stream_t stream, void** handle) {
    MI355_CHECK_ARG(handle != nullptr);
    mi355_timer* t = new mi355_timer;
    hipError_t e = hipEventCreate(&t->a);
    if (e == hipSuccess) e = hipEventCreate(&t->b);
    if (e == hipSuccess) e = hipEventRecord(t->a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        delete t;
        return mi355::fail(MI355_EHIP, "mi355_event_time_begin: %s", hipGetErrorString(e));
    }
    *handle = t;
    return MI355_OK;
}

int mi355_event_time_end(mi355_stream_t stream, void* handle, float* ms_out) {
    MI355_CHECK_ARG(handle != nullptr && ms_out != nullptr);
    mi355_timer* t = static_cast<mi355_timer*>(handle);
    hipError_t e = hipEventRecord(t->b, static_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = hipEventSynchronize(t->b);
    if (e == hipSuccess) e = hipEventElapsedTime(ms_out, t->a, t->b);
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
    if (e != hipSuccess) return mi355::fail(MI355_EHIP, "mi355_event_time_end: %s", hipGetErrorString(e));
    return MI355_OK;
}

}  // extern "C"
