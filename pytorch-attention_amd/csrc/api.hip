// api.hip -- library-level entry points: version, thread-local error text, tuning options, HIP-event stopwatch.
#include "common.h"
#include <atomic>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace mi355 {
static thread_local char g_err[512] = "";
static std::atomic<long> g_chunk_images{0};   // 0 = auto (about 200 MB of x per chunk)
static std::atomic<long> g_nt{3};        // bit0: non-temporal loads, bit1: non-temporal stores in the final pass
static std::atomic<long> g_reverse{0};
static std::atomic<long> g_gemm_variant{0};   // tile/schedule variant of the 16-bit GEMM (gemm16.hip)

static std::atomic<long> g_eca_single{1};   // ECA: one read + one write of x, halo channel rows re-summed per workgroup (chan_fused.hip)
static std::atomic<long> g_se_single{1};    // SE: x read once, channel means exchanged as 8-byte {mean, tag} granules (chan_fused.hip)
static std::atomic<long> g_cbam_single{1};  // CBAM: x read once, row bands in registers, three granule hops per band (cbam_single.hip)
static std::atomic<long> g_ws_persistent{0};  // 1 = caller keeps workspace contents between calls: granule exchanges skip their memset
static std::atomic<long> g_stem_direct{1};  // narrow conv stems: direct fp32 kernel (stem_conv.hip) vs implicit GEMM
static std::atomic<long> g_zoo_single{1};   // SimAM / SRM / GCT / LCT: single-read register-resident path (chan_stat.hip) vs two passes
static std::atomic<long> g_spin_limit{1L << 22};   // poll budget of the exchange kernels (sweeps) before they give up with an error code
static std::atomic<long> g_gemm_pa{1};       // fp32-output GEMMs: two-accumulator persistent kernel where it applies (gemm16_pa.hip)
static std::atomic<long> g_gemm_splitk{1};   // persistent GEMM: cut the tiles of the last partial round along K (gemm16_p8.hip)
std::atomic<long> g_da_fused{1};       // DoubleAttention: two-pass kernels where they apply (double_attn_fused.hip)
std::atomic<long> g_da_ranges{0};      // ... pixel ranges per image in pass 1: 0 = from the batch size, 1..32 = fixed
static std::atomic<long> g_se_occ{3};        // single-read SE: workgroups per CU (2: <= 128 VGPRs, 3: <= 80 VGPRs)

char* err_buf() { return g_err; }

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
long opt_chunk_images() { return g_chunk_images.load(std::memory_order_relaxed); }
long opt_nt() { return g_nt.load(std::memory_order_relaxed); }
long opt_reverse() { return g_reverse.load(std::memory_order_relaxed); }
long opt_eca_single() { return g_eca_single.load(std::memory_order_relaxed); }
long opt_se_single() { return g_se_single.load(std::memory_order_relaxed); }
long opt_cbam_single() { return g_cbam_single.load(std::memory_order_relaxed); }
long opt_ws_persistent() { return g_ws_persistent.load(std::memory_order_relaxed); }

// ---- epochs of the granule-exchange workspaces (chan_fused.hip, cbam_single.hip) ---------------------------------------------
// A granule is valid when it carries the tag of the CURRENT launch.  With a fresh tag per launch a slot written by any earlier
// launch -- completed or not -- can never look valid, so the region only has to be zeroed when its history is unknown: first
// use of the pointer, a different layout key, ticket counter about to wrap, or "ws_persistent" off (the default: a C caller
// that frees / reuses workspace memory between calls must not opt in).  The ticket word keeps counting across launches;
// each launch subtracts the base it was handed.
namespace {
struct WsEntry { unsigned long long key; unsigned ticket_end; };
std::mutex g_ws_mu;
std::unordered_map<const void*, WsEntry> g_ws;
std::atomic<unsigned> g_epoch{0x5EC0DE00u};
}  // namespace

// A launch recorded by hipGraph capture replays with the SAME kernel arguments, so it cannot carry a per-launch tag or a ticket base
// remembered on the host, and a captured memset -> kernel pair was observed to misbehave on replay (MI355X, ROCm 7.2: replays after an
// intervening eager launch returned stale output; tests/test_chan_attn_gpu.py::test_exchange_kernels_under_graph_capture).  The
// exchange kernels therefore step aside under capture: their callers take the multi-pass paths, which only use kernel-to-kernel
// dependencies through memory.  ws_epoch still answers safely (constant tag, zero base, nothing remembered) if it is ever asked.
bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return cs != hipStreamCaptureStatusNone;
}

WsEpoch ws_epoch(const void* region, unsigned long long key, unsigned draws, hipStream_t st) {
    WsEpoch r{};
    if (stream_is_capturing(st)) {
        r.tag = 0x6A9F0001u;
        r.fresh = true;
        r.ticket_base = 0u;
        std::lock_guard<std::mutex> lk(g_ws_mu);
        g_ws.erase(region);
        return r;
    }
    unsigned tag = g_epoch.fetch_add(1u, std::memory_order_relaxed) + 1u;
    if (tag == 0u) tag = g_epoch.fetch_add(1u, std::memory_order_relaxed) + 1u;     // 0 is what a zeroed slot holds
    r.tag = tag;
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto it = g_ws.find(region);
    const bool known = opt_ws_persistent() && it != g_ws.end() && it->second.key == key &&
                       it->second.ticket_end < 0x7FFFFFFFu - draws && tag > 0x1000u /* tags wrapped: start over */;
    r.fresh = !known;
    r.ticket_base = known ? it->second.ticket_end : 0u;
    if (opt_ws_persistent()) g_ws[region] = WsEntry{key, r.ticket_base + draws};
    else if (it != g_ws.end()) g_ws.erase(it);             // zeroed on every call from now on: what was remembered is void
    return r;
}
// Workspaces of the kernels that keep their launch tag in device memory (se_single_kernel, cbam_single_kernel): the host only has to know whether the
// region was zeroed for this shape.  Under stream capture an unknown region stays unknown (the memset the caller records runs at
// replay time, not now); a known one needs nothing.
bool ws_known(const void* region, unsigned long long key, hipStream_t st) {
    if (!opt_ws_persistent()) return false;
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto it = g_ws.find(region);
    if (it != g_ws.end() && it->second.key == key) return true;
    if (!stream_is_capturing(st)) g_ws[region] = WsEntry{key, 0u};
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

// ---- exchange-kernel failure word (common.h) -----------------------------------------------------------------------------------
namespace {
std::once_flag g_sync_once;
unsigned* g_sync_word = nullptr;
}  // namespace
unsigned* sync_err_word() {
    std::call_once(g_sync_once, [] {
        void* p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) == hipSuccess && p) {
            std::memset(p, 0, 64);
            g_sync_word = static_cast<unsigned*>(p);
        } else {
            (void)hipGetLastError();
        }
    });
    return g_sync_word;
}
unsigned* sync_err_word_on(hipStream_t st) {
    if (!g_sync_word && stream_is_capturing(st)) return nullptr;          // never allocate pinned memory inside a capture
    return sync_err_word();
}
unsigned* range_word(hipStream_t st) {
    if (!g_sync_word && stream_is_capturing(st)) return nullptr;          // never allocate pinned memory inside a capture
    unsigned* w = sync_err_word();
    return w ? w + 4 : nullptr;                                            // second quarter of the 64-byte pinned block
}
int range_pending(const char* who) {
    unsigned* w = g_sync_word ? g_sync_word + 4 : nullptr;
    if (!w || !__atomic_load_n(w, __ATOMIC_ACQUIRE)) return MI355_OK;
    const unsigned code = __atomic_exchange_n(w, 0u, __ATOMIC_ACQ_REL);
    if (!code) return MI355_OK;
    static const char* const names[] = {"?", "mi355_cast16_fwd", "mi355_layernorm16_fwd", "a 16-bit-output GEMM epilogue (mi355_linear16_fwd family)"};
    return fail(MI355_ERANGE, "%s: an EARLIER launch of %s converted a finite value of magnitude >= 65520 to fp16: that tensor holds inf "
                "where the fp32 reference is finite.  Run the module in precision 0 (strict) or 2 (bf16)", who, names[code < 4 ? code : 0]);
}
unsigned spin_limit() { return (unsigned)g_spin_limit.load(std::memory_order_relaxed); }
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
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device: remember (kernel, device ordinal) pairs, not a
// per-process flag -- a second GPU driven from the same process would otherwise launch with the default dynamic-LDS limit and fail.
int func_dynamic_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, unsigned long long> done;      // kernel -> bit mask of device ordinals (< 64)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    const unsigned long long bit = 1ull << (dev & 63);
    std::lock_guard<std::mutex> lk(mu);
    unsigned long long& m = done[fn];
    if (m & bit) return MI355_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(MI355_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize, %d) -> %s", bytes, hipGetErrorString(e));
    m |= bit;
    return MI355_OK;
}
int resident_slots(int per_cu) {
    int dev = 0, ncu = 256;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); ncu = 256; }
    return ncu * per_cu;
}

long opt_zoo_single() { return g_zoo_single.load(std::memory_order_relaxed); }
long opt_stem_direct() { return g_stem_direct.load(std::memory_order_relaxed); }
long opt_se_occ() { return g_se_occ.load(std::memory_order_relaxed); }
long opt_gemm_variant() { return g_gemm_variant.load(std::memory_order_relaxed); }
long opt_gemm_splitk() { return g_gemm_splitk.load(std::memory_order_relaxed); }
long opt_gemm_pa() { return g_gemm_pa.load(std::memory_order_relaxed); }
long opt_da_fused() { return g_da_fused.load(std::memory_order_relaxed); }
long opt_da_ranges() { return g_da_ranges.load(std::memory_order_relaxed); }
}  // namespace mi355

struct mi355_timer {
    hipEvent_t a, b;
};

extern "C" {

int mi355_version(void) { return MI355_ABI_VERSION; }
const char* mi355_last_error(void) { return mi355::err_buf(); }

int mi355_set_option(const char* key, long value) {
    MI355_CHECK_ARG(key != nullptr);
    if (std::strcmp(key, "chunk_images") == 0) {
        MI355_CHECK_ARG(value >= 0);
        mi355::g_chunk_images.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "nt") == 0) {
        MI355_CHECK_ARG(value >= 0 && value <= 3);
        mi355::g_nt.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "gemm_variant") == 0) {
        MI355_CHECK_ARG(value >= 0 && value <= 31);
        mi355::g_gemm_variant.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "eca_single") == 0) {
        MI355_CHECK_ARG(value == 0 || value == 1);
        mi355::g_eca_single.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "se_single") == 0) {
        MI355_CHECK_ARG(value == 0 || value == 1);
        mi355::g_se_single.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "cbam_single") == 0) {
        MI355_CHECK_ARG(value == 0 || value == 1);
        mi355::g_cbam_single.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "ws_persistent") == 0) {
        MI355_CHECK_ARG(value == 0 || value == 1);
        mi355::g_ws_persistent.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "zoo_single") == 0) {
        MI355_CHECK_ARG(value == 0 || value == 1);
        mi355::g_zoo_single.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "stem_direct") == 0) {
        MI355_CHECK_ARG(value == 0 || value == 1);
        mi355::g_stem_direct.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "spin_limit") == 0) {
        // 0 is accepted for ONE purpose: forcing the time-out path in tests (every exchange then fails on its first unsuccessful poll);
        // real budgets start at 1024 sweeps
        MI355_CHECK_ARG(value == 0 || (value >= 1024 && value <= (1L << 30)));
        mi355::g_spin_limit.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "gemm_splitk") == 0) {
        MI355_CHECK_ARG(value == 0 || value == 1);
        mi355::g_gemm_splitk.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "gemm_pa") == 0) {
        MI355_CHECK_ARG(value == 0 || value == 1);
        mi355::g_gemm_pa.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "da_fused") == 0) {
        MI355_CHECK_ARG(value == 0 || value == 1);
        mi355::g_da_fused.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "da_ranges") == 0) {
        MI355_CHECK_ARG(value >= 0 && value <= 32);
        mi355::g_da_ranges.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "se_occ") == 0) {
        MI355_CHECK_ARG(value == 2 || value == 3);
        mi355::g_se_occ.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    if (std::strcmp(key, "reverse") == 0) {
        MI355_CHECK_ARG(value == 0 || value == 1);
        mi355::g_reverse.store(value, std::memory_order_relaxed);
        return MI355_OK;
    }
    return mi355::fail(MI355_EINVAL, "mi355_set_option: unknown key '%s'", key);
}

int mi355_workspace_forget(const void* ws, size_t ws_bytes) {
    MI355_CHECK_ARG(ws != nullptr);
    mi355::ws_forget_range(ws, ws_bytes);
    return MI355_OK;
}

long mi355_get_option(const char* key) {
    if (key && std::strcmp(key, "chunk_images") == 0) return mi355::opt_chunk_images();
    if (key && std::strcmp(key, "nt") == 0) return mi355::opt_nt();
    if (key && std::strcmp(key, "reverse") == 0) return mi355::opt_reverse();
    if (key && std::strcmp(key, "gemm_variant") == 0) return mi355::opt_gemm_variant();
    if (key && std::strcmp(key, "eca_single") == 0) return mi355::opt_eca_single();
    if (key && std::strcmp(key, "se_single") == 0) return mi355::opt_se_single();
    if (key && std::strcmp(key, "se_occ") == 0) return mi355::opt_se_occ();
    if (key && std::strcmp(key, "gemm_splitk") == 0) return mi355::opt_gemm_splitk();
    if (key && std::strcmp(key, "gemm_pa") == 0) return mi355::opt_gemm_pa();
    if (key && std::strcmp(key, "da_fused") == 0) return mi355::opt_da_fused();
    if (key && std::strcmp(key, "da_ranges") == 0) return mi355::opt_da_ranges();
    if (key && std::strcmp(key, "spin_limit") == 0) return (long)mi355::spin_limit();
    if (key && std::strcmp(key, "zoo_single") == 0) return mi355::opt_zoo_single();
    if (key && std::strcmp(key, "stem_direct") == 0) return mi355::opt_stem_direct();
    if (key && std::strcmp(key, "ws_persistent") == 0) return mi355::opt_ws_persistent();
    if (key && std::strcmp(key, "cbam_single") == 0) return mi355::opt_cbam_single();
    mi355::fail(MI355_EINVAL, "mi355_get_option: unknown key '%s'", key ? key : "(null)");
    return -1;
}

int mi355_sync_status(void) { return mi355::sync_pending("mi355_sync_status"); }
int mi355_range_status(void) { return mi355::range_pending("mi355_range_status"); }

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
