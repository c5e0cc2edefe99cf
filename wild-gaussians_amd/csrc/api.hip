// C-ABI entry points (include/wg_rasterizer.h) -- the orchestration that Rasterizer::forward / ::backward /
// ::markVisible do in the reference (rasterizer_impl.cu:141-153, 198-340, 344-443).
#include "wg_common.h"

#include <dlfcn.h>
#include <limits>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_last_hip_error;

int hip_fail(hipError_t e, const char* where) {
    g_last_hip_error = std::string(where) + ": " + hipGetErrorString(e);
    return WG_ERR_HIP;
}

// ---- optional per-stage event timing (wg_profile_*) ----
struct StageProfiler {
    bool enabled = false;
    std::mutex mu;
    struct Rec { int stage; hipEvent_t a, b; int dev; };
    std::vector<Rec> pending;
    std::map<int, std::vector<hipEvent_t>> pool;  // per device: an event belongs to the device it was created on
    wg_stage_times totals{};
    static int device() {
        int d = 0;
        (void)hipGetDevice(&d);
        return d;
    }
    hipEvent_t get(int dev) {
        auto& p = pool[dev];
        if (!p.empty()) { hipEvent_t e = p.back(); p.pop_back(); return e; }
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};
StageProfiler g_prof;

// The backward pass of a deferred frame usually runs on ANOTHER host thread (torch's autograd engine): it finds the frame by its image
// buffer here.  One ticket per forward thread (its latest deferred frame: earlier ones were settled by that thread's later calls).
struct DeferredTicket {
    const void* image_buffer;
    const wg::HostMailbox* host;
    uint32_t seq;
    uint32_t reported_seq;  // the frame whose failure a backward call has already returned (the owning thread's settle then stays quiet)
    bool reported;
    hipStream_t stream;     // the stream the deferred forward call was queued on (what a waiting backward-side check synchronises)
};
std::mutex g_ticket_mu;
std::vector<DeferredTicket> g_tickets;

// One pinned mailbox per host thread (see wg::HostMailbox).  nullptr if pinned memory is unavailable: the forward pass
// then falls back to hipMemcpyAsync + hipStreamSynchronize.  The pinned words outlive their thread: a thread that ends hands its
// mailbox (with its sequence counter, which therefore never runs backwards) to a process-wide free list for the next thread that
// needs one, and nothing is ever given back to the runtime -- a backward call on another thread may be polling those words at that
// very moment (check_ticket), and thread-local destructors of a process that is shutting down run after the HIP runtime has gone.
struct MailboxSlot { wg::HostMailbox* host; wg::HostMailbox* dev; uint32_t seq; };
std::mutex g_mailbox_mu;
std::vector<MailboxSlot> g_mailbox_free;
struct Mailbox {
    wg::HostMailbox* host = nullptr;
    wg::HostMailbox* dev = nullptr;
    uint32_t seq = 0;
    bool tried = false;
    ~Mailbox() {
        if (!host) return;
        {   // the thread's deferred-frame ticket goes with it (its frames' verdicts were this thread's to report)
            std::lock_guard<std::mutex> l(g_ticket_mu);
            for (size_t i = 0; i < g_tickets.size();)
                if (g_tickets[i].host == host) g_tickets.erase(g_tickets.begin() + (long)i);
                else i++;
        }
        std::lock_guard<std::mutex> l(g_mailbox_mu);
        g_mailbox_free.push_back({host, dev, seq});
    }
};
thread_local Mailbox t_mailbox;
thread_local uint32_t t_last_instances_per_tile = 0;  // density of this thread's previous frame: the near / far split's "try it" hint
thread_local uint32_t t_split_backoff = 0;            // frames for which the split is not attempted after one that needed the far phase
// The split's aimed number of near instances per tile, ADAPTED per host thread (option "near_adapt", default on; a fixed "near_per_tile" wins):
// every split frame reports how many tiles ran out of near instances and the aim it ran with (the mailbox's far_report, read a frame or two
// late).  Four clean reports at the current aim lower it by 10 %; a frame in which ANY tile asked lifts it to a floor an eighth (a sixteenth for a
// handful of tiles) above the aim THAT frame ran with, where it then stays (the floor is relaxed by a twentieth every 2048 frames).  Never above the
// fixed default (1.1 x the front target): the adaptation can only shorten what
// is scattered and sorted -- at 10 M Gaussians / 4K pixels stop ~215 instances deep and 900 near instances per tile were twice what the
// deepest tile needed.  Results do not depend on it (a tile that runs out gets its far instances: the far phase).
// What it has learnt belongs to one SCENE: a cloud outside [P / 2, 2 P] of the one it learnt from starts it afresh (bench.py's config legs run a
// 3 M-Gaussian frame and then a 10 M-Gaussian frame on one thread: the first one's floor of 698 sat under the second, 2.03 instead of 1.84 ms per
// frame).  Not the image size: a training run's cameras may differ in theirs (Photo Tourism) while the depth at which pixels stop is the scene's.
struct NearAdapt {
    uint32_t cur = 0, floor = 0, clean = 0, age = 0, last_far = 0;
    int P = 0;
    bool same_scene(int P_) const { return P != 0 && (int64_t)P_ * 2 >= (int64_t)P && (int64_t)P_ <= (int64_t)P * 2; }
};
thread_local NearAdapt t_near;

// Speculative forward: what this host thread's recent frames looked like.  A prediction is made only from frames of the same image
// size and a similar number of Gaussians (training: the model grows slowly, the cameras alternate -- hence the maximum over the last
// eight frames, not the last one).
struct SpecHistory {
    static constexpr int N = 8;
    int W = 0, H = 0, P = 0, n = 0, head = 0;
    uint32_t rendered[N] = {}, longest[N] = {};
    bool usable(int P_, int W_, int H_) const {
        return n > 0 && W_ == W && H_ == H && (int64_t)P_ * 2 >= (int64_t)P && (int64_t)P_ <= (int64_t)P * 2;
    }
    void push(int P_, int W_, int H_, uint32_t R, uint32_t L) {
        if (W_ != W || H_ != H) { n = 0; head = 0; W = W_; H = H_; }
        P = P_;
        rendered[head] = R; longest[head] = L;
        head = (head + 1) % N;
        if (n < N) n++;
    }
    uint32_t max_rendered() const { uint32_t m = 0; for (int i = 0; i < n; i++) m = std::max(m, rendered[i]); return m; }
    uint32_t max_longest() const { uint32_t m = 0; for (int i = 0; i < n; i++) m = std::max(m, longest[i]); return m; }
    uint32_t last_rendered() const { return n ? rendered[(head + N - 1) % N] : 0u; }
    void clear() { n = 0; head = 0; }
};
thread_local SpecHistory t_spec;
// How long the forward calls of this thread waited for the count (wg_get_option "forward_wait_*": evidence for "no host wait in
// steady state"), and how the speculation fared.
struct WaitStats {
    uint64_t polls = 0, waited = 0, spec_frames = 0, spec_misses = 0;
    double wait_us = 0.0, last_wait_us = 0.0;
    template <typename D>
    void record(bool spun, D d) {
        polls += 1;
        last_wait_us = std::chrono::duration<double, std::micro>(d).count();
        if (spun) { waited += 1; wait_us += last_wait_us; }
    }
    void clear() { *this = WaitStats(); }
};
thread_local WaitStats t_wait;
// Deferred speculation (option "speculative_forward" = 2): the last forward call of this thread returned without looking at its frame's
// verdict.  It is looked at by the next forward / backward call of the thread (settle_deferred, below).
struct Deferred {
    bool pending = false;
    uint32_t seq = 0;
    int P = 0, W = 0, H = 0;
    hipStream_t stream = nullptr;  // the stream the frame was enqueued on (the next call may come with another one)
};
thread_local Deferred t_deferred;

// the options (wg_common.h: Options): written by wg_set_option under the mutex, copied once per call
std::mutex g_opt_mu;
wg::Options g_opt;
wg::Options options_snapshot() {
    std::lock_guard<std::mutex> l(g_opt_mu);
    return g_opt;
}

Mailbox* get_mailbox() {
    Mailbox& m = t_mailbox;
    if (!m.tried) {
        m.tried = true;
        {
            std::lock_guard<std::mutex> l(g_mailbox_mu);
            if (!g_mailbox_free.empty()) {
                const MailboxSlot s = g_mailbox_free.back();
                g_mailbox_free.pop_back();
                m.host = s.host; m.dev = s.dev; m.seq = s.seq;
                m.host->far_report = 0ull;
            }
        }
        void* h = nullptr;
        if (!m.host && hipHostMalloc(&h, sizeof(wg::HostMailbox), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                m.host = static_cast<wg::HostMailbox*>(h);
                m.dev = static_cast<wg::HostMailbox*>(d);
                std::memset(h, 0, sizeof(wg::HostMailbox));
            } else {
                (void)hipHostFree(h);
            }
        }
        (void)hipGetLastError();
    }
    return m.host ? &m : nullptr;
}

// roctx ranges around every stage (wg_set_option("roctx", 1) or WG_ROCTX=1 in the environment): `rocprofv3 --marker-trace
// --kernel-trace` then shows K1...K11 as named host ranges above the kernels they launch.  The marker library is looked up at run
// time (librocprofiler-sdk-roctx.so, else libroctx64.so), so the rasterizer has no link-time dependency on a profiler.
struct Roctx {
    int state = 0;  // 0 = not tried, 1 = available, -1 = unavailable
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool enabled = false;
    bool load() {
        if (state == 0) {
            state = -1;
            for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
                void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (!h) continue;
                push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) { state = 1; break; }
            }
        }
        return state == 1;
    }
};
Roctx g_roctx;
const char* stage_label(int stage);

struct StageScope {
    int stage; hipStream_t stream; hipEvent_t a = nullptr, b = nullptr; bool on; bool marked = false; int dev = 0;
    StageScope(int s, hipStream_t st) : stage(s), stream(st), on(g_prof.enabled) {
        if (g_roctx.enabled) { g_roctx.push(stage_label(s)); marked = true; }
        if (!on) return;
        std::lock_guard<std::mutex> l(g_prof.mu);
        dev = StageProfiler::device();
        a = g_prof.get(dev); b = g_prof.get(dev);
        if (a) (void)hipEventRecord(a, stream);
    }
    ~StageScope() {
        if (marked) g_roctx.pop();
        if (!on || !a || !b) return;
        (void)hipEventRecord(b, stream);
        std::lock_guard<std::mutex> l(g_prof.mu);
        g_prof.pending.push_back({stage, a, b, dev});
    }
};

// debug == true reproduces CHECK_CUDA (auxiliary.h:166-173): synchronise after every stage and report.
#define WG_STAGE(stage_id, expr, name)                                           \
    do {                                                                         \
        hipError_t e_;                                                           \
        {                                                                        \
            StageScope scope_(stage_id, stream);                                 \
            e_ = (expr);                                                         \
        }                                                                        \
        if (e_ != hipSuccess) return hip_fail(e_, name);                         \
        if (debug) {                                                             \
            e_ = hipStreamSynchronize(stream);                                   \
            if (e_ != hipSuccess) return hip_fail(e_, name " (debug sync)");     \
        }                                                                        \
    } while (0)

const char* stage_label(int stage) {
    static const char* names[WG_STAGE_COUNT] = {"wg:K1 preprocess", "wg:K2-K3 scan", "wg:K4 duplicate_keys", "wg:K5 sort", "wg:K6-K7 tile_ranges",
                                                "wg:K8 render_forward", "wg:K9 render_backward", "wg:K10-K11 preprocess_backward",
                                                "wg:K8 render_fixup"};
    return (stage >= 0 && stage < WG_STAGE_COUNT) ? names[stage] : "wg:?";
}

struct RoctxEnv {  // WG_ROCTX=1: ranges on from the first call, without touching the caller
    RoctxEnv() {
        const char* e = std::getenv("WG_ROCTX");
        if (e && e[0] == '1') g_roctx.enabled = g_roctx.load();
    }
} g_roctx_env;

// Reads the mailbox words of frame `seq` (waits for them, bounded).  false: not there within ~2 s.
bool read_mailbox(Mailbox* mbox, uint32_t seq, wg::BinStats& st, bool* waited) {
    volatile unsigned long long* w0 = &mbox->host->word0;
    volatile unsigned long long* w1 = &mbox->host->word1;
    const unsigned long long want = seq;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    unsigned long long a = *w0, b = *w1;
    while ((a >> 32) != want || (b >> 32) != want) {
        if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
        __builtin_ia32_pause();
        a = *w0; b = *w1;
    }
    if (waited) *waited = spins != 0u;
    if ((a >> 32) != want || (b >> 32) != want) return false;
    st.num_rendered = (uint32_t)a;
    st.max_tile_count = (uint32_t)b & wg::MAILBOX_MAX_LIST;   // (saturated at 2^30 - 1: beyond any list the sort paths distinguish)
    st.split_active = (uint32_t)(b >> 30) & 1u;
    st.spec_fail = (uint32_t)(b >> 31) & 1u;
    return true;
}

// A deferred frame's verdict, taken by the thread's next call.  WG_ERR_SPECULATION when that frame did not fit the buffer it was given:
// its image is NaN and its gradients are zero (see wg_forward_args::binning_capacity); the caller repeats the step.  The history learns the
// frame's real size either way, so the repeat fits.
int settle_deferred() {
    if (!t_deferred.pending) return WG_OK;
    t_deferred.pending = false;
    const hipStream_t stream = t_deferred.stream;
    Mailbox& mb = t_mailbox;
    wg::BinStats st{};
    bool ok = mb.host != nullptr && read_mailbox(&mb, t_deferred.seq, st, nullptr);
    if (!ok) {  // the frame's scan has not run within the bound: wait for the stream (a failed launch surfaces here)
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return hip_fail(e, "deferred forward (stream synchronize)");
        ok = mb.host != nullptr && read_mailbox(&mb, t_deferred.seq, st, nullptr);
        if (!ok) return hip_fail(hipErrorUnknown, "deferred forward: the instance count never arrived");
    }
    t_spec.push(t_deferred.P, t_deferred.W, t_deferred.H, st.num_rendered, st.max_tile_count);
    t_last_instances_per_tile = st.num_rendered / (uint32_t)std::max(1, ((t_deferred.W + wg::TILE_X - 1) / wg::TILE_X) * ((t_deferred.H + wg::TILE_Y - 1) / wg::TILE_Y));
    t_wait.spec_frames += 1;
    // The owning thread has the verdict now: the frame's ticket gives its address up (a frame that is never differentiated -- evaluation,
    // no_grad -- would otherwise leave it there for whatever frame the caller's allocator hands the same address to next).
    bool reported_already = false;
    {
        std::lock_guard<std::mutex> l(g_ticket_mu);
        for (auto& k : g_tickets)
            if (k.host == mb.host) {
                if (k.reported && k.reported_seq == t_deferred.seq) reported_already = true;   // its backward call has said so already
                if (k.seq == t_deferred.seq) k.image_buffer = nullptr;
            }
    }
    if (st.spec_fail != 0u) {
        t_wait.spec_misses += 1;
        return reported_already ? WG_OK : WG_ERR_SPECULATION;
    }
    return WG_OK;
}

void post_ticket(const void* image_buffer, const wg::HostMailbox* host, uint32_t seq, hipStream_t stream) {
    std::lock_guard<std::mutex> l(g_ticket_mu);
    for (auto& t : g_tickets)
        if (t.host == host) { t.image_buffer = image_buffer; t.seq = seq; t.stream = stream; return; }
    g_tickets.push_back({image_buffer, host, seq, 0u, false, stream});
}

// Backward side: was `image_buffer` produced by a deferred forward call whose verdict nobody has looked at yet, and did it fit?
// A ticket is checked once: the call that reads its verdict takes the address out of it (scratch addresses come back from the
// caller's allocator for other frames).
int check_ticket(const void* image_buffer, hipStream_t stream) {
    DeferredTicket t{};
    {
        std::lock_guard<std::mutex> l(g_ticket_mu);
        bool found = false;
        for (auto& k : g_tickets)
            if (k.image_buffer == image_buffer) { t = k; found = true; break; }
        if (!found) return WG_OK;
    }
    auto consume = [&](bool failed) {
        std::lock_guard<std::mutex> l(g_ticket_mu);
        for (auto& k : g_tickets)
            if (k.host == t.host && k.seq == t.seq) {
                k.image_buffer = nullptr;
                if (failed) { k.reported = true; k.reported_seq = t.seq; }
            }
    };
    // (t.host stays valid whatever its thread does meanwhile: mailboxes are pooled, never freed)
    volatile const unsigned long long* w0 = &t.host->word0;
    volatile const unsigned long long* w1 = &t.host->word1;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (;;) {
        const unsigned long long a = *w0, b = *w1;
        const uint32_t sa = (uint32_t)(a >> 32), sb = (uint32_t)(b >> 32);
        if (sa == t.seq && sb == t.seq) {
            const bool failed = ((b >> 31) & 1ull) != 0ull;
            consume(failed);
            return failed ? WG_ERR_SPECULATION : WG_OK;
        }
        // a later frame of the owning thread is in the mailbox: that thread's call has settled (and reported) this one already
        if ((int32_t)(sa - t.seq) > 0 || (int32_t)(sb - t.seq) > 0) { consume(false); return WG_OK; }
        if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            hipError_t e = hipStreamSynchronize(t.stream ? t.stream : stream);   // the frame's scan has not run yet: wait for the stream the forward call was queued on, then look again
            if (e != hipSuccess) return hip_fail(e, "deferred forward (backward-side check)");
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(6)) return hip_fail(hipErrorUnknown, "deferred forward: the instance count never arrived");
        }
        __builtin_ia32_pause();
    }
}

template <typename F>
size_t required_bytes(F carve_fn) {
    char* p = nullptr;
    carve_fn(p);
    return reinterpret_cast<size_t>(p) + wg::ALIGN;
}

}  // namespace

extern "C" {

size_t wg_geometry_buffer_size(int P) {
    const size_t p = (size_t)(P > 0 ? P : 0);
    const bool lists = p >= (size_t)options_snapshot().band_list_min_p;
    return required_bytes([&](char*& c) { wg::GeometryState::fromChunk(c, p, lists); });
}
size_t wg_image_buffer_size(int width, int height) {
    const size_t N = (size_t)(width > 0 ? width : 0) * (size_t)(height > 0 ? height : 0);
    const size_t tiles = (size_t)((width + wg::TILE_X - 1) / wg::TILE_X) * (size_t)((height + wg::TILE_Y - 1) / wg::TILE_Y);
    return required_bytes([&](char*& c) { wg::ImageState::fromChunk(c, N, tiles); });
}
size_t wg_image_accumulation_offset(int width, int height) {
    const size_t N = (size_t)(width > 0 ? width : 0) * (size_t)(height > 0 ? height : 0);
    return (((N ? N : 1) * sizeof(float)) + wg::ALIGN - 1) & ~(wg::ALIGN - 1);  // carve(): the next ALIGN-aligned address
}
size_t wg_binning_buffer_size(int R) {  // upper bound over both binning paths
    return required_bytes([&](char*& c) { wg::BinningState::fromChunk(c, (size_t)(R > 0 ? R : 0), true); });
}

// sh_second (wg_forward_args::sh_second): the tone kernels run even when a set has no tone of its own (NULL = identity: no affine, clamps at
// +infinity -- min(x, inf) = x, x * 1 + 0 = x)
static wg::ShTone device_tone(const wg_sh_tone* t, const wg_sh_tone* t2 = nullptr, bool sh_second = false) {
    wg::ShTone d;
    const float inf = std::numeric_limits<float>::infinity();
    if (t != nullptr) {
        d.enabled = 1;
        d.mul = t->mul; d.offset = t->offset;
        d.pre_clamp = t->pre_clamp_max; d.post_clamp = t->post_clamp_max;
        d.dL_dmul = t->dL_dmul; d.dL_doffset = t->dL_doffset;
    } else if (sh_second) {
        d.enabled = 1;
        d.pre_clamp = d.post_clamp = inf;
    }
    if (sh_second) {
        d.second = 1;
        d.pre_clamp2 = d.post_clamp2 = inf;
        if (t2 != nullptr) {
            d.mul2 = t2->mul; d.offset2 = t2->offset;
            d.pre_clamp2 = t2->pre_clamp_max; d.post_clamp2 = t2->post_clamp_max;
            d.dL_dmul2 = t2->dL_dmul; d.dL_doffset2 = t2->dL_doffset;
        }
    }
    return d;
}

// ---- the frame's exact_compositing, remembered per image buffer: a backward or recolouring call that carries another value
// than the forward call that made the image state would differentiate / recomposite decisions the stored per-pixel state does not hold.
// Host-side only (no device traffic): the last 64 frames, keyed by the 256-byte-aligned image-state address; an address handed out
// again by the caller's allocator is simply overwritten by the next forward call that gets it.
// Scratch of the deterministic backward: plain hipMalloc blocks of the library's own, LEASED to a call and kept between calls until
// wg_set_option("release_scratch", 1).  A call takes a block that no other call is holding (one it used before on the same stream when there
// is one: stream order then makes the reuse safe by itself), grows it when it is too small, and hands it back behind its last launch together
// with an event recorded on its stream; a later call on ANOTHER stream makes its stream wait for that event first.  Two calls in flight at
// once -- other streams, other host threads, or two threads feeding one stream -- therefore never share a block.
// Why not the runtime's stream-ordered allocator (hipMallocAsync / hipMallocFromPoolAsync), which does the same on paper: on ROCm 7.2 /
// MI355X kernels writing memory that came from it LOSE STORES.  Round 5 met it twice: (1) repeated deterministic backward passes over one
// frame with a hipStreamSynchronize between them (what retain_graph=True does) on the device's default pool: the second pass missed the sums
// of ~6 % of the Gaussians -- gone when the pool was told to keep its memory, so a pool of the library's own with release threshold =
// everything replaced it; (2) that pool with OTHER host threads calling the runtime at the same time (three callers on their own streams,
// tests/native/c_abi_driver.cpp): 10 of 14 runs had a deterministic call with thousands of Gaussians' sums missing or short, 0 of 10 with
// hipMalloc blocks, 0 of 7 with nobody in the deterministic mode or one thread alone (scripts/r5/r5_concurrent_diag.sh,
// profiles/r5/concurrent_callers_diag.log; EXPERIMENTS.md R5.10, R5.12).  Nothing else in the library uses that allocator.
struct ScratchBlock { int dev; void* p; size_t cap; hipStream_t last_stream; hipEvent_t done; bool recorded; bool busy; };
std::mutex g_pool_mu;
std::vector<ScratchBlock*> g_scratch;   // (blocks are never moved: a lease is a pointer)
hipError_t det_scratch_alloc(ScratchBlock** lease, size_t bytes, hipStream_t stream) {
    *lease = nullptr;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    ScratchBlock* b = nullptr;
    {
        std::lock_guard<std::mutex> l(g_pool_mu);
        // large enough + same stream; large enough; too small but this stream's own (or never used): it will be grown.  A too-small block
        // ANOTHER stream used last is left alone -- growing it means hipFree, which waits for the whole device, that stream's work included
        // (ADVICE r5): a new block is made instead.
        for (int pass = 0; pass < 3 && !b; pass++)
            for (ScratchBlock* x : g_scratch)
                if (!x->busy && x->dev == dev && (pass == 2 ? (x->last_stream == stream || x->p == nullptr) : x->cap >= bytes) &&
                    (pass != 0 || x->last_stream == stream)) { b = x; break; }
        if (!b) {
            b = new ScratchBlock{dev, nullptr, 0, nullptr, nullptr, false, false};
            e = hipEventCreateWithFlags(&b->done, hipEventDisableTiming);
            if (e != hipSuccess) { delete b; return e; }
            g_scratch.push_back(b);
        }
        b->busy = true;
    }
    // (outside the lock: hipFree / hipMalloc may wait for the device)
    if (b->cap < bytes) {
        if (b->p) e = hipFree(b->p);   // waits for everything in flight on the device: the block's last user included
        b->p = nullptr; b->cap = 0; b->recorded = false;
        // headroom: cameras whose instance counts vary would otherwise re-allocate (= stall the device) at every new maximum
        const size_t want = ((bytes + bytes / 4) + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        if (e == hipSuccess) {
            e = hipMalloc(&b->p, want);
            if (e == hipSuccess) b->cap = want;
            else { (void)hipGetLastError(); e = hipMalloc(&b->p, bytes); if (e == hipSuccess) b->cap = bytes; }   // (no room for the headroom: the exact size)
        }
    } else if (b->recorded && b->last_stream != stream) {
        e = hipStreamWaitEvent(stream, b->done, 0);
    }
    if (e != hipSuccess) {
        std::lock_guard<std::mutex> l(g_pool_mu);
        b->busy = false;
        return e;
    }
    *lease = b;
    return hipSuccess;
}
// behind the call's last launch on `stream`
hipError_t det_scratch_free(ScratchBlock* b, hipStream_t stream) {
    const hipError_t e = hipEventRecord(b->done, stream);
    void* drop = nullptr;
    {
        std::lock_guard<std::mutex> l(g_pool_mu);
        b->last_stream = stream;
        b->recorded = e == hipSuccess;
        if (e != hipSuccess && b->p) {   // no event to order the next user behind this call: do not hand the memory out again
            drop = b->p;
            b->p = nullptr; b->cap = 0;
        }
        b->busy = false;
    }
    if (drop) (void)hipFree(drop);   // (outside the lock: hipFree waits for the device, other callers must not wait behind it)
    return e;
}
// wg_set_option("release_scratch", 1): the blocks no call is holding go back to the system (hipFree waits for the device first)
int det_scratch_release() {
    std::vector<std::pair<int, void*>> drop;   // detached under the lock, freed outside it (hipFree waits for the device)
    {
        std::lock_guard<std::mutex> l(g_pool_mu);
        for (ScratchBlock* b : g_scratch)
            if (!b->busy && b->p) {
                drop.push_back({b->dev, b->p});
                b->p = nullptr; b->cap = 0; b->recorded = false;
            }
    }
    int rc = WG_OK;
    for (auto& d : drop) {
        int cur = 0;
        const bool sw = hipGetDevice(&cur) == hipSuccess && cur != d.first && hipSetDevice(d.first) == hipSuccess;
        if (hipFree(d.second) != hipSuccess) rc = WG_ERR_HIP;
        if (sw) (void)hipSetDevice(cur);
    }
    return rc;
}

// ---- the forward render kernel's launch-order history (binning.hip: forward_order_kernel) -------------------------------------------
// One table per device, plain hipMalloc memory of the library's own, allocated at the first forward call that wants it (never inside a
// stream capture; a call that cannot have it runs in image order) and kept: [2 * slots] tag words, then slots rows of ORDER_STRIDE costs.
// Read and written by kernels only; which row a camera takes is decided on the device.  Never freed or moved while the process lives --
// captured graphs hold its address -- except by wg_set_option("release_scratch", 1), whose caller vouches that no graph replays it.
constexpr uint32_t ORDER_STRIDE = 9216;   // tiles per row: 1080p (8160) fits one row, a 4K frame (32 400) takes four
struct OrderTable { int dev; uint32_t* mem; uint32_t slots; };
std::mutex g_order_mu;
std::vector<OrderTable> g_order_tables;
// nullptr: no table (switched off, frame too large, capturing with none allocated yet, or out of memory -- the forward pass runs without)
uint32_t* order_table_for(const wg::Options& opt, int tiles, hipStream_t stream, uint32_t* slots_out) {
    *slots_out = 0;
    if (!opt.forward_order || opt.forward_order_slots <= 0 || tiles < 64) return nullptr;
    const uint32_t per = ((uint32_t)tiles + ORDER_STRIDE - 1u) / ORDER_STRIDE;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> l(g_order_mu);
    for (const OrderTable& t : g_order_tables)
        if (t.dev == dev) {
            if (t.slots < per) return nullptr;
            *slots_out = t.slots;
            return t.mem;
        }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    const uint32_t slots = (uint32_t)opt.forward_order_slots;
    if (slots < per) return nullptr;
    const size_t bytes = (2 * (size_t)slots + (size_t)slots * ORDER_STRIDE) * sizeof(uint32_t);
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, 2 * (size_t)slots * sizeof(uint32_t)) != hipSuccess) {   // (only the tags must start empty)
        (void)hipGetLastError();
        if (p) (void)hipFree(p);
        g_order_tables.push_back({dev, nullptr, 0});   // do not try again on every call
        return nullptr;
    }
    g_order_tables.push_back({dev, static_cast<uint32_t*>(p), slots});
    *slots_out = slots;
    return static_cast<uint32_t*>(p);
}
void order_tables_release() {
    std::lock_guard<std::mutex> l(g_order_mu);
    for (OrderTable& t : g_order_tables)
        if (t.mem) {
            int cur = 0;
            const bool sw = hipGetDevice(&cur) == hipSuccess && cur != t.dev && hipSetDevice(t.dev) == hipSuccess;
            (void)hipFree(t.mem);
            if (sw) (void)hipSetDevice(cur);
        }
    g_order_tables.clear();
}

struct FrameMode { const void* image; int exact; };
std::mutex g_mode_mu;
FrameMode g_modes[64];
unsigned g_mode_head = 0;
const void* image_key(const char* image_buffer) {
    return reinterpret_cast<const void*>((reinterpret_cast<uintptr_t>(image_buffer) + wg::ALIGN - 1) & ~(uintptr_t)(wg::ALIGN - 1));
}
void remember_mode(const char* image_buffer, int exact) {
    const void* k = image_key(image_buffer);
    std::lock_guard<std::mutex> l(g_mode_mu);
    for (auto& m : g_modes)
        if (m.image == k) { m.exact = exact; return; }
    g_modes[g_mode_head++ % 64] = {k, exact};
}
// -1: not remembered (older than 64 frames: the caller is trusted)
int remembered_mode(const char* image_buffer) {
    const void* k = image_key(image_buffer);
    std::lock_guard<std::mutex> l(g_mode_mu);
    for (auto& m : g_modes)
        if (m.image == k) return m.exact;
    return -1;
}

static int forward_impl(const wg_forward_args& a);
static int recolor_impl(const wg_forward_args& a);
static int backward_impl(const wg_backward_args& a);

int wg_rasterize_forward(wg_alloc_fn geometry_alloc, void* geometry_user, wg_alloc_fn binning_alloc, void* binning_user,
                         wg_alloc_fn image_alloc, void* image_user, int P, int D, int M, const float* background, int width,
                         int height, const float* means3D, const float* shs, const float* colors_precomp,
                         const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                         const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                         float tan_fovx, float tan_fovy, float kernel_size, const float* subpixel_offset, int prefiltered,
                         float* out_color, int* radii, int debug, void* stream_) {
    wg_forward_args a{};
    a.struct_size = sizeof(a);
    a.geometry_alloc = geometry_alloc; a.geometry_user = geometry_user;
    a.binning_alloc = binning_alloc; a.binning_user = binning_user;
    a.image_alloc = image_alloc; a.image_user = image_user;
    a.P = P; a.D = D; a.M = M; a.width = width; a.height = height; a.prefiltered = prefiltered; a.debug = debug;
    a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.kernel_size = kernel_size;
    a.background = background; a.means3D = means3D; a.shs = shs; a.colors_precomp = colors_precomp; a.opacities = opacities;
    a.scales = scales; a.rotations = rotations; a.cov3D_precomp = cov3D_precomp;
    a.viewmatrix = viewmatrix; a.projmatrix = projmatrix; a.cam_pos = cam_pos; a.subpixel_offset = subpixel_offset;
    a.out_color = out_color; a.radii = radii; a.stream = stream_;
    return forward_impl(a);
}

// struct_size: a caller built against an OLDER header hands over a shorter struct (fields are only ever appended): accepted from the
// first published layout on, the missing tail reads as zero / NULL = absent.  A larger one (a newer header) is refused: its tail may ask
// for something this library does not know.
constexpr size_t FORWARD_ARGS_V05 = 288, BACKWARD_ARGS_V05 = 312;   // sizeof of the first published layouts (version 0.5, LP64)
static_assert(sizeof(wg_forward_args) >= FORWARD_ARGS_V05 && sizeof(wg_backward_args) >= BACKWARD_ARGS_V05, "fields are appended, never removed");
int wg_rasterize_forward_ex(const wg_forward_args* args) {
    if (args == nullptr || args->struct_size < FORWARD_ARGS_V05 || args->struct_size > sizeof(wg_forward_args)) return WG_ERR_INVALID_ARGUMENT;
    wg_forward_args a{};
    std::memcpy(&a, args, args->struct_size);
    a.struct_size = sizeof(a);
    return a.recolor != nullptr ? recolor_impl(a) : forward_impl(a);
}

int wg_forward_status(char* image_buffer, int width, int height, int* num_rendered, int* fits, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!image_buffer || width <= 0 || height <= 0) return WG_ERR_INVALID_ARGUMENT;
    const int gx = (width + wg::TILE_X - 1) / wg::TILE_X, gy = (height + wg::TILE_Y - 1) / wg::TILE_Y;
    wg::ImageState img = wg::ImageState::fromChunk(image_buffer, (size_t)width * height, (size_t)gx * gy);
    wg::BinStats st{};
    hipError_t e = hipMemcpyAsync(&st, img.stats, sizeof(st), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail(e, "forward status readback");
    if (num_rendered) *num_rendered = st.num_rendered > 0x7fffffffu ? 0x7fffffff : (int)st.num_rendered;
    if (fits) *fits = st.spec_fail == 0u ? 1 : 0;
    return WG_OK;
}

static int forward_impl(const wg_forward_args& a) {
    // (the body below reads the arguments under the reference interface's names)
    const wg_alloc_fn geometry_alloc = a.geometry_alloc, binning_alloc = a.binning_alloc, image_alloc = a.image_alloc;
    void* const geometry_user = a.geometry_user; void* const binning_user = a.binning_user; void* const image_user = a.image_user;
    const int P = a.P, D = a.D, M = a.M, width = a.width, height = a.height, prefiltered = a.prefiltered;
    const float scale_modifier = a.scale_modifier, tan_fovx = a.tan_fovx, tan_fovy = a.tan_fovy, kernel_size = a.kernel_size;
    const float *background = a.background, *means3D = a.means3D, *shs = a.shs, *colors_precomp = a.colors_precomp, *opacities = a.opacities,
                *scales = a.scales, *rotations = a.rotations, *cov3D_precomp = a.cov3D_precomp, *viewmatrix = a.viewmatrix,
                *projmatrix = a.projmatrix, *cam_pos = a.cam_pos, *subpixel_offset = a.subpixel_offset;
    float* const out_color = a.out_color;
    int* const radii = a.radii;
    void* const stream_ = a.stream;
    const wg_sh_tone *tone = a.tone, *tone2 = a.tone2;
    const wg_second_image* second = a.second;
    const wg_raw_gaussians* raw = a.raw;
    const bool sh_second = a.sh_second != 0;
    const int fixed_capacity = a.binning_capacity;
    if (fixed_capacity < 0 || (fixed_capacity > 0 && a.debug)) return WG_ERR_INVALID_ARGUMENT;
    const int debug = fixed_capacity > 0 ? 0 : a.debug;
    if (a.recolor != nullptr) return WG_ERR_INVALID_ARGUMENT;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    wg::Options opt = options_snapshot();
    {   // the result-affecting switches are the CALL's (wg_call_options), never the process's
        const wg_call_options dflt = WG_CALL_OPTIONS_DEFAULT;
        const wg_call_options& co = a.options ? *a.options : dflt;
        opt.exact_compositing = co.exact_compositing != 0; opt.deterministic_backward = co.deterministic_backward != 0; opt.grad_record = co.grad_record != 0;
    }
    // two colour sets over one walk: a second set of precomputed colours (wg_second_image), or the SAME SH coefficients through a
    // second tone (sh_second; wg_forward_args::sh_second).  The second image is required either way.
    float* out_color2 = nullptr;
    if (second != nullptr) {   // (the second image is written whatever P is: the background alone when there is nothing to composite)
        if (!second->out_color2) return WG_ERR_INVALID_ARGUMENT;
        if (P > 0 && !sh_second && (!second->colors_precomp2 || shs != nullptr || !colors_precomp)) return WG_ERR_INVALID_ARGUMENT;
        if (P > 0 && sh_second && (second->colors_precomp2 || shs == nullptr || colors_precomp)) return WG_ERR_INVALID_ARGUMENT;
        out_color2 = second->out_color2;
    } else if (sh_second) {
        return WG_ERR_INVALID_ARGUMENT;
    }
    // get_gaussians() inside the preprocess kernel (wg_raw_gaussians): needs the scale / rotation pair it acts on
    if (raw != nullptr && P > 0 && (!raw->filter_3D || !scales || !rotations || cov3D_precomp)) return WG_ERR_INVALID_ARGUMENT;
    const bool fixed = fixed_capacity > 0;  // no host rendezvous at all: the caller's capacity, the superset (lazy) flow, a device-side verdict
    {
        const int settled = settle_deferred();
        if (settled != WG_OK) return settled;
    }
    if (tone != nullptr && shs == nullptr && P > 0) return WG_ERR_INVALID_ARGUMENT;  // the tone acts on SH coefficients
    if (!geometry_alloc || !binning_alloc || !image_alloc) return WG_ERR_INVALID_ARGUMENT;
    if (P < 0 || width <= 0 || height <= 0 || D < 0 || D > 3) return WG_ERR_INVALID_ARGUMENT;
    if (!background || !out_color || !viewmatrix || !projmatrix) return WG_ERR_INVALID_ARGUMENT;  // subpixel_offset may be null (= zeros)
    if (P > 0) {
        if (!means3D || !opacities) return WG_ERR_INVALID_ARGUMENT;
        // exactly one colour source / one covariance source (GaussianRasterizer.forward, __init__.py:212-216)
        if ((shs == nullptr) == (colors_precomp == nullptr)) return WG_ERR_INVALID_ARGUMENT;
        if (cov3D_precomp == nullptr && (!scales || !rotations)) return WG_ERR_INVALID_ARGUMENT;
        if (shs != nullptr && (!cam_pos || M < (D + 1) * (D + 1))) return WG_ERR_INVALID_ARGUMENT;
    }

    const int gx = (width + wg::TILE_X - 1) / wg::TILE_X, gy = (height + wg::TILE_Y - 1) / wg::TILE_Y;
    const int tiles = gx * gy;

    if (fixed && (tiles > wg::BIN_MAX_TILES || opt.force_global_sort || !opt.lazy.enabled)) return WG_ERR_INVALID_ARGUMENT;  // LDS binning + lazy sort only
    const bool band_lists = (size_t)P >= (size_t)opt.band_list_min_p;  // size and carving from the same snapshot
    char* geom_chunk = geometry_alloc(required_bytes([&](char*& c) { wg::GeometryState::fromChunk(c, (size_t)P, band_lists); }), geometry_user);
    char* img_chunk = image_alloc(wg_image_buffer_size(width, height), image_user);
    if (!geom_chunk || !img_chunk) return WG_ERR_ALLOC;
    remember_mode(img_chunk, opt.exact_compositing ? 1 : 0);
    wg::GeometryState geom = wg::GeometryState::fromChunk(geom_chunk, (size_t)P, band_lists);
    wg::ImageState img = wg::ImageState::fromChunk(img_chunk, (size_t)width * height, (size_t)tiles);

    wg::FwdParams fp;
    fp.P = P; fp.D = D; fp.M = M; fp.W = width; fp.H = height; fp.gx = gx; fp.gy = gy;
    fp.means3D = means3D; fp.shs = shs; fp.colors_precomp = colors_precomp; fp.opacities = opacities;
    fp.colors_precomp2 = (out_color2 && P > 0 && !sh_second) ? second->colors_precomp2 : nullptr;
    if (raw != nullptr && P > 0) fp.filter_3D = raw->filter_3D;
    fp.nt_stream = opt.sh_stream == 1 || (opt.sh_stream < 0 && P <= opt.sh_stream_max_p);
    fp.scales = scales; fp.scale_modifier = scale_modifier; fp.rotations = rotations; fp.cov3D_precomp = cov3D_precomp;
    fp.viewmatrix = viewmatrix; fp.projmatrix = projmatrix; fp.cam_pos = cam_pos;
    fp.tan_fovx = tan_fovx; fp.tan_fovy = tan_fovy;
    fp.focal_y = height / (2.0f * tan_fovy);  // rasterizer_impl.cu:224-225
    fp.focal_x = width / (2.0f * tan_fovx);
    fp.kernel_size = kernel_size; fp.prefiltered = prefiltered;

    int num_rendered = 0;
    uint32_t max_tile_count = 0;
    bool huge_frame = false;
    Mailbox* mbox = nullptr;
    uint32_t order_slots = 0;
    uint32_t* const order_table = (P > 0 && !debug) ? order_table_for(opt, tiles, stream, &order_slots) : nullptr;
    // Near / far split of dense frames (binning.hip): attempted from band_list_min_p Gaussians on (or whenever forced), on the LDS
    // binning path with the lazy sort available; whether it is ACTIVE for this frame is decided on the device (dense enough?) and
    // comes back with the instance count.
    // (measured at 10 M Gaussians / 4K, where pixels stop ~220 instances deep: 2050 near instances per tile 300 fps, 1400 324 fps,
    // 1000 359 fps, none of them sending a tile to the far phase; 1.5 x the front target keeps a margin for deeper walks)
    //  With the difference-grid counting: 1230 / 1000 / 900 per tile 365 / 404 / 427 fps there, 1021 / 1103 / 1145 train iter/s on
    //  the dense x3 frame: a near bag that fits the 1024-key network is sorted without a selection pass.)
    const uint32_t near_default = (opt.lazy.target * 11u) / 10u;
    const bool near_adaptive = opt.near_per_tile <= 0 && opt.near_adapt != 0 && opt.near_split < 0;
    if (near_adaptive && !t_near.same_scene(P)) {
        t_near = NearAdapt();
        t_near.P = P;
        if (t_mailbox.host) t_mailbox.host->far_report = 0ull;   // (a report of the other scene's last frame)
    }
    if (near_adaptive && (t_near.cur == 0u || t_near.cur > near_default)) t_near.cur = near_default;
    // Frames whose pixels do not saturate (low opacities: after an opacity reset, early in training) walk their whole lists: every
    // band then asks for its far instances and the split only adds a second, slower scatter.  The last split frame's request mask
    // arrives through the mailbox: the number of tiles that asked.  A few deep tiles are what the far phase is for (its cost is a
    // walk of the flagged bands' far Gaussians); after a frame in which more than 2 % of the tiles asked, the split is not
    // attempted for the next 64 frames of this thread (automatic mode only).
    if (opt.near_split < 0) {
        Mailbox& mb = t_mailbox;
        unsigned long long report = 0ull;
        if (mb.host) report = __atomic_exchange_n(&mb.host->far_report, 0ull, __ATOMIC_RELAXED);   // (take it: a report is acted on once)
        if (report != 0ull) {
            const uint32_t word = (uint32_t)report - 1u, far_tiles = word & 0xffffffu, far_bands = word >> 24, report_aim = (uint32_t)(report >> 32);
            // (a frame that failed at an aim the adaptation had LOWERED says the aim was too low, not that the scene does not saturate: the aim is
            //  lifted below, the split stays on -- with the back-off a single probing step cost 64 unsplit frames: 536 -> 447 fps over 600 frames)
            if ((uint64_t)far_tiles * 50u > (uint64_t)tiles && !(near_adaptive && report_aim != 0u && report_aim < near_default)) t_split_backoff = 64;
            if (near_adaptive && report_aim != 0u) {
                t_near.last_far = far_tiles;
                // ANY tile that asked is a failure of the aim its frame ran with.  (Rounds of this controller: "more than one tile in a thousand"
                // let thirty tiles in all eight bands pass as clean while every frame paid the far scatter of every band, 536 -> 447 fps over 600
                // frames; "at most two bands" still let the aim rest where a few tiles asked in EVERY frame, each paying a far phase: 2.35 instead of
                // 1.82 ms per frame for 2000 frames at 10 M Gaussians / 4K, profiles/r6/near_trace_*.)  The floor remembers the level, so there is
                // no saw-tooth: one or two slow frames per probing step, and a probing step only when the floor has decayed (every 2048 frames).
                if (far_tiles != 0u) {
                    // the floor goes an eighth above the aim THE FAILING FRAME ran with (the report carries it) -- a sixteenth when only a handful of
                    // tiles asked (the aim sits right at the deepest tiles' need): reports lag a frame or two behind the calls when the caller does not
                    // synchronise, and two frames issued at one failing aim used to lift the floor twice (386 failed -> 435 -> 490 where 435 sufficed)
                    const uint32_t step = ((uint64_t)far_tiles * 1000u > (uint64_t)tiles || far_bands > 2u) ? report_aim / 8u : report_aim / 16u;
                    t_near.floor = std::min(near_default, std::max(t_near.floor, report_aim + step + 1u));
                    t_near.cur = std::max(t_near.cur, t_near.floor);
                    t_near.clean = 0;
                } else if (report_aim == t_near.cur && ++t_near.clean >= 4u) {   // (four clean frames AT the current aim)
                    t_near.clean = 0;
                    t_near.cur = std::max(std::max(t_near.floor, 192u), (t_near.cur * 9u) / 10u);
                }
                if (++t_near.age >= 2048u) { t_near.age = 0; t_near.floor = (t_near.floor * 19u) / 20u; }
            }
        }
    }
    const uint32_t near_per_tile = opt.near_per_tile > 0 ? (uint32_t)opt.near_per_tile : (near_adaptive ? t_near.cur : near_default);
    const bool backoff = opt.near_split < 0 && t_split_backoff > 0;
    if (backoff) t_split_backoff--;
    // Automatic mode attempts it for large scenes and whenever this host thread's previous frame was dense (a performance hint only:
    // the threshold pass costs ~15 us, the results are the same either way).
    const bool try_split = P > 0 && tiles <= wg::BIN_MAX_TILES && opt.lazy.enabled && !opt.force_global_sort && opt.near_split != 0 &&
                           wg::GeometryState::band_lists_possible((size_t)P) &&
                           (opt.near_split == 1 || P >= opt.band_list_min_p || t_last_instances_per_tile >= wg::SPLIT_DENSE_AVG) && !backoff;
    bool split_active = false;
    // Frames that attempt the split with plain SH colours colour their Gaussians LAZILY: the per-Gaussian kernel leaves the colour out (and the
    // 192-byte SH block unread), the near Gaussians are coloured once the threshold is known, the far ones only if a tile asks for its far
    // instances (preprocess.hip: GEOM_ONLY, sh_colour_kernel).  Same colours, bit for bit, for every Gaussian the walk can reach.
    const bool lazy_colour = try_split && opt.lazy_colour != 0 && P >= opt.lazy_colour_min_p && shs != nullptr && colors_precomp == nullptr && tone == nullptr && tone2 == nullptr &&
                             !sh_second && out_color2 == nullptr && raw == nullptr;   // (plain SH colours of activated parameters: the combinations the suite holds)

    // ---- what depends on the instance count, as functions of it (used by the speculative and by the classic flow alike) ----
    // Longest per-tile list decides the binning path: full register sort of every tile, lazy front sort when lists are long,
    // global radix sort (the reference's scheme) when forced, when the frame is too large for the LDS histogram, or when a
    // list exceeds the register sort and the lazy sort is switched off.
    // (an active split implies the lazy path: its buckets only hold the near instances at first)
    auto lazy_for = [&](bool split_on, uint32_t longest) {
        return split_on || (opt.lazy.enabled && !opt.force_global_sort && !huge_frame && longest > opt.lazy.min_len + opt.lazy.min_len / 4);
    };
    // lazy sort: bucket entries carry a coarse depth code above the id for the front extraction, as wide as the ids allow
    // (2^20 Gaussians or fewer: 12 bits; up to 2^24: 8 bits; more: none)
    auto code_bits_for = [&](bool lazy) {
        int code_bits = 0;
        if (lazy && opt.depth_codes && P <= (1 << 24)) {
            int id_bits = 20;
            while ((1 << id_bits) < P) id_bits++;
            code_bits = 32 - id_bits;
            if (opt.depth_codes >= 8 && opt.depth_codes <= code_bits) code_bits = opt.depth_codes;  // a narrower code than the ids allow
        }
        return code_bits;
    };
    // Everything behind the count on the LDS binning path: scatter -> per-tile sort (full, or lazy front) -> render -> [fix-up ->
    // far scatter -> fix-up].  R sizes the scatter's staging passes only; `longest` picks the sort network; `far` = the split may be
    // active (its two extra launches return at once when it is not, or when no tile asked); guard = the frame's BinStats when the
    // kernels are enqueued BEFORE the count is known (speculation), nullptr otherwise.
    auto enqueue_tail = [&](const wg::BinningState& bin, uint32_t R, uint32_t longest, bool lazy, bool far, const wg::BinStats* guard) -> int {
        const int code_bits = code_bits_for(lazy);
        uint32_t emit = R;  // what the scatter will emit, for the sizing of its staging passes: everything, or about near_per_tile per tile
        if (far) emit = (uint32_t)std::min<uint64_t>(emit, (uint64_t)near_per_tile * (uint64_t)tiles * 5u / 4u);
        WG_STAGE(WG_STAGE_DUPLICATE_KEYS, wg::launch_tile_scatter(P, geom, img, bin, gx, tiles, emit, code_bits, opt.staged_scatter, opt.staged_cap, try_split, guard, stream), "tile_scatter");
        if (lazy) WG_STAGE(WG_STAGE_SORT, wg::launch_tile_sort_lazy(img, bin, geom, tiles, code_bits, opt.lazy, try_split, guard, stream), "tile_sort_lazy");
        else WG_STAGE(WG_STAGE_SORT, wg::launch_tile_sort(img, bin, geom, tiles, longest, guard, stream), "tile_sort");
        WG_STAGE(WG_STAGE_RENDER_FORWARD,
                 wg::launch_render_forward(width, height, gx, gy, img, bin, geom, subpixel_offset, background, out_color, out_color2, lazy, opt.exact_compositing != 0, guard, order_table, order_slots, ORDER_STRIDE, stream),
                 "render_forward");
        if (lazy) {
            WG_STAGE(WG_STAGE_RENDER_FIXUP,
                     wg::launch_render_fixup(code_bits, width, height, gx, gy, img, bin, geom, subpixel_offset, background, out_color, out_color2, opt.lazy, try_split, 0, opt.exact_compositing != 0, (wg::HostMailbox*)nullptr, guard, stream),
                     "render_fixup");
            if (far) {  // all return at once unless some tile ran out of near instances with pixels still accumulating
                if (lazy_colour) WG_STAGE(WG_STAGE_PREPROCESS, wg::launch_sh_colour(fp, geom, img.split, true, stream), "sh_colour_far");
                WG_STAGE(WG_STAGE_DUPLICATE_KEYS, wg::launch_tile_scatter_far(P, geom, img, bin, gx, tiles, code_bits, guard, stream), "tile_scatter_far");
                WG_STAGE(WG_STAGE_RENDER_FIXUP,
                         wg::launch_render_fixup(code_bits, width, height, gx, gy, img, bin, geom, subpixel_offset, background, out_color, out_color2, opt.lazy, true, 1, opt.exact_compositing != 0, mbox ? mbox->dev : (wg::HostMailbox*)nullptr, guard, stream),
                         "render_fixup_far");
            }
        }
        return WG_OK;
    };

    bool rendered = false;  // the render kernels of this frame are already in the stream (a speculation that held)
    if (P > 0) {
        if (!order_table) {   // (what wg_view_image's order_key says then: no row)
            hipError_t e0 = hipMemsetAsync(img.order_key, 0xff, sizeof(uint32_t), stream);
            if (e0 != hipSuccess) return hip_fail(e0, "order_key memset");
        }
        wg::FwdOrderArgs fo;   // the forward render kernel's launch order: eight workgroups riding along in the tile scan's launch
        if (order_table) {
            fo.viewmatrix = viewmatrix; fo.projmatrix = projmatrix; fo.W = width; fo.H = height; fo.table = order_table; fo.slots = order_slots;
            fo.stride = ORDER_STRIDE; fo.order = img.order_fwd; fo.key_out = img.order_key; fo.period = (uint32_t)std::max(opt.order_period, 0);
            if (tiles > wg::BIN_MAX_TILES || opt.fused_scan) {   // no stand-alone tile scan on these paths: a launch of its own, first in the stream
                WG_STAGE(WG_STAGE_TILE_RANGES, wg::launch_forward_order(fo, tiles, stream), "forward_order");
                fo.table = nullptr;
            }
        }
        WG_STAGE(WG_STAGE_PREPROCESS, wg::launch_preprocess(fp, device_tone(tone, tone2, sh_second), geom, radii, lazy_colour, stream), "preprocess");
        wg::SpecLimits spec;   // all zero: the classic flow
        char* spec_chunk = nullptr;
        bool spec_lazy = false;
        if (tiles <= wg::BIN_MAX_TILES) {
            if (try_split)
                WG_STAGE(WG_STAGE_SCAN, wg::launch_split_threshold(P, geom, img, tiles, opt.near_split == 1, near_per_tile, stream), "split_threshold");
            if (lazy_colour) WG_STAGE(WG_STAGE_PREPROCESS, wg::launch_sh_colour(fp, geom, img.split, false, stream), "sh_colour_near");
            const bool box = opt.box_count == 1 || (opt.box_count < 0 && (P >= opt.band_list_min_p || t_last_instances_per_tile >= 1500u));
            WG_STAGE(WG_STAGE_SCAN, wg::launch_tile_count(P, geom, img, gx, tiles, try_split, box, opt.fused_scan != 0, stream), "tile_count");
            mbox = (debug || !opt.use_mailbox || fixed) ? nullptr : get_mailbox();
            if (mbox) mbox->seq += 1;
            if (fixed) {  // the caller's capacity; the lazy flow covers lists of any length
                spec.capacity = (uint32_t)fixed_capacity;
                spec.max_list = 0xffffffffu;
                spec_lazy = true;
                spec_chunk = binning_alloc(required_bytes([&](char*& c) { wg::BinningState::fromChunk(c, (size_t)spec.capacity, false); }), binning_user);
                if (!spec_chunk) return WG_ERR_ALLOC;
            }
            // ---- speculative forward (option "speculative_forward", default on) ----
            // The reference's forward pass stops in its middle for the instance count (rasterizer_impl.cu:284: it sizes the binning
            // buffer), the GPU idles while the host then launches the rest.  Frames of one training run resemble one another, so the
            // count of THIS frame is predicted from this thread's recent frames of the same shape: the binning buffer is allocated
            // with a margin, everything behind the count is enqueued at once -- each kernel guarded by the verdict tile_scan leaves
            // in BinStats::spec_fail -- and the host looks at the mailbox only after its last launch, by which time the count has
            // usually long arrived.  A frame that does not fit (more instances than the buffer holds, a list longer than the launched
            // sort network covers) runs none of the guarded kernels; the host then re-issues the tail with the real numbers, i.e.
            // falls back to the classic flow for that frame.  Results are the classic flow's, bit for bit.
            if (!fixed && mbox && opt.speculative > 0 && !opt.force_global_sort && t_spec.usable(P, width, height)) {
                const uint64_t cap64 = (uint64_t)t_spec.max_rendered() * (100u + (uint32_t)opt.spec_margin_pct) / 100u + 4096u;
                const uint32_t longest = t_spec.max_longest() + t_spec.max_longest() / 8u + 16u;
                // the split's decision is taken on the device: when it is attempted the (superset) lazy flow is enqueued
                spec_lazy = lazy_for(try_split, longest);
                const uint32_t cover = spec_lazy ? 0xffffffffu : (longest <= 1024u ? 1024u : longest <= 2048u ? 2048u : longest <= 4096u ? 4096u : wg::TILE_SORT_MAX);
                if (cap64 < 0x7fffffffull && (spec_lazy || longest <= wg::TILE_SORT_MAX)) {
                    spec.capacity = (uint32_t)cap64;
                    spec.max_list = cover;
                    spec_chunk = binning_alloc(required_bytes([&](char*& c) { wg::BinningState::fromChunk(c, (size_t)spec.capacity, false); }), binning_user);
                    if (!spec_chunk) return WG_ERR_ALLOC;
                }
            }
            WG_STAGE(WG_STAGE_SCAN, wg::launch_tile_scan(img, tiles, mbox ? mbox->dev : nullptr, mbox ? mbox->seq : 0, try_split, spec, opt.fused_scan != 0, fo, stream), "tile_scan");
            // "speculative_forward" = 2: do not even look at the verdict before returning -- the thread's next call does (settle_deferred)
            const bool deferred = !fixed && spec.capacity != 0u && opt.speculative == 2;
            if (spec.capacity != 0u) {
                wg::BinningState sbin = wg::BinningState::fromChunk(spec_chunk, (size_t)spec.capacity, false);
                const uint32_t r_hint = fixed ? (t_spec.usable(P, width, height) ? std::min(t_spec.last_rendered(), spec.capacity) : spec.capacity) : t_spec.last_rendered();
                const int st_ = enqueue_tail(sbin, r_hint, spec.max_list == 0xffffffffu ? 0u : spec.max_list, spec_lazy, try_split, img.stats);
                if (st_ != WG_OK) return st_;
                if (fixed || deferred) {
                    // A frame that does not fit leaves nothing rendered: make that impossible to miss (NaN image and accumulation) and
                    // safe to differentiate (no walked instance anywhere: the backward pass returns zeros); wg_forward_status tells.
                    WG_STAGE(WG_STAGE_RENDER_FORWARD, wg::launch_poison_unfit(img, width, height, tiles, out_color, out_color2, stream), "poison_unfit");
                    if (deferred) {
                        t_deferred.pending = true;
                        t_deferred.seq = mbox->seq;
                        t_deferred.P = P; t_deferred.W = width; t_deferred.H = height;
                        t_deferred.stream = stream;
                        post_ticket(img.final_T, mbox->host, mbox->seq, stream);   // (the state's first array: image_alloc's pointer, aligned)
                    }
                    return (int)spec.capacity;
                }
            }
        } else {
            huge_frame = true;  // tile histogram does not fit LDS: count through the per-Gaussian prefix sum instead
            {   // tile_scan, which fills the frame's BinStats, does not run on this path: wg_forward_status must not read a fresh buffer's bytes
                hipError_t e0 = hipMemsetAsync(img.stats, 0, sizeof(wg::BinStats), stream);
                if (e0 != hipSuccess) return hip_fail(e0, "stats memset");
            }
            WG_STAGE(WG_STAGE_SCAN, wg::run_scan(geom, P, stream), "inclusive_scan");
            WG_STAGE(WG_STAGE_SCAN, wg::launch_scan_overflow_check(geom, P, &img.stats->max_tile_count, stream), "scan_overflow_check");
        }
        // the one host rendezvous of the forward pass (rasterizer_impl.cu:284) -- behind the frame's last launch when speculating
        wg::BinStats st{};
        hipError_t e;
        bool have_stats = false;
        if (mbox) {
            // poll the mailbox (bounded: ~2 s, then fall back to a real synchronise so that a failed launch is reported)
            const auto t0 = std::chrono::steady_clock::now();
            bool waited = false;
            have_stats = read_mailbox(mbox, mbox->seq, st, &waited);
            t_wait.record(waited, std::chrono::steady_clock::now() - t0);
        }
        if (have_stats) {
            e = hipSuccess;
        } else if (!huge_frame) {
            e = hipMemcpyAsync(&st, img.stats, sizeof(st), hipMemcpyDeviceToHost, stream);
        } else {
            e = hipMemcpyAsync(&st.num_rendered, geom.point_offsets + (P - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&st.max_tile_count, &img.stats->max_tile_count, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        }
        if (e != hipSuccess) return hip_fail(e, "num_rendered readback");
        if (!have_stats) {
            e = hipStreamSynchronize(stream);
            if (e != hipSuccess) return hip_fail(e, "num_rendered readback sync");
        }
        if (huge_frame) {  // here max_tile_count carried the scan's overflow flag; the longest list itself is unknown on this path
            if (st.max_tile_count != 0u) return WG_ERR_OVERFLOW;
            st.max_tile_count = 0xffffffffu;
        }
        if (st.num_rendered > 0x7fffffffu) return WG_ERR_OVERFLOW;
        num_rendered = (int)st.num_rendered;
        max_tile_count = st.max_tile_count;
        split_active = try_split && !huge_frame && st.split_active != 0u;
        t_last_instances_per_tile = tiles > 0 ? st.num_rendered / (uint32_t)tiles : 0u;
        if (!huge_frame) t_spec.push(P, width, height, st.num_rendered, st.max_tile_count);
        if (spec.capacity != 0u) {
            if (st.spec_fail == 0u) rendered = true;
            t_wait.spec_frames += 1;
            t_wait.spec_misses += st.spec_fail != 0u ? 1 : 0;
        }
    } else {
        hipError_t e = hipMemsetAsync(img.ranges, 0, (size_t)tiles * sizeof(uint2), stream);
        if (e == hipSuccess) e = hipMemsetAsync(img.stats, 0, sizeof(wg::BinStats), stream);   // (no Gaussians: nothing rendered, fits)
        if (e != hipSuccess) return hip_fail(e, "ranges memset");
    }
    if (rendered) return num_rendered;

    const bool lazy = lazy_for(split_active, max_tile_count);
    const bool global_sort = opt.force_global_sort || huge_frame || (!lazy && max_tile_count > wg::TILE_SORT_MAX);
    size_t bin_bytes = required_bytes([&](char*& c) { wg::BinningState::fromChunk(c, (size_t)num_rendered, global_sort); });
    char* bin_chunk = binning_alloc(bin_bytes, binning_user);   // (a second call of this frame after a speculation that did not hold)
    if (!bin_chunk) return WG_ERR_ALLOC;
    wg::BinningState bin = wg::BinningState::fromChunk(bin_chunk, (size_t)num_rendered, global_sort);

    if (huge_frame && num_rendered == 0) {
        hipError_t e = hipMemsetAsync(img.ranges, 0, (size_t)tiles * sizeof(uint2), stream);
        if (e != hipSuccess) return hip_fail(e, "ranges memset");
    }
    if (num_rendered > 0 && !global_sort) {
        const int st_ = enqueue_tail(bin, (uint32_t)num_rendered, max_tile_count, lazy, split_active, nullptr);
        return st_ != WG_OK ? st_ : num_rendered;
    }
    if (num_rendered > 0) {
        if (!huge_frame) WG_STAGE(WG_STAGE_SCAN, wg::run_scan(geom, P, stream), "inclusive_scan");
        WG_STAGE(WG_STAGE_DUPLICATE_KEYS, wg::launch_duplicate_keys(P, geom, bin, gx, stream), "duplicate_keys");
        const int bit = (int)wg::higher_msb((uint32_t)tiles);  // rasterizer_impl.cu:303
        WG_STAGE(WG_STAGE_SORT, wg::run_sort(bin, num_rendered, 32 + bit, stream), "radix_sort_pairs");
        // tile_scan already produced ranges identical to identifyTileRanges on the sorted keys; only frames
        // too large for the LDS histogram need the key-boundary pass
        if (huge_frame) WG_STAGE(WG_STAGE_TILE_RANGES, wg::launch_tile_ranges(num_rendered, bin, img, tiles, stream), "tile_ranges");
    }
    WG_STAGE(WG_STAGE_RENDER_FORWARD,
             wg::launch_render_forward(width, height, gx, gy, img, bin, geom, subpixel_offset, background, out_color, out_color2, false, opt.exact_compositing != 0, nullptr, order_table, order_slots, ORDER_STRIDE, stream),
             "render_forward");
    return num_rendered;
}

// wg_recolor_parent (include/wg_rasterizer.h): other precomputed colours over a parent call's projection, binning and per-pixel stops
static int recolor_impl(const wg_forward_args& a) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(a.stream);
    const bool debug = false;
    const wg_recolor_parent& par = *a.recolor;
    const int P = a.P, R = par.R, width = a.width, height = a.height;
    if (a.tone || a.tone2 || a.sh_second || a.second || a.raw || a.binning_capacity != 0 || a.shs) return WG_ERR_INVALID_ARGUMENT;
    if (!a.geometry_alloc || !par.geom_buffer || !par.binning_buffer || !par.image_buffer) return WG_ERR_INVALID_ARGUMENT;
    if (P <= 0 || R < 0 || width <= 0 || height <= 0 || !a.background || !a.colors_precomp || !a.out_color) return WG_ERR_INVALID_ARGUMENT;
    const wg_call_options dflt = WG_CALL_OPTIONS_DEFAULT;
    const int exact = (a.options ? a.options->exact_compositing : dflt.exact_compositing) != 0 ? 1 : 0;
    const int parent_mode = remembered_mode(par.image_buffer);
    if (parent_mode >= 0 && parent_mode != exact) return WG_ERR_INVALID_ARGUMENT;   // the parent's per-pixel stops were taken in the other arithmetic
    const int gx = (width + wg::TILE_X - 1) / wg::TILE_X, gy = (height + wg::TILE_Y - 1) / wg::TILE_Y;
    char *pg = par.geom_buffer, *pb = par.binning_buffer, *pi = par.image_buffer;   // (fromChunk advances its argument)
    wg::GeometryState parent = wg::GeometryState::fromChunk(pg, (size_t)P, false);
    wg::BinningState bin = wg::BinningState::fromChunk(pb, (size_t)R, false);  // point_list only
    wg::ImageState img = wg::ImageState::fromChunk(pi, (size_t)width * height, (size_t)gx * gy);
    char* chunk = a.geometry_alloc(required_bytes([&](char*& c) { wg::GeometryState::fromChunk(c, (size_t)P, false); }), a.geometry_user);
    if (!chunk) return WG_ERR_ALLOC;
    wg::GeometryState geom = wg::GeometryState::fromChunk(chunk, (size_t)P, false);
    WG_STAGE(WG_STAGE_PREPROCESS, wg::launch_recolor(P, parent, geom, a.colors_precomp, a.radii, stream), "recolor");
    WG_STAGE(WG_STAGE_RENDER_FORWARD,
             wg::launch_render_forward_replay(width, height, gx, gy, img, bin, geom, a.subpixel_offset, a.background, a.out_color, exact != 0, stream),
             "render_forward_replay");
    return R;
}

int wg_rasterize_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                          const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                          const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                          const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, float kernel_size,
                          const float* subpixel_offset, const int* radii, char* geom_buffer, char* binning_buffer,
                          char* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                          float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                          float* dL_drot, int debug, void* stream_) {
    wg_backward_args a{};
    a.struct_size = sizeof(a);
    a.P = P; a.D = D; a.M = M; a.R = R; a.width = width; a.height = height; a.debug = debug;
    a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.kernel_size = kernel_size;
    a.background = background; a.means3D = means3D; a.shs = shs; a.colors_precomp = colors_precomp; a.scales = scales; a.rotations = rotations;
    a.cov3D_precomp = cov3D_precomp; a.viewmatrix = viewmatrix; a.projmatrix = projmatrix; a.campos = campos; a.subpixel_offset = subpixel_offset;
    a.radii = radii; a.geom_buffer = geom_buffer; a.binning_buffer = binning_buffer; a.image_buffer = image_buffer; a.dL_dpix = dL_dpix;
    a.dL_dmean2D = dL_dmean2D; a.dL_dconic = dL_dconic; a.dL_dopacity = dL_dopacity; a.dL_dcolor = dL_dcolor; a.dL_dmean3D = dL_dmean3D;
    a.dL_dcov3D = dL_dcov3D; a.dL_dsh = dL_dsh; a.dL_dscale = dL_dscale; a.dL_drot = dL_drot; a.stream = stream_;
    return backward_impl(a);
}

int wg_rasterize_backward_ex(const wg_backward_args* args) {
    if (args == nullptr || args->struct_size < BACKWARD_ARGS_V05 || args->struct_size > sizeof(wg_backward_args)) return WG_ERR_INVALID_ARGUMENT;
    wg_backward_args a{};
    std::memcpy(&a, args, args->struct_size);
    a.struct_size = sizeof(a);
    return backward_impl(a);
}

static int backward_impl(const wg_backward_args& a) {
    // (the body below reads the arguments under the reference interface's names)
    const int P = a.P, D = a.D, M = a.M, R = a.R, width = a.width, height = a.height, debug = a.debug;
    const float scale_modifier = a.scale_modifier, tan_fovx = a.tan_fovx, tan_fovy = a.tan_fovy, kernel_size = a.kernel_size;
    const float *background = a.background, *means3D = a.means3D, *shs = a.shs, *colors_precomp = a.colors_precomp, *scales = a.scales,
                *rotations = a.rotations, *cov3D_precomp = a.cov3D_precomp, *viewmatrix = a.viewmatrix, *projmatrix = a.projmatrix,
                *campos = a.campos, *subpixel_offset = a.subpixel_offset, *dL_dpix = a.dL_dpix;
    const int* radii = a.radii;
    char *geom_buffer = a.geom_buffer, *binning_buffer = a.binning_buffer, *image_buffer = a.image_buffer;   // (fromChunk advances its argument)
    float *const dL_dmean2D = a.dL_dmean2D, *const dL_dconic = a.dL_dconic, *const dL_dopacity = a.dL_dopacity, *const dL_dcolor = a.dL_dcolor,
          *const dL_dmean3D = a.dL_dmean3D, *const dL_dcov3D = a.dL_dcov3D, *const dL_dsh = a.dL_dsh, *const dL_dscale = a.dL_dscale,
          *const dL_drot = a.dL_drot;
    void* const stream_ = a.stream;
    const wg_sh_tone *tone = a.tone, *tone2 = a.tone2;
    const wg_second_image* second = a.second;
    const wg_raw_gaussians* raw = a.raw;
    const bool sh_second = a.sh_second != 0;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    wg::Options opt = options_snapshot();
    {   // the result-affecting switches are the CALL's (wg_call_options), never the process's
        const wg_call_options dflt = WG_CALL_OPTIONS_DEFAULT;
        const wg_call_options& co = a.options ? *a.options : dflt;
        opt.exact_compositing = co.exact_compositing != 0; opt.deterministic_backward = co.deterministic_backward != 0; opt.grad_record = co.grad_record != 0;
    }
    if (image_buffer != nullptr && P > 0) {   // the frame was composited with the other arithmetic: its stored decisions are not this call's
        const int fwd_mode = remembered_mode(image_buffer);
        if (fwd_mode >= 0 && fwd_mode != (opt.exact_compositing ? 1 : 0)) return WG_ERR_INVALID_ARGUMENT;
    }
    // two colour sets over one walk: the thirteen sums go to the gradient record (or, deterministic mode, to fourteen-float slots)
    const bool dual = second != nullptr && P > 0;
    // raw-parameter mode: the per-Gaussian kernel turns the gradients of the activated values into those of the raw parameters where it
    // WRITES them, i.e. with the gradient record
    if (raw != nullptr && P > 0 && (!raw->filter_3D || !raw->raw_opacities || !scales || !rotations || cov3D_precomp ||
                                    !(opt.grad_record || opt.deterministic_backward))) return WG_ERR_INVALID_ARGUMENT;
    if (sh_second && second == nullptr) return WG_ERR_INVALID_ARGUMENT;
    if (dual && (!second->dL_dpix2 || !(opt.grad_record || opt.deterministic_backward))) return WG_ERR_INVALID_ARGUMENT;
    if (dual && !sh_second && (!second->dL_dcolor2 || shs != nullptr)) return WG_ERR_INVALID_ARGUMENT;
    if (dual && sh_second) {   // the second set's dL/dRGB is an intermediate here (dL_dcolor2 optional, like dL_dcolor with SH colours)
        if (shs == nullptr) return WG_ERR_INVALID_ARGUMENT;
        if (tone2 != nullptr && ((tone2->mul != nullptr && tone2->dL_dmul == nullptr) || (tone2->offset != nullptr && tone2->dL_doffset == nullptr))) return WG_ERR_INVALID_ARGUMENT;
    }
    if (image_buffer != nullptr) {   // a deferred forward call's verdict, before anything is differentiated (found by its image buffer: the
        const void* key = reinterpret_cast<const void*>((reinterpret_cast<uintptr_t>(image_buffer) + wg::ALIGN - 1) & ~(uintptr_t)(wg::ALIGN - 1));
        const int verdict = check_ticket(key, stream);   // backward pass usually runs on torch's autograd thread, not the forward's)
        if (verdict != WG_OK) return verdict;
    }
    const int g_grad_record = opt.grad_record;
    if (tone != nullptr && P > 0) {
        if (shs == nullptr) return WG_ERR_INVALID_ARGUMENT;
        if ((tone->mul != nullptr && tone->dL_dmul == nullptr) || (tone->offset != nullptr && tone->dL_doffset == nullptr)) return WG_ERR_INVALID_ARGUMENT;
    }
    (void)colors_precomp;  // colours were copied into the splat records by the forward pass
    if (P < 0 || R < 0 || width <= 0 || height <= 0) return WG_ERR_INVALID_ARGUMENT;
    if (P == 0) return WG_OK;
    if (!geom_buffer || !binning_buffer || !image_buffer || !dL_dpix || !background) return WG_ERR_INVALID_ARGUMENT;
    if (!dL_dmean2D || !dL_dopacity || !dL_dmean3D || !dL_dcov3D) return WG_ERR_INVALID_ARGUMENT;
    // dL_dconic (always an intermediate) and, with SH colours, dL_dcolor (the gradient of the evaluated RGB, an intermediate there)
    // may be NULL with the gradient record: nobody reads them.  Without the record they are accumulation targets.
    if (!g_grad_record && !opt.deterministic_backward && (!dL_dconic || !dL_dcolor)) return WG_ERR_INVALID_ARGUMENT;
    if (!dL_dcolor && shs == nullptr) return WG_ERR_INVALID_ARGUMENT;
    if (shs != nullptr && (!dL_dsh || !campos)) return WG_ERR_INVALID_ARGUMENT;
    if (scales != nullptr && (!rotations || !dL_dscale || !dL_drot)) return WG_ERR_INVALID_ARGUMENT;
    if (scales == nullptr && cov3D_precomp == nullptr) return WG_ERR_INVALID_ARGUMENT;

    const int gx = (width + wg::TILE_X - 1) / wg::TILE_X, gy = (height + wg::TILE_Y - 1) / wg::TILE_Y;
    wg::GeometryState geom = wg::GeometryState::fromChunk(geom_buffer, (size_t)P, false);  // the band lists (carved last) are not needed
    wg::BinningState bin = wg::BinningState::fromChunk(binning_buffer, (size_t)R, false);  // only point_list is used
    wg::ImageState img = wg::ImageState::fromChunk(image_buffer, (size_t)width * height, (size_t)gx * gy);
    if (radii == nullptr) radii = geom.radii;  // rasterizer_impl.cu:381-384

    // grad_record (default): the per-tile pass accumulates into one 48-byte record per Gaussian (wg_common.h: GRAD_REC_FLOATS) inside the geometry buffer, cleared
    // here; the per-Gaussian kernel then WRITES the four arrays (they need no clearing by the caller).  Off: the arrays are the
    // accumulation targets and must arrive zeroed, as the reference demands of its caller (rasterize_points.cu:157-165).
    // deterministic_backward: per-instance slots (stream-ordered scratch of 40 B per tile instance + a flag byte each; only the flags
    // are cleared) + an ordered per-Gaussian sum instead of float atomics; needs the prefix sum of tiles_touched, which the LDS
    // binning path never made.
    // The record decision follows the OPTIONS alone (it is what the NULL checks above and the binding's allocation key on); only the
    // slot scratch, its scan and the ordered sum need instances to exist.  With nothing rendered the cleared record gives zeros.
    const bool det = opt.deterministic_backward != 0 && R > 0;
    const bool record = g_grad_record != 0 || opt.deterministic_backward != 0;
    // scratch of the deterministic mode (a leased block of the library's own: det_scratch_alloc), handed back on every way out of this function
    struct DetSlots {
        float* p = nullptr;
        ScratchBlock* lease = nullptr;
        hipStream_t s;
        explicit DetSlots(hipStream_t st) : s(st) {}
        ~DetSlots() { if (lease) (void)det_scratch_free(lease, s); }
        hipError_t release() { ScratchBlock* q = lease; lease = nullptr; p = nullptr; return q ? det_scratch_free(q, s) : hipSuccess; }
    } det_guard(stream);
    float*& det_slots = det_guard.p;
    unsigned char* det_flags = nullptr;
    if (det) {  // slots: 40 B per tile instance, never cleared; flags: 1 B per instance behind them, cleared
        StageScope scope_(WG_STAGE_RENDER_BACKWARD, stream);
        const size_t slot_bytes = (((size_t)R * (dual ? 14 : 10) * sizeof(float)) + 255) & ~(size_t)255;   // (the two-colour walk: thirteen sums, padded to fourteen)
        // a captured call would replay into a block that was only leased for the call's duration: refused (the atomic mode captures fine)
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) return WG_ERR_INVALID_ARGUMENT;
        hipError_t e = wg::run_scan(geom, P, stream);
        // (the flags: a whole number of 16-byte words, cleared by the launch that orders the tiles -- no memset of their own)
        if (e == hipSuccess) e = det_scratch_alloc(&det_guard.lease, slot_bytes + (((size_t)R + 15) & ~(size_t)15), stream);
        if (e == hipSuccess) det_slots = static_cast<float*>(det_guard.lease->p);
        if (e == hipSuccess) det_flags = reinterpret_cast<unsigned char*>(det_slots) + slot_bytes;
#ifdef WG_DET_POISON   // debugging aid (variant build): a slot that is read without having been written in THIS call shows up as NaN gradients
        if (e == hipSuccess) e = hipMemsetAsync(det_slots, 0xFF, slot_bytes, stream);
#endif
        if (e != hipSuccess) return hip_fail(e, "deterministic backward scratch");
    }
    // the gradient records are cleared by the launch that orders the tiles (one launch instead of a fill kernel + the ordering);
    // the deterministic mode's ordered sum writes every record in full and needs no clearing
    const bool clear_records = record && !det;
    if (clear_records && R <= 0) {
        StageScope scope_(WG_STAGE_RENDER_BACKWARD, stream);
        hipError_t e = hipMemsetAsync(geom.grad_rec, 0, (size_t)P * (wg::GRAD_REC_FLOATS + (dual ? 1 : 0)) * sizeof(float), stream);
        if (e != hipSuccess) return hip_fail(e, "gradient record memset");
    }
    if (R > 0) {
        // what the ordering launch clears on the side: the gradient records -- or, deterministic mode, the slots' flag bytes
        float* const clear_ptr = det ? reinterpret_cast<float*>(det_flags) : (clear_records ? geom.grad_rec : nullptr);
        const size_t clear_floats = det ? ((size_t)R + 3) / 4 : (size_t)P * (wg::GRAD_REC_FLOATS + (dual ? 1 : 0));
        WG_STAGE(WG_STAGE_TILE_RANGES, wg::launch_tile_order(img.tile_last, nullptr, img.order_bwd, gx * gy, clear_ptr, clear_floats, opt.backward_order_period, stream), "tile_order");
        WG_STAGE(WG_STAGE_RENDER_BACKWARD, wg::launch_render_backward(width, height, gx, gy, img, bin, geom, subpixel_offset, background, dL_dpix,
                                            dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, record, opt.exact_compositing != 0, dual ? second->dL_dpix2 : nullptr, det_slots, det_flags, (size_t)R, P, stream),
                 "render_backward");
    }

    wg::BwdParams bp;
    bp.P = P; bp.D = D; bp.M = M; bp.W = width; bp.H = height;
    bp.means3D = means3D; bp.shs = shs; bp.scales = scales; bp.scale_modifier = scale_modifier; bp.rotations = rotations;
    bp.cov3D = (cov3D_precomp != nullptr) ? cov3D_precomp : geom.cov3D;  // rasterizer_impl.cu:418
    bp.viewmatrix = viewmatrix; bp.projmatrix = projmatrix; bp.campos = campos;
    bp.tan_fovx = tan_fovx; bp.tan_fovy = tan_fovy;
    bp.focal_y = height / (2.0f * tan_fovy);
    bp.focal_x = width / (2.0f * tan_fovx);
    bp.kernel_size = kernel_size; bp.radii = radii;
    bp.dL_dcolor2 = dual ? second->dL_dcolor2 : nullptr;
    if (raw != nullptr) { bp.filter_3D = raw->filter_3D; bp.raw_opacities = raw->raw_opacities; }
    bp.nt_stream = opt.sh_stream == 1 || (opt.sh_stream < 0 && P <= opt.sh_stream_max_p);
    WG_STAGE(WG_STAGE_PREPROCESS_BACKWARD, wg::launch_preprocess_backward(bp, device_tone(tone, tone2, sh_second && dual), geom, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
                                            dL_dscale, dL_drot, record, stream),
             "preprocess_backward");
    {
        hipError_t e = det_guard.release();
        if (e != hipSuccess) return hip_fail(e, "deterministic backward scratch release");
    }
    return WG_OK;
}

int wg_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, unsigned char* present,
                    void* stream_) {
    (void)projmatrix;  // the reference projects but only tests view-space z (auxiliary.h:149-154)
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (P < 0) return WG_ERR_INVALID_ARGUMENT;
    if (P == 0) return WG_OK;
    if (!means3D || !viewmatrix || !present) return WG_ERR_INVALID_ARGUMENT;
    hipError_t e = wg::launch_mark_visible(P, means3D, viewmatrix, present, stream);
    if (e != hipSuccess) return hip_fail(e, "mark_visible");
    return WG_OK;
}

int wg_view_geometry(char* geom_buffer, int P, wg_geometry_view* out) {
    if (!geom_buffer || !out || P < 0) return WG_ERR_INVALID_ARGUMENT;
    wg::GeometryState g = wg::GeometryState::fromChunk(geom_buffer, (size_t)P, false);
    out->depths = g.depths;
    out->radii = g.radii;
    out->splats = reinterpret_cast<const float*>(g.splats);
    out->cov3D = g.cov3D;
    out->clamped = g.clamped;
    out->tiles_touched = g.tiles_touched;
    out->point_offsets = g.point_offsets;
    return WG_OK;
}

int wg_view_binning(char* binning_buffer, int R, wg_binning_view* out) {
    if (!binning_buffer || !out || R < 0) return WG_ERR_INVALID_ARGUMENT;
    wg::BinningState b = wg::BinningState::fromChunk(binning_buffer, (size_t)R, false);
    out->point_list = b.point_list;
    return WG_OK;
}

int wg_view_image(char* image_buffer, int width, int height, wg_image_view* out) {
    if (!image_buffer || !out || width <= 0 || height <= 0) return WG_ERR_INVALID_ARGUMENT;
    const int gx = (width + wg::TILE_X - 1) / wg::TILE_X, gy = (height + wg::TILE_Y - 1) / wg::TILE_Y;
    wg::ImageState img = wg::ImageState::fromChunk(image_buffer, (size_t)width * height, (size_t)gx * gy);
    out->final_T = img.final_T;
    out->accumulation = img.accum;
    out->n_contrib = img.n_contrib;
    out->ranges = reinterpret_cast<const uint32_t*>(img.ranges);
    out->tile_last = img.tile_last;
    out->tile_near = img.tile_near;
    out->split = reinterpret_cast<const uint32_t*>(img.split);
    out->order_fwd = img.order_fwd;
    out->order_key = img.order_key;
    out->order_bwd = img.order_bwd;
    return WG_OK;
}

int wg_set_option(const char* name, int value) {
    if (!name) return WG_ERR_INVALID_ARGUMENT;
    if (std::strcmp(name, "roctx") == 0) {  // WG_ERR_INVALID_ARGUMENT when no marker library can be loaded
        if (value != 0 && !g_roctx.load()) return WG_ERR_INVALID_ARGUMENT;
        g_roctx.enabled = value != 0;
        return WG_OK;
    }
    if (std::strcmp(name, "release_scratch") == 0) {   // (an action, not a setting)
        if (value == 0) return WG_OK;
        order_tables_release();
        return det_scratch_release();
    }
    std::lock_guard<std::mutex> l(g_opt_mu);
    wg::Options& o = g_opt;
    if (std::strcmp(name, "force_global_sort") == 0) { o.force_global_sort = value != 0; return WG_OK; }
    if (std::strcmp(name, "host_mailbox") == 0) { o.use_mailbox = value != 0; return WG_OK; }
    if (std::strcmp(name, "geometry_reuse") == 0) { o.geometry_reuse = value != 0; return WG_OK; }
    if (std::strcmp(name, "fused_scan") == 0) { o.fused_scan = value != 0; return WG_OK; }
    if (std::strcmp(name, "forward_order") == 0) { o.forward_order = value != 0; return WG_OK; }
    if (std::strcmp(name, "forward_order_slots") == 0) { if (value < 0 || value > (1 << 16)) return WG_ERR_INVALID_ARGUMENT; o.forward_order_slots = value; return WG_OK; }   // (read when a device's table is allocated)
    if (std::strcmp(name, "order_period") == 0) { if (value < 0 || value > 4096) return WG_ERR_INVALID_ARGUMENT; o.order_period = value; return WG_OK; }
    if (std::strcmp(name, "backward_order_period") == 0) { if (value < 0 || value > 4096) return WG_ERR_INVALID_ARGUMENT; o.backward_order_period = value; return WG_OK; }
    if (std::strcmp(name, "lazy_colour") == 0) { o.lazy_colour = value != 0; return WG_OK; }
    if (std::strcmp(name, "lazy_colour_min_p") == 0) { if (value < 0) return WG_ERR_INVALID_ARGUMENT; o.lazy_colour_min_p = value; return WG_OK; }
    if (std::strcmp(name, "sh_stream") == 0) { o.sh_stream = value < 0 ? -1 : (value != 0); return WG_OK; }
    if (std::strcmp(name, "sh_stream_max_p") == 0) { if (value < 0) return WG_ERR_INVALID_ARGUMENT; o.sh_stream_max_p = value; return WG_OK; }
    if (std::strcmp(name, "speculative_forward") == 0) {  // (a deferred frame still pending is dropped: its verdict goes unread)
        if (value < 0 || value > 2) return WG_ERR_INVALID_ARGUMENT;
        o.speculative = value; t_spec.clear(); t_wait.clear(); t_deferred.pending = false;
        return WG_OK;
    }
    if (std::strcmp(name, "spec_margin_pct") == 0) { if (value < 0 || value > 1000) return WG_ERR_INVALID_ARGUMENT; o.spec_margin_pct = value; return WG_OK; }
    if (std::strcmp(name, "band_list_min_p") == 0) { o.band_list_min_p = value > 0 ? value : 1; return WG_OK; }
    if (std::strcmp(name, "near_split") == 0) {  // (also clears the calling thread's back-off and density hint: a fresh start)
        o.near_split = value < 0 ? -1 : (value != 0);
        t_split_backoff = 0;
        t_near = NearAdapt();
        t_last_instances_per_tile = 0;
        if (t_mailbox.host) t_mailbox.host->far_report = 0ull;
        return WG_OK;
    }
    if (std::strcmp(name, "near_per_tile") == 0) { o.near_per_tile = value > 0 ? value : 0; return WG_OK; }
    if (std::strcmp(name, "near_adapt") == 0) {   // (also resets the calling thread's controller and drops a report still waiting in its mailbox)
        o.near_adapt = value != 0; t_near = NearAdapt(); t_split_backoff = 0;
        if (t_mailbox.host) t_mailbox.host->far_report = 0ull;
        return WG_OK;
    }
    if (std::strcmp(name, "box_count") == 0) { o.box_count = value < 0 ? -1 : (value != 0); return WG_OK; }
    if (std::strcmp(name, "depth_codes") == 0) {
        if (value != 0 && value != 1 && (value < 8 || value > 12)) return WG_ERR_INVALID_ARGUMENT;
        o.depth_codes = value;
        return WG_OK;
    }
    if (std::strcmp(name, "staged_scatter_cap") == 0) { o.staged_cap = value > 0 ? value : 0; return WG_OK; }
    if (std::strcmp(name, "staged_scatter") == 0) { o.staged_scatter = value < 0 ? -1 : (value != 0); return WG_OK; }
    if (std::strcmp(name, "lazy_sort") == 0) { o.lazy.enabled = value != 0; return WG_OK; }
    if (std::strcmp(name, "lazy_min_len") == 0 || std::strcmp(name, "lazy_target") == 0 || std::strcmp(name, "lazy_cap") == 0) {
        // min_len >= 256 (the selection samples 256 entries) and min_len, cap <= 2048 (the 8-keys-per-thread network)
        if (value < 1 || value > 2048) return WG_ERR_INVALID_ARGUMENT;
        if (name[5] == 'm') { if (value < 256) return WG_ERR_INVALID_ARGUMENT; o.lazy.min_len = (uint32_t)value; }
        else if (name[5] == 't') o.lazy.target = (uint32_t)value;
        else o.lazy.cap = (uint32_t)value;
        return WG_OK;
    }
    return WG_ERR_INVALID_ARGUMENT;
}

int wg_get_option(const char* name) {
    if (!name) return -1;
    if (std::strcmp(name, "roctx") == 0) return g_roctx.enabled ? 1 : 0;
    if (std::strcmp(name, "near_split_backoff") == 0) return (int)t_split_backoff;  // read-only, of the calling thread
    if (std::strcmp(name, "near_per_tile_now") == 0) return (int)t_near.cur;        // read-only: the calling thread's adapted aim (0 = none yet)
    if (std::strcmp(name, "near_floor_now") == 0) return (int)t_near.floor;         // read-only: the floor the adaptation does not go below (scripts/r6/near_trace.py)
    if (std::strcmp(name, "near_far_tiles_last") == 0) return (int)t_near.last_far;  // read-only: tiles that asked for far instances in the last reported split frame
    // read-only counters of the calling thread since the last wg_set_option("speculative_forward", ...)
    if (std::strcmp(name, "spec_frames") == 0) return (int)std::min<uint64_t>(t_wait.spec_frames, 0x7fffffffu);
    if (std::strcmp(name, "spec_misses") == 0) return (int)std::min<uint64_t>(t_wait.spec_misses, 0x7fffffffu);
    if (std::strcmp(name, "forward_polls") == 0) return (int)std::min<uint64_t>(t_wait.polls, 0x7fffffffu);
    if (std::strcmp(name, "forward_polls_waited") == 0) return (int)std::min<uint64_t>(t_wait.waited, 0x7fffffffu);
    if (std::strcmp(name, "forward_wait_us_total") == 0) return (int)std::min(t_wait.wait_us, 2147483647.0);
    if (std::strcmp(name, "forward_wait_us_last") == 0) return (int)std::min(t_wait.last_wait_us, 2147483647.0);
    const wg::Options o = options_snapshot();
    if (std::strcmp(name, "geometry_reuse") == 0) return o.geometry_reuse;
    if (std::strcmp(name, "fused_scan") == 0) return o.fused_scan;
    if (std::strcmp(name, "forward_order") == 0) return o.forward_order;
    if (std::strcmp(name, "forward_order_slots") == 0) return o.forward_order_slots;
    if (std::strcmp(name, "order_period") == 0) return o.order_period;
    if (std::strcmp(name, "backward_order_period") == 0) return o.backward_order_period;
    if (std::strcmp(name, "lazy_colour") == 0) return o.lazy_colour;
    if (std::strcmp(name, "lazy_colour_min_p") == 0) return o.lazy_colour_min_p;
    if (std::strcmp(name, "sh_stream") == 0) return o.sh_stream;
    if (std::strcmp(name, "sh_stream_max_p") == 0) return o.sh_stream_max_p;
    if (std::strcmp(name, "speculative_forward") == 0) return o.speculative;
    if (std::strcmp(name, "spec_margin_pct") == 0) return o.spec_margin_pct;
    if (std::strcmp(name, "force_global_sort") == 0) return o.force_global_sort ? 1 : 0;
    if (std::strcmp(name, "host_mailbox") == 0) return o.use_mailbox ? 1 : 0;
    if (std::strcmp(name, "lazy_sort") == 0) return o.lazy.enabled ? 1 : 0;
    if (std::strcmp(name, "depth_codes") == 0) return o.depth_codes;
    if (std::strcmp(name, "band_list_min_p") == 0) return o.band_list_min_p;
    if (std::strcmp(name, "staged_scatter") == 0) return o.staged_scatter;
    if (std::strcmp(name, "near_split") == 0) return o.near_split;
    if (std::strcmp(name, "near_per_tile") == 0) return o.near_per_tile;
    if (std::strcmp(name, "near_adapt") == 0) return o.near_adapt;
    if (std::strcmp(name, "box_count") == 0) return o.box_count;
    if (std::strcmp(name, "lazy_min_len") == 0) return (int)o.lazy.min_len;
    if (std::strcmp(name, "lazy_target") == 0) return (int)o.lazy.target;
    if (std::strcmp(name, "lazy_cap") == 0) return (int)o.lazy.cap;
    return -1;
}

int wg_profile_enable(int enable) {
    g_prof.enabled = enable != 0;
    return WG_OK;
}

int wg_profile_reset(void) {
    std::lock_guard<std::mutex> l(g_prof.mu);
    for (auto& r : g_prof.pending) { g_prof.pool[r.dev].push_back(r.a); g_prof.pool[r.dev].push_back(r.b); }
    g_prof.pending.clear();
    g_prof.totals = wg_stage_times{};
    return WG_OK;
}

int wg_profile_read(wg_stage_times* out) {
    if (!out) return WG_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> l(g_prof.mu);
    for (auto& r : g_prof.pending) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return hip_fail(e, "profile event sync");
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) return hip_fail(e, "profile event elapsed");
        g_prof.totals.total_ms[r.stage] += ms;
        g_prof.totals.launches[r.stage] += 1;
        g_prof.pool[r.dev].push_back(r.a);
        g_prof.pool[r.dev].push_back(r.b);
    }
    g_prof.pending.clear();
    *out = g_prof.totals;
    return WG_OK;
}

const char* wg_stage_name(int stage) {
    static const char* names[WG_STAGE_COUNT] = {"preprocess", "scan", "duplicate_keys", "sort", "tile_ranges",
                                                "render_forward", "render_backward", "preprocess_backward", "render_fixup"};
    return (stage >= 0 && stage < WG_STAGE_COUNT) ? names[stage] : "?";
}

const char* wg_status_string(int status) {
    switch (status) {
        case WG_OK: return "ok";
        case WG_ERR_INVALID_ARGUMENT: return "invalid argument";
        case WG_ERR_ALLOC: return "scratch allocator returned NULL";
        case WG_ERR_HIP: return "HIP runtime error";
        case WG_ERR_OVERFLOW: return "more than 2^31-1 tile instances";
        case WG_ERR_SPECULATION: return "the previous forward call of this thread (deferred speculation) did not fit its predicted binning buffer: "
                                        "its image is NaN and its gradients are zero -- repeat the step";
        default: return status > 0 ? "ok (num_rendered)" : "unknown error";
    }
}

const char* wg_last_hip_error(void) { return g_last_hip_error.c_str(); }

const char* wg_version(void) { return "wg_rasterizer 0.6 (gfx950)"; }

}  // extern "C"
