// Multi-GPU forms of the transform for a SINGLE-PROCESS host (Toyni is one Rust process: src/ntt.rs:128-141,
// src/fibonacci.rs:99-103).  Included by toyni_hip.hip (one translation unit); declarations in include/toyni_hip.h 2c.
//
//   1. toyni_ntt_host_multi_gpu   -- the batched host-slice transform sharded over devices, no exchange;
//                                    contexts are cached per (device, lane, n) for the life of the process.
//   2. toyni_ntt_slab_multi_gpu_* -- ONE size-n transform over G devices with ONE exchange (BASELINE
//                                    configs[4]): slab pass -> exchange of contiguous row blocks -> relayout -> row transforms
//                                    (the transpose-free form of header section 2b).  The exchange is either
//                                      * peer copies: every destination pulls its G blocks with hipMemcpyPeerAsync, one copy
//                                        stream per source so that all incoming xGMI links carry data at once, or
//                                      * RCCL: one ncclGroupStart / ncclSend+ncclRecv per peer / ncclGroupEnd over communicators
//                                        from ncclCommInitAll.  librccl (573 MB) is loaded on first use with dlopen, not linked:
//                                        single-GPU users of this library never pay for it, and a process that already holds an
//                                        RCCL (PyTorch) shares that copy by SONAME.
// The multi-PROCESS form of the same algorithm (one rank per GPU, torch.distributed / RCCL all-to-all) is toyni_amd/dist.py;
// both drive the same device entry points (toyni_ntt_slab_pass_device, toyni_ntt_slab_rows_device -- the relayout folded into the
// row transforms' addressing where the pass shapes allow it, toyni_ntt_slab_relayout_device + toyni_ntt_device otherwise).
#pragma once
#include <dlfcn.h>

#include <memory>

// ---- per-(device, lane, n) context cache: process lifetime, never destroyed (src/ntt.rs:128-141) ----
namespace {

struct CtxKey {
    int device, lane;
    uint32_t n;
    bool operator<(const CtxKey& o) const { return device != o.device ? device < o.device : lane != o.lane ? lane < o.lane : n < o.n; }
};

int cached_ctx(int device, int lane, uint32_t n, toyni_ntt_ctx** out) {
    static std::mutex mu;
    static std::map<CtxKey, toyni_ntt_ctx*> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find({device, lane, n});
    if (it == cache.end()) {
        toyni_ntt_ctx* c = nullptr;
        int rc = toyni_ntt_ctx_create(n, device, &c);
        if (rc) return rc;
        it = cache.emplace(CtxKey{device, lane, n}, c).first;
    }
    *out = it->second;
    return TOYNI_OK;
}

// lane index of entry d among the entries that name the same device (a device may be listed more than once)
int lane_of(const int* devices, int d) {
    int lane = 0;
    for (int i = 0; i < d; ++i) lane += devices[i] == devices[d];
    return lane;
}

// ---- RCCL, bound at first use ----
// Plain C declarations of the five entry points used (rccl.h: ncclCommInitAll :236, ncclSend :700, ncclUint32 = 3); the
// header itself is not included so that the crate's hip/ directory builds without RCCL's include path.
typedef void* rccl_comm_t;
struct RcclApi {
    int (*CommInitAll)(rccl_comm_t*, int, const int*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int version = 0;                 // NCCL_VERSION_CODE of the loaded library (major * 10000 + minor * 100 + patch for >= 2.9)
    bool ok = false;
};
// tests/test_abi.py parses /opt/rocm/include/rccl/rccl.h and checks this value and the five signatures above against it
constexpr int RCCL_UINT32 = 3;
// ncclSend / ncclRecv inside groups exist since NCCL 2.7; the grouped p2p of every RCCL that ships with ROCm >= 5 is newer.  A
// library that does not report a version at all, or an older one, is treated as absent.
constexpr int RCCL_MIN_VERSION = 2700;

const RcclApi& rccl() {
    static const RcclApi api = [] {
        RcclApi a;
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return a;
        a.CommInitAll = (decltype(a.CommInitAll))dlsym(h, "ncclCommInitAll");
        a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
        a.Send = (decltype(a.Send))dlsym(h, "ncclSend");
        a.Recv = (decltype(a.Recv))dlsym(h, "ncclRecv");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
        a.GetVersion = (decltype(a.GetVersion))dlsym(h, "ncclGetVersion");
        if (a.GetVersion && a.GetVersion(&a.version) != 0) a.version = 0;
        a.ok = a.CommInitAll && a.GroupStart && a.GroupEnd && a.Send && a.Recv && a.version >= RCCL_MIN_VERSION;
        if (!a.ok && std::getenv("TOYNI_VERBOSE"))
            std::fprintf(stderr, "toyni_hip: librccl loaded but unusable (version code %d, need >= %d, or a symbol is missing)\n", a.version, RCCL_MIN_VERSION);
        return a;
    }();
    return api;
}

// ---- one transform over G lanes ----
struct SlabLane {
    int device = 0;
    toyni_ntt_ctx* big = nullptr;   // size-n context: slab pass, relayout tables
    toyni_ntt_ctx* row = nullptr;   // size-S1 context: the row transforms
    hipStream_t stream = nullptr;   // the lane's compute stream
    std::vector<hipStream_t> pull;  // one copy stream per source lane (peer-copy exchange)
    hipEvent_t ready = nullptr;     // "my outgoing blocks are final"
    std::vector<hipEvent_t> got;    // per source lane: "its block has landed here"
    uint32_t* d_xchg = nullptr;     // [G][r][w] pieces: landing buffer (forward) / outgoing buffer (inverse); n / G words
    // host form only
    uint32_t* d_slab = nullptr;     // [M1][w]
    uint32_t* d_rows = nullptr;     // [r][S1]
    uint64_t* d_stage = nullptr;    // n / G u64
    rccl_comm_t comm = nullptr;
};

struct SlabGroup {
    std::mutex mu;                  // one transform at a time per group (blocking entry points)
    std::vector<SlabLane> lanes;
    uint32_t n = 0;
    size_t m1 = 0, s1 = 0;
    bool rccl_ready = false, host_buffers = false;
    bool distinct_devices = false;  // at least two lanes on different devices: the exchange crosses a link
    bool checked[2] = {false, false};  // first-use self-check passed, per exchange kind
    // registered groups live as long as the process; this runs for a group whose construction failed half way
    ~SlabGroup() {
        for (SlabLane& L : lanes) {
            DeviceGuard guard(L.device);
            if (L.stream) (void)hipStreamDestroy(L.stream);
            if (L.ready) (void)hipEventDestroy(L.ready);
            for (hipStream_t st : L.pull) if (st) (void)hipStreamDestroy(st);
            for (hipEvent_t ev : L.got) if (ev) (void)hipEventDestroy(ev);
            if (L.d_xchg) (void)hipFree(L.d_xchg);
            if (L.d_slab) (void)hipFree(L.d_slab);
            if (L.d_rows) (void)hipFree(L.d_rows);
            if (L.d_stage) (void)hipFree(L.d_stage);
        }
    }
};

#define MG_TRY(expr) do { int _rc = (int)(expr); if (_rc) return _rc; } while (0)

// Fault injection for the tests (measurement build only, include/toyni_hip_tools.h: toyni_tools_inject): bit 0 = every pair of
// different lanes counts as two devices WITHOUT peer access; bit 1 = the first-use self-check also runs for groups whose lanes all
// sit on one device; bit 2 = the self-check sees one corrupted word (a broken link).  Always 0 in the shipped library.
#ifdef TOYNI_TOOLS
std::atomic<unsigned> g_inject{0};
inline unsigned injected() { return g_inject.load(std::memory_order_relaxed); }
#else
constexpr unsigned injected() { return 0u; }
#endif
constexpr unsigned INJECT_DENY_PEER = 1u, INJECT_FORCE_SELF_CHECK = 2u, INJECT_CORRUPT = 4u;

// TOYNI_VERBOSE=1: one line per lane (device ordinal, PCI bus id, peer access to the other lanes) when a group is built, so that
// the first run on a multi-GPU node shows which devices and links it is using.
bool verbose() {
    static const bool v = [] { const char* e = std::getenv("TOYNI_VERBOSE"); return e && e[0] && e[0] != '0'; }();
    return v;
}

// Direct access from device `self` to device `peer` (xGMI or PCIe P2P).  hipMemcpyPeerAsync also works WITHOUT it -- the runtime
// then stages every block through host memory, silently, at a fraction of the link rate -- so a group refuses such a pair unless
// TOYNI_ALLOW_STAGED_PEER=1 says that staged copies are acceptable (the error used to be swallowed).
int enable_peer(int self, int peer) {
    if (injected() & INJECT_DENY_PEER) {
        std::fprintf(stderr, "toyni_hip: device %d has no peer access to device %d (injected by the test hook)\n", self, peer);
        return TOYNI_E_NO_PEER_ACCESS;
    }
    if (self == peer) return TOYNI_OK;
    static const bool allow_staged = [] { const char* e = std::getenv("TOYNI_ALLOW_STAGED_PEER"); return e && e[0] == '1'; }();
    int can = 0;
    hipError_t e = hipDeviceCanAccessPeer(&can, self, peer);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    if (!can) {
        if (verbose() || !allow_staged)
            std::fprintf(stderr, "toyni_hip: device %d has no peer access to device %d%s\n", self, peer,
                         allow_staged ? " (TOYNI_ALLOW_STAGED_PEER=1: copies are staged through the host)" : "");
        return allow_staged ? TOYNI_OK : TOYNI_E_NO_PEER_ACCESS;
    }
    e = hipDeviceEnablePeerAccess(peer, 0);   // the current device is `self` (DeviceGuard of the caller)
    if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return TOYNI_OK; }
    if (e != hipSuccess) (void)hipGetLastError();
    return (int)e;
}

int slab_group(const int* devices, int ndev, uint32_t n, SlabGroup** out) {
    static std::mutex mu;
    static std::map<std::pair<std::vector<int>, uint32_t>, SlabGroup*> groups;  // process lifetime
    std::lock_guard<std::mutex> lk(mu);
    const std::vector<int> key(devices, devices + ndev);
    auto it = groups.find({key, n});
    if (it != groups.end()) { *out = it->second; return TOYNI_OK; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return TOYNI_E_NO_DEVICE; }
    for (int d = 0; d < ndev; ++d) if (devices[d] < 0 || devices[d] >= count) return TOYNI_E_RANGE;
    // the shape is checked on the first lane's context before anything else is allocated
    toyni_ntt_ctx* first = nullptr;
    MG_TRY(cached_ctx(devices[0], 0, n, &first));
    const size_t m1 = toyni_ntt_ctx_first_pass_points(first);
    if (!m1) return TOYNI_E_INVALID_SIZE;  // n <= 1024: a single pass, nothing to split
    const size_t s1 = (size_t)n / m1;
    if (s1 / (size_t)ndev < 32 || m1 < (size_t)ndev) return TOYNI_E_RANGE;
    std::unique_ptr<SlabGroup> owner(new SlabGroup());  // released into the registry only when complete
    SlabGroup* g = owner.get();
    g->n = n;
    g->m1 = m1;
    g->s1 = s1;
    g->lanes.resize((size_t)ndev);
    for (int d = 0; d < ndev; ++d) {
        SlabLane& L = g->lanes[(size_t)d];
        L.device = devices[d];
        const int lane = lane_of(devices, d);
        MG_TRY(cached_ctx(L.device, lane, n, &L.big));
        MG_TRY(cached_ctx(L.device, lane, (uint32_t)g->s1, &L.row));
        DeviceGuard guard(L.device);
        MG_TRY(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        MG_TRY(hipEventCreateWithFlags(&L.ready, hipEventDisableTiming));
        L.pull.assign((size_t)ndev, nullptr);
        L.got.assign((size_t)ndev, nullptr);
        for (int sidx = 0; sidx < ndev; ++sidx) {
            MG_TRY(hipStreamCreateWithFlags(&L.pull[(size_t)sidx], hipStreamNonBlocking));
            MG_TRY(hipEventCreateWithFlags(&L.got[(size_t)sidx], hipEventDisableTiming));
        }
        MG_TRY(hipMalloc((void**)&L.d_xchg, ((size_t)n / (size_t)ndev) * sizeof(uint32_t)));
        // direct xGMI access between the lanes' devices: checked, not assumed
        for (int p = 0; p < ndev; ++p)
            if (p != d) MG_TRY(enable_peer(L.device, devices[p]));
        if (verbose()) {
            char bus[32] = "?";
            if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, L.device) != hipSuccess) (void)hipGetLastError();
            std::fprintf(stderr, "toyni_hip: slab group n=2^%d lane %d/%d -> device %d (PCI %s), slab %zu x %zu, row block %zu x %zu\n",
                         ilog2(n), d, ndev, L.device, bus, m1, s1 / (size_t)ndev, m1 / (size_t)ndev, s1);
        }
    }
    for (int a = 0; a < ndev; ++a)
        for (int b = a + 1; b < ndev; ++b) g->distinct_devices |= devices[a] != devices[b];
    groups.emplace(std::make_pair(key, n), g);
    owner.release();
    *out = g;
    return TOYNI_OK;
}

int slab_group_rccl(SlabGroup* g) {
    if (g->rccl_ready) return TOYNI_OK;
    const RcclApi& api = rccl();
    if (!api.ok) return TOYNI_E_NO_RCCL;
    const size_t G = g->lanes.size();
    std::vector<int> devs(G);
    for (size_t d = 0; d < G; ++d) devs[d] = g->lanes[d].device;
    for (size_t a = 0; a < G; ++a)
        for (size_t b = a + 1; b < G; ++b)
            if (devs[a] == devs[b]) return TOYNI_E_RANGE;  // one communicator rank per device: no duplicate lanes with RCCL
    std::vector<rccl_comm_t> comms(G, nullptr);
    if (api.CommInitAll(comms.data(), (int)G, devs.data()) != 0) return TOYNI_E_RCCL;
    for (size_t d = 0; d < G; ++d) g->lanes[d].comm = comms[d];
    g->rccl_ready = true;
    return TOYNI_OK;
}

// Block h of src lane g's [G][r][w] view -> piece g of dst lane h's [G][r][w] view; `blk` = r * w words.
//   send(g) / recv(h): base pointers of the lane's outgoing / landing buffers
// On return every lane's compute stream is ordered after the arrival of all its pieces.
template <class SendPtr, class RecvPtr>
int slab_exchange(SlabGroup* g, size_t blk, int exchange, SendPtr&& send, RecvPtr&& recv) {
    const size_t G = g->lanes.size();
    if (exchange == TOYNI_EXCHANGE_RCCL) {
        const RcclApi& api = rccl();
        if (api.GroupStart() != 0) return TOYNI_E_RCCL;
        bool queued = true;
        for (size_t a = 0; a < G && queued; ++a) {
            SlabLane& L = g->lanes[a];
            DeviceGuard guard(L.device);      // the communicator's device is current while its calls are queued
            for (size_t b = 0; b < G && queued; ++b) {  // in stream order on the lane's compute stream: no events needed
                if (b == a && G > 1) continue;          // the lane's own block never enters the communicator (below)
                queued = api.Send(send(a) + b * blk, blk, RCCL_UINT32, (int)b, L.comm, L.stream) == 0 &&
                         api.Recv(recv(a) + b * blk, blk, RCCL_UINT32, (int)b, L.comm, L.stream) == 0;
            }
        }
        const bool closed = api.GroupEnd() == 0;   // the group is closed on the error path too: RCCL's group state is per thread
        if (!(queued && closed)) return TOYNI_E_RCCL;
        for (size_t a = 0; a < G && G > 1; ++a) {  // block a of lane a: a device-local copy on the same stream (a group of one keeps
            SlabLane& L = g->lanes[a];             // the send-to-self form, which is what a one-GPU box can test of the RCCL path)
            DeviceGuard guard(L.device);
            MG_TRY(hipMemcpyAsync(recv(a) + a * blk, send(a) + a * blk, blk * sizeof(uint32_t), hipMemcpyDeviceToDevice, L.stream));
        }
        return TOYNI_OK;
    }
    for (size_t a = 0; a < G; ++a) {  // every producer marks its outgoing blocks final
        DeviceGuard guard(g->lanes[a].device);
        MG_TRY(hipEventRecord(g->lanes[a].ready, g->lanes[a].stream));
    }
    for (size_t h = 0; h < G; ++h) {  // every destination pulls, one copy stream per source: all incoming links at once
        SlabLane& D = g->lanes[h];
        DeviceGuard guard(D.device);
        for (size_t a = 0; a < G; ++a) {
            SlabLane& S = g->lanes[a];
            hipStream_t cs = D.pull[a];
            MG_TRY(hipStreamWaitEvent(cs, S.ready, 0));
            if (S.device == D.device)
                MG_TRY(hipMemcpyAsync(recv(h) + a * blk, send(a) + h * blk, blk * sizeof(uint32_t), hipMemcpyDeviceToDevice, cs));
            else
                MG_TRY(hipMemcpyPeerAsync(recv(h) + a * blk, D.device, send(a) + h * blk, S.device, blk * sizeof(uint32_t), cs));
            MG_TRY(hipEventRecord(D.got[a], cs));
            MG_TRY(hipStreamWaitEvent(D.stream, D.got[a], 0));
        }
    }
    return TOYNI_OK;
}

// Peer-copy exchange in K pieces (sub-blocks of rq = r / K rows of every destination's row block), so that the relayout and the row
// transforms of piece q run while pieces q + 1 .. are still on the links (forward), and piece q is on the links while the row
// transforms of piece q + 1 run (inverse) -- the single-process counterpart of `chunks=K` in toyni_amd/dist.py.  The landing /
// outgoing buffer d_xchg is piece-major: [K][G][rq][w].  K = TOYNI_SLAB_PIECES, default 1: with all eight lanes on ONE device there
// is nothing to overlap and the extra launches and copies are pure host-side enqueue cost (2^27: 3.8 ms per forward + inverse at
// K = 1, 5.9 at 4, 10.3 at 8); whether K > 1 pays on xGMI links has to be measured on a multi-GPU box.
// Measurement build only (toyni_tools_slab_phases, include/toyni_hip_tools.h): HIP events on LANE 0's compute stream between the stages of a
// transform -- slab pass / exchange (the time the stream waits for its incoming blocks) / relayout / row transforms -- so that
// bench.py's single-process line can say where a step's time goes.  One-piece exchanges only.
#ifdef TOYNI_TOOLS
struct SlabPhaseRec {
    std::mutex mu;
    bool on = false;
    struct Rec { hipEvent_t e[5]; bool inverse; };
    std::vector<Rec> recs;
    Rec cur{};
    SlabGroup* cur_group = nullptr;   // the group whose transform `cur` belongs to: one transform at a time is recorded
    int device = 0;
};
inline SlabPhaseRec& slab_phase_rec() { static SlabPhaseRec r; return r; }
inline void slab_phase_mark(SlabGroup* g, int k, bool inverse) {
    SlabPhaseRec& pr = slab_phase_rec();
    std::lock_guard<std::mutex> lk(pr.mu);
    if (!pr.on) return;
    SlabLane& L = g->lanes[0];
    DeviceGuard guard(L.device);
    if (k == 0) {
        // a transform that left through MG_TRY between two marks never reached k == 4: its events are released here, not leaked
        for (int j = 0; j < 5; ++j) if (pr.cur.e[j]) (void)hipEventDestroy(pr.cur.e[j]);
        pr.cur = SlabPhaseRec::Rec{};
        pr.cur.inverse = inverse;
        pr.cur_group = g;
        pr.device = L.device;
    }
    if (pr.cur_group != g) return;   // another group's transform is being recorded: its marks are not mixed into this record
    if (hipEventCreate(&pr.cur.e[k]) != hipSuccess) { pr.cur.e[k] = nullptr; (void)hipGetLastError(); return; }
    if (hipEventRecord(pr.cur.e[k], L.stream) != hipSuccess) { (void)hipEventDestroy(pr.cur.e[k]); pr.cur.e[k] = nullptr; (void)hipGetLastError(); return; }
    if (k == 4) {
        bool complete = true;
        for (int j = 0; j < 5; ++j) complete = complete && pr.cur.e[j] != nullptr;
        if (complete) pr.recs.push_back(pr.cur);   // only records with all five events are read back
        else for (int j = 0; j < 5; ++j) if (pr.cur.e[j]) (void)hipEventDestroy(pr.cur.e[j]);
        pr.cur = SlabPhaseRec::Rec{};
        pr.cur_group = nullptr;
    }
}
#define MG_PHASE(g, k, inverse) slab_phase_mark((g), (k), (inverse))
#else
#define MG_PHASE(g, k, inverse) ((void)0)
#endif

size_t slab_pieces(size_t r) {
    static const size_t want = [] { const char* env = std::getenv("TOYNI_SLAB_PIECES"); const long v = env ? std::atol(env) : 1; return (size_t)(v < 1 ? 1 : v); }();
    size_t k = 1;
    while (k * 2 <= want && k * 2 <= r) k *= 2;
    return k;
}

int slab_run_pieces(SlabGroup* g, uint32_t* const* d_slabs, uint32_t* const* d_rows, bool inverse) {
    const size_t G = g->lanes.size();
    const size_t w = g->s1 / G, r = g->m1 / G, blk = r * w, K = slab_pieces(r), rq = r / K, piece = rq * w;
    auto copy = [&](SlabLane& D, SlabLane& S, hipStream_t cs, uint32_t* dst, const uint32_t* src) -> int {
        if (S.device == D.device) return (int)hipMemcpyAsync(dst, src, piece * sizeof(uint32_t), hipMemcpyDeviceToDevice, cs);
        return (int)hipMemcpyPeerAsync(dst, D.device, src, S.device, piece * sizeof(uint32_t), cs);
    };
    const bool phases = K == 1;   // per-phase events (measurement build) make sense for the unpipelined exchange only
    if (!inverse) {
        for (size_t a = 0; a < G; ++a) {  // M1-point column transforms x w_n^(j' k1), in place on the slab; then "my blocks are final"
            SlabLane& L = g->lanes[a];
            if (a == 0 && phases) MG_PHASE(g, 0, false);
            MG_TRY(toyni_ntt_slab_pass_device(L.big, d_slabs[a], w, a * w, 0, L.stream));
            if (a == 0 && phases) MG_PHASE(g, 1, false);
            DeviceGuard guard(L.device);
            MG_TRY(hipEventRecord(L.ready, L.stream));
        }
        for (size_t q = 0; q < K; ++q) {
            for (size_t h = 0; h < G; ++h) {  // every destination pulls piece q of its row block from every source, one stream per source
                SlabLane& D = g->lanes[h];
                DeviceGuard guard(D.device);
                for (size_t a = 0; a < G; ++a) {
                    SlabLane& S = g->lanes[a];
                    hipStream_t cs = D.pull[a];
                    if (q == 0) MG_TRY(hipStreamWaitEvent(cs, S.ready, 0));
                    MG_TRY(copy(D, S, cs, D.d_xchg + q * G * piece + a * piece, d_slabs[a] + h * blk + q * piece));
                    MG_TRY(hipEventRecord(D.got[a], cs));
                    MG_TRY(hipStreamWaitEvent(D.stream, D.got[a], 0));
                }
            }
            for (size_t h = 0; h < G; ++h) {  // piece q: pieces -> rows [rq][S1], then the size-S1 transforms over j'
                SlabLane& L = g->lanes[h];
                uint32_t* part = d_rows[h] + q * rq * g->s1;
                if (h == 0 && phases) MG_PHASE(g, 2, false);   // behind the stream's waits for its incoming blocks: the exchange has landed
                if (h == 0 && phases) MG_PHASE(g, 3, false);   // (round 5: no relayout sweep -- the row transforms' first pass reads the pieces)
                MG_TRY(toyni_ntt_slab_rows_device(L.big, L.row, L.d_xchg + q * G * piece, part, rq, h * r + q * rq, G, 0, nullptr, L.stream));
                if (h == 0 && phases) MG_PHASE(g, 4, false);
            }
        }
    } else {
        for (size_t q = 0; q < K; ++q) {
            for (size_t h = 0; h < G; ++h) {  // inverse size-S1 transforms of piece q, then x w_n^-(k1 j') into outgoing pieces
                SlabLane& L = g->lanes[h];
                uint32_t* part = d_rows[h] + q * rq * g->s1;
                if (h == 0 && phases) MG_PHASE(g, 0, true);
                // inverse size-S1 transforms whose last pass writes the outgoing pieces, times w_n^-(k1 j') (no relayout sweep)
                MG_TRY(toyni_ntt_slab_rows_device(L.big, L.row, part, L.d_xchg + q * G * piece, rq, h * r + q * rq, G, 1, nullptr, L.stream));
                if (h == 0 && phases) MG_PHASE(g, 1, true);
                if (h == 0 && phases) MG_PHASE(g, 2, true);
                DeviceGuard guard(L.device);
                MG_TRY(hipEventRecord(L.ready, L.stream));   // re-recorded per piece: the waits below capture this recording
            }
            for (size_t a = 0; a < G; ++a) {  // lane a's slab receives rows of block h, piece q, from lane h
                SlabLane& D = g->lanes[a];
                DeviceGuard guard(D.device);
                for (size_t h = 0; h < G; ++h) {
                    SlabLane& S = g->lanes[h];
                    hipStream_t cs = D.pull[h];
                    MG_TRY(hipStreamWaitEvent(cs, S.ready, 0));
                    MG_TRY(copy(D, S, cs, d_slabs[a] + h * blk + q * piece, S.d_xchg + q * G * piece + a * piece));
                    if (q + 1 == K) {
                        MG_TRY(hipEventRecord(D.got[h], cs));
                        MG_TRY(hipStreamWaitEvent(D.stream, D.got[h], 0));
                    }
                }
            }
        }
        for (size_t a = 0; a < G; ++a) {
            SlabLane& L = g->lanes[a];
            if (a == 0 && phases) MG_PHASE(g, 3, true);    // behind the waits for the incoming rows
            MG_TRY(toyni_ntt_slab_pass_device(L.big, d_slabs[a], w, a * w, 1, L.stream));  // closing inverse column transforms, 1/M1
            if (a == 0 && phases) MG_PHASE(g, 4, true);
        }
    }
    return TOYNI_OK;
}

int slab_run(SlabGroup* g, uint32_t* const* d_slabs, uint32_t* const* d_rows, bool inverse, int exchange) {
    const size_t G = g->lanes.size();
    const size_t w = g->s1 / G, r = g->m1 / G, blk = r * w;
    if (exchange == TOYNI_EXCHANGE_PEER_COPY) return slab_run_pieces(g, d_slabs, d_rows, inverse);
    if (exchange != TOYNI_EXCHANGE_RCCL) return TOYNI_E_RANGE;
    MG_TRY(slab_group_rccl(g));
    if (!inverse) {
        for (size_t a = 0; a < G; ++a) {  // M1-point column transforms x w_n^(j' k1), in place on the slab
            SlabLane& L = g->lanes[a];
            if (a == 0) MG_PHASE(g, 0, false);
            MG_TRY(toyni_ntt_slab_pass_device(L.big, d_slabs[a], w, a * w, 0, L.stream));
            if (a == 0) MG_PHASE(g, 1, false);
        }
        // row block h of a slab (k1 in lane h's chunk) is contiguous: no packing
        MG_TRY(slab_exchange(g, blk, exchange, [&](size_t a) { return d_slabs[a]; }, [&](size_t h) { return g->lanes[h].d_xchg; }));
        for (size_t h = 0; h < G; ++h) {
            SlabLane& L = g->lanes[h];
            if (h == 0) MG_PHASE(g, 2, false);   // the grouped send / recv run in stream order: behind them
            if (h == 0) MG_PHASE(g, 3, false);   // (round 5: no relayout sweep)
            MG_TRY(toyni_ntt_slab_rows_device(L.big, L.row, L.d_xchg, d_rows[h], r, h * r, G, 0, nullptr, L.stream));   // pieces -> size-S1 transforms over j'
            if (h == 0) MG_PHASE(g, 4, false);
        }
    } else {
        for (size_t h = 0; h < G; ++h) {
            SlabLane& L = g->lanes[h];
            if (h == 0) MG_PHASE(g, 0, true);
            MG_TRY(toyni_ntt_slab_rows_device(L.big, L.row, d_rows[h], L.d_xchg, r, h * r, G, 1, nullptr, L.stream));   // inverse size-S1 (1/S1), x w_n^-(k1 j'), into pieces
            if (h == 0) MG_PHASE(g, 1, true);
            if (h == 0) MG_PHASE(g, 2, true);
        }
        MG_TRY(slab_exchange(g, blk, exchange, [&](size_t h) { return g->lanes[h].d_xchg; }, [&](size_t a) { return d_slabs[a]; }));
        for (size_t a = 0; a < G; ++a) {
            SlabLane& L = g->lanes[a];
            if (a == 0) MG_PHASE(g, 3, true);
            MG_TRY(toyni_ntt_slab_pass_device(L.big, d_slabs[a], w, a * w, 1, L.stream));  // closing inverse column transforms, 1/M1
            if (a == 0) MG_PHASE(g, 4, true);
        }
    }
    return TOYNI_OK;
}

// Waits for everything the group has enqueued: the copy streams first (they feed the compute streams), then the compute streams.
int slab_sync(SlabGroup* g) {
    int rc = TOYNI_OK;
    for (int phase = 0; phase < 2; ++phase)
        for (SlabLane& L : g->lanes) {
            DeviceGuard guard(L.device);
            if (phase == 0) {
                for (hipStream_t cs : L.pull) {
                    hipError_t e = cs ? hipStreamSynchronize(cs) : hipSuccess;
                    if (e != hipSuccess && rc == TOYNI_OK) rc = (int)e;
                }
            } else {
                hipError_t e = L.stream ? hipStreamSynchronize(L.stream) : hipSuccess;
                if (e != hipSuccess && rc == TOYNI_OK) rc = (int)e;
            }
        }
    return rc;
}

// Drain on EVERY exit path of a blocking entry point: an early return between enqueued uploads, peer copies, kernels and
// downloads must not hand the caller's host slice or device buffers back while a stream still reads or writes them.
struct SlabDrain {
    SlabGroup* g;
    bool armed = true;
    explicit SlabDrain(SlabGroup* g_) : g(g_) {}
    int finish() { armed = false; return slab_sync(g); }
    ~SlabDrain() { if (armed) (void)slab_sync(g); }
};

// [R][C] -> [C][R] through a 32 x 33 LDS tile, converting the element type (u32 -> u64 widens, u64 -> u32 reduces like
// BabyBear::new): the natural-order <-> row-block re-layout of the host form, on the device (PCIe then moves whole runs)
template <class IN, class OUT>
__global__ void __launch_bounds__(256) transpose_convert_kernel(const IN* __restrict__ in, OUT* __restrict__ out, uint32_t R, uint32_t C) {
    __shared__ uint32_t tile[32][33];
    const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;  // 32 x 8
    const uint32_t tiles_c = (C + 31u) / 32u;
    const uint64_t ntiles = (uint64_t)tiles_c * ((R + 31u) / 32u);
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t r0 = (uint32_t)(t / tiles_c) * 32u, c0 = (uint32_t)(t % tiles_c) * 32u;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t rr = r0 + ty + 8u * k, cc = c0 + tx;
            if (rr < R && cc < C) tile[ty + 8u * k][tx] = (uint32_t)(in[(uint64_t)rr * C + cc] % BB_P);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t cc = c0 + ty + 8u * k, rr = r0 + tx;
            if (rr < R && cc < C) out[(uint64_t)cc * R + rr] = tile[tx][ty + 8u * k];
        }
        __syncthreads();
    }
}

int slab_host_buffers(SlabGroup* g) {
    if (g->host_buffers) return TOYNI_OK;
    const size_t per = (size_t)g->n / g->lanes.size();
    for (SlabLane& L : g->lanes) {  // a call that failed half way left some of these allocated: only the missing ones are made
        DeviceGuard guard(L.device);
        if (!L.d_slab) MG_TRY(hipMalloc((void**)&L.d_slab, per * sizeof(uint32_t)));
        if (!L.d_rows) MG_TRY(hipMalloc((void**)&L.d_rows, per * sizeof(uint32_t)));
        if (!L.d_stage) MG_TRY(hipMalloc((void**)&L.d_stage, per * sizeof(uint64_t)));
    }
    g->host_buffers = true;
    return TOYNI_OK;
}

// Host slice in natural order, in place: n u64 elements.  Per lane: strided upload of its column slab (forward) or its
// row block (inverse), the device form, and the mirror-image download; the natural-order <-> row-block transposition
// runs on the device, so PCIe moves runs of S1 / G (slab) or M1 / G (rows) elements.
int slab_host_transform(const int* devices, int ndev, uint32_t n, uint64_t* h_data, bool inverse, int exchange) {
    SlabGroup* g = nullptr;
    MG_TRY(slab_group(devices, ndev, n, &g));
    std::lock_guard<std::mutex> lk(g->mu);
    SlabDrain drain(g);   // every return below waits for the lanes' streams first: h_data is the caller's memory
    MG_TRY(slab_host_buffers(g));
    const size_t G = g->lanes.size(), m1 = g->m1, s1 = g->s1, w = s1 / G, r = m1 / G, per = (size_t)n / G;
    std::vector<uint32_t*> slabs(G), rows(G);
    for (size_t a = 0; a < G; ++a) { slabs[a] = g->lanes[a].d_slab; rows[a] = g->lanes[a].d_rows; }
    for (size_t a = 0; a < G; ++a) {
        SlabLane& L = g->lanes[a];
        DeviceGuard guard(L.device);
        if (!inverse) {
            // slab [M1][w]: element (j1, c) = x[j1 S1 + a w + c]
            MG_TRY(hipMemcpy2DAsync(L.d_stage, w * sizeof(uint64_t), h_data + a * w, s1 * sizeof(uint64_t), w * sizeof(uint64_t), m1,
                                    hipMemcpyHostToDevice, L.stream));
            hipLaunchKernelGGL(narrow_kernel, dim3(grid_for(per)), dim3(256), 0, L.stream, (const uint64_t*)L.d_stage, L.d_slab, per);
        } else {
            // rows [r][S1]: element (rr, k') = X[(a r + rr) + M1 k'] -> upload [S1][r] runs, transpose on the device
            MG_TRY(hipMemcpy2DAsync(L.d_stage, r * sizeof(uint64_t), h_data + a * r, m1 * sizeof(uint64_t), r * sizeof(uint64_t), s1,
                                    hipMemcpyHostToDevice, L.stream));
            hipLaunchKernelGGL((transpose_convert_kernel<uint64_t, uint32_t>), dim3(grid_for(per, 1024)), dim3(256), 0, L.stream,
                               (const uint64_t*)L.d_stage, L.d_rows, (uint32_t)s1, (uint32_t)r);
        }
        MG_TRY(hipGetLastError());
    }
    MG_TRY(slab_run(g, slabs.data(), rows.data(), inverse, exchange));
    for (size_t a = 0; a < G; ++a) {
        SlabLane& L = g->lanes[a];
        DeviceGuard guard(L.device);
        if (!inverse) {
            hipLaunchKernelGGL((transpose_convert_kernel<uint32_t, uint64_t>), dim3(grid_for(per, 1024)), dim3(256), 0, L.stream,
                               (const uint32_t*)L.d_rows, L.d_stage, (uint32_t)r, (uint32_t)s1);
            MG_TRY(hipGetLastError());
            MG_TRY(hipMemcpy2DAsync(h_data + a * r, m1 * sizeof(uint64_t), L.d_stage, r * sizeof(uint64_t), r * sizeof(uint64_t), s1,
                                    hipMemcpyDeviceToHost, L.stream));
        } else {
            hipLaunchKernelGGL(widen_kernel, dim3(grid_for(per)), dim3(256), 0, L.stream, (const uint32_t*)L.d_slab, L.d_stage, per);
            MG_TRY(hipGetLastError());
            MG_TRY(hipMemcpy2DAsync(h_data + a * w, s1 * sizeof(uint64_t), L.d_stage, w * sizeof(uint64_t), w * sizeof(uint64_t), m1,
                                    hipMemcpyDeviceToHost, L.stream));
        }
    }
    return drain.finish();
}

// First use of a device list whose exchange crosses a link (no cross-device copy, event wait or RCCL group of more than
// one rank had ever run when this was written): one small transform (n = 2^18, 1 MiB) goes through exactly the code path of the
// real call -- same devices, same exchange kind -- and is compared with the SINGLE-device transform of the same input, forward and
// back.  A wrong block order, a copy that raced its producer or a mis-declared RCCL signature shows up here as TOYNI_E_SELF_CHECK
// instead of as a silently wrong proof.  Costs a few milliseconds once per (device list, exchange) and process.
// The probe must itself be a valid slab layout for the lane count (S1 / lanes >= 32 columns per lane, M1 >= lanes rows): 2^18 = 512 x
// 512 serves up to 16 lanes, 2^22 (M1 = 128, S1 = 2^15) up to 128 (a fixed 2^18 probe answered E_RANGE for more than 16
// lanes although the caller's own n was fine).  Only a VERDICT is cached -- ok, or a failed comparison; any other status (out of
// memory, an RCCL start-up hiccup) is returned and the check runs again at the next call.
int slab_self_check(const int* devices, int ndev, int exchange) {
    const uint32_t SELF_CHECK_N = ndev <= 16 ? 1u << 18 : 1u << 22;
    if (ndev < 2) return TOYNI_OK;
    bool distinct = false;
    for (int a = 0; a < ndev; ++a) for (int b = a + 1; b < ndev; ++b) distinct |= devices[a] != devices[b];
    if (!distinct && !(injected() & INJECT_FORCE_SELF_CHECK)) return TOYNI_OK;
    static std::mutex mu;
    static std::map<std::pair<std::vector<int>, int>, int> verdicts;  // process lifetime
    std::lock_guard<std::mutex> lk(mu);
    const std::pair<std::vector<int>, int> key(std::vector<int>(devices, devices + ndev), exchange);
    const auto it = verdicts.find(key);
    if (it != verdicts.end() && !(injected() & INJECT_FORCE_SELF_CHECK)) return it->second;
    const size_t n = SELF_CHECK_N;
    std::vector<uint64_t> x(n), ref(n), got(n);
    uint64_t state = 0x70796E69u;
    for (size_t i = 0; i < n; ++i) {  // splitmix64 residues: every block of the exchange carries different words
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        x[i] = (z ^ (z >> 31)) % BB_P;
    }
    ref = x;
    got = x;
    int rc = TOYNI_OK;
    toyni_ntt_ctx* single = nullptr;
    if ((rc = cached_ctx(devices[0], 0, SELF_CHECK_N, &single)) == TOYNI_OK) rc = toyni_ntt_host(single, ref.data(), 1, 0);
    if (rc == TOYNI_OK) rc = slab_host_transform(devices, ndev, SELF_CHECK_N, got.data(), false, exchange);
    if (rc == TOYNI_OK && (injected() & INJECT_CORRUPT)) got[n / 2 + 5] ^= 1u;
    if (rc == TOYNI_OK && std::memcmp(ref.data(), got.data(), n * sizeof(uint64_t)) != 0) rc = TOYNI_E_SELF_CHECK;
    if (rc == TOYNI_OK) rc = slab_host_transform(devices, ndev, SELF_CHECK_N, got.data(), true, exchange);
    if (rc == TOYNI_OK && std::memcmp(x.data(), got.data(), n * sizeof(uint64_t)) != 0) rc = TOYNI_E_SELF_CHECK;
    if (verbose() || rc != TOYNI_OK)
        std::fprintf(stderr, "toyni_hip: multi-device self-check (%d lanes, %s exchange, n = 2^%d against device %d alone): %s\n", ndev,
                     exchange == TOYNI_EXCHANGE_RCCL ? "RCCL" : "peer-copy", ilog2(SELF_CHECK_N), devices[0], rc == TOYNI_OK ? "ok" : toyni_error_string(rc));
    if (rc == TOYNI_OK || rc == TOYNI_E_SELF_CHECK) verdicts[key] = rc;
    return rc;
}

}  // namespace

extern "C" {

// Batched host-slice transform over several GPUs from one host process: contiguous shards, one host thread per entry,
// cached contexts (the first call per (device, lane, n) builds them), no exchange.
int toyni_ntt_host_multi_gpu(const int* devices, int ndev, uint32_t n, uint64_t* h_data, size_t batch, int inverse) {
    if (!devices || !h_data) return TOYNI_E_NULL;
    if (ndev < 1) return TOYNI_E_RANGE;
    if (!is_pow2(n) || ilog2(n) > MAX_LOG_N) return TOYNI_E_INVALID_SIZE;
    std::vector<int> status((size_t)ndev, TOYNI_OK);
    std::vector<std::thread> workers;
    const size_t base = batch / (size_t)ndev, rem = batch % (size_t)ndev;
    size_t start = 0;
    for (int d = 0; d < ndev; ++d) {
        const size_t count = base + ((size_t)d < rem ? 1 : 0);
        const size_t first = start;
        start += count;
        if (count == 0) continue;
        workers.emplace_back([&, d, first, count]() {
            toyni_ntt_ctx* ctx = nullptr;
            int rc = cached_ctx(devices[d], lane_of(devices, d), n, &ctx);
            if (rc == TOYNI_OK) rc = toyni_ntt_host(ctx, h_data + first * (size_t)n, count, inverse);
            status[(size_t)d] = rc;
        });
    }
    for (auto& w : workers) w.join();
    for (int rc : status) if (rc != TOYNI_OK) return rc;
    return TOYNI_OK;
}

int toyni_ntt_slab_multi_gpu_device(const int* devices, int ndev, uint32_t n, uint32_t* const* d_slabs, uint32_t* const* d_rows,
                                    int inverse, int exchange) {
    if (!devices || !d_slabs || !d_rows) return TOYNI_E_NULL;
    if (ndev < 1 || (ndev & (ndev - 1))) return TOYNI_E_RANGE;
    if (!is_pow2(n) || ilog2(n) > MAX_LOG_N) return TOYNI_E_INVALID_SIZE;
    if (exchange != TOYNI_EXCHANGE_PEER_COPY && exchange != TOYNI_EXCHANGE_RCCL) return TOYNI_E_RANGE;
    for (int d = 0; d < ndev; ++d) if (!d_slabs[d] || !d_rows[d]) return TOYNI_E_NULL;
    MG_TRY(slab_self_check(devices, ndev, exchange));
    SlabGroup* g = nullptr;
    MG_TRY(slab_group(devices, ndev, n, &g));
    std::lock_guard<std::mutex> lk(g->mu);
    SlabDrain drain(g);
    const int rc = slab_run(g, d_slabs, d_rows, inverse != 0, exchange);
    const int rs = drain.finish();
    return rc ? rc : rs;
}

// Host slice in natural order, in place (slab_host_transform above), after the group's first-use self-check.
int toyni_ntt_slab_multi_gpu_host(const int* devices, int ndev, uint32_t n, uint64_t* h_data, int inverse, int exchange) {
    if (!devices || !h_data) return TOYNI_E_NULL;
    if (ndev < 1 || (ndev & (ndev - 1))) return TOYNI_E_RANGE;
    if (!is_pow2(n) || ilog2(n) > MAX_LOG_N) return TOYNI_E_INVALID_SIZE;
    if (exchange != TOYNI_EXCHANGE_PEER_COPY && exchange != TOYNI_EXCHANGE_RCCL) return TOYNI_E_RANGE;
    MG_TRY(slab_self_check(devices, ndev, exchange));
    return slab_host_transform(devices, ndev, n, h_data, inverse != 0, exchange);
}

#ifdef TOYNI_TOOLS  // include/toyni_hip_tools.h
int toyni_tools_inject(unsigned flags) { g_inject.store(flags, std::memory_order_relaxed); return TOYNI_OK; }
int toyni_tools_rccl_version(void) { return rccl().version; }

int toyni_tools_slab_phases(int enable) {
    SlabPhaseRec& pr = slab_phase_rec();
    std::lock_guard<std::mutex> lk(pr.mu);
    pr.on = enable != 0;
    return TOYNI_OK;
}
// ms[0..3] = slab pass / exchange / relayout / row transforms, summed over the transforms recorded since the last read (forward and
// inverse alike: the inverse runs the same four stages in the opposite order); *transforms = how many.  Waits for the events.
int toyni_tools_slab_phases_read(float* ms, unsigned* transforms) {
    if (!ms || !transforms) return TOYNI_E_NULL;
    SlabPhaseRec& pr = slab_phase_rec();
    std::lock_guard<std::mutex> lk(pr.mu);
    for (int k = 0; k < 4; ++k) ms[k] = 0.f;
    *transforms = 0;
    DeviceGuard guard(pr.device);
    hipError_t err = hipSuccess;
    for (auto& r : pr.recs) {
        hipError_t e = hipEventSynchronize(r.e[4]);
        float d[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventElapsedTime(&d[k], r.e[k], r.e[k + 1]);
        if (e == hipSuccess) {
            // forward: pass, exchange, relayout, rows; inverse: rows, relayout, exchange, pass
            for (int k = 0; k < 4; ++k) ms[r.inverse ? 3 - k : k] += d[k];
            *transforms += 1u;
        } else if (err == hipSuccess) err = e;
        for (int k = 0; k < 5; ++k) if (r.e[k]) (void)hipEventDestroy(r.e[k]);
    }
    pr.recs.clear();
    return (int)err;
}
#endif

}  // extern "C"
#undef MG_TRY
