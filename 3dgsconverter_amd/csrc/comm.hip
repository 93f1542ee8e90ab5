// comm.hip -- the collectives of the multi-GPU paths (gsx_comm_*), two transports behind one C ABI.
//
// The reference is a single process; its SOR treats queries as independent units over one
// reference set (data_processor.py:167-173) and thresholds on numpy's f32 mean / std of the whole mean-distance array
// (:176-180).  Scale-out on one MI355X node is one process per GPU; every exchange of the data path goes through the
// entry points below, on the context's stream, called from the C library (no collective inside kernels):
//
//   * "rccl"     -- the product: librccl is dlopen'ed at first use (single-GPU users never need it) and the calls are
//                   the plain collectives over xGMI: ncclAllReduce / ncclAllGather / grouped ncclSend + ncclRecv.
//   * "hostwire" -- N processes that SHARE GPUs (RCCL refuses two ranks on one device): every rank owns an outbox
//                   in POSIX shared memory, a collective = device -> outbox, barrier, peers' outboxes -> device,
//                   barrier.  Synchronous and slow by design; it exists so that the full N-rank choreography -- the
//                   same C and Python code, only the wire differs -- runs on the one-GPU test box
//                   (tests/test_dist_gpu.py, `bench.py --gpus 2` there).  Selected by the unique id: ids made under
//                   GSX_COMM_TRANSPORT=hostwire carry a magic prefix.
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>

#include "gsx_common.h"

namespace gsx {

// ---------------------------------------------------------------- RCCL through dlopen
typedef struct { char internal[128]; } rcclUniqueId;
typedef void *rcclComm_t;
enum { RCCL_INT8 = 0, RCCL_INT64 = 4, RCCL_FLOAT32 = 7, RCCL_FLOAT64 = 8 };  // ncclDataType_t
enum { RCCL_SUM = 0, RCCL_MAX = 2, RCCL_MIN = 3 };                           // ncclRedOp_t

struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(rcclUniqueId *) = nullptr;
    int (*CommInitRank)(rcclComm_t *, int, rcclUniqueId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*CommAbort)(rcclComm_t) = nullptr;    // optional
    const char *(*GetErrorString)(int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, rcclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};
static Rccl g_rccl;

static int rccl_load()
{
    if (g_rccl.h) return 0;
    // an already loaded librccl (torch bundles one under the same SONAME family) is reused by dlopen
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) GSX_FAIL("gsx_comm: cannot load librccl (%s)", dlerror());
#define GSX_SYM(field, name)                                                          \
    *reinterpret_cast<void **>(&g_rccl.field) = dlsym(h, name);                       \
    if (!g_rccl.field) GSX_FAIL("gsx_comm: librccl has no symbol %s", name)
    GSX_SYM(GetUniqueId, "ncclGetUniqueId");
    GSX_SYM(CommInitRank, "ncclCommInitRank");
    GSX_SYM(CommDestroy, "ncclCommDestroy");
    GSX_SYM(GetErrorString, "ncclGetErrorString");
    GSX_SYM(AllReduce, "ncclAllReduce");
    GSX_SYM(AllGather, "ncclAllGather");
    GSX_SYM(Send, "ncclSend");
    GSX_SYM(Recv, "ncclRecv");
    GSX_SYM(GroupStart, "ncclGroupStart");
    GSX_SYM(GroupEnd, "ncclGroupEnd");
#undef GSX_SYM
    *reinterpret_cast<void **>(&g_rccl.CommAbort) = dlsym(h, "ncclCommAbort");
    g_rccl.h = h;
    return 0;
}

#define GSX_RCCL(call)                                                                                  \
    do {                                                                                                \
        int r__ = (call);                                                                               \
        if (r__ != 0) GSX_FAIL("%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r__) : "?"); \
    } while (0)

// ---------------------------------------------------------------- hostwire: shared-memory outboxes
constexpr int COMM_MAX_RANKS = 16;
constexpr int COMM_MAX_SEGS = 2;
static const char HW_MAGIC[8] = {'G', 'S', 'X', 'H', 'W', '0', '1', 0};

struct HwControl {                          // one per job, in /dev/shm
    std::atomic<uint32_t> ready;            // set by rank 0 once the block is initialised
    std::atomic<uint32_t> arrive;           // barrier: arrivals of the current round
    std::atomic<uint32_t> sense;            // barrier: flips when a round completes
    std::atomic<uint32_t> failed;           // a rank gave up: everybody else stops waiting
    uint32_t world;
    uint64_t outbox_bytes[COMM_MAX_RANKS];  // current size of every rank's outbox file
    // all-to-all descriptors: where rank src put the block for rank dst (bytes, inside src's outbox)
    int64_t blk_off[COMM_MAX_RANKS][COMM_MAX_RANKS][COMM_MAX_SEGS];
    int64_t blk_len[COMM_MAX_RANKS][COMM_MAX_RANKS][COMM_MAX_SEGS];
};

struct HostWire {
    char token[40] = {0};
    HwControl *ctl = nullptr;
    uint32_t my_sense = 0;
    int fd[COMM_MAX_RANKS];
    char *map[COMM_MAX_RANKS];
    size_t mapped[COMM_MAX_RANKS];
    double timeout_s = 300.0;
    HostWire()
    {
        for (int r = 0; r < COMM_MAX_RANKS; ++r) {
            fd[r] = -1;
            map[r] = nullptr;
            mapped[r] = 0;
        }
    }
};

static double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void hw_name(const HostWire &w, int which /* -1: control block */, char *out, size_t cap)
{
    if (which < 0) snprintf(out, cap, "/gsx_hw_%s_ctl", w.token);
    else snprintf(out, cap, "/gsx_hw_%s_%d", w.token, which);
}

static int hw_barrier(HostWire &w, int world)
{
    HwControl *c = w.ctl;
    w.my_sense ^= 1u;
    if (c->arrive.fetch_add(1u, std::memory_order_acq_rel) == (uint32_t)world - 1u) {
        c->arrive.store(0u, std::memory_order_relaxed);
        c->sense.store(w.my_sense, std::memory_order_release);
        return 0;
    }
    const double t0 = now_s();
    unsigned spins = 0;
    while (c->sense.load(std::memory_order_acquire) != w.my_sense) {
        if (c->failed.load(std::memory_order_relaxed)) GSX_FAIL("gsx_comm (hostwire): a peer rank failed");
        if (++spins > 2000) {
            sched_yield();
            if ((spins & 1023u) == 0 && now_s() - t0 > w.timeout_s) {
                c->failed.store(1u);
                GSX_FAIL("gsx_comm (hostwire): barrier timed out after %.0f s (a peer rank died?)", w.timeout_s);
            }
        }
    }
    return 0;
}

// my outbox holds at least `bytes`
static int hw_reserve(HostWire &w, int me, size_t bytes)
{
    if (bytes <= w.mapped[me]) return 0;
    size_t cap = std::max<size_t>(bytes + bytes / 8, 1 << 20);
    cap = (cap + 4095) & ~(size_t)4095;
    if (ftruncate(w.fd[me], (off_t)cap) != 0) GSX_FAIL("gsx_comm (hostwire): cannot grow the outbox to %zu bytes (/dev/shm full?)", cap);
    if (w.map[me]) munmap(w.map[me], w.mapped[me]);
    void *p = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_SHARED, w.fd[me], 0);
    if (p == MAP_FAILED) {
        w.map[me] = nullptr;
        w.mapped[me] = 0;
        GSX_FAIL("gsx_comm (hostwire): mmap of %zu bytes failed", cap);
    }
    w.map[me] = static_cast<char *>(p);
    w.mapped[me] = cap;
    w.ctl->outbox_bytes[me] = cap;
    return 0;
}

// peer's outbox as it is after the barrier that published it
static int hw_peer(HostWire &w, int peer, size_t need, const char **out)
{
    const size_t cur = (size_t)w.ctl->outbox_bytes[peer];
    if (need > cur) GSX_FAIL("gsx_comm (hostwire): rank %d published %zu bytes, %zu needed", peer, cur, need);
    if (w.mapped[peer] < cur) {
        if (w.map[peer]) munmap(w.map[peer], w.mapped[peer]);
        w.map[peer] = nullptr;
        w.mapped[peer] = 0;
        if (w.fd[peer] < 0) {
            char name[96];
            hw_name(w, peer, name, sizeof(name));
            w.fd[peer] = shm_open(name, O_RDONLY, 0600);
            if (w.fd[peer] < 0) GSX_FAIL("gsx_comm (hostwire): cannot open the outbox of rank %d", peer);
        }
        void *p = mmap(nullptr, cur, PROT_READ, MAP_SHARED, w.fd[peer], 0);
        if (p == MAP_FAILED) GSX_FAIL("gsx_comm (hostwire): mmap of rank %d's outbox failed", peer);
        w.map[peer] = static_cast<char *>(p);
        w.mapped[peer] = cur;
    }
    *out = w.map[peer];
    return 0;
}

}  // namespace gsx

using namespace gsx;

struct gsx_comm {
    int rank = 0, world = 1;
    rcclComm_t comm = nullptr;   // transport "rccl"
    HostWire *hw = nullptr;      // transport "hostwire"
    bool aborted = false;        // gsx_comm_abort ran: every later collective fails instead of touching a dead communicator
    bool self_wire = false;      // GSX_COMM_SELF_WIRE=1 (test knob): a rank's block for ITSELF also travels as ncclSend + ncclRecv in
                                 // the group instead of a device copy -- the grouped point-to-point calls (data type, counts, offsets,
                                 // stream) run through the real library on a box with one GPU
};

static int hw_destroy(gsx_comm *m)
{
    HostWire *w = m->hw;
    if (!w) return 0;
    char name[96];
    for (int r = 0; r < COMM_MAX_RANKS; ++r) {
        if (w->map[r]) munmap(w->map[r], w->mapped[r]);
        if (w->fd[r] >= 0) close(w->fd[r]);
    }
    hw_name(*w, m->rank, name, sizeof(name));
    shm_unlink(name);
    if (w->ctl) munmap(w->ctl, sizeof(HwControl));
    if (m->rank == 0) {   // (peers that have not attached yet fail loudly; the others hold their mapping)
        hw_name(*w, -1, name, sizeof(name));
        shm_unlink(name);
    }
    delete w;
    m->hw = nullptr;
    return 0;
}

static int hw_init(gsx_comm *m, const char *id128)
{
    HostWire *w = new HostWire();
    m->hw = w;
    memcpy(w->token, id128 + 8, 32);
    w->token[32] = 0;
    if (const char *t = getenv("GSX_HOSTWIRE_TIMEOUT_S")) w->timeout_s = std::max(1.0, atof(t));
    char name[96];
    hw_name(*w, -1, name, sizeof(name));
    int fd = -1;
    if (m->rank == 0) {
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) GSX_FAIL("gsx_comm (hostwire): cannot create %s", name);
        if (ftruncate(fd, sizeof(HwControl)) != 0) GSX_FAIL("gsx_comm (hostwire): ftruncate failed");
    } else {
        const double t0 = now_s();
        while ((fd = shm_open(name, O_RDWR, 0600)) < 0) {
            if (now_s() - t0 > w->timeout_s) GSX_FAIL("gsx_comm (hostwire): rank 0 never created %s", name);
            usleep(1000);
        }
        struct stat st;
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < sizeof(HwControl)) {
            if (now_s() - t0 > w->timeout_s) GSX_FAIL("gsx_comm (hostwire): control block never sized");
            usleep(1000);
        }
    }
    void *p = mmap(nullptr, sizeof(HwControl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) GSX_FAIL("gsx_comm (hostwire): mmap of the control block failed");
    w->ctl = static_cast<HwControl *>(p);
    if (m->rank == 0) {
        // (a fresh shm object is zero-filled: arrive = sense = failed = 0)
        w->ctl->world = (uint32_t)m->world;
        w->ctl->ready.store(1u, std::memory_order_release);
    } else {
        const double t0 = now_s();
        while (!w->ctl->ready.load(std::memory_order_acquire)) {
            if (now_s() - t0 > w->timeout_s) GSX_FAIL("gsx_comm (hostwire): rank 0 never initialised the control block");
            usleep(1000);
        }
        if (w->ctl->world != (uint32_t)m->world) GSX_FAIL("gsx_comm (hostwire): world size mismatch (%u vs %d)", w->ctl->world, m->world);
    }
    hw_name(*w, m->rank, name, sizeof(name));
    w->fd[m->rank] = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (w->fd[m->rank] < 0) GSX_FAIL("gsx_comm (hostwire): cannot create %s", name);
    GSX_CHECK(hw_reserve(*w, m->rank, 1 << 20));
    return hw_barrier(*w, m->world);   // every outbox exists
}

// one host-side reduction, the same order on every rank (so every rank holds identical bits)
template <class T, class Op>
static void hw_reduce(T *acc, const T *src, int64_t n, Op op)
{
    for (int64_t i = 0; i < n; ++i) acc[i] = op(acc[i], src[i]);
}

// A hostwire collective that fails LOCALLY (outbox cannot grow, a copy or an mmap fails) must not leave the peers spinning in
// hw_barrier until the timeout: whatever the reason, a non-zero return raises the shared `failed` flag.
static int hw_guard(gsx_comm *m, int rc)
{
    if (rc != 0 && m->hw && m->hw->ctl) m->hw->ctl->failed.store(1u);
    return rc;
}

static int hw_all_reduce(gsx_ctx *c, gsx_comm *m, void *buf_dev, int64_t count, int kind)
{
    HostWire &w = *m->hw;
    const size_t eb = (kind == GSX_COMM_F32_MAX || kind == GSX_COMM_F32_SUM) ? 4 : 8;
    const size_t bytes = eb * (size_t)count;
    GSX_HIP(hipStreamSynchronize(c->stream));
    GSX_CHECK(hw_reserve(w, m->rank, bytes));
    GSX_HIP(hipMemcpy(w.map[m->rank], buf_dev, bytes, hipMemcpyDeviceToHost));
    GSX_CHECK(hw_barrier(w, m->world));
    std::vector<char> acc(bytes);
    for (int r = 0; r < m->world; ++r) {
        const char *src;
        GSX_CHECK(hw_peer(w, r, bytes, &src));
        if (r == 0) {
            memcpy(acc.data(), src, bytes);
            continue;
        }
        switch (kind) {
        case GSX_COMM_F32_MAX: hw_reduce(reinterpret_cast<float *>(acc.data()), reinterpret_cast<const float *>(src), count, [](float a, float b) { return b > a ? b : a; }); break;
        case GSX_COMM_F32_SUM: hw_reduce(reinterpret_cast<float *>(acc.data()), reinterpret_cast<const float *>(src), count, [](float a, float b) { return a + b; }); break;
        case GSX_COMM_I64_SUM: hw_reduce(reinterpret_cast<int64_t *>(acc.data()), reinterpret_cast<const int64_t *>(src), count, [](int64_t a, int64_t b) { return a + b; }); break;
        case GSX_COMM_F64_MAX: hw_reduce(reinterpret_cast<double *>(acc.data()), reinterpret_cast<const double *>(src), count, [](double a, double b) { return b > a ? b : a; }); break;
        default: hw_reduce(reinterpret_cast<int64_t *>(acc.data()), reinterpret_cast<const int64_t *>(src), count, [](int64_t a, int64_t b) { return b < a ? b : a; }); break;
        }
    }
    GSX_HIP(hipMemcpy(buf_dev, acc.data(), bytes, hipMemcpyHostToDevice));
    return hw_barrier(w, m->world);   // nobody overwrites an outbox that is still being read
}

static int hw_all_gather(gsx_ctx *c, gsx_comm *m, const void *send_dev, void *recv_dev, int64_t bytes)
{
    HostWire &w = *m->hw;
    GSX_HIP(hipStreamSynchronize(c->stream));
    GSX_CHECK(hw_reserve(w, m->rank, (size_t)bytes));
    if (bytes) GSX_HIP(hipMemcpy(w.map[m->rank], send_dev, (size_t)bytes, hipMemcpyDeviceToHost));
    GSX_CHECK(hw_barrier(w, m->world));
    for (int r = 0; r < m->world && bytes; ++r) {
        const char *src;
        GSX_CHECK(hw_peer(w, r, (size_t)bytes, &src));
        GSX_HIP(hipMemcpy(static_cast<char *>(recv_dev) + (size_t)bytes * r, src, (size_t)bytes, hipMemcpyHostToDevice));
    }
    return hw_barrier(w, m->world);
}

static int hw_all_to_all(gsx_ctx *c, gsx_comm *m, const void *send_dev, void *recv_dev, int nseg, const int64_t *send_off,
                         const int64_t *send_cnt, const int64_t *recv_off, const int64_t *recv_cnt, size_t eb)
{
    HostWire &w = *m->hw;
    const int G = m->world, me = m->rank;
    GSX_HIP(hipStreamSynchronize(c->stream));
    size_t total = 0;
    for (int i = 0; i < nseg * G; ++i) total += eb * (size_t)send_cnt[i];
    GSX_CHECK(hw_reserve(w, me, total));
    size_t at = 0;
    for (int s = 0; s < nseg; ++s)
        for (int p = 0; p < G; ++p) {
            const size_t len = eb * (size_t)send_cnt[s * G + p];
            w.ctl->blk_off[me][p][s] = (int64_t)at;
            w.ctl->blk_len[me][p][s] = (int64_t)len;
            if (len) GSX_HIP(hipMemcpy(w.map[me] + at, static_cast<const char *>(send_dev) + eb * (size_t)send_off[s * G + p], len, hipMemcpyDeviceToHost));
            at += len;
        }
    GSX_CHECK(hw_barrier(w, G));
    for (int s = 0; s < nseg; ++s)
        for (int p = 0; p < G; ++p) {
            const size_t len = eb * (size_t)recv_cnt[s * G + p];
            if ((int64_t)len != w.ctl->blk_len[p][me][s])
                GSX_FAIL("gsx_comm (hostwire): rank %d sends %lld bytes to rank %d, which expects %zu", p, (long long)w.ctl->blk_len[p][me][s], me, len);
            if (!len) continue;
            const char *src;
            const size_t off = (size_t)w.ctl->blk_off[p][me][s];
            GSX_CHECK(hw_peer(w, p, off + len, &src));
            GSX_HIP(hipMemcpy(static_cast<char *>(recv_dev) + eb * (size_t)recv_off[s * G + p], src + off, len, hipMemcpyHostToDevice));
        }
    return hw_barrier(w, G);
}

extern "C" {

/* GSX_COMM_TRANSPORT=hostwire: an id for the shared-memory transport (ranks sharing GPUs); default: ncclGetUniqueId */
int gsx_comm_unique_id(void *out128)
{
    if (!out128) GSX_FAIL("gsx_comm_unique_id: null argument");
    const char *t = getenv("GSX_COMM_TRANSPORT");
    if (t && strcmp(t, "hostwire") == 0) {
        char *o = static_cast<char *>(out128);
        memset(o, 0, 128);
        memcpy(o, HW_MAGIC, 8);
        unsigned char rnd[16] = {0};
        int fd = open("/dev/urandom", O_RDONLY);
        if (fd < 0 || read(fd, rnd, sizeof(rnd)) != (ssize_t)sizeof(rnd)) {
            if (fd >= 0) close(fd);
            GSX_FAIL("gsx_comm_unique_id: /dev/urandom unavailable");
        }
        close(fd);
        for (int i = 0; i < 16; ++i) snprintf(o + 8 + 2 * i, 3, "%02x", rnd[i]);
        return 0;
    }
    if (t && *t && strcmp(t, "rccl") != 0) GSX_FAIL("gsx_comm_unique_id: unknown GSX_COMM_TRANSPORT '%s' (rccl | hostwire)", t);
    GSX_CHECK(rccl_load());
    rcclUniqueId id;
    GSX_RCCL(g_rccl.GetUniqueId(&id));
    memcpy(out128, &id, sizeof(id));
    return 0;
}

int gsx_comm_init(gsx_ctx *c, int rank, int world, const void *id128)
{
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) GSX_FAIL("gsx_comm_init: bad arguments");
    if (world > COMM_MAX_RANKS) GSX_FAIL("gsx_comm_init: at most %d ranks (one node)", COMM_MAX_RANKS);
    if (c->comm) GSX_FAIL("gsx_comm_init: the context already has a communicator");
    GSX_HIP(hipSetDevice(c->device));
    gsx_comm *m = new gsx_comm();
    if (const char *t = getenv("GSX_COMM_SELF_WIRE")) m->self_wire = atoi(t) != 0;
    m->rank = rank;
    m->world = world;
    if (memcmp(id128, HW_MAGIC, 8) == 0) {
        if (hw_init(m, static_cast<const char *>(id128)) != 0) {
            if (m->hw && m->hw->ctl) m->hw->ctl->failed.store(1u);
            hw_destroy(m);
            delete m;
            return 1;
        }
    } else {
        if (rccl_load() != 0) {
            delete m;
            return 1;
        }
        rcclUniqueId id;
        memcpy(&id, id128, sizeof(id));
        int r = g_rccl.CommInitRank(&m->comm, world, id, rank);
        if (r != 0) {
            delete m;
            GSX_FAIL("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        }
    }
    c->comm = m;
    return 0;
}

int gsx_comm_destroy(gsx_ctx *c)
{
    if (!c || !c->comm) return 0;
    gsx_comm *m = static_cast<gsx_comm *>(c->comm);
    if (m->comm) (void)g_rccl.CommDestroy(m->comm);
    hw_destroy(m);
    delete m;
    c->comm = nullptr;
    return 0;
}

/* a rank that gives up tells the others (hostwire: their barriers fail at once instead of timing out) */
int gsx_comm_abort(gsx_ctx *c)
{
    if (!c || !c->comm) return 0;
    gsx_comm *m = static_cast<gsx_comm *>(c->comm);
    if (m->hw && m->hw->ctl) m->hw->ctl->failed.store(1u);
    if (!m->hw && m->comm && g_rccl.CommAbort && !m->aborted) {   // frees this rank's resources, fails its queued operations
        g_rccl.CommAbort(m->comm);
        m->comm = nullptr;
    }
    m->aborted = true;
    return 0;
}

int gsx_comm_rank(gsx_ctx *c, int *rank_out, int *world_out)
{
    if (!c) GSX_FAIL("null ctx");
    gsx_comm *m = static_cast<gsx_comm *>(c->comm);
    if (rank_out) *rank_out = m ? m->rank : 0;
    if (world_out) *world_out = m ? m->world : 1;
    return 0;
}

/* 0 = no communicator, 1 = rccl, 2 = hostwire */
int gsx_comm_transport(gsx_ctx *c)
{
    if (!c || !c->comm) return 0;
    return static_cast<gsx_comm *>(c->comm)->hw ? 2 : 1;
}

static int dtype_of(int elem_bytes, int *out)
{
    if (elem_bytes == 1) { *out = RCCL_INT8; return 0; }
    if (elem_bytes == 4) { *out = RCCL_FLOAT32; return 0; }
    if (elem_bytes == 8) { *out = RCCL_INT64; return 0; }
    GSX_FAIL("gsx_comm: element size %d not supported", elem_bytes);
}

/* in place; kind: GSX_COMM_F32_MAX / F32_SUM / I64_SUM / F64_MAX / I64_MIN */
int gsx_comm_all_reduce(gsx_ctx *c, void *buf_dev, int64_t count, int kind)
{
    if (!c || !c->comm || !buf_dev || count < 0) GSX_FAIL("gsx_comm_all_reduce: no communicator / null buffer");
    if (kind < GSX_COMM_F32_MAX || kind > GSX_COMM_I64_MIN) GSX_FAIL("gsx_comm_all_reduce: unknown kind %d", kind);
    gsx_comm *m = static_cast<gsx_comm *>(c->comm);
    if (m->aborted) GSX_FAIL("gsx_comm: the communicator was aborted");
    GSX_HIP(hipSetDevice(c->device));
    if (count == 0) return 0;
    if (m->hw) return hw_guard(m, hw_all_reduce(c, m, buf_dev, count, kind));
    static const int dts[5] = {RCCL_FLOAT32, RCCL_FLOAT32, RCCL_INT64, RCCL_FLOAT64, RCCL_INT64};
    static const int ops[5] = {RCCL_MAX, RCCL_SUM, RCCL_SUM, RCCL_MAX, RCCL_MIN};
    GSX_RCCL(g_rccl.AllReduce(buf_dev, buf_dev, (size_t)count, dts[kind], ops[kind], m->comm, c->stream));
    return 0;
}

int gsx_comm_all_gather(gsx_ctx *c, const void *send_dev, void *recv_dev, int64_t bytes_per_rank)
{
    if (!c || !c->comm || !send_dev || !recv_dev || bytes_per_rank < 0) GSX_FAIL("gsx_comm_all_gather: no communicator / null buffer");
    gsx_comm *m = static_cast<gsx_comm *>(c->comm);
    if (m->aborted) GSX_FAIL("gsx_comm: the communicator was aborted");
    GSX_HIP(hipSetDevice(c->device));
    if (m->hw) return hw_guard(m, hw_all_gather(c, m, send_dev, recv_dev, bytes_per_rank));
    GSX_RCCL(g_rccl.AllGather(send_dev, recv_dev, (size_t)bytes_per_rank, RCCL_INT8, m->comm, c->stream));
    return 0;
}

/* nseg <= 2 independent exchanges in ONE group of ncclSend / ncclRecv (e.g. own rows and halo rows of the slab
 * partition): entry [s * world + p] of the four host arrays = segment s to / from peer p, in ELEMENTS of elem_bytes
 * (1, 4, 8 or 12 = a row of three floats); the local block is a device copy */
int gsx_comm_all_to_all_segs(gsx_ctx *c, const void *send_dev, void *recv_dev, int nseg, const int64_t *send_off,
                             const int64_t *send_cnt, const int64_t *recv_off, const int64_t *recv_cnt, int elem_bytes)
{
    if (!c || !c->comm || !send_off || !send_cnt || !recv_off || !recv_cnt || nseg < 1 || nseg > COMM_MAX_SEGS)
        GSX_FAIL("gsx_comm_all_to_all: no communicator / null argument / more than %d segments", COMM_MAX_SEGS);
    gsx_comm *m = static_cast<gsx_comm *>(c->comm);
    if (m->aborted) GSX_FAIL("gsx_comm: the communicator was aborted");
    GSX_HIP(hipSetDevice(c->device));
    const int G = m->world, me = m->rank;
    int dt;
    GSX_CHECK(dtype_of(elem_bytes == 12 ? 4 : elem_bytes, &dt));
    const size_t mul = elem_bytes == 12 ? 3 : 1;   // rows of 3 floats travel as floats
    const size_t eb = elem_bytes;
    for (int i = 0; i < nseg * G; ++i)
        if (send_cnt[i] < 0 || recv_cnt[i] < 0 || send_off[i] < 0 || recv_off[i] < 0) GSX_FAIL("gsx_comm_all_to_all: negative size");
    for (int s = 0; s < nseg; ++s)
        if (send_cnt[s * G + me] != recv_cnt[s * G + me]) GSX_FAIL("gsx_comm_all_to_all: local block sizes differ");
    if (m->hw) return hw_guard(m, hw_all_to_all(c, m, send_dev, recv_dev, nseg, send_off, send_cnt, recv_off, recv_cnt, eb));
    const bool self_wire = m->self_wire;
    bool remote = false;
    for (int i = 0; i < nseg * G; ++i) remote |= ((i % G) != me || self_wire) && (send_cnt[i] > 0 || recv_cnt[i] > 0);
    if (remote) {   // (an empty group still costs RCCL bookkeeping)
        GSX_RCCL(g_rccl.GroupStart());
        for (int s = 0; s < nseg; ++s)
            for (int p = 0; p < G; ++p) {
                if (p == me && !self_wire) continue;
                const int i = s * G + p;
                if (send_cnt[i] > 0)
                    GSX_RCCL(g_rccl.Send(static_cast<const char *>(send_dev) + eb * (size_t)send_off[i], (size_t)send_cnt[i] * mul, dt, p,
                                         m->comm, c->stream));
                if (recv_cnt[i] > 0)
                    GSX_RCCL(g_rccl.Recv(static_cast<char *>(recv_dev) + eb * (size_t)recv_off[i], (size_t)recv_cnt[i] * mul, dt, p,
                                         m->comm, c->stream));
            }
        GSX_RCCL(g_rccl.GroupEnd());
    }
    for (int s = 0; s < nseg && !self_wire; ++s) {
        const int i = s * G + me;
        if (send_cnt[i] > 0)
            GSX_HIP(hipMemcpyAsync(static_cast<char *>(recv_dev) + eb * (size_t)recv_off[i],
                                   static_cast<const char *>(send_dev) + eb * (size_t)send_off[i], eb * (size_t)send_cnt[i],
                                   hipMemcpyDeviceToDevice, c->stream));
    }
    return 0;
}

int gsx_comm_all_to_all_v(gsx_ctx *c, const void *send_dev, const int64_t *send_off, const int64_t *send_cnt, void *recv_dev,
                          const int64_t *recv_off, const int64_t *recv_cnt, int elem_bytes)
{
    return gsx_comm_all_to_all_segs(c, send_dev, recv_dev, 1, send_off, send_cnt, recv_off, recv_cnt, elem_bytes);
}

/* every rank's stream work up to here has completed on every rank when this returns (the launcher's barrier) */
int gsx_comm_barrier(gsx_ctx *c)
{
    if (!c || !c->comm) GSX_FAIL("gsx_comm_barrier: no communicator");
    gsx_comm *m = static_cast<gsx_comm *>(c->comm);
    if (m->aborted) GSX_FAIL("gsx_comm: the communicator was aborted");
    GSX_HIP(hipSetDevice(c->device));
    if (m->hw) {
        GSX_HIP(hipStreamSynchronize(c->stream));
        return hw_barrier(*m->hw, m->world);
    }
    GSX_CHECK(c->commscratch.reserve(64));
    GSX_HIP(hipMemsetAsync(c->commscratch.p, 0, 8, c->stream));
    GSX_RCCL(g_rccl.AllReduce(c->commscratch.p, c->commscratch.p, 1, RCCL_INT64, RCCL_SUM, m->comm, c->stream));
    GSX_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
