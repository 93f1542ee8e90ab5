// host_rows.hip -- the two host-side row operations that bracket every filter of the reference's
// DataProcessor, threaded.  No device code in this file.
//
//   data_processor.py:38,139   coords = np.column_stack((v['x'], v['y'], v['z']))   AoS -> (N,3)
//   data_processor.py:114,149  self.data = vertices[mask]                           row compaction
//
// On the splat table (62 x f4 = 248 B per row, structures.py:32-40) numpy does the first as three
// strided single-thread copies and the second as a per-element structured take: measured at 10M
// rows on the MI355X host 111 ms and 1215 ms -- against 6.4 ms for the whole host-to-host GPU
// filter (profiles/r01_e2e_probe.log).  Both are pure data movement; SURVEY.md 8(f) rank 1.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "gsx_common.h"

namespace {

int worker_count(int64_t bytes)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const int64_t by_size = bytes / (8LL << 20) + 1;  // ~8 MiB of traffic per thread at least
    return (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)(hw ? hw : 8), 64, by_size}));
}

template <class F>
void run_threads(int nt, F &&body)  // body(t) for t in [0, nt); the caller is thread 0
{
    std::vector<std::thread> th;
    th.reserve(nt > 1 ? nt - 1 : 0);
    for (int t = 1; t < nt; ++t) th.emplace_back([&body, t] { body(t); });
    body(0);
    for (auto &x : th) x.join();
}

}  // namespace

extern "C" int gsx_host_gather_f32(const void *rows, int64_t row_bytes, int64_t n, const int64_t *offsets, int ncols,
                                   float *out)
{
    if (!rows || !offsets || !out) GSX_FAIL("gsx_host_gather_f32: null argument");
    if (n < 0 || row_bytes <= 0 || ncols < 1 || ncols > 64) GSX_FAIL("gsx_host_gather_f32: bad shape");
    for (int c = 0; c < ncols; ++c)
        if (offsets[c] < 0 || offsets[c] + 4 > row_bytes) GSX_FAIL("gsx_host_gather_f32: column %d outside the row", c);
    const int nt = worker_count(n * (int64_t)(64 + 4 * ncols));
    const char *src = static_cast<const char *>(rows);
    run_threads(nt, [&](int t) {
        const int64_t r0 = n * t / nt, r1 = n * (t + 1) / nt;
        if (ncols == 3) {
            const int64_t o0 = offsets[0], o1 = offsets[1], o2 = offsets[2];
            for (int64_t r = r0; r < r1; ++r) {
                const char *p = src + r * row_bytes;
                float a, b, c;
                memcpy(&a, p + o0, 4);
                memcpy(&b, p + o1, 4);
                memcpy(&c, p + o2, 4);
                out[3 * r] = a;
                out[3 * r + 1] = b;
                out[3 * r + 2] = c;
            }
        } else {
            for (int64_t r = r0; r < r1; ++r) {
                const char *p = src + r * row_bytes;
                for (int c = 0; c < ncols; ++c) memcpy(&out[r * ncols + c], p + offsets[c], 4);
            }
        }
    });
    return 0;
}

// The same gather, COLUMN-major: out[c * n + r] -- what the writers upload (one contiguous float32 column per field: the
// compressed-PLY packers and the SH byte kernel take SoA columns).  Round 5: the compressed-PLY writer gathered its 59 columns
// one numpy strided copy at a time (most of its 460 ms per 2M splats).
extern "C" int gsx_host_gather_columns_f32(const void *rows, int64_t row_bytes, int64_t n, const int64_t *offsets, int ncols,
                                           float *out)
{
    if (!rows || !offsets || !out) GSX_FAIL("gsx_host_gather_columns_f32: null argument");
    if (n < 0 || row_bytes <= 0 || ncols < 1 || ncols > 64) GSX_FAIL("gsx_host_gather_columns_f32: bad shape");
    for (int c = 0; c < ncols; ++c)
        if (offsets[c] < 0 || offsets[c] + 4 > row_bytes) GSX_FAIL("gsx_host_gather_columns_f32: column %d outside the row", c);
    const int nt = worker_count(n * (int64_t)(64 + 4 * ncols));
    const char *src = static_cast<const char *>(rows);
    run_threads(nt, [&](int t) {
        const int64_t r0 = n * t / nt, r1 = n * (t + 1) / nt;
        // blocks of 256 rows: the block of source rows (62 KB for the standard table) stays in L1/L2 while every column is
        // written as one 1 KB run
        for (int64_t b = r0; b < r1; b += 256) {
            const int64_t e = b + 256 < r1 ? b + 256 : r1;
            for (int c = 0; c < ncols; ++c) {
                float *dst = out + (int64_t)c * n;
                const char *p = src + offsets[c];
                for (int64_t r = b; r < e; ++r) memcpy(&dst[r], p + r * row_bytes, 4);
            }
        }
    });
    return 0;
}

extern "C" int gsx_host_compact_rows(const void *rows, int64_t row_bytes, int64_t n, const uint8_t *mask, void *out,
                                     int64_t out_rows, int64_t *n_out)
{
    if (!rows || !mask || !n_out || (!out && out_rows > 0)) GSX_FAIL("gsx_host_compact_rows: null argument");
    if (n < 0 || row_bytes <= 0 || out_rows < 0) GSX_FAIL("gsx_host_compact_rows: bad shape");
    const int nt = worker_count(2 * n * row_bytes);
    std::vector<int64_t> cnt(nt + 1, 0);
    run_threads(nt, [&](int t) {
        const int64_t r0 = n * t / nt, r1 = n * (t + 1) / nt;
        int64_t c = 0;
        for (int64_t r = r0; r < r1; ++r) c += mask[r] != 0;
        cnt[t + 1] = c;
    });
    for (int t = 0; t < nt; ++t) cnt[t + 1] += cnt[t];
    *n_out = cnt[nt];
    if (cnt[nt] > out_rows) GSX_FAIL("gsx_host_compact_rows: %lld survivors do not fit %lld output rows", (long long)cnt[nt],
                                     (long long)out_rows);
    const char *src = static_cast<const char *>(rows);
    char *dst = static_cast<char *>(out);
    {
        // the output is a fresh allocation: ask for huge pages before first touch (a hint; ignored
        // where transparent huge pages are off) -- 4 KiB first-touch faults otherwise dominate: 87 ->
        // 16 ms for 10M x 248 B on the MI355X host.  (MADV_POPULATE_WRITE per thread was worse: 68 ms.)
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(dst) + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(dst) + (size_t)cnt[nt] * (size_t)row_bytes) & ~(uintptr_t)((2u << 20) - 1);
        if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
    }
    run_threads(nt, [&](int t) {
        const int64_t r0 = n * t / nt, r1 = n * (t + 1) / nt;
        char *d = dst + cnt[t] * row_bytes;
        int64_t r = r0;
        while (r < r1) {
            while (r < r1 && !mask[r]) ++r;  // skip a dropped run
            int64_t e = r;
            while (e < r1 && mask[e]) ++e;   // one memcpy per surviving run (order preserved)
            if (e > r) {
                const size_t bytes = (size_t)(e - r) * (size_t)row_bytes;
                memcpy(d, src + r * row_bytes, bytes);
                d += bytes;
            }
            r = e;
        }
    });
    return 0;
}

// rows[idx] for an ASCENDING list of distinct row indices -- what `vertices[mask]` (data_processor.py:114,149) is once the
// device chain has handed back its survivor list: no boolean mask has to be built from the list first (numpy's
// `mask[survivors] = True` on 8M indices cost more than the compaction itself).  Threaded; consecutive indices are copied as
// one run.
extern "C" int gsx_host_take_rows(const void *rows, int64_t row_bytes, int64_t n, const uint32_t *idx, int64_t n_idx, void *out)
{
    if (!rows || (!idx && n_idx > 0) || (!out && n_idx > 0)) GSX_FAIL("gsx_host_take_rows: null argument");
    if (n < 0 || row_bytes <= 0 || n_idx < 0) GSX_FAIL("gsx_host_take_rows: bad shape");
    if (n_idx == 0) return 0;
    const int nt = worker_count(2 * n_idx * row_bytes);
    const char *src = static_cast<const char *>(rows);
    char *dst = static_cast<char *>(out);
    {
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(dst) + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(dst) + (size_t)n_idx * (size_t)row_bytes) & ~(uintptr_t)((2u << 20) - 1);
        if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
    }
    std::vector<int> bad(nt, 0);
    run_threads(nt, [&](int t) {
        const int64_t i0 = n_idx * t / nt, i1 = n_idx * (t + 1) / nt;
        int64_t i = i0;
        while (i < i1) {
            int64_t e = i + 1;
            while (e < i1 && idx[e] == idx[e - 1] + 1u) ++e;   // a run of consecutive rows: one memcpy
            if ((int64_t)idx[e - 1] >= n || (i > 0 && idx[i] <= idx[i - 1])) {
                bad[t] = 1;
                return;
            }
            memcpy(dst + i * row_bytes, src + (int64_t)idx[i] * row_bytes, (size_t)(e - i) * (size_t)row_bytes);
            i = e;
        }
    });
    for (int t = 0; t < nt; ++t)
        if (bad[t]) GSX_FAIL("gsx_host_take_rows: the index list is not strictly ascending inside [0, n)");
    return 0;
}

// data_processor.py:310-313 (cap_sh_degree): self.data[f_rest_i] = 0.0 for the columns above the kept degree -- up to 45
// strided single-thread column fills in numpy; one threaded pass over the rows here.  offsets: byte offsets of the
// 4-byte columns to zero.
extern "C" int gsx_host_zero_columns(void *rows, int64_t row_bytes, int64_t n, const int64_t *offsets, int ncols)
{
    if (!rows || !offsets) GSX_FAIL("gsx_host_zero_columns: null argument");
    if (n < 0 || row_bytes <= 0 || ncols < 0 || ncols > 4096) GSX_FAIL("gsx_host_zero_columns: bad shape");
    for (int c = 0; c < ncols; ++c)
        if (offsets[c] < 0 || offsets[c] + 4 > row_bytes) GSX_FAIL("gsx_host_zero_columns: column %d outside the row", c);
    if (ncols == 0 || n == 0) return 0;
    // contiguous runs of columns become one memset per row
    std::vector<std::pair<int64_t, int64_t>> runs;   // (offset, bytes)
    std::vector<int64_t> off(offsets, offsets + ncols);
    std::sort(off.begin(), off.end());
    for (int64_t o : off) {
        if (!runs.empty() && runs.back().first + runs.back().second == o) runs.back().second += 4;
        else if (runs.empty() || runs.back().first + runs.back().second < o) runs.emplace_back(o, 4);
    }
    const int nt = worker_count(n * row_bytes);
    char *base = static_cast<char *>(rows);
    run_threads(nt, [&](int t) {
        const int64_t r0 = n * t / nt, r1 = n * (t + 1) / nt;
        for (int64_t r = r0; r < r1; ++r)
            for (const auto &ru : runs) memset(base + r * row_bytes + ru.first, 0, (size_t)ru.second);
    });
    return 0;
}

// data_processor.py:264-274 (add_rgb_from_sh): a new structured array = every old field + three u1 fields.  numpy copies
// field by field (62 strided passes over a 10M-row table); here every output row is the old row followed by `extra_bytes`
// bytes of `extra` (row-major n x extra_bytes), one threaded pass.  out_row_bytes >= row_bytes + extra_bytes (numpy's
// itemsize of the widened dtype; any padding in between is zeroed).
extern "C" int gsx_host_append_columns(const void *rows, int64_t row_bytes, int64_t n, const uint8_t *extra, int64_t extra_bytes,
                                       void *out, int64_t out_row_bytes)
{
    if (!rows || !extra || !out) GSX_FAIL("gsx_host_append_columns: null argument");
    if (n < 0 || row_bytes <= 0 || extra_bytes <= 0 || out_row_bytes < row_bytes + extra_bytes) GSX_FAIL("gsx_host_append_columns: bad shape");
    const int nt = worker_count(2 * n * out_row_bytes);
    const char *src = static_cast<const char *>(rows);
    char *dst = static_cast<char *>(out);
    {
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(dst) + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(dst) + (size_t)n * (size_t)out_row_bytes) & ~(uintptr_t)((2u << 20) - 1);
        if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
    }
    const int64_t pad = out_row_bytes - row_bytes - extra_bytes;
    run_threads(nt, [&](int t) {
        const int64_t r0 = n * t / nt, r1 = n * (t + 1) / nt;
        for (int64_t r = r0; r < r1; ++r) {
            char *d = dst + r * out_row_bytes;
            memcpy(d, src + r * row_bytes, (size_t)row_bytes);
            memcpy(d + row_bytes, extra + r * extra_bytes, (size_t)extra_bytes);
            if (pad) memset(d + row_bytes + extra_bytes, 0, (size_t)pad);
        }
    });
    return 0;
}

// Both at once (round 6): the filters' compaction `self.data = vertices[mask]` (data_processor.py:114,149) followed by add_rgb_from_sh's
// widened copy (:262-274) -- what the reference's converter does for every target that needs colours (converter.py:243-252) -- as
// ONE pass: output row j = source row idx[j] followed by the extra bytes of THAT source row (extra: n x extra_bytes, indexed like
// the source table).  idx: strictly ascending row numbers (the device chain's survivor list).
extern "C" int gsx_host_take_rows_shape(const void *rows, int64_t row_bytes, int64_t n, const uint32_t *idx, int64_t n_idx,
                                        const uint8_t *extra, int64_t extra_bytes, const int64_t *zero_offsets, int nzero, void *out,
                                        int64_t out_row_bytes);

extern "C" int gsx_host_take_rows_append(const void *rows, int64_t row_bytes, int64_t n, const uint32_t *idx, int64_t n_idx,
                                         const uint8_t *extra, int64_t extra_bytes, void *out, int64_t out_row_bytes)
{
    if (!extra || extra_bytes <= 0) GSX_FAIL("gsx_host_take_rows_append: null argument");
    return gsx_host_take_rows_shape(rows, row_bytes, n, idx, n_idx, extra, extra_bytes, nullptr, 0, out, out_row_bytes);
}

// ... and cap_sh_degree's column fill (data_processor.py:310-313) in the same pass: zero_offsets = byte offsets of the 4-byte fields
// to zero in every OUTPUT row (nzero may be 0; extra may be null with extra_bytes 0).  The lazy class's whole table shaping --
// compaction, SH cap, colours -- is then one read of the survivors and one write of the new table.
extern "C" int gsx_host_take_rows_shape(const void *rows, int64_t row_bytes, int64_t n, const uint32_t *idx, int64_t n_idx,
                                        const uint8_t *extra, int64_t extra_bytes, const int64_t *zero_offsets, int nzero, void *out,
                                        int64_t out_row_bytes)
{
    if (!rows || (!extra && extra_bytes > 0) || (!idx && n_idx > 0) || (!out && n_idx > 0) || (!zero_offsets && nzero > 0))
        GSX_FAIL("gsx_host_take_rows_shape: null argument");
    if (n < 0 || row_bytes <= 0 || n_idx < 0 || extra_bytes < 0 || out_row_bytes < row_bytes + extra_bytes || nzero < 0 || nzero > 4096)
        GSX_FAIL("gsx_host_take_rows_shape: bad shape");
    for (int c = 0; c < nzero; ++c)
        if (zero_offsets[c] < 0 || zero_offsets[c] + 4 > row_bytes) GSX_FAIL("gsx_host_take_rows_shape: column %d outside the row", c);
    if (n_idx == 0) return 0;
    std::vector<std::pair<int64_t, int64_t>> runs;   // (offset, bytes): contiguous columns become one memset per row
    {
        std::vector<int64_t> off(zero_offsets, zero_offsets + nzero);
        std::sort(off.begin(), off.end());
        for (int64_t o : off) {
            if (!runs.empty() && runs.back().first + runs.back().second == o) runs.back().second += 4;
            else if (runs.empty() || runs.back().first + runs.back().second < o) runs.emplace_back(o, 4);
        }
    }
    const int nt = worker_count(2 * n_idx * out_row_bytes);
    const char *src = static_cast<const char *>(rows);
    char *dst = static_cast<char *>(out);
    {
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(dst) + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(dst) + (size_t)n_idx * (size_t)out_row_bytes) & ~(uintptr_t)((2u << 20) - 1);
        if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
    }
    const int64_t pad = out_row_bytes - row_bytes - extra_bytes;
    std::vector<int> bad(nt, 0);
    run_threads(nt, [&](int t) {
        const int64_t i0 = n_idx * t / nt, i1 = n_idx * (t + 1) / nt;
        for (int64_t i = i0; i < i1; ++i) {
            const int64_t r = idx[i];
            if (r >= n || (i > 0 && idx[i] <= idx[i - 1])) {
                bad[t] = 1;
                return;
            }
            char *d = dst + i * out_row_bytes;
            memcpy(d, src + r * row_bytes, (size_t)row_bytes);
            for (const auto &ru : runs) memset(d + ru.first, 0, (size_t)ru.second);
            if (extra_bytes) memcpy(d + row_bytes, extra + r * extra_bytes, (size_t)extra_bytes);
            if (pad) memset(d + row_bytes + extra_bytes, 0, (size_t)pad);
        }
    });
    for (int t = 0; t < nt; ++t)
        if (bad[t]) GSX_FAIL("gsx_host_take_rows_shape: the index list is not strictly ascending inside [0, n)");
    return 0;
}

// ---- bulk copies between pageable host memory and HBM through pinned staging, threaded (round 6) ------------------------
// hipMemcpy of a PAGEABLE buffer is one runtime thread copying through one staging buffer: measured 24 GB/s up for the 2.48 GB
// splat table of the SOG writer and 14 GB/s down into freshly allocated (not yet faulted) numpy arrays, on a link that moves
// 56 GB/s from pinned memory.  Here T lanes split the transfer into contiguous slices; each lane owns two pinned 8 MiB
// buffers and a stream of its own: while the DMA of one buffer is in flight the lane's host thread fills (or drains) the
// other.  The pinned pool is allocated once per process and kept.
namespace {

constexpr size_t STAGE_CHUNK = 8u << 20;
constexpr int STAGE_LANES_MAX = 12;

struct StageLane {
    void *pin[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
};
struct StagePool {
    int device = -1;
    StageLane lanes[STAGE_LANES_MAX];
    int ready = 0;
} g_stage;
std::atomic_flag g_stage_busy = ATOMIC_FLAG_INIT;

int stage_prepare(int device, int lanes, bool with_buffers = true)
{
    if (g_stage.device != device && g_stage.ready) {   // another GPU: rebuild (one process normally drives one)
        for (int l = 0; l < g_stage.ready; ++l) {
            StageLane &s = g_stage.lanes[l];
            for (int b = 0; b < 2; ++b) {
                if (s.pin[b]) (void)hipHostFree(s.pin[b]);
                if (s.ev[b]) (void)hipEventDestroy(s.ev[b]);
            }
            if (s.stream) (void)hipStreamDestroy(s.stream);
            s = StageLane();
        }
        g_stage.ready = 0;
    }
    g_stage.device = device;
    for (int l = g_stage.ready; l < lanes; ++l) {
        StageLane &s = g_stage.lanes[l];
        for (int b = 0; b < 2; ++b) GSX_HIP(hipEventCreateWithFlags(&s.ev[b], hipEventDisableTiming));
        GSX_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        g_stage.ready = l + 1;
    }
    // the lanes' page-locked 8 MiB buffers only for the paths that copy through them: pinning 2 x 8 MiB per lane costs tens of
    // milliseconds, which the first upload of a process (pinned IN PLACE: register_copy) used to pay for nothing
    if (with_buffers)
        for (int l = 0; l < lanes; ++l)
            for (int b = 0; b < 2; ++b)
                if (!g_stage.lanes[l].pin[b]) GSX_HIP(hipHostMalloc(&g_stage.lanes[l].pin[b], STAGE_CHUNK, hipHostMallocDefault));
    return 0;
}

// Upload by pinning the caller's pages IN PLACE, chunk by chunk, on worker threads that run ahead of the DMA.  The runtime's own
// pageable copy pins a piece, copies it, pins the next: the link idles while the CPU walks page tables (measured 29 GB/s on
// the first copy of a buffer against 56 GB/s once its pages are in the runtime's pinned cache -- which any munmap in the
// process empties, so a writer that allocates and frees result arrays never sees the fast case).  Here lane t registers chunk
// t, t + L, ... (hipHostRegister: get_user_pages + IOMMU map on ITS thread), enqueues the DMA on its own stream and
// unregisters two chunks later; with L lanes the page-table walks of L chunks overlap the DMA of the others.
constexpr size_t REG_CHUNK = 32u << 20;
constexpr int REG_LANES = 6;

int register_copy(gsx_ctx *c, char *dev, char *host, size_t bytes, bool upload)
{
    if (g_stage_busy.test_and_set()) return 1;
    int rc = stage_prepare(c->device, REG_LANES, false);   // (the lanes' streams and events only)
    std::atomic<int> failed{0};
    if (rc == 0) {
        // chunk borders on 2 MiB multiples of the ADDRESS: no page is shared by two chunks (a page registered twice fails)
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(host), a1 = a0 + bytes;
        const uintptr_t first = (a0 + REG_CHUNK) & ~(uintptr_t)((2u << 20) - 1);
        std::vector<uintptr_t> cut;
        cut.push_back(a0);
        for (uintptr_t a = first; a < a1; a += REG_CHUNK) cut.push_back(a);
        cut.push_back(a1);
        const size_t nchunks = cut.size() - 1;
        run_threads(REG_LANES, [&](int t) {
            if (hipSetDevice(c->device) != hipSuccess) {
                failed = 1;
                return;
            }
            StageLane &s = g_stage.lanes[t];
            char *held[2] = {nullptr, nullptr};
            int it = 0;
            for (size_t ch = (size_t)t; ch < nchunks && !failed; ch += REG_LANES, ++it) {
                const int b = it & 1;
                if (held[b]) {   // the DMA that read this slot's chunk has finished: let its pages go
                    if (hipEventSynchronize(s.ev[b]) != hipSuccess) failed = 1;
                    (void)hipHostUnregister(held[b]);
                    held[b] = nullptr;
                }
                char *src = reinterpret_cast<char *>(cut[ch]);
                const size_t len = (size_t)(cut[ch + 1] - cut[ch]);
                if (hipHostRegister(src, len, hipHostRegisterDefault) != hipSuccess) {
                    (void)hipGetLastError();
                    failed = 2;
                    break;
                }
                held[b] = src;
                if ((upload ? hipMemcpyAsync(dev + (cut[ch] - a0), src, len, hipMemcpyHostToDevice, s.stream)
                            : hipMemcpyAsync(src, dev + (cut[ch] - a0), len, hipMemcpyDeviceToHost, s.stream)) != hipSuccess) failed = 1;
                if (hipEventRecord(s.ev[b], s.stream) != hipSuccess) failed = 1;
            }
            if (hipStreamSynchronize(s.stream) != hipSuccess) failed = 1;
            for (int b = 0; b < 2; ++b)
                if (held[b]) (void)hipHostUnregister(held[b]);
        });
    }
    g_stage_busy.clear();
    if (rc != 0) return rc;
    return failed ? 1 : 0;
}

int staged_copy(gsx_ctx *c, char *dev, char *host, size_t bytes, bool upload)
{
    GSX_HIP(hipSetDevice(c->device));
    GSX_HIP(hipStreamSynchronize(c->stream));   // everything the caller enqueued before is done (the lanes use their own streams)
    static const char *mode_env = getenv("GSX_UPLOAD_MODE");   // A/B: "plain", "lanes", "register"; default: register, lanes on failure
    const bool dbg = getenv("GSX_STAGE_DEBUG") != nullptr;
    if (upload && mode_env && !strcmp(mode_env, "plain")) {
        GSX_HIP(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
        return 0;
    }
    if (upload && bytes >= 8 * REG_CHUNK && !(mode_env && !strcmp(mode_env, "lanes"))) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = register_copy(c, dev, host, bytes, true);
        if (dbg) fprintf(stderr, "[gsx] upload of %zu MiB through pinned-in-place chunks: rc %d, %.1f GB/s\n", bytes >> 20, rc,
                         (double)bytes / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 1e9);
        if (rc == 0) return 0;
        // (a range the driver would not pin: the staging lanes below take the whole transfer again)
    }
    // downloads of >= 128 MiB: the DESTINATION pinned in place, chunk by chunk, like the upload's source -- 54 GB/s into a result
    // array whose pages a helper thread touched while the device worked (_lib.prefault), against 26-34 GB/s through the staging
    // lanes (429 + 152 MiB of compressed-PLY elements: 11.4 ms against 18-24).  GSX_DOWNLOAD_MODE=lanes: A/B.
    static const char *dmode_env = getenv("GSX_DOWNLOAD_MODE");
    if (!upload && bytes >= 4 * REG_CHUNK && !(dmode_env && !strcmp(dmode_env, "lanes"))) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = register_copy(c, dev, host, bytes, false);
        if (dbg) fprintf(stderr, "[gsx] download of %zu MiB into pages pinned in place: rc %d, %.1f GB/s\n", bytes >> 20, rc,
                         (double)bytes / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 1e9);
        if (rc == 0) return 0;
    }
    if (bytes < 4 * STAGE_CHUNK || g_stage_busy.test_and_set()) {   // small, or another thread is inside: the plain copy
        GSX_HIP(hipMemcpy(upload ? (void *)dev : (void *)host, upload ? (void *)host : (void *)dev, bytes,
                          upload ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
        return 0;
    }
    if (!upload) {
        // the destination is usually a fresh allocation: ask for huge pages before the lanes first touch it (a hint; 4 KiB
        // first-touch faults otherwise dominate the drain, as in gsx_host_take_rows)
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(host) + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(host) + bytes) & ~(uintptr_t)((2u << 20) - 1);
        if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
    }
    const unsigned hw = std::thread::hardware_concurrency();
    const int lanes = (int)std::max<size_t>(1, std::min<size_t>({(size_t)STAGE_LANES_MAX, (size_t)(hw ? hw : 4), bytes / (2 * STAGE_CHUNK)}));
    int rc = stage_prepare(c->device, lanes);
    std::atomic<int> failed{0};
    if (rc == 0) {
        const size_t nchunks = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
        run_threads(lanes, [&](int t) {
            if (hipSetDevice(c->device) != hipSuccess) {
                failed = 1;
                return;
            }
            StageLane &s = g_stage.lanes[t];
            // lane t takes the chunks t, t + lanes, ...: neighbouring lanes touch neighbouring memory at the same time
            size_t prev_off = 0, prev_len = 0;
            int prev_b = -1, it = 0;
            for (size_t ch = (size_t)t; ch < nchunks; ch += (size_t)lanes, ++it) {
                const int b = it & 1;
                const size_t off = ch * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - off);
                if (upload) {
                    if (it >= 2 && hipEventSynchronize(s.ev[b]) != hipSuccess) failed = 1;   // the DMA that last read this buffer
                    memcpy(s.pin[b], host + off, len);
                    if (hipMemcpyAsync(dev + off, s.pin[b], len, hipMemcpyHostToDevice, s.stream) != hipSuccess) failed = 1;
                    if (hipEventRecord(s.ev[b], s.stream) != hipSuccess) failed = 1;
                } else {
                    if (hipMemcpyAsync(s.pin[b], dev + off, len, hipMemcpyDeviceToHost, s.stream) != hipSuccess) failed = 1;
                    if (hipEventRecord(s.ev[b], s.stream) != hipSuccess) failed = 1;
                    if (prev_b >= 0) {   // drain the previous buffer while this DMA runs
                        if (hipEventSynchronize(s.ev[prev_b]) != hipSuccess) failed = 1;
                        memcpy(host + prev_off, s.pin[prev_b], prev_len);
                    }
                    prev_b = b;
                    prev_off = off;
                    prev_len = len;
                }
            }
            if (!upload && prev_b >= 0) {
                if (hipEventSynchronize(s.ev[prev_b]) != hipSuccess) failed = 1;
                memcpy(host + prev_off, s.pin[prev_b], prev_len);
            }
            if (hipStreamSynchronize(s.stream) != hipSuccess) failed = 1;
        });
    }
    g_stage_busy.clear();
    if (rc != 0) return rc;
    if (failed) GSX_FAIL("staged copy: a HIP call failed (%s)", hipGetErrorString(hipGetLastError()));
    return 0;
}

}  // namespace

extern "C" int gsx_dev_upload_staged(gsx_ctx *c, void *dst_dev, const void *src_host, size_t bytes)
{
    if (!c || (bytes && (!dst_dev || !src_host))) GSX_FAIL("gsx_dev_upload_staged: null argument");
    if (bytes == 0) return 0;
    return staged_copy(c, static_cast<char *>(dst_dev), const_cast<char *>(static_cast<const char *>(src_host)), bytes, true);
}

extern "C" int gsx_dev_download_staged(gsx_ctx *c, void *dst_host, const void *src_dev, size_t bytes)
{
    if (!c || (bytes && (!dst_host || !src_dev))) GSX_FAIL("gsx_dev_download_staged: null argument");
    if (bytes == 0) return 0;
    return staged_copy(c, const_cast<char *>(static_cast<const char *>(src_dev)), static_cast<char *>(dst_host), bytes, false);
}
