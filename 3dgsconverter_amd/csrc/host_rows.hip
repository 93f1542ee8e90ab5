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
#include <cstdint>
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
