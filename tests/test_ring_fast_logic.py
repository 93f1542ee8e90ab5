"""CPU: the selection argument of knn_ring_fast (csrc/sor_grid.hip) replayed in numpy.

The kernel filters candidates by a float32 squared distance d32 <= r_t^2 (1 + 1e-6), bins d32 into 64 histogram bins
(a monotone function of d32), takes the bins that hold the first k+1 candidates, thr2 = the largest d32 inside them, and
ranks only the candidates with d32 <= thr2 (1 + 1e-6) exactly in float64.  Claim: whenever the exact (k+1)-th smallest
float64 distance is <= r_t^2, the ranked subset contains the exact k+1 nearest (so its k+1 smallest exact distances are
the true ones).  This test replays the arithmetic -- including adversarial near-ties one float32 ulp apart -- against
a brute-force float64 selection."""
import numpy as np


def d32_of(q, p):
    d = q.astype(np.float32) - p.astype(np.float32)
    dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
    # fmaf(dz, dz, fmaf(dy, dy, dx * dx)) in float32: emulate the two fused steps with float64 products rounded once
    t = (dx.astype(np.float64) * dx.astype(np.float64)).astype(np.float32)
    t = (dy.astype(np.float64) * dy.astype(np.float64) + t.astype(np.float64)).astype(np.float32)
    return (dz.astype(np.float64) * dz.astype(np.float64) + t.astype(np.float64)).astype(np.float32)


def d64_of(q, p):
    d = p.astype(np.float64) - q.astype(np.float64)
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]


def select(q, pts, kk, rt):
    """-> (indices ranked exactly, exact kth or None when the kernel would leave the query to knn_ring)"""
    thr = np.float32(np.float32(rt * rt) * np.float32(1.0 + 1e-6))
    inv_thr = np.float32(1.0) / thr
    d32 = d32_of(q[None, :], pts)
    cand = np.nonzero(d32 <= thr)[0]
    if len(cand) < kk or len(cand) > 128:
        return None, None
    b = np.minimum(63, (d32[cand] * inv_thr * np.float32(64.0)).astype(np.int32))
    hist = np.bincount(b, minlength=64)
    bstar = int(np.argmax(np.cumsum(hist) >= kk))
    m32 = d32[cand][b <= bstar].max()
    thr2 = np.float32(m32 * np.float32(1.0 + 1e-6))
    fin = cand[d32[cand] <= thr2]
    if len(fin) > 64:
        return None, None
    dex = np.sort(d64_of(q, pts[fin]))
    kth = dex[kk - 1]
    if not kth <= rt * rt:
        return None, None
    return fin, dex[:kk]


def test_histogram_select_keeps_the_true_nearest():
    rng = np.random.default_rng(5)
    solved = 0
    for trial in range(3000):
        kk = int(rng.choice([2, 4, 9, 17, 33]))
        n = int(rng.integers(kk, 120))
        scale = float(rng.choice([1e-3, 1.0, 37.0, 1e4]))
        q = (rng.standard_normal(3) * scale).astype(np.float32)
        pts = (q + rng.standard_normal((n, 3)) * 0.6 * scale * 1e-2).astype(np.float32)
        pts[0] = q                                          # the query itself is among the candidates
        if trial % 3 == 0:                                  # near ties: shells one float32 ulp apart around the k-th distance
            r = np.float32(0.5 * scale * 1e-2)
            dirs = rng.standard_normal((n - 1, 3))
            dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
            radii = np.nextafter(r, np.float32(np.inf) if trial % 2 else np.float32(0), dtype=np.float32) if trial % 6 == 0 else r
            pts[1:] = (q.astype(np.float64) + dirs * np.float64(radii)).astype(np.float32)
        rt = 1.2 * scale * 1e-2
        fin, got = select(q, pts, kk, rt)
        if fin is None:
            continue
        solved += 1
        want = np.sort(d64_of(q, pts))[:kk]
        np.testing.assert_array_equal(got, want)
    assert solved > 1500


def test_row_trim_never_drops_a_point_inside_the_ball():
    """the per-row cell range of knn_ring_fast: every point within r_t of the query lies in a visited (row, x-cell)"""
    rng = np.random.default_rng(6)
    for trial in range(400):
        o = (rng.standard_normal(3) * 10).astype(np.float32)
        h = np.float32(rng.uniform(0.01, 2.0))
        inv_h = np.float32(1.0) / h
        hp = 1.0 / np.float64(inv_h)
        dims = rng.integers(5, 40, 3)
        pts = (o + rng.random((4000, 3)).astype(np.float32) * (dims * h).astype(np.float32)).astype(np.float32)
        cell = np.minimum(dims - 1, ((pts - o) * inv_h).astype(np.int32))        # cell_coord of csrc/sor_grid.hip
        q = pts[rng.integers(len(pts))]
        cq = np.minimum(dims - 1, ((q - o) * inv_h).astype(np.int32))
        rt = min(rng.uniform(0.8, 1.9), 1.9) * hp
        rtc = rt * np.float64(inv_h)
        u = (q.astype(np.float64) - o.astype(np.float64)) * np.float64(inv_h)
        inside = d64_of(q, pts) <= rt * rt
        for p_cell in cell[inside]:
            yy, zz = p_cell[1], p_cell[2]
            assert abs(yy - cq[1]) <= 2 and abs(zz - cq[2]) <= 2 and abs(p_cell[0] - cq[0]) <= 2
            dy = (yy - 1e-3 - u[1]) if u[1] < yy - 1e-3 else ((u[1] - (yy + 1.001)) if u[1] > yy + 1.001 else 0.0)
            dz = (zz - 1e-3 - u[2]) if u[2] < zz - 1e-3 else ((u[2] - (zz + 1.001)) if u[2] > zz + 1.001 else 0.0)
            rem = rtc * rtc - dy * dy - dz * dz
            assert rem >= 0.0
            dxc = np.sqrt(rem) + 1e-3
            xa, xb = int(np.floor(u[0] - dxc)), int(np.floor(u[0] + dxc))
            assert xa <= p_cell[0] <= xb


def _padded_bitonic_merge(L, BS, a, b):
    """csrc/knn_common.h: TopNet<L>::merge_block restated -- the list of L sorted values takes a block of BS values through
    c[L-BS+i] = min(a[L-BS+i], b_sorted[BS-1-i]) and a bitonic merge network of the next power of two P, whose first P - L
    positions are an imagined -inf (compare-exchanges with them are not emitted)"""
    a, b = list(a), sorted(b)
    P = 8
    while P < L:
        P *= 2
    off = P - L
    for i in range(BS):
        a[L - BS + i] = min(a[L - BS + i], b[BS - 1 - i])
    half, nce = P // 2, 0
    while half >= 1:
        for p in range(off, P):
            if p & half == 0:
                x, y = a[p - off], a[p + half - off]
                a[p - off], a[p + half - off] = min(x, y), max(x, y)
                nce += 1
        half //= 2
    return a, nce


def test_padded_bitonic_merge():
    """0-1 principle: every sorted 0/1 list x every 0/1 block, for every list length the kernels instantiate (and the
    other multiples of 4); plus random reals.  The compare-exchange counts are the ones the kernel comments quote."""
    rng = np.random.default_rng(5)
    counts = {}
    for L in range(8, 65, 4):
        for za in range(L + 1):
            for zb in range(9):
                a, b = [0] * za + [1] * (L - za), [0] * zb + [1] * (8 - zb)
                out, counts[L] = _padded_bitonic_merge(L, 8, a, b)
                assert out == sorted(a + b)[:L], (L, za, zb)
        for _ in range(200):
            a, b = sorted(rng.random(L).tolist()), rng.random(8).tolist()
            assert _padded_bitonic_merge(L, 8, a, b)[0] == sorted(a + b)[:L]
    assert {L: counts[L] for L in range(8, 65, 8)} == {8: 12, 16: 32, 24: 52, 32: 80, 40: 100, 48: 128, 56: 156, 64: 192}
    assert counts[28] == 64   # (the default k = 25: a fifth fewer compare-exchanges than the 32-entry list)


def test_leaf_window_minimum_by_doubling():
    """csrc/sor_tree.hip: tree_leaf_flags_kernel restated -- the minimum over the cap + 1 window levels that contain a sorted point is
    taken from the minima of all 64-entry windows (six doubling passes) at offsets 0, 64, ... and cap + 1 - 64 (overlapping where
    cap + 1 is no multiple of 64); for every capacity the kernel accepts, against the plain loop round 4 ran"""
    rng = np.random.default_rng(11)
    tile = 256
    for cap in (64, 65, 96, 100, 127, 128, 129, 192, 200, 255, 256):
        span = tile + cap
        a = rng.integers(0, 65, size=span + 400).astype(np.int64)
        a[span:] = 64                       # "no window starts here": what the kernel writes beyond the entries that exist
        w = a.copy()
        st = 1
        while st < 64:                      # w[t] = min a[t .. t + 2 st)
            w = np.minimum(w, np.concatenate([w[st:], np.full(st, 64)]))
            st *= 2
        for t in range(tile):
            want = a[t:t + cap + 1].min()
            got = w[t + cap + 1 - 64]
            d = 0
            while d + 64 <= cap:
                got = min(got, w[t + d])
                d += 64
            assert got == want, (cap, t)


def _next_word(lens, starts, r, off):
    """csrc/sor_tree.hip: knn_leaf's next_word restated -- the candidate ranges are ONE flat sequence cut into words of 32 candidates,
    a word taking pieces of up to four ranges; returns the word's (base, end) pieces and where the next word starts"""
    nr, fill, pieces = len(lens), 0, []
    for _ in range(4):
        ln = lens[r] if r < nr else 0
        while r < nr and off >= ln:
            r, off = r + 1, 0
            ln = lens[r] if r < nr else 0
        take, st = 0, 0
        if r < nr and fill < 32:
            take = min(ln - off, 32 - fill)
            st = starts[r] + off
        pieces.append((st - fill, fill + take))   # candidate t of the word, t in [previous end, this end), is index base + t
        fill += take
        off += take
    return pieces, r, off


def test_filter_passes_cover_every_candidate_once():
    """... and the filter takes them in passes of `park` words, each pass resuming at (cr, coff) = the position behind the last word
    it parked (round 5: boxes of more than 896 candidates).  Every candidate index must come out exactly once, in order."""
    rng = np.random.default_rng(3)
    for trial in range(300):
        nr = int(rng.integers(1, 64))
        lens = [int(x) for x in rng.choice([0, 1, 2, 5, 31, 32, 33, 70, 200], size=nr)]
        starts, at = [], 1000
        for ln in lens:
            starts.append(at)
            at += ln + int(rng.integers(0, 50))
        want = [s + i for s, ln in zip(starts, lens) for i in range(ln)]
        park = int(rng.choice([1, 2, 28, 32]))
        got, cr, coff, passes, nwords = [], 0, 0, 0, 0
        while True:
            widx, more = 0, False
            pieces, r, off = _next_word(lens, starts, cr, coff)
            while pieces[3][1] > 0 and widx < park:
                prev_end = 0
                for base, end in pieces:
                    got.extend(base + t for t in range(prev_end, end))
                    prev_end = end
                widx += 1
                nwords += 1
                cr, coff = r, off
                pieces, r, off = _next_word(lens, starts, cr, coff)
            more = pieces[3][1] > 0
            passes += 1
            if not more:
                break
        assert got == want, (trial, lens, park)
        assert passes == max(1, -(-nwords // park))


def test_kmeans_strip_items_cover_every_operand_word_once():
    """csrc/kmeans_cs.h: every wave of the centroid-stationary assign kernel cuts the operand words of ITS rows out of its LDS strip;
    item -> (row, 16-dimension slice, half) must hit every (point tile, slice, hi/lo lane) slot of the block exactly once, for the three
    shapes the launcher uses (waves, points per block) and D = 45 (three slices), 24 (two), 9 (one)"""
    for waves, block in ((16, 128), (4, 64), (2, 64)):
        for d in (45, 24, 9):
            ns = -(-(d + 3) // 16)          # km_dp(D) / 16: the dimensions + three |c|^2 pieces, padded to 16
            rpw = block // waves
            seen = set()
            for wv in range(waves):
                nitems = rpw * ns * 2
                for item in range(nitems):
                    rl, rest = item % rpw, item // rpw
                    ij, half = rest % ns, rest // ns
                    r = wv * rpw + rl
                    slot = (r >> 5, ij, (half << 5) | (r & 31))
                    assert half < 2 and slot not in seen
                    seen.add(slot)
                    # the 8 dimensions this item reads from the strip: row rl of the wave, dimensions 16 ij + 8 half ...
                    assert rl * d + min(16 * ij + 8 * half, d - 1) < rpw * d
            assert len(seen) == (block // 32) * ns * 64
