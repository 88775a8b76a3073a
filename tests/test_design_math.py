"""Design-level claims of the gfx950 kernels, checked on the CPU (no device needed).

1. The 3-way bf16 operand split of csrc/split.h (DZN_PREC_F32_SPLIT) is EXACT, and the six products the
   kernels keep differ from the fp32 product by at most 2^-22 |a w| (emulated with numpy bit arithmetic).
2. The LDS images of the contraction / conv / attention kernels are bank-conflict free for
   ds_read_b128 under the MI355X lane-group model (MI355X_MICROARCH.md, LDS table: four 16-lane groups
   {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63}; bank = (addr/4) % 64).
   The address formulas below are the ones the kernels use.
3. The k-permutation of the pre-split weight planes is a bijection that matches the two fp32 slots a lane
   group reads.
"""
import numpy as np
import pytest


# ----------------------------------------------------------------------------- 1. split arithmetic
def _bf16_rne(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest bf16 (ties to even), returned as float32"""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return (((u + r) >> 16) << 16).astype(np.uint32).view(np.float32)


def _split3(x):
    hi = _bf16_rne(x)
    r1 = (x - hi).astype(np.float32)          # exact in fp32
    mid = _bf16_rne(r1)
    r2 = (r1 - mid).astype(np.float32)        # exact in fp32
    lo = _bf16_rne(r2)
    return hi, mid, lo


def test_three_way_split_is_exact_and_six_products_suffice():
    g = np.random.default_rng(0)
    n = 400_000
    mant = g.standard_normal(n).astype(np.float32)
    x = (mant * np.exp2(g.integers(-40, 40, n)).astype(np.float32)).astype(np.float32)
    w = (g.standard_normal(n) * np.exp2(g.integers(-20, 20, n))).astype(np.float32)
    xh, xm, xl = _split3(x)
    assert np.array_equal(((xh + xm).astype(np.float32) + xl).astype(np.float32), x)      # bit exact
    assert np.all(np.abs(xm) <= np.abs(x) * 2.0 ** -8) and np.all(np.abs(xl) <= np.abs(x) * 2.0 ** -16)
    wh, wm, wl = _split3(w)
    exact = x.astype(np.float64) * w.astype(np.float64)
    six = sum(a.astype(np.float64) * b.astype(np.float64)
              for a, b in ((xh, wh), (xh, wm), (xm, wh), (xh, wl), (xm, wm), (xl, wh)))
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -22, rel.max()          # the three dropped cross terms
    # each kept product is exact in the bf16 MFMA (8 x 8 significant bits) and in fp32
    for a, b in ((xh, wh), (xm, wm)):
        p = a.astype(np.float64) * b.astype(np.float64)
        assert np.array_equal(p.astype(np.float32).astype(np.float64), p)


# ----------------------------------------------------------------------------- 2. LDS layouts
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
          list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def _worst_conflict(addr_of_lane) -> int:
    worst = 0
    for grp in GROUPS:
        banks = {}
        for lane in grp:
            a = addr_of_lane(lane)
            assert a % 16 == 0
            for b in range(4):
                banks.setdefault((a // 4 + b) % 64, set()).add(a)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def _wswz(row):   # gemm_split.hip: wswz / gemm_split_pre.hip: pswz
    return (0x78 >> (2 * ((row >> 2) & 3))) & 3


def test_contraction_fragment_reads_are_conflict_free():
    for base in range(0, 256, 16):
        # fp32 A tile, 128-B rows, slot XOR ((row >> 1) & 7): slots lq and 4 + lq (gemm_split.hip read_a)
        for first in (0, 4):
            a = lambda l: (base + (l & 15)) * 128 + ((((first + (l >> 4)) ^ (((base + (l & 15)) >> 1) & 7))) << 4)
            assert _worst_conflict(a) == 1
        # bf16 plane rows of 64 B, chunk XOR g((row >> 2) & 3): W planes, pre-split A planes
        w = lambda l: (base + (l & 15)) * 64 + (((l >> 4) ^ _wswz(base + (l & 15))) << 4)
        assert _worst_conflict(w) == 1
        # the same rows WITHOUT the XOR conflict 2-way: the swizzle is what makes them free
        lin = lambda l: (base + (l & 15)) * 64 + ((l >> 4) << 4)
        assert _worst_conflict(lin) == 2


def test_conv_and_attention_fragment_reads_are_conflict_free():
    # conv_split.hip: pixel rows of 64 B, chunk XOR ((px >> 1) & 3), ANY pixel offset (the dw = 0, 1, 2 shifts)
    for base in range(0, 140):
        c = lambda l: (base + (l & 15)) * 64 + (((l >> 4) ^ (((base + (l & 15)) >> 1) & 3)) << 4)
        assert _worst_conflict(c) == 1
    # attention_split.hip: K / V^T planes, 128-B rows, slot (half*4 + lq) XOR ((row >> 1) & 7)
    for base in (0, 16, 32, 48):
        for half in (0, 1):
            k = lambda l: (base + (l & 15)) * 128 + (((half * 4 + (l >> 4)) ^ (((base + (l & 15)) >> 1) & 7)) << 4)
            assert _worst_conflict(k) == 1


# ----------------------------------------------------------------------------- 3. weight-plane k order
def test_weight_plane_k_permutation_matches_the_a_slots():
    pos = [8 * ((k & 15) >> 2) + (k & 3) + 4 * (k >> 4) for k in range(32)]     # split_weights_kernel
    assert sorted(pos) == list(range(32))
    for q in range(4):                       # lane group q reads positions 8q .. 8q+7 of a plane row ...
        ks = [k for k in range(32) if pos[k] // 8 == q]
        ks_in_order = sorted(ks, key=lambda k: pos[k])
        # ... = fp32 slot q (k = 4q .. 4q+3) followed by slot 4 + q (k = 16+4q .. 16+4q+3) of the A row
        assert ks_in_order == list(range(4 * q, 4 * q + 4)) + list(range(16 + 4 * q, 16 + 4 * q + 4))


# ----------------------------------------------------------------------------- 4. device linkage, as an algorithm
def _linkage_model(e, BLK):
    """csrc/linkage.hip step for step in numpy: lower bounds lb / guesses nb per row, the global argmin over per-block
    minima (bmin / barg), the merged row's bound taken exact from the partial minima of the NEW row, stale bounds rescanned
    over ACTIVE columns only (retired columns keep their last distances), Lance-Williams centroid update in scipy's
    operation order.  Returns the dendrogram and the number of rescans."""
    n = len(e)
    nblk = (n + BLK - 1) // BLK
    x64 = e.astype(np.float64)
    D = np.sqrt(((x64[:, None, :] - x64[None, :, :]) ** 2).sum(-1))
    np.fill_diagonal(D, np.inf)
    lb = D.min(axis=1)
    nb = D.argmin(axis=1)
    size = np.ones(n, dtype=np.int64)
    cid = np.arange(n)
    Z = np.zeros((n - 1, 4))

    def block_minima():
        bm = np.full(nblk, np.inf)
        ba = np.full(nblk, 2 ** 31 - 1, dtype=np.int64)
        for b in range(nblk):
            seg = lb[b * BLK:(b + 1) * BLK]
            j = int(np.argmin(seg))                     # lowest row on ties
            bm[b], ba[b] = seg[j], b * BLK + j
        return bm, ba

    def refresh(bm, ba, row):
        b = row // BLK
        seg = lb[b * BLK:(b + 1) * BLK]
        j = int(np.argmin(seg))
        bm[b], ba[b] = seg[j], b * BLK + j

    bm, ba = block_minima()
    hp = None
    rescans = 0
    for k in range(n - 1):
        if hp is not None:                              # select_kernel, k > 0: exact bound of the previous merge's row
            hi_prev, hp_val, hp_idx = hp
            order = np.lexsort((hp_idx, hp_val))
            v, i = hp_val[order[0]], hp_idx[order[0]]
            lb[hi_prev], nb[hi_prev] = v, (i if v < np.inf else -1)
            refresh(bm, ba, hi_prev)
        while True:
            o = np.lexsort((ba, bm))[0]                 # lowest value, then lowest row index
            d, x = bm[o], int(ba[o])
            y = int(nb[x])
            if y >= 0 and D[x, y] == d:
                break
            row = np.where(size != 0, D[x], np.inf)     # block_argmin_active
            j = int(np.argmin(row))
            lb[x], nb[x] = row[j], j
            refresh(bm, ba, x)
            rescans += 1
        lo, hi = (x, y) if x < y else (y, x)
        nlo, nhi = int(size[lo]), int(size[hi])
        Z[k] = (min(cid[lo], cid[hi]), max(cid[lo], cid[hi]), d, nlo + nhi)
        size[lo], size[hi], cid[hi] = 0, nlo + nhi, n + k
        lb[lo] = lb[hi] = np.inf
        nb[hi] = -1
        # update_kernel
        sx, sy, dxy = float(nlo), float(nhi), d
        newrow = np.full(n, np.inf)
        for z in range(n):
            if z == hi or size[z] == 0:
                continue
            dxi, dyi = D[lo, z], D[hi, z]
            nd = np.sqrt((((sx * dxi * dxi) + (sy * dyi * dyi)) - (sx * sy * dxy * dxy) / (sx + sy)) / (sx + sy))
            newrow[z] = nd
            D[hi, z] = D[z, hi] = nd                    # column lo is NOT blanked
            if nb[z] == lo:
                nb[z] = hi
            if nd < lb[z]:
                lb[z], nb[z] = nd, hi
        hp_val = np.full(nblk, np.inf)
        hp_idx = np.full(nblk, 2 ** 31 - 1, dtype=np.int64)
        for b in range(nblk):
            seg = newrow[b * BLK:(b + 1) * BLK]
            j = int(np.argmin(seg))
            hp_val[b], hp_idx[b] = seg[j], b * BLK + j
        hp = (hi, hp_val, hp_idx)
        bm, ba = block_minima()
    return Z, rescans


def _linkage_step_model(e, BLK):
    """numpy model of csrc/linkage.hip's r3 loop: ONE launch per step (step_kernel).  Every workgroup reduces the same
    published records (minimum bound of its rows, that row's neighbour, exact flag) + the one row whose bound is fixed up
    at the start of the launch, and all reach the same decision: MERGE (bound exact) or RESCAN (stale).  What a launch
    reads was written by the PREVIOUS launch (the model keeps `pub` = the published copy and builds `new` separately);
    words of the previous merge's pair are substituted, not read."""
    n = len(e)
    G = (n + BLK - 1) // BLK
    x64 = e.astype(np.float64)
    D = np.sqrt(((x64[:, None, :] - x64[None, :, :]) ** 2).sum(-1))
    np.fill_diagonal(D, np.inf)
    lb = D.min(axis=1)
    nb = D.argmin(axis=1).astype(np.int64)
    size = np.ones(n, dtype=np.int64)
    cid = np.arange(n)
    Z = np.zeros((n - 1, 4))
    BIG = 2 ** 31 - 1
    exf = np.ones(n, dtype=bool)        # bound of the row is exact (D[z, nb[z]] == lb[z])

    def publish(excl):
        recs = []
        for w in range(G):
            rows = np.arange(w * BLK, min((w + 1) * BLK, n))
            vals = np.where(rows == excl, np.inf, lb[rows])
            j = int(np.argmin(vals))                        # lowest row on ties
            v, x = vals[j], int(rows[j])
            y, exact = -1, False
            if v < np.inf:
                y = int(nb[x])
                exact = y >= 0 and D[x, y] == v
                # the kernel does not look D up: it keeps a per-row flag, maintained by the row's owner (below)
                assert exact == bool(exf[x] and y >= 0), (x, y)
            recs.append((v, x, y, exact))
        return recs

    def partial_minima(row):
        parts = []
        for w in range(G):
            seg = row[w * BLK:(w + 1) * BLK]
            j = int(np.argmin(seg))
            parts.append((seg[j], w * BLK + j))
        return parts

    pub = dict(kind="none", k=0, rec=publish(-1), part=None)
    launches = rescans = 0
    while True:
        prev = pub
        if prev["kind"] == "done":
            break
        launches += 1
        assert launches < 4 * n + 8
        ev, ex, ey = np.inf, BIG, -1
        if prev["kind"] in ("merge", "rescan"):             # fix-ups owed by the previous step (owner threads)
            ev, idx = min(prev["part"], key=lambda t: (t[0], t[1]))
            ex = prev["hi"] if prev["kind"] == "merge" else prev["x"]
            ey = idx if ev < np.inf else -1
            lb[ex], nb[ex], exf[ex] = ev, ey, ey >= 0
            if prev["kind"] == "merge":
                size[prev["hi"]] = prev["nlo"] + prev["nhi"]
                cid[prev["hi"]] = n + prev["k"] - 1
                size[prev["lo"]] = 0
        k = prev["k"]
        cand = [(v, x) for v, x, _, _ in prev["rec"]] + [(ev, ex)]
        d, x = min(cand)
        assert d < np.inf and 0 <= x < n
        if x == ex:
            y, exact = ey, True
        else:
            (y, exact), = [(yy, xx) for v, r, yy, xx in prev["rec"] if r == x and v == d]
        exact = exact and y >= 0
        if not exact:
            rescans += 1
            row = np.where(size != 0, D[x], np.inf)
            pub = dict(kind="rescan", k=k, x=x, part=partial_minima(row), rec=publish(x))
            continue
        lo, hi = (x, y) if x < y else (y, x)
        nlo, nhi = int(size[lo]), int(size[hi])
        Z[k] = (min(cid[lo], cid[hi]), max(cid[lo], cid[hi]), d, nlo + nhi)
        newrow = np.full(n, np.inf)
        for z in range(n):
            if z == hi:
                D[hi, lo] = np.inf
                lb[z], nb[z], exf[z] = np.inf, -1, False
            elif z == lo:
                lb[z], exf[z] = np.inf, False
            elif size[z] != 0:
                dxi, dyi = D[lo, z], D[hi, z]
                sx, sy = float(nlo), float(nhi)
                nd = np.sqrt((((sx * dxi * dxi) + (sy * dyi * dyi)) - (sx * sy * d * d) / (sx + sy)) / (sx + sy))
                newrow[z] = nd
                D[hi, z] = D[z, hi] = nd
                if nb[z] == lo:
                    nb[z], exf[z] = hi, nd == lb[z]         # a guess; lb stays a lower bound
                elif nb[z] == hi:
                    exf[z] = nd == lb[z]                    # the remembered neighbour's distance moved
                if nd < lb[z]:
                    lb[z], nb[z], exf[z] = nd, hi, True
        pub = dict(kind="done" if k + 1 >= n - 1 else "merge", k=k + 1, lo=lo, hi=hi, nlo=nlo, nhi=nhi,
                   part=partial_minima(newrow), rec=publish(-1))
    return Z, rescans, launches


def test_linkage_step_model_equals_scipy():
    """The one-launch-per-step loop (r3) is again the same greedy centroid linkage: dendrogram equal to scipy's, and the
    launch count stays below the 2 (n - 1) of the two-kernel loop."""
    from scipy.cluster.hierarchy import linkage
    r = np.random.default_rng(11)
    for n, K, BLK in ((2, 1, 4), (3, 1, 4), (23, 3, 4), (60, 4, 8), (61, 5, 3), (40, 2, 64), (97, 6, 16), (200, 5, 32)):
        cent = r.standard_normal((K, 12))
        e = (cent[r.integers(0, K, n)] + 0.3 * r.standard_normal((n, 12))).astype(np.float32)
        e /= np.linalg.norm(e, axis=1, keepdims=True)
        Z, rescans, launches = _linkage_step_model(e, BLK)
        Zs = linkage(e.astype(np.float64), method="centroid", metric="euclidean")
        assert np.array_equal(Z[:, [0, 1, 3]], Zs[:, [0, 1, 3]]), (n, K, BLK)
        assert np.abs(Z[:, 2] - Zs[:, 2]).max() < 1e-12
        assert launches == n - 1 + rescans and launches <= 2 * (n - 1) + 1


def _linkage_top2_model(e):
    """Row-level numpy model of csrc/linkage.hip's r6 rules (step2_kernel): every row remembers its TWO nearest neighbours
       I1: l1 <= D[z][c] for every active c != z;            e1: D[z][n1] == l1
       I2: l2 <= D[z][c] for every active c not in {z, n1};  e2: D[z][n2] == l2
    so that a row whose nearest neighbour is retired by a merge falls back on an exact second instead of a rescan.
    -> (Z, steps, rescans)"""
    INF = np.inf
    n = len(e)
    x = e.astype(np.float64)
    D = np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(-1))
    np.fill_diagonal(D, INF)
    size = np.ones(n, np.int64)
    cid = np.arange(n)
    Z = np.zeros((n - 1, 4))
    l1, l2 = np.zeros(n), np.zeros(n)
    n1, n2 = np.zeros(n, np.int64), np.zeros(n, np.int64)
    e1, e2 = np.ones(n, bool), np.ones(n, bool)

    def top2_of(row, mask):
        v = np.where(mask, row, INF)
        i1 = int(np.argmin(v))
        v1 = v[i1]
        v[i1] = INF
        i2 = int(np.argmin(v))
        v2 = v[i2]
        return v1, (i1 if np.isfinite(v1) else -1), v2, (i2 if np.isfinite(v2) else -1)

    for z in range(n):
        l1[z], n1[z], l2[z], n2[z] = top2_of(D[z], size != 0)
        e1[z], e2[z] = n1[z] >= 0, n2[z] >= 0
    k = steps = rescans = 0
    while k < n - 1:
        steps += 1
        assert steps < 4 * n + 8
        lbm = np.where(size != 0, l1, INF)
        xw = int(np.argmin(lbm))
        d = lbm[xw]
        assert np.isfinite(d)
        if not (e1[xw] and n1[xw] >= 0):                   # RESCAN: refills both slots
            rescans += 1
            l1[xw], n1[xw], l2[xw], n2[xw] = top2_of(D[xw], size != 0)
            e1[xw], e2[xw] = n1[xw] >= 0, n2[xw] >= 0
            continue
        y = int(n1[xw])
        assert D[xw, y] == d
        lo, hi = (xw, y) if xw < y else (y, xw)
        nlo, nhi = int(size[lo]), int(size[hi])
        Z[k] = (min(cid[lo], cid[hi]), max(cid[lo], cid[hi]), d, nlo + nhi)
        newrow = np.full(n, INF)
        size[lo] = 0
        for z in range(n):
            if z == hi or z == lo or size[z] == 0:
                continue
            dxi, dyi = D[lo, z], D[hi, z]
            sx, sy = float(nlo), float(nhi)
            v = np.sqrt((((sx * dxi * dxi) + (sy * dyi * dyi)) - (sx * sy * d * d) / (sx + sy)) / (sx + sy))
            newrow[z] = v
            D[hi, z] = D[z, hi] = v
            d1 = n1[z] in (lo, hi)
            d2 = n2[z] in (lo, hi)
            if not d1:
                if v < l1[z]:
                    l2[z], n2[z], e2[z] = l1[z], n1[z], e1[z]
                    l1[z], n1[z], e1[z] = v, hi, True
                elif not d2:
                    if v < l2[z]:
                        l2[z], n2[z], e2[z] = v, hi, True
                elif v <= l2[z]:
                    l2[z], n2[z], e2[z] = v, hi, True
                else:
                    n2[z], e2[z] = hi, False
            elif not d2 and e2[z]:
                if v < l2[z]:
                    l1[z], n1[z], e1[z] = v, hi, True
                else:
                    l1[z], n1[z], e1[z] = l2[z], n2[z], True
                    n2[z], e2[z] = hi, False
            else:
                if v <= l2[z]:
                    l1[z], n1[z], e1[z] = v, hi, True
                else:
                    l1[z], n1[z], e1[z] = l2[z], hi, False
                e2[z] = False
                if d2:
                    n2[z] = hi
            # the invariants the rules promise
            act = size != 0
            act[z] = False
            act[lo] = False
            assert l1[z] <= D[z][act].min()
            if e1[z]:
                assert D[z, n1[z]] == l1[z] == D[z][act].min()
                rest = act.copy()
                rest[n1[z]] = False
                if rest.any():
                    assert l2[z] <= D[z][rest].min()
                    if e2[z]:
                        assert D[z, n2[z]] == l2[z] == D[z][rest].min()
        D[hi, lo] = D[lo, hi] = INF
        size[hi] = nlo + nhi
        cid[hi] = n + k
        m = size != 0
        m[hi] = False
        l1[hi], n1[hi], l2[hi], n2[hi] = top2_of(newrow, m)
        e1[hi], e2[hi] = n1[hi] >= 0, n2[hi] >= 0
        l1[lo] = INF
        k += 1
    return Z, steps, rescans


def test_linkage_top2_model_equals_scipy():
    """(r6) two remembered neighbours per row: the dendrogram still equals scipy's, the stated invariants hold after every
    update (asserted inside the model), and the loop needs fewer steps than the one-neighbour loop on the same data."""
    from scipy.cluster.hierarchy import linkage
    r = np.random.default_rng(11)
    tot2 = tot1 = 0
    for n, K in ((2, 1), (3, 1), (23, 3), (60, 4), (61, 5), (97, 6), (200, 5)):
        cent = r.standard_normal((K, 12))
        e = (cent[r.integers(0, K, n)] + 0.3 * r.standard_normal((n, 12))).astype(np.float32)
        e /= np.linalg.norm(e, axis=1, keepdims=True)
        Z, steps, rescans = _linkage_top2_model(e)
        Zs = linkage(e.astype(np.float64), method="centroid", metric="euclidean")
        assert np.array_equal(Z[:, [0, 1, 3]], Zs[:, [0, 1, 3]]), (n, K)
        assert np.abs(Z[:, 2] - Zs[:, 2]).max() < 1e-12
        assert steps == n - 1 + rescans
        _, rescans1, launches1 = _linkage_step_model(e, 16)
        tot2 += steps
        tot1 += launches1
    assert tot2 < 0.8 * tot1, (tot2, tot1)


def test_linkage_algorithm_model_equals_scipy():
    """The restructured selection of csrc/linkage.hip (block minima, exact bound of the merged row, masked rescans) is a
    different route to the SAME greedy centroid linkage: the model reproduces scipy's dendrogram on clustered unit vectors
    for several block sizes, including n not a multiple of the block and blocks larger than n."""
    from scipy.cluster.hierarchy import linkage
    r = np.random.default_rng(7)
    for n, K, BLK in ((23, 3, 4), (60, 4, 8), (61, 5, 3), (40, 2, 64), (97, 6, 16)):
        cent = r.standard_normal((K, 12))
        e = (cent[r.integers(0, K, n)] + 0.3 * r.standard_normal((n, 12))).astype(np.float32)
        e /= np.linalg.norm(e, axis=1, keepdims=True)
        Z, rescans = _linkage_model(e, BLK)
        Zs = linkage(e.astype(np.float64), method="centroid", metric="euclidean")
        assert np.array_equal(Z[:, [0, 1, 3]], Zs[:, [0, 1, 3]]), (n, K, BLK)
        assert np.abs(Z[:, 2] - Zs[:, 2]).max() < 1e-12
        assert rescans < 2 * n                          # stale bounds are the exception, not the rule


# ---------------------------------------------------------------------------------------------------------------------
# (r4) Ring schedules of the fused BasicBlock kernels (csrc/resblock_fused.hip, csrc/resblock_ws.hip): a host model of
# WHICH ring slot every phase reads and writes and WHERE the barriers sit, checked for (a) every read finds the row it
# wants, (b) no slot is overwritten between a barrier-free pair of (read, write) by different wavefronts.
def _run_intervals(intervals, X, M, x_slot, m_slot):
    for iv in intervals:
        reads = [(ring, row) for _, kind, ring, row in iv if kind == "r"]
        writes = [(ring, row) for _, kind, ring, row in iv if kind == "w"]
        for ring, row in reads:                       # (a) the wanted row is resident
            store, sl = (X, x_slot(row)) if ring == "X" else (M, m_slot(row))
            assert store.get(sl) == row, (ring, row, sl, store.get(sl))
        for ring, row in writes:                      # (b) nothing read in this interval lives in the slot being written
            sl = x_slot(row) if ring == "X" else m_slot(row)
            for ring2, row2 in reads:
                if ring2 == ring:
                    sl2 = x_slot(row2) if ring == "X" else m_slot(row2)
                    assert sl2 != sl, f"{ring} slot {sl}: row {row} written while row {row2} is read in the same interval"
        for ring, row in writes:
            (X if ring == "X" else M)[x_slot(row) if ring == "X" else m_slot(row)] = row


@pytest.mark.parametrize("H", [1, 2, 3, 5, 40, 80])
def test_resblock_fused_ring_schedule(H):
    """resblock_fused.hip: X ring of 4 slots (row q at (q + 1) & 3), M ring of 3 slots (row m at (m + 1) % 3).  Step r, r = -1 .. H - 1:
    [phase A reads x rows r, r + 1, r + 2] | barrier | [store intermediate row r + 1; store x row r + 3] | barrier |
    [phase B reads intermediate rows r - 1, r, r + 1 (r >= 0)]."""
    x_slot, m_slot = (lambda q: (q + 1) & 3), (lambda m: (m + 1) % 3)
    X, M = {}, {}
    _run_intervals([[("all", "w", "X", q) for q in (-1, 0, 1)] + [("all", "w", "M", -1)]], X, M, x_slot, m_slot)    # prologue
    for r in range(-1, H):
        ivs = []
        a = [("all", "r", "X", q) for q in (r, r + 1, r + 2)] if r + 1 < H else []
        ivs.append(a)
        w = [("all", "w", "M", r + 1)]
        if r + 3 <= H:
            w.append(("all", "w", "X", r + 3))
        ivs.append(w)
        ivs.append([("all", "r", "M", m) for m in (r - 1, r, r + 1)] if r >= 0 else [])
        # phase B of step r and phase A of step r + 1 are NOT separated by a barrier: merge them into one interval
        _run_intervals(ivs[:2], X, M, x_slot, m_slot)
        nxt = [("all", "r", "X", q) for q in (r + 1, r + 2, r + 3)] if (r + 1 < H and r + 2 < H) else []
        _run_intervals([ivs[2] + nxt], X, M, x_slot, m_slot)


@pytest.mark.parametrize("H", [1, 2, 3, 5, 40, 80])
def test_resblock_ws_ring_schedule(H):
    """resblock_ws.hip: both rings have 4 slots (row at (row + 1) & 3); ONE barrier per step; in step r (r = -1 .. H) the producers
    read x rows r, r + 1, r + 2 and write intermediate row r + 1 (if r + 1 <= H) while — in the same interval — the consumers
    read intermediate rows r - 2, r - 1, r for output row r - 1 (if 0 <= r - 1 < H) and write x row r + 3 (if r + 3 <= H)."""
    sl = lambda q: (q + 1) & 3      # noqa: E731
    X, M = {}, {}
    _run_intervals([[("cons", "w", "X", q) for q in (-1, 0, 1)] + [("prod", "w", "M", -1)]], X, M, sl, sl)
    outputs = []
    for r in range(-1, H + 1):
        iv = []
        if r + 1 <= H:
            if r + 1 < H:
                iv += [("prod", "r", "X", q) for q in (r, r + 1, r + 2)]
            iv.append(("prod", "w", "M", r + 1))
        if r + 3 <= H:
            iv.append(("cons", "w", "X", r + 3))
        if 0 <= r - 1 < H:
            iv += [("cons", "r", "M", m) for m in (r - 2, r - 1, r)]
            outputs.append(r - 1)
        _run_intervals([iv], X, M, sl, sl)
    assert outputs == list(range(H))


def test_column_group_tile_order_is_a_permutation():
    """csrc/gemm_split.hip "COLUMN GROUPS": with G groups the linear tile index walks group-major (all row blocks of the first
    cg column tiles, then the next group ...).  Restated here: every (row block, column tile) is produced exactly once for any
    grid and any G <= 8, groups are contiguous column ranges of near-equal width, and G = 1 is the row-block-major order."""
    def tile(t, tiles_m, tiles_n, groups):
        if groups <= 1:
            return t // tiles_n, t % tiles_n
        base, rem = tiles_n // groups, tiles_n % groups
        g, cg, cs, r = 0, base + (rem > 0), 0, t
        while r >= tiles_m * cg:
            r -= tiles_m * cg
            cs += cg
            g += 1
            cg = base + (g < rem)
        tm = r // cg
        return tm, cs + r - tm * cg

    for tiles_m, tiles_n in ((1, 2), (3, 5), (7, 8), (1166, 8), (1749, 15), (13, 16), (5, 9)):
        for groups in (1, 2, 3, 4, 8):
            if groups > tiles_n:
                continue
            seen = [tile(t, tiles_m, tiles_n, groups) for t in range(tiles_m * tiles_n)]
            assert sorted(seen) == [(m, n) for m in range(tiles_m) for n in range(tiles_n)], (tiles_m, tiles_n, groups)
            if groups > 1:     # group-major: the column index never decreases by more than a group's width along the walk
                first_cols = [n for m, n in seen if m == 0]
                assert first_cols == sorted(first_cols)
                width = -(-tiles_n // groups)
                quarter = seen[: tiles_m * (tiles_n // groups)]
                assert max(n for _, n in quarter) < width
