#!/usr/bin/env python3
"""CPU experiment (numpy, no GPU): how many inner nodes does an 8x8 block of primary rays enter in different hierarchies over the
SAME leaf boxes?  The search tree is free to be any hierarchy of nested boxes over the reference's leaves,
so its quality is a lever: (a) the radix tree over properly normalised Morton codes (what librtmi builds today), (b) a top-down
binned surface-area-heuristic tree, (c) bottom-up locally-ordered clustering (PLOC) over the Morton order.
    python tools/tree_quality.py man tenryuu          (uses the oracle as a scene / camera / leaf-box source: a tool, not product)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.setrecursionlimit(100000)


def area(lo, hi):
    d = np.maximum(hi - lo, 0)
    return 2 * (d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0])


class Tree:
    """binary tree over leaves 0..n-1: arrays left/right (child >= 0 inner index, < 0: ~leaf), lo/hi per inner node, root = 0"""
    def __init__(self, n):
        self.left = np.zeros(n - 1, np.int64); self.right = np.zeros(n - 1, np.int64)
        self.lo = np.zeros((n - 1, 3)); self.hi = np.zeros((n - 1, 3)); self.count = 0

    def new(self):
        self.count += 1
        return self.count - 1


def morton(c, lo, hi):
    q = np.clip(((c - lo) / np.maximum(hi - lo, 1e-30) * 1024).astype(np.int64), 0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    return (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])


def build_lbvh(llo, lhi):
    n = len(llo)
    c = (llo + lhi) * 0.5
    codes = morton(c, llo.min(0), lhi.max(0))
    order = np.argsort(codes, kind='stable')
    codes = codes[order]
    t = Tree(n)

    def rec(a, b):                      # leaves order[a..b], returns child id
        if a == b:
            return ~int(order[a])
        me = t.new()
        if codes[a] == codes[b]:
            m = (a + b) // 2
        else:
            bit = int(codes[a] ^ codes[b]).bit_length() - 1
            m = a + int(np.searchsorted(codes[a:b + 1] >> bit, (codes[a] >> bit) + 1)) - 1
        l, r = rec(a, m), rec(m + 1, b)
        t.left[me], t.right[me] = l, r
        ids = order[a:b + 1]
        t.lo[me], t.hi[me] = llo[ids].min(0), lhi[ids].max(0)
        return me
    rec(0, n - 1)
    return t


def build_sah(llo, lhi, bins=16):
    n = len(llo)
    c = (llo + lhi) * 0.5
    t = Tree(n)

    def rec(ids):
        if len(ids) == 1:
            return ~int(ids[0])
        me = t.new()
        lo, hi = llo[ids].min(0), lhi[ids].max(0)
        t.lo[me], t.hi[me] = lo, hi
        best = (np.inf, None)
        clo, chi = c[ids].min(0), c[ids].max(0)
        for ax in range(3):
            if chi[ax] <= clo[ax]:
                continue
            b = np.minimum(((c[ids, ax] - clo[ax]) / (chi[ax] - clo[ax]) * bins).astype(np.int64), bins - 1)
            cnt = np.bincount(b, minlength=bins)
            blo = np.full((bins, 3), np.inf); bhi = np.full((bins, 3), -np.inf)
            np.minimum.at(blo, b, llo[ids]); np.maximum.at(bhi, b, lhi[ids])
            llo_acc = np.minimum.accumulate(blo, 0); lhi_acc = np.maximum.accumulate(bhi, 0)
            rlo_acc = np.minimum.accumulate(blo[::-1], 0)[::-1]; rhi_acc = np.maximum.accumulate(bhi[::-1], 0)[::-1]
            cl = np.cumsum(cnt)
            for s in range(bins - 1):
                nl, nr = cl[s], len(ids) - cl[s]
                if nl == 0 or nr == 0:
                    continue
                cost = area(llo_acc[s], lhi_acc[s]) * nl + area(rlo_acc[s + 1], rhi_acc[s + 1]) * nr
                if cost < best[0]:
                    best = (cost, b <= s)
        if best[1] is None:
            m = len(ids) // 2
            mask = np.zeros(len(ids), bool); mask[:m] = True
        else:
            mask = best[1]
        l, r = rec(ids[mask]), rec(ids[~mask])
        t.left[me], t.right[me] = l, r
        return me
    rec(np.arange(n))
    return t


def build_ploc(llo, lhi, radius=16):
    """PLOC (Meister & Bittner 2018): clusters in Morton order; every cluster finds the neighbour within +-radius with the smallest
    merged area; mutual pairs merge; repeat."""
    n = len(llo)
    c = (llo + lhi) * 0.5
    order = np.argsort(morton(c, llo.min(0), lhi.max(0)), kind='stable')
    t = Tree(n)
    ids = [~int(i) for i in order]                   # cluster -> child id
    clo = llo[order].copy(); chi = lhi[order].copy()
    inner = []                                       # (left, right, lo, hi) in creation order; root is last
    while len(ids) > 1:
        m = len(ids)
        best = np.full(m, np.inf); nb = np.full(m, -1)
        for off in range(1, radius + 1):
            if off >= m:
                break
            a = area(np.minimum(clo[:-off], clo[off:]), np.maximum(chi[:-off], chi[off:]))
            i = np.arange(m - off)
            upd = a < best[i]
            best[i[upd]] = a[upd]; nb[i[upd]] = i[upd] + off
            j = i + off
            upd = a < best[j]
            best[j[upd]] = a[upd]; nb[j[upd]] = i[upd]
        mutual = nb[nb] == np.arange(m)
        keep_ids, keep_lo, keep_hi = [], [], []
        for i in range(m):
            if mutual[i]:
                j = nb[i]
                if i < j:
                    lo, hi = np.minimum(clo[i], clo[j]), np.maximum(chi[i], chi[j])
                    inner.append((ids[i], ids[j], lo, hi))
                    keep_ids.append(len(inner) - 1); keep_lo.append(lo); keep_hi.append(hi)
            else:
                keep_ids.append(ids[i]); keep_lo.append(clo[i]); keep_hi.append(chi[i])
        ids = keep_ids; clo = np.array(keep_lo); chi = np.array(keep_hi)
    # renumber: root = 0
    k = len(inner)
    remap = lambda x: x if x < 0 else k - 1 - x
    for idx, (l, r, lo, hi) in enumerate(inner):
        me = k - 1 - idx
        t.left[me], t.right[me], t.lo[me], t.hi[me] = remap(l), remap(r), lo, hi
    t.count = k
    return t


def sah_cost(t, llo, lhi):
    return float(area(t.lo[:t.count], t.hi[:t.count]).sum() / area(t.lo[0], t.hi[0]))


def depth(t):
    d = np.zeros(t.count, np.int64)
    best = 0
    stack = [(0, 1)]
    while stack:
        i, k = stack.pop()
        best = max(best, k)
        for ch in (t.left[i], t.right[i]):
            if ch >= 0:
                stack.append((int(ch), k + 1))
    return best


def hit_any(lo, hi, o, inv):
    with np.errstate(all='ignore'):
        t0 = (lo - o) * inv; t1 = (hi - o) * inv
        near = np.fmax.reduce(np.fmin(t0, t1), axis=1); far = np.fmin.reduce(np.fmax(t0, t1), axis=1)
        return bool(((near <= far) & (far > 0)).any())


def block_steps(t, llo, lhi, o, inv, quad=False):
    """inner nodes entered by the block (every child box is tested by its parent's step); quad: a step covers two levels"""
    steps = 0; leaves = 0
    stack = [0]
    while stack:
        i = stack.pop()
        steps += 1
        kids = [t.left[i], t.right[i]]
        if quad == 1:                               # open inner children in place: up to four grandchildren per step
            kids = [g for ch in kids for g in ((t.left[ch], t.right[ch]) if ch >= 0 else (ch,))]
        elif quad == 2:                             # the library's records: the slot with the largest surface area is opened, twice
            for _ in range(2):
                inner = [(float(area(t.lo[ch], t.hi[ch])), k) for k, ch in enumerate(kids) if ch >= 0]
                if not inner:
                    break
                k = max(inner)[1]
                ch = kids[k]
                kids[k] = t.left[ch]; kids.append(t.right[ch])
        for ch in kids:
            if ch >= 0:
                if hit_any(t.lo[ch], t.hi[ch], o, inv):
                    stack.append(int(ch))
            else:
                if hit_any(llo[~ch], lhi[~ch], o, inv):
                    leaves += 1
    return steps, leaves


def main():
    import oracle as orc
    from conftest import packed_scene
    names = sys.argv[1:] or ['man']
    res = (1920, 1080)
    rng = np.random.default_rng(1)
    for name in names:
        sc = orc.Scene(packed_scene(name))
        ob = orc.BVH(sc, 0)
        n = ob.n
        llo = ob.nodes['lo'][:n, :3].astype(np.float64); lhi = ob.nodes['hi'][:n, :3].astype(np.float64)
        cam = sc.demo_camera(*res)
        rays = orc.generate_rays(cam)
        names_ = rays.dtype.names
        print(name, n, 'leaves; ray fields', names_)
        O = np.array(rays['origin'][:, :3], np.float64); D = np.array(rays['direction'][:, :3], np.float64)
        with np.errstate(all='ignore'):
            INV = 1.0 / D
        # the sample: every 80th block of the kernel's own enumeration (64x64 tiles row-major, 8x8 blocks row-major inside), with the
        # kernel's own step count when tools/block_costs.py has left it in gpurun_out/
        tiles_x = (res[0] + 63) // 64
        costf = os.path.join(ROOT, 'gpurun_out', 'block_cost_%s.npy' % name)
        gpu_cost = (np.load(costf) & 0xFFFF) if os.path.exists(costf) else None
        nblocks = tiles_x * ((res[1] + 63) // 64) * 64
        sample = np.arange(7, nblocks, 80)
        trees = {}
        nodef = os.path.join(ROOT, 'gpurun_out', 'search_nodes_%s.npy' % name)
        if os.path.exists(nodef):                   # the library's own search tree (tools/block_costs.py)
            sn = np.load(nodef)
            t = Tree(n)
            conv = lambda ch: ~int(ch) if ch < n else int(ch) - n
            for i in range(n - 1):
                t.left[i] = conv(sn['data'][n + i, 1]); t.right[i] = conv(sn['data'][n + i, 2])
            t.lo[:] = sn['boundsMin'][n:, :3]; t.hi[:] = sn['boundsMax'][n:, :3]; t.count = n - 1
            trees['library'] = t
            print('  library search tree: SAH cost %.1f, depth %d' % (sah_cost(t, llo, lhi), depth(t)))
        for label, fn in (('lbvh', build_lbvh), ('sah', build_sah), ('ploc16', build_ploc)):
            t0 = time.time(); trees[label] = fn(llo, lhi); dt = time.time() - t0
            print('  %-7s built in %5.1f s: SAH cost %.1f, depth %d' % (label, dt, sah_cost(trees[label], llo, lhi), depth(trees[label])), flush=True)
        tot = {k: [0, 0, 0, 0] for k in trees}
        gpu_total = 0; used = 0
        for b in sample:
            tile, blk = b // 64, b % 64
            y0, x0 = (tile // tiles_x) * 64 + (blk >> 3) * 8, (tile % tiles_x) * 64 + (blk & 7) * 8
            if y0 + 8 > res[1] or x0 + 8 > res[0]:
                continue
            used += 1
            if gpu_cost is not None:
                gpu_total += int(gpu_cost[b])
            px = ((y0 + np.arange(8))[:, None] * res[0] + x0 + np.arange(8)[None, :]).ravel()
            for k, t in trees.items():
                s2, l2 = block_steps(t, llo, lhi, O[px], INV[px], False)
                s4, l4 = block_steps(t, llo, lhi, O[px], INV[px], 1)
                sg, lg = block_steps(t, llo, lhi, O[px], INV[px], 2)
                tot[k][0] += s2; tot[k][1] += s4; tot[k][2] += l2; tot[k][3] += sg
        print('  %d sampled blocks; the kernel itself took %s steps on them' % (used, gpu_total if gpu_cost is not None else '?'))
        for k, v in tot.items():
            print('  %-7s binary steps %8d  two-level steps %8d  largest-opened-twice steps %8d  leaves %7d' % (k, v[0], v[1], v[3], v[2]))


if __name__ == '__main__':
    main()


def build_hybrid(llo, lhi, cut=32, top='sah'):
    """the radix tree's sub-trees of at most `cut` leaves kept, the hierarchy above them rebuilt (top-down binned SAH or PLOC over the
    sub-trees' boxes)"""
    base = build_lbvh(llo, lhi)
    n = len(llo)
    # leaves below every inner node
    cnt = np.zeros(base.count, np.int64)
    for i in range(base.count - 1, -1, -1):          # children have larger indices (pre-order numbering of rec)
        c = 0
        for ch in (base.left[i], base.right[i]):
            c += 1 if ch < 0 else cnt[ch]
        cnt[i] = c
    roots = []                                       # sub-tree roots at the cut (child ids: >= 0 inner, < 0 leaf)
    stack = [0]
    while stack:
        i = stack.pop()
        for ch in (base.left[i], base.right[i]):
            if ch >= 0 and cnt[ch] > cut:
                stack.append(int(ch))
            else:
                roots.append(int(ch))
    rlo = np.array([llo[~r] if r < 0 else base.lo[r] for r in roots]); rhi = np.array([lhi[~r] if r < 0 else base.hi[r] for r in roots])
    topt = (build_sah if top == 'sah' else build_ploc)(rlo, rhi)
    # splice: new tree = top nodes (ids 0..k-1) + kept base nodes (renumbered behind them)
    k = topt.count
    t = Tree(n)
    t.count = 0
    remap = {}
    def copy_sub(r):                                  # base sub-tree -> t, returns child id
        if r < 0:
            return r
        me = k + len(remap); remap[r] = me
        return me
    # first assign ids to all kept base inner nodes
    keep = []
    for r in roots:
        if r >= 0:
            st = [r]
            while st:
                i = st.pop(); keep.append(i)
                for ch in (base.left[i], base.right[i]):
                    if ch >= 0:
                        st.append(int(ch))
    ids = {b: k + j for j, b in enumerate(keep)}
    conv = lambda ch: ch if ch < 0 else ids[int(ch)]
    for b, me in ids.items():
        t.left[me], t.right[me], t.lo[me], t.hi[me] = conv(base.left[b]), conv(base.right[b]), base.lo[b], base.hi[b]
    convtop = lambda ch: ch if ch >= 0 else (roots[~ch] if roots[~ch] < 0 else ids[roots[~ch]])
    for i in range(k):
        t.left[i], t.right[i], t.lo[i], t.hi[i] = convtop(topt.left[i]), convtop(topt.right[i]), topt.lo[i], topt.hi[i]
    t.count = k + len(keep)
    return t, len(roots)
