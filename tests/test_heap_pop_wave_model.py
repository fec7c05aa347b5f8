"""The wave-parallel heappop of the planner kernels (csrc/avp_plan_kernels.h: pl_heap_pop_wave), restated as a scalar model
and compared with CPython's heapq.heappop -- the reference's open list (hybrid_a_star.py:96, Node.__lt__ :61-68) -- on
arbitrary arrays: heaps, heaps whose order was broken by in-place key changes (the reference lowers an open node's f
without re-heapifying, :224-230), plain unordered arrays, ties. The device code is compared with the oracle on the GPU
(tests/test_gpu_plan*.py); this test pins the derivation the kernel comment states: six levels per fetch, the walk on
fetched values, the final _siftdown as ONE comparison of the last element with the path's original entries."""
import heapq
import random


class _Key:
    def __init__(self, f, n):
        self.f, self.n = f, n

    def __lt__(self, o):
        return self.f < o.f


def pop_wave_model(h):
    """h: list of (f, node). Returns (popped node, array after the pop)."""
    nh = len(h) - 1
    last = h[nh]
    if nh == 0:
        return last[1], []
    root = h[0][1]
    cur1, level, P = 1, 0, 1              # 1-based: the hole, path entries recorded, root of the fetched subtree
    my_pos, my_ent = {}, {}
    while True:
        fetched = {}
        for lane in range(63):            # lane r - 1 holds relative position r of the subtree under P (levels 0 .. 5)
            r = lane + 1
            k = r.bit_length() - 1
            abs1 = (P << k) + (r - (1 << k))
            if abs1 <= nh:
                fetched[lane] = h[abs1 - 1]
        rr, leaf = 1, False
        for _ in range(5):
            left1 = 2 * cur1
            if left1 > nh:
                leaf = True
                break
            cl = 2 * rr - 1
            cf, cn = fetched[cl]
            right = 0
            if left1 + 1 <= nh:
                rf, rn = fetched[cl + 1]
                if not (cf < rf):
                    right, cf, cn = 1, rf, rn
            my_pos[level], my_ent[level] = cur1 - 1, (cf, cn)
            level += 1
            cur1, rr = left1 + right, 2 * rr + right
        if leaf:
            break
        P = cur1
    stop = [t for t in range(level) if not (last[0] < my_ent[t][0])]
    j = max(stop) + 1 if stop else 0
    out = list(h[:nh])
    for t in range(j):
        out[my_pos[t]] = my_ent[t]
    out[my_pos[j] if j < level else cur1 - 1] = last
    return root, out


def test_wave_heappop_model_equals_heapq():
    rng = random.Random(20260927)
    sizes = [1, 2, 3, 4, 5, 7, 31, 32, 33, 62, 63, 64, 65, 100, 127, 128, 500, 1000, 2047, 2048, 3000, 9000]
    for trial in range(6000):
        n = rng.choice(sizes)
        vals = [rng.choice([rng.random(), round(rng.random() * 5) / 5]) for _ in range(n)]
        hh = [_Key(v, i) for i, v in enumerate(vals)]
        kind = trial % 3
        if kind < 2:
            heapq.heapify(hh)
        if kind == 1:
            for _ in range(rng.randint(1, 4)):            # in-place key changes, no re-heapify
                hh[rng.randrange(n)].f = rng.random()
        h = [(x.f, x.n) for x in hh]
        root, out = pop_wave_model(h)
        r = heapq.heappop(hh)
        assert r.n == root, (trial, n)
        assert [(x.f, x.n) for x in hh] == out, (trial, n)
