"""The rule csrc/sor_tree.hip builds its leaves by (tree_leaf_flags_kernel), restated in numpy and checked against
brute-force counting: in the sorted Morton keys, the node of point i at bit level b (all keys equal to key[i] above bit b) holds
more than C points  <=>  key[j] and key[j + C] agree above b for some j in [i - C, i].  Hence the smallest such b is a sliding
minimum over a[j] = bitlength(key[j] ^ key[j + C]) and the leaf -- the largest node with at most C points -- is one level below.
(The GPU suite pins the kernel itself through the masks; this pins the rule.)"""
import numpy as np
import pytest


def morton(q, bits):
    k = np.zeros(len(q), dtype=np.uint64)
    for t in range(bits):
        for a in range(3):
            k |= ((q[:, a] >> t) & 1).astype(np.uint64) << np.uint64(3 * t + a)
    return k


def leaf_levels(keys, cap, top):
    n = len(keys)
    a = np.full(n, top + 1, dtype=np.int64)
    for j in range(n - cap):
        a[j] = (int(keys[j]) ^ int(keys[j + cap])).bit_length()
    split = np.array([a[max(0, i - cap):i + 1].min() for i in range(n)])
    return np.maximum(split - 1, 0)


@pytest.mark.parametrize("seed", range(6))
def test_sliding_minimum_gives_the_largest_node_with_at_most_cap_points(seed):
    rng = np.random.default_rng(seed)
    bits, cap = 6, 8
    n = int(rng.integers(1, 500))
    pts = np.concatenate([rng.random((n, 3)), 0.5 + 0.02 * rng.standard_normal((n // 2, 3)), np.full((n // 3, 3), 0.25)])
    q = (np.clip(pts, 0, 0.999) * (1 << bits)).astype(np.int64)
    keys = np.sort(morton(q, bits))
    levels = leaf_levels(keys, cap, 3 * bits)
    for i in range(len(keys)):
        want = 0
        for b in range(3 * bits, -1, -1):
            if int(np.sum((keys >> np.uint64(b)) == (keys[i] >> np.uint64(b)))) <= cap:
                want = b
                break
        assert levels[i] == want, (i, levels[i], want)
    # leaves are runs of equal (key >> level); every point of a run carries the same level; a run holds <= cap points unless it
    # is one over-full cell of the finest level
    head = np.ones(len(keys), bool)
    head[1:] = [(int(keys[i]) >> int(levels[i])) != (int(keys[i - 1]) >> int(levels[i])) for i in range(1, len(keys))]
    starts = np.flatnonzero(head)
    for s, e in zip(starts, np.append(starts[1:], len(keys))):
        assert len(set(levels[s:e].tolist())) == 1
        assert e - s <= cap or levels[s] == 0
