"""The per-wave list builders of the blend kernels (csrc/render_common.h) and the reductions per half of a wave (csrc/wave_reduce.h).

Round 8 stopped running every (quadrant, Gaussian) entry over all 64 lanes of a wave: the forward and the tracking backward walk
one list per HALF-wave (build_half_lists), the mapping backward PAIRS neighbouring entries of a wave's list that live in different
halves of the quadrant (build_paired_lists), and a step that serves two entries reduces its sums per half.  The parity tests
hold the kernels' results against the oracle; these tests hold the builders themselves against what they promise:
  * every entry of a wave's list is served exactly once;
  * each half of a wave meets its entries in tile-list order (a pixel's blending order is the list order);
  * a paired step holds an upper-only and a lower-only entry, and the pairs are those of the documented rule -- ranks
    (2 m, 2 m + 1) first, then (2 m + 1, 2 m + 2) where neither was taken;
  * lists end in sentinels (record offset 32 * 128) so that the loops may read past the end by their unroll."""
import numpy as np
import pytest
import torch

from dgr_amd import _capi

pytestmark = pytest.mark.gpu

NB, LD, SENT = 128, 136, 128 * 32


def build(codes):
    lib = _capi.load()
    dev = torch.device("cuda:0")
    c = torch.from_numpy(np.ascontiguousarray(codes, np.uint8)).to(dev)
    paired = torch.zeros(4 * 280, dtype=torch.int32, device=dev)
    halves = torch.zeros(4 * 280, dtype=torch.int32, device=dev)
    rc = lib.dgr_debug_lane_lists(_capi.stream_handle(), c.data_ptr(), paired.data_ptr(), halves.data_ptr())
    assert rc == 0, _capi.last_error()
    torch.cuda.synchronize()
    out = []
    for t in (paired, halves):
        a = t.cpu().numpy().view(np.uint32).reshape(4, 280)
        out.append([dict(n=int(r[0]), split=(int(r[1]) | int(r[2]) << 32, int(r[3]) | int(r[4]) << 32), a=r[5:5 + LD], b=r[5 + LD:5 + 2 * LD])
                    for r in a])
    return out


def reference_pairs(types):
    """The documented rule on the list of types (1 upper only, 2 lower only, 3 both) -> list of steps, each (rank,) or (rank, rank)."""
    n = len(types)
    mg = lambda i, j: 0 <= i < n and 0 <= j < n and types[i] * types[j] == 2  # noqa: E731
    A = {m: mg(2 * m, 2 * m + 1) for m in range(-1, n // 2 + 2)}
    second = set()
    pair_of = {}
    for r in range(n):
        if r % 2 == 0 and A[r // 2]:
            pair_of[r] = r + 1
            second.add(r + 1)
        elif r % 2 == 1 and not A[(r - 1) // 2] and not A[(r + 1) // 2] and mg(r, r + 1):
            pair_of[r] = r + 1
            second.add(r + 1)
    return [((r, pair_of[r]) if r in pair_of else (r,)) for r in range(n) if r not in second]


def check_paired(codes, paired):
    for w in range(4):
        types = [(int(c) >> (2 * w)) & 3 for c in codes]
        slots = [i for i, t in enumerate(types) if t]
        tl = [types[i] for i in slots]
        p = paired[w]
        n = len(slots)
        if n > 64:  # long lists stay unpaired
            assert p["n"] == n and p["split"] == (0, 0)
            assert [int(x) // 32 for x in p["a"][:n]] == slots and [int(x) // 32 for x in p["b"][:n]] == slots
        else:
            steps = reference_pairs(tl)
            assert p["n"] == len(steps), (w, p["n"], len(steps))
            for s, st in enumerate(steps):
                up, lo = int(p["a"][s]) // 32, int(p["b"][s]) // 32
                is_split = (p["split"][0] >> s) & 1
                if len(st) == 1:
                    assert not is_split and up == lo == slots[st[0]], (w, s)
                else:
                    e = sorted(st, key=lambda r: tl[r])  # type 1 (upper) first
                    assert is_split and tl[e[0]] == 1 and tl[e[1]] == 2, (w, s)
                    assert up == slots[e[0]] and lo == slots[e[1]], (w, s)
            assert p["split"][1] == 0 and p["split"][0] >> len(steps) == 0
            # what the builder promises, independent of the rule: every entry once, each half in list order
            served = []
            for s in range(p["n"]):
                up, lo = int(p["a"][s]) // 32, int(p["b"][s]) // 32
                served += [up] if up == lo else [up, lo]
            assert sorted(served) == slots
            for half, want in ((p["a"], (1, 3)), (p["b"], (2, 3))):
                seen = [int(x) // 32 for x in half[:p["n"]] if types[int(x) // 32] in want]
                assert seen == [i for i in slots if types[i] in want]
        assert all(int(x) == SENT for x in p["a"][p["n"]:p["n"] + 4]) and all(int(x) == SENT for x in p["b"][p["n"]:p["n"] + 4])


def check_halves(codes, halves):
    for w in range(4):
        up = [i for i, c in enumerate(codes) if (int(c) >> (2 * w)) & 1]
        lo = [i for i, c in enumerate(codes) if (int(c) >> (2 * w + 1)) & 1]
        h = halves[w]
        n = max(len(up), len(lo))
        assert h["n"] == n
        assert [int(x) // 32 for x in h["a"][:len(up)]] == up and [int(x) // 32 for x in h["b"][:len(lo)]] == lo
        assert all(int(x) == SENT for x in h["a"][len(up):n + 4]) and all(int(x) == SENT for x in h["b"][len(lo):n + 4])


def draw(rng, fill, p_both=0.44):
    """128 codes: every quadrant takes a slot with probability `fill`; of those 44 % in both halves, the rest split evenly."""
    codes = np.zeros(NB, np.uint8)
    for w in range(4):
        take = rng.random(NB) < fill
        kind = rng.random(NB)
        t = np.where(kind < p_both, 3, np.where(kind < p_both + (1 - p_both) / 2, 1, 2))
        codes |= (np.where(take, t, 0) << (2 * w)).astype(np.uint8)
    return codes


@pytest.mark.parametrize("fill", [0.02, 0.17, 0.45, 0.8, 1.0])
def test_random_batches(fill):
    rng = np.random.default_rng(int(fill * 100))
    for _ in range(12):
        codes = draw(rng, fill)
        paired, halves = build(codes)
        check_paired(codes, paired)
        check_halves(codes, halves)


def test_corner_cases():
    both = np.full(NB, 0xFF, np.uint8)
    cases = {
        "nothing": np.zeros(NB, np.uint8),
        "everything in both halves": both,
        "upper halves only": np.full(NB, 0x55, np.uint8),
        "alternating upper / lower": np.where(np.arange(NB) % 2 == 0, 0x55, 0xAA).astype(np.uint8),
        "upper upper lower lower": np.where(np.arange(NB) % 4 < 2, 0x55, 0xAA).astype(np.uint8),
        "one entry": np.eye(1, NB, 77, dtype=np.uint8)[0] * 0x02,
        "exactly 64 entries, alternating": np.where(np.arange(NB) < 64, np.where(np.arange(NB) % 2 == 0, 0x01, 0x02), 0).astype(np.uint8),
        "65 entries": np.where(np.arange(NB) < 65, np.where(np.arange(NB) % 2 == 0, 0x01, 0x02), 0).astype(np.uint8),
        "a pair across the two staging waves": np.where(np.arange(NB) == 63, 0x01, np.where(np.arange(NB) == 64, 0x02, 0)).astype(np.uint8),
    }
    for name, codes in cases.items():
        paired, halves = build(codes)
        check_paired(codes, paired)
        check_halves(codes, halves)
    # alternating singles pair up completely: half the steps
    paired, _ = build(cases["exactly 64 entries, alternating"])
    assert paired[0]["n"] == 32 and paired[0]["split"][0] == (1 << 32) - 1


def half_reduce(x):
    lib = _capi.load()
    dev = torch.device("cuda:0")
    t = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    f = [torch.zeros(64, device=dev) for _ in range(3)]
    i = [torch.zeros(64, dtype=torch.int32, device=dev) for _ in range(3)]
    rc = lib.dgr_debug_half_reduce(_capi.stream_handle(), t.data_ptr(), *[o.data_ptr() for o in f], *[o.data_ptr() for o in i])
    assert rc == 0, _capi.last_error()
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in f], [o.cpu().numpy() for o in i]


def test_half_reductions_sum_each_half_on_its_own():
    rng = np.random.default_rng(3)
    x = rng.integers(-64, 64, size=(12, 64)).astype(np.float32)  # integer-valued: every order sums exactly
    (r0, r1, h3), (s0, s1, c3) = half_reduce(x)
    for half in range(2):
        lanes = slice(32 * half, 32 * half + 32)
        tot = x[:, lanes].sum(1)
        assert np.array_equal(r0[lanes], tot[s0[lanes]])
        for lane in range(32 * half, 32 * half + 32):
            if s1[lane] >= 0:
                assert r1[lane] == tot[s1[lane]]
            if c3[lane] >= 0:
                assert h3[lane] == tot[c3[lane]]
        # every value has a lane that delivers it: 0..7 through r0, 8..11 through r1, 0..2 of the three-value network
        assert sorted(set(s0[lanes].tolist())) == list(range(8))
        assert sorted(set(v for v in s1[lanes].tolist() if v >= 0)) == [8, 9, 10, 11]
        assert sorted(v for v in c3[lanes].tolist() if v >= 0) == [0, 1, 2]


@pytest.mark.parametrize("comp", range(12))
def test_half_reductions_one_hot(comp):
    for lane in (0, 5, 15, 16, 31, 32, 40, 47, 48, 63):
        x = np.zeros((12, 64), np.float32)
        x[comp, lane] = 3.0
        (r0, r1, h3), (s0, s1, c3) = half_reduce(x)
        mine = np.arange(64) // 32 == lane // 32
        assert np.array_equal(r0, np.where(mine & (s0 == comp), 3.0, 0.0))
        assert np.array_equal(np.where(s1 >= 0, r1, 0.0), np.where(mine & (s1 == comp), 3.0, 0.0))
        assert np.array_equal(np.where(c3 >= 0, h3, 0.0), np.where(mine & (c3 == comp), 3.0, 0.0))


def half_reduce16(x):
    lib = _capi.load()
    dev = torch.device("cuda:0")
    t = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    f = [torch.zeros(64, device=dev) for _ in range(2)]
    i = [torch.zeros(64, dtype=torch.int32, device=dev) for _ in range(2)]
    rc = lib.dgr_debug_half_reduce16(_capi.stream_handle(), t.data_ptr(), *[o.data_ptr() for o in f], *[o.data_ptr() for o in i])
    assert rc == 0, _capi.last_error()
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in f], [o.cpu().numpy() for o in i]


def test_sixteen_value_half_reductions_sum_each_half_on_its_own():
    """The paired step of the FULL backward (csrc/render_full.hip): wave_reduce16d_head + quad sums leave every half's sixteen totals in
    its own lanes -- values 0..7 through r0, 8..15 through r1, each named by wave_reduce16d_half_slot0 / 1."""
    rng = np.random.default_rng(4)
    x = rng.integers(-64, 64, size=(16, 64)).astype(np.float32)  # integer-valued: every order sums exactly
    (r0, r1), (s0, s1) = half_reduce16(x)
    for half in range(2):
        lanes = slice(32 * half, 32 * half + 32)
        tot = x[:, lanes].sum(1)
        assert np.array_equal(r0[lanes], tot[s0[lanes]]) and np.array_equal(r1[lanes], tot[s1[lanes]])
        assert sorted(set(s0[lanes].tolist())) == list(range(8)) and sorted(set(s1[lanes].tolist())) == list(range(8, 16))
        # the lanes that deliver (lane % 4 == 0) cover every value exactly once per half
        d0, d1 = s0[lanes][::4], s1[lanes][::4]
        assert sorted(d0.tolist()) == list(range(8)) and sorted(d1.tolist()) == list(range(8, 16))


@pytest.mark.parametrize("comp", range(16))
def test_sixteen_value_half_reductions_one_hot(comp):
    for lane in (0, 5, 15, 16, 31, 32, 40, 47, 48, 63):
        x = np.zeros((16, 64), np.float32)
        x[comp, lane] = 3.0
        (r0, r1), (s0, s1) = half_reduce16(x)
        mine = np.arange(64) // 32 == lane // 32
        assert np.array_equal(r0, np.where(mine & (s0 == comp), 3.0, 0.0))
        assert np.array_equal(r1, np.where(mine & (s1 == comp), 3.0, 0.0))
