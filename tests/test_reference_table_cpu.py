"""CPU: the host replay behind PHL_BUILD_REFERENCE_TABLE (csrc/phl_reftable.hip, reached through the C ABI's test
hook; no device needed) against the oracle's faithful-table mode, which is pinned bit for bit to the reference
engine above its first table doubling (tests/golden/PIN_REPORT.json)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from oracle import phl_oracle as po


def _replay(keys_clean, cand_vid):
    import phl

    lib = phl.load_library()
    M, d = keys_clean.shape
    N = cand_vid.size
    keys_out = np.empty((M + 128, d), np.int16)
    cand_out = np.empty(N, np.int32)
    hidden = np.empty(64, np.int32)
    M_ref, nh, nb = C.c_int64(0), C.c_int(0), C.c_int(0)
    rc = lib.phl_debug_reference_table(keys_clean.ctypes.data_as(C.c_void_p), cand_vid.ctypes.data_as(C.c_void_p), M, d, N,
                                       keys_out.ctypes.data_as(C.c_void_p), M + 128, C.byref(M_ref),
                                       cand_out.ctypes.data_as(C.c_void_p), hidden.ctypes.data_as(C.c_void_p), 64,
                                       C.byref(nh), C.byref(nb))
    assert rc == 0, lib.phl_last_error()
    return keys_out[:M_ref.value].copy(), cand_out, hidden[:nh.value].copy(), nb.value


def _check(ref):
    Oc = po.Oracle(ref)                       # defect-free table: what the device build numbers first
    Of = po.Oracle(ref, faithful_table=True)  # == the reference engine
    keys_c = np.ascontiguousarray(Oc.keys())
    cand = np.ascontiguousarray(Oc.replay()[0].ravel())
    keys_r, cand_r, hidden, nb = _replay(keys_c, cand)
    assert len(keys_r) == Of.M
    assert np.array_equal(keys_r, Of.keys())
    assert np.array_equal(cand_r, Of.replay()[0].ravel())
    # a hidden vertex is never anybody's blur neighbour in the reference
    nbr = Of.neighbors()
    assert not np.isin(nbr, hidden).any() or len(hidden) == 0
    # ... and of the vertices sharing a key at most one is reachable (a key whose only vertex was filed from a
    # stale slot at the last doubling and never looked up again is hidden too, without being a duplicate)
    _, inv, cnt = np.unique(keys_r, axis=0, return_inverse=True, return_counts=True)
    inv = inv.ravel()
    for grp in np.nonzero(cnt > 1)[0]:
        members = np.nonzero(inv == grp)[0]
        assert len(set(members.tolist()) - set(hidden.tolist())) <= 1
    return Of.M - Oc.M, len(hidden)


@pytest.fixture(params=["simulation", "analytic"])
def replay_mode(request, monkeypatch):
    """Both forms of the replay: the table simulation, and the analytic form (which falls back to the simulation
    where it does not apply)."""
    monkeypatch.setenv("PHL_REPLAY_FAST", "1" if request.param == "analytic" else "0")
    return request.param


@pytest.mark.parametrize("n,d,scale,seed", [(3000, 5, 3.0, 1), (20000, 5, 8.0, 21), (60000, 5, 6.0, 22), (200000, 3, 40.0, 23),
                                            (9000, 8, 2.0, 5), (30000, 2, 300.0, 6), (12000, 5, 9.0, 7), (50000, 4, 12.0, 8)])
def test_replay_equals_faithful_oracle_random(n, d, scale, seed, replay_mode):
    rng = np.random.default_rng(seed)
    ref = (rng.random((n, d), dtype=np.float32) * np.float32(scale)).astype(np.float32)
    extra, nh = _check(ref)
    print(f"n={n} d={d}: {extra} duplicate vertices, {nh} hidden")


def test_replay_equals_faithful_oracle_many_seeds(replay_mode):
    """Sweep seeds so that all the sub-cases occur: stale slot == proper slot (no duplicate), in-flight key new /
    already present, the duplicate found again only after the next doubling, several doublings."""
    seen = set()
    for seed in range(40):
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.integers(6000, 40000))
        d = int(rng.integers(2, 7))
        ref = (rng.random((n, d), dtype=np.float32) * np.float32(rng.choice([6.0, 10.0, 25.0]))).astype(np.float32)
        seen.add(_check(ref))
    assert any(e > 0 for e, _ in seen) and any(e == 0 for e, _ in seen), seen


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "growth_*.npz"))),
                         ids=os.path.basename)
def test_replay_on_stored_growth_cases(path, replay_mode):
    from _golden_util import load_growth_case

    g = load_growth_case(path)
    extra, _ = _check(g["ref"])
    assert extra == g["M"] - g["clean_M"]


def test_threshold_hit_exactly_at_the_last_vertex():
    """M == 2^k - 1 after splat: the doubling happens inside blur()'s first neighbour lookup (latent
    hazard).  Build such a lattice by truncating a random feature set at the pixel that creates vertex 16383."""
    for seed in range(30):
        rng = np.random.default_rng(99 + seed)
        ref = (rng.random((12000, 5), dtype=np.float32) * np.float32(9.0)).astype(np.float32)
        vid = po.Oracle(ref).replay()[0]
        first = int(np.nonzero(vid.max(1) >= 16382)[0][0])    # pixel whose lookups create vertex id 16382 (the 16383rd)
        if vid[first].max() != 16382:
            continue                                           # that pixel goes on to create more: M would overshoot
        O = po.Oracle(ref[:first + 1])
        assert O.M == 16383
        _check(ref[:first + 1])
        Of = po.Oracle(ref[:first + 1], faithful_table=True)
        keys_r, _, _, nb = _replay(np.ascontiguousarray(O.keys()), np.ascontiguousarray(O.replay()[0].ravel()))
        assert nb != -2 and nb == Of.neighbors()[0, 0, 0]
        return
    pytest.fail("no prefix with exactly 16383 vertices in 30 draws")


def test_skipping_the_last_doubling_changes_nothing(monkeypatch):
    """The replay does not carry out the LAST doubling of a build (csrc/phl_reftable.hip, `frozen`): what lookups in
    the re-filed table return is derived from the old table.  Both forms -- every doubling simulated, and the last one
    skipped -- must produce the same vertices, candidate resolutions, hidden list and blur neighbour, on vertex counts
    well inside an epoch, just above a doubling and just below the next one (where the skip must not apply)."""
    cases = 0
    for seed in range(60):
        rng = np.random.default_rng(5000 + seed)
        d = int(rng.integers(2, 7))
        # aim the vertex count at interesting places relative to the thresholds 16383, 32767, 65535
        target = int(rng.choice([17000, 20000, 30000, 32600, 32900, 34000, 50000, 65400, 66000, 90000]))
        n = max(2000, target // (d + 1) + int(rng.integers(0, 400)))
        ref = (rng.random((n, d), dtype=np.float32) * np.float32(rng.choice([40.0, 80.0]))).astype(np.float32)   # ~every candidate its own vertex
        Oc = po.Oracle(ref)
        if Oc.M < 16383:
            continue
        keys_c = np.ascontiguousarray(Oc.keys())
        cand = np.ascontiguousarray(Oc.replay()[0].ravel())
        monkeypatch.setenv("PHL_REPLAY_SKIP_FINAL", "1")
        a = _replay(keys_c, cand)
        monkeypatch.setenv("PHL_REPLAY_SKIP_FINAL", "0")
        b = _replay(keys_c, cand)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3], (seed, Oc.M)
        cases += 1
    assert cases >= 40


def test_analytic_replay_equals_the_simulation(monkeypatch):
    """phl_reference_table_fast derives doubling times, stale probes and the order of a key's entries by counting,
    without a table (csrc/phl_reftable.hip); wherever it applies its result must be the simulation's: vertices,
    candidate resolutions, hidden list.  120 random lattices around the thresholds 16383 / 32767 / 65535 / 131071."""
    applied = fell_back = with_dups = 0
    for seed in range(120):
        rng = np.random.default_rng(9000 + seed)
        d = int(rng.integers(2, 7))
        target = int(rng.choice([16500, 17000, 20000, 30000, 32700, 32800, 34000, 50000, 65500, 65600, 70000, 100000, 131200, 150000]))
        n = max(2000, target // (d + 1) + int(rng.integers(0, 300)))
        scale = float(rng.choice([40.0, 80.0])) if seed % 2 else float(rng.choice([6.0, 9.0, 12.0]))
        if seed % 2 == 0:
            n = int(rng.integers(20000, 90000))          # vertices shared by many pixels: keys looked up again and again
        ref = (rng.random((n, d), dtype=np.float32) * np.float32(scale)).astype(np.float32)
        Oc = po.Oracle(ref)
        if Oc.M < 16383:
            continue
        keys_c = np.ascontiguousarray(Oc.keys())
        cand = np.ascontiguousarray(Oc.replay()[0].ravel())
        monkeypatch.setenv("PHL_REPLAY_FAST", "0")
        want = _replay(keys_c, cand)
        monkeypatch.setenv("PHL_REPLAY_FAST", "2")
        try:
            got = _replay(keys_c, cand)
        except AssertionError:
            fell_back += 1                                # not applicable here (wrap, or a doubling inside blur)
            continue
        applied += 1
        with_dups += int(len(want[0]) > Oc.M)
        assert np.array_equal(got[0], want[0]), (seed, "keys")
        assert np.array_equal(got[1], want[1]), (seed, "candidate resolution")
        assert np.array_equal(got[2], want[2]) and got[3] == want[3], (seed, "hidden / blur neighbour")
    print(f"analytic replay applied {applied}x (with duplicate vertices {with_dups}x), fell back {fell_back}x")
    assert applied >= 60 and with_dups >= 10 and fell_back <= applied // 4


# ---- the analytic replay's occupancy check (probe_paths_do_not_wrap) ------------------------------------------------
def ref_homes(keys, cap):
    """Home slots under the reference's hash (permutohedral.h:109-116; size_t arithmetic) for a power-of-two capacity."""
    h = np.zeros(len(keys), np.uint64)
    with np.errstate(over="ignore"):
        for i in range(keys.shape[1]):
            h = (h + keys[:, i].astype(np.int64).astype(np.uint64)) * np.uint64(2531011)
    return (h & np.uint64(cap - 1)).astype(np.int64)


def probe_paths(keys, n_clean, extra, stale, cap, check, on_device=0):
    import phl

    lib = phl.load_library()
    keys = np.ascontiguousarray(keys, np.int16)
    ex, st, ck = (np.ascontiguousarray(a, np.int32) for a in (extra, stale, check))
    res = C.c_int(-1)
    rc = lib.phl_debug_probe_paths(keys.ctypes.data_as(C.c_void_p), n_clean, keys.shape[1], ex.ctypes.data_as(C.c_void_p), len(ex),
                                   st.ctypes.data_as(C.c_void_p), len(st), cap, ck.ctypes.data_as(C.c_void_p), len(ck),
                                   on_device, C.byref(res))
    assert rc == 0, lib.phl_last_error()
    return res.value


def brute_force_paths(homes_all, cap, check_homes):
    """Insert every entry by linear probing (order does not matter for which slots end up full)."""
    full = np.zeros(cap, bool)
    for h in homes_all:
        while full[h]:
            h = (h + 1) % cap
        full[h] = True
    return int(all((~full[h:]).any() for h in check_homes))


def keys_with_homes(rng, cap, d=5, count=1 << 18):
    keys = rng.integers(-3000, 3000, (count, d)).astype(np.int16)
    keys = np.unique(keys, axis=0)
    return keys, ref_homes(keys, cap)


def test_occupancy_check_sees_a_full_table_tail():
    cap = 1 << 15
    rng = np.random.default_rng(5)
    keys, homes = keys_with_homes(rng, cap)
    tail = np.nonzero(homes >= cap - 4)[0]
    assert len(tail) >= 8
    body = np.nonzero(homes < cap - 64)[0][:9000]
    # the last four slots full (six entries want them): every path from there wraps
    sel = np.concatenate([body, tail[:6]])
    k = keys[sel]
    n = len(k)
    assert probe_paths(k, n, [], [], cap, [n - 1]) == 0
    assert probe_paths(k, n, [], [], cap, [0]) == (1 if homes[sel[0]] < cap - 4 else 0)
    # one entry in the tail: slots behind it are free
    sel = np.concatenate([body, tail[:1]])
    k = keys[sel]
    want = 1 if homes[tail[0]] < cap - 1 else 0
    assert probe_paths(k, len(k), [], [], cap, [len(k) - 1]) == want


def test_occupancy_check_equals_brute_force_probing():
    rng = np.random.default_rng(17)
    for trial in range(12):
        cap = 1 << int(rng.choice([15, 16]))
        keys, homes = keys_with_homes(rng, cap, count=1 << 17)
        n = int(rng.integers(cap // 8, cap // 2 - 1))
        # bias towards trouble: half of the cases pile entries onto the table's end
        if trial % 2:
            order = np.argsort(-homes, kind="stable")
            pile = order[: int(rng.integers(4, 200))]
            rest = rng.permutation(np.setdiff1d(np.arange(len(keys)), pile))[: n - len(pile)]
            sel = np.concatenate([rest, pile])
        else:
            sel = rng.permutation(len(keys))[:n]
        k, hk = keys[sel], homes[sel]
        extra = rng.integers(0, n, int(rng.integers(0, 4)))
        stale = rng.integers(0, n, int(rng.integers(0, 3)))
        all_homes = np.concatenate([hk, hk[extra], ref_homes(k[stale], cap // 2)]) if len(stale) else np.concatenate([hk, hk[extra]])
        for c in list(rng.integers(0, n, 3)) + [n - 1, int(np.argmax(hk))]:
            want = brute_force_paths(all_homes.tolist(), cap, [int(hk[c])])
            assert probe_paths(k, n, extra, stale, cap, [c]) == want, (trial, cap, n, c)
