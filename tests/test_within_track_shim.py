"""CPU check of the engine's nearest-track-point search (mpopis_amd/csrc/car_dynamics.h: within_track with an anchor) against
the oracle's literal findmin (car_racing_tracks.jl:68-92).  The header is compiled for the host (tests/shim/host_shim.cpp, test
infrastructure only).  Covers the three paths: ring candidates certified by ring_r2, neighbour-list scan, full scan; anchors
carried along random walks with occasional jumps; every bundled track, a hairpin that folds back on itself, and tiny tracks."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM_SRC = os.path.join(HERE, "shim", "host_shim.cpp")
SHIM_SO = os.path.join(HERE, "shim", "libhost_shim.so")
HDR = os.path.join(os.path.dirname(HERE), "mpopis_amd", "csrc", "car_dynamics.h")
dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def shim():
    if (not os.path.exists(SHIM_SO)) or os.path.getmtime(SHIM_SO) < max(os.path.getmtime(SHIM_SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-o", SHIM_SO, SHIM_SRC])
    L = C.CDLL(SHIM_SO)
    L.shim_within_anchor.argtypes = [C.c_int, dp, dp, dp, C.c_double, C.c_double, C.POINTER(C.c_int), dp]
    return L


def _tracks():
    from mpopis_amd.engine import default_track, BUNDLED_TRACKS
    out = [(n, default_track(name=n)) for n in BUNDLED_TRACKS]
    t = np.linspace(0, 1, 40)                                   # hairpin: two straights 6 m apart (non-ring points nearer than ring ones)
    hx = np.concatenate([100 * t, 100 * t[::-1]]); hy = np.concatenate([np.zeros(40), np.full(40, 6.0)])
    out.append(("hairpin", (np.ascontiguousarray(hx), np.ascontiguousarray(hy), np.full(80, 15.0))))
    for P in (2, 3, 5):
        a = np.linspace(0, 2 * np.pi, P, endpoint=False)
        out.append(("ring%d" % P, (np.ascontiguousarray(30 * np.cos(a)), np.ascontiguousarray(30 * np.sin(a)), np.full(P, 15.0))))
    return out


@pytest.mark.parametrize("name,track", _tracks(), ids=[n for n, _ in _tracks()])
def test_anchored_search_matches_findmin(shim, oracle, name, track):
    tx, ty, tw = (np.ascontiguousarray(a, dtype=np.float64) for a in track)
    rng = np.random.default_rng(abs(hash(name)) % 2 ** 31)
    P = len(tx)
    anchor = C.c_int(-1)
    dist = C.c_double()
    k = int(rng.integers(0, P))
    pos = np.array([tx[k], ty[k]]) + rng.normal(0, 3, 2)
    nq = 1500
    for q in range(nq):
        if q % 97 == 0:                                         # teleport (stale anchor far away) / reset the anchor
            k = int(rng.integers(0, P)); pos = np.array([tx[k], ty[k]]) + rng.normal(0, 12, 2)
            if q % 2: anchor = C.c_int(-1)
        else:
            pos = pos + rng.normal(0, 2.5, 2)                   # a car moves <= 3 m per model step
            if q % 13 == 0: pos = pos + rng.normal(0, 10, 2)    # occasional large excursion off the track
        w = shim.shim_within_anchor(P, tx.ctypes.data_as(dp), ty.ctypes.data_as(dp), tw.ctypes.data_as(dp), float(pos[0]), float(pos[1]),
                                    C.byref(anchor), C.byref(dist))
        rw, rd = oracle.within_track((tx, ty, tw), pos)
        assert bool(w) == rw and abs(dist.value - rd) <= 1e-9 * max(1.0, rd), (name, q, pos, dist.value, rd)
        d2 = (tx - pos[0]) ** 2 + (ty - pos[1]) ** 2              # the anchor left behind is the first minimum (up to exact-tie rounding)
        assert d2[anchor.value] <= d2.min() * (1 + 1e-12) + 1e-12


@pytest.mark.parametrize("name,track", _tracks(), ids=[n for n, _ in _tracks()])
def test_ring_fast_path_matches_findmin_and_general_search(shim, oracle, name, track):
    """The rollout kernels' straight-line paths (ring table; three certified candidates, else five under the wider certificate; no ties) against the oracle's literal findmin AND bit for bit
    against the general anchored search from the same anchor: a rollout's cost must not depend on which of the two a wave took."""
    tx, ty, tw = (np.ascontiguousarray(a, dtype=np.float64) for a in track)
    L = shim
    L.shim_within_ring.argtypes = [C.c_int, dp, dp, dp, C.c_double, C.c_double, C.POINTER(C.c_int), dp, C.POINTER(C.c_int)]
    rng = np.random.default_rng(abs(hash(name)) % 2 ** 31 + 1)
    P = len(tx)
    anchor = C.c_int(-1)
    k = int(rng.integers(0, P))
    pos = np.array([tx[k], ty[k]]) + rng.normal(0, 3, 2)
    nfast = nfast5 = 0
    nq = 1500
    for q in range(nq):
        if q % 97 == 0:
            k = int(rng.integers(0, P)); pos = np.array([tx[k], ty[k]]) + rng.normal(0, 12, 2)
            if q % 2: anchor = C.c_int(-1)
        else:
            pos = pos + rng.normal(0, 2.0, 2)
            if q % 13 == 0: pos = pos + rng.normal(0, 10, 2)
        a_gen, a_ring = C.c_int(anchor.value), C.c_int(anchor.value)
        d_gen, d_ring, fast = C.c_double(), C.c_double(), C.c_int()
        args = (P, tx.ctypes.data_as(dp), ty.ctypes.data_as(dp), tw.ctypes.data_as(dp), float(pos[0]), float(pos[1]))
        w_gen = L.shim_within_anchor(*args, C.byref(a_gen), C.byref(d_gen))
        w_ring = L.shim_within_ring(*args, C.byref(a_ring), C.byref(d_ring), C.byref(fast))
        nfast += fast.value == 1
        nfast5 += fast.value == 2                                # the five-candidate tier (a lane beyond the three-point certificate)
        assert w_ring == w_gen and a_ring.value == a_gen.value and d_ring.value == d_gen.value, (name, q, pos, fast.value)   # bit-identical
        rw, rd = oracle.within_track((tx, ty, tw), pos)
        assert bool(w_ring) == rw and abs(d_ring.value - rd) <= 1e-9 * max(1.0, rd)
        anchor = a_ring
    if P >= 8 and name != "hairpin":
        assert nfast > nq // 3, (name, nfast)                    # the fast path is common even on this random walk with excursions
        assert nfast5 > nq // 50, (name, nfast5)                 # ... and the second tier picks up a good part of what the first one misses
