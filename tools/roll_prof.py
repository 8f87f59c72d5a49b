"""dev: distribution of the per-wave run times of the last 1-car rollout launch (needs tools/ab/libprof.so copied over the library, see roll_prof.sh)"""
import sys, os, ctypes as C; sys.path.insert(0, os.getcwd())
import numpy as np
from mpopis_amd.engine import Engine
from mpopis_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = Engine("car", 1, "musigmaaismppi", 4096, 50, batch=B, lam=10.0, ais_its=10, lam_ais=20.0, cov=[0.0625, 0.1], seed=20240000)
eng.bench_policy_steps(3)
L = _lib.lib()
buf = (C.c_ulonglong * (3 * 8192))()
L.mpopis_debug_roll_prof.argtypes = [C.c_void_p]
for rep in range(3):
    eng.bench_policy_steps(1)
    L.mpopis_debug_roll_prof(buf)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 3)[: B * 64]
    t0, t1, hw = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2]
    base = t0.min()
    st, en = (t0 - base) / 100.0, (t1 - base) / 100.0          # us
    dur = en - st
    print("launch span %.1f us | start: max %.1f us | end: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f | dur: min %.1f median %.1f max %.1f" % (
        en.max(), st.max(), en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max(), dur.min(), np.median(dur), dur.max()))
    xcc = (hw >> np.uint64(32)).astype(int) & 0xf
    hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7; simd = (hwid >> 4) & 0x3
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simdkey = key * 4 + simd
    wid = hwid & 0xf
    sets = {}
    for sk, w in zip(simdkey, wid): sets.setdefault(int(sk), []).append(int(w))
    from collections import Counter
    print("  wave-slot sets per SIMD:", Counter(tuple(sorted(v)) for v in sets.values()).most_common(4))
    u, cnt = np.unique(simdkey, return_counts=True)
    print("  SIMDs used %d, waves per SIMD: min %d max %d ; CUs used %d" % (len(u), cnt.min(), cnt.max(), len(np.unique(key))))
    print("  per-XCC median end:", [round(float(np.median(en[xcc == x])), 1) for x in range(8)])
    # idle estimate: for each SIMD, the time of its last wave end vs launch span
    last = np.array([en[simdkey == s].max() for s in u])
    print("  SIMD last-end: min %.1f median %.1f max %.1f ; mean(last)/span = %.3f" % (last.min(), np.median(last), last.max(), last.mean() / en.max()))
eng.close()
