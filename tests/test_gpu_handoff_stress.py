"""The last-workgroup hand-offs (kernels_select.hip: elite early break of the chip-wide rank sorts; kernels_invsqrt.hip: trace partials -> Lanczos preparation)
publish their payload with agent-scope stores + a store-acknowledge wait and take an ACQUIRE ticket -- no release fence.  A stale read there would flip the
early break (a different iteration count) or feed the Lanczos run wrong spectrum bounds.  Stress: a :cmamppi handle whose sort and trace run as multi-workgroup
launches spread over the XCDs (K = 4096 at <= 2 slots: k_sortperm_rank_multi; K = 9000: k_sortperm_rank_big; cs = 180: six trace workgroups per slot) is
driven through a 30-step closed loop alone, then through the same 30 steps again WHILE a second handle saturates the chip with rollout / sampler traffic on its own streams (dirty lines in
every XCD's L2).  Same seeds, so iteration counts, controls and costs must be bit-identical between the quiet and the loaded run."""
import threading
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ncars,K,T,B", [(3, 4096, 30, 2), (1, 9000, 20, 1)])
def test_handoffs_are_bit_stable_beside_heavy_traffic(ncars, K, T, B):
    from mpopis_amd import build
    build.build()
    from mpopis_amd.engine import Engine

    def run(steps, loaded):
        eng = Engine("car", ncars, "cmamppi", K, T, batch=B, lam=10.0, ais_its=4, elite_threshold=0.8, cma_sigma=0.75, cov=np.tile([0.0625, 0.1], ncars), seed=31337)
        stop = threading.Event()
        th = None
        if loaded:
            noise = Engine("car", 1, "musigmaaismppi", 4096, 50, batch=32, lam=10.0, ais_its=6, lam_ais=20.0, cov=[0.0625, 0.1], seed=7)

            def churn():                                      # ctypes releases the GIL during the call: the two handles really overlap on the device
                while not stop.is_set():
                    noise.bench_policy_steps(2)
            th = threading.Thread(target=churn, daemon=True)
            th.start()
        out = []
        try:
            for _ in range(steps):
                try:
                    g = eng.policy_step(None)
                    out.append((g["iters_run"].copy(), g["control"].copy(), g["cost"].copy()))
                    eng.env_step(g["control"])                # closed loop: :cmamppi's Σ update needs the state to move
                except Exception as e:                        # the reference's own PosDefException may end a :cmamppi run: then both runs must end alike
                    out.append(("error", getattr(e, "code", None)))
                    break
        finally:
            stop.set()
            if th is not None:
                th.join()
                noise.close()
            eng.close()
        return out

    quiet = run(30, False)
    loaded = run(30, True)
    print("\n[hand-off stress] cars=%d K=%d cs=%d: %d closed-loop steps compared%s" % (ncars, K, 2 * ncars * T, len(quiet), " (ended by the reference's own error)" if isinstance(quiet[-1][0], str) else ""))
    assert len(quiet) == len(loaded) and len(quiet) >= 5
    for s, (a, b) in enumerate(zip(quiet, loaded)):
        if isinstance(a[0], str) or isinstance(b[0], str):
            assert isinstance(a[0], str) and isinstance(b[0], str) and a[1] == b[1], (s, a, b)
            continue
        assert np.array_equal(a[0], b[0]), (s, a[0], b[0])
        assert np.array_equal(a[1], b[1]), s
        assert np.array_equal(a[2], b[2]), s
