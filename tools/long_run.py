import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpopis_amd as M
for pt in [":mppi", ":gmppi", ":imppi", ":cemppi", ":cmamppi", ":μaismppi", ":μΣaismppi", ":pmcmppi"]:
    t = time.time()
    try:
        rec, summ = M.simulate_car_racing(num_trials=16, num_steps=300, policy_type=pt, num_samples=375, horizon=50, ais_its=10, seed=42, quiet=True)
        print(pt, "ok %.1fs" % (time.time() - t), "steps med %.0f" % np.median(rec[:, 2]), "laps-t1 med %.0f" % np.median(rec[:, 4]), "viol T/B/C", rec[:, 13].sum(), rec[:, 12].sum(), rec[:, 14].sum(), "status", rec[:, 16].min(), "rew/step %.1f" % np.mean(rec[:, 3]))
    except Exception as e:
        print(pt, "EXC", repr(e)[:200])
rec, summ = M.simulate_car_racing(num_trials=8, num_steps=150, num_cars=3, policy_type=":cemppi", num_samples=512, horizon=50, seed=3, quiet=True)
print("3 cars cemppi", "steps", rec[:, 2], "status", rec[:, 16].min())
rec, _ = M.simulate_mountaincar(num_trials=8, num_steps=200, policy_type=":cemppi", seed=5, quiet=True); print("mountaincar steps", rec[:, 1])
rec, _ = M.simulate_cartpole(num_trials=8, num_steps=200, policy_type=":cemppi", seed=5, quiet=True); print("cartpole steps", rec[:, 1])
# the benched configuration through the closed loop, large batch (two-part schedule active), 2 laps
t = time.time()
rec, summ = M.simulate_car_racing(num_trials=64, num_steps=200, policy_type=":μΣaismppi", num_samples=4096, horizon=50, ais_its=10, seed=20240000, quiet=True)
print("C5 closed loop 64 trials x 200 steps: %.1fs" % (time.time() - t), "steps med %.0f" % np.median(rec[:, 2]), "lap1 med %.0f" % np.median(rec[:, 4]),
      "viol T/B", rec[:, 13].sum(), rec[:, 12].sum(), "status", rec[:, 16].min(), "rollouts %.3e" % rec[:, 15].sum(), "rew/step %.1f" % np.mean(rec[:, 3]))
t = time.time()
try:    # long CMA closed loops can end in the reference's own PosDefException (negative-weight Σ update, :598)
    rec, summ = M.simulate_car_racing(num_trials=8, num_steps=30, num_cars=3, policy_type=":cmamppi", num_samples=4096, horizon=50, ais_its=10, seed=11, quiet=True)
    print("C4 closed loop 8 trials x 30 steps: %.1fs" % (time.time() - t), "steps", rec[:, 2], "status", rec[:, 16].min())
except Exception as e:
    print("C4 closed loop:", repr(e)[:160])
