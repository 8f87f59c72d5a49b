"""Independent NumPy re-derivation of the MPOPIS hot path (vectorised over samples, LAPACK-backed).

TEST INFRASTRUCTURE ONLY.  Written separately from oracle/mpopis_oracle.c, directly from the
reference sources, to cross-check the C restatement: the env dynamics are vectorised over K (a
different code shape from the scalar C loop) and the dense linear algebra goes through
numpy.linalg (LAPACK -- what Julia's cholesky / eigen / inv call) instead of the C stand-ins.
Citations are /root/reference paths.  PARITY UNPINNED like the C oracle.
"""
import numpy as np

DEG = np.pi / 180.0


def car_params():
    # src/envs/car_racing.jl:68-93
    return dict(m=2000.0, Izz=3764.0, h=0.3, lf=1.53, lr=1.23, CD0=241.0, CD1=25.1, Caf=150000.0,
                Car=280000.0, muf=0.9, mur=0.9, dmax=18.0 * DEG, ddotmax=90.0 * DEG, Fxmax=7200.0,
                Fxmin=22500.0, lbrake=0.6, ldrive=0.0, blim=45.0 * DEG, dt=0.1, ddt=0.01)


def tire_fy(alpha, mu, Ca, fz, fx):
    # src/envs/car_racing.jl:252-260
    fymax = np.sqrt(np.maximum((mu * fz) ** 2 - fx ** 2, 1e-8))
    ta = np.tan(alpha)
    lin = -Ca * ta + (Ca ** 2 / (3 * fymax)) * np.abs(ta) * ta - (Ca ** 3 / (27 * fymax ** 2)) * ta ** 3
    sat = -fymax * np.sign(alpha)
    return np.where(np.abs(alpha) < np.arctan(3 * fymax / Ca), lin, sat)


def car_step(p, S, A):
    """S: (K,8) states, A: (K,2) actions -> new (K,8).  src/envs/car_racing.jl:282-344"""
    x, y, psi, Vx, Vy, r, d = (S[:, i].copy() for i in range(7))
    tgt = A[:, 0] * p["dmax"] - d
    rate = np.minimum(np.abs(tgt) / p["dt"], p["ddotmax"]) * np.sign(tgt)
    pedal = A[:, 1]
    L = p["lr"] + p["lf"]
    for _ in range(int(round(p["dt"] / p["ddt"]))):
        d = d + rate * p["ddt"]
        af = np.arctan2(Vy + p["lf"] * r, Vx) - d
        ar = np.arctan2(Vy - p["lr"] * r, Vx)
        aero = (p["CD0"] + p["CD1"] * np.abs(Vx)) * np.sign(Vx)
        fx = p["Fxmax"] * np.maximum(pedal, 0.0) + p["Fxmin"] * np.minimum(pedal, 0.0) * np.sign(Vx)
        lam = np.where(pedal <= 0, p["lbrake"], p["ldrive"])
        fxf, fxr = lam * fx, (1 - lam) * fx
        fzf = (p["m"] * p["lr"] * 9.81 - p["h"] * fx) / L     # :262-272 ('f': l_t=l_r, h_cm negated)
        fzr = (p["m"] * p["lf"] * 9.81 + p["h"] * fx) / L
        fyf = tire_fy(af, p["muf"], p["Caf"], fzf, fxf)
        fyr = tire_fy(ar, p["mur"], p["Car"], fzr, fxr)
        rdd = (1 / p["Izz"]) * (p["lf"] * (fxf * np.sin(d) + fyf * np.cos(d)) - p["lr"] * fyr)
        Vyd = (1 / p["m"]) * (fyf * np.cos(d) + fxf * np.sin(d) + fyr) - r * Vx
        Vxd = (1 / p["m"]) * (fxf * np.cos(d) - fyf * np.sin(d) + fxr - aero) + r * Vy
        r = r + rdd * p["ddt"]
        Vx = Vx + Vxd * p["ddt"]
        Vy = Vy + Vyd * p["ddt"]
        psi = psi + r * p["ddt"]
        psi = np.arctan2(np.sin(psi), np.cos(psi))
        x = x + (Vx * np.cos(psi) - Vy * np.sin(psi)) * p["ddt"]
        y = y + (Vx * np.sin(psi) + Vy * np.cos(psi)) * p["ddt"]
    return np.stack([x, y, psi, Vx, Vy, r, d, pedal], axis=1)


def within_track(track, pos):
    """pos (K,2) -> (within (K,), dist (K,)).  src/envs/car_racing_tracks/car_racing_tracks.jl:68-92"""
    tx, ty, tw = track
    P = len(tx)
    d2 = (tx[None, :] - pos[:, :1]) ** 2 + (ty[None, :] - pos[:, 1:2]) ** 2
    i0 = np.argmin(d2, axis=1)                   # first minimum, like findmin
    im, ip = (i0 - 1) % P, (i0 + 1) % P
    dm = np.hypot(tx[im] - pos[:, 0], ty[im] - pos[:, 1])
    dp = np.hypot(tx[ip] - pos[:, 0], ty[ip] - pos[:, 1])
    i2 = np.where(dm <= dp, im, ip)
    p1 = np.stack([tx[i0], ty[i0]], 1)
    p2 = np.stack([tx[i2], ty[i2]], 1)
    t = np.sum((pos - p1) * (p2 - p1), 1) / np.sum((p2 - p1) ** 2, 1)
    proj = p1 + t[:, None] * (p2 - p1)
    dist = np.hypot(proj[:, 0] - pos[:, 0], proj[:, 1] - pos[:, 1])
    return dist < tw[i0], dist


def car_reward(p, track, S):
    # src/envs/car_racing.jl:201-213
    within, dist = within_track(track, S[:, :2])
    beta = np.arctan2(S[:, 4], S[:, 3])
    return (-1e6 * (~within) - 5000.0 * (np.abs(beta) > p["blim"]) - dist + 2.0 * np.hypot(S[:, 3], S[:, 4]))


def multicar_reward(p, track, S, ncars):
    # src/envs/multi-car_racing.jl:145-158
    rew = np.zeros(S.shape[0])
    for i in range(ncars):
        rew += car_reward(p, track, S[:, 8 * i:8 * i + 8])
        for j in range(i + 1, ncars):
            dd = np.hypot(S[:, 8 * j] - S[:, 8 * i], S[:, 8 * j + 1] - S[:, 8 * i + 1])
            rew += -dd - 11000.0 * (dd <= 4.0)
    return rew


def simulate_model(p, track, ncars, x0, Ucur, E, T, gamma=0.0, Sigma_inv=None, U_orig=None):
    """src/mppi_mpopi_policies.jl:261-278 + src/utils.jl:129-144; E is cs x K."""
    cs, K = E.shape
    as_ = 2 * ncars
    V = Ucur[:, None] + E
    cc = np.zeros(K)
    if gamma != 0.0:
        cc = (gamma * U_orig) @ Sigma_inv @ (V - U_orig[:, None])
    ctrl = np.clip(V, -1.0, 1.0).T.reshape(K, T, as_)
    S = np.tile(np.asarray(x0, dtype=float)[None, :], (K, 1))
    cost = np.zeros(K)
    for t in range(T):
        for c in range(ncars):
            S[:, 8 * c:8 * c + 8] = car_step(p, S[:, 8 * c:8 * c + 8], ctrl[:, t, 2 * c:2 * c + 2])
        r = car_reward(p, track, S) if ncars == 1 else multicar_reward(p, track, S, ncars)
        cost -= r
    return cost + cc


def compute_weights(lam, cost):
    # src/utils.jl:79-86
    w = np.exp(-1 / lam * (cost - cost.min()))
    return w / w.sum()


def cma_constants(K, cs, thr):
    # src/mppi_mpopi_policies.jl:513-525
    m, n = K, cs
    m_elite = int(np.round((1.0 - thr) * m))      # np.round is half-to-even like Julia's round
    ws = np.log((m + 1) / 2) - np.log(np.arange(1, m + 1))
    ws[:m_elite] /= ws[:m_elite].sum()
    mu_eff = 1 / np.sum(ws[:m_elite] ** 2)
    cs_ = (mu_eff + 2) / (n + mu_eff + 5)
    ds = 1 + 2 * max(0, np.sqrt((mu_eff - 1) / (n + 1)) - 1) + cs_
    cS = (4 + mu_eff / n) / (n + 4 + 2 * mu_eff / n)
    c1 = 2 / ((n + 1.3) ** 2 + mu_eff)
    cmu = min(1 - c1, 2 * (mu_eff - 2 + 1 / mu_eff) / ((n + 2) ** 2 + mu_eff))
    ws[m_elite:] *= -(1 + c1 / cmu) / ws[m_elite:].sum()
    Ecma = n ** 0.5 * (1 - 1 / (4 * n) + 1 / (21 * n ** 2))
    return dict(m_elite=m_elite, ws=ws, mu_eff=mu_eff, c_sigma=cs_, d_sigma=ds, c_Sigma=cS, c1=c1, c_mu=cmu, E_cma=Ecma)


def alias_table(w):
    # StatsBase.make_alias_table! (recalled)
    n = len(w)
    a = np.asarray(w, dtype=float) * (n / 1.0)
    alias = np.arange(n)
    larges = [i for i in range(n) if a[i] > 1.0]
    smalls = [i for i in range(n) if a[i] < 1.0]
    while larges and smalls:
        s = smalls.pop()
        l = larges.pop()
        alias[s] = l
        a[l] = (a[l] - 1.0) + a[s]
        (larges if a[l] > 1.0 else smalls).append(l)
    for s in smalls:
        a[s] = 1.0
    return a, alias


def policy_call(kind, p, track, ncars, x0, U, Sigma, Z, K, T, lam, N=10, lam_ais=20.0, thr=0.8,
                cma_sigma=0.75, res=None, alpha=1.0):
    """pol(env) for the G-variants (car env) -- returns dict(control, cost, weights, E, U_next, ...).
    Z: (N, K, cs).  Mirrors src/mppi_mpopi_policies.jl:221-238 and :303-817."""
    cs = len(U)
    as_ = 2 * ncars
    gamma = lam * (1 - alpha)
    U_orig = U.copy()
    Uc = U.copy()
    Sig = Sigma.copy()
    sigma = cma_sigma
    if kind == "gmppi":
        N = 1
    cm = cma_constants(K, cs, thr) if kind == "cmamppi" else None
    m_elite = cm["m_elite"] if cm else int(np.round(K * (1 - thr)))
    p_s, p_S = np.zeros(cs), np.zeros(cs)
    ridx = []
    iters = 0
    for n in range(1, N + 1):
        A = sigma ** 2 * Sig if (kind == "cmamppi" and N > 1) else Sig
        L = np.linalg.cholesky(A)
        Sinv = np.linalg.inv(A) if gamma != 0.0 else None
        E = L @ Z[n - 1].T                                   # cs x K
        cost = simulate_model(p, track, ncars, x0, Uc, E, T, gamma, Sinv, U_orig)
        iters = n
        if n < N:
            if kind in ("imppi", "muaismppi", "musigmaaismppi"):
                ws = compute_weights(lam if kind == "imppi" else lam_ais, cost)
                mu = (E @ ws) / ws.sum()
                if kind == "musigmaaismppi":
                    Ec = E - mu[:, None]
                    Sig = (Ec * ws[None, :]) @ Ec.T / ws.sum() + 10e-9 * np.eye(cs)
                Uc = Uc + mu
            elif kind == "pmcmppi":
                ws = compute_weights(lam_ais, cost)
                acc, al = alias_table(ws)
                di, du = res[0][n - 1], res[1][n - 1]
                idx = np.where(du < acc[di], di, al[di])
                ridx.append(idx)
                Ep = E[:, idx]
                mu = Ep.mean(axis=1)
                Sig = np.cov(Ep, ddof=1) + 10e-9 * np.eye(cs)
                Uc = Uc + mu
            else:
                order = np.argsort(cost, kind="stable")
                el = E[:, order[:m_elite]]
                if np.max(np.abs(np.diff(cost[order[:m_elite]]))) < 10e-3:
                    break
                if kind == "cemppi":
                    Sig = np.cov(el, ddof=0) + 10e-9 * np.eye(cs)
                    Uc = Uc + el.mean(axis=1)
                else:
                    ds = el / sigma
                    dw = el @ cm["ws"][:m_elite]
                    Uc = Uc + sigma * dw
                    lamv, Vv = np.linalg.eigh(Sig)
                    Cm = (Vv * lamv ** -0.5) @ Vv.T
                    p_s = (1 - cm["c_sigma"]) * p_s + np.sqrt(cm["c_sigma"] * (2 - cm["c_sigma"]) * cm["mu_eff"]) * (Cm @ dw)
                    sigma = sigma * np.exp(cm["c_sigma"] / cm["d_sigma"] * (np.linalg.norm(p_s) / cm["E_cma"] - 1))
                    h = int(np.linalg.norm(p_s) / np.sqrt(1 - (1 - cm["c_sigma"]) ** (2 * n)) < (1.4 + 2 / (cs + 1)) * cm["E_cma"])
                    p_S = (1 - cm["c_Sigma"]) * p_S + h * np.sqrt(cm["c_Sigma"] * (2 - cm["c_Sigma"]) * cm["mu_eff"]) * dw
                    dsf = ds.flatten(order="F")            # Julia linear indexing is column-major
                    tsum = 0.0
                    for ii in range(K):
                        d = dsf[order[ii]]
                        w0 = cm["ws"][ii] if cm["ws"][ii] >= 0 else n * cm["ws"][ii] / np.linalg.norm(Cm * d) ** 2
                        tsum += w0 * d * d
                    Sig = (1 - cm["c1"] - cm["c_mu"]) * Sig + cm["c1"] * (np.outer(p_S, p_S) + (1 - h) * cm["c_Sigma"] * (2 - cm["c_Sigma"]) * Sig) + cm["c_mu"] * tsum
                    Sig = np.triu(Sig) + np.triu(Sig, 1).T
    E = E + (Uc - U_orig)[:, None]
    w = compute_weights(lam, cost)
    wc = U_orig + E @ w
    control = np.clip(wc[:as_], -1.0, 1.0)
    U_next = U_orig.copy()
    if T > 1:
        U_next[:cs - as_] = wc[as_:]
    else:
        U_next = wc
    return dict(control=control, cost=cost, weights=w, E=E, U_next=U_next, iters_run=iters,
                Sigma_last=A, U_last=Uc, res_idx0=np.array(ridx))
