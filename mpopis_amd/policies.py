"""Host-side mirror of the reference's policy API (src/mppi_mpopi_policies.jl) over the HIP engine.

    pol = GMPPI_Policy(env, num_samples=1024, horizon=50, λ=10.0, cov_mat=[0.0625, 0.1])
    act = pol(env)            # src/mppi_mpopi_policies.jl:121 / :221  -> one launch sequence on the MI355X
    env(act)

Same constructor keywords, same error behaviour (MPOPISError codes mirror error()/PosDefException),
same semantics including the reference's quirks (aliasing of U/U0 in get_controls_roll_U!, CMA's
scalar rank-µ term).  Randomness: the reference's MersenneTwister stream is not reproduced; a policy
draws from per-policy Philox4x32-10 streams seeded with seed!(pol, seed), or consumes injected
standard normals (`pol(env, Z=...)`) for results-parity tests.
"""
import numpy as np

from .engine import Engine, _f64
from ._lib import MPOPISError, ERR_ARG


class MPPI_Logger:
    """src/mppi_mpopi_policies.jl:2-6"""

    def __init__(self):
        self.trajectories, self.traj_costs, self.traj_weights = None, None, None


class Information_Theoretic:
    def __init__(self, λ):
        self.λ = λ


class _Params:
    pass


class AbstractPathIntegralPolicy:
    _kind = None

    def __init__(self, env, num_samples=50, horizon=50, λ=1.0, α=1.0, U0=(0.0,), cov_mat=(1.0,), weight_method="IT",
                 elite_threshold=0.8, rng=None, log=False, seed=0, device=0, _extra=None):
        if weight_method != "IT":
            raise MPOPISError(ERR_ARG, "No cost method implemented for %s" % weight_method)       # :83-90
        extra = _extra or {}
        as_ = env.as_
        U0 = _f64(U0).reshape(-1)
        if U0.size == as_:
            U0 = np.tile(U0, horizon)                                                               # :61-63
        if U0.size != as_ * horizon:
            raise MPOPISError(ERR_ARG, "U0 must be length of action space or control space")       # :64
        cov = np.asarray(cov_mat, dtype=np.float64)
        self.env = env
        self._eng = Engine(env.kind, env.ncars, self._kind, num_samples=num_samples, horizon=horizon, batch=1, lam=λ,
                           alpha=α, seed=seed, device=device, log_trajectories=log,
                           track=env.track.arrays() if env.kind == "car" else None, env_params=env._param_vector(),
                           cov=cov, U0=U0, **extra)
        p = _Params()
        p.num_samples, p.horizon, p.λ, p.α, p.U0 = num_samples, horizon, λ, α, U0
        p.ss, p.as_, p.cs, p.log = env.ss, as_, as_ * horizon, log
        p.weight_method = Information_Theoretic(λ)
        self.params = p
        self.rng = rng
        self.logger = MPPI_Logger()
        full = np.diag(cov) if cov.ndim == 1 else cov
        rep = 1 if self._kind == "mppi" else horizon
        self.Σ = np.kron(np.eye(rep), full) if full.shape[0] == as_ and not (self._kind != "mppi" and as_ == p.cs) else full

    # pol.U lives on the device; expose it like the reference field
    @property
    def U(self):
        return self._eng.get_U()[0]

    @U.setter
    def U(self, v):
        self._eng.set_U(_f64(v)[None])

    def seed(self, seed):
        """Random.seed!(pol, seed): src/MPOPIS.jl:54"""
        self._eng.seed(int(seed) - 1)          # slot 0 draws from seed+0+1

    def __call__(self, env, Z=None, res_i0=None, res_u=None, return_info=False):
        """control = pol(env)."""
        if Z is None and not return_info:
            # the production call: one ABI call, one host wait (mpopis_policy_call); pol.U stays resident and is rolled on the device
            out = self._eng.policy_call(env.state[None], [env.t], [int(env.done)], None, want_cost=self.params.log)
            if self.params.log:
                self.logger.traj_costs = out["cost"][0]
                self.logger.traj_weights = out["weights"][0]
                self.logger.trajectories = list(self._eng.get_trajectories()[0])
            return out["control"][0].copy()
        self._eng.set_state(env.state[None], [env.t], [int(env.done)])
        Zb = None if Z is None else _f64(Z)[None]
        ri = None if res_i0 is None else np.asarray(res_i0)[None]
        ru = None if res_u is None else _f64(res_u)[None]
        out = self._eng.policy_step(Zb, ri, ru, want_E=return_info, minimal=not (return_info or self.params.log))
        if self.params.log:                                                                         # :140-143,:233-236
            self.logger.traj_costs = out["cost"][0]
            self.logger.traj_weights = out["weights"][0]
            self.logger.trajectories = list(self._eng.get_trajectories()[0])
        control = out["control"][0].copy()
        if return_info:
            return control, {k: v[0] for k, v in out.items()}
        return control

    def close(self):
        self._eng.close()


def calculate_trajectory_costs(pol, env, Z=None, res_i0=None, res_u=None):
    """(trajectory_cost, E, weights) = calculate_trajectory_costs(pol, env)  (e.g. :303,:434,:532,:644,:709,:782).
    Note: like the functor, this advances pol.U (the engine fuses the roll); save/restore pol.U to undo."""
    _, info = pol(env, Z=Z, res_i0=res_i0, res_u=res_u, return_info=True)
    E = info["E"]
    return info["cost"], (E if pol._kind == "mppi" else E.T), info["weights"]


def simulate_model(pol, env, E, Σ_inv=None, U_orig=None):
    """simulate_model(pol, env, E, Σ_inv, U_orig) -> trajectory_cost (:261-278); E is cs x K."""
    E = np.asarray(E, dtype=np.float64)
    U = pol.U
    return pol._eng.rollout_costs(U[None], E.T[None], x0=env.state[None], U_orig=None if U_orig is None else _f64(U_orig)[None],
                                  Sigma_inv=Σ_inv)[0]


class MPPI_Policy(AbstractPathIntegralPolicy):
    _kind = "mppi"


class AbstractGMPPI_Policy(AbstractPathIntegralPolicy):
    pass


class GMPPI_Policy(AbstractGMPPI_Policy):
    _kind = "gmppi"


class IMPPI_Policy(AbstractGMPPI_Policy):
    _kind = "imppi"

    def __init__(self, env, opt_its=10, **kw):
        super().__init__(env, _extra=dict(ais_its=opt_its), **kw)
        self.opt_its = opt_its


class CEMPPI_Policy(AbstractGMPPI_Policy):
    _kind = "cemppi"

    def __init__(self, env, opt_its=10, ce_elite_threshold=0.8, Σ_est="mle", **kw):
        Σ_est = str(Σ_est).lstrip(":")
        if Σ_est not in ("mle", "lw", "ss", "rblw", "oas"):
            raise MPOPISError(ERR_ARG, "CEMPPI_Policy - Not a valid Σ estimation method")          # :425
        super().__init__(env, _extra=dict(ais_its=opt_its, elite_threshold=ce_elite_threshold, sigma_est=Σ_est), **kw)
        self.opt_its, self.ce_elite_threshold = opt_its, ce_elite_threshold


class CMAMPPI_Policy(AbstractGMPPI_Policy):
    _kind = "cmamppi"

    def __init__(self, env, opt_its=10, σ=1.0, elite_perc_threshold=0.8, **kw):
        super().__init__(env, _extra=dict(ais_its=opt_its, cma_sigma=σ, elite_threshold=elite_perc_threshold), **kw)
        self.opt_its, self.σ = opt_its, σ


class μAISMPPI_Policy(AbstractGMPPI_Policy):
    _kind = "μaismppi"

    def __init__(self, env, opt_its=10, λ_ais=20.0, **kw):
        super().__init__(env, _extra=dict(ais_its=opt_its, lam_ais=λ_ais), **kw)
        self.opt_its, self.λ_ais = opt_its, λ_ais


class μΣAISMPPI_Policy(AbstractGMPPI_Policy):
    _kind = "μΣaismppi"

    def __init__(self, env, opt_its=10, λ_ais=20.0, **kw):
        super().__init__(env, _extra=dict(ais_its=opt_its, lam_ais=λ_ais), **kw)
        self.opt_its, self.λ_ais = opt_its, λ_ais


class PMCMPPI_Policy(AbstractGMPPI_Policy):
    _kind = "pmcmppi"

    def __init__(self, env, opt_its=10, λ_ais=20.0, **kw):
        super().__init__(env, _extra=dict(ais_its=opt_its, lam_ais=λ_ais), **kw)
        self.opt_its, self.λ_ais = opt_its, λ_ais


# ASCII aliases
muAISMPPI_Policy = μAISMPPI_Policy
muSigmaAISMPPI_Policy = μΣAISMPPI_Policy


def seed_(obj, seed):
    """seed!(pol, seed) / seed!(env, seed)"""
    if hasattr(obj, "seed"):
        obj.seed(seed)


def get_policy(policy_type, env, num_samples, horizon, λ, α, U0, cov_mat, pol_log, ais_its, λ_ais,
               ce_elite_threshold, ce_Σ_est, cma_σ, cma_elite_threshold, **kw):
    """src/examples/example_utils.jl:12-130 -- same positional signature, symbols as ':cemppi' or 'cemppi'."""
    pt = str(policy_type).lstrip(":")
    common = dict(num_samples=num_samples, horizon=horizon, λ=λ, α=α, U0=U0, cov_mat=cov_mat, log=pol_log, **kw)
    if pt == "mppi":
        return MPPI_Policy(env, **common)
    if pt == "gmppi":
        return GMPPI_Policy(env, **common)
    if pt == "imppi":
        return IMPPI_Policy(env, opt_its=ais_its, **common)
    if pt == "cemppi":
        return CEMPPI_Policy(env, opt_its=ais_its, ce_elite_threshold=ce_elite_threshold, Σ_est=ce_Σ_est, **common)
    if pt == "cmamppi":
        return CMAMPPI_Policy(env, opt_its=ais_its, σ=cma_σ, elite_perc_threshold=cma_elite_threshold, **common)
    if pt in ("μΣaismppi", "musigmaaismppi"):
        return μΣAISMPPI_Policy(env, opt_its=ais_its, λ_ais=λ_ais, **common)
    if pt in ("μaismppi", "muaismppi"):
        return μAISMPPI_Policy(env, opt_its=ais_its, λ_ais=λ_ais, **common)
    if pt == "pmcmppi":
        return PMCMPPI_Policy(env, opt_its=ais_its, λ_ais=λ_ais, **common)
    raise MPOPISError(ERR_ARG, "No policy_type of %s" % policy_type)                                # :127
