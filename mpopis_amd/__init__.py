"""mpopis_amd -- MI355X-native MPPI/MPOPI sampling engine behind MPOPIS's policy API.

Product package: HIP kernels + C ABI (csrc/, include/mpopis.h) and the host-side mirror of the
reference's operator interface.  There is no CPU fallback: without libmpopis_hip.so and a visible
MI355X every compute call raises.
"""
from ._lib import MPOPISError  # noqa: F401
from .engine import Engine, default_track  # noqa: F401
from .envs import (CarRacingEnv, CarRacingEnvParams, MultiCarRacingEnv, MountainCarEnv, CartPoleEnv, Track, state, reward,  # noqa: F401
                   action_space, is_terminated, within_track, calculate_β, exceed_β)
from .policies import (MPPI_Policy, GMPPI_Policy, IMPPI_Policy, CEMPPI_Policy, CMAMPPI_Policy, μAISMPPI_Policy,  # noqa: F401
                       μΣAISMPPI_Policy, PMCMPPI_Policy, muAISMPPI_Policy, muSigmaAISMPPI_Policy, get_policy,
                       calculate_trajectory_costs, simulate_model, AbstractGMPPI_Policy, AbstractPathIntegralPolicy)
from .examples import simulate_car_racing, simulate_mountaincar, simulate_cartpole, quantile_ci, shard_trials  # noqa: F401


def block_diagm(A, rep_number):
    """src/utils.jl:9-21"""
    import numpy as np
    A = np.asarray(A, dtype=np.float64)
    if A.ndim == 1:
        return np.diag(np.tile(A, rep_number))
    return np.kron(np.eye(rep_number), A)
