"""ctypes binding of libmpopis_hip.so (the C ABI in include/mpopis.h).

The product has NO CPU fallback: if the HIP library is missing or no MI355X is visible, every
compute entry point raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPOPIS_HIP_LIB: load another build of the library (julia/MPOPISHip.jl honours the same variable).  The A/B scripts under tools/ab/ point it
# at a variant build instead of overwriting the product binary.
LIB_PATH = os.environ.get("MPOPIS_HIP_LIB") or os.path.join(_HERE, "lib", "libmpopis_hip.so")

ENV_MOUNTAINCAR, ENV_CAR, ENV_CARTPOLE = 0, 1, 2
ENV_IDS = {"mountaincar": ENV_MOUNTAINCAR, "car": ENV_CAR, "cartpole": ENV_CARTPOLE}
POLICY_IDS = {"mppi": 0, "gmppi": 1, "imppi": 2, "cemppi": 3, "cmamppi": 4,
              "μaismppi": 5, "muaismppi": 5, "μΣaismppi": 6, "musigmaaismppi": 6, "pmcmppi": 7}
SIGMA_EST_IDS = {"mle": 0, "ss": 1, "lw": 2, "rblw": 3, "oas": 4}
ERR_ARG, ERR_NOT_PD, ERR_ACTION, ERR_HIP, ERR_NUMERIC = -1, -2, -3, -4, -5
RECORD_LEN = 16

# every symbol include/mpopis.h declares (checked by tests/test_abi.py without a GPU)
ABI_SYMBOLS = [
    "mpopis_abi_version", "mpopis_last_error", "mpopis_create", "mpopis_destroy",
    "mpopis_set_env_params", "mpopis_set_track", "mpopis_set_action_bounds", "mpopis_reset",
    "mpopis_set_state", "mpopis_get_state", "mpopis_set_U", "mpopis_get_U", "mpopis_set_Sigma",
    "mpopis_seed", "mpopis_seed_slots", "mpopis_get_Sigma", "mpopis_rollout_costs", "mpopis_policy_step", "mpopis_env_step",
    "mpopis_env_query", "mpopis_get_trajectories", "mpopis_set_state_noise", "mpopis_run_trials", "mpopis_timing_enable", "mpopis_timing_read",
    "mpopis_timing_reset", "mpopis_bench_policy_steps",
    "mpopis_policy_call", "mpopis_set_overlap", "mpopis_comm_unique_id", "mpopis_comm_init", "mpopis_gather_summary", "mpopis_comm_destroy", "mpopis_comm_count",
]


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("env_kind", C.c_int32), ("num_cars", C.c_int32), ("policy", C.c_int32),
                ("num_samples", C.c_int32), ("horizon", C.c_int32), ("batch", C.c_int32), ("ais_its", C.c_int32),
                ("sigma_est", C.c_int32), ("log_trajectories", C.c_int32),
                ("lambda_", C.c_double), ("alpha", C.c_double), ("lambda_ais", C.c_double),
                ("elite_threshold", C.c_double), ("cma_sigma", C.c_double), ("seed", C.c_uint64)]


class Noise(C.Structure):
    _fields_ = [("Z", C.POINTER(C.c_double)), ("res_i0", C.POINTER(C.c_int32)), ("res_u", C.POINTER(C.c_double))]


class MPOPISError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mpopis error %d: %s" % (code, msg))
        self.code = code


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def lib():
    """Load the HIP engine; fail loudly when it is missing (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: run `python -m mpopis_amd.build`%s (the engine has no CPU fallback)"
                              % (LIB_PATH, " or unset MPOPIS_HIP_LIB" if os.environ.get("MPOPIS_HIP_LIB") else ""))
        L = C.CDLL(LIB_PATH)
        H = C.c_void_p
        L.mpopis_abi_version.restype = C.c_int
        L.mpopis_last_error.restype = C.c_char_p
        L.mpopis_last_error.argtypes = [H]
        L.mpopis_create.argtypes = [C.POINTER(Config), C.POINTER(H)]
        L.mpopis_destroy.argtypes = [H]
        L.mpopis_destroy.restype = None
        L.mpopis_set_env_params.argtypes = [H, _dp, C.c_int32]
        L.mpopis_set_track.argtypes = [H, _dp, _dp, _dp, C.c_int32]
        L.mpopis_set_action_bounds.argtypes = [H, _dp, _dp]
        L.mpopis_reset.argtypes = [H]
        L.mpopis_set_state.argtypes = [H, _dp, _ip, _ip]
        L.mpopis_get_state.argtypes = [H, _dp, _ip, _ip]
        L.mpopis_set_U.argtypes = [H, _dp]
        L.mpopis_get_U.argtypes = [H, _dp]
        L.mpopis_set_Sigma.argtypes = [H, _dp, C.c_int32]
        L.mpopis_seed.argtypes = [H, C.c_uint64]
        L.mpopis_seed_slots.argtypes = [H, C.POINTER(C.c_uint64)]
        L.mpopis_get_Sigma.argtypes = [H, _dp]
        L.mpopis_rollout_costs.argtypes = [H, _dp, _dp, _dp, _dp, _dp, _dp]
        L.mpopis_policy_step.argtypes = [H, C.POINTER(Noise), _dp, _dp, _dp, _dp, _ip, _ip]
        L.mpopis_policy_call.argtypes = [H, _dp, _ip, _ip, _dp, C.POINTER(Noise), _dp, _dp, _dp, _ip]
        L.mpopis_env_step.argtypes = [H, _dp, _dp]
        L.mpopis_env_query.argtypes = [H, _dp, _ip, _dp, _dp]
        L.mpopis_get_trajectories.argtypes = [H, _dp]
        L.mpopis_set_state_noise.argtypes = [H, C.c_double, C.c_double, C.c_double]
        L.mpopis_run_trials.argtypes = [H, C.c_int32, C.c_int32, _dp, _dp]
        L.mpopis_timing_enable.argtypes = [H, C.c_int32]
        L.mpopis_timing_read.argtypes = [H, C.c_char_p, C.c_int32, _dp, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        L.mpopis_timing_reset.argtypes = [H]
        L.mpopis_bench_policy_steps.argtypes = [H, C.c_int32, _dp, _dp]
        L.mpopis_set_overlap.argtypes = [H, C.c_int32]
        L.mpopis_comm_unique_id.argtypes = [C.c_char_p]
        L.mpopis_comm_init.argtypes = [H, C.c_char_p, C.c_int32, C.c_int32]
        L.mpopis_gather_summary.argtypes = [H, _dp, C.c_int32, C.c_int32, _dp, _ip]
        L.mpopis_comm_destroy.argtypes = [H]
        L.mpopis_comm_count.argtypes = [H, _ip]
        _lib = L
    return _lib


def check(h, rc):
    if rc != 0:
        msg = lib().mpopis_last_error(h)
        raise MPOPISError(rc, msg.decode() if msg else "")
