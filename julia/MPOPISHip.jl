# MPOPISHip.jl -- the reference-side binding for libmpopis_hip.so (include/mpopis.h).
#
# EXPERIMENTAL, SOURCE ONLY: Julia is not installed in the build image, so this file has never been executed
# there (expect to fix typos on first load); it is the thin `ccall` layer a MPOPIS maintainer adds.  It plugs in exactly where the
# reference already dispatches on the env type for its EnvPool backend
# (src/mppi_mpopi_policies.jl:148 vs :186, :240 vs :261): more specific methods of
#     (pol::AbstractPathIntegralPolicy)(env)         -> mpopis_policy_call (one ccall, one host wait per MPC step)
#     simulate_model(pol, env, E, Σ_inv, U_orig)     -> mpopis_rollout_costs
# for env::CarRacingEnv / MultiCarRacingEnv / MountainCarEnv / CartPoleEnv.  Every other env type keeps falling
# through to the Julia CPU methods unchanged.  Policy symbols, constructors, `get_policy`,
# `seed!`, `pol.U`, `pol.Σ`, `pol.logger` are untouched.
module MPOPISHip

using MPOPIS
using MPOPIS: AbstractPathIntegralPolicy, AbstractGMPPI_Policy, MPPI_Policy, GMPPI_Policy, IMPPI_Policy,
              CEMPPI_Policy, CMAMPPI_Policy, μAISMPPI_Policy, μΣAISMPPI_Policy, PMCMPPI_Policy,
              CarRacingEnv, MultiCarRacingEnv
using ReinforcementLearning: MountainCarEnv, CartPoleEnv
import MPOPIS: simulate_model

const LIB = get(ENV, "MPOPIS_HIP_LIB", joinpath(@__DIR__, "..", "mpopis_amd", "lib", "libmpopis_hip.so"))

function __init__()
    v = ccall((:mpopis_abi_version, LIB), Cint, ())
    v >= 3 || error("libmpopis_hip.so speaks ABI version $v; this binding needs >= 3 (mpopis_policy_call)")
end

# mirrors `mpopis_config` (include/mpopis.h) field for field
struct Config
    device::Int32; env_kind::Int32; num_cars::Int32; policy::Int32
    num_samples::Int32; horizon::Int32; batch::Int32; ais_its::Int32
    sigma_est::Int32; log_trajectories::Int32
    lambda::Float64; alpha::Float64; lambda_ais::Float64; elite_threshold::Float64; cma_sigma::Float64
    seed::UInt64
end

policy_id(::MPPI_Policy) = 0;  policy_id(::GMPPI_Policy) = 1;  policy_id(::IMPPI_Policy) = 2
policy_id(::CEMPPI_Policy) = 3; policy_id(::CMAMPPI_Policy) = 4; policy_id(::μAISMPPI_Policy) = 5
policy_id(::μΣAISMPPI_Policy) = 6; policy_id(::PMCMPPI_Policy) = 7

env_kind(::MountainCarEnv) = (0, 0)
env_kind(::CartPoleEnv) = (2, 0)
env_kind(::CarRacingEnv) = (1, 1)
env_kind(env::MultiCarRacingEnv) = (1, env.N)

ais_its(pol) = hasproperty(pol, :opt_its) ? pol.opt_its : 1
lam_ais(pol) = hasproperty(pol, :λ_ais) ? pol.λ_ais : 0.0
elite(pol::CEMPPI_Policy) = pol.ce_elite_threshold
elite(pol::CMAMPPI_Policy) = 1.0 - pol.m_elite / pol.params.num_samples
elite(pol) = 0.8
cma_sigma(pol::CMAMPPI_Policy) = pol.σ
cma_sigma(pol) = 1.0
# MPOPIS_SIGMA_EST_*: CEMPPI_Policy stores the estimator object, not the symbol (src/mppi_mpopi_policies.jl:386,414-426)
# Told apart by type NAME (SimpleCovariance / LinearShrinkage are CovarianceEstimation.jl types; going by name needs neither that package in
# this module's environment nor MPOPIS re-exporting them).
sigma_est(pol::CEMPPI_Policy) = sigma_est_id(pol.Σ_estimation_method)
sigma_est(pol) = 0
function sigma_est_id(m)
    nameof(typeof(m)) === :SimpleCovariance && return 0                                   # :mle
    nameof(typeof(m)) === :LinearShrinkage || error("MPOPISHip: unknown Σ estimation method $(typeof(m))")
    return Dict(:ss => 1, :lw => 2, :rblw => 3, :oas => 4)[m.shrinkage]
end

car_param_vector(env::CarRacingEnv) = Float64[getfield(env.params, f) for f in fieldnames(typeof(env.params))] |>
                                      v -> vcat(v, env.dt, env.δt)
car_param_vector(env::MultiCarRacingEnv) = car_param_vector(env.envs[1])
track_of(env::CarRacingEnv) = env.track
track_of(env::MultiCarRacingEnv) = env.envs[1].track

check(h, rc) = rc == 0 ? nothing : error(unsafe_string(ccall((:mpopis_last_error, LIB), Cstring, (Ptr{Cvoid},), h)))

# One engine handle per policy object.  Weak keys: the table must not keep a policy alive (its finalizer destroys the handle; an IdDict
# entry would pin the policy for ever, and the finalizer would never run).  Policies are `mutable struct`s, so keys compare by identity.
const HANDLES = WeakKeyDict{Any,Ptr{Cvoid}}()
# seed!(pol, s) issued BEFORE the first pol(env) -- the order the reference harness uses (src/examples/car_example.jl:187-188 precede :205) --
# is kept here and consumed when the handle is created, so the harness seed reaches the device streams.
const PENDING_SEED = WeakKeyDict{Any,UInt64}()

# Device seed of a policy nobody seeded: taken from a COPY of pol.rng (deterministic given the policy's own rng state, and pol.rng is not advanced).
default_seed(pol) = rand(copy(pol.rng), UInt64)

function handle(pol, env)
    get!(HANDLES, pol) do
        kind, ncars = env_kind(env)
        seed0 = haskey(PENDING_SEED, pol) ? pop!(PENDING_SEED, pol) : default_seed(pol)
        cfg = Config(0, kind, ncars, policy_id(pol), pol.params.num_samples, pol.params.horizon, 1, ais_its(pol),
                     sigma_est(pol), pol.params.log, pol.params.λ, pol.params.α, lam_ais(pol), elite(pol), cma_sigma(pol), seed0)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:mpopis_create, LIB), Cint, (Ref{Config}, Ref{Ptr{Cvoid}}), cfg, out)
        rc == 0 || error(unsafe_string(ccall((:mpopis_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL)))
        h = out[]
        if kind == 1
            p = car_param_vector(env); tr = track_of(env)
            check(h, ccall((:mpopis_set_env_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32), h, p, length(p)))
            check(h, ccall((:mpopis_set_track, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32),
                           h, tr.x′, tr.y′, tr.lane_width′, length(tr.x′)))
        elseif kind == 2
            pr = env.params
            p = Float64[pr.gravity, pr.masscart, pr.masspole, pr.totalmass, pr.halflength, pr.polemasslength, pr.forcemag,
                        pr.dt, pr.thetathreshold, pr.xthreshold, pr.max_steps]
            check(h, ccall((:mpopis_set_env_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32), h, p, length(p)))
        else
            pr = env.params
            p = Float64[pr.min_pos, pr.max_pos, pr.max_speed, pr.goal_pos, pr.goal_velocity, pr.power, pr.gravity, pr.max_steps]
            check(h, ccall((:mpopis_set_env_params, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32), h, p, length(p)))
        end
        Σ = Matrix{Float64}(pol.Σ)                       # column-major, as×as (:mppi) or cs×cs
        check(h, ccall((:mpopis_set_Sigma, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32), h, Σ, size(Σ, 1)))
        finalizer(_ -> ccall((:mpopis_destroy, LIB), Cvoid, (Ptr{Cvoid},), h), pol)
        h
    end
end

env_t(env) = hasproperty(env, :t) ? Int32(env.t) : Int32(0)

# ---- control = pol(env) ------------------------------------------------------------------------------
function hip_policy_call(pol::AbstractPathIntegralPolicy, env)
    h = handle(pol, env)
    K, as = pol.params.num_samples, pol.params.as
    x = Vector{Float64}(MPOPIS.state(env)); t = Int32[env_t(env)]; done = Int32[env.done]
    want_log = pol.params.log                              # costs / weights are only copied back when the logger wants them
    control = Vector{Float64}(undef, as); cost = Vector{Float64}(undef, want_log ? K : 0); w = Vector{Float64}(undef, want_log ? K : 0)
    # ONE ccall, one host wait: state in, pol.U in and (rolled in place, same array as params.U₀) out, control / cost / weights out
    GC.@preserve x t done control cost w begin
        check(h, ccall((:mpopis_policy_call, LIB), Cint,
                       (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                       h, x, t, done, pol.U, C_NULL, control, want_log ? pointer(cost) : C_NULL, want_log ? pointer(w) : C_NULL, C_NULL))
    end
    if pol.params.log
        pol.logger.traj_costs = cost; pol.logger.traj_weights = w
        # pol.logger.trajectories[k][t, :] = env.state (src/utils.jl:139-141): K matrices of T x ss, which is exactly the
        # per-sample block layout mpopis_get_trajectories returns (column-major T x ss per sample)
        T, ss = pol.params.horizon, pol.params.ss
        buf = Vector{Float64}(undef, K * T * ss)
        GC.@preserve buf check(h, ccall((:mpopis_get_trajectories, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), h, buf))
        for k in 1:K
            copyto!(pol.logger.trajectories[k], 1, buf, (k - 1) * T * ss + 1, T * ss)
        end
    end
    return as == 1 ? control : reshape(control, as, 1)        # get_model_controls returns as×1 (utils.jl:55-67)
end

for E in (:CarRacingEnv, :MultiCarRacingEnv, :MountainCarEnv, :CartPoleEnv)
    @eval (pol::MPPI_Policy)(env::$E) = hip_policy_call(pol, env)
    @eval (pol::AbstractGMPPI_Policy)(env::$E) = hip_policy_call(pol, env)
end

# ---- trajectory_cost = simulate_model(pol, env, E, Σ_inv, U_orig) ----------------------------------------
function hip_simulate_model(pol::AbstractGMPPI_Policy, env, E::Matrix{Float64}, Σ_inv::Matrix{Float64}, U_orig::Vector{Float64})
    h = handle(pol, env)
    cost = Vector{Float64}(undef, pol.params.num_samples)
    x = Vector{Float64}(MPOPIS.state(env))
    γ = pol.params.λ * (1 - pol.params.α)
    GC.@preserve x E Σ_inv U_orig cost begin
        check(h, ccall((:mpopis_rollout_costs, LIB), Cint,
                       (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                       h, x, pol.U, U_orig, E, γ == 0 ? C_NULL : pointer(Σ_inv), cost))
    end
    return cost
end
for E in (:CarRacingEnv, :MultiCarRacingEnv, :MountainCarEnv, :CartPoleEnv)
    @eval simulate_model(pol::AbstractGMPPI_Policy, env::$E, E::Matrix{Float64}, Σ_inv::Matrix{Float64}, U_orig::Vector{Float64}) =
        hip_simulate_model(pol, env, E, Σ_inv, U_orig)
end

# Random.seed!(pol, seed) also reseeds the device streams.  NOTE for the maintainer: once a policy is bound to a handle, E is drawn on the device
# from Philox4x32-10 streams keyed by this seed (slot b of a handle draws from seed + b, hence `seed - 1` below for the one-slot handles this
# binding creates: mpopis_seed / mpopis_create give slot 0 the key `arg + 1`), NOT from `pol.rng`; `pol.rng` is still seeded so that code which
# reads it directly (state noise in the example harness) behaves as before.  Results therefore differ from a CPU run with the same seed in the
# draws, not in the algorithm -- pass explicit noise (`mpopis_noise`) to compare the two paths number for number.
# A seed that arrives before the handle exists (the harness order: construct, seed!, first pol(env)) is parked in PENDING_SEED and becomes the
# handle's creation seed; tests/abi_client.c runs the same three-step order against the C ABI (lazy handle, parked seed) and checks that two
# runs give identical controls.
# The method is MORE SPECIFIC than the reference's `Random.seed!(pol::AbstractPathIntegralPolicy, seed)` (src/MPOPIS.jl:54: concrete policy
# types, `seed::Integer`), so nothing is overwritten (method overwriting is an error during precompilation).
const BoundPolicy = Union{MPPI_Policy, GMPPI_Policy, IMPPI_Policy, CEMPPI_Policy, CMAMPPI_Policy, μAISMPPI_Policy, μΣAISMPPI_Policy, PMCMPPI_Policy}
function MPOPIS.seed!(pol::BoundPolicy, seed::Integer)
    MPOPIS.Random.seed!(pol.rng, seed)
    s = (seed - 1) % UInt64                                # wraps like the C side's uint64_t arithmetic
    if haskey(HANDLES, pol)
        check(HANDLES[pol], ccall((:mpopis_seed, LIB), Cint, (Ptr{Cvoid}, UInt64), HANDLES[pol], s))
    else
        PENDING_SEED[pol] = s
    end
    pol
end

end # module
