# gen_golden.jl -- PINS THE ORACLE: run it wherever Julia + MPOPIS (v0.2.0) exist, commit the JSON files it writes to
# tests/golden/julia_*.json, and `pytest tests/test_julia_golden.py` checks oracle/mpopis_oracle.c against the REAL reference
# (that test skips loudly while the files are absent).  Julia is not installed in the build image, so this script has not
# been executed there; it only uses the reference's public entry points:
#     get_policy (src/examples/example_utils.jl:20-128), seed! (src/MPOPIS.jl:54),
#     calculate_trajectory_costs (src/mppi_mpopi_policies.jl:186,303,347,434,532,644,709,782), pol(env) (:121,:221),
#     env(action), reward(env), within_track (src/envs/car_racing.jl:238,201; car_racing_tracks.jl:68)
#
#     julia --project=<MPOPIS checkout> tools/gen_golden.jl            # write the vectors
#     julia --project=<MPOPIS checkout> tools/gen_golden.jl --check    # write them, then re-load every file and assert shapes / draw counts
#
# --check makes one run self-validating: whoever has Julia for five minutes gets either usable files or an assertion that says which
# file and which field is off (wrong number of randn! blocks for the policy, E not cs x K, resampling draws missing, ...).
#
# The reference draws from a MersenneTwister that the engine does not reproduce; parity is defined on identical standard
# normals / resampling draws.  They are captured by WRAPPING the policy's RNG (nothing is re-derived): every randn! block,
# every integer-range draw and every uniform the policy consumes is logged in consumption order.
using MPOPIS, Random, LinearAlgebra, JSON
import Random: rand, randn, randn!, AbstractRNG

mutable struct RecRNG <: AbstractRNG
    inner::MersenneTwister
    normals::Vector{Vector{Float64}}      # one entry per randn! call (column-major contents) or scalar randn
    ints::Vector{Int}                     # results of rand(rng, 1:n)            (alias sampler: column index, :805)
    unis::Vector{Float64}                 # results of rand(rng) :: Float64      (alias sampler: accept test)
end
RecRNG(seed) = RecRNG(MersenneTwister(seed), Vector{Float64}[], Int[], Float64[])
Random.seed!(r::RecRNG, seed) = (Random.seed!(r.inner, seed); empty!(r.normals); empty!(r.ints); empty!(r.unis); r)
Base.copy(r::RecRNG) = RecRNG(copy(r.inner), copy(r.normals), copy(r.ints), copy(r.unis))
# raw bit generators: forwarded, not logged
Random.rng_native_52(::RecRNG) = UInt64
rand(r::RecRNG, ::Random.SamplerType{T}) where {T<:Union{Bool,Int8,UInt8,Int16,UInt16,Int32,UInt32,Int64,UInt64,Int128,UInt128}} = rand(r.inner, T)
# what Distributions calls: randn!(rng, x) for MvNormal (then unwhiten!), rand(rng, 1:K) and rand(rng) for the alias table
function randn!(r::RecRNG, A::AbstractArray{Float64})
    randn!(r.inner, A); push!(r.normals, vec(copy(A))); A
end
randn(r::RecRNG, ::Type{Float64}=Float64) = (x = randn(r.inner); push!(r.normals, [x]); x)
rand(r::RecRNG, sp::Random.SamplerTrivial{Random.CloseOpen01{Float64}}) = (u = rand(r.inner, sp); push!(r.unis, u); u)
rand(r::RecRNG, sp::Random.SamplerRangeNDL) = (i = rand(r.inner, sp); push!(r.ints, Int(i)); i)
rand(r::RecRNG, sp::Random.SamplerRangeFast) = (i = rand(r.inner, sp); push!(r.ints, Int(i)); i)

outdir = joinpath(@__DIR__, "..", "tests", "golden")
mkpath(outdir)

make_env(num_cars) = num_cars == 1 ? CarRacingEnv(rng=MersenneTwister(1)) : MultiCarRacingEnv(num_cars, rng=MersenneTwister(1))

const CTORS = Dict(:mppi => MPPI_Policy, :gmppi => GMPPI_Policy, :imppi => IMPPI_Policy, :cemppi => CEMPPI_Policy,
                   :cmamppi => CMAMPPI_Policy, :μaismppi => μAISMPPI_Policy, :μΣaismppi => μΣAISMPPI_Policy, :pmcmppi => PMCMPPI_Policy)

# the policy structs are parametric in their RNG type (`mutable struct X{R<:AbstractRNG}`, src/mppi_mpopi_policies.jl:107,284,...),
# so the recording RNG goes in through the constructors' own `rng` keyword (same keywords get_policy passes, example_utils.jl:20-128)
function make_policy(policy_type, env, K, T, N, num_cars, Σ_est, seed)
    rec = RecRNG(seed)
    kw = Dict{Symbol,Any}(:num_samples => K, :horizon => T, :λ => 10.0, :α => 1.0, :U₀ => zeros(2num_cars),
                          :cov_mat => block_diagm([0.0625, 0.1], num_cars), :log => false, :rng => rec)
    if policy_type == :cemppi
        kw[:opt_its] = N; kw[:ce_elite_threshold] = 0.8; kw[:Σ_est] = Σ_est
    elseif policy_type == :cmamppi
        kw[:opt_its] = N; kw[:σ] = 0.75; kw[:elite_perc_threshold] = 0.8
    elseif policy_type in (:μaismppi, :μΣaismppi, :pmcmppi)
        kw[:opt_its] = N; kw[:λ_ais] = 20.0
    elseif policy_type == :imppi
        kw[:opt_its] = N
    end
    return CTORS[policy_type](env; kw...), rec
end

function dump_case(name, policy_type; K=48, T=6, N=4, num_cars=1, Σ_est=:mle, seed=11, warm_steps=0)
    env = make_env(num_cars)
    for _ in 1:warm_steps                       # move off the reset state (curved track section, non-zero slip)
        env(repeat([0.05, 0.6], num_cars))
    end
    # (1) calculate_trajectory_costs: cost, E, weights
    pol, rec = make_policy(policy_type, env, K, T, N, num_cars, Σ_est, seed)
    U0 = copy(pol.U)
    res = MPOPIS.calculate_trajectory_costs(pol, env)
    cost, E = res[1], res[2]
    weights = length(res) >= 3 ? res[3] : MPOPIS.compute_weights(pol.params.weight_method, cost)
    out = Dict{String,Any}("policy" => String(policy_type), "K" => K, "T" => T, "N" => N, "num_cars" => num_cars,
                           "sigma_est" => String(Σ_est), "lambda" => 10.0, "alpha" => 1.0, "lambda_ais" => 20.0,
                           "elite_threshold" => 0.8, "cma_sigma" => 0.75, "cov" => repeat([0.0625, 0.1], num_cars),
                           "state" => collect(Float64, state(env)), "U0" => U0, "cost" => cost, "weights" => weights,
                           "normals" => rec.normals, "ints_1based" => rec.ints, "unis" => rec.unis,
                           "track_x" => env.track.x′, "track_y" => env.track.y′, "track_w" => env.track.lane_width′)
    if policy_type == :mppi
        out["E"] = [collect(Float64, e) for e in vec(E)]          # K x T array of as-vectors, column-major (k fastest)
    else
        out["E"] = vec(collect(Float64, E))                       # cs x K column-major (after the final shift)
    end
    # (2) the functor: control and the rolled pol.U, from a fresh policy with the same seed (same draws)
    pol2, rec2 = make_policy(policy_type, env, K, T, N, num_cars, Σ_est, seed)
    control = pol2(env)
    out["control"] = vec(collect(Float64, control))
    out["U_after"] = collect(Float64, pol2.U)
    @assert rec2.normals == rec.normals
    open(joinpath(outdir, "julia_$(name).json"), "w") do io
        JSON.print(io, out)
    end
    println("wrote julia_$(name).json: ", length(rec.normals), " randn! blocks, ", length(rec.ints), " resampling draws")
end

dump_case("mppi", :mppi; K=20, T=8, N=1)
dump_case("gmppi", :gmppi; N=1)
dump_case("imppi", :imppi)
dump_case("muais", :μaismppi)
dump_case("musigma", :μΣaismppi)
dump_case("musigma_warm", :μΣaismppi; warm_steps=25, K=120, T=10)
dump_case("ce_mle", :cemppi; Σ_est=:mle)
for est in (:ss, :lw, :rblw, :oas)                                # CovarianceEstimation.jl shrinkage estimators (third party)
    dump_case("ce_$(est)", :cemppi; Σ_est=est, K=60)
end
dump_case("cma", :cmamppi; K=64)
dump_case("pmc", :pmcmppi)
dump_case("gmppi_3car", :gmppi; N=1, num_cars=3)
dump_case("musigma_2car", :μΣaismppi; num_cars=2, K=64)
dump_case("cma_3car", :cmamppi; num_cars=3, K=96, T=5)

# ---- env steps / reward / within_track: single car and 3 cars -----------------------------------------------------------------
for nc in (1, 3)
    env = make_env(nc)
    steps = []
    for a in ([0.0, 0.0], [1.0, 0.5], [-0.3, -1.0], [0.2, 1.0], [-1.0, 1.0], [0.7, -0.4])
        act = repeat(a, nc)
        env(act)
        push!(steps, Dict("a" => act, "state" => collect(Float64, state(env)), "reward" => reward(env)))
    end
    open(joinpath(outdir, "julia_env_steps_$(nc)car.json"), "w") do io
        JSON.print(io, steps)
    end
end
env = make_env(1)
wt = []
for p in ([0.0, 0.0], [10.0, 5.0], [-20.0, 40.0], [3.0, 80.0], [60.0, 100.0], [-14.9, 0.0], [15.1, 0.0])
    w, d = MPOPIS.within_track(env.track, p)
    push!(wt, Dict("pos" => p, "within" => w, "dist" => d))
end
open(joinpath(outdir, "julia_within_track.json"), "w") do io
    JSON.print(io, Dict("track_x" => env.track.x′, "track_y" => env.track.y′, "track_w" => env.track.lane_width′, "queries" => wt))
end
# MountainCar (RL.jl dynamics + the reward override of src/examples/mountaincar_example.jl:4-22)
try
    mc = MPOPIS.MountainCarEnv(continuous=true, rng=MersenneTwister(1))
    mc.state[1] = -0.5; mc.state[2] = 0.0
    ms = []
    for a in (1.0, -1.0, 0.3, 1.0, 1.0)
        mc([a]); push!(ms, Dict("a" => a, "state" => collect(Float64, state(mc)), "reward" => reward(mc)))
    end
    open(joinpath(outdir, "julia_mountaincar_steps.json"), "w") do io
        JSON.print(io, ms)
    end
catch err
    @warn "MountainCar vectors skipped" err
end

# ---- --check: re-load what was just written and assert it has the layout tests/test_julia_golden.py consumes ---------------------------------
function check_outputs()
    files = filter(f -> startswith(f, "julia_") && endswith(f, ".json"), readdir(outdir))
    @assert !isempty(files) "no julia_*.json in $outdir"
    npol = 0
    for f in sort(files)
        d = JSON.parsefile(joinpath(outdir, f))
        if f == "julia_within_track.json"
            @assert length(d["track_x"]) == length(d["track_y"]) == length(d["track_w"]) "$f: track arrays differ in length"
            @assert all(q -> haskey(q, "pos") && haskey(q, "within") && haskey(q, "dist"), d["queries"]) "$f: query fields"
            continue
        elseif startswith(f, "julia_env_steps_") || f == "julia_mountaincar_steps.json"
            @assert d isa Vector && !isempty(d) "$f: expected a list of steps"
            @assert all(st -> haskey(st, "a") && haskey(st, "state") && haskey(st, "reward"), d) "$f: step fields"
            continue
        end
        npol += 1
        K, T, N, nc = d["K"], d["T"], d["N"], d["num_cars"]
        as, cs = 2nc, 2nc * T
        pol = d["policy"]
        @assert length(d["cost"]) == K "$f: cost has $(length(d["cost"])) entries, K = $K"
        @assert length(d["weights"]) == K "$f: weights"
        @assert length(d["control"]) == as "$f: control has $(length(d["control"])) entries, as = $as"
        @assert length(d["U0"]) == cs && length(d["U_after"]) == cs "$f: U0 / U_after must have cs = $cs entries"
        @assert length(d["state"]) == 8nc "$f: state"
        @assert abs(sum(d["weights"]) - 1) < 1e-9 "$f: weights do not sum to 1"
        nb = length(d["normals"])
        if pol == "mppi"
            # rand(rng, P, K, T) with P = MvNormal(as x as): K*T draws of an as-vector (src/mppi_mpopi_policies.jl:193)
            @assert sum(length, d["normals"]) == K * T * as "$f: :mppi consumed $(sum(length, d["normals"])) normals, expected K*T*as = $(K * T * as)"
            @assert length(d["E"]) == K * T "$f: E must hold K*T as-vectors"
        else
            iters = pol == "gmppi" ? 1 : N
            # one randn! block of cs*K per executed iteration; CE / CMA may stop early (:459-461, :567-569)
            @assert 1 <= nb <= iters "$f: $nb randn! blocks for N = $iters"
            @assert all(b -> length(b) == cs * K, d["normals"]) "$f: a randn! block is not cs*K = $(cs * K) long"
            @assert length(d["E"]) == cs * K "$f: E must be cs x K"
            (pol in ("gmppi", "imppi", "μaismppi", "μΣaismppi", "pmcmppi")) && @assert nb == iters "$f: $pol draws every iteration ($nb of $iters blocks)"
        end
        if pol == "pmcmppi"
            # Categorical alias sampler: one integer + one uniform per resampled column, (N-1) resampling rounds of K (:804-805)
            @assert length(d["ints_1based"]) == (N - 1) * K "$f: $(length(d["ints_1based"])) alias integers, expected (N-1)*K = $((N - 1) * K)"
            @assert length(d["unis"]) == (N - 1) * K "$f: alias uniforms"
            @assert all(i -> 1 <= i <= K, d["ints_1based"]) "$f: alias integer out of 1:K"
        else
            @assert isempty(d["ints_1based"]) && isempty(d["unis"]) "$f: $pol must not consume resampling draws"
        end
    end
    @assert npol >= 16 "only $npol policy cases written"
    println("check ok: ", length(files), " files, ", npol, " policy cases -- commit tests/golden/julia_*.json and run pytest tests/test_julia_golden.py")
end
("--check" in ARGS) && check_outputs()
