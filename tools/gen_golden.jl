# gen_golden.jl -- for whoever has a Julia runtime + MPOPIS installed: dumps golden vectors of the
# reference's hot path so the CPU oracle (oracle/mpopis_oracle.c) can be PINNED against the real
# reference.  Never required by CI (Julia is absent from the build image; parity is "unpinned"
# until these vectors exist).  Output: tests/golden/julia_*.json (inputs + expected outputs).
using MPOPIS, Random, LinearAlgebra, JSON

function dump_case(name, policy_type; K=48, T=6, N=4, num_cars=1, seed=11)
    env = num_cars == 1 ? CarRacingEnv(rng=MersenneTwister()) : MultiCarRacingEnv(num_cars, rng=MersenneTwister())
    pol = MPOPIS.get_policy(policy_type, env, K, T, 10.0, 1.0, zeros(2num_cars), block_diagm([0.0625, 0.1], num_cars), false,
                            N, 20.0, 0.8, :mle, 0.75, 0.8)
    seed!(pol, seed)
    rng0 = copy(pol.rng)
    U0 = copy(pol.U)
    cost, E, w = MPOPIS.calculate_trajectory_costs(pol, env)
    out = Dict("policy" => String(policy_type), "K" => K, "T" => T, "N" => N, "num_cars" => num_cars, "seed" => seed,
               "state" => state(env), "U0" => U0, "cost" => cost, "E" => vec(E), "weights" => w)
    # the raw standard normals the policy consumed (to feed the oracle as injected noise)
    cs = length(U0)
    out["Z"] = [vec(randn(rng0, cs, K)) for _ in 1:(policy_type == :gmppi ? 1 : N)]   # valid for the non-resampling variants
    open(joinpath(@__DIR__, "..", "tests", "golden", "julia_$(name).json"), "w") do io
        JSON.print(io, out)
    end
end

for (name, pt) in (("gmppi", :gmppi), ("muais", :μaismppi), ("musigma", :μΣaismppi), ("ce", :cemppi), ("cma", :cmamppi))
    dump_case(name, pt)
end
# single env steps / rewards
env = CarRacingEnv(rng=MersenneTwister())
steps = []
for a in ([0.0, 0.0], [1.0, 0.5], [-0.3, -1.0])
    env(a); push!(steps, Dict("a" => a, "state" => copy(env.state), "reward" => reward(env)))
end
open(joinpath(@__DIR__, "..", "tests", "golden", "julia_env_steps.json"), "w") do io
    JSON.print(io, steps)
end
