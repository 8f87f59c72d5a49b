// kernels_sample.hip -- proposal sampling on device:
//   randn!(rng, cs x K)  +  unwhiten!: E = L*Z   (Distributions.rand(rng, MvNormal(Σ), K) as called at
//   src/mppi_mpopi_policies.jl:308,359,448,556,657,724,797; rand(rng,P,K,T) at :193 for :mppi)
// The reference stream (Julia MersenneTwister + ziggurat) is not reproduced; the engine draws from
// Philox4x32-10 counter streams + Box-Muller (bit-identical to the oracle's generator up to the
// last-ulp differences of log/sin/cos) or consumes injected normals.  Counter = (index in the reference's
// linear draw order) / 4 -- four normals per call, philox.h; key = per-trial seed; stream = (mpc step, AIS iteration).
#include "engine.h"
#include "philox.h"

namespace mpopis {

__global__ void k_rng_tab_init(double* g_rng_tab) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < kRngTabLog) {
        const double c = (i == kRngTabLog - 1) ? 1.0 : 0.5 * (1.0 + (i + 0.5) / kRngTabLog);
        g_rng_tab[2 * i] = 1.0 / c; g_rng_tab[2 * i + 1] = log(c);
    } else if (i < kRngTabLog + kRngTabSc) {
        const int j = i - kRngTabLog;
        g_rng_tab[2 * i] = sinpi((2.0 * j + 1.0) / kRngTabSc); g_rng_tab[2 * i + 1] = cospi((2.0 * j + 1.0) / kRngTabSc);
    }
}
void launch_rng_tab_init(double* gtab, hipStream_t s) { hipLaunchKernelGGL(k_rng_tab_init, dim3((kRngTabLog + kRngTabSc + 255) / 256), dim3(256), 0, s, gtab); }

// Z[b][r][k] (K fastest) = standard normal number `lin` of the reference's draw order, scaled by
// dscale[r] when the Cholesky factor is diagonal (then Z is already E).
//   G-variants: lin = k*cs + r            (randn! fills the cs x K matrix column-major)
//   :mppi     : lin = (t*K + k)*as + a    with r = t*as + a  (k fastest, then t)
__global__ void __launch_bounds__(256) k_sample_normal(double* __restrict__ Z, int cs, int K, int as, int mppi_order,
                                                       const uint64_t* seeds, uint32_t slo, uint32_t shi,
                                                       const double* dscale, const int* active, const double* __restrict__ rng_tab) {
    const int b = blockIdx.z;
    if (active && !active[b]) return;
    __shared__ double sh_tab[kRngTabDoubles];
    stage_rng_tab(sh_tab, rng_tab, threadIdx.x, 256);
    __syncthreads();
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (k >= K) return;
    uint64_t lin;
    if (mppi_order) { const int t = r / as, a = r - t * as; lin = ((uint64_t)t * K + k) * as + a; }
    else lin = (uint64_t)k * cs + r;
    double z0, z1;
    philox_normal_pair(seeds[b], slo, shi, lin >> 1, sh_tab, &z0, &z1);
    double z = (lin & 1) ? z1 : z0;
    if (dscale) z *= dscale[(size_t)b * cs + r];
    Z[((size_t)b * cs + r) * K + k] = z;
}

// Pair version: one Philox call + one Box-Muller per TWO normals.  Needs the two members of a draw pair
// (linear indices 2p, 2p+1) to sit in adjacent rows of the same sample: cs even (G order) / as even (:mppi).
__global__ void __launch_bounds__(256) k_sample_normal_pair(double* __restrict__ Z, int cs, int K, int as, int mppi_order,
                                                            const uint64_t* seeds, uint32_t slo, uint32_t shi,
                                                            const double* dscale, const int* active, const double* __restrict__ rng_tab) {
    const int b = blockIdx.z;
    if (active && !active[b]) return;
    __shared__ double sh_tab[kRngTabDoubles];
    stage_rng_tab(sh_tab, rng_tab, threadIdx.x, 256);
    __syncthreads();
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int r = 2 * blockIdx.y;                               // rows r, r+1
    if (k >= K) return;
    uint64_t lin;
    if (mppi_order) { const int t = r / as, a = r - t * as; lin = ((uint64_t)t * K + k) * as + a; }
    else lin = (uint64_t)k * cs + r;
    double z0, z1;
    philox_normal_pair(seeds[b], slo, shi, lin >> 1, sh_tab, &z0, &z1);
    if (dscale) { z0 *= dscale[(size_t)b * cs + r]; z1 *= dscale[(size_t)b * cs + r + 1]; }
    Z[((size_t)b * cs + r) * K + k] = z0;
    Z[((size_t)b * cs + r + 1) * K + k] = z1;
}

// Quad version (G order, 4 | cs): rows 4 rq .. 4 rq + 3 of sample k are the four normals of one Philox call
__global__ void __launch_bounds__(256) k_sample_normal_quad(double* __restrict__ Z, int cs, int K, const uint64_t* seeds, uint32_t slo, uint32_t shi,
                                                            const double* dscale, const int* active, const double* __restrict__ rng_tab) {
    const int b = blockIdx.z;
    if (active && !active[b]) return;
    __shared__ double sh_tab[kRngTabDoubles];
    stage_rng_tab(sh_tab, rng_tab, threadIdx.x, 256);
    __syncthreads();
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int r = 4 * blockIdx.y;
    if (k >= K) return;
    double z[4];
    philox_normal_quad(seeds[b], slo, shi, ((uint64_t)k * cs + r) >> 2, sh_tab, z);
#pragma unroll
    for (int i = 0; i < 4; ++i) Z[((size_t)b * cs + r + i) * K + k] = dscale ? z[i] * dscale[(size_t)b * cs + r + i] : z[i];
}

void launch_sample_normal(double* Z, int B, int cs, int K, int as, int mppi_order, const uint64_t* seeds,
                          uint32_t slo, uint32_t shi, const double* dscale, const int* active, hipStream_t s, const double* rng_tab) {
    const bool pairable = mppi_order ? (as % 2 == 0) : (cs % 2 == 0);
    if (!mppi_order && cs % 4 == 0)
        hipLaunchKernelGGL(k_sample_normal_quad, dim3((K + 255) / 256, cs / 4, B), dim3(256), 0, s, Z, cs, K, seeds, slo, shi, dscale, active, rng_tab);
    else if (pairable)
        hipLaunchKernelGGL(k_sample_normal_pair, dim3((K + 255) / 256, cs / 2, B), dim3(256), 0, s, Z, cs, K, as, mppi_order, seeds, slo, shi, dscale, active, rng_tab);
    else
        hipLaunchKernelGGL(k_sample_normal, dim3((K + 255) / 256, cs, B), dim3(256), 0, s, Z, cs, K, as, mppi_order, seeds, slo, shi, dscale, active, rng_tab);
}

// resampling draws for :pmcmppi (rand(rng, Categorical(ws), K), :805): i uniform in [0,K), u in [0,1)
__global__ void __launch_bounds__(256) k_sample_resample_draws(int32_t* di, double* du, int K, const uint64_t* seeds,
                                                               uint32_t slo, uint32_t shi, const int* active) {
    MPOPIS_HI_PRIO();
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    uint32_t r[4];
    const uint64_t seed = seeds[b];
    philox4x32_10((uint32_t)k, 0u, slo, shi, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const uint64_t a = ((uint64_t)r[1] << 32) | r[0], bb = ((uint64_t)r[3] << 32) | r[2];
    const double two_m53 = 1.0 / 9007199254740992.0;
    int i = (int)((double)(a >> 11) * two_m53 * K);
    if (i >= K) i = K - 1;
    di[(size_t)b * K + k] = i;
    du[(size_t)b * K + k] = (double)(bb >> 11) * two_m53;
}
void launch_sample_resample_draws(int32_t* di, double* du, int B, int K, const uint64_t* seeds, uint32_t slo, uint32_t shi,
                                  const int* active, hipStream_t s) {
    hipLaunchKernelGGL(k_sample_resample_draws, dim3((K + 255) / 256, B), dim3(256), 0, s, di, du, K, seeds, slo, shi, active);
}

}  // namespace mpopis
