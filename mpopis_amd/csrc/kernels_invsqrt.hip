// kernels_invsqrt.hip -- what CMAMPPI_Policy needs from C = Σ^-0.5 (src/mppi_mpopi_policies.jl:580) WITHOUT forming C.
//
// The reference computes the dense matrix C = Σ^-0.5 (LinearAlgebra: symmetric eigen-decomposition) every AIS iteration,
// but only ever consumes
//     C * δw                      (:581, the evolution path p_σ)                        -- a vector
//     norm(C * δs[order[ii]])     (:593, δs[..] is a SCALAR through linear indexing)    -- |δ| ||C||_F, i.e. tr(Σ^-1)
// so the n^3 matrix function (round 1: 16 coupled Newton-Schulz iterations = 48 batched n x n GEMMs per update, 48 % of
// a C4 step) is replaced by two O(n^2 m) pieces, one workgroup per trial slot / block column:
//   k_trtri_fro_pair   ||L^-1||_F^2 for the Cholesky factor that the sampler already holds (σ²Σ = L L'):
//                      tr(Σ^-1) = σ² ||L^-1||_F^2.  Block column J of X = L^-1 depends on L only, so the n/16 block
//                      columns are independent workgroups; X stays in LDS, only the sum of squares leaves.
//   k_lanczos_invsqrt  y = Σ^-0.5 δw = ||δw|| V_m f(T_m) e_1: Lanczos with full re-orthogonalisation (CGS2) on Σ,
//                      f(T_m) e_1 by the 64-node quadrature of invsqrt_quad.h (lane = node, every node one SPD tridiagonal
//                      solve).  The same solves give the residual of each shifted system, hence a rigorous error bound
//                      (λ_min(Σ) >= 1/tr(Σ^-1)), which is the stopping rule (1e-13 relative).  CMA covariances are
//                      low-rank updates of a two-eigenvalue block-diagonal matrix: m = 4 ... 40.
#include "engine.h"
#include "invsqrt_quad.h"

namespace mpopis {

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LDS traffic between the lanes of ONE wave: make earlier LDS writes visible / keep later ones from moving up
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 1/x for a normal-range x: v_rcp_f64 + two Newton steps (<= 1 ulp; an IEEE division is ~25 dependent instructions on a lone wave)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

constexpr int kTB = 16;                      // block size of the triangular inverse

// L(r, c) of the n x n column-major factor, padded with the identity beyond n
__device__ __forceinline__ double ld_L(const double* __restrict__ L, int n, int r, int c) {
    return (r < n && c < n) ? L[r + (size_t)c * n] : ((r == c) ? 1.0 : 0.0);
}

}  // namespace

// part[b][J] = sum of squares of the entries of block column J of L^-1 (rows/cols < n).
// Block row I of block column J:  X(I) = -L(I,I)^-1 sum_{J <= K < I} L(I,K) X(K).  The chain over I is serial; everything else is arranged
// so that the chip stays full at large batches AND the chain stays short at one slot:
//   * the inverses of the diagonal blocks are formed once per slot by a pre-kernel (k_trtri_diag, one wave per block) and live in global memory;
//   * block column J is paired with block column nb-1-J in one workgroup (W waves each): a pair always holds nb+1 blocks of X, so every
//     workgroup has the same LDS footprint (48 KB at n = 300: three per CU, all 640 workgroups of 64 slots resident at once) and the same
//     number of row steps;
//   * the 16 x 16 block products run on the matrix cores, the L block of the NEXT product / row is requested before the current one is used
//     (L does not depend on anything computed here), so a row step is 4 MFMAs per K block + one wave-local LDS exchange + two barriers.
// n = 300: 71 us at 1-8 slots, 100 us at 64 slots (the round-2 kernel -- one 110 KB workgroup of 16 waves per block column, FP64 FMAs from
// LDS broadcasts -- took 102 us / 313 us alone and 606 us beside the sort on the second stream).
constexpr int kTriW = 2;                                          // waves per block column

// dinv[b][I][i][k] = (L(I,I)^-1)[i][k], one wave per diagonal block
__global__ void __launch_bounds__(64) k_trtri_diag(const double* __restrict__ Lall, size_t Lstride, int n, int nb, double* __restrict__ dinv, const int* active, int hiprio) {
    if (hiprio) MPOPIS_HI_PRIO();
    const int b = blockIdx.y, I = blockIdx.x;
    if (active && !active[b]) return;
    __shared__ double Ld[256], Di[256];
    const double* L = Lall + (size_t)b * Lstride;
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) { Ld[i * 16 + 4 * g + q] = ld_L(L, n, 16 * I + i, 16 * I + 4 * g + q); Di[i * 16 + 4 * g + q] = 0.0; }
    wave_lds_sync();
    if (lane < 16) {                                              // column c of L(I,I)^-1 by forward substitution (rows in order)
        const int c = lane;
#pragma unroll 1
        for (int r = 0; r < 16; ++r) {
            double sacc = (r == c) ? 1.0 : 0.0;
#pragma unroll 1
            for (int k = 0; k < r; ++k) sacc = fma(-Ld[r * 16 + k], Di[k * 16 + c], sacc);
            Di[r * 16 + c] = sacc / Ld[r * 16 + r];
        }
    }
    wave_lds_sync();
    double* o = dinv + ((size_t)b * nb + I) * 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[lane + 64 * q] = Di[lane + 64 * q];
}

// What a Lanczos run (k_lanczos_invsqrt below) needs besides A and b depends on A and L alone: fro = scale * sum_J part[J] (= tr(A^-1)), the spectrum bounds
// M = ||A||_inf >= λ_max and m = min(1 / fro, M / 2) <= λ_min, and the 64 quadrature nodes for [m, M] (invsqrt_quad.h: an AGM / Landen recurrence with a
// sine and an arcsine per level, ~12 us on one wave).  Until round 5 every Lanczos launch computed them itself at its first step -- 17 us on the chain
// rollout -> sort -> δw -> Σ^-0.5 δw -> paths -> Σ' -> Cholesky -> sampler.  Now the LAST workgroup of a slot's k_trtri_fro_pair launch to finish does it (no
// extra launch: beside a rollout that fills every SIMD's registers a new workgroup waits ~200 us for a place), i.e. on the second stream beside the rollout
// wherever the engine runs the trace there.      prep[b] = { fro, M, m, usable (1 / 0), shift[64], weight[64] }
constexpr int kLanPrep = 4 + 2 * 64;         // doubles per slot
// this workgroup's share of the column abs sums of A (columns c_lo .. c_hi-1): lane-strided partial sums, then the wave butterfly -- the order the
// Lanczos kernels themselves used.  Four columns x five row chunks of loads in flight per wave (more would cost the trace kernel its third workgroup per CU).
__device__ __forceinline__ double lanczos_colmax(const double* __restrict__ A, int n, int c_lo, int c_hi, int tid, int nthreads) {
    const int lane = tid & 63, wv = tid >> 6, nw = nthreads >> 6;
    double colmax = 0.0;
    for (int c0 = c_lo + wv; c0 < c_hi; c0 += 4 * nw) {
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        for (int r0 = 0; r0 < n; r0 += 320) {
            double x[4][5];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double* col = A + (size_t)min(c0 + nw * u, n - 1) * n;
#pragma unroll
                for (int q = 0; q < 5; ++q) x[u][q] = col[min(r0 + lane + 64 * q, n - 1)];
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                if (r0 + lane + 64 * q < n) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u] += fabs(x[u][q]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const double t = wave_sum(a[u]); if (c0 + nw * u < c_hi) colmax = fmax(colmax, t); }
    }
    return colmax;
}
// the rest, by ONE wave: fro, the bounds, this lane's node
__device__ __forceinline__ void lanczos_prep_slot(double Mhi, const double* part, int nb, double scale, double* __restrict__ o, int lane, double* agm_a, double* agm_c) {
    double fro = 0.0;
    for (int J0 = 0; J0 < nb; J0 += 8) {                        // same summation order as a plain loop, but eight loads in flight (agent scope: other workgroups wrote them)
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            t[u] = __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)&part[min(J0 + u, nb - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
        for (int u = 0; u < 8; ++u) if (J0 + u < nb) fro += t[u];
    }
    fro *= scale;
    const double mlo = fmin(1.0 / fro, 0.5 * Mhi);
    double sh = 0.0, wt = 0.0;
    const bool ok = invsqrt_quad_node(mlo, Mhi, lane, 64, &sh, &wt, agm_a, agm_c);      // uniform: depends on mlo / Mhi only (false also for NaN bounds); AGM sequence in LDS
    if (lane == 0) { o[0] = fro; o[1] = Mhi; o[2] = mlo; o[3] = ok ? 1.0 : 0.0; }
    o[4 + lane] = sh; o[4 + 64 + lane] = wt;
}

typedef double v4f64_t __attribute__((ext_vector_type(4)));

// Block products on the matrix cores.  __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc) with av = A[li][4kk + lk], bv = Bm[li][4kk + lk]
// (li = lane & 15, lk = lane >> 4) accumulates acc[r] += sum_k A[li][k] Bm[lk + 4r][k], i.e. acc[r] = (A Bm')[li][lk + 4r] -- the idiom of
// the Cholesky kernels (kernels_linalg.hip, tile_update).  P = L(I,K) X(K): A = the L block straight from global memory (4 loads per lane
// and block instead of 16), Bm' = X(K) from LDS (row-major: Bm[j][k] = X[k][j] is the read Xs[(4kk + lk) 16 + li]).
template <int W>
__global__ void __launch_bounds__(128 * W) k_trtri_fro_pair(const double* __restrict__ Lall, size_t Lstride, int n, int nb, const double* __restrict__ dinv,
                                                            double* __restrict__ part, const int* active, int hiprio,
                                                            const double* __restrict__ Aall, const double* __restrict__ scale, double* __restrict__ prep, unsigned long long* __restrict__ sync2) {
    if (hiprio) MPOPIS_HI_PRIO();
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    extern __shared__ __attribute__((aligned(16))) double sh_tri[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, grp = wv / W, wq = wv % W, gt = tid - grp * W * 64;   // group, wave within group, thread within group
    double my_colmax = 0.0;
    if (prep) {                                                               // this workgroup's columns of ||A||_inf (independent of L: the loads run under the start-up below)
        const int per = (n + (int)gridDim.x - 1) / (int)gridDim.x;
        my_colmax = lanczos_colmax(Aall + (size_t)b * n * n, n, (int)blockIdx.x * per, min(n, ((int)blockIdx.x + 1) * per), tid, 128 * W);
    }
    const int JA = blockIdx.x, JB = nb - 1 - JA;
    const bool haveB = JB > JA;
    const int J = grp == 0 ? JA : JB, nbj = nb - J;
    const bool live = grp == 0 || haveB;
    double* Xs = sh_tri + (grp == 0 ? 0 : (size_t)(nb - JA) * 256);          // [nbj][16][16] block column of X = L^-1 (row-major blocks)
    double* Ws = sh_tri + (size_t)(nb + 1) * 256 + (size_t)grp * W * 256;    // [W][256] per wave: raw partial product, then -Dinv * partial
    double* red = sh_tri + (size_t)(nb + 1) * 256 + (size_t)2 * W * 256;     // [2 W]
    const double* L = Lall + (size_t)b * Lstride;
    const double* Db = dinv + (size_t)b * nb * 256;
    const int li = lane & 15, lk = lane >> 4;
    if (live) for (int e = gt; e < 256; e += W * 64) Xs[e] = Db[(size_t)J * 256 + e];      // X(J) = L(J,J)^-1
    // operands of the first row step, requested before the first barrier: they depend on nothing computed here
    double la[4], da[4];
    auto load_L = [&](int I, int K, double* dst) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) dst[kk] = ld_L(L, n, 16 * I + li, 16 * K + 4 * kk + lk);
    };
    auto load_D = [&](int I, double* dst) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) dst[kk] = -Db[(size_t)I * 256 + li * 16 + 4 * kk + lk];     // -(L(I,I)^-1)[li][4kk + lk] (zero above the diagonal)
    };
    if (live && J + 1 < nb && wq == 0) { load_L(J + 1, J + wq, la); load_D(J + 1, da); }
    __syncthreads();
    const int nsteps = nb - JA - 1;                                           // the longer column of the pair
    for (int st = 0; st < nsteps; ++st) {
        const int I = J + 1 + st;
        const bool row = live && I < nb;
        const int nact = row ? min(W, I - J) : 0;
        if (wq < nact) {
            // ---- P = sum_{K = J+wq, J+wq+W, ... < I} L(I,K) X(K), the next block's loads in flight during the MFMAs
            v4f64_t acc = {0.0, 0.0, 0.0, 0.0};
            for (int K = J + wq; K < I; K += W) {
                double ln[4] = {0.0, 0.0, 0.0, 0.0};
                if (K + W < I) load_L(I, K + W, ln);
                const double* xk = Xs + (size_t)(K - J) * 256;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xk[(4 * kk + lk) * 16 + li], la[kk], acc, 0, 0, 0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) la[kk] = ln[kk];
            }
            double* w = Ws + wq * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r) w[li * 16 + lk + 4 * r] = acc[r];      // P[li][lk + 4r]
            wave_lds_sync();
            // ---- Q = -L(I,I)^-1 P
            v4f64_t q = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) q = __builtin_amdgcn_mfma_f64_16x16x4f64(w[(4 * kk + lk) * 16 + li], da[kk], q, 0, 0, 0);
            wave_lds_sync();                                                  // every lane has read the raw block: overwrite it in place
#pragma unroll
            for (int r = 0; r < 4; ++r) w[li * 16 + lk + 4 * r] = q[r];
        }
        // next row's first operands (global memory, independent of this row's result)
        if (live && I + 1 < nb && wq < min(W, I + 1 - J)) { load_L(I + 1, J + wq, la); load_D(I + 1, da); }
        __syncthreads();
        if (row) {
            for (int e = gt; e < 256; e += W * 64) {                          // X(I) = sum of the waves' blocks
                double sacc = 0.0;
                for (int w = 0; w < nact; ++w) sacc += Ws[w * 256 + e];
                Xs[(size_t)(I - J) * 256 + e] = sacc;
            }
        }
        __syncthreads();
    }
    double s = 0.0;
    if (live) for (int e = gt; e < nbj * 256; e += W * 64) {
        const int r = 16 * (J + (e >> 8)) + ((e & 255) >> 4), c = 16 * J + (e & 15);
        const double v = Xs[e];
        if (r < n && c < n) s = fma(v, v, s);
    }
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    if (live && gt == 0) {
        double t = 0.0;
        for (int w = 0; w < W; ++w) t += red[grp * W + w];
        // (agent scope: written through, so that another workgroup's agent-scope load sees it -- the L2s of the eight XCDs are not coherent for plain stores)
        __hip_atomic_store((unsigned long long*)&part[(size_t)b * nb + J], (unsigned long long)__double_as_longlong(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!prep) return;
    // ||A||_inf: every workgroup's share into the slot's maximum (bit patterns of non-negative doubles order like the numbers; a NaN sum wins and makes the
    // bounds unusable, as it should).  The slot's last workgroup to arrive prepares the Lanczos run and zeroes both words for the next launch.
    __shared__ int sh_last;
    __shared__ double agm_a[kQuadAgmMax + 1], agm_c[kQuadAgmMax + 1];
    unsigned long long* sy = sync2 + 2 * (size_t)b;
    {
        unsigned long long kbits = (unsigned long long)__double_as_longlong(my_colmax);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(kbits, o, 64); kbits = (t > kbits) ? t : kbits; }
        if (lane == 0) atomicMax(&sy[1], kbits);
    }
    // this workgroup's part[] entries and its maximum have arrived (vmcnt: write-through stores and atomics are acknowledged by memory) before its ticket is
    // drawn -- no device-scope fence: on this part a release at agent scope writes the whole L2 back (tens of us once the rollouts' E lines are dirty in it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (the ticket is an ACQUIRE at agent scope -- a buffer invalidate here, no L2 write-back -- so that the last workgroup's reads of part[] / sy[1] happen-after
    // every other workgroup's ticket under the HIP memory model; the release side is the store-acknowledge wait above: gfx9 family only, see kernels_select.hip)
    if (tid == 0) sh_last = (__hip_atomic_fetch_add(&sy[0], 1ull, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!sh_last || wv != 0) return;
    const double Mhi = __longlong_as_double((long long)__hip_atomic_load(&sy[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (lane == 0) { __hip_atomic_store(&sy[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&sy[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    lanczos_prep_slot(Mhi, part + (size_t)b * nb, nb, scale ? scale[b] : 1.0, prep + (size_t)b * kLanPrep, lane, agm_a, agm_c);
}

// Lanczos runs until the error bound is met, at most n steps (with full re-orthogonalisation the Krylov space is then
// exhausted and the result exact): generic dense covariances need m ~ n, CMA's low-rank-updated ones m = 4 ... 40.
//
// Everything on the serial chain of one Lanczos step is arranged around the ~1.5-2 us a dependent global-memory round trip
// costs a lone workgroup on this part:
//   * w = A v: each wave owns columns c = wv, wv + 16, ...; the loads of kLanCG columns x kLanNQ row chunks (30 per lane) are
//     issued back to back before any is consumed (4 round trips at n = 300, instead of one per column);
//   * the Lanczos basis lives in LDS (first `nvl` vectors; later ones in the global workspace), so both Gram-Schmidt passes
//     are LDS-only;
//   * the stopping rule is O(1) per step: for the shifted systems (T_m + s_j I) z = e_1 the LEADING pivots d_m and the last
//     solution component z_m = g_m follow from the previous step by a two-term recurrence (lane = quadrature node, state in
//     registers).  The full coefficient vector c = sum_j w_j (T_m + s_j I)^-1 e_1 (O(m) back substitution per node) is formed
//     once, after convergence.  ||c|| >= λ_max^-1/2 >= M^-1/2 turns the relative test into a (slightly conservative) absolute one.
constexpr int kLanThreads = 1024, kLanWaves = kLanThreads / 64;
constexpr int kLanNQ = 5;                    // rows per lane and pass of the mat-vec (64*5 = 320 >= cs = 300 in one pass)
constexpr int kLanCG = 4;                    // columns whose loads are in flight together
constexpr int kLanPivLds = 16;               // pivot rows of the final back substitution kept in LDS (longer runs: global workspace)
constexpr double kLanTol = 1e-13;
constexpr int kLanRed = kLanWaves + 4 + 16;  // block-reduction scratch (+ spare)

// y[b] = A[b]^-1/2 bvec[b];  fro[b] = scale[b] * sum_J part[b][J]  (= tr(A^-1) = ||A^-1/2||_F^2)
// status: MPOPIS_ERR_NUMERIC when the spectrum bounds are unusable (non-finite input, M/m beyond 1e14)
//
// COOP = true: G workgroups per matrix.  One CU reads A at ~57 GB/s from L2, which is what a Lanczos step costs at n = 300 (720 KB
// per mat-vec: 12.6 of 19 us).  Here workgroup g keeps the columns [g nc, (g+1) nc) of A (= rows, A symmetric) in LDS for the whole
// run and produces the entries w[c] = A[:, c] . v of its columns; the slices are exchanged through global memory as self-validating
// 8-byte granules {32 data bits | 32 tag bits} (two per double, both tagged with (launch epoch, step)), written
// with agent-scope stores straight from the reducing lane and polled by the consumers -- no flag, no drain, no fence.  Everything after the
// mat-vec (orthogonalisation, stopping rule, quadrature) is replicated: every workgroup holds the full vectors and takes identical
// decisions on identical bits.  Waits are bounded (2 s -> MPOPIS_ERR_HIP).
#ifdef LAN_PROF
__device__ unsigned long long g_lprof[64 * 8];
void debug_read_lprof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lprof), sizeof(g_lprof)); }
#define LPROF(j_, k_) do { if (blockIdx.x == 0 && tid == 0 && (j_) < 48) g_lprof[(j_) * 8 + (k_)] = wall_clock64(); } while (0)
#define LPROFG(k_) do { if (b == 0 && threadIdx.x == 0 && g < 8) g_lprof[(48 + g) * 8 + (k_)] = wall_clock64(); } while (0)      // start-up of every workgroup of slot 0
#else
#define LPROF(j_, k_) do { } while (0)
#define LPROFG(k_) do { } while (0)
#endif
// returns 1 when a cluster member gave up waiting for its partners (COOP only; the caller lets workgroup 0 of the cluster redo the slot alone), else 0
// Workspace of slot b: Gs spill regions of (n + 1 + 128) n doubles (cooperative runs: one per workgroup), and never less than 2 n^2 doubles -- what the dense
// fall-back (dense_invsqrt_slot: V and the working copy of A) needs when the quadrature cannot resolve the spectrum.
__host__ __device__ inline size_t invsqrt_slot_doubles(int n, int Gs) {
    const size_t a = (size_t)Gs * (size_t)(n + 1 + 128) * n, d = (size_t)2 * n * n;
    return a > d ? a : d;
}

// ---- dense fall-back: A^-1/2 b and tr(A^-1) by a symmetric eigen-decomposition, for the (cold) case cond(A) > 1e14 ---------------------------------------
// The reference forms Σ^-0.5 with eigen() (LinearAlgebra symmetric.jl, ^(A::Symmetric, p): src/mppi_mpopi_policies.jl:580), which has no conditioning limit;
// the Lanczos + 64-node quadrature above resolves m/M down to 1e-14.  Beyond that ONE workgroup runs a cyclic two-sided Jacobi iteration on a copy M of A
// (global workspace; the slot's Lanczos region is idle: its run never starts), accumulating V: the same method as the oracle's orc_sym_eig, in the parallel
// round-robin order (n/2 disjoint rotations per round: phase 1 the angles, phase 2 the column rotations of M and V, phase 3 the row rotations of M) instead of
// the row-cyclic one.  Stops like the oracle (off-diagonal mass <= 1e-30 of the diagonal's, at most 60 sweeps).  Then y = V diag(λ^-1/2) V' b, fro = sum 1/λ;
// a non-positive eigenvalue is the reference's DomainError / the oracle's -2 (MPOPIS_ERR_NOT_PD).  ~0.1 s per call at n = 300: a cold path, not a fast one.
__device__ __noinline__ void dense_invsqrt_slot(const double* __restrict__ A, const double* __restrict__ bv, double* M, double* V, double* __restrict__ y,
                                                double* __restrict__ fro_out, int* __restrict__ status_b, int n, double* sh /* >= 3 n + 40 doubles of LDS */) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, NT = kLanThreads;
    const int ne = (n + 1) & ~1, np = ne / 2;
    double* Cc = sh;                       // [np]
    double* Ss = Cc + np;                  // [np]
    int* Pq = reinterpret_cast<int*>(Ss + np);      // [2 np] ints
    double* red = Ss + np + np + 2;        // [2 * kLanWaves]
    double* ev = red + 2 * kLanWaves;      // [n]  (ev, then t = V' b scaled)
    for (int e = tid; e < n * n; e += NT) { M[e] = A[e]; V[e] = (e / n == e % n) ? 1.0 : 0.0; }
    __syncthreads();
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int e = tid; e < n * n; e += NT) { const double v = M[e]; if (e / n == e % n) dg = fma(v, v, dg); else off = fma(v, v, off); }
        off = wave_sum(off); dg = wave_sum(dg);
        __syncthreads();
        if (lane == 0) { red[wv] = off; red[kLanWaves + wv] = dg; }
        __syncthreads();
        off = 0.0; dg = 0.0;
        for (int w = 0; w < kLanWaves; ++w) { off += red[w]; dg += red[kLanWaves + w]; }
        if (off <= 1e-30 * (dg > 0.0 ? dg : 1.0) || !(off == off)) break;            // uniform: every thread sums the same partials in the same order
        for (int r = 0; r < ne - 1; ++r) {
            // ---- phase 1: the np disjoint pairs of round r (round-robin tournament) and their rotation angles
            for (int k = tid; k < np; k += NT) {
                int a = (k == 0) ? ne - 1 : (r + k) % (ne - 1);
                int c = (k == 0) ? r : (r + ne - 1 - k) % (ne - 1);
                int p = a < c ? a : c, q = a < c ? c : a;
                double cc = 1.0, ss = 0.0;
                if (q < n) {
                    const double apq = M[p + (size_t)q * n];
                    if (apq != 0.0) {
                        const double app = M[p + (size_t)p * n], aqq = M[q + (size_t)q * n];
                        const double theta = (aqq - app) / (2.0 * apq);
                        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        cc = 1.0 / sqrt(t * t + 1.0); ss = t * cc;
                    }
                } else { p = -1; }
                Cc[k] = cc; Ss[k] = ss; Pq[2 * k] = p; Pq[2 * k + 1] = q;
            }
            __syncthreads();
            // ---- phase 2: columns p, q of M and of V
            for (int e = tid; e < np * n; e += NT) {
                const int k = e / n, i = e - k * n, p = Pq[2 * k], q = Pq[2 * k + 1];
                if (p < 0) continue;
                const double cc = Cc[k], ss = Ss[k];
                if (ss == 0.0) continue;
                const double kp = M[i + (size_t)p * n], kq = M[i + (size_t)q * n], vp = V[i + (size_t)p * n], vq = V[i + (size_t)q * n];
                M[i + (size_t)p * n] = cc * kp - ss * kq; M[i + (size_t)q * n] = ss * kp + cc * kq;
                V[i + (size_t)p * n] = cc * vp - ss * vq; V[i + (size_t)q * n] = ss * vp + cc * vq;
            }
            __syncthreads();
            // ---- phase 3: rows p, q of M (lanes along the pairs: neighbouring lanes touch different rows of the same column)
            for (int e = tid; e < np * n; e += NT) {
                const int j = e / np, k = e - j * np, p = Pq[2 * k], q = Pq[2 * k + 1];
                if (p < 0) continue;
                const double cc = Cc[k], ss = Ss[k];
                if (ss == 0.0) continue;
                const double pk = M[p + (size_t)j * n], qk = M[q + (size_t)j * n];
                M[p + (size_t)j * n] = cc * pk - ss * qk; M[q + (size_t)j * n] = ss * pk + cc * qk;
            }
            __syncthreads();
        }
    }
    // ---- eigenvalues, y = V diag(λ^-1/2) V' b, fro = sum 1/λ
    int bad = 0;
    double fr = 0.0;
    for (int j = tid; j < n; j += NT) { const double l = M[j + (size_t)j * n]; ev[j] = l; if (!(l > 0.0)) bad = 1; else fr += 1.0 / l; }
    bad = __syncthreads_or(bad);
    fr = wave_sum(fr);
    if (lane == 0) red[wv] = fr;
    __syncthreads();
    if (bad) {
        for (int i = tid; i < n; i += NT) y[i] = 0.0;
        if (tid == 0) status_raise(status_b, MPOPIS_ERR_NOT_PD);
        return;
    }
    if (tid == 0) { double t = 0.0; for (int w = 0; w < kLanWaves; ++w) t += red[w]; *fro_out = t; }
    __syncthreads();
    for (int j = wv; j < n; j += kLanWaves) {                    // t_j = (v_j . b) λ_j^-1/2: one wave per eigenvector
        double t = 0.0;
        for (int i = lane; i < n; i += 64) t = fma(V[i + (size_t)j * n], bv[i], t);
        t = wave_sum(t);
        if (lane == 0) ev[j] = t / sqrt(ev[j]);
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s = fma(V[i + (size_t)j * n], ev[j], s);
        y[i] = s;
    }
}

template <bool COOP>
__device__ __forceinline__ int lanczos_run(const int b, const int g, const double* __restrict__ Aall, const double* __restrict__ bvec, size_t bstride,
                                           const double* __restrict__ prep,
                                           double* __restrict__ Vall, double* __restrict__ yall, double* __restrict__ fro_out,
                                           int* __restrict__ msteps, int n, int nvl, int* status,
                                           int G, int Gs, unsigned long long* xbuf, unsigned long long epoch, int* timeouts,
                                           unsigned long long wait_ticks) {
    extern __shared__ __attribute__((aligned(16))) double sh_lan[];
    const int nc = COOP ? (n + G - 1) / G : 0, c_lo = g * nc, c_hi = min(n, c_lo + nc);       // own columns (COOP)
    double* part_v = sh_lan;                            // [kLanWaves][n]  (COOP: the column slab [nc][n])
    double* vcur = part_v + (COOP ? (size_t)nc * n : (size_t)kLanWaves * n);      // [n]
    double* wv_ = vcur + n;                             // [n]  the working vector w
    double* coef = wv_ + n;                             // [n + 1]
    double* alpha = coef + n + 1;                       // [n]
    double* beta = alpha + n;                           // [n]
    double* cvec = beta + n;                            // [n]
    double* red = cvec + n;                             // [kLanRed]
    double* pivl = red + kLanRed;                 // [2][kLanPivLds][64]  d_i and e_i = β_i/d_i of the back substitution
    double* Vl = pivl + 2 * kLanPivLds * 64;            // [nvl][n]  Lanczos basis, LDS-resident part
    __shared__ double red2[2][kLanWaves];
    const double* A = Aall + (size_t)b * n * n;
    const double* bv = bvec + (size_t)b * bstride;
    double* Vg = Vall + (size_t)b * invsqrt_slot_doubles(n, Gs) + (size_t)g * (size_t)(n + 1 + 128) * n;  // basis vectors v_0 .. v_n (only k >= nvl are ever touched); Gs regions per slot
    double* pivg = Vg + (size_t)(n + 1) * n;                    // [2][n][64]
    double* y = yall + (size_t)b * n;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool writer = !COOP || g == 0;                        // replicated results: one workgroup writes them
    unsigned long long* xb = COOP ? xbuf + (size_t)b * 4 * (n + G) : nullptr;     // [2 parities][n + G][2 granules]
    if (COOP) {                                                 // slab <- own columns (contiguous in memory), 8 loads in flight per thread
        const double* src = A + (size_t)c_lo * n;
        const int cnt = (c_hi - c_lo) * n;
        for (int e0 = tid; e0 < cnt; e0 += kLanThreads * 8) {
            double av[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) av[u] = src[min(e0 + u * kLanThreads, cnt - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (e0 + u * kLanThreads < cnt) part_v[e0 + u * kLanThreads] = av[u];
        }
    }
    LPROFG(1);
    auto block_sum = [&](double v) -> double {
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wv] = v;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kLanWaves; ++w) t += red[w];
        return t;
    };
    // spectrum bounds and this lane's quadrature node: k_lanczos_prep, queued behind the trace of L^-1 (off the per-iteration chain)
    const double* pp = prep + (size_t)b * kLanPrep;
    const double fro = pp[0], Mhi = pp[1], mlo = pp[2];
    const bool quad_ok = pp[3] != 0.0;
    const double q_shift = pp[4 + lane], q_weight = pp[4 + 64 + lane];
    const double q_wres = q_weight / (mlo + q_shift);           // residual -> error weight of this lane's node
    if (tid == 0 && writer) fro_out[b] = fro;
    // ---- v_0 = b / ||b|| --------------------------------------------------------------------------------------------------
    double s2 = 0.0;
    for (int i = tid; i < n; i += kLanThreads) { const double v = bv[i]; s2 = fma(v, v, s2); }
    const double nrm_b = sqrt(block_sum(s2));
    if (!(nrm_b > 0.0) || !(fro > 0.0)) {                             // δw = 0 -> y = 0; NaN input / unusable trace -> numeric error
        if (writer) {
            for (int i = tid; i < n; i += kLanThreads) y[i] = 0.0;
            if (tid == 0) { msteps[b] = 0; if (nrm_b > 0.0 || !(nrm_b == nrm_b)) status_raise(&status[b], MPOPIS_ERR_NUMERIC); }      // never hides an earlier error of the slot
        }
        return 0;
    }
    if (!quad_ok) {                                             // spectrum bounds beyond what the node table resolves (M/m > 1e14) or unusable
        if (writer) {
            if (Mhi == Mhi && mlo == mlo && Mhi < INFINITY && mlo > 0.0) {      // finite bounds: the dense fall-back (the reference's eigen-based Σ^-0.5 has no such limit)
                if (tid == 0) msteps[b] = -1;
                double* Vd = Vall + (size_t)b * invsqrt_slot_doubles(n, Gs);
                __syncthreads();
                dense_invsqrt_slot(A, bv, Vd + (size_t)n * n, Vd, y, &fro_out[b], &status[b], n, sh_lan);
            } else {
                for (int i = tid; i < n; i += kLanThreads) y[i] = 0.0;
                if (tid == 0) { msteps[b] = 0; status_raise(&status[b], MPOPIS_ERR_NUMERIC); }
            }
        }
        return 0;
    }
    for (int i = tid; i < n; i += kLanThreads) { const double v = bv[i] / nrm_b; vcur[i] = v; if (nvl > 0) Vl[i] = v; else Vg[i] = v; }
    __syncthreads();
    const int mcap = n;
    int m = 0;
    double rd_prev = 1.0, g_prev = 0.0;                         // pivot recurrence of this lane's node
    for (int j = 0; j < mcap; ++j) {
        // ---- w = A v_j : lanes along rows (coalesced column reads), waves split the columns, kLanCG columns of loads in flight ----
        LPROF(j, 0);
        if (j == 0) LPROFG(2);
        if constexpr (COOP) {
            // own columns: w[c] = A[:, c] . v from the LDS slab, one wave per column (up to 3 columns per wave interleaved); the
            // reducing lane stores the entry to LDS and publishes it as a granule pair
            // BOTH granules of a pair carry the same 32-bit tag (22 epoch bits | step + 1): each 8-byte store is atomic, so a granule is
            // either this step's or an older one's, and an older one never has this tag (the host clears the buffer when the epoch bits wrap)
            const unsigned long long tlo = ((unsigned long long)(((unsigned)(epoch & 0x3fffffull) << 10) | (unsigned)(j + 1))) << 32, thi = tlo;
            unsigned long long* xp = xb + (size_t)(j & 1) * 2 * (n + G);
            auto publish = [&](int idx, double v) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
                __hip_atomic_store(&xp[2 * idx], (bits & 0xffffffffull) | tlo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&xp[2 * idx + 1], (bits >> 32) | thi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            for (int cb = c_lo + wv; cb < c_hi; cb += 3 * kLanWaves) {
                double s0 = 0.0, s1 = 0.0, s2c = 0.0;
                const int c1 = cb + kLanWaves, c2 = cb + 2 * kLanWaves;
                for (int i = lane; i < n; i += 64) {
                    const double vi = vcur[i];
                    const double x0 = part_v[(size_t)(cb - c_lo) * n + i];
                    const double x1 = (c1 < c_hi) ? part_v[(size_t)(c1 - c_lo) * n + i] : 0.0;
                    const double x2 = (c2 < c_hi) ? part_v[(size_t)(c2 - c_lo) * n + i] : 0.0;
                    s0 = fma(x0, vi, s0); s1 = fma(x1, vi, s1); s2c = fma(x2, vi, s2c);
                }
                s0 = wave_sum(s0); s1 = wave_sum(s1); s2c = wave_sum(s2c);
                if (lane == 0) {
                    wv_[cb] = s0; publish(cb, s0);
                    if (c1 < c_hi) { wv_[c1] = s1; publish(c1, s1); }
                    if (c2 < c_hi) { wv_[c2] = s2c; publish(c2, s2c); }
                }
            }
            // the other workgroups' entries: one thread per entry polls its granule pair
            LPROF(j, 1);
            if (j == 0) LPROFG(3);
            int timed_out = 0;
            for (int idx = tid; idx < n; idx += kLanThreads) {
                if (idx >= c_lo && idx < c_hi) continue;
                unsigned long long lo, hi;
                const unsigned long long t0 = wall_clock64();
                unsigned spins = 0;
                for (;;) {
                    lo = __hip_atomic_load(&xp[2 * idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hi = __hip_atomic_load(&xp[2 * idx + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((lo & 0xffffffff00000000ull) == tlo && (hi & 0xffffffff00000000ull) == thi) break;
                    if ((++spins & 255u) == 0 && wall_clock64() - t0 > wait_ticks) { timed_out = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                const double v = __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32)));
                wv_[idx] = v;
            }
            if (__syncthreads_or(timed_out)) {                  // a partner is not running: workgroup 0 of this cluster redoes the slot alone (k_lanczos_invsqrt)
                if (tid == 0) atomicAdd(timeouts, 1);
                return 1;
            }
            if (j == 0) LPROFG(4);
        } else {
        for (int r0 = 0; r0 < n; r0 += 64 * kLanNQ) {
                double acc[kLanNQ];
#pragma unroll
                for (int q = 0; q < kLanNQ; ++q) acc[q] = 0.0;
                for (int c0 = wv; c0 < n; c0 += kLanWaves * kLanCG) {
                    double a[kLanCG][kLanNQ];
#pragma unroll
                    for (int u = 0; u < kLanCG; ++u) {
                        // wave-uniform column base + one lane offset + immediate 512 q: rows past n read into the next column / the
                        // padding behind the last slot (kInvsqrtPadDoubles, engine.h) and are never used
                        const double* col = A + (size_t)min(c0 + kLanWaves * u, n - 1) * n + r0;
#pragma unroll
                        for (int q = 0; q < kLanNQ; ++q) a[u][q] = col[lane + 64 * q];
                    }
#pragma unroll
                    for (int u = 0; u < kLanCG; ++u) {
                        const int c = c0 + kLanWaves * u;
                        const double vc = (c < n) ? vcur[c] : 0.0;
#pragma unroll
                        for (int q = 0; q < kLanNQ; ++q) acc[q] = fma(a[u][q], vc, acc[q]);
                    }
                }
#pragma unroll
                for (int q = 0; q < kLanNQ; ++q) if (r0 + lane + 64 * q < n) part_v[(size_t)wv * n + r0 + lane + 64 * q] = acc[q];
            }
            __syncthreads();
            for (int i = tid; i < n; i += kLanThreads) {
                double s = 0.0;
#pragma unroll
                for (int w = 0; w < kLanWaves; ++w) s += part_v[(size_t)w * n + i];
                wv_[i] = s;
            }
        }
        if (!COOP) __syncthreads();
        LPROF(j, 2);
        // ---- orthogonalise against v_0..v_j twice (classical Gram-Schmidt, CGS2); α_j = the v_j coefficient -------------------
        double a_j = 0.0, w2 = 0.0;
        const int jl = min(j + 1, nvl);                                             // vectors 0 .. jl-1 in LDS, jl .. j in global memory
        for (int pass = 0; pass < 2; ++pass) {
            for (int k = wv; k <= j; k += kLanWaves) {
                double s = 0.0;
                if (k < jl) { for (int i = lane; i < n; i += 64) s = fma(Vl[(size_t)k * n + i], wv_[i], s); }
                else { for (int i = lane; i < n; i += 64) s = fma(Vg[(size_t)k * n + i], wv_[i], s); }
                s = wave_sum(s);
                if (lane == 0) coef[k] = s;
            }
            __syncthreads();
            a_j += coef[j];
            for (int i = tid; i < n; i += kLanThreads) {
                double s = wv_[i];
                for (int k = 0; k < jl; ++k) s = fma(-coef[k], Vl[(size_t)k * n + i], s);
                for (int k = jl; k <= j; ++k) s = fma(-coef[k], Vg[(size_t)k * n + i], s);
                wv_[i] = s;
                if (pass == 1) w2 = fma(s, s, w2);                                  // ||w||^2 of the final vector: the entries this thread has just written
            }
            if (pass == 0) __syncthreads();
        }
        LPROF(j, 3);
        // β_j = ||w||: the barrier that publishes the second pass's w is the reduction's own (partials in the buffer of this step's parity: the previous
        // reduction's readers are two barriers behind)
        double bt;
        {
            const double v = wave_sum(w2);
            if (lane == 0) red2[j & 1][wv] = v;
            __syncthreads();
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < kLanWaves; ++w) t += red2[j & 1][w];
            bt = sqrt(t);
        }
        if (tid == 0) { alpha[j] = a_j; beta[j] = bt; }
        m = j + 1;
        // ---- stopping rule, O(1): leading pivot d_m = α_m + s - β_{m-1}²/d_{m-1}, last solution component z_m = g_m = -β_{m-1} g_{m-1}/d_m
        //      (g_1 = 1/d_1); residual of node j's shifted system = β_m |z_m|, error <= sum_j w_j β_m |z_m| / (λ_min + s_j)  (per unit ||b||)
        bool conv;
        {
            const double bprev = (j > 0) ? beta[j - 1] : 0.0;                        // written one step (and several barriers) ago
            const double d = a_j + q_shift - ((j > 0) ? bprev * bprev * rd_prev : 0.0);
            const double rd = fast_rcp(d);
            const double gcur = (j > 0) ? -bprev * g_prev * rd : rd;
            rd_prev = rd; g_prev = gcur;
            const double eb = bt * wave_sum(q_wres * fabs(gcur));
            conv = (eb <= kLanTol * (1.0 / sqrt(Mhi))) || (bt <= 1e-14 * Mhi) || (m >= mcap);      // identical in every wave (same node table, same recurrences): no flag, no barrier
        }
        // LOCKSTEP ASSUMPTION (documented because a divergence would deadlock the next barrier): a_j and bt are sums of the SAME LDS partials (red2) in the SAME
        // order in every thread, q_* are per-lane copies of one table, so `conv` is bit-identical in every wave.  alpha[j] / beta[j] were written by tid 0 --
        // a lane of wave 0 -- and are read after the loop by wave 0 only: same-wave program order, no barrier needed on the break path.
        LPROF(j, 4);
        if (conv) break;
        const double rbt = fast_rcp(bt);
        for (int i = tid; i < n; i += kLanThreads) {
            const double v = wv_[i] * rbt;
            vcur[i] = v;
            if (j + 1 < nvl) Vl[(size_t)(j + 1) * n + i] = v; else Vg[(size_t)(j + 1) * n + i] = v;
        }
        __syncthreads();
    }
    LPROF(m, 0);
    // ---- c = sum_nodes w (T_m + s I)^-1 e_1: LDL' per node (lane), forward g_i, back z_i = g_i - e_i z_{i+1}; wave 0 -----------------
    if (wv == 0) {
        auto dpiv = [&](int i) -> double& { return i < kLanPivLds ? pivl[i * 64 + lane] : pivg[(size_t)i * 64 + lane]; };
        auto epiv = [&](int i) -> double& { return i < kLanPivLds ? pivl[(kLanPivLds + i) * 64 + lane] : pivg[((size_t)n + i) * 64 + lane]; };
        double rd = 1.0, g = 0.0;                                                    // rd = 1 / d_{i-1}
        for (int i = 0; i < m; ++i) {
            const double bp = (i > 0) ? beta[i - 1] : 0.0;
            const double dn = alpha[i] + q_shift - ((i > 0) ? bp * bp * rd : 0.0);
            rd = fast_rcp(dn);
            g = (i > 0) ? -bp * g * rd : rd;
            dpiv(i) = g;                                                            // g_i
            epiv(i) = (i + 1 < m) ? beta[i] * rd : 0.0;                              // e_i = β_i / d_i
        }
        double z = 0.0;
        for (int i = m - 1; i >= 0; --i) {
            z = dpiv(i) - epiv(i) * z;
            const double ci = wave_sum(q_weight * z);
            if (lane == 0) cvec[i] = ci;
        }
    }
    __syncthreads();
    const int ml = min(m, nvl);
    for (int i = tid; i < n; i += kLanThreads) {
        double s = 0.0;
        for (int k = 0; k < ml; ++k) s = fma(cvec[k], Vl[(size_t)k * n + i], s);
        for (int k = ml; k < m; ++k) s = fma(cvec[k], Vg[(size_t)k * n + i], s);
        if (writer) y[i] = nrm_b * s;
    }
    if (tid == 0 && writer) msteps[b] = m;
    LPROF(m, 1);
    return 0;
}

// COOP = true: G workgroups per slot.  A cluster whose members do not all run (bounded waits) is redone by its workgroup 0 with the one-workgroup
// algorithm IN THIS LAUNCH (same LDS allocation, laid out for one workgroup: nvl_solo basis vectors) -- until round 5 a second launch of the
// one-workgroup kernel, predicated on a per-slot flag, was queued behind every cluster launch for that: 6 us on the per-iteration chain for a kernel
// that did nothing.
template <bool COOP>
__global__ void __launch_bounds__(kLanThreads) k_lanczos_invsqrt(const double* __restrict__ Aall, const double* __restrict__ bvec, size_t bstride,
                                                                 const double* __restrict__ prep,
                                                                 double* __restrict__ Vall, double* __restrict__ yall, double* __restrict__ fro_out,
                                                                 int* __restrict__ msteps, int n, int nvl, int nvl_solo, int* status, const int* active,
                                                                 int G, int Gs, unsigned long long* xbuf, unsigned long long epoch, int* timeouts,
                                                                 unsigned long long wait_ticks, int test_drop) {
    MPOPIS_HI_PRIO();
    const int b = COOP ? blockIdx.x / G : blockIdx.x, g = COOP ? blockIdx.x % G : 0;
    if (active && !active[b]) return;
    LPROFG(0);
    if (COOP && test_drop && g == G - 1) return;                // test hook: a partner that never runs
    const int rc = lanczos_run<COOP>(b, g, Aall, bvec, bstride, prep, Vall, yall, fro_out, msteps, n, nvl, status, G, Gs, xbuf, epoch, timeouts, wait_ticks);
    if constexpr (COOP) {
        if (rc && g == 0) {
            __syncthreads();
            (void)lanczos_run<false>(b, 0, Aall, bvec, bstride, prep, Vall, yall, fro_out, msteps, n, nvl_solo, status, 1, Gs, nullptr, 0ull, nullptr, 0ull);
        }
    }
}

size_t invsqrt_workspace_doubles(int B, int n, int regions_per_slot) { return (size_t)B * invsqrt_slot_doubles(n, regions_per_slot); }   // cooperative runs: one spill region per workgroup
int invsqrt_max_n() {
    // dynamic LDS of k_lanczos_invsqrt: (kLanWaves + 6) n + ... doubles, and of k_trtri_fro_pair: 16 (n + 31) + 1028 doubles, both <= 150 KiB
    return std::min((int)((150 * 1024 / 8 - 64 - kLanRed - 2 * kLanPivLds * 64) / (kLanWaves + 8)), (int)((150 * 1024 / 8 - 1028) / 16 - 31));
}

// y = A^-1/2 b and fro = tr(A^-1) per slot, from A (n x n, SPD) and the Cholesky factor L of scale*A (scale: per-slot, nullable).
// A must stay readable for kInvsqrtPadDoubles doubles behind its last slot (the mat-vec reads whole 64-row chunks).
// part[b][nb] = per-block-column partial sums of ||L^-1||_F^2 (only L is read: may run on another stream beside the sort / elite mean)
size_t trtri_dinv_doubles(int B, int n) { return (size_t)B * ((n + kTB - 1) / kTB) * 256; }
// hiprio = false: the launch runs beside a throughput kernel of the same handle (the rollout) and must not take its issue slots
size_t lanczos_prep_doubles(int B) { return (size_t)B * kLanPrep; }
// A / scale / prep / sync2 (nullable together; sync2: [B][2] words, zero before the first launch): the slot's last workgroup also leaves the
// Lanczos run's spectrum bounds and quadrature nodes in prep (lanczos_prep_slot)
void launch_trtri_fro(const double* L, size_t Lstride, double* part, int B, int n, const int* active, hipStream_t s, double* dinv, bool hiprio,
                      const double* A, const double* scale, double* prep, unsigned long long* sync2) {
    const int nb = (n + kTB - 1) / kTB;
    const size_t lds = ((size_t)(nb + 1) * 256 + 2 * kTriW * 256 + 2 * kTriW) * sizeof(double);
    static std::atomic<unsigned long long> seenp{0};
    ensure_dyn_lds((const void*)k_trtri_fro_pair<kTriW>, 150 * 1024, seenp);
    hipLaunchKernelGGL(k_trtri_diag, dim3(nb, B), dim3(64), 0, s, L, Lstride, n, nb, dinv, active, hiprio ? 1 : 0);
    hipLaunchKernelGGL(k_trtri_fro_pair<kTriW>, dim3((nb + 1) / 2, B), dim3(128 * kTriW), lds, s, L, Lstride, n, nb, dinv, part, active, hiprio ? 1 : 0,
                       A, scale, (prep && sync2) ? prep : nullptr, sync2);
}

// y = A^-1/2 b and fro = scale * sum(part) (prep: what launch_trtri_fro left for this A; 1/fro is also the quadrature's lower spectrum bound).
// xbuf / epoch (nullable): exchange buffer (invsqrt_coop_words, zero-initialised once) and launch counter of the cooperative variant.
size_t invsqrt_coop_words(int B, int n) { return (size_t)B * 4 * (n + 16); }
int invsqrt_coop_groups(int B, int n, int share) {
    static const int env_G = [] { const char* e = getenv("MPOPIS_LANCZOS_G"); return e ? atoi(e) : -1; }();
    int G = env_G >= 0 ? env_G : 8;
    if (G > 16) G = 16;
    if (G < 2 || n < 160 || n > 1000 || B * G * share > coop_max_workgroups()) return 1;                 // small matrices: one CU streams them from L2 fast enough; clusters must be co-resident
    const size_t fixed = (size_t)6 * n + 1 + kLanRed + 2 * kLanPivLds * 64 + (size_t)((n + G - 1) / G) * n;
    return (fixed + (size_t)4 * n) * sizeof(double) <= 150 * 1024 ? G : 1;
}
void launch_lanczos_invsqrt(const double* A, const double* prep, const double* bvec, size_t bstride,
                            double* V, double* y, double* fro, int* msteps, int B, int n, int* status, const int* active, hipStream_t s,
                            int regions_per_slot, const CoopCtx& coop) {
    static std::atomic<unsigned long long> seen2{0}, seen3{0};
    unsigned long long* const xbuf = coop.flags;
    const int G = coop.usable() ? std::min(invsqrt_coop_groups(B, n, coop.share), regions_per_slot) : 1;
    const size_t fixed1 = (size_t)(kLanWaves + 6) * n + 1 + kLanRed + 2 * kLanPivLds * 64;              // doubles (one-workgroup layout)
    if (G > 1) {
        const size_t fixed = (size_t)6 * n + 1 + kLanRed + 2 * kLanPivLds * 64 + (size_t)((n + G - 1) / G) * n;
        const int nvl = (int)std::min<size_t>(n + 1, (150 * 1024 / sizeof(double) - fixed) / n);
        const size_t lds = (fixed + (size_t)nvl * n) * sizeof(double);
        // the cluster's own fall-back (its workgroup 0 alone, one-workgroup layout in the same allocation)
        const int nvl_solo = lds / sizeof(double) > fixed1 ? (int)std::min<size_t>(n + 1, (lds / sizeof(double) - fixed1) / n) : 0;
        ensure_dyn_lds((const void*)k_lanczos_invsqrt<true>, 150 * 1024, seen3);
        const unsigned long long ep = ++*coop.epoch;
        if ((ep & 0x3fffffull) == 0) (void)hipMemsetAsync(xbuf, 0, invsqrt_coop_words(B, n) * sizeof(unsigned long long), s);   // tag wrap: no stale granule may alias
        hipLaunchKernelGGL(k_lanczos_invsqrt<true>, dim3(B * G), dim3(kLanThreads), lds, s, A, bvec, bstride, prep, V, y, fro, msteps, n, nvl, nvl_solo,
                           status, active, G, regions_per_slot, xbuf, ep, coop.timeouts, coop_wait_ticks(), coop_test_drop());
        return;
    }
    const int nvl1 = (int)std::min<size_t>(n + 1, (150 * 1024 / sizeof(double) - fixed1) / n);          // basis vectors that fit next to it
    const size_t lds1 = (fixed1 + (size_t)nvl1 * n) * sizeof(double);
    ensure_dyn_lds((const void*)k_lanczos_invsqrt<false>, 150 * 1024, seen2);
    hipLaunchKernelGGL(k_lanczos_invsqrt<false>, dim3(B), dim3(kLanThreads), lds1, s, A, bvec, bstride, prep, V, y, fro, msteps, n, nvl1, nvl1,
                       status, active, 1, regions_per_slot, (unsigned long long*)nullptr, 0ull, (int*)nullptr, 0ull, 0);
}

}  // namespace mpopis
