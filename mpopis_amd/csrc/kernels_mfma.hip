// kernels_mfma.hip -- the two dense FP64 contractions of the AIS loop on the CDNA4 matrix cores
// (v_mfma_f64_16x16x4_f64):
//   E = L * Z              unwhiten! of rand(rng, MvNormal(Σ′), K)        src/mppi_mpopi_policies.jl:448,556,724,797
//   Σ′ = X W X' / den + εI  StatsBase.mean_and_cov (weighted / resampled) :730-733,:806-808; cov(elite') :464
// These are the only real contractions on the path (2·cs²·K flop each: 82 MFLOP at cs=100,K=4096, 737 MFLOP at
// cs=300).  On MI355X the FP64 MFMA rate equals the FP64 vector rate (78.6 TF), so the win is operand traffic and
// issue slots: one A and one B register pair per lane feed 2048 flops.
// Fragment layout (f64 16x16x4): A lane l = A[i=l&15][k=l>>4]; B lane l = B[k=l>>4][j=l&15];
// D lane l, reg r = D[row=(l>>4)+4r][col=l&15].
#include "engine.h"
#include <type_traits>
#include "philox.h"

namespace mpopis {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int kTrmmTiles = 8;       // row tiles (of 16) per workgroup pass => 32 accumulator VGPR pairs

// E[b][i][k] = sum_{j<=i} L[b][i][j] Z[b][j][k];  L n x n column-major lower (upper part stored as zeros);
// Z, E [n][K].  grid (ceil(K/64), row groups, B), 4 waves; wave = 16 samples x (<= 8 row tiles).
// L is streamed in 16-column chunks through a double-buffered LDS panel shared by the 4 waves (one
// barrier per chunk); the next chunk's global loads (L panel + Z operands) are in flight during the
// current chunk's MFMAs.
constexpr int kTrmmRows = kTrmmTiles * 16;          // 128 rows per pass
constexpr int kTrmmLd = kTrmmRows + 16;             // LDS row stride = 16 (mod 32) doubles: conflict-free operand reads
// L is lower triangular (the sampler's E = L*Z): row tiles above a chunk's block row are skipped.
// RNG = true: Z is never materialised -- every wave draws the 16 x 16 block of standard normals it needs for the
//              current chunk from the Philox streams (same counters as k_sample_normal_pair, i.e. the same numbers), two
//              Box-Muller pairs per lane (one Philox call), in the MFMA B-operand pattern.  Needs 4 | n and all rows in one pass (n <= 128).
// oscale2 (nullable, [B]): E = sqrt(oscale2[b]) L Z -- :cmamppi draws from MvNormal(σ²Σ′) (:550-554) and keeps the factor of Σ′ itself: chol(σ²Σ′) = σ chol(Σ′),
// so the step size only scales the output and the factorisation does not have to wait for it
struct RngArgs { const uint64_t* seeds; uint32_t slo, shi; const double* tab; const double* panel; size_t pstride; const double* oscale2; };
// 4 waves per SIMD (128 VGPRs; the LDS panel allows 4 workgroups per CU): measured 5 % faster than the default 3 for the fused
// sampler (Philox / Box-Muller VALU work of one wave fills the slots in which another waits on the matrix cores)
// LD: LDS row stride of the panel in doubles, = 16 (mod 32) for conflict-free operand reads.  kTrmmLd = 144 holds all 8 row tiles; 112 (7 row
// tiles, no padding needed: 112 = 16 mod 32) is what every 1-car H = 50 shape needs and leaves room for the Box-Muller tables (6 KB) at four
// workgroups per CU: 2 x 16 x 112 x 8 + 6 KB = 34.7 KB (with the 144-stride panel the round-3 tables of 3 KB were all that fitted: 159.7 of 160 KB)
template <bool RNG, int LD = kTrmmLd>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_trmm_LZ_mfma(const double* __restrict__ L, size_t Lstride, const double* __restrict__ Z,
                                                      double* __restrict__ E, int n, int K, const int* active, RngArgs rng, int tpg) {
    __shared__ double Ls[2][16][LD];
    constexpr int TT = LD >= kTrmmRows ? kTrmmTiles : LD / 16;       // row tiles this instantiation can hold
    __shared__ double sh_tab[RNG ? kRngTabDoubles : 1];
    const int b = blockIdx.z;
    if (active && !active[b]) return;
    if (RNG) { stage_rng_tab(sh_tab, rng.tab, threadIdx.x, 256); __syncthreads(); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k0 = (blockIdx.x * 4 + wv) * 16;
    const int t0 = blockIdx.y * tpg;                           // first row tile of this group (tpg <= kTrmmTiles row tiles per group, balanced)
    const int nt_total = (n + 15) / 16;
    const int nt = min(tpg, nt_total - t0);
    const double* Lb = L + (size_t)b * Lstride;
    const double* Zb = Z + (size_t)b * n * K;
    double* Eb = E + (size_t)b * n * K;
    const int li = lane & 15, lk = lane >> 4;
    const int kcol = min(k0 + li, K - 1);
    const bool wave_on = k0 < K;
    v4f64 acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int jend = min(n, (t0 + nt) * 16);                   // lower triangular: j <= i
    // staging map: thread -> row si of the pass, columns sj + 2u (u < 8) of the chunk
    const int si = threadIdx.x & (kTrmmRows - 1), sj = threadIdx.x >> 7;
    const int gi = t0 * 16 + si, gic = min(gi, n - 1);
    double lreg[8], bz[4];
    const uint64_t seed = RNG ? rng.seeds[b] : 0;
    // MFMA slot (q, lk) of a chunk carries column j0 + 4*lk + q of L (and row j0 + 4*lk + q of Z): a lane's four B operands are
    // then four ADJACENT rows of one sample = the four normals of one Philox call (philox.h), drawn by the lane that consumes them
    // (no cross-lane redistribution).  The LDS panel stores column c in row (c & 3) * 4 + (c >> 2), so that the A-operand
    // reads keep their conflict-free pattern Ls[4q + lk][..].
    auto draw_chunk = [&](int j0) {                             // RNG: bz[q] = N(0,1) number (k0+li)*n + j0+4lk+q of the stream
        // (rows beyond n draw from counters past the sample's range: finite normals that meet the panel's zero columns beyond n, so nothing is zeroed here)
        const int kk = min(k0 + li, K - 1);
#ifdef MPOPIS_DEV_NO_DRAW                                       // dev (tools/kbench_c5.hip): the kernel with the drawing compiled out = what a zero-cost generator would leave
        bz[0] = bz[1] = bz[2] = bz[3] = (double)(kk + j0 + lk) * 1e-3;
#else
        philox_normal_quad(seed, rng.slo, rng.shi, ((uint64_t)kk * n + j0 + 4 * lk) >> 2, sh_tab, bz);             // (4 | n: sample_trmm_fusable)
#endif
    };
    // RNG: L comes as the pre-arranged, zero-filled panel copy the Cholesky kernel wrote (k_potrf_lds, Lpanel): row p of chunk c is LDS row p,
    // so staging is 8 plain loads + 8 plain LDS stores per thread and chunk (the generic path spends ~10 VALU per element on clamps,
    // address arithmetic and the triangle predicate -- on the datapath the Philox / Box-Muller work and the MFMAs share)
    const double* Pb = RNG ? rng.panel + (size_t)b * rng.pstride + (size_t)(sj * 8) * kPanelRows + si : nullptr;
    auto load_chunk = [&](int j0) {
        if (RNG) {
            const double* pc = Pb + (size_t)j0 * kPanelRows;       // chunk j0 / 16: 16 rows of kPanelRows
#pragma unroll
            for (int u = 0; u < 8; ++u) lreg[u] = pc[(size_t)u * kPanelRows];
            draw_chunk(j0);
            return;
        }
#pragma unroll
        // unconditional loads from clamped addresses (a predicated load costs an exec-masked block + vmcnt(0) each);
        // out-of-range / upper-triangle entries are zeroed when the chunk is written to LDS / used
        for (int u = 0; u < 8; ++u) { const int j = min(j0 + sj + 2 * u, n - 1); lreg[u] = Lb[(size_t)gic + (size_t)j * n]; }
        if (RNG) draw_chunk(j0);
        else {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int j = min(j0 + 4 * lk + q, n - 1); bz[q] = Zb[(size_t)j * K + kcol]; }
        }
    };
    load_chunk(0);
    int buf = 0;
    for (int j0 = 0; j0 < jend; j0 += 16, buf ^= 1) {
        if (RNG) {
            if (LD >= kTrmmRows || si < LD) {
#pragma unroll
                for (int u = 0; u < 8; ++u) Ls[buf][sj * 8 + u][si] = lreg[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jc = sj + 2 * u, j = j0 + jc;
                Ls[buf][(jc & 3) * 4 + (jc >> 2)][si] = (gi < n && j < n && j <= gi) ? lreg[u] : 0.0;
            }
        }
        double bc[4];
        // rows of Z beyond n: the generic path clamps its loads and zeroes the operand; the RNG path needs neither -- the panel's columns beyond n
        // are zero and a drawn normal is finite (|z| <= 6.76), so those products vanish on their own
#pragma unroll
        for (int q = 0; q < 4; ++q) bc[q] = (RNG || j0 + 4 * lk + q < n) ? bz[q] : 0.0;
        __syncthreads();
        if (j0 + 16 < jend) load_chunk(j0 + 16);                // prefetch: overlaps the MFMAs below
        const int tfirst = max(0, j0 / 16 - t0);                // row tiles above the chunk's block row are all zero
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            if (t >= tfirst && t < nt) {                        // wave-uniform
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[buf][4 * q + lk][t * 16 + li], bc[q], acc[t], 0, 0, 0);
            }
        }
    }
    const double osc = rng.oscale2 ? sqrt(rng.oscale2[b]) : 1.0;
    if (wave_on && k0 + li < K) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            if (t < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = (t0 + t) * 16 + lk + 4 * r;
                    if (i < n) Eb[(size_t)i * K + k0 + li] = rng.oscale2 ? osc * acc[t][r] : acc[t][r];
                }
            }
        }
    }
}

void launch_trmm_LZ_mfma(const double* L, size_t Lstride, const double* Z, double* E, int B, int n, int K, const int* active, hipStream_t s, const double* oscale2) {
    // row tiles dealt evenly to the row groups (19 tiles at n = 300: 7 / 7 / 5 instead of 8 / 8 / 3 -- the last group stages every chunk of L and Z
    // for its few tiles, so a short group is mostly staging)
    const int nt = (n + 15) / 16, ng = (nt + kTrmmTiles - 1) / kTrmmTiles, tpg = (nt + ng - 1) / ng;
    hipLaunchKernelGGL((k_trmm_LZ_mfma<false>), dim3((K + 63) / 64, ng, B), dim3(256), 0, s, L, Lstride, Z, E, n, K, active,
                       RngArgs{nullptr, 0, 0, nullptr, nullptr, 0, oscale2}, tpg);
}
// E = L * randn(n, K) with the normals drawn inside the kernel (no Z buffer); returns false if the shape needs the 2-kernel path
bool sample_trmm_fusable(int n) { return !(n & 3) && (n + 15) / 16 <= kTrmmTiles; }     // (a lane's four rows = one Philox call)
bool launch_sample_trmm_fused(const double* L, size_t Lstride, double* E, int B, int n, int K, const uint64_t* seeds, uint32_t slo, uint32_t shi,
                              const int* active, hipStream_t s, const double* rng_tab, const double* panel, size_t pstride, const double* oscale2) {
    if (!sample_trmm_fusable(n) || !panel) return false;
    if (n <= 112) hipLaunchKernelGGL((k_trmm_LZ_mfma<true, 112>), dim3((K + 63) / 64, 1, B), dim3(256), 0, s, L, Lstride, (const double*)nullptr, E, n, K, active,
                                     RngArgs{seeds, slo, shi, rng_tab, panel, pstride, oscale2}, 7);
    else hipLaunchKernelGGL((k_trmm_LZ_mfma<true>), dim3((K + 63) / 64, 1, B), dim3(256), 0, s, L, Lstride, (const double*)nullptr, E, n, K, active,
                            RngArgs{seeds, slo, shi, rng_tab, panel, pstride, oscale2}, kTrmmTiles);
    return true;
}
// ---------------------------------------------------------------------------------------------
// Scatter matrix partials.  A workgroup (4 waves) owns a group of <= 28 lower-triangular 16x16 tile pairs
// (7 per wave) and one K split; it streams its k range in chunks staged through LDS as centred rows
// Xs[row][kk] (row stride kc+2 doubles => conflict-free ds_read_b64 in the MFMA operand pattern).
// part[b][split][pair][lane*4 + r]
// ---------------------------------------------------------------------------------------------
constexpr int kPairsPerWave = 7;
constexpr int kPairsPerBlock = 4 * kPairsPerWave;

__device__ __forceinline__ void decode_pair(int q, int* ta, int* tb) {      // q -> (ta >= tb), row-major lower enumeration
    int a = 0;
    while (q >= a + 1) { q -= a + 1; ++a; }
    *ta = a; *tb = q;
}

// SQ = true stages ((x - μ) * rscale[row])^2 instead of (x - μ): the fourth-moment scatter Σ_k z_a² z_b² needed by the
// Schäfer-Strimmer shrinkage intensity (CE's Σ_est = :ss).
// ROWS (cs + ones row in 97..112 => exactly 7 row tiles, KC = 64: every 1-car H = 50 configuration): the 28 tile pairs are dealt to the waves
// as whole tile ROWS of the lower triangle -- wave w owns rows R1 = 3 + w and R2 = 2 - w (7 pairs each) -- so a wave reads tiles 0..R1 once per
// k-step and feeds all its MFMAs from them (4..7 operand tiles instead of 14), and a chunk is stored k-permuted, column k of a row at
// (k & 3) kRowsQ + (k >> 2), so the operands of two consecutive k-steps are one 16-byte ds_read_b128.  Row stride 76 / quarter stride 18 doubles:
// conflict-free in the four 16-lane groups a b128 read is serviced in (MI355X_MICROARCH.md, LDS table).  LDS read time per chunk
// is a fifth of the pair-list form's (224 ds_read_b64 per wave and chunk beside 112 MFMAs).  What bought the time, measured step by step at 64
// trials: rows + b128 alone 86 -> 86 us; + two images / one barrier, staging straight-line 86 -> 84; + staging and the next loads issued between
// the MFMA groups 77; + accumulators resident across chunks 75; + 8 waves (two MFMA streams per SIMD) 74 -- the pair-list form idled the matrix
// pipe through its staging pass and two barriers per chunk, LDS bandwidth was never the limit.  The K range is cut differently into partials (two
// per workgroup, interleaved by k-step group), so Σ' agrees with the pair-list form to rounding (1e-16 relative), not bit for bit.
constexpr int kRowsS = 76, kRowsQ = 18;
typedef double v2f64 __attribute__((ext_vector_type(2)));
template <int R1, int R2, class Side>
__device__ __forceinline__ void wcov_rows_chunk(const double* __restrict__ xl, v4f64 (&acc)[kPairsPerWave], Side&& side) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {                               // (xl points at this wave's half of the chunk: k-steps 8 half .. 8 half + 7)
        side(j);                                                // a quarter of this thread's share of the next chunk's staging, issued in the shadow of these 14 MFMAs
        v2f64 x[R1 + 1];
#pragma unroll
        for (int t = 0; t <= R1; ++t) x[t] = *reinterpret_cast<const v2f64*>(xl + t * 16 * kRowsS + 2 * j);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int tb = 0; tb <= R1; ++tb) acc[tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[R1][h], x[tb][h], acc[tb], 0, 0, 0);
#pragma unroll
            for (int tb = 0; tb <= R2; ++tb) acc[R1 + 1 + tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[R2 < 0 ? 0 : R2][h], x[tb][h], acc[R1 + 1 + tb], 0, 0, 0);
        }
    }
}

template <int KC, bool SQ, bool ROWS = false>
__global__ void __launch_bounds__(ROWS ? 512 : 256, ROWS ? 1 : 2) k_wcov_mfma_partial(const double* __restrict__ X, const double* __restrict__ w, const int32_t* __restrict__ idx,
                                                           const double* __restrict__ mu, const double* __restrict__ rscale,
                                                           double* __restrict__ part, int cs, int K, int m,
                                                           int ksplit, int npairs, const int* active, int aug,
                                                           const double* __restrict__ costp, const unsigned long long* __restrict__ cminp, double neg_inv_lambda) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.z;
    if (active && !active[b]) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int nt = (cs + 15) / 16, rows_pad = nt * 16;
    static_assert(!ROWS || (KC == 64 && !SQ), "row form: 64-column chunks, plain scatter");
    constexpr int S = ROWS ? kRowsS : KC + 1;                   // LDS row stride (doubles); pair-list form: odd strides measured best (tools/kbench)
    constexpr int NTHR = ROWS ? 512 : 256;
    constexpr int kMaxLd = ROWS ? 14 : (KC == 64) ? 28 : 32;    // staged elements per thread and chunk (cs <= 112 resp. 512 rows)
    constexpr int kRowStep = NTHR / KC;
    double* Xs = smem;                                          // [rows_pad][S]
    double* ws = smem + (size_t)rows_pad * S;                   // [KC]
    double* wsl = ws + KC;                                      // [per] (weights-from-costs form) the k range's unnormalised weights
    const double* Xb = X + (size_t)b * cs * K;
    const double* wb = w ? w + (size_t)b * K : nullptr;
    const int32_t* ib = idx ? idx + (size_t)b * K : nullptr;
    const double* mub = mu + (size_t)b * cs;
    // this wave's tile pairs
    int pa[kPairsPerWave], pb[kPairsPerWave];
    const int qbase = blockIdx.y * kPairsPerBlock + wv * kPairsPerWave;
#pragma unroll
    for (int p = 0; p < kPairsPerWave; ++p) {
        if (ROWS) { const int r1 = 3 + (wv & 3), r2 = 2 - (wv & 3); pa[p] = (p <= r1) ? r1 : r2; pb[p] = (p <= r1) ? p : p - r1 - 1; }
        else if (qbase + p < npairs) decode_pair(qbase + p, &pa[p], &pb[p]); else { pa[p] = 0; pb[p] = 0; }   // dummy tile, result dropped
    }
    v4f64 acc[kPairsPerWave];
#pragma unroll
    for (int p = 0; p < kPairsPerWave; ++p) acc[p] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int per = ((m + ksplit - 1) / ksplit + KC - 1) / KC * KC;          // k range per split, multiple of KC
    // (row form: a workgroup of 8 waves covers TWO splits' worth of columns; waves 0-3 / 4-7 take the first / second four k-step pairs of every
    //  chunk and write partial 2 blockIdx.x + half -- two MFMA streams per SIMD, one LDS image pair)
    const int wgper = ROWS ? 2 * per : per;
    const int kbeg = blockIdx.x * wgper, kend = min(m, kbeg + wgper);
    // staging map: this thread always handles column kk of a chunk and rows r0 + u*kRowStep
    const int skk = threadIdx.x % KC, sr0 = threadIdx.x / KC;
    // !SQ: rows are staged UNcentred and the finish kernel subtracts μ μ' Σw (Σ w (x-μ)(x-μ)' = Σ w x x' - μ μ' Σw for
    // μ = Σ w x / Σw; x and μ are O(0.1..1), so the cancellation costs ~1e-16 absolute) -- this frees 56 VGPRs per lane.
    double xreg[kMaxLd], mureg[SQ ? kMaxLd : 1], rsreg[SQ ? kMaxLd : 1], wreg = 0.0;
    if (SQ) {
#pragma unroll
        for (int u = 0; u < kMaxLd; ++u) {
            mureg[u] = mub[min(sr0 + u * kRowStep, cs - 1)];
            rsreg[u] = rscale[(size_t)b * cs + min(sr0 + u * kRowStep, cs - 1)];
        }
    }
    if (costp) {
        // Weights from costs: w_k = exp(-1/λ (c_k - ρ)) (compute_weights, utils.jl:79-86) with ρ = the minimum the rollout kernel accumulated.
        // Left UNnormalised: the moments divide by Σ_k w_k, which comes out of the same pass (ones row x ones row of the augmented scatter),
        // so the separate reweighting launch between rollout and moments disappears.  Done before the main loop, while the staging registers
        // are still free.
        const double rho = cost_unkey(cminp[b]);
        // (stored as sqrt(w_k) = exp(-1/(2λ) (c_k - ρ)): the rows are staged as sqrt(w_k) x_k, see below)
        for (int kq = kbeg + (int)threadIdx.x; kq < kend; kq += NTHR) wsl[kq - kbeg] = exp(0.5 * neg_inv_lambda * (costp[(size_t)b * K + kq] - rho));
        __syncthreads();
    }
    bool kin_cur = false;
    const bool weighted = costp || wb;                          // uniform
    auto load_chunk = [&](int c0) {                             // unconditional loads from clamped addresses
        const int kq = min(c0 + skk, kend - 1);
        const int col = ib ? ib[kq] : kq;
        wreg = costp ? wsl[kq - kbeg] : (wb ? sqrt(wb[col]) : 1.0);      // sqrt(w_k)
#pragma unroll
        for (int u = 0; u < kMaxLd; ++u) xreg[u] = Xb[(size_t)min(sr0 + u * kRowStep, cs - 1) * K + col];
    };
    // per-pair operand bases in the MFMA lane pattern (row = tile*16 + li, column offset lk)
    const double* pA[kPairsPerWave]; const double* pB[kPairsPerWave];
#pragma unroll
    for (int p = 0; p < kPairsPerWave; ++p) { pA[p] = Xs + (size_t)(pa[p] * 16 + li) * S + lk; pB[p] = Xs + (size_t)(pb[p] * 16 + li) * S + lk; }
    // Rows are staged as sqrt(w_k) x_k: Σ_k w_k x_a x_b = Σ_k (sqrt(w_k) x_a)(sqrt(w_k) x_b), so the weight costs one multiply per staged
    // element (28 per thread and chunk) instead of one per MFMA operand (112 per lane and chunk, on the datapath the MFMAs share).
    auto stage_chunk = [&](double* dst, int c0) {               // the chunk held in xreg / wreg -> LDS
        kin_cur = (c0 + skk) < kend;
        const double sw = weighted ? wreg : 1.0;
        if (ROWS) {
            // straight-line form: 7 row tiles = 28 rows per thread exactly (no store guard), rows below 96 are data rows whatever cs in 97..112 is,
            // and a column beyond the k range is zeroed through its weight
            const double swk = kin_cur ? sw : 0.0;
            double* d0 = dst + (size_t)sr0 * S + ((skk & 3) * kRowsQ + (skk >> 2));
#pragma unroll
            for (int u = 0; u < kMaxLd; ++u) {
                const int row = sr0 + u * kRowStep;
                double v = weighted ? xreg[u] * swk : (kin_cur ? xreg[u] : 0.0);
                if (u * kRowStep + kRowStep > 96) v = (row < cs) ? v : ((aug && row == cs) ? swk : 0.0);
                d0[(size_t)u * kRowStep * S] = v;
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < kMaxLd; ++u) {
            const int row = sr0 + u * kRowStep;
            double v = xreg[u];
            if (SQ) { v = (v - mureg[u]) * rsreg[u]; v *= v; }
            if (weighted) v *= sw;
            // zero padded; aug: the first padding row carries ones (times sqrt(w_k)), so row cs of the scatter is Σ_k w_k x_k (the weighted
            // mean comes out of the same pass over X and the separate E·w kernel is not needed) and its diagonal entry is Σ_k w_k
            const int pos = ROWS ? (skk & 3) * kRowsQ + (skk >> 2) : skk;
            if (row < rows_pad) dst[(size_t)row * S + pos] = (kin_cur && row < cs) ? v : ((aug && row == cs && kin_cur) ? sw : 0.0);
        }
    };
    if (kbeg < kend) load_chunk(kbeg);
    if (ROWS) {
        // two LDS images, one barrier per chunk, one workgroup (8 waves) per CU: a wave writes its share of chunk c+1 (loaded during chunk c-1) into the
        // other image and issues the loads of chunk c+2 BETWEEN the MFMA groups of its half of chunk c (56 MFMAs) -- where the single-image form had two
        // barriers and the whole staging pass between two MFMA phases (the matrix pipe idled ~45 % of the kernel)
        double* Xs1 = wsl + (costp ? wgper : 0);
        if (kbeg < kend) { stage_chunk(Xs, kbeg); load_chunk(kbeg + KC); }      // (clamped addresses: a chunk beyond the range loads valid memory)
        __syncthreads();
        // (the loop is instantiated once per wave role, so the accumulators stay in their registers across chunks)
        auto run = [&](auto r1c, auto r2c) {
        int it = 0;
        for (int c0 = kbeg; c0 < kend; c0 += KC, ++it) {
            double* cur = (it & 1) ? Xs1 : Xs;
            double* nxt = (it & 1) ? Xs : Xs1;
            // chunk c+1 sits in xreg / wreg: it is written to the other image, and chunk c+2 loaded into the freed registers, four rows at a time
            // between the MFMA groups of chunk c (straight-line; for the last chunks this stages zeros / loads clamped addresses that nobody reads)
            const bool kin1 = (c0 + KC + skk) < kend;
            const double swk = kin1 ? (weighted ? wreg : 1.0) : 0.0;
            const int kq2 = min(c0 + 2 * KC + skk, kend - 1), col2 = ib ? ib[kq2] : kq2;
            const double wreg2 = costp ? wsl[kq2 - kbeg] : (wb ? sqrt(wb[col2]) : 1.0);
            double* d0 = nxt + (size_t)sr0 * S + ((skk & 3) * kRowsQ + (skk >> 2));
            const double* g0 = Xb + col2;
            auto side = [&](int j) {
#pragma unroll
                for (int uu = 0; uu < 4; ++uu) {
                    const int u = 4 * j + uu;
                    if (u < kMaxLd) {
                        const int row = sr0 + u * kRowStep;
                        double v = weighted ? xreg[u] * swk : (kin1 ? xreg[u] : 0.0);
                        if (u * kRowStep + kRowStep > 96) v = (row < cs) ? v : ((aug && row == cs) ? swk : 0.0);
                        d0[(size_t)u * kRowStep * S] = v;
                        xreg[u] = g0[(size_t)min(row, cs - 1) * K];
                    }
                }
            };
            const double* xl = cur + (size_t)li * S + lk * kRowsQ + 8 * (wv >> 2);
            wcov_rows_chunk<decltype(r1c)::value, decltype(r2c)::value>(xl, acc, side);
            wreg = wreg2;
            __syncthreads();
        }
        };
        switch (wv & 3) {
            case 0: run(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{}); break;
            case 1: run(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{}); break;
            case 2: run(std::integral_constant<int, 5>{}, std::integral_constant<int, 0>{}); break;
            default: run(std::integral_constant<int, 6>{}, std::integral_constant<int, -1>{}); break;
        }
    } else {
        for (int c0 = kbeg; c0 < kend; c0 += KC) {
#ifdef MPOPIS_DEV_NO_STAGE                                      // dev (tools/kbench_c5.hip): only the first chunk is staged -- the MFMA + barrier floor of the kernel
            if (c0 == kbeg) stage_chunk(Xs, c0);
            __syncthreads();
#else
            stage_chunk(Xs, c0);
            __syncthreads();
            if (c0 + KC < kend) load_chunk(c0 + KC);            // next chunk's loads fly during the MFMAs
#endif
#pragma unroll
            for (int kk0 = 0; kk0 < KC; kk0 += 4) {
#pragma unroll
                for (int p = 0; p < kPairsPerWave; ++p)         // sqrt(w_k) x_a  x  sqrt(w_k) x_b
                    acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(pA[p][kk0], pB[p][kk0], acc[p], 0, 0, 0);
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int p = 0; p < kPairsPerWave; ++p) {
        const int q = ROWS ? pa[p] * (pa[p] + 1) / 2 + pb[p] : qbase + p;
        if (q < npairs) {
            const int split = ROWS ? 2 * blockIdx.x + (wv >> 2) : blockIdx.x;
            double* pp = part + (((size_t)b * ksplit + split) * npairs + q) * 256 + lane * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) pp[r] = acc[p][r];
        }
    }
}

// S = (1/den) sum_splits part + ridge*I, written symmetric (lower triangle drives both halves)
// mu_corr != nullptr: the partials are the UNcentred scatter Σ w x x'; subtract μ μ' wtot (wtot = Σ of the weights used:
// Σ_k w_k, or m for unweighted gathered columns)
__global__ void __launch_bounds__(256) k_wcov_mfma_finish(const double* __restrict__ part, const double* __restrict__ w, double* __restrict__ Sg,
                                                          int cs, int K, int ksplit, int npairs, double den, double ridge, const int* active,
                                                          const double* __restrict__ mu_corr, double wtot_unweighted,
                                                          double* __restrict__ mu_aug, double* __restrict__ u_add, const double* __restrict__ wsum,
                                                          unsigned long long* cmin_reset, const double* __restrict__ mu_shift) {
    MPOPIS_HI_PRIO();
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    __shared__ double sh[4];
    __shared__ double sden, swtot;
    __shared__ double smu[32];
    const int q = blockIdx.x;
    int ta, tb;
    decode_pair(q, &ta, &tb);
    const int e = threadIdx.x;                                  // e = lane*4 + r  (lane 0..63, r 0..3)
    const int lane = e >> 2, r = e & 3;
    const int ia = ta * 16 + (lane >> 4) + 4 * r, ibb = tb * 16 + (lane & 15);
    // Every sum over the K-split partials this block needs -- its own entry, the 32 mean entries, the sum of weights -- is loaded BEFORE the first
    // barrier: the kernel is three dependent global round trips otherwise (a few workgroups' worth of data, ~2 us each).
    // sum over the K splits in split order (deterministic), eight loads in flight at a time: as a plain loop every addition waits for its own
    // load -- ksplit dependent L2 round trips, 24 us at 8 resident trials (32 splits), more than the partial kernel itself
    auto split_sum = [&](size_t pair, int elem) -> double {
        const double* src = part + ((size_t)b * ksplit * npairs + pair) * 256 + elem;
        const size_t stride = (size_t)npairs * 256;
        double acc = 0.0;
        for (int sp0 = 0; sp0 < ksplit; sp0 += 8) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = src[(size_t)min(sp0 + u, ksplit - 1) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (sp0 + u < ksplit) acc += t[u];
        }
        return acc;
    };
    double v = split_sum(q, e);
    double msum = 0.0;
    if (mu_aug && threadIdx.x < 32) {
        // μ_j = (row cs of the augmented scatter)_j / Σw: element (cs % 16, j % 16) of tile pair (cs / 16, j / 16)
        const int t = (threadIdx.x < 16) ? ta : tb, jl = threadIdx.x & 15;
        const int taug = cs >> 4, il = cs & 15;
        const int qa = taug * (taug + 1) / 2 + t, ea = (((il & 3) * 16 + jl) << 2) + (il >> 2);
        msum = split_sum(qa, ea);
    }
    if (cmin_reset) {
        // weights-from-costs form: Σ_k w_k is the (ones row, ones row) entry of the augmented scatter
        if (threadIdx.x == 32) {
            const int taug = cs >> 4, il = cs & 15;
            const int qa = taug * (taug + 1) / 2 + taug, ea = (((il & 3) * 16 + il) << 2) + (il >> 2);
            const double t = split_sum(qa, ea);
            swtot = t; sden = (den == 0.0) ? t : den;
            if (blockIdx.x == 0) cmin_reset[b] = ~0ull;                        // the next rollout launch starts a fresh minimum
        }
        __syncthreads();
    } else if (w && wsum) { if (threadIdx.x == 0) { swtot = wsum[b]; sden = (den == 0.0) ? swtot : den; } __syncthreads(); }   // precomputed by k_weights
    else if (w) {                                               // Σ_k w_k (ProbabilityWeights)
        double sacc = 0.0;
        for (int k = threadIdx.x; k < K; k += 256) sacc += w[(size_t)b * K + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sacc;
        __syncthreads();
        if (threadIdx.x == 0) { swtot = sh[0] + sh[1] + sh[2] + sh[3]; sden = (den == 0.0) ? swtot : den; }
        __syncthreads();
    } else { if (threadIdx.x == 0) { sden = den; swtot = wtot_unweighted; } __syncthreads(); }
    const double inv = 1 / sden;
    if (mu_aug) {
        if (threadIdx.x < 32) {
            const int jl = threadIdx.x & 15, taug = cs >> 4;
            const double m = msum / swtot;
            smu[threadIdx.x] = m;
            if (ta == taug && threadIdx.x >= 16 && tb * 16 + jl < cs) {
                const double mfull = mu_shift ? m + mu_shift[(size_t)b * cs + tb * 16 + jl] : m;      // data were shifted by mu_shift (launch_gather_cols): Σ′ is unaffected
                mu_aug[(size_t)b * cs + tb * 16 + jl] = mfull;
                if (u_add) u_add[(size_t)b * cs + tb * 16 + jl] += mfull;      // pol.U += μ′ (:734), one writer per entry
            }
        }
        __syncthreads();
    }
    if (ia >= cs || ibb >= cs) return;
    if (ta == tb && ibb > ia) return;
    if (mu_aug) v = fma(-smu[ia - ta * 16] * swtot, smu[16 + ibb - tb * 16], v);
    else if (mu_corr) v = fma(-mu_corr[(size_t)b * cs + ia] * swtot, mu_corr[(size_t)b * cs + ibb], v);
    v = v * inv;
    if (ia == ibb) v += ridge;
    Sg[(size_t)b * cs * cs + (size_t)ia + (size_t)ibb * cs] = v;
    Sg[(size_t)b * cs * cs + (size_t)ibb + (size_t)ia * cs] = v;
}

// rs[b][a] = 1/sqrt(S[b][a][a])
__global__ void __launch_bounds__(256) k_inv_sd(const double* __restrict__ Sg, double* __restrict__ rs, int cs, const int* active) {
    MPOPIS_HI_PRIO();
    const int b = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x;
    if ((active && !active[b]) || a >= cs) return;
    rs[(size_t)b * cs + a] = 1.0 / sqrt(Sg[(size_t)b * cs * cs + (size_t)a * (cs + 1)]);
}
// LinearShrinkage(DiagonalUnequalVariance(), :ss) [third-party CovarianceEstimation.jl, restated from Schäfer & Strimmer 2005,
// UNPINNED like the oracle's cov_ss_cols]: on standardised data r_ab = S_ab/(sd_a sd_b),
//   λ* = Σ_{a≠b} Var^(r_ab) / Σ_{a≠b} r_ab²,  Var^(r_ab) = n/(n-1)³ (Q_ab - n r_ab²),  Q_ab = Σ_k z_ka² z_kb²;
// S ← λ* diag(S) + (1-λ*) S  (+ ridge on the diagonal).  One workgroup per slot.
// standardized = 0 gives the :lw variant (same intensity on the unstandardised scatter: rs == 1, Q = Σ_k xc_a² xc_b²).
__global__ void __launch_bounds__(256) k_ss_shrink(double* __restrict__ Sg, const double* __restrict__ Q, const double* __restrict__ rs,
                                                   int cs, int m, double ridge, const int* active) {
    MPOPIS_HI_PRIO();
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    __shared__ double sh[8];
    __shared__ double lam_sh;
    double* S = Sg + (size_t)b * cs * cs;
    const double* Qb = Q + (size_t)b * cs * cs;
    const double* r = rs + (size_t)b * cs;
    double num = 0.0, den = 0.0;
    for (int e = threadIdx.x; e < cs * cs; e += 256) {
        const int a = e % cs, c = e / cs;
        if (a == c) continue;
        const double rab = S[e] * r[a] * r[c];
        num += Qb[e] - (double)m * rab * rab;
        den += rab * rab;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { num += __shfl_xor(num, o, 64); den += __shfl_xor(den, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = num; sh[4 + (threadIdx.x >> 6)] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        num = (sh[0] + sh[1] + sh[2] + sh[3]) * ((double)m / ((double)(m - 1) * (m - 1) * (m - 1)));
        den = sh[4] + sh[5] + sh[6] + sh[7];
        double lam = den > 0 ? num / den : 1.0;
        lam_sh = fmin(fmax(lam, 0.0), 1.0);
    }
    __syncthreads();
    const double keep = 1 - lam_sh;
    for (int e = threadIdx.x; e < cs * cs; e += 256) {
        const int a = e % cs, c = e / cs;
        S[e] = (a == c) ? S[e] + ridge : S[e] * keep;
    }
}
// LinearShrinkage(DiagonalCommonVariance(), :rblw / :oas): closed forms in tr(S), tr(S²) (Chen et al. 2010), F = tr(S)/p I
__global__ void __launch_bounds__(256) k_common_shrink(double* __restrict__ Sg, int cs, int m, int oas, double ridge, const int* active) {
    MPOPIS_HI_PRIO();
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    __shared__ double sh[8];
    __shared__ double lam_sh, f_sh;
    double* S = Sg + (size_t)b * cs * cs;
    double tr = 0.0, tr2 = 0.0;
    for (int e = threadIdx.x; e < cs * cs; e += 256) { const double v = S[e]; tr2 = fma(v, v, tr2); if (e % cs == e / cs) tr += v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { tr += __shfl_xor(tr, o, 64); tr2 += __shfl_xor(tr2, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = tr; sh[4 + (threadIdx.x >> 6)] = tr2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tr = sh[0] + sh[1] + sh[2] + sh[3]; tr2 = sh[4] + sh[5] + sh[6] + sh[7];
        const double p = cs, n = m, dd = tr2 - tr * tr / p;
        double lam = oas ? ((1 - 2 / p) * tr2 + tr * tr) / ((n + 1 - 2 / p) * dd) : ((n - 2) / n * tr2 + tr * tr) / ((n + 2) * dd);
        lam_sh = (dd > 0) ? fmin(fmax(lam, 0.0), 1.0) : 1.0;
        f_sh = tr / p;
    }
    __syncthreads();
    const double lam = lam_sh, f = f_sh;
    for (int e = threadIdx.x; e < cs * cs; e += 256) {
        const bool dg = (e % cs) == (e / cs);
        S[e] = (1 - lam) * S[e] + (dg ? lam * f + ridge : 0.0);
    }
}
void launch_common_shrink(double* S, int B, int cs, int m, int oas, double ridge, const int* active, hipStream_t s) {
    hipLaunchKernelGGL(k_common_shrink, dim3(B), dim3(256), 0, s, S, cs, m, oas, ridge, active);
}
__global__ void __launch_bounds__(256) k_fill_f64(double* p, double v, size_t n) { const size_t i = blockIdx.x * (size_t)256 + threadIdx.x; if (i < n) p[i] = v; }
void launch_fill_f64(double* p, double v, size_t n, hipStream_t s) { hipLaunchKernelGGL(k_fill_f64, dim3((n + 255) / 256), dim3(256), 0, s, p, v, n); }
void launch_ss_shrink(double* S, const double* Q, double* rs_ws, int B, int cs, int m, double ridge, const int* active, hipStream_t s) {
    hipLaunchKernelGGL(k_ss_shrink, dim3(B), dim3(256), 0, s, S, Q, rs_ws, cs, m, ridge, active);
}
void launch_inv_sd(const double* S, double* rs, int B, int cs, const int* active, hipStream_t s) {
    hipLaunchKernelGGL(k_inv_sd, dim3((cs + 255) / 256, B), dim3(256), 0, s, S, rs, cs, active);
}

static int wcov_kc(int cs) { return cs <= 112 ? 64 : 16; }
size_t wcov_mfma_workspace_doubles(int B, int cs, int ksplit) {
    const int nt = (cs + 15) / 16;
    return (size_t)B * ksplit * (nt * (nt + 1) / 2) * 256;
}
bool wcov_mfma_can_emit_mean(int cs) { return (cs & 15) != 0; }      // needs a padding row for the ones
void launch_wcov_mfma(const double* X, const double* w, const int32_t* idx, int m, const double* mu, double* S, double* part,
                      int B, int cs, int K, int ksplit, int sel_batch, double den, double ridge, const int* active, hipStream_t s, const double* rscale,
                      double* mu_out, double* u_add, const double* wsum, const double* cost, unsigned long long* cmin, double neg_inv_lambda,
                      const double* mu_shift) {
    const int aug = (mu_out && !rscale && wcov_mfma_can_emit_mean(cs)) ? 1 : 0;   // mu_out: also produce μ = Σ w x / Σw (mu is then unused)
    const bool from_cost = cost && cmin && aug && !idx && wcov_weights_from_cost_ok(cs, K, ksplit);
    if (!from_cost) { cost = nullptr; cmin = nullptr; }
    const int nt = (cs + 15) / 16, npairs = nt * (nt + 1) / 2;
    const int kc = wcov_kc(cs);
    const int per = ((m + ksplit - 1) / ksplit + kc - 1) / kc * kc;
    static const int env_rows = [] { const char* e = getenv("MPOPIS_WCOV_ROWS"); return e ? atoi(e) : 1; }();
    // row form: 7 row tiles (kc == 64 then), an even number of partials, and enough 8-wave workgroups for most of the CUs (measured: wins from 16
    // resident C5 trials up -- 30 -> 28 us at 16, 86 -> 74 us at 64 -- and loses ~5 us at 8, where it fields 128 workgroups)
    // The choice goes by the handle's WHOLE batch (sel_batch), not by this launch's share of it: the part-chains of a multi-stream schedule are in
    // flight together, and a slot's result must not depend on the schedule (the two forms agree to rounding, not bit for bit).
    const bool rows = env_rows && nt == 7 && !rscale && !(ksplit & 1) && sel_batch >= 0 && (long long)(sel_batch > 0 ? sel_batch : B) * (ksplit / 2) >= (env_rows > 1 ? 1 : 192);
    const size_t lds = ((size_t)nt * 16 * (rows ? 2 * kRowsS : kc + 1) + kc + (from_cost ? (rows ? 2 * per : per) : 0)) * sizeof(double);
    static std::atomic<unsigned long long> seen[5];
    ensure_dyn_lds((const void*)k_wcov_mfma_partial<64, false, true>, 160 * 1024, seen[4]);
    ensure_dyn_lds((const void*)k_wcov_mfma_partial<64, false>, 96 * 1024, seen[0]);
    ensure_dyn_lds((const void*)k_wcov_mfma_partial<16, false>, 96 * 1024, seen[1]);
    ensure_dyn_lds((const void*)k_wcov_mfma_partial<64, true>, 96 * 1024, seen[2]);
    ensure_dyn_lds((const void*)k_wcov_mfma_partial<16, true>, 96 * 1024, seen[3]);
    const dim3 grid(ksplit, (npairs + kPairsPerBlock - 1) / kPairsPerBlock, B);
    if (rscale) {
        if (kc == 64) hipLaunchKernelGGL((k_wcov_mfma_partial<64, true>), grid, dim3(256), lds, s, X, w, idx, mu, rscale, part, cs, K, m, ksplit, npairs, active, aug, cost, cmin, neg_inv_lambda);
        else          hipLaunchKernelGGL((k_wcov_mfma_partial<16, true>), grid, dim3(256), lds, s, X, w, idx, mu, rscale, part, cs, K, m, ksplit, npairs, active, aug, cost, cmin, neg_inv_lambda);
    } else {
        if (rows)          hipLaunchKernelGGL((k_wcov_mfma_partial<64, false, true>), dim3(ksplit / 2, 1, B), dim3(512), lds, s, X, w, idx, mu, rscale, part, cs, K, m, ksplit, npairs, active, aug, cost, cmin, neg_inv_lambda);
        else if (kc == 64) hipLaunchKernelGGL((k_wcov_mfma_partial<64, false>), grid, dim3(256), lds, s, X, w, idx, mu, rscale, part, cs, K, m, ksplit, npairs, active, aug, cost, cmin, neg_inv_lambda);
        else          hipLaunchKernelGGL((k_wcov_mfma_partial<16, false>), grid, dim3(256), lds, s, X, w, idx, mu, rscale, part, cs, K, m, ksplit, npairs, active, aug, cost, cmin, neg_inv_lambda);
    }
    hipLaunchKernelGGL(k_wcov_mfma_finish, dim3(npairs, B), dim3(256), 0, s, part, w, S, cs, K, ksplit, npairs, den, ridge, active,
                       rscale ? (const double*)nullptr : mu, (double)m, aug ? mu_out : (double*)nullptr, aug ? u_add : (double*)nullptr, wsum, cmin, aug ? mu_shift : (const double*)nullptr);
}
// the weights-from-costs form needs the ones row (cs not a multiple of 16) and keeps a split's weights in LDS
bool wcov_weights_from_cost_ok(int cs, int K, int ksplit) {
    const int kc = wcov_kc(cs);
    const int per = ((K + ksplit - 1) / ksplit + kc - 1) / kc * kc;
    return wcov_mfma_can_emit_mean(cs) && per <= 1024;
}

}  // namespace mpopis
