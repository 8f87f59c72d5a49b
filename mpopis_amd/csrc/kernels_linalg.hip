// kernels_linalg.hip -- the small dense FP64 linear algebra the reference gets from LAPACK via
// Distributions/PDMats/StatsBase, restated as device kernels (one workgroup per trial slot):
//   MvNormal(Σ') -> PDMat -> cholesky(Σ')            src/mppi_mpopi_policies.jl:192,307,352,447,551-553,650,723,796
//   invcov(P) as used in the control cost             :194,309,353,449,555,651,725,798 (only the row γ U_orig' Σ⁻¹ is needed)
//   mean(elite, dims=2) / mean of resampled columns / CMA δw (gather + mean)  :465,:807,:573-576
// (the covariance contractions live in kernels_mfma.hip)
#include "engine.h"
#include "linalg_diag.h"

namespace mpopis {

// ---------------------------------------------------------------------------------------------
// Blocked right-looking Cholesky, one workgroup per matrix, panel width 16 (k_potrf_lds: working copy in LDS, cs <= 128, all
// 1-car configs; k_potrf_global: working copy = the output buffer, L2 resident, cs = 300 for 3 cars).  Per panel:
//   (1) one wave factors the 16x16 diagonal block (linalg_diag.h, diag16_factor: 4x4 sub-blocks, one LDS exchange each),
//   (2) the panel below it is L21 = A21 * L11^-T as a blocked forward substitution on the matrix cores (panel_solve_tile:
//       7 x v_mfma_f64_16x16x4 per 16-row tile against the 4x4 diagonal inverses and the sub-diagonal blocks of L11),
//   (3) the rank-16 trailing update runs on the matrix cores too, 4 MFMAs per 16x16 tile, read-modify-write; the wave that owns
//       the next diagonal block updates it first and factors it while the others finish the update (look-ahead).
// scale[b] (nullable) multiplies A first (CMA: MvNormal(σ²Σ), :551).  On a non-positive pivot
// status[b] = MPOPIS_ERR_NOT_PD and active[b] = 0 (the reference throws PosDefException).
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// LDS-resident Cholesky for n <= 128 (every 1-car configuration), one workgroup of 8 waves per matrix, whole matrix in LDS.
// The panel step is bound by its two serial pieces, both in linalg_diag.h: the diagonal block (one wave, ~2.0 us) and the panel
// solve (blocked substitution on the matrix cores); the trailing update overlaps the next diagonal block (look-ahead).
// History at n = 100: per-pivot diagonal block + per-row substitution 63 us -> 4x4 sub-blocked block with explicit inverse 39 us
// -> without the inverse (blocked MFMA substitution), one Newton step per pivot, branch-free row selection 29 us.
// ---------------------------------------------------------------------------------------------
// Lpanel (nullable): second copy of the factor in the layout the fused sampler stages through LDS (kernels_mfma.hip, k_trmm_LZ_mfma<true>):
// [chunk = j / 16][p = (j & 3) 4 + ((j & 15) >> 2)][i < 128], zero above the diagonal and beyond n -- the sampler then copies whole rows
// without address clamps or triangle predicates.
__global__ void __launch_bounds__(512) k_potrf_lds(const double* __restrict__ A, size_t Astride, double* __restrict__ Lout,
                                                   int n, int npad, const double* scale, int* status, int* active,
                                                   double* __restrict__ Lpanel, size_t pstride) {
    MPOPIS_HI_PRIO();
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int failed;
    __shared__ DiagScratch dsh;
    constexpr int NTHR = 512, NW = NTHR / 64;
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double* Ab = A + (size_t)b * Astride;
    double* Lb = Lout + (size_t)b * n * n;
    const double sc = scale ? scale[b] : 1.0;
    double* W = smem;                                  // [npad][npad] column-major, identity tail beyond n
    const int ldw = npad, m = npad;
    if (tid == 0) failed = 0;
    {   // copy in the lower triangle only (see k_potrf): column pair c (columns c and m-1-c) holds m+1 triangle entries
        const int npairs2 = m / 2, tchunks = (m + 1 + 63) / 64;
        for (int c0 = wv * 4; c0 < npairs2; c0 += NW * 4) {
            for (int tc = 0; tc < tchunks; tc += 2) {
                double av[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = min(c0 + (u >> 1), npairs2 - 1), t = min((tc + (u & 1)) * 64 + lane, m);
                    const int j = (t < m - c) ? c : m - 1 - c, i = (t < m - c) ? c + t : m - 1 - c + (t - (m - c));
                    av[u] = Ab[(size_t)min(i, n - 1) + (size_t)min(j, n - 1) * n];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + (u >> 1), t = (tc + (u & 1)) * 64 + lane;
                    if (c < npairs2 && t <= m && tc + (u & 1) < tchunks) {
                        const int j = (t < m - c) ? c : m - 1 - c, i = (t < m - c) ? c + t : m - 1 - c + (t - (m - c));
                        W[(size_t)i + (size_t)j * ldw] = (i < n && j < n) ? sc * av[u] : ((i == j) ? 1.0 : 0.0);
                    }
                }
            }
        }
    }
    __syncthreads();
    const int li = lane & 15, lk = lane >> 4;
    auto factor_diag_inv = [&](int j0) {
        const int rounds = (j0 + kNB >= m) ? min(4, (n - j0 + 3) / 4) : 4;     // last block: only the sub-blocks that hold matrix rows
        const bool bad = diag16_factor(lane,
            [&](int i, int c) { return W[(size_t)(j0 + i) + (size_t)(j0 + c) * ldw]; },
            [&](int i, int c, double v) { W[(size_t)(j0 + i) + (size_t)(j0 + c) * ldw] = v; }, dsh, rounds);
        if (bad && lane == 0) failed = 1;
    };
    // rows r0 .. r0+15 of the panel below the diagonal block: L21 tile = A21 tile * L11^-T on the matrix cores (in place: every lane
    // writes back exactly the entries it read)
    auto panel_tile = [&](const PanelOps& o, int j0, int r0) {
        double a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = W[(size_t)(r0 + li) + (size_t)(j0 + q * 4 + lk) * ldw];
        panel_solve_tile(o, a);
#pragma unroll
        for (int q = 0; q < 4; ++q) W[(size_t)(r0 + li) + (size_t)(j0 + q * 4 + lk) * ldw] = a[q];
    };
    auto trail_pair = [&](int j0, int t1, int q) {
        int ta = 0, qq = q;
        while (qq >= ta + 1) { qq -= ta + 1; ++ta; }
        const int r0 = (t1 + ta) * 16, c0 = (t1 + qq) * 16;
        v4f64_l acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int col = j0 + kk * 4 + lk;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(W[(size_t)(c0 + li) + (size_t)col * ldw], W[(size_t)(r0 + li) + (size_t)col * ldw], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = r0 + li, c = c0 + lk + 4 * r;
            if (i >= c) W[(size_t)i + (size_t)c * ldw] -= acc[r];
        }
    };
    if (wv == 0) factor_diag_inv(0);
    __syncthreads();
    for (int j0 = 0; j0 < m; j0 += kNB) {
        if (failed) break;
        const int i1 = j0 + kNB, ntile = (m - i1) / 16, t1 = i1 / 16;
        if (wv < ntile) {
            const PanelOps o = panel_solve_operands(lane, dsh);
            for (int t = wv; t < ntile; t += NW) panel_tile(o, j0, i1 + 16 * t);
        }
        __syncthreads();
        if (ntile > 0) {
            const int npair = ntile * (ntile + 1) / 2;
            if (wv == 0) { trail_pair(j0, t1, 0); factor_diag_inv(i1); }         // look-ahead: next diagonal block while the others update
            else for (int q = wv; q < npair; q += NW - 1) trail_pair(j0, t1, q);
        }
        __syncthreads();
    }
    if (failed) {
        if (tid == 0) { if (status) status_raise(&status[b], MPOPIS_ERR_NOT_PD); if (active) active[b] = 0; }
        return;
    }
    for (int j = wv; j < n; j += NW)
        for (int i = lane; i < n; i += 64) Lb[(size_t)i + (size_t)j * n] = (i >= j) ? W[(size_t)i + (size_t)j * ldw] : 0.0;
    if (Lpanel) {
        double* Lp = Lpanel + (size_t)b * pstride;
        const int nch = (n + 15) / 16;
        for (int e = tid; e < nch * 16 * kPanelRows; e += NTHR) {
            const int i = e & (kPanelRows - 1), pr = (e / kPanelRows) & 15, c = e / (16 * kPanelRows);
            const int j = c * 16 + (pr & 3) * 4 + (pr >> 2);                  // p = (jc & 3) 4 + (jc >> 2)  <=>  jc = 4 (p & 3) + (p >> 2)
            Lp[e] = (i < n && j < n && i >= j) ? W[(size_t)i + (size_t)j * ldw] : 0.0;
        }
    }
}

// Global-memory variant for matrices that do not fit in LDS (cs = 300: three cars), one workgroup of 16 waves: the working
// matrix is the output buffer (L2 resident); the solved panel strip P[m][17] -- both operands of the rank-16 trailing update --
// lives in LDS, only the trailing read-modify-write goes to L2.  Diagonal blocks and panel solves as in k_potrf_lds.
__global__ void __launch_bounds__(1024) k_potrf_global(const double* __restrict__ A, size_t Astride, double* __restrict__ Lout,
                                                       int n, const double* scale, int* status, int* active, int* redo) {
    MPOPIS_HI_PRIO();
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int failed;
    __shared__ DiagScratch dsh;
    constexpr int NTHR = 1024, NW = NTHR / 64, kPS = kNB + 1;
    const int b = blockIdx.x;
    if (redo) {                                         // fall-back pass behind k_potrf_coop: only the slots whose cluster gave up
        if (!redo[b]) return;                           // (the common case: nothing to do)
        __syncthreads();
        if (threadIdx.x == 0) redo[b] = 0;
    }
    if (active && !active[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const double* Ab = A + (size_t)b * Astride;
    double* W = Lout + (size_t)b * n * n;
    const double sc = scale ? scale[b] : 1.0;
    const int ldw = n, m = n;
    double* P = smem;                                   // [m][17] solved panel strip
    if (tid == 0) failed = 0;
    for (int e0 = tid; e0 < m * m; e0 += NTHR * 8) {    // W = lower(sc * A), zeros above the diagonal
        double av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = min(e0 + u * NTHR, m * m - 1); av[u] = Ab[(size_t)(e % m) + (size_t)(e / m) * n]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * NTHR;
            if (e < m * m) { const int i = e % m, j = e / m; W[(size_t)i + (size_t)j * ldw] = (i >= j) ? sc * av[u] : 0.0; }
        }
    }
    __threadfence_block();
    __syncthreads();
    auto factor_diag_inv = [&](int j0) {
        int ln = lane;                                   // opaque (see k_potrf_reg): keeps the routine's 20 selection weights out of the panel loop's live set
        asm volatile("" : "+v"(ln));
        const bool bad = diag16_factor(ln,
            [&](int i, int c) { return (j0 + i < m && j0 + c < m) ? W[(size_t)(j0 + i) + (size_t)(j0 + c) * ldw] : ((i == c) ? 1.0 : 0.0); },
            [&](int i, int c, double v) { if (j0 + i < m && j0 + c < m) W[(size_t)(j0 + i) + (size_t)(j0 + c) * ldw] = v; }, dsh);
        if (bad && lane == 0) failed = 1;
    };
    auto panel_tile = [&](const PanelOps& o, int j0, int r0) {   // L21 tile = A21 tile * L11^-T on the matrix cores; result to W (L2) and P (LDS)
        const int ra = r0 + li;
        double a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int col = j0 + q * 4 + lk; a[q] = (ra < m && col < m) ? W[(size_t)ra + (size_t)col * ldw] : 0.0; }
        panel_solve_tile(o, a);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = q * 4 + lk;
            if (ra < m) { P[(size_t)ra * kPS + c] = a[q]; if (j0 + c < m) W[(size_t)ra + (size_t)(j0 + c) * ldw] = a[q]; }
        }
    };
    auto trail_pair = [&](int j0, int t1, int q) {       // one 16x16 tile of the rank-16 trailing update: pair q -> tile (t1 + ta, t1 + tb), ta >= tb
        int ta = 0, qq = q;
        while (qq >= ta + 1) { qq -= ta + 1; ++ta; }
        const int r0 = (t1 + ta) * 16, c0 = (t1 + qq) * 16;
        v4f64_l acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ra = r0 + li, rb = c0 + li;
            const double av = (ra < m) ? P[(size_t)ra * kPS + kk * 4 + lk] : 0.0;
            const double bv = (rb < m) ? P[(size_t)rb * kPS + kk * 4 + lk] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc, 0, 0, 0);     // transposed tile: lanes run along i
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = r0 + li, c = c0 + lk + 4 * r;
            if (i < m && c < m && i >= c) W[(size_t)i + (size_t)c * ldw] -= acc[r];
        }
    };
    if (wv == 0) factor_diag_inv(0);
    __threadfence_block();
    __syncthreads();
    for (int j0 = 0; j0 < m; j0 += kNB) {
        if (failed) break;
        const int i1 = j0 + kNB;                          // first row below the panel (>= m after the last, possibly partial, panel)
        const int ntile = (i1 < m) ? (m - i1 + 15) / 16 : 0, t1 = i1 / 16;
        if (wv < ntile) {
            const PanelOps o = panel_solve_operands(lane, dsh);
            for (int t = wv; t < ntile; t += NW) panel_tile(o, j0, i1 + 16 * t);
        }
        __threadfence_block();
        __syncthreads();
        if (ntile > 0) {
            const int npair = ntile * (ntile + 1) / 2;
            if (wv == 0) { trail_pair(j0, t1, 0); __threadfence_block(); factor_diag_inv(i1); }   // look-ahead: next diagonal block first
            else for (int q = wv; q < npair; q += NW - 1) trail_pair(j0, t1, q);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (failed && tid == 0) { if (status) status_raise(&status[b], MPOPIS_ERR_NOT_PD); if (active) active[b] = 0; }
}

// ---------------------------------------------------------------------------------------------
// Register-resident variant for 128 < n <= 304 (cs = 300: three cars), one workgroup of 8 waves per matrix.  The lower triangle of a
// 304 x 304 matrix is 190 tiles of 16 x 16 -- 361 KB, more than a CU's LDS, but a CU's register file is 512 KB: every wave keeps up
// to 17 tiles in the accumulator layout of v_mfma_f64_16x16x4 (lane (li, lk), element q <-> row li, column lk + 4 q) for the whole
// factorisation, 136 of its 256 registers at two waves per SIMD, and its first seven tiles -- block columns 0 .. 3, done after the
// fourth panel -- in a private corner of LDS (the diagonal-block routine needs ~110 registers of its own).  Nothing of the working
// matrix ever moves between waves: per panel j only the solved panel strip (both operands of the rank-16 update, 304 x 16) goes
// through LDS, and the 16 x 16 diagonal block through the scratch of diag16_factor.  Tile number t (column-major order of the lower
// triangle) lives in slot t / 8 of wave t % 8, so the trailing tiles of every stage are spread evenly (+-1), a wave's active tiles
// are always the tail of its slot list and its tiles of one block column are consecutive slots.  Per panel j:
//   A  every wave solves its (at most three) tiles of block column j against the diagonal block, three independent chains of
//      panel_solve_tile interleaved (the tile IS the operand), and writes them to the strip and to the output   | barrier
//   B  the owner of diagonal block j+1 updates it and factors it at raised priority (look-ahead); every wave updates its trailing
//      tiles from the strip in groups of four slots of straight-line code (slots that are not active read the strip's zero block):
//      product of the two strip rows first, then subtracted -- the arithmetic of k_potrf_global in the same order, so the factor
//      has the same bits                                                                                        | barrier
// The barriers wait for LDS only: the output stores drain behind the arithmetic, nothing in the kernel reads them back.
// Against k_potrf_coop at n = 300: no hand-offs between CUs (19 x ~8 us there); one CU's matrix rate is ample (4560 MFMAs).
// ---------------------------------------------------------------------------------------------
#ifdef POTRF_PROF
__device__ unsigned long long g_rprof[8 * 32 * 6];
void debug_read_rprof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rprof), sizeof(g_rprof)); }
#define RPROF(j_, k_) do { if (b == 0 && lane == 0) g_rprof[(wv * 32 + (j_)) * 6 + (k_)] = wall_clock64(); } while (0)
#else
#define RPROF(j_, k_) do { } while (0)
#endif
constexpr int kRegWaves = 8, kRegSlots = 24, kRegLdsSlots = 7, kRegMinPan = 16, kRegMaxPan = 19, kRegPS = kNB + 1;
static_assert(kRegMaxPan * (kRegMaxPan + 1) / 2 <= kRegWaves * kRegSlots, "every tile of the lower triangle needs a slot");
static_assert(kRegMaxPan < 32 && kRegSlots % 4 == 0, "tile key = 32 * block column + block row; update groups of four slots");
size_t potrf_reg_lds_bytes(int n) {
    const int npad = (n + kNB - 1) / kNB * kNB;
    return ((size_t)npad * kRegPS + (size_t)kRegWaves * kRegLdsSlots * 256) * sizeof(double);
}
__global__ void __launch_bounds__(64 * kRegWaves) k_potrf_reg(const double* __restrict__ A, size_t Astride, double* __restrict__ Lout,
                                                              int n, const double* scale, int* status, int* active) {
    __builtin_amdgcn_s_setprio(2);
    extern __shared__ __attribute__((aligned(16))) double smem[];              // strip[npad][17], then the LDS tile slots [wave][slot][q][lane]
    __shared__ DiagScratch dsh;
    __shared__ double dtile[kNB][kNB + 1];
    __shared__ int failed;
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npan = (n + kNB - 1) / kNB, npad = npan * kNB;
    const double* Ab = A + (size_t)b * Astride;
    double* Lb = Lout + (size_t)b * n * n;
    double* P = smem;
    double* Tl = smem + (size_t)npad * kRegPS + (size_t)wv * kRegLdsSlots * 256 + lane;      // this lane's entries of LDS slot s: Tl[(s * 4 + q) * 64]
    const double sc = scale ? scale[b] : 1.0;
    if (tid == 0) failed = 0;
    RPROF(31, 0);
    // the strip's rows 0 .. 15 are never written (block row 0 holds only the first diagonal block): the zero block
    for (int e = tid; e < kNB * kRegPS; e += 64 * kRegWaves) P[e] = 0.0;
    // slot s <-> tile number 8 s + wv; key = 32 * block column + block row, -1: no tile
    int key[kRegSlots];
    {
        int c = 0, pp = wv;
#pragma unroll
        for (int s = 0; s < kRegSlots; ++s) {
            while (c < npan && pp >= npan - c) { pp -= npan - c; ++c; }
            key[s] = (c < npan) ? 32 * c + c + pp : -1;
            pp += kRegWaves;
        }
    }
    v4f64_l tile[kRegSlots - kRegLdsSlots];
    auto get = [&](int s) -> v4f64_l {                                         // s is a compile-time constant wherever this is called
        if (s >= kRegLdsSlots) return tile[s - kRegLdsSlots];
        v4f64_l t;
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = Tl[(s * 4 + q) * 64];
        return t;
    };
    auto put = [&](int s, const v4f64_l& t) {
        if (s >= kRegLdsSlots) { tile[s - kRegLdsSlots] = t; return; }
#pragma unroll
        for (int q = 0; q < 4; ++q) Tl[(s * 4 + q) * 64] = t[q];
    };
#pragma unroll
    for (int s = 0; s < kRegSlots; ++s) {                                      // tiles <- sc * A, identity beyond n (a CU pulls ~50 GB/s: 8 us at n = 300)
        v4f64_l t = {0.0, 0.0, 0.0, 0.0};
        if (key[s] >= 0) {
            const int i = (key[s] & 31) * kNB + li;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = (key[s] >> 5) * kNB + lk + 4 * q;
                const double av = Ab[(unsigned)(min(i, n - 1) + min(col, n - 1) * n)];      // 32-bit element index (n <= 304)
                t[q] = (i < n && col < n) ? sc * av : ((i == col) ? 1.0 : 0.0);
            }
        }
        put(s, t);
    }
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };   // LDS visibility only: the output stores stay in flight
    // the diagonal block (wave-uniform calls): through LDS into the layout of diag16_factor; factor -> output, dsh
    auto factor_diag = [&](const v4f64_l& t) {                                 // -> dsh (L with zeros above the diagonal, the 4x4 inverses); the output copy follows in phase A
#pragma unroll
        for (int q = 0; q < 4; ++q) dtile[li][lk + 4 * q] = t[q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int ln = lane;                                                         // opaque: the routine's per-lane selection weights (20 doubles) are loop
        asm volatile("" : "+v"(ln));                                           // invariants otherwise, hoisted out of the panel loop and spilled
        const bool bad = diag16_factor(ln, [&](int i, int c) { return dtile[i][c]; }, [&](int, int, double) {}, dsh);
        if (bad && lane == 0) failed = 1;
    };
    auto update = [&](v4f64_l& t, int r, int c) {                              // t -= (strip rows of block r) x (strip rows of block c)'
        v4f64_l acc = {0.0, 0.0, 0.0, 0.0};
        const double* pa = P + (size_t)(r * kNB + li) * kRegPS + lk;
        const double* pb = P + (size_t)(c * kNB + li) * kRegPS + lk;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[kk * 4], pa[kk * 4], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] -= acc[q];
    };
#define MPOPIS_REG_CASES(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) M(20) M(21) M(22) M(23)
    RPROF(31, 1);
    lds_barrier();
    RPROF(31, 2);
    if (wv == 0) factor_diag(get(0));                                          // tile 0 = block (0, 0)
    lds_barrier();
    int td = 0;                                                                // tile number of diagonal block j
    for (int j = 0; j < npan; ++j) {
        if (failed) break;
        const int j0 = j * kNB;
        RPROF(j, 0);
        // ---- A: block column j below the diagonal block: tiles td + 1 .. td + npan - 1 - j, mine = those congruent wv mod 8 ----------
        {
            const PanelOps o = panel_solve_operands(lane, dsh);
            const int t1 = td + 1 + ((wv - (td + 1)) & 7), tlast = td + npan - 1 - j;
            const int cnt = (t1 <= tlast) ? ((tlast - t1) >> 3) + 1 : 0;       // 0 .. 3
            const int r1 = j + (t1 - td);
            auto emit = [&](const double (&a)[4], int r) {
                const int i = r * kNB + li;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = q * 4 + lk;
                    P[(size_t)i * kRegPS + cl] = a[q];
                    if (i < n && j0 + cl < n) Lb[(unsigned)(i + (j0 + cl) * n)] = a[q];
                }
            };
            for (int done = 0; done < cnt; done += 2) {                        // two tiles at a time: two independent chains of 7 MFMAs, interleaved
                v4f64_l x0 = {0.0, 0.0, 0.0, 0.0}, x1 = x0;                     // (an FP64 MFMA occupies the pipe for 64 cycles; a third chain costs registers)
                switch ((t1 >> 3) + done) {
#define MPOPIS_REG_FETCH2(S) case S: x0 = get(S); x1 = get((S) + 1 < kRegSlots ? (S) + 1 : (S)); break;
                    MPOPIS_REG_CASES(MPOPIS_REG_FETCH2)
#undef MPOPIS_REG_FETCH2
                }
                double a0[4] = {x0[0], x0[1], x0[2], x0[3]}, a1[4] = {x1[0], x1[1], x1[2], x1[3]};
                if (cnt - done >= 2) {
                    panel_solve_tile(o, a0); panel_solve_tile(o, a1);
                    emit(a0, r1 + 8 * done); emit(a1, r1 + 8 * done + 8);
                } else {
                    panel_solve_tile(o, a0);
                    emit(a0, r1 + 8 * done);
                }
            }
            RPROF(j, 1);
            if (wv == ((j + 4) & 7)) {                                         // the diagonal block itself (dsh.L: zeros above the diagonal included)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = lk + 4 * q;
                    if (j0 + li < n && j0 + cl < n) Lb[(unsigned)(j0 + li + (j0 + cl) * n)] = dsh.L[li][cl];
                }
            }
            // the zeros above this panel's diagonal block: in phase B, by the waves that wait for the factoring one there -- except behind the last panel
            if (j + 1 == npan)
                for (int cl = wv; cl < kNB; cl += kRegWaves)
                    if (j0 + cl < n) for (int i = lane; i < j0; i += 64) Lb[(unsigned)(i + (j0 + cl) * n)] = 0.0;
        }
        RPROF(j, 2);
        lds_barrier();
        RPROF(j, 3);
        if (j + 1 == npan) break;
        // ---- B: rank-16 update of everything to the right; the next diagonal block first, factored at once ---------------------
        const int td1 = td + npan - j;                                          // tile number of diagonal block j + 1
        if (wv == (td1 & 7)) {
            v4f64_l t = {0.0, 0.0, 0.0, 0.0};
            switch (td1 >> 3) {
#define MPOPIS_REG_FETCH1(S) case S: t = get(S); break;
                MPOPIS_REG_CASES(MPOPIS_REG_FETCH1)
#undef MPOPIS_REG_FETCH1
            }
            update(t, j + 1, j + 1);
            RPROF(j, 0);
            __builtin_amdgcn_s_setprio(3);
            factor_diag(t);
            __builtin_amdgcn_s_setprio(2);
        }
        RPROF(j, 4);
        {
            const int s0 = (td1 - wv + 8) >> 3;                                 // first slot with tile number > td1
            auto group = [&](int g) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int s = 4 * g + u;                                    // (compile-time after unrolling)
                    const bool act = s >= s0 && key[s] >= 0;
                    const int k = act ? key[s] : 0;                             // inactive: block (0, 0) of the strip = zeros, the tile keeps its bits
                    v4f64_l t = get(s); update(t, k & 31, k >> 5); put(s, t);
                }
            };
            switch (s0 >> 2) {
                case 0: group(0); [[fallthrough]];
                case 1: group(1); [[fallthrough]];
                case 2: group(2); [[fallthrough]];
                case 3: group(3); [[fallthrough]];
                case 4: group(4); [[fallthrough]];
                case 5: group(5);
            }
        }
        RPROF(j, 5);
        if (wv != (td1 & 7) && j0 > 0) {                                        // zeros above the diagonal block of columns j0 .. j0+15 (rows 0 .. j0-1): seven waves share
            for (int cl = (wv - (td1 & 7) - 1) & 7; cl < kNB; cl += kRegWaves - 1)      // the 16 columns while the eighth factors -- nobody waits for these stores
                for (int i = lane; i < j0; i += 64) Lb[(unsigned)(i + (j0 + cl) * n)] = 0.0;
        }
        lds_barrier();
        td = td1;
    }
#undef MPOPIS_REG_CASES
    if (failed && tid == 0) { if (status) status_raise(&status[b], MPOPIS_ERR_NOT_PD); if (active) active[b] = 0; }
}

// ---------------------------------------------------------------------------------------------
// Cooperative variant for matrices that do not fit one CU's LDS (cs = 300): G workgroups ("cluster") per matrix.  A single CU
// moves ~57 GB/s from L2 and has 1/256 of the chip's FP64 matrix rate -- k_potrf_global spends most of its 234 us on the
// trailing read-modify-write.  Here the 16-wide block columns (panels) are dealt round-robin to the G workgroups and live in
// their owner's LDS for the whole factorisation; the only traffic between CUs is each solved panel, once: its owner writes it to
// the output buffer (where it has to go anyway) with agent-scope write-through stores, drains them, and sets flag[b][j]; the other
// workgroups poll that flag (one lane, relaxed agent-scope loads), read the panel with agent-scope loads into an LDS strip and
// update their own panels from it on the matrix cores.  The owner of panel j+1 updates that panel first, factors, solves and
// publishes it, and only then updates the rest of its panels (look-ahead), so the critical path per panel is one hand-off + one
// 16x16 factorisation + one panel product.
// Flags are monotonic: a launch publishes 2*epoch (ok) / 2*epoch + 1 (not positive definite: everybody leaves), epoch = a per
// workspace launch counter, so they are never reset.  Every wait is bounded (2 s of the 100 MHz clock -> MPOPIS_ERR_HIP): a
// cluster whose workgroups are not all resident cannot hang the device.  No placement is assumed (any CU / XCD).
// ---------------------------------------------------------------------------------------------
#ifdef POTRF_PROF
__device__ unsigned long long g_prof[8 * 32 * 6];
void debug_read_prof(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(g_prof)); }
#define PROF_MARK(g_, j_, k_) do { if (b == 0 && tid == 0) g_prof[((g_) * 32 + (j_)) * 6 + (k_)] = wall_clock64(); } while (0)
#else
#define PROF_MARK(g_, j_, k_) do { } while (0)
#endif
constexpr int kCoopThreads = 1024, kCoopWaves = kCoopThreads / 64, kCoopMaxOwn = 16, kCoopPS = kNB + 1;
__device__ __forceinline__ int coop_ld(int rows) { return (rows & 31) ? rows : rows + 16; }   // column stride = 16 mod 32 doubles: the 4 k-groups of an MFMA operand read hit different bank halves
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }

__global__ void __launch_bounds__(kCoopThreads) k_potrf_coop(const double* __restrict__ A, size_t Astride, double* __restrict__ Lout, int n, int G, int S,
                                                             const double* scale, int* status, int* active,
                                                             unsigned long long* flags, unsigned long long epoch, int* redo, int* timeouts,
                                                             unsigned long long wait_ticks, int test_drop) {
    MPOPIS_HI_PRIO();
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ DiagScratch dsh;
    __shared__ int sh_fail, sh_wait, p_off[kCoopMaxOwn], p_ld[kCoopMaxOwn];
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    if (active && !active[b]) return;
    if (test_drop && g == G - 1) return;                // test hook: a partner that never runs
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int npan = (n + kNB - 1) / kNB, npad = npan * kNB;
    // ownership: blocks of S consecutive panels dealt round-robin; own panel q <-> panel (q / S) S G + g S + q % S
    auto own_panel = [&](int q) { return (q / S) * S * G + g * S + (q % S); };
    auto owner = [&](int j) { return (j / S) % G; };
    auto qidx = [&](int j) { return (j / (S * G)) * S + (j % S); };
    int nown = 0;
    while (nown < kCoopMaxOwn && own_panel(nown) < npan) ++nown;
    const double* Ab = A + (size_t)b * Astride;
    double* Lb = Lout + (size_t)b * n * n;
    unsigned long long* fl = flags + (size_t)b * npan;
    const double sc = scale ? scale[b] : 1.0;
    const unsigned long long ok_val = 2 * epoch;
    if (tid == 0) {
        int off = 0;
        for (int q = 0; q < nown; ++q) { const int rows = npad - own_panel(q) * kNB; p_off[q] = off; p_ld[q] = coop_ld(rows); off += p_ld[q] * kNB; }
        sh_fail = 0;
    }
    __syncthreads();
    double* P = smem + p_off[nown - 1] + p_ld[nown - 1] * kNB;      // received panel strip [npad][17]
    // ---- own panels <- lower(sc * A), identity beyond n -----------------------------------------------------------------------
    for (int q = 0; q < nown; ++q) {
        const int c0 = own_panel(q) * kNB, rows = npad - c0, ld = p_ld[q], cnt = rows * kNB;
        double* W = smem + p_off[q];
        for (int e0 = tid; e0 < cnt; e0 += kCoopThreads * 4) {
            double av[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(e0 + u * kCoopThreads, cnt - 1), i = c0 + e % rows, col = c0 + e / rows;
                av[u] = Ab[(size_t)min(i, n - 1) + (size_t)min(col, n - 1) * n];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * kCoopThreads;
                if (e < cnt) { const int il = e % rows, cl = e / rows, i = c0 + il, col = c0 + cl; W[il + cl * ld] = (i < n && col < n) ? sc * av[u] : ((i == col) ? 1.0 : 0.0); }
            }
        }
    }
    __syncthreads();
    // operand (row r, k) of panel j: from the owner's own storage or from the received strip
    auto tile_update = [&](double* Wc, int ldc, int c, int ta, const double* src, int src_ld, int src_row0, bool strip) {
        // Wc: panel c (local row 0 = global row 16c); tile rows 16 ta ..; src(r, k) = strip ? src[r * 17 + k] : src[(r - src_row0) + k * src_ld]
        const int r0 = ta * kNB, c0 = c * kNB;
        v4f64_l acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = kk * 4 + lk;
            const double av = strip ? src[(size_t)(r0 + li) * kCoopPS + k] : src[(r0 + li - src_row0) + k * src_ld];
            const double bv = strip ? src[(size_t)(c0 + li) * kCoopPS + k] : src[(c0 + li - src_row0) + k * src_ld];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc, 0, 0, 0);     // acc[r] <-> (row r0 + li, column c0 + lk + 4r)
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Wc[(r0 - c0 + li) + (lk + 4 * r) * ldc] -= acc[r];
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };   // LDS visibility only: does not wait for the global stores in flight
    int pending = -1;                                   // own panel whose stores are issued but whose flag is not yet set
    auto publish = [&](int c) {                         // every wave: its stores have left; then one flag store
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&fl[c], ok_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // factor the diagonal block of own panel q (= panel c), solve the panel, write it out; the flag follows at once when the next panel
    // belongs to another workgroup, otherwise after this workgroup's next panel update (its stores drain meanwhile)
    auto factor_solve_publish = [&](int q, int c) -> bool {
        double* W = smem + p_off[q];
        const int ld = p_ld[q], c0 = c * kNB, h = npan - c;
        if (pending >= 0) {                             // the previous panel's stores were issued a whole panel update ago: announce it now, before
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the diagonal block, so that the others work on it while this workgroup factors
            lds_barrier();
            if (tid == 0) __hip_atomic_store(&fl[pending], ok_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (wv == 0) {
            int ln = lane;                              // opaque (see k_potrf_reg)
            asm volatile("" : "+v"(ln));
            const bool bad = diag16_factor(ln, [&](int i, int cc) { return W[i + cc * ld]; }, [&](int i, int cc, double v) { W[i + cc * ld] = v; }, dsh);
            if (bad && lane == 0) sh_fail = 1;
        }
        __syncthreads();
        PROF_MARK(g, c > 0 ? c - 1 : 31, 5);
        if (sh_fail) {
            if (tid == 0) {
                if (status) status_raise(&status[b], MPOPIS_ERR_NOT_PD);
                if (active) active[b] = 0;
                for (int jj = c; jj < npan; ++jj) __hip_atomic_store(&fl[jj], ok_val + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return false;
        }
        if (1 + wv < h) {
            const PanelOps o = panel_solve_operands(lane, dsh);
            for (int t = 1 + wv; t < h; t += kCoopWaves) {        // L21 tile = A21 tile * L11^-T
                double a[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = W[(t * kNB + li) + (q * 4 + lk) * ld];
                panel_solve_tile(o, a);
                const int i = c0 + t * kNB + li;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = q * 4 + lk;
                    W[(t * kNB + li) + cl * ld] = a[q];
                    if (i < n && c0 + cl < n) st_agent(&Lb[(size_t)i + (size_t)(c0 + cl) * n], a[q]);
                }
            }
        }
        pending = -1;
        if (c + 1 < npan && owner(c + 1) == g) { pending = c; lds_barrier(); }   // solved panel visible in LDS for this workgroup's next update
        else publish(c);
        // rows 0 .. 16c+15 of these columns: zeros above the diagonal, the factored diagonal block (nobody in the cluster reads them)
        for (int e = tid; e < (c0 + kNB) * kNB; e += kCoopThreads) {
            const int i = e % (c0 + kNB), cl = e / (c0 + kNB);
            if (i < n && c0 + cl < n) Lb[(size_t)i + (size_t)(c0 + cl) * n] = (i >= c0 + cl) ? W[(i - c0) + cl * ld] : 0.0;
        }
        return true;
    };
    // wait for panel j and read its rows below the diagonal block into the strip
    auto receive = [&](int j) -> bool {
        if (tid == 0) {
            const unsigned long long t0 = wall_clock64();
            unsigned long long v;
            int st = 0;
            while ((v = __hip_atomic_load(&fl[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < ok_val) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > wait_ticks) { st = 2; break; }
            }
            if (!st && (v & 1)) st = 1;
            if (st == 2) { redo[b] = 1; atomicAdd(timeouts, 1); }      // a partner is not running: the slot is recomputed by the one-workgroup kernel queued behind this launch
            sh_wait = st;
        }
        __syncthreads();
        if (sh_wait) return false;
        const int r1 = (j + 1) * kNB, rows = npad - r1, cnt = rows * kNB, j0 = j * kNB;
        for (int e0 = tid; e0 < cnt; e0 += kCoopThreads * 5) {      // one batch of loads in flight for cs = 300 (4608 entries)
            double av[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int e = min(e0 + u * kCoopThreads, cnt - 1), r = r1 + e % rows, k = e / rows;
                av[u] = ld_agent(&Lb[(size_t)min(r, n - 1) + (size_t)min(j0 + k, n - 1) * n]);
            }
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int e = e0 + u * kCoopThreads;
                if (e < cnt) { const int r = r1 + e % rows, k = e / rows; P[(size_t)r * kCoopPS + k] = (r < n && j0 + k < n) ? av[u] : 0.0; }
            }
        }
        __syncthreads();
        return true;
    };
    if (g == 0 && !factor_solve_publish(0, 0)) return;
    for (int j = 0; j + 1 < npan; ++j) {
        const bool mine = owner(j) == g;
        PROF_MARK(g, j, 0);
        if (!mine && !receive(j)) return;
        PROF_MARK(g, j, 1);
        const double* src = mine ? smem + p_off[qidx(j)] : P;
        const int src_ld = mine ? p_ld[qidx(j)] : 0, src_row0 = j * kNB;
        const bool nxt = owner(j + 1) == g;
        if (nxt) {
            const int q = qidx(j + 1), c = j + 1;
            for (int ta = c + wv; ta < npan; ta += kCoopWaves) tile_update(smem + p_off[q], p_ld[q], c, ta, src, src_ld, src_row0, !mine);
            lds_barrier();
            PROF_MARK(g, j, 2);
            if (!factor_solve_publish(q, c)) return;
            PROF_MARK(g, j, 3);
        }
        // the rest of the owned panels (c > j + 1): tiles dealt to the waves across panels
        int idx = wv;
        for (int q = 0; q < nown; ++q) {
            const int c = own_panel(q);
            if (c <= j + 1) continue;
            const int h = npan - c;
            for (; idx < h; idx += kCoopWaves) tile_update(smem + p_off[q], p_ld[q], c, c + idx, src, src_ld, src_row0, !mine);
            idx -= h;
        }
        lds_barrier();
        PROF_MARK(g, j, 4);
    }
    // (pending is never left set: the last panel has no successor, so it is published at once)
}

// k_potrf_reg needs nearly a whole CU's LDS (strip + tile slots + ~6 KB static: 158.4 KB at n = 304 of the 160 KB a workgroup may have on gfx950); asked of the
// device once per ordinal, so that a part with less falls back to the cluster / one-workgroup kernels instead of failing to launch
static bool potrf_reg_fits_device() {
    static std::atomic<int> ok[64];                             // 0 unknown, 1 fits, 2 does not
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int v = ok[dev & 63].load(std::memory_order_relaxed);
    if (v == 0) {
        int lds = 0;
        const bool got = hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess;
        v = (got && (size_t)lds >= potrf_reg_lds_bytes(kRegMaxPan * kNB) + sizeof(DiagScratch) + sizeof(double) * kNB * (kNB + 1) + 64) ? 1 : 2;      // + the kernel's static LDS
        ok[dev & 63].store(v, std::memory_order_relaxed);
    }
    return v == 1;
}

size_t potrf_coop_flag_words(int B, int n) { return (size_t)B * ((n + kNB - 1) / kNB); }

// coop_flags / coop_epoch: per-handle workspace (potrf_coop_flag_words, zero-initialised once) and launch counter; null -> never cooperative
int coop_max_workgroups() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int v = cus[dev & 63].load(std::memory_order_relaxed);
    if (v == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 2;
        v = std::max(1, n);
        cus[dev & 63].store(v, std::memory_order_relaxed);
    }
    static const int env_cap = [] { const char* e = getenv("MPOPIS_COOP_MAX_WG"); return e ? atoi(e) : 0; }();      // tests: pretend a smaller / larger device
    return env_cap > 0 ? env_cap : v;
}
unsigned long long coop_wait_ticks() {
    static const unsigned long long t = [] { const char* e = getenv("MPOPIS_COOP_WAIT_US"); const long long us = e ? atoll(e) : 1000000ll; return (unsigned long long)std::max(1ll, us) * 100ull; }();
    return t;
}
int coop_test_drop() {
    static const int v = [] { const char* e = getenv("MPOPIS_COOP_TEST_DROP"); return e ? atoi(e) : 0; }();
    return v;
}

size_t potrf_panel_doubles(int n) { return n <= kPanelRows ? (size_t)((n + 15) / 16) * 16 * kPanelRows : 0; }

void launch_potrf(const double* A, size_t Astride, double* L, int B, int n, const double* scale, int* status, int* active, hipStream_t s,
                  const CoopCtx& coop, double* panel, size_t pstride) {
    const int npad = (n + kNB - 1) / kNB * kNB, npan = npad / kNB;
    const size_t bytes = (size_t)npad * npad * sizeof(double);
    if (bytes <= 150 * 1024) {
        static std::atomic<unsigned long long> seen{0};
        ensure_dyn_lds((const void*)k_potrf_lds, 150 * 1024, seen);
        hipLaunchKernelGGL(k_potrf_lds, dim3(B), dim3(512), bytes, s, A, Astride, L, n, npad, scale, status, active, n <= kPanelRows ? panel : nullptr, pstride);
        return;
    }
    static const int env_reg = [] { const char* e = getenv("MPOPIS_POTRF_REG"); return e ? atoi(e) : 1; }();      // tests / A-B: 0 = the cluster and global kernels only
    // n = 241 .. 304: the register-resident kernel (n = 300: 133 us at one slot, 140 us at 64 -- clusters 160, one-workgroup global 225; below 16 panels
    // its fixed costs -- 8 us of loads through one CU, ~5 us per panel whatever its height -- lose to the clusters: n = 240: 123 vs 120, n = 144: 84 vs 68)
    if (env_reg && npan >= kRegMinPan && npan <= kRegMaxPan && potrf_reg_fits_device()) {
        static std::atomic<unsigned long long> seenr{0};
        const size_t rbytes = potrf_reg_lds_bytes(n);
        ensure_dyn_lds((const void*)k_potrf_reg, (int)potrf_reg_lds_bytes(kRegMaxPan * kNB), seenr);
        hipLaunchKernelGGL(k_potrf_reg, dim3(B), dim3(64 * kRegWaves), rbytes, s, A, Astride, L, n, scale, status, active);
        return;
    }
    static const int env_G = [] { const char* e = getenv("MPOPIS_POTRF_G"); return e ? atoi(e) : -1; }();
    static const int env_S = [] { const char* e = getenv("MPOPIS_POTRF_S"); return e ? atoi(e) : -1; }();
    // panels per ownership block.  S = 2 halves the hand-offs on the critical path but measured no faster (162 vs 158 us at n = 300): the
    // per-panel chain is barriers + diagonal block + solve + store drain (~6 of 8.2 us), not the hand-off itself
    const int S = env_S > 0 ? env_S : 1;
    int G = env_G >= 0 ? env_G : 6;
    const int nblk = (npan + S - 1) / S;
    if (G > nblk) G = nblk;
    size_t coop_lds = 0;
    int nown0 = 0;
    if (G >= 2 && coop.usable()) {
        size_t own = 0;                                           // workgroup 0 owns the tallest panels
        for (int q = 0;; ++q) {
            const int c = (q / S) * S * G + (q % S);
            if (c >= npan) break;
            const int rows = npad - c * kNB;
            own += (size_t)((rows & 31) ? rows : rows + 16) * kNB; ++nown0;
        }
        coop_lds = (own + (size_t)npad * kCoopPS) * sizeof(double);
    }
    // clusters should be co-resident: one workgroup per CU (LDS), the grid at most the device's CU count.  Not assumed: kernels of other
    // streams / processes may hold CUs, so a cluster that gives up (bounded waits) marks its slot in coop.redo and the one-workgroup kernel
    // queued right behind recomputes exactly those slots (A is not modified).
    const size_t strip = (size_t)n * (kNB + 1) * sizeof(double);                             // panel strip
    static std::atomic<unsigned long long> seen2{0};
    ensure_dyn_lds((const void*)k_potrf_global, 150 * 1024, seen2);
    if (coop_lds && coop_lds <= 150 * 1024 && nown0 <= kCoopMaxOwn && B * G * coop.share <= coop_max_workgroups()) {
        static std::atomic<unsigned long long> seen3{0};
        ensure_dyn_lds((const void*)k_potrf_coop, 150 * 1024, seen3);
        const unsigned long long epoch = ++*coop.epoch;
        hipLaunchKernelGGL(k_potrf_coop, dim3(B * G), dim3(kCoopThreads), coop_lds, s, A, Astride, L, n, G, S, scale, status, active, coop.flags, epoch,
                           coop.redo, coop.timeouts, coop_wait_ticks(), coop_test_drop());
        hipLaunchKernelGGL(k_potrf_global, dim3(B), dim3(1024), strip, s, A, Astride, L, n, scale, status, active, coop.redo);
        return;
    }
    hipLaunchKernelGGL(k_potrf_global, dim3(B), dim3(1024), strip, s, A, Astride, L, n, scale, status, active, (int*)nullptr);
}

// g = Σ⁻¹ (γ U_orig) through the Cholesky factor (Σ symmetric => row vector γ U_orig' Σ⁻¹ = g').
// Slow path: only taken when α != 1 (γ != 0); no BASELINE config uses it.
__global__ void __launch_bounds__(256) k_chol_solve_gvec(const double* __restrict__ L, size_t Lstride, const double* __restrict__ Uorig,
                                                         double gamma, double* __restrict__ g, int n, const int* active, const double* inv_scale2) {
    MPOPIS_HI_PRIO();
    extern __shared__ __attribute__((aligned(16))) double y[];
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    const double* Lb = L + (size_t)b * Lstride;
    for (int i = threadIdx.x; i < n; i += 256) y[i] = gamma * Uorig[(size_t)b * n + i];
    __syncthreads();
    for (int j = 0; j < n; ++j) {                     // forward: L y = γU
        if (threadIdx.x == 0) y[j] = y[j] / Lb[(size_t)j + (size_t)j * n];
        __syncthreads();
        const double yj = y[j];
        for (int i = j + 1 + threadIdx.x; i < n; i += 256) y[i] = fma(-Lb[(size_t)i + (size_t)j * n], yj, y[i]);
        __syncthreads();
    }
    for (int j = n - 1; j >= 0; --j) {                // backward: L' g = y
        if (threadIdx.x == 0) y[j] = y[j] / Lb[(size_t)j + (size_t)j * n];
        __syncthreads();
        const double yj = y[j];
        for (int i = threadIdx.x; i < j; i += 256) y[i] = fma(-Lb[(size_t)j + (size_t)i * n], yj, y[i]);
        __syncthreads();
    }
    const double isc = inv_scale2 ? 1.0 / inv_scale2[b] : 1.0;
    for (int i = threadIdx.x; i < n; i += 256) g[(size_t)b * n + i] = inv_scale2 ? y[i] * isc : y[i];
}
void launch_chol_solve_gvec(const double* L, size_t Lstride, const double* Uorig, double gamma, double* g, int B, int n, const int* active, hipStream_t s,
                            const double* inv_scale2) {
    hipLaunchKernelGGL(k_chol_solve_gvec, dim3(B), dim3(256), n * sizeof(double), s, L, Lstride, Uorig, gamma, g, n, active, inv_scale2);
}

// Level-1 entry: caller supplies Σ_inv; g[j] = Σ_i (γ U_orig[i]) Σ_inv[i][j]   (:272)
__global__ void __launch_bounds__(256) k_gvec_from_inv(const double* __restrict__ Sinv, const double* __restrict__ Uorig, double gamma,
                                                       double* __restrict__ g, int n) {
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double v = 0.0;
    for (int i = 0; i < n; ++i) v = fma(gamma * Uorig[(size_t)b * n + i], Sinv[(size_t)i + (size_t)j * n], v);
    g[(size_t)b * n + j] = v;
}
void launch_gvec_from_inv(const double* Sinv, const double* Uorig, double gamma, double* g, int B, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_gvec_from_inv, dim3((n + 255) / 256, B), dim3(256), 0, s, Sinv, Uorig, gamma, g, n);
}

// mean over gathered columns: mu[r] = (1/m) Σ_j X[r][idx[j]] ; optionally weighted by cw[j] (CMA δw, no division)
__global__ void __launch_bounds__(256) k_gather_mean(const double* __restrict__ X, const int32_t* __restrict__ idx, const double* __restrict__ cw,
                                                     double* __restrict__ mu, size_t mu_stride, int cs, int K, int m, int divide, const int* active) {
    MPOPIS_HI_PRIO();
    const int b = blockIdx.y, r = blockIdx.x;
    if (active && !active[b]) return;
    __shared__ double sh[4];
    const double* x = X + ((size_t)b * cs + r) * K;
    const int32_t* ib = idx + (size_t)b * K;          // idx arrays are K long per slot (order / resample)
    double acc = 0.0;
    // index -> value is two dependent global round trips per element: issue them 8 elements at a time (m = K = 4096 for :pmcmppi is 16
    // elements per thread -- 32 serial round trips as a plain loop); same summation order as the plain loop
    for (int j0 = threadIdx.x; j0 < m; j0 += 256 * 8) {
        int ii[8]; double xv[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int j = min(j0 + 256 * u, m - 1); ii[u] = ib[j]; wv[u] = cw ? cw[j] : 1.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = x[ii[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (j0 + 256 * u < m) acc = cw ? fma(wv[u], xv[u], acc) : acc + xv[u];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double t = sh[0] + sh[1] + sh[2] + sh[3]; mu[(size_t)b * mu_stride + r] = divide ? t / m : t; }
}
// Xout[b][r][k] = X[b][r][idx[b][k]]: E′ = E[:, idx] (:806) as a contiguous matrix (coalesced index reads and writes; the random reads stay inside
// one 8K-byte row), so that the moments of the resampled columns run through the un-gathered fast path of the scatter kernel
// shift (nullable, [B][cs]): the first resampled column E′[:, 1] is subtracted from every column and stored there.  Mean and covariance of the
// shifted data are those of E′ up to adding the shift back to the mean (the caller does), and the one-pass moments (Σ x x' - n μ μ')/(n - 1) then
// cancel against a small mean even when the resampled set collapses onto a few distinct columns far from the origin -- the reference's
// mean_and_cov (:807) centres before it squares.
__global__ void __launch_bounds__(256) k_gather_cols(const double* __restrict__ X, const int32_t* __restrict__ idx, double* __restrict__ Xout, int cs, int K,
                                                     const int* active, double* __restrict__ shift) {
    const int b = blockIdx.z, r = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if ((active && !active[b]) || k >= K) return;
    const double* xr = X + ((size_t)b * cs + r) * K;
    const double x0 = shift ? xr[idx[(size_t)b * K]] : 0.0;
    if (shift && k == 0) shift[(size_t)b * cs + r] = x0;
    Xout[((size_t)b * cs + r) * K + k] = xr[idx[(size_t)b * K + k]] - x0;
}
void launch_gather_cols(const double* X, const int32_t* idx, double* Xout, int B, int cs, int K, const int* active, hipStream_t s, double* shift) {
    hipLaunchKernelGGL(k_gather_cols, dim3((K + 255) / 256, cs, B), dim3(256), 0, s, X, idx, Xout, cs, K, active, shift);
}
void launch_gather_mean(const double* X, const int32_t* idx, const double* cw, double* mu, int B, int cs, int K, int m, int divide,
                        const int* active, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_mean, dim3(cs, B), dim3(256), 0, s, X, idx, cw, mu, (size_t)cs, cs, K, m, divide, active);
}
void launch_gather_mean_strided(const double* X, const int32_t* idx, const double* cw, double* out, size_t out_stride, int B, int cs, int K, int m,
                                const int* active, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_mean, dim3(cs, B), dim3(256), 0, s, X, idx, cw, out, out_stride, cs, K, m, 0, active);
}

}  // namespace mpopis
