// kernels_select.hip -- order statistics and multinomial resampling on device:
//   order = sortperm(trajectory_cost)                       src/mppi_mpopi_policies.jl:455,563
//   early break: maximum(abs.(diff(elite_traj_cost))) < 10e-3   :458-461,:566-569
//   Categorical(ws) -> AliasTable (StatsBase.make_alias_table!) + rand(rng, ., K)   :804-805
// One workgroup per trial slot while K <= 8192 (sort) / K <= 7168 (alias) lets the slot live entirely in LDS; chip-wide / global-workspace forms beyond.
#include "engine.h"
#include <type_traits>

namespace mpopis {

// CE/CMA early break (:458-461 / :566-569), fused into the sort kernels: if max_j |c[order[j+1]] - c[order[j]]| < 10e-3 over the elite
// set, the slot leaves the AIS loop (active[b] = 0) -- nothing of this iteration's update is applied.  skey: the sorted costs in LDS
// (visible to the whole workgroup); m_elite <= 0: no check.
__device__ __forceinline__ void elite_break_tail(const double* skey, int m_elite, int* active, int b) {
    __shared__ double eb_red[16];
    if (m_elite < 2 || !active) return;
    double mx = -INFINITY;
    for (int j = threadIdx.x; j + 1 < m_elite; j += blockDim.x) mx = fmax(mx, fabs(skey[j + 1] - skey[j]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) eb_red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nw; ++w) mx = fmax(mx, eb_red[w]);
        if (mx < 10e-3) active[b] = 0;
    }
}

// Bitonic sort of (cost, index) pairs under the total order (cost asc, index asc) == Julia's stable sortperm.
// n = next pow2 >= max(K, 64), padded with (+inf, index).  Thread t keeps the EPT consecutive entries t EPT .. t EPT + EPT-1 in
// registers: compare-exchange steps with stride < EPT stay inside the thread, strides < 64 EPT are wave shuffles (no barrier), only
// the strides that cross waves (>= 64 EPT: 10 of the 78 steps at n = 4096) go through LDS with a barrier.  (The all-LDS version
// with one barrier per step took 57 us at K = 4096.)
template <int EPT>
__global__ void __launch_bounds__(1024) k_sortperm(const double* __restrict__ cost, int32_t* __restrict__ order, int K, int n,
                                                   int m_elite, int* active) {
    MPOPIS_HI_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* key_l = reinterpret_cast<double*>(smem);
    int32_t* idx_l = reinterpret_cast<int32_t*>(smem + (size_t)n * sizeof(double));
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    const int base = threadIdx.x * EPT;
    double k[EPT];
    int id[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) { const int e = base + r; k[r] = (e < K) ? cost[(size_t)b * K + e] : INFINITY; id[r] = e; }
    // a <- min(a, b) if keep_min else max(a, b), under (key, index) order (indices are distinct: never equal)
    auto cmpx = [](double& ka, int& ia, double kb, int ib, bool keep_min) {
        const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
        if (a_gt_b == keep_min) { ka = kb; ia = ib; }
    };
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride < EPT) {                                             // inside the thread (compile-time register pairs)
#pragma unroll
                for (int S = 1; S < EPT; S <<= 1) {
                    if (stride == S) {
#pragma unroll
                        for (int r = 0; r < EPT; ++r) {
                            if ((r & S) == 0) {
                                const bool up = (((base + r) & size) == 0);
                                const bool a_gt_b = (k[r] > k[r + S]) || (k[r] == k[r + S] && id[r] > id[r + S]);
                                if (a_gt_b == up) { const double tk = k[r]; k[r] = k[r + S]; k[r + S] = tk; const int ti = id[r]; id[r] = id[r + S]; id[r + S] = ti; }
                            }
                        }
                    }
                }
            } else if (stride < EPT * 64) {                                 // inside the wave
                const int lm = stride / EPT;
#pragma unroll
                for (int r = 0; r < EPT; ++r) {
                    const double kb = __shfl_xor(k[r], lm, 64);
                    const int ib = __shfl_xor(id[r], lm, 64);
                    const int e = base + r;
                    cmpx(k[r], id[r], kb, ib, ((e & stride) == 0) == ((e & size) == 0));
                }
            } else {                                                        // across waves
#pragma unroll
                for (int r = 0; r < EPT; ++r) { key_l[base + r] = k[r]; idx_l[base + r] = id[r]; }
                __syncthreads();
                const int pb = base ^ stride;
#pragma unroll
                for (int r = 0; r < EPT; ++r) {
                    const int e = base + r;
                    cmpx(k[r], id[r], key_l[pb + r], idx_l[pb + r], ((e & stride) == 0) == ((e & size) == 0));
                }
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int r = 0; r < EPT; ++r) { if (base + r < K) order[(size_t)b * K + base + r] = id[r]; key_l[base + r] = k[r]; }
    __syncthreads();
    elite_break_tail(key_l, m_elite, active, b);
}

// All-LDS bitonic network, one barrier per compare-exchange step: fastest for the mid sizes (n = 512, 1024), where the wave
// shuffles of the register version cost more than its saved barriers.
__global__ void __launch_bounds__(1024) k_sortperm_lds(const double* __restrict__ cost, int32_t* __restrict__ order, int K, int n,
                                                       int m_elite, int* active) {
    MPOPIS_HI_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* key = reinterpret_cast<double*>(smem);
    int32_t* idx = reinterpret_cast<int32_t*>(smem + (size_t)n * sizeof(double));
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { key[i] = (i < K) ? cost[(size_t)b * K + i] : INFINITY; idx[i] = i; }
    __syncthreads();
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < n / 2; t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1));          // index with the `stride` bit clear
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const double ka = key[lo], kb = key[hi];
                const int ia = idx[lo], ib = idx[hi];
                const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
                if (a_gt_b == up) { key[lo] = kb; key[hi] = ka; idx[lo] = ib; idx[hi] = ia; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < K; i += blockDim.x) order[(size_t)b * K + i] = idx[i];
    elite_break_tail(key, m_elite, active, b);
}

// K <= 256 (C3: K = 150): rank sort.  Thread i counts the entries that precede entry i in the (cost, index) order -- K broadcast LDS
// reads and compares, no barrier after the staging -- and writes order[rank] = i.  (C3: sort + early break 13.7 -> 9.2 us per iteration with
// the 36-step bitonic network gone: a lone wave pays ~8 cycles per instruction whatever it does, so the instruction count is the cost.)
__global__ void __launch_bounds__(256) k_sortperm_rank(const double* __restrict__ cost, int32_t* __restrict__ order, int K, int m_elite, int* active) {
    MPOPIS_HI_PRIO();
    __shared__ __attribute__((aligned(16))) double c[256], sc[256];
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    const int i = threadIdx.x;
    const double cr = (i < K) ? cost[(size_t)b * K + i] : INFINITY;
    // the order is (cost, index) with NaN placed like +inf (Julia's isless sorts NaN last): compared as raw doubles every NaN entry would get
    // rank 0, ranks would repeat and order[] would keep stale indices (the slot is already flagged MPOPIS_ERR_ACTION by the rollout, but the
    // CE / CMA update behind this sort still gathers through order[] before the host sees the error)
    const double ci = (cr != cr) ? INFINITY : cr;
    c[i] = ci;
    __syncthreads();
    int rank = 0;
    const int K4 = (i < K) ? (K & ~3) : 0, Kr = (i < K) ? K : 0;
    sc[i] = INFINITY;
    for (int j = 0; j < K4; j += 4) {
        const double c0 = c[j], c1 = c[j + 1], c2 = c[j + 2], c3 = c[j + 3];
        rank += ((c0 < ci) || (c0 == ci && j < i)) + ((c1 < ci) || (c1 == ci && j + 1 < i)) + ((c2 < ci) || (c2 == ci && j + 2 < i)) +
                ((c3 < ci) || (c3 == ci && j + 3 < i));
    }
    for (int j = K4; j < Kr; ++j) rank += ((c[j] < ci) || (c[j] == ci && j < i));
    __syncthreads();
    if (i < K) { order[(size_t)b * K + rank] = i; sc[rank] = cr; }
    __syncthreads();
    elite_break_tail(sc, m_elite, active, b);
}


// ---- elite early break (:458-461 / :566-569) by the LAST workgroup of the slot (chip-wide rank sorts) ----------------------------------------
// Hand-off of the elite keys: the keys are agent-scope stores (written through to where every XCD's agent-scope loads find them); a workgroup
// draws its ticket once its own stores are acknowledged (s_waitcnt vmcnt(0): on gfx9 -- gfx950 included -- stores count on vmcnt, there is no vscnt).
// The ticket itself is an ACQUIRE at agent scope (a buffer invalidate on this part, no L2 write-back), so that under the HIP memory model the last
// workgroup's reads of skey happen-after every other workgroup's ticket; the release side stays the explicit store-acknowledge wait instead of a
// release fence (which writes the whole L2 back: 23 -> 21 us per sort at one C4 slot, and much more beside dirty rollout data).
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "the vmcnt-acknowledged hand-off below is written for the gfx9 family (stores counted on vmcnt, agent-scope stores written through)"
#endif
__device__ __forceinline__ void elite_break_last_workgroup(double* __restrict__ skey, int* __restrict__ done, int m_elite, int* active, int b, int K,
                                                           int* sh_last, double* eb) {
    const int tid = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) *sh_last = (__hip_atomic_fetch_add(&done[b], 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1);
    __syncthreads();
    if (!*sh_last) return;
    double mx = -INFINITY;
    for (int q = tid; q + 1 < m_elite; q += 256) {
        const double a0 = __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)&skey[(size_t)b * K + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const double a1 = __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)&skey[(size_t)b * K + q + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        mx = fmax(mx, fabs(a1 - a0));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) eb[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
        mx = fmax(fmax(eb[0], eb[1]), fmax(eb[2], eb[3]));
        if (mx < 10e-3) active[b] = 0;
        done[b] = 0;                                                          // ready for the next launch (stream order)
    }
}

// Large K at a few slots (C4: K = 4096, 1-16 trials): the bitonic network above is one workgroup per slot and ~49 us of pure latency (78 dependent
// compare-exchange steps) on a chip that is otherwise idle.  Rank sort across the whole chip instead: K^2 independent comparisons.  A workgroup owns
// kRmE = 16 entries; thread (el, js) counts how many entries of the js-th sixteenth of the slot precede entry el in the (cost, index) order (all K costs
// in LDS: broadcast reads), the 16 partial counts are added, and order[rank] = entry -- the same permutation as the stable sort, by construction.
// The elite early break needs the sorted costs: every workgroup publishes skey[rank] for rank < m_elite (agent-scope stores), and the LAST workgroup
// of the slot to finish (a per-slot arrival counter) evaluates max |diff| -- one launch, no host round trip.  Work grows with B K^2: used for one or two slots.
constexpr int kRmE = 16;
__global__ void __launch_bounds__(256) k_sortperm_rank_multi(const double* __restrict__ cost, int32_t* __restrict__ order, int K, int m_elite, int* active,
                                                             double* __restrict__ skey, int* __restrict__ done) {
    MPOPIS_HI_PRIO();
    extern __shared__ __attribute__((aligned(16))) double c_l[];            // [K]
    __shared__ int sh_rank[kRmE];
    __shared__ int sh_last;
    __shared__ double eb[4];
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const int tid = threadIdx.x, el = tid & (kRmE - 1), js = tid >> 4;
    for (int j = tid; j < K; j += 256) { const double v = cost[(size_t)b * K + j]; c_l[j] = (v != v) ? INFINITY : v; }      // NaN ranks like +inf (see k_sortperm_rank)
    if (tid < kRmE) sh_rank[tid] = 0;
    __syncthreads();
    const int e = blockIdx.x * kRmE + el;
    const double ce = (e < K) ? c_l[e] : INFINITY;
    const int JR = ((K + 15) / 16 + 3) & ~3, j0 = js * JR, j1 = min(K, j0 + JR);
    int cnt = 0;
    int j = j0;
    for (; j + 4 <= j1; j += 4) {
        const double c0 = c_l[j], c1 = c_l[j + 1], c2 = c_l[j + 2], c3 = c_l[j + 3];
        cnt += ((c0 < ce) || (c0 == ce && j < e)) + ((c1 < ce) || (c1 == ce && j + 1 < e)) + ((c2 < ce) || (c2 == ce && j + 2 < e)) +
               ((c3 < ce) || (c3 == ce && j + 3 < e));
    }
    for (; j < j1; ++j) cnt += ((c_l[j] < ce) || (c_l[j] == ce && j < e));
    cnt += __shfl_xor(cnt, 16, 64);                                           // the four js values of this wave
    cnt += __shfl_xor(cnt, 32, 64);
    if ((tid & 63) < kRmE) atomicAdd(&sh_rank[el], cnt);                      // ... and the four waves
    __syncthreads();
    if (tid < kRmE && e < K) {
        const int rank = sh_rank[el];
        order[(size_t)b * K + rank] = e;
        if (skey && rank < m_elite) __hip_atomic_store((unsigned long long*)&skey[(size_t)b * K + rank], (unsigned long long)__double_as_longlong(ce), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (m_elite < 2 || !active || !skey || !done) return;
    elite_break_last_workgroup(skey, done, m_elite, active, b, K, &sh_last, eb);
}

// Any K (the reference's sortperm has no size limit, src/mppi_mpopi_policies.jl:455,563): the chip-wide rank sort with the slot's costs streamed through
// LDS in chunks of kRbChunk instead of held whole (k_sortperm_rank_multi keeps all K costs of the slot in LDS: K <= 12288; the bitonic kernels K <= 8192).
// Same counting, same (cost, index) order, same hand-off of the elite keys to the slot's last workgroup.  Work is B K^2 comparisons: the path for
// K > 8192 only (K = 16384: 2.7e8 per slot).
constexpr int kRbChunk = 4096;
__global__ void __launch_bounds__(256) k_sortperm_rank_big(const double* __restrict__ cost, int32_t* __restrict__ order, int K, int m_elite, int* active,
                                                           double* __restrict__ skey, int* __restrict__ done) {
    MPOPIS_HI_PRIO();
    __shared__ __attribute__((aligned(16))) double c_l[kRbChunk];
    __shared__ int sh_rank[kRmE];
    __shared__ int sh_last;
    __shared__ double eb[4];
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const int tid = threadIdx.x, el = tid & (kRmE - 1), js = tid >> 4;
    if (tid < kRmE) sh_rank[tid] = 0;
    const int e = blockIdx.x * kRmE + el;
    double ce = INFINITY;
    if (e < K) { const double v = cost[(size_t)b * K + e]; ce = (v != v) ? INFINITY : v; }      // NaN ranks like +inf (see k_sortperm_rank)
    int cnt = 0;
    for (int c0 = 0; c0 < K; c0 += kRbChunk) {
        const int nc = min(kRbChunk, K - c0);
        __syncthreads();
        for (int j = tid; j < nc; j += 256) { const double v = cost[(size_t)b * K + c0 + j]; c_l[j] = (v != v) ? INFINITY : v; }
        __syncthreads();
        const int JR = ((nc + 15) / 16 + 3) & ~3, j0 = js * JR, j1 = min(nc, j0 + JR);
        int j = j0;
        for (; j + 4 <= j1; j += 4) {
            const double c0v = c_l[j], c1 = c_l[j + 1], c2 = c_l[j + 2], c3 = c_l[j + 3];
            const int g = c0 + j;
            cnt += ((c0v < ce) || (c0v == ce && g < e)) + ((c1 < ce) || (c1 == ce && g + 1 < e)) + ((c2 < ce) || (c2 == ce && g + 2 < e)) +
                   ((c3 < ce) || (c3 == ce && g + 3 < e));
        }
        for (; j < j1; ++j) cnt += ((c_l[j] < ce) || (c_l[j] == ce && c0 + j < e));
    }
    cnt += __shfl_xor(cnt, 16, 64);
    cnt += __shfl_xor(cnt, 32, 64);
    if ((tid & 63) < kRmE) atomicAdd(&sh_rank[el], cnt);
    __syncthreads();
    if (tid < kRmE && e < K) {
        const int rank = sh_rank[el];
        order[(size_t)b * K + rank] = e;
        if (skey && rank < m_elite) __hip_atomic_store((unsigned long long*)&skey[(size_t)b * K + rank], (unsigned long long)__double_as_longlong(ce), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (m_elite < 2 || !active || !skey || !done) return;
    elite_break_last_workgroup(skey, done, m_elite, active, b, K, &sh_last, eb);
}

// order = sortperm(cost) per slot, and (m_elite >= 2) the elite early-break check on the sorted costs.
// skey ([B][K] doubles) / done ([B] ints, zero between launches): workspace of the chip-wide rank sort (nullptr: never used)
void launch_sortperm(const double* cost, int32_t* order, int B, int K, int m_elite, int* active, hipStream_t s, double* skey, int* done) {
    if (K <= 256) { hipLaunchKernelGGL(k_sortperm_rank, dim3(B), dim3(256), 0, s, cost, order, K, m_elite, active); return; }
    static const int env_rm = [] { const char* e = getenv("MPOPIS_SORT_MULTI"); return e ? atoi(e) : 1; }();       // 0: bitonic only (A/B)
    // measured, C4 (K = 4096): 49.5 -> 20.5 us per sort at one slot (step 5.97 -> 5.72 ms); at 8 slots 52 vs ~60 us, at 16 worse -- so: few slots only
    if (env_rm && skey && done && K >= 2048 && (long long)B * K * K <= 2ll * 4096 * 4096) {
        static std::atomic<unsigned long long> seenm{0};
        ensure_dyn_lds((const void*)k_sortperm_rank_multi, 96 * 1024, seenm);
        hipLaunchKernelGGL(k_sortperm_rank_multi, dim3((K + kRmE - 1) / kRmE, B), dim3(256), (size_t)K * sizeof(double), s, cost, order, K, m_elite, active, skey, done);
        return;
    }
    if (K > 8192) {                                                         // beyond the one-workgroup bitonic network: chunked chip-wide rank sort, any K
        hipLaunchKernelGGL(k_sortperm_rank_big, dim3((K + kRmE - 1) / kRmE, B), dim3(256), 0, s, cost, order, K, m_elite, active, skey, done);
        return;
    }
    int n = 512;
    while (n < K) n <<= 1;
    const size_t bytes = (size_t)n * (sizeof(double) + sizeof(int32_t));
    static std::atomic<unsigned long long> seen8{0};
    if (n >= 8192) {
        ensure_dyn_lds((const void*)k_sortperm<8>, 150 * 1024, seen8);
        hipLaunchKernelGGL(k_sortperm<8>, dim3(B), dim3(n / 8), bytes, s, cost, order, K, n, m_elite, active);
    } else if (n >= 2048) hipLaunchKernelGGL(k_sortperm<4>, dim3(B), dim3(n / 4), bytes, s, cost, order, K, n, m_elite, active);
    else hipLaunchKernelGGL(k_sortperm_lds, dim3(B), dim3(n / 2), bytes, s, cost, order, K, n, m_elite, active);
}

// StatsBase.make_alias_table!(w, 1.0, a, alias): Vose's construction with LIFO stacks of smalls and
// larges, executed in the reference's exact operation order (the result must be bit-identical for
// identical w).  The wave classifies with ballots (keeps index order), then runs the pairing loop (see below).
template <bool GLOBAL>
__global__ void __launch_bounds__(64) k_alias_build(const double* __restrict__ w, double* accept, int32_t* alias,
                                                    int K, const int* active, const int* need, int32_t* stack_ws) {
    MPOPIS_HI_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // GLOBAL = false: a / alias / the two stacks live in LDS (K <= kAliasLdsMaxK: 20 bytes per entry).  GLOBAL = true: any K (the reference's Categorical
    // has no size limit, :804) -- a and alias are built in place in the output arrays, the stacks in a global workspace (2K ints per slot).  ONE wave runs
    // the kernel, so every access is same-wave program order through this CU's write-through L1; the barriers / fences below order them as they do in LDS.
    // (volatile only in the global form: there every access is followed by its own wait, which is what orders a lane-0 store before another lane's later load)
    using DP = std::conditional_t<GLOBAL, volatile double*, double*>;
    using IP = std::conditional_t<GLOBAL, volatile int32_t*, int32_t*>;
    DP a = GLOBAL ? (DP)(accept + (size_t)blockIdx.x * K) : reinterpret_cast<DP>(smem);
    IP al = GLOBAL ? (IP)(alias + (size_t)blockIdx.x * K) : reinterpret_cast<IP>(smem + (size_t)K * 8);
    IP larges = GLOBAL ? (IP)(stack_ws + (size_t)blockIdx.x * 2 * K) : al + K;
    IP smalls = larges + K;
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    if (need && !need[b]) return;                               // the parallel construction was certain of every decision
    const int lane = threadIdx.x;
    const double ac = (double)K / 1.0;                          // n / wsum
    int kl = 0, ks = 0;
    // classification in index order; the loads of 16 rounds (1024 weights) are issued together (a dependent global round trip
    // costs a lone wave ~1.5 us: 64 of them would be 100 us)
    for (int c0 = 0; c0 < K; c0 += 64 * 16) {
        double wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = w[(size_t)b * K + min(c0 + 64 * u + lane, K - 1)];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = c0 + 64 * u + lane;
            if (c0 + 64 * u < K) {                                  // wave-uniform
                const double ai = (i < K) ? wv[u] * ac : 1.0;
                if (i < K) { a[i] = ai; al[i] = i; }
                const unsigned long long ml = __ballot(ai > 1.0), ms = __ballot(ai < 1.0);
                const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                if (ai > 1.0) larges[kl + __popcll(ml & lt)] = i;
                else if (ai < 1.0) smalls[ks + __popcll(ms & lt)] = i;
                kl += __popcll(ml); ks += __popcll(ms);
            }
        }
    }
    __syncthreads();
    // The pairing loop is sequential by definition (and its order fixes which index a later draw maps to), but a naive
    // transcription pays ~4 dependent LDS round trips per pairing (0.85 ms at K = 4096).  What the LIFO discipline implies:
    //   * a large is re-popped until it drops to <= 1, so it can stay in registers while it absorbs smalls;
    //   * no other large is ever pushed, so the larges stack only shrinks: the wave prefetches its top 64 entries (index and a)
    //     in one round trip, lane j <- j-th from the top, and walks them with v_readlane;
    //   * the same for the smalls stack, except that an exhausted large lands on top of it -- that one is kept in registers as the
    //     `pending` small and is the first thing the next large absorbs, exactly as in the reference.
    // The sequential part then runs on registers with the reference's operations in the reference's order (same bits); alias[s] = l
    // is recorded per lane and written when a batch of smalls is retired.
    {
        auto rl_d = [](double v, int j) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), j), __builtin_amdgcn_readlane(__double2loint(v), j)); };
        int idxS = 0, aliasS = -1, nS = 0, posS = 0;  double valS = 0.0;      // smalls batch: lane j <-> smalls[ks - 1 - j]
        int idxL = 0, nL = 0, posL = 0;               double valL = 0.0;      // larges batch (already popped: kl excludes it)
        int pend_idx = -1;                            double pend_val = 0.0;  // exhausted large = top of the smalls stack
        auto flushS = [&]() { if (lane < posS) al[idxS] = aliasS; ks -= posS; nS = 0; posS = 0; };
        auto loadS = [&]() { nS = min(64, ks); idxS = (lane < nS) ? smalls[ks - 1 - lane] : 0; valS = (lane < nS) ? a[idxS] : 0.0; posS = 0; };
        auto loadL = [&]() { nL = min(64, kl); idxL = (lane < nL) ? larges[kl - 1 - lane] : 0; valL = (lane < nL) ? a[idxL] : 0.0; posL = 0; kl -= nL; };
        while (true) {
            if (pend_idx < 0 && posS == nS) { flushS(); if (ks == 0) break; loadS(); }        // `ks > 0` of the reference's loop test
            if (posL == nL) { if (kl == 0) break; loadL(); }                                   // `kl > 0`
            const int l = __builtin_amdgcn_readlane(idxL, posL);
            double a_l = rl_d(valL, posL);
            posL += 1;
            bool alive = true;
            if (pend_idx >= 0) {                                                               // s = the large that just became small
                if (lane == 0) al[pend_idx] = l;
                a_l = (a_l - 1.0) + pend_val;
                pend_idx = -1;
                alive = a_l > 1.0;
            }
            bool dry = false;
            while (alive) {
                if (posS == nS) { flushS(); if (ks == 0) { dry = true; break; } loadS(); }
                const double as = rl_d(valS, posS);
                if (lane == posS) aliasS = l;                                                  // alias[s] = l
                posS += 1;
                a_l = (a_l - 1.0) + as;                                                        // a[l] = (a[l] - 1.0) + a[s]
                alive = a_l > 1.0;
            }
            if (lane == 0) a[l] = a_l;
            if (dry) break;                                                                    // smalls ran out, this large stays > 1
            pend_idx = l; pend_val = a_l;                                                      // pushed on the smalls
        }
        flushS();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int i = lane; i < ks; i += 64) a[smalls[i]] = 1.0;                                // "should be redundant, except for rounding"
        if (pend_idx >= 0 && lane == 0) a[pend_idx] = 1.0;
    }
    __syncthreads();
    if (!GLOBAL) for (int i = lane; i < K; i += 64) { accept[(size_t)b * K + i] = a[i]; alias[(size_t)b * K + i] = al[i]; }
}
// ---------------------------------------------------------------------------------------------
// The same table without the sequential loop.  What the LIFO discipline of make_alias_table! does, in cumulative terms: pop
// order = descending index; with e_i = a - 1 of the i-th popped large, d_j = 1 - a of the j-th popped small and their prefix sums
// E_i, D_j, the large in charge when small j is popped is the first one whose cumulative capacity exceeds the deficits absorbed so
// far, i(j) = min{i : E_i > D_{j-1}}; large i is exhausted by the first small m with D_m >= E_i, lands on the smalls stack with
// a = 1 - (D_m - E_i) and is absorbed by large i+1 (alias[L_i] = L_{i+1}); smalls left over when the larges run out, the last
// exhausted large and entries with a == 1 end with a = 1 and alias = self; a large that is never exhausted keeps a = 1 + (E_i - D_last).
// That is two prefix scans and one binary search per entry instead of K dependent steps (420 us at K = 4096).
// Exactness: alias[] is index work and must equal the sequential result.  The sequential partial sums differ from the scanned ones
// only by rounding (<= 2K operations of magnitude <= K: 4 K^2 eps), so every decision whose margin exceeds that bound is the
// sequential decision; if ANY comparison of a slot is closer than the bound (exact ties included), the slot is flagged and the
// sequential kernel redoes it.  accept[] of exhausted / partly used larges is 1 - (D - E) from the scans instead of the
// sequentially rounded chain: a floating-point deviation of <= 4 K^2 eps (typically 1e-13); all other entries are bit-identical.
// ---------------------------------------------------------------------------------------------
constexpr int kAliasParThreads = 1024;
constexpr int kAliasLdsMaxK = 7168;                                     // 20 bytes of LDS per entry in k_alias_build<false>
__global__ void __launch_bounds__(kAliasParThreads) k_alias_build_par(const double* __restrict__ w, double* __restrict__ accept, int32_t* __restrict__ alias,
                                                                      int K, const int* active, int* need) {
    MPOPIS_HI_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* pre = reinterpret_cast<double*>(smem);              // E_0 .. E_{nL-1} from the front, D_0 .. D_{nS-1} from the back (pre[K-1-j])
    int32_t* Lidx = reinterpret_cast<int32_t*>(smem + (size_t)K * 8);
    __shared__ double wsE[16], wsD[16];
    __shared__ int wcL[16], wcS[16], sh_unsafe;
    const int b = blockIdx.x;
    if (active && !active[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double ac = (double)K / 1.0;
    constexpr int EPT = 8;                                      // K <= 8192
    const int per = (K + kAliasParThreads - 1) / kAliasParThreads;             // entries per thread, consecutive in pop order (descending index)
    double av[EPT];
    double sE = 0.0, sD = 0.0; int cL = 0, cS = 0;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int r = tid * per + u;                            // pop-order position
        av[u] = 1.0;
        if (u < per && r < K) {
            av[u] = w[(size_t)b * K + (K - 1 - r)] * ac;
            if (av[u] > 1.0) { sE += av[u] - 1.0; ++cL; } else if (av[u] < 1.0) { sD += 1.0 - av[u]; ++cS; }
        }
    }
    if (tid == 0) sh_unsafe = 0;
    // exclusive block scans of (sE, sD, cL, cS) over the threads
    double xE = sE, xD = sD; int xL = cL, xS = cS;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double tE = __shfl_up(xE, o, 64), tD = __shfl_up(xD, o, 64); const int tL = __shfl_up(xL, o, 64), tS = __shfl_up(xS, o, 64);
        if (lane >= o) { xE += tE; xD += tD; xL += tL; xS += tS; }
    }
    if (lane == 63) { wsE[wv] = xE; wsD[wv] = xD; wcL[wv] = xL; wcS[wv] = xS; }
    __syncthreads();
    double oE = 0.0, oD = 0.0; int oL = 0, oS = 0, nL = 0, nS = 0;
    for (int q = 0; q < kAliasParThreads / 64; ++q) {
        if (q < wv) { oE += wsE[q]; oD += wsD[q]; oL += wcL[q]; oS += wcS[q]; }
        nL += wcL[q]; nS += wcS[q];
    }
    double runE = oE + (xE - sE), runD = oD + (xD - sD);        // exclusive prefixes of this thread
    int rL = oL + (xL - cL), rS = oS + (xS - cS);
    const int myL0 = rL, myS0 = rS;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        if (av[u] > 1.0) { runE += av[u] - 1.0; pre[rL] = runE; Lidx[rL] = K - 1 - (tid * per + u); ++rL; }
        else if (av[u] < 1.0) { runD += 1.0 - av[u]; pre[K - 1 - rS] = runD; ++rS; }
    }
    __syncthreads();
    const double tol = fmax(4.0 * (double)K * (double)K * 2.220446049250313e-16, 1e-12);
    auto Eat = [&](int i) { return pre[i]; };
    auto Dat = [&](int j) { return pre[K - 1 - j]; };
    const double Dtot = (nS > 0) ? Dat(nS - 1) : 0.0;            // all deficits, as the scan accumulated them
    int unsafe = 0;
    rL = myL0; rS = myS0;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int r = tid * per + u;
        if (!(u < per && r < K)) continue;
        const int i = K - 1 - r;
        double acc = 1.0; int al = i;
        if (av[u] < 1.0) {                                      // small number rS of the pop order
            const double lo = (rS > 0) ? Dat(rS - 1) : 0.0;
            int a0 = 0, a1 = nL;                                // first i with E_i > lo
            while (a0 < a1) { const int mid = (a0 + a1) >> 1; if (Eat(mid) > lo) a1 = mid; else a0 = mid + 1; }
            if (a0 < nL) { acc = av[u]; al = Lidx[a0]; if (!(Eat(a0) - lo > tol)) unsafe |= 1; }
            if (a0 > 0 && !(lo - Eat(a0 - 1) > tol)) unsafe |= 2;
            ++rS;
        } else if (av[u] > 1.0) {                               // large number rL
            const double Ei = Eat(rL), Ep = (rL > 0) ? Eat(rL - 1) : 0.0;
            int a0 = 0, a1 = nS;                                // first m with D_m >= E_i
            while (a0 < a1) { const int mid = (a0 + a1) >> 1; if (Dat(mid) >= Ei) a1 = mid; else a0 = mid + 1; }
            // The weights sum to 1, so total excess = total deficit in exact arithmetic: the LAST large meets the last small in an exact tie
            // that rounding decides either way -- harmlessly: nothing is popped after it, its alias stays itself and its acceptance is 1 (or
            // 1 + 1e-13: "always accept" both ways).  Its own comparisons are therefore exempt from the margin test (a small whose
            // pairing depends on that tie is still caught by its own test).
            const bool last = (rL + 1 == nL);
            if (a0 < nS) {                                      // exhausted by small a0
                if (!last && !(Dat(a0) - Ei > tol)) unsafe |= 4;
                if (!last) { acc = 1.0 - (Dat(a0) - Ei); al = Lidx[rL + 1]; }                  // absorbed by the next large; the last one ends at 1.0
            } else {
                if (!last && !(Ei - Dtot > tol)) unsafe |= 8;
                acc = (Dtot > Ep) ? fmax(1.0, 1.0 + (Ei - Dtot)) : av[u];                      // partly used / untouched: still a large
            }
            if (!last && a0 > 0 && !(Ei - Dat(a0 - 1) > tol)) unsafe |= 16;
            ++rL;
        }
        accept[(size_t)b * K + i] = acc;
        alias[(size_t)b * K + i] = al;
    }
    if (unsafe) atomicOr(&sh_unsafe, unsafe);
    __syncthreads();
    if (tid == 0) need[b] = sh_unsafe;
}

// need_ws (nullable): B ints of workspace; with it the parallel construction runs first and the sequential kernel only redoes the slots
// that the parallel one could not certify
void launch_alias_build(const double* w, double* accept, int32_t* alias, int B, int K, const int* active, hipStream_t s, int* need_ws, int32_t* stack_ws) {
    const size_t bytes = (size_t)K * (8 + 4 + 4 + 4);
    static std::atomic<unsigned long long> seen{0}, seenp{0};
    static const int env_par = [] { const char* e = getenv("MPOPIS_ALIAS_PAR"); return e ? atoi(e) : 1; }();
    if (K > kAliasLdsMaxK) {                                    // beyond what LDS holds: the sequential construction on global arrays (stack_ws: B x 2K ints)
        hipLaunchKernelGGL(k_alias_build<true>, dim3(B), dim3(64), 0, s, w, accept, alias, K, active, (const int*)nullptr, stack_ws);
        return;
    }
    const bool par = need_ws && env_par && K <= 8192;
    if (par) {
        ensure_dyn_lds((const void*)k_alias_build_par, 150 * 1024, seenp);
        hipLaunchKernelGGL(k_alias_build_par, dim3(B), dim3(kAliasParThreads), (size_t)K * 12, s, w, accept, alias, K, active, need_ws);
        if (getenv("MPOPIS_ALIAS_DEBUG")) { std::vector<int> nd(B); (void)hipStreamSynchronize(s); (void)hipMemcpy(nd.data(), need_ws, B * 4, hipMemcpyDeviceToHost); fprintf(stderr, "alias need:"); for (int v : nd) fprintf(stderr, " %d", v); fprintf(stderr, "\n"); }
    }
    ensure_dyn_lds((const void*)k_alias_build<false>, 160 * 1024, seen);
    hipLaunchKernelGGL(k_alias_build<false>, dim3(B), dim3(64), bytes, s, w, accept, alias, K, active, par ? (const int*)need_ws : (const int*)nullptr, (int32_t*)nullptr);
}
int alias_lds_max_K() { return kAliasLdsMaxK; }

// rand(rng, s::AliasTable): i = rand(1:n); u = rand(); u < accept[i] ? i : alias[i]
__global__ void __launch_bounds__(256) k_alias_sample(const double* __restrict__ accept, const int32_t* __restrict__ alias,
                                                      const int32_t* __restrict__ di, size_t di_stride, const double* __restrict__ du,
                                                      int32_t* __restrict__ out, int32_t* __restrict__ log, size_t log_stride, int K,
                                                      const int* active) {
    MPOPIS_HI_PRIO();
    const int b = blockIdx.y;
    if (active && !active[b]) return;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const int i = di[(size_t)b * di_stride + k];
    const double u = du[(size_t)b * di_stride + k];
    const int r = (u < accept[(size_t)b * K + i]) ? i : alias[(size_t)b * K + i];
    out[(size_t)b * K + k] = r;
    if (log) log[(size_t)b * log_stride + k] = r;
}
void launch_alias_sample(const double* accept, const int32_t* alias, const int32_t* di, size_t di_stride, const double* du,
                         int32_t* out, int32_t* log, size_t log_stride, int B, int K, const int* active, hipStream_t s) {
    hipLaunchKernelGGL(k_alias_sample, dim3((K + 255) / 256, B), dim3(256), 0, s, accept, alias, di, di_stride, du, out, log, log_stride, K, active);
}

}  // namespace mpopis
