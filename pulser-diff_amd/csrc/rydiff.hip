// rydiff.hip — MI355X (gfx950, wave64) kernels + the C ABI declared in include/rydiff.h.
//
// The hot path of pulser_diff.backend.TorchEmulator (pulser_diff/backend.py:488-494 calling
// pulser_diff/hamiltonian.py:526-546 on every solver sub-step) re-designed matrix-free:
//
//   K0 k_expand_coeffs   per-exponential effective coefficients from the sampled tables (hamiltonian.py:532-542)
//   K1 k_factor_*        y = gamma*x + beta*H(t)x : one factor of the product-form propagator (never builds H)
//   K2 k_expect_diag     <psi|O|psi> for diagonal O (utils.py:79-81), wave-shuffle + LDS reduction
//   K3 k_factor_bwd_*    adjoint of K1 fused with the gradient contractions (replaces the autograd tape, derivative.py:40,76)
//   K4 k_inject          cotangent injection  lambda += grad_states + 2*ge*O*psi
//   K5 k_scatter_grads   per-exponential coefficient gradients -> table / tsave gradients
//   K6 k_build_udiag / k_ugrad   static interaction diagonal (hamiltonian.py:333-344,368-404) and its gradient
//
// No H matrix, no sparse algebra, no accumulator vectors: every factor pass reads the state once and writes it once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/rydiff.h"
#include "plan.hpp"
#include "poly.hpp"

using namespace rydiff;

// ------------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------------
// gradient accumulators are replicated so that concurrent blocks do not serialise on one address
constexpr int kGradReplicas = 64;
constexpr int kMaxRemote = 6;  // up to 2^6 GPUs in a state-sharded run
constexpr int kShardMaxBits = 6;  // natively driven sharded runs: up to 2^6 ranks

static thread_local std::string g_last_error;  // the only mutable per-thread state (include/rydiff.h: rydiff_last_error)

static int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(RYDIFF_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));               \
    } while (0)

#define LAUNCH_CHECK()                                                                                 \
    do {                                                                                               \
        hipError_t _e = hipGetLastError();                                                             \
        if (_e != hipSuccess) return fail(RYDIFF_EHIP, std::string("kernel launch: ") + hipGetErrorString(_e)); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
// HIP's __popc returns UNSIGNED: `count - __popc(x)` with count < popcount (a ones-counting detuning group has count 0) would wrap
__device__ __forceinline__ int popc_i(uint32_t v) { return int(__popc(v)); }

struct GroupArgs {
    int ga, gd;
    uint32_t amask[kMaxGroups];  // amplitude-index bit masks of the flip groups
    uint32_t dmask[kMaxGroups];  // amplitude-index bit masks of the detuning groups
    int dcnt[kMaxGroups];        // qubits per detuning group (0: the group counts ones, RydProblem.det_ones_terms)
    uint32_t cond;               // flip groups whose flips act only where the sibling qubit (index bit ^ 1) is 1 (amp_conditioned_terms)
};

// conditioned flip of index bit `bit` (one-hot): does it act on amplitude x?  Flip and sibling are different bits, so the partner
// x ^ bit passes the same test.
__device__ __forceinline__ bool flip_acts(uint32_t cond_groups, int q, uint32_t x, uint32_t bit) {
    if (!(cond_groups >> q & 1u)) return true;
    const uint32_t sib = (bit & 0x55555555u) ? (bit << 1) : (bit >> 1);
    return (x & sib) != 0u;
}

// dense two-qubit terms of the generator (include/rydiff.h: pair terms)
struct PairArgs {
    int n = 0;
    const double2* tab = nullptr;  // [n][2][16]: forward table, then its conjugate transpose
    uint32_t ma[RYDIFF_MAX_PAIR_TERMS];
    uint32_t mb[RYDIFF_MAX_PAIR_TERMS];
    // which relative flips delta = own ^ s the block of term t has at all (bit delta: some T[own][own ^ delta] != 0, in the block or its
    // conjugate transpose; bit 0 = the diagonal, 1 = flip b, 2 = flip a, 3 = flip both).  Collapse operators populate few of them —
    // dephasing (Z (x) Z) the diagonal only, relaxation / depolarizing the diagonal and the double flip — and the kernels skip the rest
    // uniformly: no coefficient reads, no partner reads.
    uint8_t dl[RYDIFF_MAX_PAIR_TERMS];
};

// sum_p sum_s T_p[4*own + s] * v[x with the pair's bits set to s];  which = 0: T, 1: T^dagger
__device__ __forceinline__ double2 pair_apply(const PairArgs& pa, int which, const double2* __restrict__ v, uint32_t x) {
    double2 acc = make_double2(0.0, 0.0);
    for (int t = 0; t < pa.n; ++t) {
        const uint32_t ma = pa.ma[t], mb = pa.mb[t];
        const int own = ((x & ma) ? 2 : 0) | ((x & mb) ? 1 : 0);
        const double2* __restrict__ row = pa.tab + (size_t(t) * 2 + which) * 16 + own * 4;
        const unsigned dm = pa.dl[t];
#pragma unroll
        for (int dlt = 0; dlt < 4; ++dlt) {
            if (!(dm >> dlt & 1u)) continue;  // uniform
            const double2 c = row[own ^ dlt];
            if (c.x == 0.0 && c.y == 0.0) continue;
            const double2 q = v[x ^ ((dlt & 2) ? ma : 0u) ^ ((dlt & 1) ? mb : 0u)];
            acc.x += c.x * q.x - c.y * q.y;
            acc.y += c.x * q.y + c.y * q.x;
        }
    }
    return acc;
}

struct FactorArgs {
    const double2* xin;
    double2* xout;
    const double* udiag;
    const double* coef;   // record of this exponential, trajectory 0: c_re[ga], c_im[ga], dcoef[gd]
    long coef_bstride;    // doubles between trajectories' records (0: shared)
    uint32_t dim;
    double gr, gi, br, bi;  // gamma, beta
    GroupArgs g;
    // optional: coefficient record passed by value, and contributions of vectors owned by OTHER GPUs (state sharding):
    //   y += rc_k * remote_k[x]   (the flip terms of the qubits that select the GPU; see pulser-diff_amd/sharded.py)
    int use_inline;
    double coef_inline[3 * kMaxGroups];
    int n_remote;
    const double2* remote[kMaxRemote];
    double rc[2 * kMaxRemote];
    PairArgs pair;
    // state-sharded run driven natively (ChainArgs documents the fields): slabs as trajectories, rank qubits as partner slabs
    int sh_bits = 0, sh_nl = 0, sh_rank_first = 0, sh_self = 0;
    const double2* sh_rem[kShardMaxBits] = {};
    int sh_grp[kShardMaxBits] = {};
    // fused <y|O|y> of the vector this launch produces (k_factor_direct_global only; last factor of a time step)
    const double* obs = nullptr;   // [n_obs][dim]
    double* expect_slot = nullptr; // &expect_out[0][k][0]
    int n_obs = 0;
    long exp_ostride = 0;          // n_tsave * B
};

struct FactorBwdArgs {
    const double2* gin;   // cotangent w.r.t. the factor's output
    const double2* xin;   // the factor's input (recomputed chain)
    double2* gout;        // cotangent w.r.t. the factor's input
    const double* udiag;
    const double* coef;
    long coef_bstride;
    double* ge;           // gradient record of this exponential, trajectory 0, replica 0: gcre[ga], gcim[ga], gd[gd], gtau
    long ge_bstride;
    long ge_rstride;      // doubles between replicas (NC+1)
    double* wtot;         // optional [dim]: accumulates Re(beta*conj(g)*x) for the U_ij gradient
    uint32_t dim;
    double gr, gi, br, bi;
    GroupArgs g;
    PairArgs pair;
    // Fused cotangent injection (replaces a separate k_inject launch and the host-side decision whether one is needed):
    // when gout is the cotangent at a save point k — xin is then the state there — add
    //   grad_states[k][b][x] + 2 * sum_o grad_expect[o][k][b] * obs[o][x] * xin[x]
    const double2* inj_gstate = nullptr;  // grad_states[k] ([B][dim]) or nullptr
    const double* inj_gexp = nullptr;     // &grad_expect[0][k][0] or nullptr
    const double* inj_obs = nullptr;      // [n_obs][dim]
    int inj_n_obs = 0;
    long inj_ostride = 0;                 // n_tsave * B
    long obs_bstride = 0, obs_ostride = 0;  // observable table: [n_obs][dim] (0, dim); sharded: one slab per rank (dim, B * dim)
    // state-sharded run (ChainArgs documents the fields): the cotangent slabs of the partner ranks enter the adjoint matvec, and —
    // through the re-indexed contraction below — the drive gradients of the rank qubits
    int sh_bits = 0, sh_nl = 0, sh_rank_first = 0, sh_self = 0;
    const double2* sh_rem[kShardMaxBits] = {};
    int sh_grp[kShardMaxBits] = {};
};

// the injected cotangent at amplitude x of trajectory b (see FactorBwdArgs); wave-uniform control flow
__device__ __forceinline__ double2 injected_cotangent(const double2* inj_gstate, const double* inj_gexp, const double* inj_obs,
                                                      int n_obs, long ostride, long obs_ostride, long obs_bstride, int b, size_t boff,
                                                      uint32_t x, const double2& psi) {
    double2 add = make_double2(0.0, 0.0);
    if (inj_gexp) {
        double wsum = 0.0;
        for (int o = 0; o < n_obs; ++o) {
            const double ge = inj_gexp[o * ostride + b];
            if (ge != 0.0) wsum += ge * inj_obs[size_t(o) * obs_ostride + size_t(b) * obs_bstride + x];
        }
        add.x = 2.0 * wsum * psi.x;
        add.y = 2.0 * wsum * psi.y;
    }
    if (inj_gstate) {
        const double2 g = inj_gstate[boff + x];
        add.x += g.x;
        add.y += g.y;
    }
    return add;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// one value per block: wave shuffle -> LDS -> one global atomic
__device__ __forceinline__ void block_atomic_add(double v, double* dst, double* lds /* >= 4 doubles */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 0; w < nw; ++w) s += lds[w];
        unsafeAtomicAdd(dst, s);
    }
}

// ------------------------------------------------------------------------------------------------
// K6: static interaction diagonal  U(x) = sum_{i<j} U_ij n_i(x) n_j(x),  n_j = 1 - bit_{N-1-j}(x)
// ------------------------------------------------------------------------------------------------
// sharded runs: one table per slab (blockIdx.y), evaluated at the global index x | (rank << nl)
__global__ void k_build_udiag(double* __restrict__ udiag, const double* __restrict__ u_pairs, int N, uint32_t dim, int nl = 0,
                              int rank_first = 0) {
    const uint32_t xl = blockIdx.x * blockDim.x + threadIdx.x;
    if (xl >= dim) return;
    udiag += size_t(blockIdx.y) * dim;
    const uint32_t x = xl | (nl ? ((uint32_t(rank_first) + blockIdx.y) << nl) : 0u);
    double s = 0.0;
    int k = 0;
    for (int i = 0; i < N; ++i) {
        const bool ni = !((x >> (N - 1 - i)) & 1u);
        for (int j = i + 1; j < N; ++j, ++k) {
            const bool nj = !((x >> (N - 1 - j)) & 1u);
            if (ni && nj) s += u_pairs[k];
        }
    }
    udiag[xl] = s;
}

// split form of the interaction diagonal for one tile layout (chain_kernels.hpp):
//   U(x) = utt[i] + vr[t][LT] + sum_{tile bits a with n_a(i)=1} vr[t][a],   x = x(t, i);   LT = 12 or 13 tile bits
__global__ void k_build_split(double* __restrict__ utt, double* __restrict__ vr, const double* __restrict__ u_pairs,
                              int N, int lo, int hs, int hb, unsigned tiles, int LT) {
    const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
    auto gbit = [&](int b) { return b < lo ? b : hs + (b - lo); };          // tile bit -> index bit
    auto upair = [&](int ib, int jb) {                                       // index bits -> U_ij
        int qi = N - 1 - ib, qj = N - 1 - jb;
        if (qi > qj) { int tmp = qi; qi = qj; qj = tmp; }
        return u_pairs[qi * (2 * N - qi - 1) / 2 + (qj - qi - 1)];
    };
    if (id < (1u << LT)) {
        double s = 0.0;
        for (int a = 0; a < LT; ++a)
            for (int b = a + 1; b < LT; ++b)
                if (!(id >> a & 1u) && !(id >> b & 1u)) s += upair(gbit(a), gbit(b));
        utt[id] = s;
    } else if (id - (1u << LT) < tiles) {
        const unsigned t = id - (1u << LT);
        const int midlow = hs - lo;
        const unsigned xbase = ((t & ((1u << midlow) - 1u)) << lo) | ((t >> midlow) << (hs + hb));
        uint32_t tile_bits = 0;
        for (int a = 0; a < LT; ++a) tile_bits |= 1u << gbit(a);
        double* row = vr + size_t(t) * 16;
        double urr = 0.0;
        for (int ib = 0; ib < N; ++ib) {
            if (tile_bits >> ib & 1u) continue;
            if (xbase >> ib & 1u) continue;  // n = 0
            for (int jb = ib + 1; jb < N; ++jb)
                if (!(tile_bits >> jb & 1u) && !(xbase >> jb & 1u)) urr += upair(ib, jb);
        }
        for (int a = 0; a < LT; ++a) {
            double v = 0.0;
            for (int jb = 0; jb < N; ++jb)
                if (!(tile_bits >> jb & 1u) && !(xbase >> jb & 1u)) v += upair(gbit(a), jb);
            row[a] = v;
        }
        row[LT] = urr;
        for (int c = LT + 1; c < 16; ++c) row[c] = 0.0;
    }
}

// g_u[pair] = sum_x n_i n_j wtot[x]
// sharded runs (slabs > 0): wtot holds one slab of 2^nl weights per rank of the call; amplitude x of slab b sits at the global index
// x | (rank_first + b) << nl — every rank adds its part, the caller sums g_u over the ranks
__global__ void k_ugrad(double* __restrict__ g_u, const double* __restrict__ wtot, int N, uint32_t dim, int slabs = 0, int nl = 0,
                        int rank_first = 0) {
    __shared__ double lds[8];
    const int pair = blockIdx.y;
    int i = 0, rem = pair;
    while (rem >= N - 1 - i) {
        rem -= N - 1 - i;
        ++i;
    }
    const int j = i + 1 + rem;
    const uint32_t mi = 1u << (N - 1 - i), mj = 1u << (N - 1 - j);
    double s = 0.0;
    if (slabs > 0) {
        for (int b = 0; b < slabs; ++b) {
            const uint32_t hi = uint32_t(rank_first + b) << nl;
            for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < dim; x += gridDim.x * blockDim.x)
                if (!((x | hi) & mi) && !((x | hi) & mj)) s += wtot[size_t(b) * dim + x];
        }
    } else {
        for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < dim; x += gridDim.x * blockDim.x)
            if (!(x & mi) && !(x & mj)) s += wtot[x];
    }
    block_atomic_add(s, g_u + pair, lds);
}

// ------------------------------------------------------------------------------------------------
// table statistics for the spectral bound (over sample index i, all trajectories):
//   stats[0] = max_i sum_g |c_g[i]| * count_g         (norm of the flip part, exact for commuting single-qubit terms)
//   stats[1] = max_i sum_g max(+dcoef_g[i],0)*count_g  stats[2] = max_i sum_g max(-dcoef_g[i],0)*count_g
//   stats[3] = sum of max(U_ij, 0)     stats[5] = sum of max(-U_ij, 0)   (the doubled register of the master-equation path
//                                                                        carries -U_ij on its column qubits)
//   stats[4] = max_i sum_g |Im c_g[i]|   (non-zero: some drive has a phase)
// all non-negative doubles -> their bit patterns order like unsigned integers (atomicMax on u64).
// ------------------------------------------------------------------------------------------------
struct StatsArgs {
    const double2* amp;
    const double* det;
    const double* u_pairs;
    int n_samples, Ka, Kd, n_pairs, Bc;
    int ga, gd;
    uint64_t amem[kMaxGroups], dmem[kMaxGroups];
    int acnt[kMaxGroups], dcnt[kMaxGroups];
    uint32_t dones;  // detuning groups that count ones: (count - popcount) ranges over [-dcnt, 0] instead of [0, dcnt]
};

__global__ void k_table_stats(unsigned long long* __restrict__ stats, StatsArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i < a.n_samples) {
        double flip = 0.0, dpos = 0.0, dneg = 0.0, imabs = 0.0;
        for (int g = 0; g < a.ga; ++g) {
            double re = 0.0, im = 0.0;
            for (int k = 0; k < a.Ka; ++k)
                if (a.amem[g] >> k & 1ull) {
                    double2 v = a.amp[(size_t(b) * a.Ka + k) * a.n_samples + i];
                    re += v.x;
                    im += v.y;
                }
            flip += sqrt(re * re + im * im) * a.acnt[g];
            imabs += fabs(im);
        }
        for (int g = 0; g < a.gd; ++g) {
            double d = 0.0;
            for (int k = 0; k < a.Kd; ++k)
                if (a.dmem[g] >> k & 1ull) d += 2.0 * a.det[(size_t(b) * a.Kd + k) * a.n_samples + i];
            if (a.dones >> g & 1u) d = -d;
            if (d > 0.0) dpos += d * a.dcnt[g];
            else dneg += -d * a.dcnt[g];
        }
        atomicMax(stats + 0, (unsigned long long)__double_as_longlong(flip));
        atomicMax(stats + 1, (unsigned long long)__double_as_longlong(dpos));
        atomicMax(stats + 2, (unsigned long long)__double_as_longlong(dneg));
        atomicMax(stats + 4, (unsigned long long)__double_as_longlong(imabs));
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        double sp = 0.0, sn = 0.0;
        for (int k = 0; k < a.n_pairs; ++k) {
            const double u = a.u_pairs[k];
            if (u > 0.0) sp += u;
            else sn -= u;
        }
        stats[3] = (unsigned long long)__double_as_longlong(sp);
        stats[5] = (unsigned long long)__double_as_longlong(sn);
    }
}

// ------------------------------------------------------------------------------------------------
// K0: effective coefficients of every exponential.  record = c_re[ga], c_im[ga], dcoef[gd]
//   c_g   = sum_{terms k in group g} sum_q w[e][q] * amp_k[idx[e][q]]        (hamiltonian.py:542)
//   dcoef = 2 * sum_{terms k in group g} sum_q w[e][q] * det_k[idx[e][q]]    (hamiltonian.py:538-540)
// ------------------------------------------------------------------------------------------------
// per-exponential metadata as it lives on the device (uploaded through kernel arguments, see upload_words)
struct StageDev {     // forward: the two samples entering the coefficient combination and their weights (hamiltonian.py:532-542)
    double w0, w1;
    int32_t i0, i1;
};
struct StageBwdDev {  // backward: how the exponential's duration and interpolation time depend on tsave
    double tau_scale, tnw0, tnw1;
    int32_t tn0, tn1, t_hi, t_lo;  // -1: not a tsave point
};
static_assert(sizeof(StageDev) == 24 && sizeof(StageBwdDev) == 40, "stage records are uploaded as 8-byte words");

struct ExpandArgs {
    const double2* amp;
    const double* det;
    const StageDev* st;  // [E]
    double* coef;        // [Bc][E][NC]
    int E, n_samples, Ka, Kd, NC, ga, gd;
    uint64_t amem[kMaxGroups], dmem[kMaxGroups];
};

__global__ void k_expand_coeffs(ExpandArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (e >= a.E) return;
    double* rec = a.coef + (size_t(b) * a.E + e) * a.NC;
    const StageDev sd = a.st[e];
    const int idx[2] = {sd.i0, sd.i1};
    const double w[2] = {sd.w0, sd.w1};
    for (int g = 0; g < a.ga; ++g) {
        double re = 0.0, im = 0.0;
        for (int k = 0; k < a.Ka; ++k)
            if (a.amem[g] >> k & 1ull) {
                const double2* t = a.amp + (size_t(b) * a.Ka + k) * a.n_samples;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (w[q] != 0.0) {
                        re += w[q] * t[idx[q]].x;
                        im += w[q] * t[idx[q]].y;
                    }
            }
        rec[g] = re;
        rec[a.ga + g] = im;
    }
    for (int g = 0; g < a.gd; ++g) {
        double d = 0.0;
        for (int k = 0; k < a.Kd; ++k)
            if (a.dmem[g] >> k & 1ull) {
                const double* t = a.det + (size_t(b) * a.Kd + k) * a.n_samples;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (w[q] != 0.0) d += w[q] * t[idx[q]];
            }
        rec[2 * a.ga + g] = 2.0 * d;
    }
}

// ------------------------------------------------------------------------------------------------
// K1 (direct variant): one amplitude per thread, partners fetched from global memory (L2 / Infinity Cache).
//   y[x] = (gamma + beta*d(x)) psi[x] + beta * sum_g [ c_g * sum_{j in g, bit_j(x)=1} psi[x^m_j]
//                                                   + conj(c_g) * sum_{j in g, bit_j(x)=0} psi[x^m_j] ]
//   d(x) = U(x) + sum_g dcoef_g * (#qubits of g in |r>)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double diag_value(const double* __restrict__ udiag, const double* __restrict__ cf, const GroupArgs& g,
                                             uint32_t x, uint32_t xglob) {
    double d = udiag[x];
    for (int q = 0; q < g.gd; ++q) d += cf[2 * g.ga + q] * double(g.dcnt[q] - popc_i(xglob & g.dmask[q]));
    return d;
}
__device__ __forceinline__ double diag_value(const double* __restrict__ udiag, const double* __restrict__ cf, const GroupArgs& g,
                                             uint32_t x) {
    return diag_value(udiag, cf, g, x, x);
}

__global__ __launch_bounds__(256) void k_factor_direct(FactorArgs a) {
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (x >= a.dim) return;
    const size_t boff = size_t(blockIdx.y) * a.dim;
    const double2* __restrict__ xin = a.xin + boff;
    const double* __restrict__ cf = a.use_inline ? a.coef_inline : a.coef + blockIdx.y * a.coef_bstride;
    const unsigned rank = unsigned(a.sh_rank_first) + blockIdx.y;
    const uint32_t xglob = a.sh_bits ? (x | (rank << a.sh_nl)) : x;  // sharded: the diagonal lives at the global index
    const double d = diag_value(a.udiag + (a.sh_bits ? boff : 0), cf, a.g, x, xglob);
    const double2 v = xin[x];
    const double dr = a.gr + a.br * d, di = a.gi + a.bi * d;
    double ar = dr * v.x - di * v.y, ai = dr * v.y + di * v.x;
    for (int k = 0; k < a.sh_bits; ++k) {  // flips of the rank qubits: partner slabs
        if (a.sh_grp[k] < 0) continue;
        const double cr = cf[a.sh_grp[k]], ci = (rank >> k & 1u) ? cf[a.g.ga + a.sh_grp[k]] : -cf[a.g.ga + a.sh_grp[k]];
        const double kr = a.br * cr - a.bi * ci, ki = a.br * ci + a.bi * cr;
        const double2 rv = a.sh_self ? a.xin[size_t(blockIdx.y ^ (1u << k)) * a.dim + x] : a.sh_rem[k][boff + x];
        ar += kr * rv.x - ki * rv.y;
        ai += kr * rv.y + ki * rv.x;
    }
    for (int k = 0; k < a.n_remote; ++k) {
        const double2 rv = a.remote[k][boff + x];
        ar += a.rc[2 * k] * rv.x - a.rc[2 * k + 1] * rv.y;
        ai += a.rc[2 * k] * rv.y + a.rc[2 * k + 1] * rv.x;
    }
    for (int q = 0; q < a.g.ga; ++q) {
        double s1r = 0.0, s1i = 0.0, s0r = 0.0, s0i = 0.0;
        uint32_t m = a.g.amask[q];
        while (m) {
            const uint32_t bit = m & (0u - m);
            m ^= bit;
            if (!flip_acts(a.g.cond, q, x, bit)) continue;
            const double2 p = xin[x ^ bit];
            if (x & bit) {
                s1r += p.x;
                s1i += p.y;
            } else {
                s0r += p.x;
                s0i += p.y;
            }
        }
        const double cr = cf[q], ci = cf[a.g.ga + q];
        // beta*c and beta*conj(c)
        const double b1r = a.br * cr - a.bi * ci, b1i = a.br * ci + a.bi * cr;
        const double b0r = a.br * cr + a.bi * ci, b0i = -a.br * ci + a.bi * cr;
        ar += b1r * s1r - b1i * s1i + b0r * s0r - b0i * s0i;
        ai += b1r * s1i + b1i * s1r + b0r * s0i + b0i * s0r;
    }
    if (a.pair.n) {  // beta * (dense two-qubit terms)
        const double2 pv = pair_apply(a.pair, 0, xin, x);
        ar += a.br * pv.x - a.bi * pv.y;
        ai += a.br * pv.y + a.bi * pv.x;
    }
    a.xout[boff + x] = make_double2(ar, ai);
}

// ------------------------------------------------------------------------------------------------
// K3 (direct variant): adjoint of one factor + gradient contractions.
//   gout = (conj(gamma) + conj(beta) H) gin
//   dL/dRe c_g += Re( beta * sum_x conj(gin[x]) * (partner sums of xin) )      dL/dIm c_g likewise with +-i
//   dL/ddcoef_g += sum_x cnt_g(x) * Re( beta conj(gin[x]) xin[x] )
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_factor_bwd_direct(FactorBwdArgs a) {
    __shared__ double lds[8];
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    const bool live = x < a.dim;
    const size_t boff = size_t(blockIdx.y) * a.dim;
    const double2* __restrict__ gin = a.gin + boff;
    const double2* __restrict__ xin = a.xin + boff;
    const double* __restrict__ cf = a.coef + blockIdx.y * a.coef_bstride;
    double* __restrict__ ge = a.ge + blockIdx.y * a.ge_bstride + (blockIdx.x % kGradReplicas) * a.ge_rstride;
    const uint32_t xs = live ? x : 0u;
    const unsigned rank = unsigned(a.sh_rank_first) + blockIdx.y;
    const uint32_t xglob = a.sh_bits ? (xs | (rank << a.sh_nl)) : xs;  // sharded: the diagonal lives at the global index
    const double d = diag_value(a.udiag + (a.sh_bits ? boff : 0), cf, a.g, xs, xglob);
    double2 gy = gin[xs];
    double2 xi = xin[xs];
    if (!live) {
        gy = make_double2(0.0, 0.0);
        xi = make_double2(0.0, 0.0);
    }
    // adjoint matvec with conj(gamma), conj(beta)
    const double dr = a.gr + a.br * d, di = -(a.gi + a.bi * d);
    double ar = dr * gy.x - di * gy.y, ai = dr * gy.y + di * gy.x;
    // a_ = beta * conj(gy)
    const double pr = a.br * gy.x + a.bi * gy.y, pi = a.bi * gy.x - a.br * gy.y;
    const double r = pr * xi.x - pi * xi.y;  // Re(beta conj(gy) xi)
    if (a.wtot && live) unsafeAtomicAdd(a.wtot + (a.sh_bits ? boff : 0) + x, r);  // sharded: one weight slab per rank (k_ugrad)
    for (int q = 0; q < a.g.ga; ++q) {
        double s1r = 0.0, s1i = 0.0, s0r = 0.0, s0i = 0.0;  // partner sums of gin: for the matvec AND for the contraction
        uint32_t m = a.g.amask[q];
        while (m) {
            const uint32_t bit = m & (0u - m);
            m ^= bit;
            if (!flip_acts(a.g.cond, q, xs, bit)) continue;
            const double2 p = gin[xs ^ bit];
            if (xs & bit) {
                s1r += p.x; s1i += p.y;
            } else {
                s0r += p.x; s0i += p.y;
            }
        }
        for (int k = 0; k < a.sh_bits; ++k) {  // flips of the rank qubits of this group: the partner ranks' cotangent slabs
            if (a.sh_grp[k] != q) continue;
            const double2 p = a.sh_self ? a.gin[size_t(blockIdx.y ^ (1u << k)) * a.dim + xs] : a.sh_rem[k][boff + xs];
            if (rank >> k & 1u) {
                s1r += p.x; s1i += p.y;
            } else {
                s0r += p.x; s0i += p.y;
            }
        }
        const double cr = cf[q], ci = cf[a.g.ga + q];
        // conj(beta)*c and conj(beta)*conj(c)
        const double b1r = a.br * cr + a.bi * ci, b1i = a.br * ci - a.bi * cr;
        const double b0r = a.br * cr - a.bi * ci, b0i = -a.br * ci - a.bi * cr;
        ar += b1r * s1r - b1i * s1i + b0r * s0r - b0i * s0i;
        ai += b1r * s1i + b1i * s1r + b0r * s0i + b0i * s0r;
        // S1 = sum_x a_(x) t1(x), S0 = sum_x a_(x) t0(x) with t1 / t0 the partner sums of xin over the bits that are 1 / 0 in x;
        // g_cre = Re(S1+S0), g_cim = -Im(S1-S0).  Re-indexed over the partner (the flip is an involution that toggles the bit):
        // S1 = sum_y xin(y) beta conj(s0(y)), S0 = sum_y xin(y) beta conj(s1(y)) — the cotangent's partner sums, which the matvec
        // needs anyway, and the OWN tape element only: no partner loads of the tape vector.
        double gre = 0.0, gim = 0.0;
        if (live) {
            const double q0r = a.br * s0r + a.bi * s0i, q0i = a.bi * s0r - a.br * s0i;  // beta conj(s0)
            const double q1r = a.br * s1r + a.bi * s1i, q1i = a.bi * s1r - a.br * s1i;  // beta conj(s1)
            const double S1r = q0r * xi.x - q0i * xi.y, S1i = q0r * xi.y + q0i * xi.x;
            const double S0r = q1r * xi.x - q1i * xi.y, S0i = q1r * xi.y + q1i * xi.x;
            gre = S1r + S0r;
            gim = -(S1i - S0i);
        }
        block_atomic_add(gre, ge + q, lds);
        block_atomic_add(gim, ge + a.g.ga + q, lds);
    }
    for (int q = 0; q < a.g.gd; ++q) {
        const double v = live ? r * double(a.g.dcnt[q] - popc_i(xglob & a.g.dmask[q])) : 0.0;
        block_atomic_add(v, ge + 2 * a.g.ga + q, lds);
    }
    if (a.pair.n && live) {  // conj(beta) * (pair terms)^dagger applied to the cotangent
        const double2 pv = pair_apply(a.pair, 1, gin, x);
        ar += a.br * pv.x + a.bi * pv.y;
        ai += a.br * pv.y - a.bi * pv.x;
    }
    if ((a.inj_gexp || a.inj_gstate) && live) {
        const double2 add = injected_cotangent(a.inj_gstate, a.inj_gexp, a.inj_obs, a.inj_n_obs, a.inj_ostride, a.obs_ostride, a.obs_bstride,
                                               blockIdx.y, boff, x, xi);
        ar += add.x;
        ai += add.y;
    }
    if (live) a.gout[boff + x] = make_double2(ar, ai);
}

// ------------------------------------------------------------------------------------------------
// Direct kernels for ONE GLOBAL DRIVE on a register of exactly NQ qubits (every bit in the amplitude mask; no remote
// vectors, no pair terms).  The generic kernels above walk the set bits of a runtime mask: one partner load, one wait per bit
// — on 13..18 qubits, where a pass is a few microseconds, that chain of N dependent L2 latencies IS the kernel time.  Here the
// loop over the NQ bits is unrolled, so all partner loads are in flight together and the plain / signed partner sums replace
// the per-bit branch (c*s1 + conj(c)*s0 = cr*(s1+s0) + i*ci*(s1-s0)).
// ------------------------------------------------------------------------------------------------
// ONEXCD (12 and 13 qubits: <= 32 workgroups): the grid is 8x oversubscribed and only the workgroups that the round-robin dispatch
// places on XCD (trajectory % 8) work, so that a trajectory's vectors stay in ONE XCD's L2 from pass to pass — the partner
// loads then hit that L2 instead of crossing the fabric (placement is a speed matter only: results do not depend on it).
// Measured forward steps/s with / without: N=13 39.7 k / 23.9 k, N=14 24.3 k / 22.0 k (but its adjoint 10 % slower), N=15 21.8 k /
// 28.6 k, N=16 13.2 k / 24.6 k — one XCD's 32 CUs are not enough from 14 qubits on.
template <int NQ, bool ONEXCD>
__global__ __launch_bounds__(256) void k_factor_direct_global(FactorArgs a) {
    if (ONEXCD && (blockIdx.x & 7u) != (blockIdx.y & 7u)) return;
    const uint32_t x = (ONEXCD ? (blockIdx.x >> 3) : blockIdx.x) * 256u + threadIdx.x;  // dim = 2^NQ is a multiple of 256
    const size_t boff = size_t(blockIdx.y) * a.dim;
    const double2* __restrict__ xin = a.xin + boff;
    const double* __restrict__ cf = a.use_inline ? a.coef_inline : a.coef + blockIdx.y * a.coef_bstride;
    double2 p[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) p[j] = xin[x ^ (1u << j)];
    const double2 v = xin[x];
    const double d = diag_value(a.udiag, cf, a.g, x);
    double tsr = 0.0, tsi = 0.0, dsr = 0.0, dsi = 0.0;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const double sgn = (x >> j & 1u) ? 1.0 : -1.0;
        tsr += p[j].x;
        tsi += p[j].y;
        dsr = fma(sgn, p[j].x, dsr);
        dsi = fma(sgn, p[j].y, dsi);
    }
    const double dr = a.gr + a.br * d, di = a.gi + a.bi * d;
    const double cr = cf[0], ci = cf[1];
    // F = cr*ts + i*ci*ds
    const double fr = cr * tsr - ci * dsi, fi = cr * tsi + ci * dsr;
    const double2 y = make_double2(dr * v.x - di * v.y + a.br * fr - a.bi * fi, dr * v.y + di * v.x + a.br * fi + a.bi * fr);
    a.xout[boff + x] = y;
    if (a.obs) {  // wave-uniform: <y|O|y> for diagonal observables straight from the register that holds y
        __shared__ double lds[8];
        const double w = y.x * y.x + y.y * y.y;
        for (int o = 0; o < a.n_obs; ++o) block_atomic_add(a.obs[size_t(o) * a.dim + x] * w, a.expect_slot + o * a.exp_ostride + blockIdx.y, lds);
    }
}

template <int NQ, bool ONEXCD>
__global__ __launch_bounds__(256) void k_factor_bwd_direct_global(FactorBwdArgs a) {
    __shared__ double lds[8];
    __shared__ double lds3[12];  // 3 values x 4 waves
    if (ONEXCD && (blockIdx.x & 7u) != (blockIdx.y & 7u)) return;
    const uint32_t wg = ONEXCD ? (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t x = wg * 256u + threadIdx.x;
    const size_t boff = size_t(blockIdx.y) * a.dim;
    const double2* __restrict__ gin = a.gin + boff;
    const double2* __restrict__ xin = a.xin + boff;
    const double* __restrict__ cf = a.coef + blockIdx.y * a.coef_bstride;
    double* __restrict__ ge = a.ge + blockIdx.y * a.ge_bstride + (wg % kGradReplicas) * a.ge_rstride;
    double2 pg[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) pg[j] = gin[x ^ (1u << j)];
    const double2 gy = gin[x], xi = xin[x];
    const double d = diag_value(a.udiag, cf, a.g, x);
    double gsr = 0.0, gsi = 0.0, gdr = 0.0, gdi = 0.0;  // plain / signed partner sums of the cotangent
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const double sgn = (x >> j & 1u) ? 1.0 : -1.0;
        gsr += pg[j].x;
        gsi += pg[j].y;
        gdr = fma(sgn, pg[j].x, gdr);
        gdi = fma(sgn, pg[j].y, gdi);
    }
    const double cr = cf[0], ci = cf[1];
    // adjoint matvec: conj(gamma + beta d) gy + conj(beta) (cr*gs + i*ci*gd)
    const double dr = a.gr + a.br * d, di = -(a.gi + a.bi * d);
    const double fr = cr * gsr - ci * gdi, fi = cr * gsi + ci * gdr;
    double2 go = make_double2(dr * gy.x - di * gy.y + a.br * fr + a.bi * fi, dr * gy.y + di * gy.x + a.br * fi - a.bi * fr);
    if (a.inj_gexp || a.inj_gstate) {
        const double2 add = injected_cotangent(a.inj_gstate, a.inj_gexp, a.inj_obs, a.inj_n_obs, a.inj_ostride, a.obs_ostride, a.obs_bstride,
                                               blockIdx.y, boff, x, xi);
        go.x += add.x;
        go.y += add.y;
    }
    a.gout[boff + x] = go;
    // contractions with a_ = beta * conj(gy):  dL/dRe c = Re(a_ * xs),  dL/dIm c = -Im(a_ * xd)
    const double pr = a.br * gy.x + a.bi * gy.y, pi = a.bi * gy.x - a.br * gy.y;
    const double r = pr * xi.x - pi * xi.y;  // Re(beta conj(gy) xi)
    if (a.wtot) unsafeAtomicAdd(a.wtot + x, r);
    // dL/dRe c = Re sum_x a_(x) xs(x), dL/dIm c = -Im sum_x a_(x) xd(x) with xs / xd the plain / signed partner sums of the TAPE
    // vector — re-indexed over the partner: sum_x a_ xs = sum_y xin(y) beta conj(gs(y)), sum_x a_ xd = -sum_y xin(y) beta conj(gd(y))
    // (flipping bit j toggles its sign): the cotangent's partner sums and the own tape element, no partner loads of the tape.
    const double qsr = a.br * gsr + a.bi * gsi, qsi = a.bi * gsr - a.br * gsi;  // beta conj(gs)
    const double qdr = a.br * gdr + a.bi * gdi, qdi = a.bi * gdr - a.br * gdi;  // beta conj(gd)
    // the two drive gradients and the first detuning gradient share ONE workgroup reduction (one pair of barriers)
    double v0 = wave_sum(qsr * xi.x - qsi * xi.y), v1 = wave_sum(qdr * xi.y + qdi * xi.x);
    double v2 = wave_sum(a.g.gd > 0 ? r * double(a.g.dcnt[0] - popc_i(x & a.g.dmask[0])) : 0.0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        lds3[wave] = v0;
        lds3[4 + wave] = v1;
        lds3[8 + wave] = v2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double sum = lds3[4 * threadIdx.x] + lds3[4 * threadIdx.x + 1] + lds3[4 * threadIdx.x + 2] + lds3[4 * threadIdx.x + 3];
        if (threadIdx.x < 2 || a.g.gd > 0) unsafeAtomicAdd(ge + threadIdx.x, sum);
    }
    for (int q = 1; q < a.g.gd; ++q) block_atomic_add(r * double(a.g.dcnt[q] - popc_i(x & a.g.dmask[q])), ge + 2 + q, lds);
}

// dL/dtau of one exponential:  Re< g, -i H x >  = Im( sum_x conj(g[x]) (H x)[x] )
struct DotHArgs {
    const double2* g;
    const double2* x;
    const double* udiag;
    const double* coef;
    long coef_bstride;
    double* out;  // ge record + NC (gtau slot), trajectory 0
    long out_bstride;
    long out_rstride;
    uint32_t dim;
    int b_first;  // the grid's y dimension covers trajectories b_first, b_first + 1, ...
    GroupArgs gr;
    PairArgs pair;
};

__global__ __launch_bounds__(256) void k_dot_hx(DotHArgs a) {
    __shared__ double lds[8];
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    const bool live = x < a.dim;
    const uint32_t xs = live ? x : 0u;
    const int bt = a.b_first + int(blockIdx.y);
    const size_t boff = size_t(bt) * a.dim;
    const double2* __restrict__ xin = a.x + boff;
    const double* __restrict__ cf = a.coef + bt * a.coef_bstride;
    const double d = diag_value(a.udiag, cf, a.gr, xs);
    const double2 v = xin[xs];
    double hr = d * v.x, hi = d * v.y;
    for (int q = 0; q < a.gr.ga; ++q) {
        double s1r = 0.0, s1i = 0.0, s0r = 0.0, s0i = 0.0;
        uint32_t m = a.gr.amask[q];
        while (m) {
            const uint32_t bit = m & (0u - m);
            m ^= bit;
            if (!flip_acts(a.gr.cond, q, xs, bit)) continue;
            const double2 p = xin[xs ^ bit];
            if (xs & bit) { s1r += p.x; s1i += p.y; } else { s0r += p.x; s0i += p.y; }
        }
        const double cr = cf[q], ci = cf[a.gr.ga + q];
        hr += cr * s1r - ci * s1i + cr * s0r + ci * s0i;
        hi += cr * s1i + ci * s1r + cr * s0i - ci * s0r;
    }
    if (a.pair.n) {
        const double2 pv = pair_apply(a.pair, 0, xin, xs);
        hr += pv.x;
        hi += pv.y;
    }
    const double2 g = (a.g + boff)[xs];
    // Im(conj(g) * h) = g.x*hi - g.y*hr
    const double val = live ? (g.x * hi - g.y * hr) : 0.0;
    block_atomic_add(val, a.out + bt * a.out_bstride + (blockIdx.x % kGradReplicas) * a.out_rstride, lds);
}

// ------------------------------------------------------------------------------------------------
// K2: expectation values of diagonal observables, one launch per saved state.
// ------------------------------------------------------------------------------------------------
// obs_bstride / obs_ostride: 0 / dim for one observable table shared by the batch; sharded runs: dim / B*dim (one slab per rank)
__global__ __launch_bounds__(256) void k_expect_diag(const double2* __restrict__ psi, const double* __restrict__ obs,
                                                     double* __restrict__ out /* [n_obs][n_tsave][B] */, int n_obs,
                                                     int n_tsave, int k, int B, uint32_t dim, long obs_bstride = 0) {
    __shared__ double lds[8];
    const int b = blockIdx.y;
    const double2* __restrict__ p = psi + size_t(b) * dim;
    const size_t ostride = obs_bstride ? size_t(B) * dim : dim;
    for (int o = 0; o < n_obs; ++o) {
        double s = 0.0;
        for (uint32_t x = blockIdx.x * 256u + threadIdx.x; x < dim; x += gridDim.x * 256u) {
            const double2 v = p[x];
            s += obs[size_t(o) * ostride + size_t(b) * obs_bstride + x] * (v.x * v.x + v.y * v.y);
        }
        block_atomic_add(s, out + (size_t(o) * n_tsave + k) * B + b, lds);
    }
}

// ------------------------------------------------------------------------------------------------
// K4: lambda[b][x] (+)= grad_states[k][b][x] + 2 * sum_o ge[o][k][b] * obs[o][x] * psi_k[b][x]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_inject(double2* __restrict__ lam, const double2* __restrict__ gstate,
                                                const double2* __restrict__ psi, const double* __restrict__ obs,
                                                const double* __restrict__ gexp, int n_obs, int n_tsave, int k, int B,
                                                uint32_t dim, int overwrite, long obs_ostride, long obs_bstride) {
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (x >= dim) return;
    const int b = blockIdx.y;
    const size_t o_ = size_t(b) * dim + x;
    double2 acc = overwrite ? make_double2(0.0, 0.0) : lam[o_];
    if (gstate) {
        const double2 g = gstate[o_];
        acc.x += g.x;
        acc.y += g.y;
    }
    if (gexp && n_obs > 0) {
        double wsum = 0.0;
        for (int o = 0; o < n_obs; ++o) wsum += gexp[(size_t(o) * n_tsave + k) * B + b] * obs[size_t(o) * obs_ostride + size_t(b) * obs_bstride + x];
        const double2 v = psi[o_];
        acc.x += 2.0 * wsum * v.x;
        acc.y += 2.0 * wsum * v.y;
    }
    lam[o_] = acc;
}

// which save points carry a non-zero expectation cotangent (a loss on the final time leaves all others empty):
// flags[k] = any_{o,b} gexp[o][k][b] != 0
__global__ void k_cotangent_flags(const double* __restrict__ gexp, int n_obs, int n_tsave, int B, int32_t* __restrict__ flags) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_tsave) return;
    int any = 0;
    for (int o = 0; o < n_obs; ++o)
        for (int b = 0; b < B; ++b) any |= gexp[(size_t(o) * n_tsave + k) * B + b] != 0.0;
    flags[k] = any;
}

// ------------------------------------------------------------------------------------------------
// K5: scatter per-exponential coefficient gradients back onto the sampled tables and tsave.
// one thread per (exponential, trajectory); atomics because several exponentials touch one sample.
// ------------------------------------------------------------------------------------------------
struct ScatterArgs {
    const double* ge;       // [Bc][E][kGradReplicas][NC+1]
    const StageDev* st;     // [E]
    const StageBwdDev* sb;  // [E] (only read when g_tsave)
    double inv_dt;          // d w1 / d t = -d w0 / d t = 1/dt   (hamiltonian.py:538,542)
    const double2* amp;     // tables (for d coef / d t)
    const double* det;
    double2* g_amp;
    double* g_det;
    double* g_tsave;
    int E, n_samples, Ka, Kd, NC, ga, gd;
    uint64_t amem[kMaxGroups], dmem[kMaxGroups];
};

__global__ void k_scatter_grads(ScatterArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (e >= a.E) return;
    const double* reps = a.ge + (size_t(b) * a.E + e) * kGradReplicas * (a.NC + 1);
    double rec[2 * kMaxGroups + kMaxGroups + 1];
    for (int c = 0; c <= a.NC; ++c) {
        double sum = 0.0;
        for (int r = 0; r < kGradReplicas; ++r) sum += reps[size_t(r) * (a.NC + 1) + c];  // fixed order
        rec[c] = sum;
    }
    double dLdt = 0.0;
    const StageDev sd = a.st[e];
    const double wq[2] = {sd.w0, sd.w1};
    const int iq[2] = {sd.i0, sd.i1};
    const double dwq[2] = {a.g_tsave ? -a.inv_dt : 0.0, a.g_tsave ? a.inv_dt : 0.0};
    for (int k = 0; k < a.Ka; ++k) {
        double gr = 0.0, gi = 0.0;
        for (int g = 0; g < a.ga; ++g)
            if (a.amem[g] >> k & 1ull) {
                gr += rec[g];
                gi += rec[a.ga + g];
            }
        const double2* t = a.amp + (size_t(b) * a.Ka + k) * a.n_samples;
        for (int q = 0; q < 2; ++q) {
            const double w = wq[q], dw = dwq[q];
            const int i = iq[q];
            if (a.g_amp && w != 0.0) {
                double* dst = reinterpret_cast<double*>(a.g_amp + (size_t(b) * a.Ka + k) * a.n_samples + i);
                unsafeAtomicAdd(dst, w * gr);
                unsafeAtomicAdd(dst + 1, w * gi);
            }
            if (dw != 0.0) dLdt += dw * (gr * t[i].x + gi * t[i].y);
        }
    }
    for (int k = 0; k < a.Kd; ++k) {
        double gd = 0.0;
        for (int g = 0; g < a.gd; ++g)
            if (a.dmem[g] >> k & 1ull) gd += rec[2 * a.ga + g];
        const double* t = a.det + (size_t(b) * a.Kd + k) * a.n_samples;
        for (int q = 0; q < 2; ++q) {
            const double w = wq[q], dw = dwq[q];
            const int i = iq[q];
            if (a.g_det && w != 0.0) unsafeAtomicAdd(a.g_det + (size_t(b) * a.Kd + k) * a.n_samples + i, 2.0 * w * gd);
            if (dw != 0.0) dLdt += dw * 2.0 * gd * t[i];
        }
    }
    if (a.g_tsave) {
        const StageBwdDev sb = a.sb[e];
        const double gtau = rec[a.NC] * sb.tau_scale;
        if (sb.tn0 >= 0) unsafeAtomicAdd(a.g_tsave + sb.tn0, dLdt * sb.tnw0);
        if (sb.tn1 >= 0) unsafeAtomicAdd(a.g_tsave + sb.tn1, dLdt * sb.tnw1);
        if (sb.t_hi >= 0) unsafeAtomicAdd(a.g_tsave + sb.t_hi, gtau);
        if (sb.t_lo >= 0) unsafeAtomicAdd(a.g_tsave + sb.t_lo, -gtau);
    }
}

#include "chain_kernels.hpp"
#include "persist_kernels.hpp"
#include "lane_kernels.hpp"
static_assert(sizeof(PersistFactor) == 48, "plan.hpp sizes the factor table with 48 bytes per entry");

// ------------------------------------------------------------------------------------------------
// Host metadata -> device WITHOUT a copy engine or a synchronisation: the words travel as kernel arguments (the runtime
// copies them into the launch packet before hipLaunchKernel returns, so the host buffer may die right away) and one
// small workgroup writes them out.  Used for the per-exponential records (24 / 40 bytes each) and the pair tables.
// ------------------------------------------------------------------------------------------------
constexpr int kUploadWords = 448;  // 3584 bytes of payload per launch (kernel arguments are limited to 4 KiB)
struct UploadChunk {
    unsigned long long w[kUploadWords];
};

__global__ __launch_bounds__(256) void k_upload(unsigned long long* __restrict__ dst, UploadChunk c, int n) {
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = c.w[i];
}

// ------------------------------------------------------------------------------------------------
// Factor table of the one-launch sweeps (k_persist / k_lanes), built ON THE DEVICE from the per-exponential durations:
// the scalars of factor f of a sub-exponential of duration tau are
//   beta = -tau / (rho_d z_f),  gamma = 1 + tau sigma / (rho_d z_f),  both times exp(-i tau sigma) p(0) for the last factor
// (factor_scalars on the host).  One thread per tsave interval walks its exponentials, sub-steps and factors.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxDegreeDev = 96;
struct PTableArgs {
    PersistFactor* out;
    const double* tau_sub;      // [E]
    const int32_t* nsub;        // [E]
    const int32_t* step_begin;  // [T+1]
    const int32_t* step_first;  // [T]: index of the interval's first factor
    int T, degree;
    double sigma, rho_design, p0r, p0i;
    double roots[2 * kMaxDegreeDev];
};

__global__ __launch_bounds__(64) void k_build_ptable(PTableArgs a) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= a.T) return;
    int idx = a.step_first[k];
    const int first = idx;
    const int e_end = a.step_begin[k + 1];
    for (int e = a.step_begin[k]; e < e_end; ++e) {
        const double tau = a.tau_sub[e];
        const int ns = a.nsub[e];
        double sn, cs;
        sincos(-tau * a.sigma, &sn, &cs);
        const double kr = cs * a.p0r - sn * a.p0i, ki = cs * a.p0i + sn * a.p0r;  // kappa = exp(-i tau sigma) p(0)
        for (int sub = 0; sub < ns; ++sub)
            for (int f = 0; f < a.degree; ++f, ++idx) {
                // 1 / (rho_d z)
                const double zr = a.rho_design * a.roots[2 * f], zi = a.rho_design * a.roots[2 * f + 1];
                const double inv = 1.0 / (zr * zr + zi * zi);
                const double ir = zr * inv, ii = -zi * inv;
                double br = -tau * ir, bi = -tau * ii;
                double gr = 1.0 + tau * a.sigma * ir, gi = tau * a.sigma * ii;
                if (f == a.degree - 1) {
                    const double nbr = br * kr - bi * ki, nbi = br * ki + bi * kr;
                    const double ngr = gr * kr - gi * ki, ngi = gr * ki + gi * kr;
                    br = nbr; bi = nbi; gr = ngr; gi = ngi;
                }
                const bool last = (e == e_end - 1) && (sub == ns - 1) && (f == a.degree - 1);
                PersistFactor pf;
                pf.gr = gr; pf.gi = gi; pf.br = br; pf.bi = bi;
                pf.stage = e;
                pf.save_index = last ? k + 1 : 0;
                pf.step_first = first;
                pf.pad = 0;
                a.out[idx] = pf;
            }
    }
}

// ---- chained tile passes (chain_kernels.hpp) ------------------------------------------------------------------------
// Tile layouts: every layout keeps a contiguous low run of amplitudes so that global accesses stay coalesced.  LT = tile bits:
// 12 (k_chain: 64 KiB of LDS, 4 amplitudes per thread at 1024 threads) or 13 (k_chain_wide: 128 KiB, two register halves).
//   two layouts:    A = [0,LT)            B = [0,2LT-N) u [LT,N)
//   three layouts:  A = [0,LT)            B = [0,LT-8) u [LT,LT+8)        C = [0,2LT+8-N) u [LT+8,N)
// With three layouts a factor takes two launches (start in A or C, middle pass in B, finish in C or A — the finishing
// launch also starts the next factor), 4R+3W instead of 2R+2W: still far better than 16-byte runs in a two-layout B.
// Which (LT, layout count) a chain uses: chain_geom() below.
struct LayoutDesc {
    int lo, hs, hb;
    uint32_t bits;  // amplitude-index bits covered by the tile
};

struct ChainGeom {
    int lt;       // tile bits: kTileBits (12) or kWideTileBits (13)
    int layouts;  // 2 or 3
};

// split-diagonal tables of one tile size: [3 layouts][2^LT + tiles * 16] doubles; one set per tile size (plan.hpp: off_split)
double* split_tables(const Plan& pl, char* ws, int lt) {
    return reinterpret_cast<double*>(ws + pl.off_split) + pl.split_off_doubles[lt - kSmallTileBits];
}

LayoutDesc chain_layout(int N, int which, const ChainGeom& g) {
    LayoutDesc d{};
    const bool three = g.layouts == 3;
    if (which == 0) {  // A
        d.lo = g.lt;
        d.hs = g.lt;
        d.hb = 0;
    } else if (which == 1) {  // B
        d.hs = g.lt;
        d.hb = three ? 8 : N - g.lt;
        d.lo = g.lt - d.hb;
    } else {  // C (three-layout mode only)
        d.hs = g.lt + 8;
        d.hb = N - d.hs;
        d.lo = g.lt - d.hb;
    }
    d.bits = ((1u << d.lo) - 1u) | (((1u << d.hb) - 1u) << d.hs);
    return d;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

struct Runtime {
    Plan pl;
    PolyDesign poly;
    double sigma = 0.0, width = 1.0, rho_design = 1.0;
    int64_t total_factors = 0;
    int max_step_factors = 0;
    int flags = 0;
    bool real_amp_grad = false;  // RydProblem.real_amp_grad: dL/dIm(amp) is not wanted
    bool prefer_direct = false;  // few tiles in flight: one-amplitude-per-thread kernels instead of the chained tile passes
    bool small_tiles = false;    // ~2^19 amplitudes in flight: chained passes on tiles of 2^11 amplitudes (256 tiles: one per CU)
    // RydProblem.kernel_variant decoded (include/rydiff.h); nothing about the kernel choice lives outside this struct
    int variant = 0;              // 0 auto | 1 direct | 2..4 chained tiles | 8 auto with LDS-tile kernels below 7 qubits
    bool generic_direct = false;  // variant 9: direct kernels without the unrolled global-drive instantiations
    bool plain_tile_order = false;  // variant 12: no line-sharing tile swizzle (ChainArgs.tile_swz)
    int force_three = 0;          // 1: variant 7, three tile layouts wherever they are legal; 2: variant 11, two layouts up to 24 qubits
    int tile_mode = 0;            // 0 automatic | 12: variant 13, 2^12-amplitude tiles everywhere | 13: variant 14, wide tiles from 14 qubits
                                  // 11 / 10: variants 15 / 16, tiles of 2^11 / 2^10 amplitudes where two layouts are legal
    bool force_xcd = false;       // variant 10: trajectory-per-XCD placement of the chained tiles forced
    int chain_lgt = 9;            // log2(threads per tile workgroup) of explicitly chosen chained variants
    // state-sharded run: where the partner slabs arrive and who moves them (RydProblem.shard_recv / shard_exchange)
    void* const* shard_recv = nullptr;
    int (*shard_exchange)(void*, int, const void*, size_t) = nullptr;
    void* shard_user = nullptr;
    GroupArgs garg{};
    PairArgs parg{};
};

// RydProblem.kernel_variant -> Runtime (include/rydiff.h lists the values)
int decode_variant(const RydProblem* p, Runtime& rt) {
    int v = p->kernel_variant;
    if (v < 0 || v > 16 || v == 5 || v == 6) return fail(RYDIFF_EINVAL, "kernel_variant must be 0..4 or 7..16");
    rt.generic_direct = v == 9;
    if (v == 9) v = 1;
    rt.force_three = v == 7 ? 1 : (v == 11 ? 2 : 0);
    rt.plain_tile_order = v == 12;
    rt.tile_mode = v == 13 ? 12 : (v == 14 ? 13 : (v == 15 ? 11 : (v == 16 ? 10 : 0)));
    if (v == 7 || v == 11 || v >= 12) v = 0;
    rt.force_xcd = v == 10;
    if (v == 10) v = 0;
    rt.variant = v;
    rt.chain_lgt = v == 3 ? 8 : (v == 4 ? 10 : 9);
    rt.shard_recv = p->shard_recv;
    rt.shard_exchange = p->shard_exchange;
    rt.shard_user = p->shard_user;
    return RYDIFF_OK;
}

// Tile size and layout count of the chained passes of one direction (forward / adjoint chains are independent: what they share is
// the complete vectors, which are in plain amplitude order).  Measured on MI355X (profiles/r03_wide_tiles.txt): 2^13-amplitude tiles
// (k_chain_wide) win the forward and the adjoint passes at 21-24 qubits (two layouts up to 24: runs of 512 / 256 / 128 / 64 bytes);
// the adjoint WITH signed sums (drive phase gradients) works in register quarters there (in halves it spilled 37 VGPRs and lost at
// 21 and 24 qubits).  Explicit chained variants (2..4, 7, 10, 11) keep the 2^12 tiles they were written for.  (`bwd` is kept in the
// signature: the two directions choose independently, split-diagonal tables exist per tile size.)
ChainGeom chain_geom(const Runtime& rt, bool bwd) {
    (void)bwd;
    const int N = rt.pl.NL;
    int lt = kTileBits;
    if (rt.pl.ga.flagged) lt = kTileBits;  // conditioned flips: sibling pairs must stay inside a tile (even lo and hs)
    else if (rt.tile_mode == 13) lt = N > kWideTileBits ? kWideTileBits : kTileBits;
    else if (rt.tile_mode == 10 || rt.tile_mode == 11) lt = (N > rt.tile_mode && N <= 2 * rt.tile_mode - 2) ? rt.tile_mode : kTileBits;  // two layouts, runs >= 64 bytes
    else if (rt.small_tiles) lt = 11;
    else if (rt.tile_mode == 0 && rt.variant == 0 && !rt.force_three && !rt.force_xcd && ((N >= 21 && N <= 24) || N >= 29))
        lt = kWideTileBits;  // (29, 30 qubits: three layouts of wide tiles keep runs of 512 / 256 bytes in the third; 2^12 tiles end at 28)
    ChainGeom g{lt, 2};
    if (lt == kWideTileBits) g.layouts = N <= 24 ? 2 : 3;
    else if (rt.force_three == 2 && N <= 24) g.layouts = 2;
    else if (N >= 23 || (rt.force_three == 1 && N >= 21)) g.layouts = 3;
    return g;
}

// metadata words -> device through kernel arguments (k_upload): asynchronous, the host buffer may die on return
int upload_words(hipStream_t stream, void* dst, const void* src, size_t bytes) {
    const size_t nwords = (bytes + 7) / 8;  // every destination region is 256-byte aligned and padded (plan.hpp: take)
    const unsigned char* sp = static_cast<const unsigned char*>(src);
    unsigned long long* dp = static_cast<unsigned long long*>(dst);
    for (size_t w0 = 0; w0 < nwords; w0 += kUploadWords) {
        const int n = int(std::min<size_t>(kUploadWords, nwords - w0));
        UploadChunk c;
        const size_t have = std::min<size_t>(size_t(n) * 8, bytes - w0 * 8);
        memcpy(c.w, sp + w0 * 8, have);
        if (have < size_t(n) * 8) memset(reinterpret_cast<unsigned char*>(c.w) + have, 0, size_t(n) * 8 - have);
        hipLaunchKernelGGL(k_upload, dim3(1), dim3(256), 0, stream, dp + w0, c, n);
        LAUNCH_CHECK();
    }
    return RYDIFF_OK;
}

std::mutex g_poly_mutex;
std::vector<PolyDesign> g_poly_cache;

PolyDesign cached_design(double rho, double tol) {
    std::lock_guard<std::mutex> lk(g_poly_mutex);
    for (const auto& d : g_poly_cache)
        if (d.rho == rho && d.tol == tol) return d;
    PolyDesign d = design_polynomial(rho, tol);
    if (g_poly_cache.size() > 64) g_poly_cache.clear();
    g_poly_cache.push_back(d);
    return d;
}

void fill_group_args(const Plan& pl, GroupArgs& g) {
    g.ga = pl.ga.n;
    g.gd = pl.gd.n;
    for (int q = 0; q < pl.ga.n; ++q) g.amask[q] = pl.ga.amp_index_mask[q];
    for (int q = 0; q < pl.gd.n; ++q) {
        g.dmask[q] = pl.gd.amp_index_mask[q];
        g.dcnt[q] = pl.gd.count[q];
    }
    g.cond = pl.ga.flagged;
}

// half width of the generator's numerical range (same widening as finish_runtime)
double generator_half_width(const Plan& pl, double lo, double hi) { return std::max(0.5 * (hi - lo), 1e-9) + pl.pair_radius; }

// apply spectral bounds: sub-steps, design rho, polynomial, factor counts
int finish_runtime(Runtime& rt, double lo, double hi) {
    Plan& pl = rt.pl;
    if (!(hi >= lo) || !std::isfinite(hi) || !std::isfinite(lo)) return fail(RYDIFF_EINVAL, "non-finite spectral bounds (NaN/Inf in the coefficient tables?)");
    lo -= pl.pair_radius;  // dense two-qubit (dissipator) terms: keep the whole numerical range inside the design interval
    hi += pl.pair_radius;
    rt.sigma = 0.5 * (hi + lo);
    rt.width = std::max(0.5 * (hi - lo), 1e-9);
    double rho_d = 1e-6;
    for (auto& s : pl.stages) {
        const double rho = s.tau * rt.width;
        s.nsub = std::max(1, int(std::ceil(rho / kRhoCap)));
        rho_d = std::max(rho_d, rho / s.nsub);
    }
    // quantise rho upward a little so that optimisation epochs with slowly drifting tables reuse the cached design
    const double q = std::pow(2.0, std::ceil(std::log2(rho_d) * 16.0) / 16.0);
    rt.rho_design = q;
    rt.poly = cached_design(rt.rho_design, pl.tol);
    if (rt.poly.degree < 1 || rt.poly.roots.empty()) return fail(RYDIFF_EINVAL, "polynomial design failed");
    rt.total_factors = 0;
    rt.max_step_factors = 0;
    for (int k = 0; k < pl.T; ++k) {
        int f = 0;
        for (int e = pl.step_begin[k]; e < pl.step_begin[k + 1]; ++e) f += pl.stages[e].nsub * rt.poly.degree;
        rt.total_factors += f;
        rt.max_step_factors = std::max(rt.max_step_factors, f);
    }
    fill_group_args(pl, rt.garg);
    if (pl.ga.flagged) {
        // conditioned flips (three-level registers): the one-launch kernels up to 12 qubits (their tile IS the register), beyond
        // that the generic one-amplitude-per-thread kernels (never the unrolled global-drive ones) while few tiles are in flight and
        // the chained passes on 2^12-amplitude tiles (sibling pairs stay inside a tile: chain_geom) beyond
        rt.generic_direct = true;
    }
    rt.parg.n = pl.n_pair;
    for (int t = 0; t < pl.n_pair; ++t) {
        rt.parg.ma[t] = pl.pair_ma[t];
        rt.parg.mb[t] = pl.pair_mb[t];
        unsigned dm = 0;  // relative flips present in the block or its conjugate transpose (PairArgs.dl)
        for (int w = 0; w < 2; ++w)
            for (int own = 0; own < 4; ++own)
                for (int s = 0; s < 4; ++s) {
                    const double* e = pl.pair_tab.data() + size_t(t) * 64 + size_t(w) * 32 + size_t(own * 4 + s) * 2;
                    if (e[0] != 0.0 || e[1] != 0.0) dm |= 1u << (own ^ s);
                }
        rt.parg.dl[t] = uint8_t(dm);
    }
    return RYDIFF_OK;
}

int run_stats(const RydProblem* p, const Plan& pl, void* scratch, hipStream_t stream, double& lo, double& hi, int& flags) {
    StatsArgs sa{};
    sa.amp = static_cast<const double2*>(p->amp_tables);
    sa.det = p->det_tables;
    sa.u_pairs = p->u_pairs;
    sa.n_samples = pl.n_samples;
    sa.Ka = pl.Ka;
    sa.Kd = pl.Kd;
    sa.n_pairs = pl.N * (pl.N - 1) / 2;
    sa.Bc = pl.Bc;
    sa.ga = pl.ga.n;
    sa.gd = pl.gd.n;
    for (int g = 0; g < pl.ga.n; ++g) {
        sa.amem[g] = pl.ga.members[g];
        sa.acnt[g] = pl.ga.count[g];
    }
    for (int g = 0; g < pl.gd.n; ++g) {
        sa.dmem[g] = pl.gd.members[g];
        sa.dcnt[g] = pl.gd.nq[g];
    }
    sa.dones = pl.gd.flagged;
    HIP_TRY(hipMemsetAsync(scratch, 0, 8 * sizeof(double), stream));
    const int ns = std::max(pl.n_samples, 1);
    dim3 grid((ns + 127) / 128, pl.Bc);
    hipLaunchKernelGGL(k_table_stats, grid, dim3(128), 0, stream, static_cast<unsigned long long*>(scratch), sa);
    LAUNCH_CHECK();
    double host[6] = {0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(host, scratch, sizeof(host), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));  // the ONE synchronisation of the library: the bounds decide how many launches follow
    // interpolation weights: KRYLOV_SE uses convex combinations (sum |w| = 1); keep the general bound
    double wsum = 1.0;
    for (const auto& s : pl.stages) {
        double a = 0.0;
        for (int q = 0; q < 4; ++q) a += std::fabs(s.w[q]);
        wsum = std::max(wsum, a);
    }
    // Gershgorin: the diagonal ranges over [-sum of negative U_ij, +sum of positive U_ij] (occupations are 0/1) plus the
    // detuning range; the flip part has norm host[0] exactly (commuting single-qubit terms)
    hi = host[3] + wsum * (host[1] + host[0]);
    lo = -host[5] - wsum * (host[2] + host[0]);
    flags = (host[4] != 0.0) ? 1 : 0;  // bit 0: some flip coefficient has a non-zero imaginary part (phase != 0)
    return RYDIFF_OK;
}

void fill_info(const Runtime& rt, double lo, double hi, size_t ws, RydPlanInfo* info) {
    info->spectral_lo = lo;
    info->spectral_hi = hi;
    info->rho_design = rt.rho_design;
    info->degree = rt.poly.degree;
    info->n_stages = int(rt.pl.stages.size());
    info->max_step_factors = rt.max_step_factors;
    info->flags = rt.flags;
    info->total_factors = rt.total_factors;
    info->workspace_bytes = ws;
}

// A tile pass keeps one CU busy for ~10 us per tile whatever the register size, so with few tiles in flight (one 13..18-qubit
// trajectory: 2..64 tiles on 256 CUs) the one-amplitude-per-thread kernels, which spread over the whole chip, are faster.
// Measured crossover (tools/time_small.py, bench.py --workload c4 --batch b, variants 0 / 1): forward-only runs up to 2^18
// amplitudes in flight (N=13: 5.3 vs 9.4 us per pass, N=16 B=4: +13 %), and the same with gradients since the direct kernels
// keep the full tape too and have unrolled instantiations for one global drive (N=18, 200 steps: 50 ms vs 69 ms chained; N=19:
// 83 vs 78 ms).  Explicit kernel variants are left alone (A/B tests).
// Around 2^19 amplitudes in flight (one 19-qubit trajectory, 2 x 18, 4 x 17, 8 x 16 ...) tiles of 2^11 amplitudes give 256 tiles — one
// per CU — where 2^12 tiles fill half of the chip and the direct kernels move every partner through the fabric: forward pass
// 10.8-11.1 -> 9.0 us, fwd+grad +10 ... +20 % (profiles/r03_small_tiles.txt).  Forward-only runs: the whole range (2^18, 2^19]; with
// gradients from 7 * 2^16 amplitudes and 14 qubits on (below, the direct adjoint stays ahead).  Automatic choice only.
bool small_tiles_win(const Runtime& rt, bool with_gradients) {
    const Plan& pl = rt.pl;
    if (rt.variant != 0 || rt.force_three || rt.force_xcd || rt.tile_mode != 0 || pl.shard_bits || pl.n_pair || pl.ga.flagged) return false;
    if (pl.N < 13 || pl.N > 19) return false;
    const size_t amps = size_t(pl.B) << pl.N;
    if (amps > (size_t(1) << 19)) return false;
    return with_gradients ? (amps >= (size_t(7) << 16) && pl.N >= 14) : amps > (size_t(1) << 18);
}

bool few_tiles(const Runtime& rt, bool with_gradients) {
    const Plan& pl = rt.pl;
    if (rt.small_tiles) return false;
    if (rt.variant != 0 || rt.force_three || rt.force_xcd || (rt.tile_mode != 0 && rt.tile_mode != kTileBits) || pl.shard_bits) return false;
    // forward only: crossover at 2^18 amplitudes in flight (N = 19: 12.5 us direct vs 10.8 us chained per pass).  With gradients the
    // direct ADJOINT pass (own tape element only, partner reads of the cotangent served by L2) stays ahead of the chained one up to
    // 2^19 (11.6-12.8 vs 14.0-14.7 us) and the pair of passes wins by 1-7 % there (profiles/r02_crossover_direct_vs_chained.txt);
    // at 2^20 (C3, C4's 16 x 2^16) the chained tiles win both passes.
    return (size_t(pl.B) << pl.N) <= (size_t(1) << (with_gradients ? 19 : 18));
}

// The full per-factor tape (the adjoint sweep recomputes nothing) goes with the launch-per-factor ADJOINT kernels, chained or
// direct (12 qubits: the one-launch forward sweep writes it); up to 11 qubits the adjoint sweep is one launch too and keeps
// one state per tsave.
// ... and the one-launch adjoint sweeps (<= 11 qubits), which in tape mode walk the factors without recomputing anything.
bool full_tape_possible(const Plan& pl) {
    if (pl.shard_bits) return false;
    if (pl.N <= kPersistBwdMaxQubits) return pl.ga.n <= kPersistGroups && pl.gd.n <= kPersistGroups;
    return pl.n_pair == 0;
}

// PARTIAL tape (need_tape = 3, RydProblem.tape_steps = K): region A = one state per tsave (T + 1 entries, as tape mode 1), region B = the
// intermediate factor outputs (every factor output that is not a step's last) of the LAST K tsave intervals, in run order.  The adjoint
// sweep recomputes the factor inputs of the earlier intervals only.  What the full tape is to a run that fits in HBM, this is to the
// part of a run that fits.  Launch-per-factor sweeps only (13 qubits and up; no pair terms, not sharded).
bool partial_tape_possible(const Runtime& rt) {
    return full_tape_possible(rt.pl) && rt.pl.N > kTileBits;
}

int64_t step_factor_count(const Runtime& rt, int k) {
    int64_t f = 0;
    for (int e = rt.pl.step_begin[k]; e < rt.pl.step_begin[k + 1]; ++e) f += int64_t(rt.pl.stages[e].nsub) * rt.poly.degree;
    return f;
}

struct TapeMap {
    int k0 = 0;                     // first tsave interval whose intermediate factor outputs are on the tape
    std::vector<int64_t> bprefix;   // [T + 1]: region-B entries before interval k (0 up to k0)
    int64_t entries = 0;            // region A + region B
};

TapeMap partial_tape_map(const Runtime& rt, int tape_steps) {
    const Plan& pl = rt.pl;
    TapeMap m;
    m.k0 = std::max(0, pl.T - std::max(tape_steps, 0));
    m.bprefix.assign(pl.T + 1, 0);
    for (int k = 0; k < pl.T; ++k) m.bprefix[k + 1] = m.bprefix[k] + (k >= m.k0 ? std::max<int64_t>(step_factor_count(rt, k) - 1, 0) : 0);
    m.entries = int64_t(pl.T + 1) + m.bprefix[pl.T];
    return m;
}

// common prologue of forward / backward: plan, (optional) stats, carve, upload metadata, expand coefficients, udiag.
// With `info` given nothing in here waits for the device.
int prepare(const RydProblem* p, const RydPlanInfo* info, void* workspace, size_t workspace_bytes, int need_tape,
            bool need_backward, hipStream_t stream, Runtime& rt) {
    std::string err;
    if (!p) return fail(RYDIFF_EINVAL, "null problem");
    int rc = decode_variant(p, rt);
    if (rc) return rc;
    if (!build_plan(p, rt.pl, err)) return fail(err.find("not implemented") != std::string::npos ? RYDIFF_ENOTIMPL : RYDIFF_EINVAL, err);
    if (!workspace) return fail(RYDIFF_EINVAL, "null workspace");
    double lo, hi;
    if (info) {
        lo = info->spectral_lo;
        hi = info->spectral_hi;
        rt.flags = info->flags;
    } else {
        if (workspace_bytes < RYDIFF_PLAN_SCRATCH_BYTES) return fail(RYDIFF_EWORKSPACE, "workspace too small");
        rc = run_stats(p, rt.pl, workspace, stream, lo, hi, rt.flags);
        if (rc) return rc;
    }
    rt.real_amp_grad = p->real_amp_grad != 0;
    // the stage list of the continuous solver depends on the spectral width: rebuild it now that the width is known
    if (!build_plan(p, rt.pl, err, generator_half_width(rt.pl, lo, hi))) return fail(RYDIFF_EINVAL, err);
    rc = finish_runtime(rt, lo, hi);
    if (rc) return rc;
    Plan& pl = rt.pl;
    if (pl.shard_bits && pl.n_pair)
        return fail(RYDIFF_ENOTIMPL, "state-sharded runs do not take dense pair terms");
    if (pl.shard_bits) rt.generic_direct = true;  // (the unrolled direct kernels know nothing about rank qubits)
    if (need_tape == 2 && !full_tape_possible(pl)) need_tape = 1;  // full tape only with chained passes
    if (need_tape == 3 && (!partial_tape_possible(rt) || p->tape_steps < 1)) need_tape = 1;
    rt.small_tiles = small_tiles_win(rt, need_backward || need_tape != 0);
    rt.prefer_direct = few_tiles(rt, need_backward || need_tape != 0);
    const size_t need = carve(pl, need_tape, need_backward, std::max(rt.max_step_factors - 1, 1), rt.total_factors,
                              need_tape == 3 ? partial_tape_map(rt, p->tape_steps).entries : 0);
    if (workspace_bytes < need)
        return fail(RYDIFF_EWORKSPACE, "workspace too small: need " + std::to_string(need) + " bytes, got " + std::to_string(workspace_bytes));
    char* ws = static_cast<char*>(workspace);
    const size_t E = pl.stages.size();
    {   // per-exponential records -> device (as kernel arguments: no copy engine, no synchronisation)
        std::vector<StageDev> sd(E);
        for (size_t e = 0; e < E; ++e) sd[e] = {pl.stages[e].w[0], pl.stages[e].w[1], pl.stages[e].idx[0], pl.stages[e].idx[1]};
        rc = upload_words(stream, ws + pl.off_meta_idx, sd.data(), E * sizeof(StageDev));
        if (rc) return rc;
    }
    if (pl.n_pair) {
        rc = upload_words(stream, ws + pl.off_pair, pl.pair_tab.data(), pl.pair_tab.size() * sizeof(double));
        if (rc) return rc;
        rt.parg.tab = reinterpret_cast<const double2*>(ws + pl.off_pair);
    }
    if (pl.NC > 0) {
        ExpandArgs ea{};
        ea.amp = static_cast<const double2*>(p->amp_tables);
        ea.det = p->det_tables;
        ea.st = reinterpret_cast<const StageDev*>(ws + pl.off_meta_idx);
        ea.coef = reinterpret_cast<double*>(ws + pl.off_coef);
        ea.E = int(E);
        ea.n_samples = pl.n_samples;
        ea.Ka = pl.Ka;
        ea.Kd = pl.Kd;
        ea.NC = pl.NC;
        ea.ga = pl.ga.n;
        ea.gd = pl.gd.n;
        for (int g = 0; g < pl.ga.n; ++g) ea.amem[g] = pl.ga.members[g];
        for (int g = 0; g < pl.gd.n; ++g) ea.dmem[g] = pl.gd.members[g];
        dim3 grid((unsigned(E) + 127) / 128, pl.Bc);
        hipLaunchKernelGGL(k_expand_coeffs, grid, dim3(128), 0, stream, ea);
        LAUNCH_CHECK();
    }
    double* udiag = reinterpret_cast<double*>(ws + pl.off_udiag);
    if (pl.N > 1) {
        if (pl.shard_bits)  // one table per slab, evaluated at the global index
            hipLaunchKernelGGL(k_build_udiag, dim3((pl.dim + 255) / 256, pl.B), dim3(256), 0, stream, udiag, p->u_pairs, pl.N, uint32_t(pl.dim),
                               pl.NL, pl.rank_first);
        else
            hipLaunchKernelGGL(k_build_udiag, dim3((pl.dim + 255) / 256), dim3(256), 0, stream, udiag, p->u_pairs, pl.N, uint32_t(pl.dim));
        LAUNCH_CHECK();
    } else {
        HIP_TRY(hipMemsetAsync(udiag, 0, pl.dim * sizeof(double), stream));
    }
    if (pl.NL > kTileBits && pl.NL <= 30) {  // split diagonal for the tile layouts of the chained passes
        // (sharded runs: the layouts of the NL slab qubits, rows for every tile of the WHOLE register — rank bits on top)
        // one table set per tile size in use (the forward and the adjoint chains choose theirs independently: chain_geom)
        bool built[4] = {false, false, false, false};
        for (int bwd = 0; bwd <= (need_backward ? 1 : 0); ++bwd) {
            const ChainGeom g = chain_geom(rt, bwd != 0);
            if (built[g.lt - kSmallTileBits]) continue;
            built[g.lt - kSmallTileBits] = true;
            const unsigned tiles = unsigned((size_t(1) << pl.N) >> g.lt);
            const size_t tile_amps = size_t(1) << g.lt;
            double* split = split_tables(pl, ws, g.lt);
            const size_t per_layout = tile_amps + size_t(tiles) * 16;
            for (int l = 0; l < g.layouts; ++l) {
                const LayoutDesc d = chain_layout(pl.NL, l, g);
                double* utt = split + l * per_layout;
                hipLaunchKernelGGL(k_build_split, dim3(unsigned((tile_amps + tiles + 255) / 256)), dim3(256), 0, stream, utt, utt + tile_amps,
                                   p->u_pairs, pl.N, d.lo, d.hs, d.hb, tiles, g.lt);
                LAUNCH_CHECK();
            }
        }
    }
    return RYDIFF_OK;
}

struct FactorScalars {
    double gr, gi, br, bi;
};

// state-sharded run with partner ranks elsewhere: tell the caller which slab the partners need next (phase 0, right after the
// launch that produced it) and when the received slabs are about to be read (phase 1); see RydProblem.shard_exchange
int shard_signal(const Runtime& rt, int phase, const void* src) {
    if (!rt.pl.shard_bits || rt.pl.shard_self) return RYDIFF_OK;
    if (rt.shard_exchange(rt.shard_user, phase, src, rt.pl.dim * sizeof(double2)) != 0)
        return fail(RYDIFF_EHIP, phase == 0 ? "shard_exchange failed to post the slab exchange" : "shard_exchange failed to wait for the partner slabs");
    return RYDIFF_OK;
}

// scalars of factor f of one sub-exponential of duration tau_sub
FactorScalars factor_scalars(const Runtime& rt, double tau_sub, int f) {
    using cd = std::complex<double>;
    const cd z = rt.poly.roots[f];
    // p(x) ~ exp(-i*rho_d*x) with x = tau_sub*(H - sigma)/rho_d, spectrum of x inside [-1,1] because
    // tau_sub*width <= rho_d.  One factor: (1 - x/z) = [1 + tau_sub*sigma/(rho_d z)] - [tau_sub/(rho_d z)] H
    const cd denom = rt.rho_design * z;
    cd beta = -tau_sub / denom;
    cd gamma = cd(1.0, 0.0) + tau_sub * rt.sigma / denom;
    if (f == rt.poly.degree - 1) {
        const cd kappa = std::exp(cd(0.0, -tau_sub * rt.sigma)) * rt.poly.p0;
        beta *= kappa;
        gamma *= kappa;
    }
    return {gamma.real(), gamma.imag(), beta.real(), beta.imag()};
}

struct ChainItem {
    int stage;
    FactorScalars s;
};

void build_step_chain(const Runtime& rt, int k, std::vector<ChainItem>& chain) {
    chain.clear();
    const Plan& pl = rt.pl;
    for (int e = pl.step_begin[k]; e < pl.step_begin[k + 1]; ++e) {
        const Stage& st = pl.stages[e];
        const double tau_sub = st.tau / st.nsub;
        for (int s = 0; s < st.nsub; ++s)
            for (int f = 0; f < rt.poly.degree; ++f) chain.push_back({e, factor_scalars(rt, tau_sub, f)});
    }
}

// one global drive on a 12..20-qubit register without pair terms: the unrolled direct kernels (k_factor_direct_global)
bool direct_global_ok(const Runtime& rt) {
    const Plan& pl = rt.pl;
    return !rt.generic_direct && !pl.shard_bits && pl.N >= 12 && pl.N <= 20 && pl.n_pair == 0 && pl.ga.n == 1 &&
           pl.ga.amp_index_mask[0] == (1u << pl.N) - 1u;
}

// state-sharded runs: flip group behind every rank bit (index bit NL + k), -1 if that qubit is not driven
void shard_groups(const Plan& pl, int (&grp)[kShardMaxBits]) {
    for (int k = 0; k < kShardMaxBits; ++k) {
        grp[k] = -1;
        if (k >= pl.shard_bits) continue;
        for (int g = 0; g < pl.ga.n; ++g)
            if (pl.ga.amp_index_mask[g] >> (pl.NL + k) & 1u) grp[k] = g;
    }
}

// obs / expect_slot: fuse <y|O|y> into this launch where the kernel can (returns *fused = true then)
int launch_factor(const Runtime& rt, char* ws, const double2* xin, double2* xout, int stage, const FactorScalars& s, hipStream_t stream,
                  const double* obs = nullptr, double* expect_slot = nullptr, bool* fused = nullptr) {
    const Plan& pl = rt.pl;
    if (fused) *fused = false;
    FactorArgs fa{};
    fa.xin = xin;
    fa.xout = xout;
    fa.udiag = reinterpret_cast<const double*>(ws + pl.off_udiag);
    fa.coef = reinterpret_cast<const double*>(ws + pl.off_coef) + size_t(stage) * pl.NC;
    fa.coef_bstride = pl.Bc > 1 ? long(pl.stages.size()) * pl.NC : 0;
    fa.dim = uint32_t(pl.dim);
    fa.gr = s.gr;
    fa.gi = s.gi;
    fa.br = s.br;
    fa.bi = s.bi;
    fa.g = rt.garg;
    fa.pair = rt.parg;
    if (pl.shard_bits) {
        fa.sh_bits = pl.shard_bits;
        fa.sh_nl = pl.NL;
        fa.sh_rank_first = pl.rank_first;
        fa.sh_self = pl.shard_self ? 1 : 0;
        for (int k = 0; k < pl.shard_bits; ++k) fa.sh_rem[k] = pl.shard_self ? nullptr : static_cast<const double2*>(rt.shard_recv[k]);
        shard_groups(pl, fa.sh_grp);
        for (int q = 0; q < fa.g.ga; ++q) fa.g.amask[q] &= uint32_t(pl.dim - 1);  // in-slab flips only; the rank bits are the partner slabs
    }
    dim3 grid(unsigned((pl.dim + 255) / 256), pl.B);
    if (direct_global_ok(rt)) {
        if (obs && expect_slot) {
            fa.obs = obs;
            fa.expect_slot = expect_slot;
            fa.n_obs = pl.n_obs;
            fa.exp_ostride = long(pl.T + 1) * pl.B;
            if (fused) *fused = true;
        }
        const dim3 grid8(grid.x * 8, grid.y);  // ONEXCD instantiations: 8x oversubscribed grid
        switch (pl.N) {
#define RYDIFF_CASE1(NQ) case NQ: hipLaunchKernelGGL((k_factor_direct_global<NQ, true>), grid8, dim3(256), 0, stream, fa); break;
#define RYDIFF_CASE(NQ) case NQ: hipLaunchKernelGGL((k_factor_direct_global<NQ, false>), grid, dim3(256), 0, stream, fa); break;
            RYDIFF_CASE1(12) RYDIFF_CASE1(13)  // one XCD has the CUs for <= 32 workgroups; beyond, spreading wins (measured)
            RYDIFF_CASE(14) RYDIFF_CASE(15) RYDIFF_CASE(16) RYDIFF_CASE(17) RYDIFF_CASE(18) RYDIFF_CASE(19) RYDIFF_CASE(20)
#undef RYDIFF_CASE
#undef RYDIFF_CASE1
        }
    } else {
        hipLaunchKernelGGL(k_factor_direct, grid, dim3(256), 0, stream, fa);
    }
    LAUNCH_CHECK();
    return RYDIFF_OK;
}

// ---- chained tile passes: launch schedule (layouts are defined next to the kernel includes) ------------------------
// One launch of a chain: which layout, which index bits the partial already covers, which factor's partial it extends
// (`fin`, -1: none; `completes`: the factor is complete afterwards) and which factor it starts (`sta`, -1: none).
struct KernelStep {
    int layout;
    uint32_t covered;
    int fin;
    bool completes;
    int sta;
};

void chain_schedule(int N, const ChainGeom& geom, int F, std::vector<KernelStep>& ks) {
    ks.clear();
    if (geom.layouts == 2) {
        for (int k = 0; k <= F; ++k)
            ks.push_back({k & 1, k > 0 ? chain_layout(N, (k - 1) & 1, geom).bits : 0u, k - 1, true, k < F ? k : -1});
        return;
    }
    const uint32_t bbits = chain_layout(N, 1, geom).bits;
    auto end_layout = [](int m) { return (m & 1) ? 2 : 0; };  // factor m starts in A (even m) or C (odd m)
    for (int m = 0; m <= F; ++m) {
        const uint32_t cov = m > 0 ? (chain_layout(N, end_layout(m - 1), geom).bits | bbits) : 0u;
        ks.push_back({end_layout(m), cov, m - 1, true, m < F ? m : -1});
        if (m < F) ks.push_back({1, chain_layout(N, end_layout(m), geom).bits, m, false, -1});
    }
}

uint32_t to_tile_mask(const LayoutDesc& d, int lt, uint32_t index_mask) {
    uint32_t m = 0;
    for (int b = 0; b < lt; ++b) {
        const int gb = b < d.lo ? b : d.hs + (b - d.lo);
        if (index_mask >> gb & 1u) m |= 1u << b;
    }
    return m;
}

bool chain_enabled(const Runtime& rt) {
    const int N = rt.pl.NL;
    if (rt.variant == 1 || rt.pl.n_pair) return false;  // pair terms: direct kernels
    // conditioned flips (three-level registers): measured (tools/time_three_level.py) the chained passes win up to 20 qubits (10 atoms:
    // 29.6 -> 20.6 us per pass, fwd+grad +23 %); beyond, the 2^12 tiles' short runs / third layout and the 512-thread signed-sum adjoint
    // lose to the generic direct kernels (22 qubits: 758 vs 687 steps/s fwd+grad).  Explicit chained variants still take them (tests).
    if (rt.pl.ga.flagged && N > 20 && rt.variant == 0) return false;
    if (rt.pl.shard_bits) return N > kTileBits && chain_geom(rt, false).layouts == 2;  // sharded: two-layout chains on the slab qubits (<= 22; wide tiles: <= 24)
    return N > kTileBits && N <= (chain_geom(rt, false).lt == kWideTileBits ? 30 : 28) && !rt.prefer_direct;
}

struct ChainStep {
    // kernel j finishes factor `fin` (if has_p) and starts factor `sta` (if has_q)
    const double2* u;
    const double2* p;
    double2* v_out;
    double2* q_out;
    int fin_stage, sta_stage;
    FactorScalars fin, sta;
    int has_p, has_q, write_v;
    bool completes = true;  // the finish stage yields the complete vector (false: middle pass of a three-layout chain)
    int layout, prev_layout;
    uint32_t covered_bits = 0;  // index bits whose flips the incoming partial already contains
    // backward mode
    bool bwd = false;
    const double2* x_fin = nullptr;
    const double2* x_sta = nullptr;
    double cb_fin_r = 0, cb_fin_i = 0, cb_sta_r = 0, cb_sta_i = 0;
    double* wtot = nullptr;
    // fused cotangent injection (adjoint): save point the vector completed by this launch belongs to, -1: none
    int inject_k = -1;
    // fused expectation (forward)
    const double* obs = nullptr;
    double* expect_slot = nullptr;
    int n_obs = 0;
    long exp_ostride = 0;
};

template <int LT, int LGT, bool CPLX, bool BWD, bool FAST = false, bool RES = false>
int launch_chain_t(const ChainArgs& ca, unsigned tiles, hipStream_t stream) {
    static_assert(LT == kTileBits || (LT >= kSmallTileBits && LT <= kWideTileBits && LGT == 10 && !RES), "other tile sizes: 1024 threads, no L2-resident placement");
    if constexpr (!FAST) {  // one global drive, at most one detuning group: the loop-free instantiation
        if (ca.ga == 1 && ca.sta_mask[0] == (1u << LT) - 1u && !ca.cond)  // (any number of detuning groups)
            return launch_chain_t<LT, LGT, CPLX, BWD, true, RES>(ca, tiles, stream);
    }
    // tile + reduction scratch: one double per wave (forward), [4 ga + gd] slots per wave (adjoint: parked gradient partials)
    const size_t nw = (size_t(1) << LGT) / 64;
    const size_t max_lds = (size_t(1) << LT) * sizeof(double2) + 256 + (BWD ? size_t(5) * kMaxGroups * nw * sizeof(double) : 0);
    const size_t lds = (size_t(1) << LT) * sizeof(double2) + 256 + (BWD ? size_t(4 * ca.ga + ca.gd) * nw * sizeof(double) : 0);
    void (*kern)(ChainArgs);
    if constexpr (LT == kWideTileBits) {  // register halves — quarters for the adjoint with signed sums (k_chain<13, ...> would spill)
        kern = k_chain_wide<LT, CPLX, BWD, FAST, (BWD && CPLX) ? 2 : 4>;
    } else {
        kern = k_chain<LT, LGT, CPLX, BWD, FAST, RES>;
    }
    // once per instantiation and process; idempotent, so a race between two first callers is harmless
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load(std::memory_order_acquire)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(max_lds)));
        attr_set.store(true, std::memory_order_release);
    }
    const dim3 grid = ca.xcd_place ? dim3(tiles * 8u, unsigned(ca.b_count + 7) / 8u) : dim3(tiles, unsigned(ca.b_count));
    hipLaunchKernelGGL(kern, grid, dim3(1 << LGT), lds, stream, ca);
    LAUNCH_CHECK();
    return RYDIFF_OK;
}

template <int LT, int LGT, bool RES = false>
int launch_chain_l(const ChainArgs& ca, unsigned tiles, bool cplx, bool bwd, hipStream_t stream) {
    // the adjoint needs both partner sums (plain and signed) unless the coefficients are real AND the caller only uses the
    // real part of the amplitude gradients (RydProblem.real_amp_grad): then `cplx` arrives false here
    if (bwd) return cplx ? launch_chain_t<LT, LGT, true, true, false, RES>(ca, tiles, stream) : launch_chain_t<LT, LGT, false, true, false, RES>(ca, tiles, stream);
    return cplx ? launch_chain_t<LT, LGT, true, false, false, RES>(ca, tiles, stream) : launch_chain_t<LT, LGT, false, false, false, RES>(ca, tiles, stream);
}

// cotangents handed to the backward call (fused injection, see ChainArgs / FactorBwdArgs)
struct InjectSource {
    const double2* gstate = nullptr;  // grad_states [n_tsave][B][dim] or nullptr
    const double* gexp = nullptr;     // grad_expect [n_obs][n_tsave][B] or nullptr
    const double* obs = nullptr;      // [n_obs][dim]
    int n_obs = 0;
    bool any() const { return gstate || gexp; }
};

// the trajectories a chain covers and how its launches are placed (Runtime::xcd_group)
struct BatchSlice {
    int first = 0, count = 0;
    bool xcd = false;  // trajectory-per-XCD placement + L2-resident in-place vectors
};

int launch_chain(const Runtime& rt, char* ws, const ChainStep& cs, const BatchSlice& bs, const InjectSource& inj, hipStream_t stream) {
    const Plan& pl = rt.pl;
    const ChainGeom geom = chain_geom(rt, cs.bwd);
    const LayoutDesc X = chain_layout(pl.NL, cs.layout, geom);
    ChainArgs ca{};
    ca.u = cs.u;
    ca.p = cs.p ? cs.p : cs.u;  // (always loadable: the kernel requests u, p and the tape vectors outside of control flow)
    ca.v_out = cs.v_out;
    ca.q_out = cs.q_out;
    {
        const size_t tile_amps = size_t(1) << geom.lt;
        const size_t per_layout = tile_amps + size_t((size_t(1) << pl.N) >> geom.lt) * 16;
        const double* split = split_tables(pl, ws, geom.lt) + size_t(cs.layout) * per_layout;
        ca.utt = split;
        ca.vr = split + tile_amps;
    }
    const double* coef = reinterpret_cast<const double*>(ws + pl.off_coef);
    ca.coef_fin = coef + size_t(std::max(cs.fin_stage, 0)) * pl.NC;
    ca.coef_sta = coef + size_t(std::max(cs.sta_stage, 0)) * pl.NC;
    ca.coef_bstride = pl.Bc > 1 ? long(pl.stages.size()) * pl.NC : 0;
    ca.fb_r = cs.fin.br;
    ca.fb_i = cs.fin.bi;
    ca.fg_r = cs.fin.gr;
    ca.fg_i = cs.fin.gi;
    ca.completes = (cs.has_p && cs.completes) ? 1 : 0;
    ca.sg_r = cs.sta.gr;
    ca.sg_i = cs.sta.gi;
    ca.sb_r = cs.sta.br;
    ca.sb_i = cs.sta.bi;
    ca.lo = X.lo;
    ca.hs = X.hs;
    ca.hb = X.hb;
    ca.dim = uint32_t(pl.dim);
    ca.has_p = cs.has_p;
    ca.has_q = cs.has_q;
    ca.write_v = cs.write_v;
    ca.ga = pl.ga.n;
    ca.gd = pl.gd.n;
    ca.cond = pl.ga.flagged;
    ca.xcd_place = bs.xcd ? 1 : 0;
    ca.resident = bs.xcd ? 1 : 0;
    ca.b_first = bs.first;
    ca.b_count = bs.count;
    const uint32_t prev_bits = cs.covered_bits;
    for (int g = 0; g < pl.ga.n; ++g) {
        ca.fin_mask[g] = to_tile_mask(X, geom.lt, pl.ga.amp_index_mask[g] & ~prev_bits);
        ca.sta_mask[g] = to_tile_mask(X, geom.lt, pl.ga.amp_index_mask[g]);
    }
    for (int g = 0; g < pl.gd.n; ++g) {
        ca.dmask[g] = pl.gd.amp_index_mask[g];
        ca.dcnt[g] = pl.gd.count[g];
    }
    ca.obs = cs.obs;
    ca.expect_slot = cs.expect_slot;
    ca.n_obs = cs.n_obs;
    ca.exp_ostride = cs.exp_ostride;
    ca.obs_bstride = pl.shard_bits ? long(pl.dim) : 0;
    ca.obs_ostride = pl.shard_bits ? long(pl.B) * long(pl.dim) : long(pl.dim);
    if (pl.shard_bits) {
        ca.sh_bits = pl.shard_bits;
        ca.sh_nl = pl.NL;
        ca.sh_rank_first = pl.rank_first;
        ca.sh_self = pl.shard_self ? 1 : 0;
        for (int k = 0; k < pl.shard_bits; ++k) ca.sh_rem[k] = pl.shard_self ? nullptr : static_cast<const double2*>(rt.shard_recv[k]);
        shard_groups(pl, ca.sh_grp);
    }
    if (cs.bwd) {
        double* ge = reinterpret_cast<double*>(ws + pl.off_ge);
        const long ge_rec = long(kGradReplicas) * (pl.NC + 1);
        ca.x_fin = cs.x_fin ? cs.x_fin : cs.u;
        ca.x_sta = cs.x_sta ? cs.x_sta : cs.u;
        ca.ge_fin = ge + size_t(std::max(cs.fin_stage, 0)) * ge_rec;
        ca.ge_sta = ge + size_t(std::max(cs.sta_stage, 0)) * ge_rec;
        ca.ge_bstride = pl.Bc > 1 ? long(pl.stages.size()) * ge_rec : 0;
        ca.ge_rstride = pl.NC + 1;
        ca.cb_fin_r = cs.cb_fin_r;
        ca.cb_fin_i = cs.cb_fin_i;
        ca.cb_sta_r = cs.cb_sta_r;
        ca.cb_sta_i = cs.cb_sta_i;
        ca.wtot = cs.wtot;
        if (cs.inject_k >= 0 && cs.has_p && inj.any()) {
            const size_t sv = size_t(pl.B) * pl.dim;
            ca.inj_gstate = inj.gstate ? inj.gstate + size_t(cs.inject_k) * sv : nullptr;
            ca.inj_gexp = inj.gexp ? inj.gexp + size_t(cs.inject_k) * pl.B : nullptr;
            ca.inj_obs = inj.obs;
            ca.inj_n_obs = inj.n_obs;
            ca.inj_ostride = long(pl.T + 1) * pl.B;
        }
    }
    const unsigned tiles = unsigned(pl.dim >> geom.lt);
    if (X.lo < 3 && !bs.xcd && !rt.plain_tile_order && tiles % (8u << (3 - X.lo)) == 0) ca.tile_swz = 3 - X.lo;
    const bool cplx = (rt.flags & 1) != 0 || (cs.bwd && !rt.real_amp_grad);
    // auto: 1024 threads per tile for the forward passes, 512 for the (register-hungrier) adjoint passes
    // (the real-drive adjoint, without the signed sums, fits 1024 threads too: measured 2710 -> 2767 steps/s on C3)
    const int lgt = rt.variant == 0 ? ((cs.bwd && cplx) ? 9 : 10) : rt.chain_lgt;
    if (geom.lt == kWideTileBits) return launch_chain_l<kWideTileBits, 10>(ca, tiles, cplx, cs.bwd, stream);
    if (geom.lt == 11) return launch_chain_l<11, 10>(ca, tiles, cplx, cs.bwd, stream);
    if (geom.lt == 10) return launch_chain_l<10, 10>(ca, tiles, cplx, cs.bwd, stream);
    if (bs.xcd)  // L2-resident placement (automatic thread counts only: variant 10 decodes to 0)
        return lgt == 9 ? launch_chain_l<kTileBits, 9, true>(ca, tiles, cplx, cs.bwd, stream) : launch_chain_l<kTileBits, 10, true>(ca, tiles, cplx, cs.bwd, stream);
    switch (lgt) {
        case 8: return launch_chain_l<kTileBits, 8>(ca, tiles, cplx, cs.bwd, stream);
        case 10: return launch_chain_l<kTileBits, 10>(ca, tiles, cplx, cs.bwd, stream);
        default: return launch_chain_l<kTileBits, 9>(ca, tiles, cplx, cs.bwd, stream);
    }
}

// Run `items` (factors, in order) as a chain starting from the complete vector `start`.
//   dst(i)   : where the complete output of factor i (0-based) goes, or nullptr to skip storing it (only legal for the last)
//   on_done(i, ptr): called after the launch that completed factor i
// skip_last_finish: do not finish the last factor (its output is not needed) — used by the backward recompute.
template <class DstFn, class DoneFn, class ExpFn>
int run_chain(const Runtime& rt, char* ws, const std::vector<ChainItem>& items, const double2* start, DstFn dst, DoneFn on_done,
              ExpFn exp_slot, bool skip_last_finish, const BatchSlice& bs, hipStream_t stream) {
    const Plan& pl = rt.pl;
    double2* pp[2] = {reinterpret_cast<double2*>(ws + pl.off_pp0), reinterpret_cast<double2*>(ws + pl.off_pp1)};
    const int F = int(items.size()) - (skip_last_finish ? 1 : 0);  // the last factor is not even started then
    if (F <= 0) return RYDIFF_OK;
    std::vector<KernelStep> ks;
    chain_schedule(pl.NL, chain_geom(rt, false), F, ks);
    const double2* cur = start;
    int rcx = shard_signal(rt, 0, cur);  // partners need the chain's start vector for the first completing launch
    if (rcx) return rcx;
    const InjectSource no_inj{};
    // L2-resident placement: partials are rewritten IN PLACE (a workgroup reads and writes only its own tile elements), so
    // the live set of a trajectory is one complete vector + one partial
    auto ppsel = [&](size_t k) { return bs.xcd ? pp[0] : pp[k & 1]; };
    for (size_t k = 0; k < ks.size(); ++k) {
        const KernelStep& st = ks[k];
        ChainStep cs{};
        cs.layout = st.layout;
        cs.prev_layout = k > 0 ? ks[k - 1].layout : -1;
        cs.covered_bits = st.covered;
        cs.u = cur;
        cs.has_p = st.fin >= 0;
        cs.has_q = st.sta >= 0;
        cs.p = cs.has_p ? ppsel(k - 1) : nullptr;
        cs.q_out = cs.has_q ? ppsel(k) : nullptr;
        cs.write_v = cs.has_p;
        cs.completes = st.completes;
        if (cs.has_p) {
            cs.v_out = st.completes ? dst(st.fin) : ppsel(k);  // a middle pass hands the extended partial on
            cs.fin_stage = items[st.fin].stage;
            cs.fin = items[st.fin].s;
            if (!cs.v_out) return fail(RYDIFF_EINVAL, "internal: chain destination missing");
            if (st.completes) exp_slot(st.fin, cs);
        } else {
            cs.fin_stage = -1;
        }
        cs.sta_stage = cs.has_q ? items[st.sta].stage : -1;
        if (cs.has_q) cs.sta = items[st.sta].s;
        int rc = cs.has_p ? shard_signal(rt, 1, nullptr) : RYDIFF_OK;  // this launch reads the partners' copies of `cur`
        if (rc) return rc;
        rc = launch_chain(rt, ws, cs, bs, no_inj, stream);
        if (rc) return rc;
        if (cs.has_p && st.completes) {
            cur = cs.v_out;
            if (k + 1 < ks.size()) {  // the next launch completes the next factor from the partners' copies of this vector
                rc = shard_signal(rt, 0, cur);
                if (rc) return rc;
            }
            rc = on_done(st.fin, cs.v_out);
            if (rc) return rc;
        }
    }
    return RYDIFF_OK;
}

// Adjoint sweep of consecutive tsave intervals as ONE chain.  `items` are the forward factors (in forward order), xs[i] the
// input of factor i, `lam_in` the cotangent w.r.t. the output of the last one; the cotangent w.r.t. the first one's input ends
// up in lam_bufs[cl].  save_k[i] >= 0: the input of factor i is the state at save point save_k[i] — the launch that completes
// the cotangent there also adds the cotangent injected at that save point (fused).  on_stage_end(stage, lam, x_out) is called
// with the complete cotangent at every exponential's output.
template <class StageEndFn>
int run_chain_bwd(const Runtime& rt, char* ws, const std::vector<ChainItem>& items, const std::vector<const double2*>& xs,
                  const std::vector<int>& save_k, const double2* lam_in, double2* lam_bufs[2], int& cl, double* wtot,
                  StageEndFn on_stage_end, const BatchSlice& bs, const InjectSource& inj, hipStream_t stream) {
    const Plan& pl = rt.pl;
    double2* pp[2] = {reinterpret_cast<double2*>(ws + pl.off_pp0), reinterpret_cast<double2*>(ws + pl.off_pp1)};
    const int M = int(items.size());
    std::vector<KernelStep> ks;
    chain_schedule(pl.NL, chain_geom(rt, true), M, ks);
    const double2* cur = lam_in;
    int rcx = shard_signal(rt, 0, cur);  // sharded: the partners need the incoming cotangent for the first completing launch
    if (rcx) return rcx;
    auto ppsel = [&](size_t k) { return bs.xcd ? pp[0] : pp[k & 1]; };
    // adjoint factor index a = 0..M-1 corresponds to forward factor f = M-1-a
    for (size_t k = 0; k < ks.size(); ++k) {
        const KernelStep& st = ks[k];
        ChainStep cs{};
        cs.bwd = true;
        cs.layout = st.layout;
        cs.prev_layout = k > 0 ? ks[k - 1].layout : -1;
        cs.covered_bits = st.covered;
        cs.u = cur;
        cs.has_p = st.fin >= 0;
        cs.has_q = st.sta >= 0;
        cs.p = cs.has_p ? ppsel(k - 1) : nullptr;
        cs.q_out = cs.has_q ? ppsel(k) : nullptr;
        cs.write_v = cs.has_p;
        cs.completes = st.completes;
        cs.wtot = wtot;
        if (cs.has_p) {
            const int f = M - 1 - st.fin;  // forward factor whose adjoint this launch extends / completes
            const ChainItem& it = items[f];
            cs.fin_stage = it.stage;
            cs.fin = {it.s.gr, -it.s.gi, it.s.br, -it.s.bi};
            cs.cb_fin_r = it.s.br;
            cs.cb_fin_i = it.s.bi;
            cs.x_fin = xs[f];
            if (st.completes) {
                if (!bs.xcd) cl ^= 1;  // L2-resident placement rewrites the cotangent in place
                cs.v_out = lam_bufs[cl];
                if (!bs.xcd && cs.v_out == cur) return fail(RYDIFF_EINVAL, "internal: cotangent ping-pong clash");
                cs.inject_k = save_k[f];
            } else {
                cs.v_out = ppsel(k);
            }
        }
        if (cs.has_q) {
            const int f = M - 1 - st.sta;
            const ChainItem& it = items[f];
            cs.sta_stage = it.stage;
            cs.sta = {it.s.gr, -it.s.gi, it.s.br, -it.s.bi};
            cs.cb_sta_r = it.s.br;
            cs.cb_sta_i = it.s.bi;
            cs.x_sta = xs[f];
            // the cotangent `cur` at the output of the chain's last factor: exponential boundary for dL/dtau
            if (st.sta == 0) {
                int rc = on_stage_end(it.stage, cur, xs[M]);
                if (rc) return rc;
            }
        }
        int rc = (cs.has_p && st.completes) ? shard_signal(rt, 1, nullptr) : RYDIFF_OK;  // this launch reads the partners' copies of `cur`
        if (rc) return rc;
        rc = launch_chain(rt, ws, cs, bs, inj, stream);
        if (rc) return rc;
        if (cs.has_p && st.completes) {
            cur = cs.v_out;  // complete cotangent at the INPUT of forward factor f = output of forward factor f-1
            if (k + 1 < ks.size()) {  // the next completing launch needs the partners' copies of this cotangent
                rc = shard_signal(rt, 0, cur);
                if (rc) return rc;
            }
            const int f = M - 1 - st.fin;
            if (f >= 1 && items[f].stage != items[f - 1].stage) {
                rc = on_stage_end(items[f - 1].stage, cur, xs[f]);
                if (rc) return rc;
            }
        }
    }
    return RYDIFF_OK;
}

// ---- persistent small-N forward (k_persist) ------------------------------------------------------------------------
bool persist_enabled(const Runtime& rt) { return rt.variant != 1 && rt.pl.N <= kTileBits && !rt.pl.shard_bits; }

// every amplitude and every detuning group is ONE qubit and there are more than two of either (stochastic-noise runs, several local
// channels): the per-bit form of the forward sweep (k_persist<..., PERBIT>) instead of the generic group loops
bool per_bit_terms(const PersistArgs& pa) {
    auto single = [](uint32_t m) { return m != 0 && (m & (m - 1)) == 0; };
    if (pa.pair.n || pa.cond || (pa.ga <= 2 && pa.gd <= 2)) return false;
    uint32_t seen = 0;
    for (int g = 0; g < pa.ga; ++g) {
        if (!single(pa.amask[g]) || (seen & pa.amask[g])) return false;
        seen |= pa.amask[g];
    }
    seen = 0;
    for (int g = 0; g < pa.gd; ++g) {
        if (!single(pa.dmask[g]) || (seen & pa.dmask[g]) || pa.dcnt[g] != 1) return false;
        seen |= pa.dmask[g];
    }
    return true;
}

template <int LT, bool CPLX>
int launch_persist_t(const PersistArgs& pa, int B, hipStream_t stream) {
    constexpr int LGT = LT < 10 ? LT : 10;
    const dim3 block(LGT < 6 ? 64 : (1 << LGT));
    if (per_bit_terms(pa) && pa.NC <= 3 * LT)
        hipLaunchKernelGGL((k_persist<LT, LGT, CPLX, false, false, kPersistGroups, true>), dim3(B), block, 0, stream, pa);
    else if (pa.ga == 1 && pa.gd <= 1 && pa.amask[0] == (1u << LT) - 1u && pa.pair.n == 0 && pa.cond == 0)
        hipLaunchKernelGGL((k_persist<LT, LGT, CPLX, true, true>), dim3(B), block, 0, stream, pa);
    else if (pa.ga <= 2 && pa.gd <= 2)
        hipLaunchKernelGGL((k_persist<LT, LGT, CPLX, true, false, 2>), dim3(B), block, 0, stream, pa);
    else if (pa.ga <= kPersistGroups && pa.gd <= kPersistGroups)
        hipLaunchKernelGGL((k_persist<LT, LGT, CPLX, true>), dim3(B), block, 0, stream, pa);
    else
        hipLaunchKernelGGL((k_persist<LT, LGT, CPLX, false>), dim3(B), block, 0, stream, pa);
    LAUNCH_CHECK();
    return RYDIFF_OK;
}

template <int LT, bool CPLX>
int launch_lanes_fwd_t(const PersistArgs& pa, int B, hipStream_t stream) {
    if (pa.ga == 1 && pa.gd <= 1 && pa.amask[0] == (1u << LT) - 1u && pa.pair.n == 0 && pa.cond == 0)
        hipLaunchKernelGGL((k_lanes_fwd<LT, CPLX, true>), dim3(B), dim3(64), 0, stream, pa);
    else if (pa.ga <= 2 && pa.gd <= 2)
        hipLaunchKernelGGL((k_lanes_fwd<LT, CPLX, false, 2>), dim3(B), dim3(64), 0, stream, pa);
    else
        hipLaunchKernelGGL((k_lanes_fwd<LT, CPLX, false>), dim3(B), dim3(64), 0, stream, pa);
    LAUNCH_CHECK();
    return RYDIFF_OK;
}

// one amplitude per lane of one wave (lane_kernels.hpp); variant 8 keeps the LDS-tile kernels for A/B tests
bool lanes_enabled(int variant, int N, int ga, int gd, int n_pair) {
    return variant != 8 && N <= kLaneMaxQubits && n_pair <= kLanePairMax && ga <= kPersistGroups && gd <= kPersistGroups;
}

template <bool CPLX>
int launch_persist(int variant, int N, const PersistArgs& pa, int B, hipStream_t stream) {
    if (lanes_enabled(variant, N, pa.ga, pa.gd, pa.pair.n) && pa.n_factors > 0) {  // (up to 4 groups: also ahead of the per-bit sweep, 0.83 vs 0.90 us per factor at 4 qubits)
        switch (N) {
            case 1: return launch_lanes_fwd_t<1, CPLX>(pa, B, stream);
            case 2: return launch_lanes_fwd_t<2, CPLX>(pa, B, stream);
            case 3: return launch_lanes_fwd_t<3, CPLX>(pa, B, stream);
            case 4: return launch_lanes_fwd_t<4, CPLX>(pa, B, stream);
            case 5: return launch_lanes_fwd_t<5, CPLX>(pa, B, stream);
            default: return launch_lanes_fwd_t<6, CPLX>(pa, B, stream);
        }
    }
    switch (N) {
        case 1: return launch_persist_t<1, CPLX>(pa, B, stream);
        case 2: return launch_persist_t<2, CPLX>(pa, B, stream);
        case 3: return launch_persist_t<3, CPLX>(pa, B, stream);
        case 4: return launch_persist_t<4, CPLX>(pa, B, stream);
        case 5: return launch_persist_t<5, CPLX>(pa, B, stream);
        case 6: return launch_persist_t<6, CPLX>(pa, B, stream);
        case 7: return launch_persist_t<7, CPLX>(pa, B, stream);
        case 8: return launch_persist_t<8, CPLX>(pa, B, stream);
        case 9: return launch_persist_t<9, CPLX>(pa, B, stream);
        case 10: return launch_persist_t<10, CPLX>(pa, B, stream);
        case 11: return launch_persist_t<11, CPLX>(pa, B, stream);
        default: return launch_persist_t<12, CPLX>(pa, B, stream);
    }
}

template <int LT, bool CPLX>
int launch_persist_bwd_t(const PersistBwdArgs& pa, int B, hipStream_t stream) {
    constexpr int LGT = LT < 10 ? LT : 9;  // 1024+ amplitudes: 512 threads, so the accumulators stay in registers
    if (pa.ga == 1 && pa.gd <= 1 && pa.amask[0] == (1u << LT) - 1u && pa.pair.n == 0 && pa.cond == 0)
        hipLaunchKernelGGL((k_persist_bwd<LT, LGT, CPLX, true>), dim3(B), dim3(LGT < 6 ? 64 : (1 << LGT)), 0, stream, pa);
    else if (pa.ga <= 2 && pa.gd <= 2)
        hipLaunchKernelGGL((k_persist_bwd<LT, LGT, CPLX, false, 2>), dim3(B), dim3(LGT < 6 ? 64 : (1 << LGT)), 0, stream, pa);
    else
        hipLaunchKernelGGL((k_persist_bwd<LT, LGT, CPLX>), dim3(B), dim3(LGT < 6 ? 64 : (1 << LGT)), 0, stream, pa);
    LAUNCH_CHECK();
    return RYDIFF_OK;
}

template <int LT, bool CPLX>
int launch_lanes_bwd_t(const PersistBwdArgs& pa, int B, hipStream_t stream) {
    if (pa.tape_full) {  // every factor input is on the tape: one descending walk, nothing recomputed
        if (pa.ga == 1 && pa.gd <= 1 && pa.amask[0] == (1u << LT) - 1u && pa.pair.n == 0 && pa.cond == 0)
            hipLaunchKernelGGL((k_lanes_bwd_tape<LT, CPLX, true>), dim3(B), dim3(64), 0, stream, pa);
        else if (pa.ga <= 2 && pa.gd <= 2)
            hipLaunchKernelGGL((k_lanes_bwd_tape<LT, CPLX, false, 2>), dim3(B), dim3(64), 0, stream, pa);
        else
            hipLaunchKernelGGL((k_lanes_bwd_tape<LT, CPLX, false>), dim3(B), dim3(64), 0, stream, pa);
        LAUNCH_CHECK();
        return RYDIFF_OK;
    }
    if (pa.ga == 1 && pa.gd <= 1 && pa.amask[0] == (1u << LT) - 1u && pa.pair.n == 0 && pa.cond == 0)
        hipLaunchKernelGGL((k_lanes_bwd<LT, CPLX, true>), dim3(B), dim3(64), 0, stream, pa);
    else if (pa.ga <= 2 && pa.gd <= 2)
        hipLaunchKernelGGL((k_lanes_bwd<LT, CPLX, false, 2>), dim3(B), dim3(64), 0, stream, pa);
    else
        hipLaunchKernelGGL((k_lanes_bwd<LT, CPLX, false>), dim3(B), dim3(64), 0, stream, pa);
    LAUNCH_CHECK();
    return RYDIFF_OK;
}

template <bool CPLX>
int launch_persist_bwd(int variant, int N, const PersistBwdArgs& pa, int B, hipStream_t stream) {
    if (lanes_enabled(variant, N, pa.ga, pa.gd, pa.pair.n) && pa.n_factors > 0) {
        switch (N) {
            case 1: return launch_lanes_bwd_t<1, CPLX>(pa, B, stream);
            case 2: return launch_lanes_bwd_t<2, CPLX>(pa, B, stream);
            case 3: return launch_lanes_bwd_t<3, CPLX>(pa, B, stream);
            case 4: return launch_lanes_bwd_t<4, CPLX>(pa, B, stream);
            case 5: return launch_lanes_bwd_t<5, CPLX>(pa, B, stream);
            default: return launch_lanes_bwd_t<6, CPLX>(pa, B, stream);
        }
    }
    switch (N) {
        case 1: return launch_persist_bwd_t<1, CPLX>(pa, B, stream);
        case 2: return launch_persist_bwd_t<2, CPLX>(pa, B, stream);
        case 3: return launch_persist_bwd_t<3, CPLX>(pa, B, stream);
        case 4: return launch_persist_bwd_t<4, CPLX>(pa, B, stream);
        case 5: return launch_persist_bwd_t<5, CPLX>(pa, B, stream);
        case 6: return launch_persist_bwd_t<6, CPLX>(pa, B, stream);
        case 7: return launch_persist_bwd_t<7, CPLX>(pa, B, stream);
        case 8: return launch_persist_bwd_t<8, CPLX>(pa, B, stream);
        case 9: return launch_persist_bwd_t<9, CPLX>(pa, B, stream);
        case 10: return launch_persist_bwd_t<10, CPLX>(pa, B, stream);
        default: return launch_persist_bwd_t<11, CPLX>(pa, B, stream);
    }
}

// Factor table of the one-launch sweeps: every factor of the run, in order, with the save point its output belongs to (0 = none).
// Built ON THE DEVICE (k_build_ptable) from four small per-interval / per-exponential arrays that travel as kernel arguments.
int build_persist_table_device(const Runtime& rt, char* ws, hipStream_t stream, int* n_factors) {
    const Plan& pl = rt.pl;
    const size_t E = pl.stages.size();
    if (rt.poly.degree > kMaxDegreeDev) return fail(RYDIFF_ENOTIMPL, "polynomial degree beyond the on-device factor table builder");
    if (size_t(rt.total_factors) * sizeof(PersistFactor) > pl.ptable_bytes) return fail(RYDIFF_EWORKSPACE, "internal: factor table does not fit");
    std::vector<int32_t> begin(pl.step_begin.begin(), pl.step_begin.end()), first(pl.T + 1, 0), nsub(E);
    std::vector<double> tau(E);
    for (size_t e = 0; e < E; ++e) {
        nsub[e] = pl.stages[e].nsub;
        tau[e] = pl.stages[e].tau / pl.stages[e].nsub;
    }
    for (int k = 0; k < pl.T; ++k) {
        int64_t f = 0;
        for (int e = pl.step_begin[k]; e < pl.step_begin[k + 1]; ++e) f += int64_t(pl.stages[e].nsub) * rt.poly.degree;
        first[k + 1] = int32_t(first[k] + f);
    }
    int rc = upload_words(stream, ws + pl.off_pm_begin, begin.data(), begin.size() * sizeof(int32_t));
    if (!rc) rc = upload_words(stream, ws + pl.off_pm_first, first.data(), first.size() * sizeof(int32_t));
    if (!rc) rc = upload_words(stream, ws + pl.off_pm_tau, tau.data(), tau.size() * sizeof(double));
    if (!rc) rc = upload_words(stream, ws + pl.off_pm_nsub, nsub.data(), nsub.size() * sizeof(int32_t));
    if (rc) return rc;
    PTableArgs ta{};
    ta.out = reinterpret_cast<PersistFactor*>(ws + pl.off_ptable);
    ta.tau_sub = reinterpret_cast<const double*>(ws + pl.off_pm_tau);
    ta.nsub = reinterpret_cast<const int32_t*>(ws + pl.off_pm_nsub);
    ta.step_begin = reinterpret_cast<const int32_t*>(ws + pl.off_pm_begin);
    ta.step_first = reinterpret_cast<const int32_t*>(ws + pl.off_pm_first);
    ta.T = pl.T;
    ta.degree = rt.poly.degree;
    ta.sigma = rt.sigma;
    ta.rho_design = rt.rho_design;
    ta.p0r = rt.poly.p0.real();
    ta.p0i = rt.poly.p0.imag();
    for (int f = 0; f < rt.poly.degree; ++f) {
        ta.roots[2 * f] = rt.poly.roots[f].real();
        ta.roots[2 * f + 1] = rt.poly.roots[f].imag();
    }
    hipLaunchKernelGGL(k_build_ptable, dim3(unsigned(pl.T + 63) / 64), dim3(64), 0, stream, ta);
    LAUNCH_CHECK();
    *n_factors = int(rt.total_factors);
    return RYDIFF_OK;
}

}  // namespace



namespace {

// Trajectory-per-XCD placement of the chained tile passes ("Batches of L2-sized trajectories"): a launch
// covers a GROUP of 8 m trajectories, trajectory -> XCD by workgroup id % 8, vectors rewritten in place with plain loads and
// stores, so that the complete vector + partial of m trajectories (m * 32 * 2^N bytes) stay in each XCD's 4 MiB L2 from pass to
// pass and only the write-back crosses the fabric.  Every group runs its WHOLE sweep before the next one starts.  Returns the
// group size (0: off).  Placement changes speed only: results are the same as with the plain grid (A/B-tested).
int xcd_group_size(const Runtime& rt, bool adjoint) {
    const Plan& pl = rt.pl;
    if (!chain_enabled(rt) || chain_geom(rt, adjoint).layouts != 2 || chain_geom(rt, adjoint).lt != kTileBits || pl.shard_bits) return 0;
    const size_t live = size_t(32) << pl.N;    // complete vector + partial of one trajectory
    const size_t budget = size_t(3) << 20;     // of the 4 MiB L2 (the rest: tape lines on their way out, tables)
    const int m = int(std::max<size_t>(1, budget / live));
    if (rt.force_xcd) return 8 * m;
    // Measured (profiles/r02_xcd_placement.txt): a launch of <= 256 tiles is bound by the ~10 us one tile keeps its CU busy, not by
    // the fabric, so SEVERAL groups in sequence lose to one launch over the whole batch (16 qubits x 32: 38 vs 27 us per pass).
    // Where ONE group covers the batch the forward passes gain (16 qubits x 8: 14.2 -> 9.8 us; 13 qubits x 64: 12.1 -> 9.6 us);
    // the adjoint passes, which stream two tape vectors anyway, do not (17.1 vs 17.3 us).
    if (rt.variant != 0 || adjoint || pl.B < 8 || pl.B > 8 * m || live > budget) return 0;
    return 8 * m;
}

// RydPlanInfo.kernel_fwd / kernel_bwd: the instantiations launch_chain / launch_factor / the one-launch sweeps will pick for this
// problem (same tests as there), spelled the way rocprofv3 prints them.  Reporting only.
void describe_kernels(const Runtime& rt, const RydProblem* p, bool backward, RydPlanInfo* info) {
    const Plan& pl = rt.pl;
    auto b = [](bool v) { return v ? "true" : "false"; };
    info->kernel_fwd[0] = info->kernel_bwd[0] = 0;
    if (info->kernel_family == 3) {
        const bool fast = pl.ga.n == 1 && (pl.ga.amp_index_mask[0] & ((1u << pl.NL) - 1u)) == (1u << pl.NL) - 1u && !pl.ga.flagged;
        for (int bwd = 0; bwd <= (backward ? 1 : 0); ++bwd) {
            const bool cplx = (rt.flags & 1) != 0 || (bwd && !p->real_amp_grad);
            const int lgt = rt.variant == 0 ? ((bwd && cplx) ? 9 : 10) : rt.chain_lgt;
            const bool res = xcd_group_size(rt, bwd != 0) > 0;
            if (chain_geom(rt, bwd != 0).lt == kWideTileBits)
                std::snprintf(bwd ? info->kernel_bwd : info->kernel_fwd, sizeof(info->kernel_fwd), "k_chain_wide<%d,%s,%s,%s>", kWideTileBits,
                              b(cplx), b(bwd != 0), b(fast));
            else
                std::snprintf(bwd ? info->kernel_bwd : info->kernel_fwd, sizeof(info->kernel_fwd), "k_chain<%d,%d,%s,%s,%s,%s>", chain_geom(rt, bwd != 0).lt,
                              chain_geom(rt, bwd != 0).lt == kTileBits ? lgt : 10, b(cplx), b(bwd != 0), b(fast), b(res));
        }
    } else if (info->kernel_family == 2) {
        if (direct_global_ok(rt)) {
            std::snprintf(info->kernel_fwd, sizeof(info->kernel_fwd), "k_factor_direct_global<%d,%s>", pl.N, b(pl.N <= 13));
            if (backward) std::snprintf(info->kernel_bwd, sizeof(info->kernel_bwd), "k_factor_bwd_direct_global<%d,%s>", pl.N, b(pl.N <= 13));
        } else {
            std::snprintf(info->kernel_fwd, sizeof(info->kernel_fwd), "k_factor_direct");
            if (backward) std::snprintf(info->kernel_bwd, sizeof(info->kernel_bwd), "k_factor_bwd_direct");
        }
    } else {
        std::snprintf(info->kernel_fwd, sizeof(info->kernel_fwd), info->kernel_family == 0 ? "k_lanes_fwd (N=%d)" : "k_persist<%d,...>", pl.N);
        if (backward) std::snprintf(info->kernel_bwd, sizeof(info->kernel_bwd), info->kernel_family == 0 ? "k_lanes_bwd (N=%d)" : "k_persist_bwd<%d,...>", pl.N);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* rydiff_last_error(void) { return g_last_error.c_str(); }
const char* rydiff_version(void) { return "rydiff 0.4 (gfx950)"; }
size_t rydiff_sizeof_problem(void) { return sizeof(RydProblem); }
size_t rydiff_sizeof_plan_info(void) { return sizeof(RydPlanInfo); }

#ifdef RYDIFF_TIMELINE
int rydiff_debug_timeline(unsigned long long* host_buf, int n_entries) {  // tuning builds only
    return hipMemcpyFromSymbol(host_buf, HIP_SYMBOL(g_timeline), size_t(n_entries) * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif

int rydiff_design_polynomial(double rho, double tol, int max_degree, int* degree, double* roots_reim, double* p0_reim, double* max_err) {
    if (!(rho > 0.0) || !degree || !roots_reim || max_degree < 1) return fail(RYDIFF_EINVAL, "bad arguments");
    PolyDesign d = design_polynomial(rho, tol, max_degree);
    if (d.degree < 1) return fail(RYDIFF_EINVAL, "polynomial design failed");
    *degree = d.degree;
    for (int i = 0; i < d.degree; ++i) {
        roots_reim[2 * i] = d.roots[i].real();
        roots_reim[2 * i + 1] = d.roots[i].imag();
    }
    if (p0_reim) {
        p0_reim[0] = d.p0.real();
        p0_reim[1] = d.p0.imag();
    }
    if (max_err) *max_err = d.max_err;
    return RYDIFF_OK;
}

int rydiff_plan(const RydProblem* p, int need_tape, int need_backward, void* scratch, void* stream_, RydPlanInfo* info) {
    if (!p) return fail(RYDIFF_EINVAL, "null problem");
    if (!info || !scratch) return fail(RYDIFF_EINVAL, "null info or scratch");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Runtime rt;
    std::string err;
    int rc = decode_variant(p, rt);
    if (rc) return rc;
    if (!build_plan(p, rt.pl, err)) return fail(err.find("not implemented") != std::string::npos ? RYDIFF_ENOTIMPL : RYDIFF_EINVAL, err);
    double lo, hi;
    rc = run_stats(p, rt.pl, scratch, stream, lo, hi, rt.flags);
    if (rc) return rc;
    if (!build_plan(p, rt.pl, err, generator_half_width(rt.pl, lo, hi))) return fail(RYDIFF_EINVAL, err);
    rc = finish_runtime(rt, lo, hi);
    if (rc) return rc;
    int tm = need_tape;
    if (tm < 0 || tm > 3) return fail(RYDIFF_EINVAL, "need_tape must be 0..3");
    if (tm == 2 && !full_tape_possible(rt.pl)) tm = 1;
    if (tm == 3 && (!partial_tape_possible(rt) || p->tape_steps < 1)) tm = 1;
    const size_t ws = carve(rt.pl, tm, need_backward != 0, std::max(rt.max_step_factors - 1, 1), rt.total_factors,
                            tm == 3 ? partial_tape_map(rt, p->tape_steps).entries : 0);
    fill_info(rt, lo, hi, ws, info);
    info->tape_mode = tm;
    rt.small_tiles = small_tiles_win(rt, need_backward != 0 || tm != 0);
    rt.prefer_direct = few_tiles(rt, need_backward != 0 || tm != 0);
    if (persist_enabled(rt)) info->kernel_family = lanes_enabled(rt.variant, rt.pl.N, rt.pl.ga.n, rt.pl.gd.n, rt.pl.n_pair) ? 0 : 1;
    else info->kernel_family = chain_enabled(rt) ? 3 : 2;
    describe_kernels(rt, p, need_backward != 0, info);
    return RYDIFF_OK;
}

int rydiff_forward(const RydProblem* p, const RydPlanInfo* info, const void* psi0, void* states_out, double* expect_out,
                   void* workspace, size_t workspace_bytes, int need_tape, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!psi0) return fail(RYDIFF_EINVAL, "null psi0");
    Runtime rt;
    // need_tape = 2 together with states_out: the caller wants the stored states AND a gradient later — where the full tape is
    // granted the factor outputs go to the workspace tape and the states at the save points are copied out of it
    int rc = prepare(p, info, workspace, workspace_bytes, need_tape >= 2 ? need_tape : (states_out ? 0 : need_tape), false, stream, rt);
    if (rc) return rc;
    const Plan& pl = rt.pl;
    char* ws = static_cast<char*>(workspace);
    const size_t sv = size_t(pl.B) * pl.dim;  // complex elements per saved state
    double2* buf[2] = {reinterpret_cast<double2*>(ws + pl.off_buf0), reinterpret_cast<double2*>(ws + pl.off_buf1)};
    const bool full_ws_tape = pl.tape_mode == 2;  // (prepare downgrades the request where the full tape is not possible)
    const bool partial_tape = pl.tape_mode == 3;  // save-point states + the factor outputs of the trailing intervals, both in the workspace
    const TapeMap tmap = partial_tape ? partial_tape_map(rt, p->tape_steps) : TapeMap{};
    double2* sout = static_cast<double2*>(states_out);
    // final_state_only: no per-step states; the state at the last evaluation time is copied to states_out at the end
    double2* final_dst = nullptr;
    if (p->final_state_only && states_out) {
        if (need_tape || persist_enabled(rt))
            return fail(RYDIFF_EINVAL, "final_state_only needs the launch-per-factor kernels (more than 12 qubits or a sharded run) and no tape");
        final_dst = sout;
        sout = nullptr;
        states_out = nullptr;
    }
    double2* tape = (states_out && !full_ws_tape && !partial_tape) ? sout : (pl.tape_mode ? reinterpret_cast<double2*>(ws + pl.off_tape) : nullptr);
    double2* copy_out = (states_out && (full_ws_tape || partial_tape)) ? sout : nullptr;  // states at the save points, copied from the workspace tape
    double2* tape_b = partial_tape ? tape + size_t(pl.T + 1) * sv : nullptr;  // region B of the partial tape
    const double2* cur = static_cast<const double2*>(psi0);
    if (tape) {
        HIP_TRY(hipMemcpyAsync(tape, psi0, pl.state_bytes, hipMemcpyDeviceToDevice, stream));
        cur = tape;
    }
    if (copy_out) HIP_TRY(hipMemcpyAsync(copy_out, psi0, pl.state_bytes, hipMemcpyDeviceToDevice, stream));
    const double* obs = p->obs_diag;
    const bool want_exp = expect_out && pl.n_obs > 0;
    const unsigned red_blocks = unsigned(std::min<size_t>((pl.dim + 255) / 256, 1024));
    const long obs_bstride = pl.shard_bits ? long(pl.dim) : 0;  // sharded: one observable slab per rank
    if (want_exp) {
        HIP_TRY(hipMemsetAsync(expect_out, 0, size_t(pl.n_obs) * (pl.T + 1) * pl.B * sizeof(double), stream));
        hipLaunchKernelGGL(k_expect_diag, dim3(red_blocks, pl.B), dim3(256), 0, stream, cur, obs, expect_out, pl.n_obs, pl.T + 1, 0, pl.B, uint32_t(pl.dim),
                           obs_bstride);
        LAUNCH_CHECK();
    }
    std::vector<ChainItem> chain;
    if (persist_enabled(rt)) {
        // whole trajectory in one launch, from a factor table built on the device
        int n_factors = 0;
        rc = build_persist_table_device(rt, ws, stream, &n_factors);
        if (rc) return rc;
        PersistArgs pa{};
        pa.psi0 = static_cast<const double2*>(psi0);
        pa.states = full_ws_tape ? copy_out : tape;
        pa.tape_all = full_ws_tape ? tape : nullptr;
        pa.udiag = reinterpret_cast<const double*>(ws + pl.off_udiag);
        pa.coef = reinterpret_cast<const double*>(ws + pl.off_coef);
        pa.coef_bstride = pl.Bc > 1 ? long(pl.stages.size()) * pl.NC : 0;
        pa.NC = pl.NC;
        pa.factors = reinterpret_cast<const PersistFactor*>(ws + pl.off_ptable);
        pa.n_factors = n_factors;
        pa.obs = want_exp ? obs : nullptr;
        pa.expect = expect_out;
        pa.n_obs = want_exp ? pl.n_obs : 0;
        pa.n_tsave = pl.T + 1;
        pa.B = pl.B;
        pa.dim = uint32_t(pl.dim);
        pa.ga = pl.ga.n;
        pa.gd = pl.gd.n;
        pa.pair = rt.parg;
        for (int g = 0; g < pl.ga.n; ++g) pa.amask[g] = pl.ga.amp_index_mask[g];
        pa.cond = pl.ga.flagged;
        for (int g = 0; g < pl.gd.n; ++g) {
            pa.dmask[g] = pl.gd.amp_index_mask[g];
            pa.dcnt[g] = pl.gd.count[g];
        }
        return (rt.flags & 1) ? launch_persist<true>(rt.variant, pl.N, pa, pl.B, stream) : launch_persist<false>(rt.variant, pl.N, pa, pl.B, stream);
    }
    if (chain_enabled(rt)) {
        // one chain over the whole run: factor i of step k; complete outputs at step ends go to the tape
        std::vector<ChainItem> all;
        std::vector<int> step_of_end;  // for factor i: k+1 if it ends step k, else 0
        std::vector<int64_t> b_entry;  // partial tape: region-B entry of factor i's output, -1: not kept
        for (int k = 0; k < pl.T; ++k) {
            build_step_chain(rt, k, chain);
            for (size_t i = 0; i < chain.size(); ++i) {
                all.push_back(chain[i]);
                step_of_end.push_back(i + 1 == chain.size() ? k + 1 : 0);
                if (partial_tape) b_entry.push_back((k >= tmap.k0 && i + 1 < chain.size()) ? tmap.bprefix[k] + int64_t(i) : -1);
            }
        }
        const bool full_tape = full_ws_tape;
        const int grp = xcd_group_size(rt, false);
        for (int b0 = 0; b0 < pl.B; b0 += (grp ? grp : pl.B)) {
            const BatchSlice bs{b0, std::min(grp ? grp : pl.B, pl.B - b0), grp > 0};
            int flip = 0;
            auto dst = [&](int i) -> double2* {
                if (full_tape) return tape + size_t(i + 1) * sv;  // entry g = output of global factor g (entry 0 = psi0)
                if (step_of_end[i] && tape) return tape + size_t(step_of_end[i]) * sv;
                if (partial_tape && b_entry[i] >= 0) return tape_b + size_t(b_entry[i]) * sv;
                if (bs.xcd) return buf[0];  // rewritten in place: the trajectory's lines stay in its XCD's L2
                flip ^= 1;
                return buf[flip];
            };
            auto done = [&](int i, const double2* out) -> int {
                if (final_dst && step_of_end[i] == pl.T)
                    HIP_TRY(hipMemcpyAsync(final_dst + size_t(bs.first) * pl.dim, out + size_t(bs.first) * pl.dim,
                                           size_t(bs.count) * pl.dim * sizeof(double2), hipMemcpyDeviceToDevice, stream));
                if (copy_out && step_of_end[i])
                    HIP_TRY(hipMemcpyAsync(copy_out + size_t(step_of_end[i]) * sv + size_t(bs.first) * pl.dim, out + size_t(bs.first) * pl.dim,
                                           size_t(bs.count) * pl.dim * sizeof(double2), hipMemcpyDeviceToDevice, stream));
                return RYDIFF_OK;
            };
            auto exp_slot = [&](int i, ChainStep& cs) {
                if (want_exp && step_of_end[i]) {  // the pass that completes a step's last factor also reduces <O>
                    cs.obs = obs;
                    cs.n_obs = pl.n_obs;
                    cs.exp_ostride = long(pl.T + 1) * pl.B;
                    cs.expect_slot = expect_out + size_t(step_of_end[i]) * pl.B;
                }
            };
            rc = run_chain(rt, ws, all, cur, dst, done, exp_slot, false, bs, stream);
            if (rc) return rc;
        }
        return RYDIFF_OK;
    }
    int pp = 0;
    const bool full_tape_direct = full_ws_tape && tape;
    bool exp_fused = false;
    size_t gfac = 0;  // global factor index: with the full tape entry g + 1 = output of factor g (entry 0 = psi0)
    for (int k = 0; k < pl.T; ++k) {
        build_step_chain(rt, k, chain);
        for (size_t i = 0; i < chain.size(); ++i, ++gfac) {
            const bool last = (i + 1 == chain.size());
            double2* dst;
            if (full_tape_direct) dst = tape + (gfac + 1) * sv;
            else if (last && tape) dst = tape + size_t(k + 1) * sv;
            else if (partial_tape && k >= tmap.k0) dst = tape_b + size_t(tmap.bprefix[k] + int64_t(i)) * sv;
            else {
                dst = buf[pp];
                if (dst == cur) dst = buf[pp ^ 1];
                pp ^= 1;
            }
            const bool want_here = want_exp && last;  // the launch that completes the step also reduces <O> where it can
            rc = shard_signal(rt, 0, cur);  // sharded: partners need `cur` ...
            if (!rc) rc = shard_signal(rt, 1, nullptr);  // ... and this launch reads theirs
            if (rc) return rc;
            rc = launch_factor(rt, ws, cur, dst, chain[i].stage, chain[i].s, stream, want_here ? obs : nullptr,
                               want_here ? expect_out + size_t(k + 1) * pl.B : nullptr, &exp_fused);
            if (rc) return rc;
            cur = dst;
        }
        if (copy_out) HIP_TRY(hipMemcpyAsync(copy_out + size_t(k + 1) * sv, cur, pl.state_bytes, hipMemcpyDeviceToDevice, stream));
        if (want_exp && !exp_fused) {
            hipLaunchKernelGGL(k_expect_diag, dim3(red_blocks, pl.B), dim3(256), 0, stream, cur, obs, expect_out, pl.n_obs, pl.T + 1, k + 1, pl.B, uint32_t(pl.dim),
                               obs_bstride);
            LAUNCH_CHECK();
        }
    }
    if (final_dst) HIP_TRY(hipMemcpyAsync(final_dst, cur, pl.state_bytes, hipMemcpyDeviceToDevice, stream));
    return RYDIFF_OK;
}

int rydiff_backward(const RydProblem* p, const RydPlanInfo* info, const void* states, const void* grad_states,
                    const double* grad_expect, void* g_amp, double* g_det, double* g_u, double* g_tsave, void* g_psi0,
                    void* workspace, size_t workspace_bytes, int need_tape, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!states && !need_tape) return fail(RYDIFF_EINVAL, "backward needs the trajectory: pass states or use the workspace tape");
    Runtime rt;
    int rc = prepare(p, info, workspace, workspace_bytes, need_tape >= 2 ? need_tape : (states ? 0 : need_tape), true, stream, rt);
    if (rc) return rc;
    const Plan& pl = rt.pl;
    char* ws = static_cast<char*>(workspace);
    const size_t sv = size_t(pl.B) * pl.dim;
    const size_t E = pl.stages.size();
    if (!states && !pl.tape_mode) return fail(RYDIFF_EINVAL, "backward needs the trajectory: pass states or use the workspace tape");
    // the full workspace tape (written by a forward call with need_tape = 2) is preferred over `states`
    const double2* tape = (pl.tape_mode >= 2 || !states) ? reinterpret_cast<const double2*>(ws + pl.off_tape) : static_cast<const double2*>(states);
    double2* lam[2] = {reinterpret_cast<double2*>(ws + pl.off_buf0), reinterpret_cast<double2*>(ws + pl.off_buf1)};
    double2* chainbuf = reinterpret_cast<double2*>(ws + pl.off_chain);
    double* ge = reinterpret_cast<double*>(ws + pl.off_ge);
    double* wtot = g_u ? reinterpret_cast<double*>(ws + pl.off_wtot) : nullptr;
    const double* udiag = reinterpret_cast<const double*>(ws + pl.off_udiag);
    const double* coef = reinterpret_cast<const double*>(ws + pl.off_coef);
    const long coef_bstride = pl.Bc > 1 ? long(E) * pl.NC : 0;
    const long ge_rec = long(kGradReplicas) * (pl.NC + 1);
    const long ge_bstride = pl.Bc > 1 ? long(E) * ge_rec : 0;
    const double2* gst = static_cast<const double2*>(grad_states);
    const double* obs = p->obs_diag;
    const bool have_gexp = grad_expect && pl.n_obs > 0;
    InjectSource inj{};
    inj.gstate = gst;
    inj.gexp = have_gexp ? grad_expect : nullptr;
    inj.obs = obs;
    inj.n_obs = have_gexp ? pl.n_obs : 0;
    // observable table [n_obs][dim]; state-sharded runs: one slab per rank of the call, [n_obs][B][dim]
    const long obs_bstride = pl.shard_bits ? long(pl.dim) : 0;
    const long obs_ostride = pl.shard_bits ? long(pl.B) * long(pl.dim) : long(pl.dim);
    if (pl.shard_bits && g_tsave)
        return fail(RYDIFF_ENOTIMPL, "state-sharded runs: no gradient w.r.t. the evaluation times (pass g_tsave = NULL)");

    HIP_TRY(hipMemsetAsync(ge, 0, size_t(pl.Bc) * E * ge_rec * sizeof(double), stream));
    if (wtot) HIP_TRY(hipMemsetAsync(wtot, 0, pl.dim * (pl.shard_bits ? size_t(pl.B) : 1) * sizeof(double), stream));
    dim3 grid(unsigned((pl.dim + 255) / 256), pl.B);
    int cl = 0;
    // where the state at tsave[k] lives: one entry per tsave, or (full tape) one entry per factor pass
    const bool full_tape = pl.tape_mode == 2;
    const bool partial_tape = pl.tape_mode == 3;
    const TapeMap tmap = partial_tape ? partial_tape_map(rt, p->tape_steps) : TapeMap{};
    const double2* tape_b = partial_tape ? tape + size_t(pl.T + 1) * sv : nullptr;
    auto taped = [&](int k) { return full_tape || (partial_tape && k >= tmap.k0); };  // every factor input of interval k is on the tape
    std::vector<int64_t> fprefix(pl.T + 1, 0);
    for (int k = 0; k < pl.T; ++k) {
        int64_t f = 0;
        for (int e = pl.step_begin[k]; e < pl.step_begin[k + 1]; ++e) f += int64_t(pl.stages[e].nsub) * rt.poly.degree;
        fprefix[k + 1] = fprefix[k] + f;
    }
    auto state_at = [&](int k) -> const double2* { return tape + size_t(full_tape ? fprefix[k] : k) * sv; };

    // small registers: the whole reverse sweep in one launch (k_persist_bwd)
    // (4096 amplitudes would need 8 per thread plus the accumulators: past the register file, so N = 12 keeps the launch-per-factor sweep)
    // (with the full tape both one-launch adjoints — one wave up to 6 qubits, one workgroup up to 11 — walk the tape)
    const bool lanes_tape = full_tape && lanes_enabled(rt.variant, pl.N, pl.ga.n, pl.gd.n, pl.n_pair);
    const bool persisted = persist_enabled(rt) && pl.N <= kPersistBwdMaxQubits && pl.ga.n <= kPersistGroups &&
                           pl.gd.n <= kPersistGroups &&
                           (rt.max_step_factors <= kStageChunk || lanes_tape);
    if (persisted) {
        int n_factors = 0;
        rc = build_persist_table_device(rt, ws, stream, &n_factors);
        if (rc) return rc;
        if (!full_tape && rt.max_step_factors - 1 > pl.chain_slots) return fail(RYDIFF_EWORKSPACE, "internal: chain buffers too small");
        int32_t* dflags = nullptr;
        if (have_gexp) {  // stays on the device: the sweep skips save points without an expectation cotangent
            dflags = reinterpret_cast<int32_t*>(ws + pl.off_meta2);
            hipLaunchKernelGGL(k_cotangent_flags, dim3(unsigned(pl.T + 1 + 255) / 256), dim3(256), 0, stream, grad_expect,
                               pl.n_obs, pl.T + 1, pl.B, dflags);
            LAUNCH_CHECK();
        }
        PersistBwdArgs pa{};
        pa.gflags = dflags;
        pa.tape = tape;
        pa.tape_full = full_tape ? 1 : 0;
        pa.save_entry = reinterpret_cast<const int32_t*>(ws + pl.off_pm_first);
        pa.chainbuf = chainbuf;
        pa.gstate = gst;
        pa.gexp = have_gexp ? grad_expect : nullptr;
        pa.obs = obs;
        pa.udiag = udiag;
        pa.coef = coef;
        pa.coef_bstride = coef_bstride;
        pa.NC = pl.NC;
        pa.factors = reinterpret_cast<const PersistFactor*>(ws + pl.off_ptable);
        pa.n_factors = n_factors;
        pa.ge = ge;
        pa.ge_bstride = ge_bstride;
        pa.ge_sstride = ge_rec;
        pa.wtot = wtot;
        pa.mu_out = lam[cl];
        pa.want_tau = g_tsave ? 1 : 0;
        pa.n_obs = have_gexp ? pl.n_obs : 0;
        pa.n_tsave = pl.T + 1;
        pa.B = pl.B;
        pa.dim = uint32_t(pl.dim);
        pa.ga = pl.ga.n;
        pa.gd = pl.gd.n;
        pa.pair = rt.parg;
        for (int g = 0; g < pl.ga.n; ++g) pa.amask[g] = pl.ga.amp_index_mask[g];
        pa.cond = pl.ga.flagged;
        for (int g = 0; g < pl.gd.n; ++g) {
            pa.dmask[g] = pl.gd.amp_index_mask[g];
            pa.dcnt[g] = pl.gd.count[g];
        }
        rc = (rt.flags & 1) ? launch_persist_bwd<true>(rt.variant, pl.N, pa, pl.B, stream) : launch_persist_bwd<false>(rt.variant, pl.N, pa, pl.B, stream);
        if (rc) return rc;
    } else {
        // cotangent at the final time; the cotangents of the earlier save points are added by the launch that completes the
        // adjoint state there (fused injection: no separate launches, and no host-side look at grad_expect)
        hipLaunchKernelGGL(k_inject, grid, dim3(256), 0, stream, lam[cl], gst ? gst + size_t(pl.T) * sv : nullptr,
                           state_at(pl.T), obs, have_gexp ? grad_expect : nullptr, pl.n_obs, pl.T + 1, pl.T, pl.B,
                           uint32_t(pl.dim), 1, obs_ostride, obs_bstride);
        LAUNCH_CHECK();
    }

    const bool chained = chain_enabled(rt);
    const int grp = (!persisted && chained) ? xcd_group_size(rt, true) : 0;
    const int cl0 = cl;
    auto sweep = [&](const BatchSlice& bs) -> int {
        std::vector<ChainItem> chain, part;
        std::vector<const double2*> xs;
        std::vector<int> save_k;
        cl = cl0;
        const dim3 grid_s(grid.x, unsigned(bs.count));
        auto dot_h = [&](int stage, const double2* g, const double2* xout) -> int {
            if (!g_tsave) return RYDIFF_OK;
            DotHArgs da{};
            da.g = g;
            da.x = xout;
            da.udiag = udiag;
            da.coef = coef + size_t(stage) * pl.NC;
            da.coef_bstride = coef_bstride;
            da.out = ge + size_t(stage) * ge_rec + pl.NC;
            da.out_bstride = ge_bstride;
            da.out_rstride = pl.NC + 1;
            da.dim = uint32_t(pl.dim);
            da.b_first = bs.first;
            da.gr = rt.garg;
            da.pair = rt.parg;
            hipLaunchKernelGGL(k_dot_hx, grid_s, dim3(256), 0, stream, da);
            LAUNCH_CHECK();
            return RYDIFF_OK;
        };
        for (int k = pl.T - 1; k >= 0; --k) {
            // With every factor input on the tape, consecutive intervals run as ONE chain: the launch that finishes the adjoint
            // of interval k's first factor (and adds the cotangent injected at save point k) also starts interval k-1's last one.
            const int k_hi = k;
            if (taped(k_hi) && chained) {
                while (k > 0 && taped(k - 1) && fprefix[k_hi + 1] - fprefix[k - 1] < (int64_t(1) << 16)) --k;
            }
            const bool on_tape = taped(k_hi);  // (then every interval of the merged chain is)
            chain.clear();
            save_k.clear();
            for (int kk = k; kk <= k_hi; ++kk) {
                build_step_chain(rt, kk, part);
                for (size_t i = 0; i < part.size(); ++i) save_k.push_back(i == 0 ? kk : -1);
                chain.insert(chain.end(), part.begin(), part.end());
            }
            const int M = int(chain.size());
            if (!on_tape && M - 1 > pl.chain_slots) return fail(RYDIFF_EWORKSPACE, "internal: chain buffers too small");
            // the factor inputs x_0 .. x_{M-1}: on the tape, or recomputed
            xs.assign(M + 1, nullptr);
            xs[0] = state_at(k);
            if (full_tape) {
                for (int i = 1; i < M; ++i) xs[i] = tape + size_t(fprefix[k] + i) * sv;  // every factor input is on the tape
            } else if (on_tape) {  // partial tape: interval kk's first input is its save-point state, the others sit in region B
                int i = 0;
                for (int kk = k; kk <= k_hi; ++kk) {
                    const int mk = int(fprefix[kk + 1] - fprefix[kk]);
                    for (int j = 0; j < mk; ++j, ++i) xs[i] = j == 0 ? state_at(kk) : tape_b + size_t(tmap.bprefix[kk] + j - 1) * sv;
                }
            } else if (chained && M > 1) {
                for (int i = 1; i < M; ++i) xs[i] = chainbuf + size_t(i - 1) * sv;
                auto dst = [&](int i) -> double2* { return chainbuf + size_t(i) * sv; };
                auto done = [&](int, const double2*) -> int { return RYDIFF_OK; };
                auto no_exp = [&](int, ChainStep&) {};
                int rc2 = run_chain(rt, ws, chain, xs[0], dst, done, no_exp, true, bs, stream);
                if (rc2) return rc2;
            } else {
                for (int i = 1; i < M; ++i) {
                    double2* dst = chainbuf + size_t(i - 1) * sv;
                    int rc2 = shard_signal(rt, 0, xs[i - 1]);  // (sharded recompute: the partners need this factor input, this launch theirs)
                    if (!rc2) rc2 = shard_signal(rt, 1, nullptr);
                    if (rc2) return rc2;
                    rc2 = launch_factor(rt, ws, xs[i - 1], dst, chain[i - 1].stage, chain[i - 1].s, stream);
                    if (rc2) return rc2;
                    xs[i] = dst;
                }
            }
            xs[M] = state_at(k_hi + 1);
            if (chained) {
                int rc2 = run_chain_bwd(rt, ws, chain, xs, save_k, lam[cl], lam, cl, wtot, dot_h, bs, inj, stream);
                if (rc2) return rc2;
                continue;
            }
            for (int i = M; i >= 1; --i) {
                const ChainItem& it = chain[i - 1];
                // dL/dtau of an exponential is taken at its output (end of its last factor)
                const bool stage_end = (i == M) || (chain[i].stage != it.stage);
                if (stage_end) {
                    int rc2 = dot_h(it.stage, lam[cl], xs[i]);
                    if (rc2) return rc2;
                }
                FactorBwdArgs ba{};
                ba.gin = lam[cl];
                ba.xin = xs[i - 1];
                ba.gout = lam[cl ^ 1];
                ba.udiag = udiag;
                ba.coef = coef + size_t(it.stage) * pl.NC;
                ba.coef_bstride = coef_bstride;
                ba.ge = ge + size_t(it.stage) * ge_rec;
                ba.ge_bstride = ge_bstride;
                ba.ge_rstride = pl.NC + 1;
                ba.wtot = wtot;
                ba.dim = uint32_t(pl.dim);
                ba.gr = it.s.gr;
                ba.gi = it.s.gi;
                ba.br = it.s.br;
                ba.bi = it.s.bi;
                ba.g = rt.garg;
                ba.pair = rt.parg;
                ba.obs_bstride = obs_bstride;
                ba.obs_ostride = obs_ostride;
                if (pl.shard_bits) {  // partner ranks' cotangent slabs (exchanged like the forward slabs: shard_signal)
                    ba.sh_bits = pl.shard_bits;
                    ba.sh_nl = pl.NL;
                    ba.sh_rank_first = pl.rank_first;
                    ba.sh_self = pl.shard_self ? 1 : 0;
                    for (int kq = 0; kq < pl.shard_bits; ++kq) ba.sh_rem[kq] = pl.shard_self ? nullptr : static_cast<const double2*>(rt.shard_recv[kq]);
                    shard_groups(pl, ba.sh_grp);
                    for (int q = 0; q < ba.g.ga; ++q) ba.g.amask[q] &= uint32_t(pl.dim - 1);  // in-slab flips only (as in launch_factor)
                    int rcs = shard_signal(rt, 0, lam[cl]);  // the partners need this rank's cotangent ...
                    if (!rcs) rcs = shard_signal(rt, 1, nullptr);  // ... and this launch reads theirs
                    if (rcs) return rcs;
                }
                if (save_k[i - 1] >= 0 && inj.any()) {  // gout is the cotangent at save point k: add what is injected there
                    const int ks = save_k[i - 1];
                    ba.inj_gstate = inj.gstate ? inj.gstate + size_t(ks) * sv : nullptr;
                    ba.inj_gexp = inj.gexp ? inj.gexp + size_t(ks) * pl.B : nullptr;
                    ba.inj_obs = inj.obs;
                    ba.inj_n_obs = inj.n_obs;
                    ba.inj_ostride = long(pl.T + 1) * pl.B;
                }
                if (direct_global_ok(rt)) {
                    const dim3 grid8(grid.x * 8, grid.y);
                    switch (pl.N) {
#define RYDIFF_CASE1(NQ) case NQ: hipLaunchKernelGGL((k_factor_bwd_direct_global<NQ, true>), grid8, dim3(256), 0, stream, ba); break;
#define RYDIFF_CASE(NQ) case NQ: hipLaunchKernelGGL((k_factor_bwd_direct_global<NQ, false>), grid, dim3(256), 0, stream, ba); break;
                        RYDIFF_CASE1(12) RYDIFF_CASE1(13)
                        RYDIFF_CASE(14) RYDIFF_CASE(15) RYDIFF_CASE(16) RYDIFF_CASE(17) RYDIFF_CASE(18) RYDIFF_CASE(19) RYDIFF_CASE(20)
#undef RYDIFF_CASE
#undef RYDIFF_CASE1
                    }
                } else {
                    hipLaunchKernelGGL(k_factor_bwd_direct, grid, dim3(256), 0, stream, ba);
                }
                LAUNCH_CHECK();
                cl ^= 1;
            }
        }
        return RYDIFF_OK;
    };
    if (!persisted) {
        for (int b0 = 0; b0 < pl.B; b0 += (grp ? grp : pl.B)) {
            rc = sweep(BatchSlice{b0, std::min(grp ? grp : pl.B, pl.B - b0), grp > 0});
            if (rc) return rc;
        }
    }
    if (g_psi0) HIP_TRY(hipMemcpyAsync(g_psi0, lam[cl], pl.state_bytes, hipMemcpyDeviceToDevice, stream));

    // scatter to tables / tsave
    if (g_amp) HIP_TRY(hipMemsetAsync(g_amp, 0, size_t(pl.Bc) * pl.Ka * pl.n_samples * 16, stream));
    if (g_det) HIP_TRY(hipMemsetAsync(g_det, 0, size_t(pl.Bc) * pl.Kd * pl.n_samples * 8, stream));
    if (g_tsave) HIP_TRY(hipMemsetAsync(g_tsave, 0, size_t(pl.T + 1) * 8, stream));
    if ((g_amp && pl.Ka) || (g_det && pl.Kd) || g_tsave) {
        char* m = ws + pl.off_meta2;  // (the save-point flags of the one-launch adjoint, which share the region, are dead by now)
        if (g_tsave) {
            std::vector<StageBwdDev> sb(E);
            for (size_t e = 0; e < E; ++e) {
                const Stage& st = pl.stages[e];
                sb[e] = {st.tau_scale, st.tnw[0], st.tnw[1], st.tn[0], st.tn[1], st.t_hi, st.t_lo};
            }
            rc = upload_words(stream, m, sb.data(), E * sizeof(StageBwdDev));
            if (rc) return rc;
        }
        ScatterArgs sa{};
        sa.ge = ge;
        sa.st = reinterpret_cast<const StageDev*>(ws + pl.off_meta_idx);
        sa.sb = reinterpret_cast<const StageBwdDev*>(m);
        sa.inv_dt = pl.dt > 0.0 ? 1.0 / pl.dt : 0.0;
        sa.amp = static_cast<const double2*>(p->amp_tables);
        sa.det = p->det_tables;
        sa.g_amp = static_cast<double2*>(g_amp);
        sa.g_det = g_det;
        sa.g_tsave = g_tsave;
        sa.E = int(E);
        sa.n_samples = pl.n_samples;
        sa.Ka = pl.Ka;
        sa.Kd = pl.Kd;
        sa.NC = pl.NC;
        sa.ga = pl.ga.n;
        sa.gd = pl.gd.n;
        for (int g = 0; g < pl.ga.n; ++g) sa.amem[g] = pl.ga.members[g];
        for (int g = 0; g < pl.gd.n; ++g) sa.dmem[g] = pl.gd.members[g];
        hipLaunchKernelGGL(k_scatter_grads, dim3((unsigned(E) + 63) / 64, pl.Bc), dim3(64), 0, stream, sa);
        LAUNCH_CHECK();
    }
    if (g_u) {
        const int npairs = pl.N * (pl.N - 1) / 2;
        if (npairs > 0) {
            HIP_TRY(hipMemsetAsync(g_u, 0, size_t(npairs) * 8, stream));
            const unsigned nb = unsigned(std::min<size_t>((pl.dim + 255) / 256, 256));
            hipLaunchKernelGGL(k_ugrad, dim3(nb, npairs), dim3(256), 0, stream, g_u, wtot, pl.N, uint32_t(pl.dim),
                               pl.shard_bits ? pl.B : 0, pl.NL, pl.rank_first);
            LAUNCH_CHECK();
        }
    }
    return RYDIFF_OK;
}

int rydiff_apply_factor(const RydProblem* p, const double* c_amp_reim, const double* c_det, const double* gamma_reim,
                        const double* beta_reim, const void* x, void* y, int n_remote, const void* const* remote,
                        const double* remote_coef_reim, int reuse_diag, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!p) return fail(RYDIFF_EINVAL, "null problem");
    if (!x || !y || !workspace || !gamma_reim || !beta_reim) return fail(RYDIFF_EINVAL, "null buffer");
    if ((p->n_amp_terms > 0 && !c_amp_reim) || (p->n_det_terms > 0 && !c_det))
        return fail(RYDIFF_EINVAL, "missing coefficient values (c_amp_reim / c_det) for the problem's terms");
    if (n_remote < 0 || n_remote > kMaxRemote || (n_remote > 0 && (!remote || !remote_coef_reim)))
        return fail(RYDIFF_EINVAL, "bad remote vector list");
    RydProblem q = *p;
    double dummy_t[2] = {0.0, 1.0};
    if (q.n_tsave < 2 || !q.tsave) {
        q.n_tsave = 2;
        q.tsave = dummy_t;
    }
    q.solver = RYDIFF_SOLVER_KRYLOV_SE;
    Runtime rt;
    std::string err;
    if (!build_plan(&q, rt.pl, err)) return fail(RYDIFF_EINVAL, err);
    Plan& pl = rt.pl;
    if (pl.n_pair) return fail(RYDIFF_ENOTIMPL, "rydiff_apply_factor does not take pair terms");
    fill_group_args(pl, rt.garg);
    const size_t need = align_up(pl.dim * sizeof(double));
    if (workspace_bytes < need) return fail(RYDIFF_EWORKSPACE, "workspace too small: need " + std::to_string(need));
    FactorArgs fa{};
    fa.use_inline = 1;
    for (int g = 0; g < pl.ga.n; ++g)
        for (int k = 0; k < pl.Ka; ++k)
            if (pl.ga.members[g] >> k & 1ull) {
                fa.coef_inline[g] += c_amp_reim[2 * k];
                fa.coef_inline[pl.ga.n + g] += c_amp_reim[2 * k + 1];
            }
    for (int g = 0; g < pl.gd.n; ++g)
        for (int k = 0; k < pl.Kd; ++k)
            if (pl.gd.members[g] >> k & 1ull) fa.coef_inline[2 * pl.ga.n + g] += 2.0 * c_det[k];
    double* udiag = static_cast<double*>(workspace);
    if (!reuse_diag) {
        if (pl.N > 1) {
            hipLaunchKernelGGL(k_build_udiag, dim3((pl.dim + 255) / 256), dim3(256), 0, stream, udiag, p->u_pairs, pl.N, uint32_t(pl.dim));
            LAUNCH_CHECK();
        } else {
            HIP_TRY(hipMemsetAsync(udiag, 0, pl.dim * sizeof(double), stream));
        }
    }
    fa.xin = static_cast<const double2*>(x);
    fa.xout = static_cast<double2*>(y);
    fa.udiag = udiag;
    fa.coef = nullptr;
    fa.coef_bstride = 0;
    fa.dim = uint32_t(pl.dim);
    fa.gr = gamma_reim[0];
    fa.gi = gamma_reim[1];
    fa.br = beta_reim[0];
    fa.bi = beta_reim[1];
    fa.g = rt.garg;
    fa.n_remote = n_remote;
    for (int k = 0; k < n_remote; ++k) {
        fa.remote[k] = static_cast<const double2*>(remote[k]);
        fa.rc[2 * k] = remote_coef_reim[2 * k];
        fa.rc[2 * k + 1] = remote_coef_reim[2 * k + 1];
    }
    hipLaunchKernelGGL(k_factor_direct, dim3(unsigned((pl.dim + 255) / 256), pl.B), dim3(256), 0, stream, fa);
    LAUNCH_CHECK();
    return RYDIFF_OK;
}

int rydiff_apply_hamiltonian(const RydProblem* p, const double* c_amp_reim, const double* c_det, const void* x, void* y,
                             void* workspace, size_t workspace_bytes, void* stream_) {
    const double zero[2] = {0.0, 0.0}, one[2] = {1.0, 0.0};
    return rydiff_apply_factor(p, c_amp_reim, c_det, zero, one, x, y, 0, nullptr, nullptr, 0, workspace, workspace_bytes, stream_);
}

}  // extern "C"
