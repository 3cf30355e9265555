// persist_kernels.hpp — persistent single-workgroup propagator and adjoint for small registers (N <= 12); included by
// rydiff.hip after chain_kernels.hpp (uses partner_sums).
//
// A 2^N <= 4096 amplitude state (<= 64 KiB) fits one workgroup's registers + LDS, so the WHOLE trajectory — every factor
// of every time step — runs in ONE launch per direction, one workgroup per trajectory (BASELINE configs 1 and 2, the
// notebook-sized training loops of model.py).  That removes ~10^4 launch latencies per pass.  What is left is a chain of
// ~10^4 dependent iterations, so every global-memory latency inside the loop counts: the per-factor records (scalars +
// coefficient record of the factor's exponential) are therefore staged through LDS 64 factors at a time by >= 64 threads
// (two dependent latencies per 64 factors instead of two per factor), the first observable lives in registers, and the
// adjoint keeps the recomputed factor inputs of a time step in LDS when they fit.
#pragma once

constexpr int kPersistGroups = 4;               // amplitude / detuning groups handled by the persistent adjoint
constexpr int kPersistBwdMaxQubits = 11;         // largest register the persistent adjoint handles
constexpr int kStageChunk = 64;                 // factors staged per refill
constexpr int kStageNC = 3 * kPersistGroups;    // coefficient records up to this size are staged (else read from global)
constexpr int kParkAmps = 2048;                 // LDS budget (amplitudes) for the adjoint's recomputed factor inputs

struct PersistFactor {
    double gr, gi, br, bi;
    int stage;
    int save_index;  // k+1 when this factor ends tsave interval k (state is stored / observed), else 0
    int step_first;  // global index of the first factor of this factor's tsave interval
    int pad;
};

// dense two-qubit terms with the whole register in the LDS tile: sum_p sum_s T_p[4*own + s] * tile[x with the pair's bits = s]
__device__ __forceinline__ double2 pair_apply_tile(const PairArgs& pa, int which, const double2* stab /* LDS copy of pa.tab */,
                                                    const double2* tile, unsigned x) {
    double2 acc = make_double2(0.0, 0.0);
    for (int t = 0; t < pa.n; ++t) {
        const uint32_t ma = pa.ma[t], mb = pa.mb[t];
        const int own = ((x & ma) ? 2 : 0) | ((x & mb) ? 1 : 0);
        const double2* row = stab + (t * 2 + which) * 16 + own * 4;
        const unsigned dm = pa.dl[t];  // relative flips the block has at all (uniform): the others cost nothing
#pragma unroll
        for (int dlt = 0; dlt < 4; ++dlt) {
            if (!(dm >> dlt & 1u)) continue;
            const double2 c = row[own ^ dlt];
            const double2 q = tile[x ^ ((dlt & 2) ? ma : 0u) ^ ((dlt & 1) ? mb : 0u)];
            acc.x += c.x * q.x - c.y * q.y;
            acc.y += c.x * q.y + c.y * q.x;
        }
    }
    return acc;
}

struct PersistArgs {
    const double2* psi0;      // [B][dim]
    double2* states;          // [n_tsave][B][dim] or nullptr
    double2* tape_all;        // full tape [(n_factors + 1)][B][dim] (entry g + 1 = output of factor g) or nullptr
    const double* udiag;      // [dim]
    const double* coef;       // [Bc][E][NC]
    long coef_bstride;
    int NC;
    const PersistFactor* factors;
    int n_factors;
    const double* obs;        // [n_obs][dim] or nullptr
    double* expect;           // [n_obs][n_tsave][B]
    int n_obs, n_tsave, B;
    uint32_t dim;
    int ga, gd;
    uint32_t amask[kMaxGroups];
    uint32_t dmask[kMaxGroups];
    int dcnt[kMaxGroups];
    uint32_t cond;            // amplitude groups with conditioned flips (three-level registers, RydProblem.amp_conditioned_terms)
    PairArgs pair;
};

// stage factors [f0, f0 + count) and (when they fit) their coefficient records; all threads of the workgroup take part
template <int NTL>
__device__ __forceinline__ void stage_factors(const PersistFactor* __restrict__ factors, int f0, int count,
                                              const double* __restrict__ coef_b, int NC, bool stage_coef,
                                              PersistFactor* sfac, double (*scoef)[kStageNC]) {
    for (int t = int(threadIdx.x); t < count; t += NTL) {
        const PersistFactor p = factors[f0 + t];
        sfac[t] = p;
        if (stage_coef) {
            const double* __restrict__ src = coef_b + size_t(p.stage) * NC;
            for (int c = 0; c < NC; ++c) scoef[t][c] = src[c];
        }
    }
}

// SMALLG: at most kPersistGroups amplitude and detuning groups — group loops are unrolled (masks / counts stay in
// scalar registers instead of being re-read from the kernel-argument segment every factor) and the coefficient
// records come from the LDS stage; otherwise generic loops over up to kMaxGroups groups with coefficients from global.
// FAST (implies SMALLG): one amplitude group driving every qubit (a global channel) and at most one detuning group — no
// group loops, no mask tests.
// GLMAX: group slots the SMALLG instantiation loops over (2 when there are at most two amplitude and two detuning groups).
// PERBIT (forward sweep; !SMALLG): every amplitude group and every detuning group is ONE qubit — the term structure of the stochastic-noise
// runs (backend._run_noisy: one single-qubit amplitude and detuning term per atom, per-trajectory tables) and of sequences with several
// local channels.  The generic path loops over the groups (a partner-sum call with zeroed accumulators per group, a popcount per detuning
// group and amplitude, 3 N coefficients read from global memory per factor): 4-7 x the time of the global-drive sweep.  Here the
// coefficient record is staged in LDS like SMALLG's, the detuning diagonal is a per-thread sum over the lane bits plus the register bits,
// and every partner read is multiplied by its own qubit's coefficient (beta c where the own bit is set, beta conj(c) where not).
template <int LT, int LGT, bool CPLX, bool SMALLG, bool FAST = false, int GLMAX = kPersistGroups, bool PERBIT = false>
__global__ __launch_bounds__((1 << LGT) < 64 ? 64 : (1 << LGT)) void k_persist(PersistArgs a) {
    static_assert(!PERBIT || (!SMALLG && !FAST), "PERBIT replaces the generic group loops");
    constexpr int GL = FAST ? 1 : GLMAX;
    constexpr int NT = 1 << LGT, R = 1 << (LT - LGT), NTL = NT < 64 ? 64 : NT, NW = NTL / 64;
    __shared__ __attribute__((aligned(16))) double2 tile[1 << LT];
    __shared__ double red[NW];
    __shared__ PersistFactor sfac[kStageChunk];
    __shared__ double scoef[SMALLG ? kStageChunk : 1][kStageNC];
    __shared__ double scoef_pb[PERBIT ? kStageChunk : 1][3 * LT];  // PERBIT: c_re[ga], c_im[ga], dcoef[gd] of the staged factors
    int ga_of[LT], gd_of[LT];  // PERBIT: group (coefficient slot) of tile bit b = amplitude-index bit b, -1: not driven (uniform)
    if constexpr (PERBIT) {
#pragma unroll
        for (int bb = 0; bb < LT; ++bb) {
            ga_of[bb] = gd_of[bb] = -1;
            for (int g = 0; g < a.ga; ++g)
                if (a.amask[g] == (1u << bb)) ga_of[bb] = g;
            for (int g = 0; g < a.gd; ++g)
                if (a.dmask[g] == (1u << bb)) gd_of[bb] = g;
        }
    }
    __shared__ double2 spair[RYDIFF_MAX_PAIR_TERMS * 32];
    for (int i = int(threadIdx.x); i < a.pair.n * 32; i += NTL) spair[i] = a.pair.tab[i];  // published by the first barrier
    const unsigned tid0 = threadIdx.x;
    const unsigned tid = tid0;
    const bool active = tid < NT;  // tiny registers run on a partial first wave; the rest only helps staging
    const int b = blockIdx.x;
    const size_t boff = size_t(b) * a.dim;
    const double* __restrict__ coef_b = a.coef + size_t(b) * a.coef_bstride;
    // up to 6 qubits the state is one amplitude per lane of ONE wave: partners through lane exchanges, the LDS tile and
    // its two barriers per factor are only kept for the dense pair terms
    const bool lanes = LT <= 6 && a.pair.n == 0;
    double2 v[R];
    double ud[R], ob0[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned x = unsigned(r) * NT + tid;
        v[r] = make_double2(0.0, 0.0);
        ud[r] = ob0[r] = 0.0;
        if (active) {
            v[r] = a.psi0[boff + x];
            ud[r] = a.udiag[x];
            if (a.n_obs > 0) ob0[r] = a.obs[x];
            tile[x] = v[r];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) asm volatile("" : "+v"(ud[r]), "+v"(ob0[r]));  // these loads are complete before the loop:
    __syncthreads();                                                         // no vmcnt(0) behind the state stores inside it
    for (int f = 0; f < a.n_factors; ++f) {
        const int fs = f % kStageChunk;
        if (fs == 0) {
            const int count = a.n_factors - f < kStageChunk ? a.n_factors - f : kStageChunk;
            stage_factors<NTL>(a.factors, f, count, coef_b, a.NC, SMALLG, sfac, scoef);
            if constexpr (PERBIT) {
                for (int t = int(threadIdx.x); t < count * a.NC; t += NTL) {
                    const int ft = t / a.NC, c = t - ft * a.NC;
                    scoef_pb[ft][c] = coef_b[size_t(a.factors[f + ft].stage) * a.NC + c];
                }
            }
            __syncthreads();
        }
        unsigned tid = tid0;
        asm volatile("" : "+v"(tid));  // keep per-lane constants (signs, popcounts, LDS addresses) out of long-lived registers
        const PersistFactor pf = sfac[fs];
        const double* __restrict__ cfg = coef_b + size_t(pf.stage) * a.NC;  // !SMALLG
        const double* cfs = scoef[SMALLG ? fs : 0];                          // SMALLG (LDS)
        auto cf = [&](int c) -> double { return SMALLG ? cfs[c] : cfg[c]; };
        double2 q[R];
        if constexpr (PERBIT) {
            if (active) {
                const double* cpb = scoef_pb[fs];
                double dlane = 0.0;  // detuning of the lane bits that are in |r> (bit = 0)
#pragma unroll
                for (int bb = 0; bb < LGT; ++bb)
                    if (gd_of[bb] >= 0 && !(tid >> bb & 1u)) dlane += cpb[2 * a.ga + gd_of[bb]];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    double d = ud[r] + dlane;
#pragma unroll
                    for (int bb = LGT; bb < LT; ++bb)
                        if (gd_of[bb] >= 0 && !(r >> (bb - LGT) & 1)) d += cpb[2 * a.ga + gd_of[bb]];
                    const double dr = pf.gr + pf.br * d, di = pf.gi + pf.bi * d;
                    q[r].x = dr * v[r].x - di * v[r].y;
                    q[r].y = dr * v[r].y + di * v[r].x;
                }
                double2 sreal[R];  // !CPLX (phase-free drives): sum_b c_b * partner_b with REAL c_b, multiplied by beta once at the end
#pragma unroll
                for (int r = 0; r < R; ++r) sreal[r] = make_double2(0.0, 0.0);
                auto flip_bit = [&](auto bc) {
                    constexpr int bb = decltype(bc)::value;
                    if (ga_of[bb] < 0) return;  // uniform
                    if constexpr (!CPLX) {  // 2 FMAs per partner instead of 4, no per-lane coefficient choice
                        const double cr = cpb[ga_of[bb]];
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            double2 pv;
                            if constexpr (bb >= LGT) pv = v[r ^ (1 << (bb >= LGT ? bb - LGT : 0))];
                            else if constexpr (LT <= 6) pv = lanes ? lane_xor<(bb < 6 ? bb : 0)>(v[r]) : tile[(unsigned(r) * NT + tid) ^ (1u << bb)];
                            else pv = tile[(unsigned(r) * NT + tid) ^ (1u << bb)];
                            sreal[r].x = fma(cr, pv.x, sreal[r].x);
                            sreal[r].y = fma(cr, pv.y, sreal[r].y);
                        }
                        return;
                    }
                    const double cr = cpb[ga_of[bb]], ci = cpb[a.ga + ga_of[bb]];
                    const double k1r = pf.br * cr - pf.bi * ci, k1i = pf.br * ci + pf.bi * cr;  // beta c       (own bit set)
                    const double k0r = pf.br * cr + pf.bi * ci, k0i = pf.bi * cr - pf.br * ci;  // beta conj(c) (own bit clear)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        double2 pv;
                        bool own;
                        if constexpr (bb >= LGT) {
                            pv = v[r ^ (1 << (bb >= LGT ? bb - LGT : 0))];
                            own = (r >> (bb >= LGT ? bb - LGT : 0)) & 1;
                        } else {
                            if constexpr (LT <= 6) pv = lanes ? lane_xor<(bb < 6 ? bb : 0)>(v[r]) : tile[(unsigned(r) * NT + tid) ^ (1u << bb)];
                            else pv = tile[(unsigned(r) * NT + tid) ^ (1u << bb)];
                            own = (tid >> bb) & 1u;
                        }
                        const double kr = own ? k1r : k0r, ki = own ? k1i : k0i;
                        q[r].x += kr * pv.x - ki * pv.y;
                        q[r].y += kr * pv.y + ki * pv.x;
                    }
                };
                static_for<0, LT>(flip_bit);
                if constexpr (!CPLX) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        q[r].x += pf.br * sreal[r].x - pf.bi * sreal[r].y;
                        q[r].y += pf.br * sreal[r].y + pf.bi * sreal[r].x;
                    }
                }
            }
        } else if (active) {
            double dsh[R];  // time-dependent part of the diagonal
#pragma unroll
            for (int r = 0; r < R; ++r) dsh[r] = ud[r];
            auto det_group = [&](int g) {
                const double c = cf(2 * a.ga + g);
#pragma unroll
                for (int r = 0; r < R; ++r) dsh[r] += c * double(a.dcnt[g] - popc_i((unsigned(r) * NT + tid) & a.dmask[g]));
            };
            if constexpr (SMALLG) {
#pragma unroll
                for (int g = 0; g < GL; ++g)
                    if (g < a.gd) det_group(g);
            } else {
                for (int g = 0; g < a.gd; ++g) det_group(g);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double dr = pf.gr + pf.br * dsh[r], di = pf.gi + pf.bi * dsh[r];
                q[r].x = dr * v[r].x - di * v[r].y;
                q[r].y = dr * v[r].y + di * v[r].x;
            }
            auto amp_group = [&](int g) {
                double2 ts[R], ds[R];
                if constexpr (LT <= 6) {
                    if (lanes) do { if (!FAST && (a.cond >> g & 1u)) partner_sums_lanes<LT, CPLX, false, true>(v[0], a.amask[g], tid, ts[0], ds[0]); else partner_sums_lanes<LT, CPLX, FAST>(v[0], a.amask[g], tid, ts[0], ds[0]); } while (0);
                    else do { if (!FAST && (a.cond >> g & 1u)) partner_sums<LT, LGT, CPLX, false, true>(tile, v, a.amask[g], tid, ts, ds); else partner_sums<LT, LGT, CPLX, FAST>(tile, v, a.amask[g], tid, ts, ds); } while (0);
                } else {
                    do { if (!FAST && (a.cond >> g & 1u)) partner_sums<LT, LGT, CPLX, false, true>(tile, v, a.amask[g], tid, ts, ds); else partner_sums<LT, LGT, CPLX, FAST>(tile, v, a.amask[g], tid, ts, ds); } while (0);
                }
                const double cr = cf(g), ci = cf(a.ga + g);
                const double k1r = pf.br * cr, k1i = pf.bi * cr, k2r = -pf.bi * ci, k2i = pf.br * ci;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    q[r].x += k1r * ts[r].x - k1i * ts[r].y;
                    q[r].y += k1r * ts[r].y + k1i * ts[r].x;
                    if (CPLX) {
                        q[r].x += k2r * ds[r].x - k2i * ds[r].y;
                        q[r].y += k2r * ds[r].y + k2i * ds[r].x;
                    }
                }
            };
            if constexpr (SMALLG) {
#pragma unroll
                for (int g = 0; g < GL; ++g)
                    if (FAST || g < a.ga) amp_group(g);
            } else {
                for (int g = 0; g < a.ga; ++g) amp_group(g);
            }
            if (a.pair.n) {  // beta * (dense two-qubit terms)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double2 pv = pair_apply_tile(a.pair, 0, spair, tile, unsigned(r) * NT + tid);
                    q[r].x += pf.br * pv.x - pf.bi * pv.y;
                    q[r].y += pf.br * pv.y + pf.bi * pv.x;
                }
            }
        }
        if (!lanes) __syncthreads();  // every partner read of the old vector is done
        if (active) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                v[r] = q[r];
                if (!lanes) tile[unsigned(r) * NT + tid] = q[r];
            }
            if (a.tape_all) {  // full tape: the adjoint sweep (per-factor launches at 12 qubits) recomputes nothing
                double2* dst = a.tape_all + (size_t(f + 1) * a.B + b) * a.dim;
#pragma unroll
                for (int r = 0; r < R; ++r) dst[unsigned(r) * NT + tid] = v[r];
            }
        }
        if (pf.save_index) {
            if (a.states && active) {
                double2* dst = a.states + (size_t(pf.save_index) * a.B + b) * a.dim;
#pragma unroll
                for (int r = 0; r < R; ++r) dst[unsigned(r) * NT + tid] = v[r];
            }
            for (int o = 0; o < a.n_obs; ++o) {
                double e = 0.0;
                if (active) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double w = o == 0 ? ob0[r] : a.obs[size_t(o) * a.dim + unsigned(r) * NT + tid];
                        e += w * (v[r].x * v[r].x + v[r].y * v[r].y);
                    }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) e += __shfl_down(e, off, 64);
                if (NW > 1) {
                    if ((tid & 63) == 0) red[tid >> 6] = e;
                    __syncthreads();
                    if (tid == 0) {
                        e = 0.0;
                        for (int w = 0; w < NW; ++w) e += red[w];
                    }
                    if (a.n_obs > 1) __syncthreads();  // red is reused by the next observable
                }
                if (tid == 0) a.expect[(size_t(o) * a.n_tsave + pf.save_index) * a.B + b] = e;  // single writer: deterministic
            }
        }
        if (!lanes) __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent ADJOINT sweep: the whole reverse pass in ONE launch.  Per tsave interval (last to first) the workgroup
//   1. adds the cotangent of the interval's end point (grad_states + 2 * grad_expect * O * psi),
//   2. recomputes the interval's factor inputs x_1 .. x_{M-1} from the saved state; each thread parks ITS OWN amplitudes
//      (LDS when the interval fits, else a global scratch slot) and reads the same locations back — no extra barriers,
//   3. walks the factors backwards: mu <- (conj(gamma) + conj(beta) H) mu, fused with the gradient contractions
//      <F_g mu, x>, Re(beta conj(mu) x) n(x) and, at the end of every exponential, Im<H mu, x_out> (= dL/dtau).
// Per-exponential gradient sums stay in registers until the exponential is complete: one workgroup reduction each.
// ---------------------------------------------------------------------------------------------------------------------
struct PersistBwdArgs {
    const double2* tape;      // [n_tsave][B][dim] saved states; tape_full: [(n_factors + 1)][B][dim], entry f = input of factor f
    int tape_full;            // the forward sweep kept every factor output: nothing is recomputed
    const int32_t* save_entry;  // tape_full: tape entry of the state at save point k ([n_tsave])
    double2* chainbuf;        // [slots][B][dim]
    const double2* gstate;    // [n_tsave][B][dim] or nullptr
    const double* gexp;       // [n_obs][n_tsave][B] or nullptr
    const int32_t* gflags;    // [n_tsave]: grad_expect has a non-zero entry at this save point (with gexp)
    const double* obs;        // [n_obs][dim]
    const double* udiag;      // [dim]
    const double* coef;       // [Bc][E][NC]
    long coef_bstride;
    int NC;
    const PersistFactor* factors;
    int n_factors;
    double* ge;               // [Bc][E][replicas][NC+1]
    long ge_bstride, ge_sstride;
    double* wtot;             // [dim] or nullptr
    double2* mu_out;          // [B][dim]: cotangent w.r.t. psi0
    int want_tau;
    int n_obs, n_tsave, B;
    uint32_t dim;
    int ga, gd;
    uint32_t amask[kPersistGroups];
    uint32_t dmask[kPersistGroups];
    int dcnt[kPersistGroups];
    uint32_t cond;            // as in PersistArgs
    PairArgs pair;
};

template <int LT, int LGT, bool CPLX, bool FAST = false, int GLMAX = kPersistGroups>  // FAST / GLMAX: as in k_persist
__global__ __launch_bounds__((1 << LGT) < 64 ? 64 : (1 << LGT)) void k_persist_bwd(PersistBwdArgs a) {
    constexpr int GL = FAST ? 1 : GLMAX;
    constexpr int NT = 1 << LGT, R = 1 << (LT - LGT), NTL = NT < 64 ? 64 : NT, NW = NTL / 64;
    constexpr int NV = 3 * kPersistGroups + 1;
    constexpr int PARK = (1 << LT) <= kParkAmps ? kParkAmps : 1;   // parked factor inputs (amplitudes) in LDS
    constexpr int PARK_SLOTS = (1 << LT) <= kParkAmps ? (kParkAmps >> LT) : 0;
    // registers to spare: the states at both ends of the interval stay in registers (the start state of one interval is
    // the end state of the next one processed) and the next start state is fetched one interval ahead
    constexpr bool KEEPX = R <= 2;
    __shared__ __attribute__((aligned(16))) double2 tile[1 << LT];
    __shared__ __attribute__((aligned(16))) double2 park[PARK];
    __shared__ double red[NV * NW];
    __shared__ PersistFactor sfac[kStageChunk];
    __shared__ double scoef[kStageChunk][kStageNC];
    __shared__ int sflag[kStageChunk];
    __shared__ double2 spair[RYDIFF_MAX_PAIR_TERMS * 32];
    for (int i = int(threadIdx.x); i < a.pair.n * 32; i += NTL) spair[i] = a.pair.tab[i];  // published by the first refill barrier
    const unsigned tid0 = threadIdx.x;
    const bool active = tid0 < NT;
    const int b = blockIdx.x;
    const size_t boff = size_t(b) * a.dim;
    const size_t sv = size_t(a.B) * a.dim;
    const double* __restrict__ coef_b = a.coef + size_t(b) * a.coef_bstride;
    const bool lanes = LT <= 6 && a.pair.n == 0;  // one amplitude per lane of one wave: lane exchanges instead of the LDS tile
    // the host only takes this path when an interval fits the staging window and the coefficient record fits a stage row
    double2 mu[R], xend[KEEPX ? R : 1], xnext[KEEPX ? R : 1];
    double ud[R], wt[R];
    const int n_save = a.factors[a.n_factors - 1].save_index;  // = T
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned x = unsigned(r) * NT + tid0;
        mu[r] = make_double2(0.0, 0.0);
        wt[r] = 0.0;
        ud[r] = active ? a.udiag[x] : 0.0;
        if (KEEPX) {
            // (tape_full: the tape holds every factor output; save point k sits at entry save_entry[k])
            xend[r] = active ? a.tape[size_t(a.tape_full ? a.save_entry[n_save] : n_save) * sv + boff + x] : make_double2(0.0, 0.0);  // state at the final time
            xnext[r] = active ? a.tape[size_t(a.tape_full ? a.save_entry[n_save - 1] : n_save - 1) * sv + boff + x] : make_double2(0.0, 0.0);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) asm volatile("" : "+v"(ud[r]));  // the loads above are complete before the loops start
    double acc_re[kPersistGroups], acc_im[kPersistGroups], acc_det[kPersistGroups], acc_tau = 0.0;
#pragma unroll
    for (int g = 0; g < kPersistGroups; ++g) acc_re[g] = acc_im[g] = acc_det[g] = 0.0;

    // mu += grad_states[k] + 2 * sum_o grad_expect[o][k] * obs[o] * psi_k   (psi_k in registers)
    auto state_elem = [&](int k, int r) -> double2 {
        return a.tape[size_t(a.tape_full ? a.save_entry[k] : k) * sv + boff + unsigned(r) * NT + tid0];
    };
    auto inject = [&](int k, bool flagged) {
        if (!active) return;
        if (a.gstate) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double2 gq = a.gstate[size_t(k) * sv + boff + unsigned(r) * NT + tid0];
                mu[r].x += gq.x;
                mu[r].y += gq.y;
            }
        }
        if (a.gexp && flagged) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned x = unsigned(r) * NT + tid0;
                double wsum = 0.0;
                for (int o = 0; o < a.n_obs; ++o) wsum += a.gexp[(size_t(o) * a.n_tsave + k) * a.B + b] * a.obs[size_t(o) * a.dim + x];
                const double2 psi = KEEPX ? xend[r] : state_elem(k, r);
                mu[r].x += 2.0 * wsum * psi.x;
                mu[r].y += 2.0 * wsum * psi.y;
            }
        }
    };

    int w0 = a.n_factors, w1 = -1;  // staged window [w0, w1] of global factor indices (empty)
    int fend = a.n_factors - 1;
    while (fend >= 0) {
        if (!(fend <= w1 && fend >= w0 && sfac[fend - w0].step_first >= w0)) {
            // the interval ending at fend is not (fully) staged: refill with the kStageChunk factors ending at fend
            __syncthreads();
            w1 = fend;
            w0 = fend - (kStageChunk - 1) > 0 ? fend - (kStageChunk - 1) : 0;
            stage_factors<NTL>(a.factors, w0, w1 - w0 + 1, coef_b, a.NC, true, sfac, scoef);
            __syncthreads();
            for (int t = int(tid0); t <= w1 - w0; t += NTL) {
                const int si = sfac[t].save_index;
                sflag[t] = (si && a.gflags) ? a.gflags[si] : 1;
            }
            __syncthreads();
        }
        auto factor_at = [&](int f) -> PersistFactor { return sfac[f - w0]; };
        const PersistFactor pend = factor_at(fend);
        const int k1 = pend.save_index;  // this interval ends at tsave[k1]; xend = state there
        const int fbeg = pend.step_first;
        inject(k1, sflag[fend - w0] != 0);
        const int M = fend - fbeg + 1;
        const bool park_lds = (M - 1) <= PARK_SLOTS;
        // x_{i+1} (output of the interval's factor i) is parked in LDS when the interval fits, else in a global slot;
        // every thread reads back only what it stored itself
        auto park_store = [&](int i, unsigned idx, const double2& val) {
            if (park_lds) park[(size_t(i) << LT) + idx] = val;
            else a.chainbuf[size_t(i) * sv + boff + idx] = val;
        };
        auto park_load = [&](int i, unsigned idx) -> double2 {
            if (a.tape_full) return a.tape[size_t(fbeg + i + 1) * sv + boff + idx];  // x_{i+1} = output of the interval's factor i
            return park_lds ? park[(size_t(i) << LT) + idx] : a.chainbuf[size_t(i) * sv + boff + idx];
        };

        // ---- state at the start of the interval, then the factor inputs x_1 .. x_{M-1}
        double2 x0[KEEPX ? R : 1], v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (KEEPX) v[r] = x0[r] = xnext[r];
            else v[r] = active ? state_elem(k1 - 1, r) : make_double2(0.0, 0.0);
        }
        if (KEEPX && k1 >= 2 && active) {
#pragma unroll
            for (int r = 0; r < R; ++r) xnext[r] = state_elem(k1 - 2, r);
        }
        // (full tape: the factor inputs are already in global memory — nothing to recompute)
        for (int i = 0; i + 1 < M && !a.tape_full; ++i) {
            unsigned tid = tid0;
            asm volatile("" : "+v"(tid));  // keep per-lane constants (signs, popcounts, LDS addresses) out of long-lived registers
            const PersistFactor pf = factor_at(fbeg + i);
            const double* cf = scoef[fbeg + i - w0];
            if (!lanes) {
                if (active) {
#pragma unroll
                    for (int r = 0; r < R; ++r) tile[unsigned(r) * NT + tid] = v[r];
                }
                __syncthreads();
            }
            double2 q[R];
            if (active) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const unsigned x = unsigned(r) * NT + tid;
                    double d = ud[r];
#pragma unroll
                    for (int g = 0; g < GL; ++g)
                        if (g < a.gd) d += cf[2 * a.ga + g] * double(a.dcnt[g] - popc_i(x & a.dmask[g]));
                    const double dr = pf.gr + pf.br * d, di = pf.gi + pf.bi * d;
                    q[r].x = dr * v[r].x - di * v[r].y;
                    q[r].y = dr * v[r].y + di * v[r].x;
                }
#pragma unroll
                for (int g = 0; g < GL; ++g) {
                    if (FAST || g < a.ga) {
                        double2 ts[R], ds[R];
                        if constexpr (LT <= 6) {
                            if (lanes) do { if (!FAST && (a.cond >> g & 1u)) partner_sums_lanes<LT, CPLX, false, true>(v[0], a.amask[g], tid, ts[0], ds[0]); else partner_sums_lanes<LT, CPLX, FAST>(v[0], a.amask[g], tid, ts[0], ds[0]); } while (0);
                            else do { if (!FAST && (a.cond >> g & 1u)) partner_sums<LT, LGT, CPLX, false, true>(tile, v, a.amask[g], tid, ts, ds); else partner_sums<LT, LGT, CPLX, FAST>(tile, v, a.amask[g], tid, ts, ds); } while (0);
                        } else {
                            do { if (!FAST && (a.cond >> g & 1u)) partner_sums<LT, LGT, CPLX, false, true>(tile, v, a.amask[g], tid, ts, ds); else partner_sums<LT, LGT, CPLX, FAST>(tile, v, a.amask[g], tid, ts, ds); } while (0);
                        }
                        const double cr = cf[g], ci = cf[a.ga + g];
                        const double k1r = pf.br * cr, k1i = pf.bi * cr, k2r = -pf.bi * ci, k2i = pf.br * ci;
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            q[r].x += k1r * ts[r].x - k1i * ts[r].y;
                            q[r].y += k1r * ts[r].y + k1i * ts[r].x;
                            if (CPLX) {
                                q[r].x += k2r * ds[r].x - k2i * ds[r].y;
                                q[r].y += k2r * ds[r].y + k2i * ds[r].x;
                            }
                        }
                    }
                }
                if (a.pair.n) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double2 pv = pair_apply_tile(a.pair, 0, spair, tile, unsigned(r) * NT + tid);
                        q[r].x += pf.br * pv.x - pf.bi * pv.y;
                        q[r].y += pf.br * pv.y + pf.bi * pv.x;
                    }
                }
            }
            if (!lanes) __syncthreads();
            if (active) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    v[r] = q[r];
                    park_store(i, unsigned(r) * NT + tid, q[r]);
                }
            }
        }

        // ---- adjoint of the interval's factors, last to first; v holds the input of the factor being processed
        // (full tape: the inputs come from global memory, requested one factor ahead: vn = input of factor i - 1)
        double2 vn[R];
        if (a.tape_full && active) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                v[r] = M >= 2 ? park_load(M - 2, unsigned(r) * NT + tid0) : (KEEPX ? x0[r] : state_elem(k1 - 1, r));  // input of factor M-1
                vn[r] = M >= 3 ? park_load(M - 3, unsigned(r) * NT + tid0) : (KEEPX ? x0[r] : state_elem(k1 - 1, r));  // input of factor M-2
            }
        }
        for (int i = M - 1; i >= 0; --i) {
            unsigned tid = tid0;
            asm volatile("" : "+v"(tid));
            const int f = fbeg + i;
            const PersistFactor pf = factor_at(f);
            const double* cf = scoef[f - w0];
            const bool stage_end = (i == M - 1) || (factor_at(f + 1).stage != pf.stage);
            const bool stage_begin = (i == 0) || (factor_at(f - 1).stage != pf.stage);
            if (active) {
                if (a.tape_full) {
                    if (i != M - 1) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            v[r] = vn[r];
                            if (i >= 1) vn[r] = i >= 2 ? park_load(i - 2, unsigned(r) * NT + tid) : (KEEPX ? x0[r] : state_elem(k1 - 1, r));
                        }
                    }
                } else if (i != M - 1) {
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        v[r] = i == 0 ? (KEEPX ? x0[r] : state_elem(k1 - 1, r)) : park_load(i - 1, unsigned(r) * NT + tid);
                }
                if (!lanes) {
#pragma unroll
                    for (int r = 0; r < R; ++r) tile[unsigned(r) * NT + tid] = mu[r];
                }
            }
            if (!lanes) __syncthreads();
            if (active) {
                double2 hm[R];  // H mu
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const unsigned x = unsigned(r) * NT + tid;
                    double d = ud[r];
#pragma unroll
                    for (int g = 0; g < GL; ++g)
                        if (g < a.gd) d += cf[2 * a.ga + g] * double(a.dcnt[g] - popc_i(x & a.dmask[g]));
                    hm[r].x = d * mu[r].x;
                    hm[r].y = d * mu[r].y;
                    const double pr = pf.br * mu[r].x + pf.bi * mu[r].y, pi = pf.bi * mu[r].x - pf.br * mu[r].y;
                    const double rr = pr * v[r].x - pi * v[r].y;  // Re(beta conj(mu) x)
                    wt[r] += rr;
#pragma unroll
                    for (int g = 0; g < GL; ++g)
                        if (g < a.gd) acc_det[g] += rr * double(a.dcnt[g] - popc_i(x & a.dmask[g]));
                }
#pragma unroll
                for (int g = 0; g < GL; ++g) {
                    if (FAST || g < a.ga) {
                        const double cr = cf[g], ci = cf[a.ga + g];
                        double z1r = 0.0, z1i = 0.0, z2r = 0.0, z2i = 0.0;
                        double2 ts[R], ds[R];
                        if constexpr (LT <= 6) {
                            if (lanes) do { if (!FAST && (a.cond >> g & 1u)) partner_sums_lanes<LT, true, false, true>(mu[0], a.amask[g], tid, ts[0], ds[0]); else partner_sums_lanes<LT, true, FAST>(mu[0], a.amask[g], tid, ts[0], ds[0]); } while (0);
                            else do { if (!FAST && (a.cond >> g & 1u)) partner_sums<LT, LGT, true, false, true>(tile, mu, a.amask[g], tid, ts, ds); else partner_sums<LT, LGT, true, FAST>(tile, mu, a.amask[g], tid, ts, ds); } while (0);
                        } else {
                            do { if (!FAST && (a.cond >> g & 1u)) partner_sums<LT, LGT, true, false, true>(tile, mu, a.amask[g], tid, ts, ds); else partner_sums<LT, LGT, true, FAST>(tile, mu, a.amask[g], tid, ts, ds); } while (0);
                        }
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            // (F_g mu) = cr * ts + i * ci * ds
                            hm[r].x += cr * ts[r].x - ci * ds[r].y;
                            hm[r].y += cr * ts[r].y + ci * ds[r].x;
                            z1r += ts[r].x * v[r].x + ts[r].y * v[r].y;
                            z1i += ts[r].x * v[r].y - ts[r].y * v[r].x;
                            z2r += ds[r].x * v[r].x + ds[r].y * v[r].y;
                            z2i += ds[r].x * v[r].y - ds[r].y * v[r].x;
                        }
                        acc_re[g] += pf.br * z1r - pf.bi * z1i;
                        acc_im[g] += pf.br * z2i + pf.bi * z2r;
                    }
                }
                if (a.pair.n) {  // the generator is M = K + T with K Hermitian: M^dagger mu = K mu + T^dagger mu
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double2 pv = pair_apply_tile(a.pair, 1, spair, tile, unsigned(r) * NT + tid);
                        hm[r].x += pv.x;
                        hm[r].y += pv.y;
                    }
                }
                if (stage_end && a.want_tau) {  // dL/dtau = Im<mu, M x_out> = Im<M^dagger mu, x_out>
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double2 o = (i == M - 1) ? (KEEPX ? xend[r] : state_elem(k1, r)) : park_load(i, unsigned(r) * NT + tid);
                        acc_tau += hm[r].x * o.y - hm[r].y * o.x;
                    }
                }
                // mu <- conj(gamma) mu + conj(beta) H mu
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double nx = pf.gr * mu[r].x + pf.gi * mu[r].y + pf.br * hm[r].x + pf.bi * hm[r].y;
                    const double ny = pf.gr * mu[r].y - pf.gi * mu[r].x + pf.br * hm[r].y - pf.bi * hm[r].x;
                    mu[r] = make_double2(nx, ny);
                }
            }
            if (stage_begin) {  // the exponential is complete: reduce its gradient record over the workgroup
                double vals[NV];
#pragma unroll
                for (int g = 0; g < kPersistGroups; ++g) {
                    vals[g] = acc_re[g];
                    vals[kPersistGroups + g] = acc_im[g];
                    vals[2 * kPersistGroups + g] = acc_det[g];
                    acc_re[g] = acc_im[g] = acc_det[g] = 0.0;
                }
                vals[3 * kPersistGroups] = acc_tau;
                acc_tau = 0.0;
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    if (FAST && q != 3 * kPersistGroups && (q % kPersistGroups) != 0) continue;  // only group 0 exists
                    double s = vals[q];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
                    if ((tid0 & 63) == 0) red[q * NW + (tid0 >> 6)] = s;
                }
                __syncthreads();
                if (tid0 < NV) {  // NTL >= 64 > NV
                    double s = 0.0;
                    for (int w = 0; w < NW; ++w) s += red[tid0 * NW + w];
                    const int kind = int(tid0) / kPersistGroups, g = int(tid0) % kPersistGroups;
                    int slot = -1;
                    if (kind == 0 && g < a.ga) slot = g;
                    else if (kind == 1 && g < a.ga) slot = a.ga + g;
                    else if (kind == 2 && g < a.gd) slot = 2 * a.ga + g;
                    else if (kind == 3 && g == 0 && a.want_tau) slot = a.NC;
                    if (slot >= 0) {
                        double* rec = a.ge + size_t(b) * a.ge_bstride + size_t(pf.stage) * a.ge_sstride +
                                      size_t(b % kGradReplicas) * (a.NC + 1);
                        unsafeAtomicAdd(rec + slot, s);
                    }
                }
            }
            if (!lanes || stage_begin) __syncthreads();  // partner reads of mu (and the reduction scratch) are done
        }
        if (KEEPX) {
#pragma unroll
            for (int r = 0; r < R; ++r) xend[r] = x0[r];  // the next interval (one earlier) ends where this one started
        }
        fend = fbeg - 1;
    }
    inject(0, a.gflags ? a.gflags[0] != 0 : true);
    if (active) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned x = unsigned(r) * NT + tid0;
            a.mu_out[boff + x] = mu[r];
            if (a.wtot) unsafeAtomicAdd(a.wtot + x, wt[r]);
        }
    }
}
