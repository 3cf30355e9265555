// chain_kernels.hpp — LDS-tiled "chained two-layout" factor passes (included by rydiff.hip).
//
// One factor of the product-form propagator is  v_f = (gamma_f + beta_f H) v_{f-1},  H = D + sum_j flips_j.
// A workgroup owns a TILE of 2^LT amplitudes in LDS/registers; flips on the tile's LT bits are local.  Two tile
// layouts alternate between launches:
//     layout A : tile = amplitude-index bits [0, LT)                       (one contiguous 2^LT run)
//     layout B : tile = bits [0, 2LT-N) u [LT, N)                         (2^(N-LT) runs of 2^(2LT-N) amplitudes)
// so that A u B covers all N bits (13 <= N <= 2 LT).  Kernel j (layout X_j)
//     1. FINISHES factor j:   v_j = p_j + beta_j * flips_{X_j \ X_{j-1}} v_{j-1}      (p_j = partial from kernel j-1)
//     2. STARTS  factor j+1:  p_{j+1} = (gamma_{j+1} + beta_{j+1} D) v_j + beta_{j+1} * flips_{X_j} v_j
// and writes v_j (complete) and p_{j+1} (partial).  Every launch therefore reads two vectors and writes two, once,
// fully coalesced, whatever N is — no partner-tile loads, no dependence on workgroup->XCD placement — and one matrix-
// free application of H costs exactly one launch.
//
// Per thread: R = 2^(LT-LGT) amplitudes i = r*NT + tid, so the top LT-LGT tile bits are REGISTER bits (flips are
// register renaming) and only the low LGT tile bits go through LDS (conflict-free ds_read_b128: consecutive lanes
// read consecutive 16-byte slots of a permuted 1 KiB span).
//
// Tile sizes (a run-time property of a chain: chain_geom() in rydiff.hip; the forward and the adjoint chain choose independently):
//   k_chain<12, 8 | 9 | 10>   2^12 amplitudes, 256 / 512 / 1024 threads: the workhorse (13..20 and 25..28 qubits, batches)
//   k_chain<11, 10>, <10, 10>  2^11 / 2^10 amplitudes: around 2^19 amplitudes in flight (256 tiles of 2^11: one per CU); tuning variants
//   k_chain_wide<13>           2^13 amplitudes in two register halves (below): 21..24 qubits with two layouts, 29 / 30 with three
#pragma once

constexpr int kSmallTileBits = 10;  // smallest tile (tuning variants 15 / 16: 2^11 / 2^10 amplitudes per workgroup)
constexpr int kTileBits = 12;      // k_chain: 2^12 amplitudes per workgroup (64 KiB of LDS)
constexpr int kWideTileBits = 13;  // k_chain_wide: 2^13 amplitudes per workgroup (128 KiB of the 160 KiB LDS), 21-24 qubits

// Streaming store: the vectors written by a pass are read next by OTHER workgroups in the other tile layout, never by
// this one, so there is no point keeping the lines dirty in this XCD's L2 until the end-of-kernel write-back.
__device__ __forceinline__ void stream_store(double2* p, const double2& v) {
#ifdef RYDIFF_PLAIN_STORES
    *p = v;
#else
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
#endif
}

// Streaming load: every vector element is read exactly once per pass, so there is no point allocating it in the
// vector L1 (global_load_dwordx4 ... nt; measured 14.8 -> 14.3 us per pass on the 20-qubit workload)
__device__ __forceinline__ double2 stream_load(const double2* p) {
#ifdef RYDIFF_PLAIN_LOADS
    return *p;
#else
    double2 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    return v;
#endif
}

struct ChainArgs {
    const double2* u;      // complete v_{j-1}
    const double2* p;      // partial of factor j (unused when !has_p)
    double2* v_out;        // complete v_j (written when write_v)
    double2* q_out;        // partial of factor j+1 (written when has_q)
    const double* utt;       // split interaction diagonal: U(x) = utt[i] + vr[t][LT] + sum_{tile bits a with n_a = 1} vr[t][a]
    const double* vr;        // [tiles][16]
    const double* coef_fin;  // coefficient record of factor j     (trajectory 0)
    const double* coef_sta;  // coefficient record of factor j+1
    long coef_bstride;
    double fb_r, fb_i;                 // beta_j
    double fg_r, fg_i;                 // gamma_j (adjoint mode, recovered contraction: see REC in k_chain)
    int completes;                     // this launch's finish stage yields the COMPLETE v_j (three-layout chains: the middle pass does not)
    double sg_r, sg_i, sb_r, sb_i;     // gamma_{j+1}, beta_{j+1}
    int lo, hs, hb;                    // layout: tile bits [0,lo) -> index bits [0,lo); tile bits [lo,LT) -> index bits [hs,hs+hb)
    uint32_t dim;
    int has_p, has_q, write_v;
    int ga, gd;
    uint32_t fin_mask[kMaxGroups];     // per flip group: TILE-bit mask handled by the finish stage
    uint32_t sta_mask[kMaxGroups];     // per flip group: TILE-bit mask handled by the start stage
    uint32_t dmask[kMaxGroups];        // detuning groups: amplitude-INDEX masks
    int dcnt[kMaxGroups];
    // flip groups with CONDITIONED flips (three-level registers, RydProblem.amp_conditioned_terms): the flip of a qubit acts only where its
    // sibling qubit (amplitude-index bit ^ 1) is 1.  Needs tile layouts that keep sibling pairs together — even lo and hs: the 2^12 tiles
    // of an even register — so that the sibling of tile bit b is tile bit b ^ 1 (k_chain only; chain_geom keeps such problems on them).
    uint32_t cond;
    // Trajectory-per-XCD placement (speed only, results do not depend on it): workgroups are dispatched round-robin over the
    // 8 XCDs (workgroup id % 8), each XCD has its own 4 MiB L2.  With xcd_place the grid is (8 * tiles_per_traj, ceil(B/8)):
    // workgroup w of row y works on tile w / 8 of trajectory 8 y + w % 8, so that ALL tiles of one trajectory, in every
    // layout, run on one XCD and its vectors stay in that L2 from pass to pass (no fabric traffic).  0: grid (tiles, B).
    int xcd_place;
    // Line-sharing tiles on one XCD (speed only): in a layout whose low run is shorter than a 128-byte line (lo < 3: 22 qubits, and
    // the forced two-layout variants beyond) 2^(3-lo) CONSECUTIVE tiles own interleaved pieces of the same lines.  With the plain
    // grid they run on different XCDs (workgroup id % 8), so every line is fetched into 2^(3-lo) L2s and written back in pieces.
    // tile_swz = 3 - lo maps workgroup w to tile  (w & ~(8 G - 1)) | (w % 8) * G | (w / 8) % G,  G = 2^tile_swz: the G tiles of a
    // line run on ONE XCD, 8 workgroup ids apart — the second tile's loads hit the line the first one fetched.  0: identity.
    int tile_swz;
    int b_first, b_count;              // this launch covers trajectories [b_first, b_first + b_count)
    int resident;                      // launch the RES instantiation: plain loads / stores (lines stay in the XCD's L2)
    // backward (adjoint) mode only: u/p/v/q are cotangents, gamma/beta above are already conjugated
    const double2* x_fin;   // input of the factor being finished (own elements only)
    const double2* x_sta;   // input of the factor being started
    double* ge_fin;         // gradient record of the finished factor's exponential (trajectory 0, replica 0)
    double* ge_sta;
    long ge_bstride, ge_rstride;
    double cb_fin_r, cb_fin_i, cb_sta_r, cb_sta_i;  // un-conjugated beta of the two factors (contraction weights)
    double* wtot;           // optional U_ij-gradient accumulator [dim]
    // fused cotangent injection (adjoint mode): the vector the finish stage completes is the cotangent at a save point k whose
    // state is x_fin; add  grad_states[k] + 2 sum_o grad_expect[o][k][b] obs[o][x] x_fin[x]  (replaces k_inject launches and the
    // host-side decision whether one is needed)
    const double2* inj_gstate;  // grad_states[k] ([B][dim]) or nullptr
    const double* inj_gexp;     // &grad_expect[0][k][0] or nullptr
    const double* inj_obs;      // [n_obs][dim]
    int inj_n_obs;
    long inj_ostride;           // n_tsave * B
    // STATE-SHARDED run (forward passes only): the vectors are SLABS of 2^nl amplitudes,
    // trajectory b of the launch is the slab of rank sh_rank_first + b (the rank id = the top sh_bits bits of the global
    // amplitude index).  The tile geometry is that of the nl local qubits; the diagonal is evaluated at the GLOBAL index; the
    // flips of the sh_bits "rank qubits" are the partner ranks' slabs, added — times beta * (c or conj c) — by the launch
    // that completes a factor.
    int sh_bits;                       // 0: not sharded
    int sh_nl;
    int sh_rank_first;
    int sh_self;                       // 1: every rank lives on this device, partner of trajectory b for rank bit k is trajectory b ^ (1 << k) of `u`
    const double2* sh_rem[kShardMaxBits];  // sh_self = 0: slab received from the partner of rank bit k (holds ITS complete v_{j-1}), trajectory 0
    int sh_grp[kShardMaxBits];         // flip group (coefficient record slot) of the qubit behind rank bit k, -1: not driven
    long obs_bstride, obs_ostride;     // observable table: [n_obs][dim] (0, dim), sharded: one slab per rank [n_obs][ranks][2^nl] (dim, ranks * dim)
    // fused expectation values of the COMPLETE vector produced by the finish stage (forward mode, step ends)
    const double* obs;      // [n_obs][dim] or nullptr
    double* expect_slot;    // &expect_out[0][k][0]
    int n_obs;
    long exp_ostride;       // n_tsave * B
};

// sum over the workgroup, then one atomic into the replica slot
template <int NT>
__device__ __forceinline__ void wg_atomic_add(double v, double* dst, double* red /* >= NT/64 doubles */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s2 = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) s2 += red[w];
        unsafeAtomicAdd(dst, s2);
    }
}

// Deferred gradient reductions of the adjoint pass: every per-thread contribution is reduced over its wave right away and
// parked per wave in an LDS slot (no barrier); ONE barrier at the end of the kernel, then thread s sums slot s over the
// waves and issues the atomic.  (A workgroup reduction per value costs two barriers and ~1.8 k cycles; the adjoint pass has
// five of them for one amplitude and one detuning group: 9 k of its 52 k cycles.)
template <int NW>
__device__ __forceinline__ void park2(double v0, double v1, double* red, int slot0, int slot1) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {  // two independent shuffle chains interleave
        v0 += __shfl_down(v0, off, 64);
        v1 += __shfl_down(v1, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {  // (+=: a slot may take several contributions per launch; each wave owns its entry)
        red[slot0 * NW + (threadIdx.x >> 6)] += v0;
        red[slot1 * NW + (threadIdx.x >> 6)] += v1;
    }
}

template <int NW>
__device__ __forceinline__ void park1(double v0, double* red, int slot0) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v0 += __shfl_down(v0, off, 64);
    if ((threadIdx.x & 63) == 0) red[slot0 * NW + (threadIdx.x >> 6)] += v0;
}

// partner sums over the tile bits in `mask`:  ts[r] = sum of partners, ds[r] = sum(+partner if own bit set else -partner)
// COND (one-launch kernels only: the tile IS the register, tile bit = amplitude-index bit): conditioned flips of a three-level
// register (RydProblem.amp_conditioned_terms) — the flip of bit b counts only for amplitudes whose sibling bit b ^ 1 is 1.
template <int LT, int LGT, bool CPLX, bool FULL = false, bool COND = false>  // FULL: every tile bit is in the mask (no per-bit tests)
__device__ __forceinline__ void partner_sums(const double2* __restrict__ tile, const double2 (&reg)[1 << (LT - LGT)],
                                             uint32_t mask, unsigned tid, double2 (&ts)[1 << (LT - LGT)],
                                             double2 (&ds)[1 << (LT - LGT)]) {
    constexpr int NT = 1 << LGT, R = 1 << (LT - LGT);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        ts[r] = make_double2(0.0, 0.0);
        ds[r] = make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int b = 0; b < LT; ++b) {
        if (FULL || (mask >> b & 1u)) {  // wave-uniform
            if (b < LGT) {
                const double sgn = (tid >> b & 1u) ? 1.0 : -1.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (COND && !((unsigned(r) * NT + tid) >> (b ^ 1) & 1u)) continue;
                    const double2 q = tile[(unsigned(r) * NT + tid) ^ (1u << b)];
                    ts[r].x += q.x;
                    ts[r].y += q.y;
                    if (CPLX) {
                        ds[r].x = fma(sgn, q.x, ds[r].x);
                        ds[r].y = fma(sgn, q.y, ds[r].y);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (COND && !((unsigned(r) * NT + tid) >> (b ^ 1) & 1u)) continue;
                    const double2 q = reg[r ^ (1 << (b - LGT))];
                    ts[r].x += q.x;
                    ts[r].y += q.y;
                    if (CPLX) {
                        if (r >> (b - LGT) & 1) {
                            ds[r].x += q.x;
                            ds[r].y += q.y;
                        } else {
                            ds[r].x -= q.x;
                            ds[r].y -= q.y;
                        }
                    }
                }
            }
        }
    }
}

// ---- registers of ONE wave as the tile (2^LT <= 64 amplitudes, one per lane): partners come through DPP lane exchanges
// (bits 0..3: quad permutes / row rotates, no LDS at all) or ds_bpermute (bits 4, 5) — no tile write, no barrier.
template <int CTRL, int BANK>
__device__ __forceinline__ int dpp_mov(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, BANK, false);
}

// every destination lane has a valid source lane in these patterns, so the destination needs no initial value
template <int CTRL, int BANK>
__device__ __forceinline__ int dpp_mov_all(int src) {
    return __builtin_amdgcn_mov_dpp(src, CTRL, 0xF, BANK, true);
}

template <int B>
__device__ __forceinline__ int lane_xor_i32(int w) {
    if constexpr (B == 0) return dpp_mov_all<0xB1, 0xF>(w);        // quad_perm [1,0,3,2]
    else if constexpr (B == 1) return dpp_mov_all<0x4E, 0xF>(w);   // quad_perm [2,3,0,1]
    else if constexpr (B == 2) {                                    // lanes with bit 2 set take lane-4 (row_ror:4), the others lane+4 (row_ror:12)
        const int r = dpp_mov_all<0x124, 0xA>(w);
        return dpp_mov<0x12C, 0x5>(r, w);
    } else if constexpr (B == 3) return dpp_mov_all<0x128, 0xF>(w);  // row_ror:8
    else return __shfl_xor(w, 1 << B, 64);
}

template <int B>
__device__ __forceinline__ double2 lane_xor(const double2& v) {
    const int a0 = lane_xor_i32<B>(__double2loint(v.x)), a1 = lane_xor_i32<B>(__double2hiint(v.x));
    const int b0 = lane_xor_i32<B>(__double2loint(v.y)), b1 = lane_xor_i32<B>(__double2hiint(v.y));
    return make_double2(__hiloint2double(a1, a0), __hiloint2double(b1, b0));
}

template <int LT, bool CPLX, bool FULL = false, bool COND = false>
__device__ __forceinline__ void partner_sums_lanes(const double2& own, uint32_t mask, unsigned tid, double2& ts, double2& ds) {
    static_assert(LT <= 6, "one amplitude per lane");
    ts = make_double2(0.0, 0.0);
    ds = make_double2(0.0, 0.0);
    auto bit = [&](auto bc) {
        constexpr int b = decltype(bc)::value;
        if constexpr (b < LT) {
            if (FULL || (mask >> b & 1u)) {  // wave-uniform
                double2 q = lane_xor<b>(own);  // (every lane takes part in the exchange; the condition only masks the sum)
                if (COND && !(tid >> (b ^ 1) & 1u)) q = make_double2(0.0, 0.0);
                ts.x += q.x;
                ts.y += q.y;
                if (CPLX) {
                    const double sgn = (tid >> b & 1u) ? 1.0 : -1.0;
                    ds.x = fma(sgn, q.x, ds.x);
                    ds.y = fma(sgn, q.y, ds.y);
                }
            }
        }
    };
    bit(std::integral_constant<int, 0>{});
    bit(std::integral_constant<int, 1>{});
    bit(std::integral_constant<int, 2>{});
    bit(std::integral_constant<int, 3>{});
    bit(std::integral_constant<int, 4>{});
    bit(std::integral_constant<int, 5>{});
}

#ifdef RYDIFF_TIMELINE
__device__ unsigned long long g_timeline[4096 * 8];  // tuning builds: per-workgroup phase timestamps of the last launch
#define RYDIFF_TL(slot)                                                                                   \
    do {                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096 && a.has_p && a.has_q) g_timeline[blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define RYDIFF_TL(slot)
#endif

// FAST: exactly one amplitude group, every tile bit in its start-stage mask and no partner-tile loads (a global drive): straight-line
// code instead of the runtime group loops, no mask tests in the start stage.  The first detuning group is straight-line too; further
// ones (local detuning channels next to the global drive) cost a uniform loop over the diagonal only.
// RES: the vectors of a trajectory are meant to STAY in the XCD's L2 (trajectory-per-XCD placement): plain loads / stores instead
// of the streaming (non-temporal) ones.  A template parameter on purpose: selecting the access flavour at run time made the
// compiler merge both flavours into plain accesses and the 20-qubit pass lost its streaming hints (15.4 vs 13.9 us).
// REC (adjoint of one global phase-free drive, BWD && FAST && !CPLX): the tape vector of a factor is read ONCE, by the launch that
// completes the factor's adjoint.  With mu' = (gamma~ + beta~ (d + c F)) mu that launch holds mu (u), mu' (acc) and x (x_fin), so
//     Re(beta <F mu, x>) = Re <mu' - (gamma~ + beta~ d) mu, x> / c                (elementwise: no partner sums, no second layout)
// and the detuning / U_ij weights Re(beta conj(mu) x) are elementwise in (u, x_fin) as well.  The start stage then is the forward
// start stage: the launch moves 3R + 2W instead of 4R + 2W.  The quotient loses log2(1 / |beta c|) bits, so a factor whose
// |beta c| is below kRecMinBetaC (amplitude ~0: pulse edges, the padded last sample) keeps the exact contraction of the partner
// sums with the tape vector in BOTH of its launches; the two launches of a factor take the same branch because they test the same
// record and the same scalars.
constexpr double kRecMinBetaC = 6.0e-5;  // 2^-14: at most 14 of 53 bits lost in the recovered contraction

template <int LT, int LGT, bool CPLX, bool BWD, bool FAST = false, bool RES = false>
__global__ __launch_bounds__(1 << LGT) void k_chain(ChainArgs a) {
    constexpr int NT = 1 << LGT, R = 1 << (LT - LGT);
    constexpr bool REC = BWD && FAST && !CPLX;
    const int GA = FAST ? 1 : a.ga;            // amplitude groups looped over
    const int GD = FAST ? (a.gd ? 1 : 0) : a.gd;
    extern __shared__ __attribute__((aligned(16))) double2 tile[];
    // behind the tile: NT/64 doubles (forward: fused expectation) or [4 ga + gd][NT/64] parked gradient partials (adjoint)
    double* red = reinterpret_cast<double*>(tile + (size_t(1) << LT));
    constexpr int NW = NT / 64;
    const int n_slots = BWD ? 4 * a.ga + a.gd : 0;  // fin: (re, im) per group | sta: det per group | sta: (re, im) per group
    if (BWD) {
        for (int s = int(threadIdx.x); s < n_slots * NW; s += NT) red[s] = 0.0;  // published by the barrier after the tile write
    }
    auto flush_gradients = [&](double* ge_fin_, double* ge_sta_) {  // (placed after the early exit of ragged groups: uniform per workgroup)
        __syncthreads();
        for (int s = int(threadIdx.x); s < n_slots; s += NT) {
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += red[s * NW + w];
            double* dst;
            if (s < 2 * a.ga) dst = ge_fin_ + ((s & 1) ? a.ga : 0) + (s >> 1);
            else if (s < 2 * a.ga + a.gd) dst = (REC ? ge_fin_ : ge_sta_) + 2 * a.ga + (s - 2 * a.ga);
            else dst = ge_sta_ + (((s - 2 * a.ga - a.gd) & 1) ? a.ga : 0) + ((s - 2 * a.ga - a.gd) >> 1);
            if (sum != 0.0) unsafeAtomicAdd(dst, sum);
        }
    };
    const unsigned tid = threadIdx.x;
    unsigned t = a.xcd_place ? (blockIdx.x >> 3) : blockIdx.x;                             // tile of the trajectory
    if (a.tile_swz) {
        const unsigned w = blockIdx.x, G = 1u << a.tile_swz;
        t = (w & ~(8u * G - 1u)) | ((w & 7u) << a.tile_swz) | ((w >> 3) & (G - 1u));
    }
    const unsigned bl = a.xcd_place ? (blockIdx.y * 8u + (blockIdx.x & 7u)) : blockIdx.y;  // trajectory within the launch
    if (bl >= unsigned(a.b_count)) return;                                                 // ragged last row of 8
    const unsigned bt = unsigned(a.b_first) + bl;                                          // trajectory
    const size_t boff = size_t(bt) * a.dim;
    const unsigned rank = unsigned(a.sh_rank_first) + bl;                 // sharded runs: this slab's rank id ...
    const unsigned rank_hi = a.sh_bits ? (rank << a.sh_nl) : 0u;           // ... = the top bits of the global amplitude index
    const unsigned t_glob = a.sh_bits ? (t + (rank << (a.sh_nl - LT))) : t;  // row of the (global) split-diagonal table
    auto ld = [&](const double2* p) -> double2 {
        if constexpr (RES) return *p;
        else return stream_load(p);
    };
    auto st = [&](double2* p, const double2& v) {
        if constexpr (RES) *p = v;
        else stream_store(p, v);
    };
    const unsigned lomask = (1u << a.lo) - 1u;
    const int midlow = a.hs - a.lo;
    const unsigned xbase = ((t & ((1u << midlow) - 1u)) << a.lo) | ((t >> midlow) << (a.hs + a.hb));

#ifdef RYDIFF_STAGGER
    // de-synchronise the workgroups' load / store phases (all tiles are co-resident and start together)
    for (unsigned sl = 0; sl < (blockIdx.x % RYDIFF_STAGGER_PHASES); ++sl) __builtin_amdgcn_s_sleep(RYDIFF_STAGGER);
#endif
    RYDIFF_TL(0);
    double2 uu[R], acc[R];
    unsigned xg[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned i = unsigned(r) * NT + tid;
        xg[r] = xbase | (i & lomask) | ((i >> a.lo) << a.hs);
        uu[r] = ld(a.u + boff + xg[r]);
    }
    // Loads OUTSIDE of control flow (the host passes valid pointers even for a chain's first / last launch, where the values are
    // not used): the compiler then places the wait counters per use — the tile write waits for u only, the partial is awaited
    // where it is accumulated, after the finish stage's partner sums — instead of conservatively at the join of the branches
    // (measured on the 20-qubit pass: 14.16 -> 13.91 us forward, fwd+grad +2.9 %; profiles/r02_uncond_loads_ab.txt).
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = ld(a.p + boff + xg[r]);
    // sharded runs: flips of the rank qubits = the partner slabs' complete v_{j-1} at the same local index, times
    // beta * (c or conj c).  Requested together with u and p so that all of a tile's global loads are in flight at once.
    // (adjoint mode: u is the cotangent, fb = conj(beta): the same sum is the rank-qubit part of (gamma~ + beta~ H) mu)
    double2 remacc[R];
    if (a.has_p && a.completes && a.sh_bits) {
#pragma unroll
        for (int r = 0; r < R; ++r) remacc[r] = make_double2(0.0, 0.0);
        const double* __restrict__ cfs = a.coef_fin + bt * a.coef_bstride;
        for (int k = 0; k < a.sh_bits; ++k) {
            if (a.sh_grp[k] < 0) continue;
            const double cr = cfs[a.sh_grp[k]], ci = (rank >> k & 1u) ? cfs[a.ga + a.sh_grp[k]] : -cfs[a.ga + a.sh_grp[k]];  // row g: c, row r: conj c
            const double kr = a.fb_r * cr - a.fb_i * ci, ki = a.fb_r * ci + a.fb_i * cr;
            const double2* __restrict__ src = a.sh_self ? a.u + size_t(unsigned(a.b_first) + (bl ^ (1u << k))) * a.dim : a.sh_rem[k] + size_t(bl) * a.dim;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double2 pv = src[xg[r]];  // plain load: with every rank on one device each slab is read by its 3 partners too
                remacc[r].x += kr * pv.x - ki * pv.y;
                remacc[r].y += kr * pv.y + ki * pv.x;
            }
        }
    }
    double dg[R];  // tile-local part of the interaction diagonal (32 KiB table shared by all tiles: L2-resident)
#pragma unroll
    for (int r = 0; r < R; ++r) dg[r] = a.utt[unsigned(r) * NT + tid];
    double2 xf[R], xs[R];
    // the real-drive adjoint (no signed sums) has the registers to request the second tape vector up front as well; REC reads it
    // only for the rare factor that keeps the exact contraction (below)
    constexpr bool XS_EARLY = BWD && !CPLX && !REC;
    // REC: which scheme the finished / the started factor uses (wave-uniform; NaN-safe: !(>=) is the exact scheme)
    bool exact_fin = true, exact_sta = true;
    if constexpr (REC) {
        exact_fin = !(fabs(a.coef_fin[bt * a.coef_bstride]) * sqrt(a.fb_r * a.fb_r + a.fb_i * a.fb_i) >= kRecMinBetaC);
        exact_sta = !(fabs(a.coef_sta[bt * a.coef_bstride]) * sqrt(a.sb_r * a.sb_r + a.sb_i * a.sb_i) >= kRecMinBetaC);
    }
    if (BWD) {
#pragma unroll
        for (int r = 0; r < R; ++r) xf[r] = stream_load(a.x_fin + boff + xg[r]);
    }
    if (XS_EARLY) {
#pragma unroll
        for (int r = 0; r < R; ++r) xs[r] = stream_load(a.x_sta + boff + xg[r]);
    }
#ifndef RYDIFF_ABLATE_SYNC
#pragma unroll
    for (int r = 0; r < R; ++r) tile[unsigned(r) * NT + tid] = uu[r];
    __syncthreads();
#endif
    RYDIFF_TL(1);
    double* ge_fin = nullptr;
    double* ge_sta = nullptr;
    if (BWD) {
        const long goff = bt * a.ge_bstride + (blockIdx.x % kGradReplicas) * a.ge_rstride;
        ge_fin = a.ge_fin + goff;
        ge_sta = a.ge_sta + goff;
    }

    // interaction diagonal: tile-local table + remote part of this tile + cross terms of the tile bits that are in |r> (n = 1 - bit)
    auto interaction_diagonal = [&](double (&du)[R]) {
        double vloc[LT];
        const double* __restrict__ vrow = a.vr + size_t(t_glob) * 16;
#ifdef RYDIFF_ABLATE_COEF
        for (int b2 = 0; b2 < LT; ++b2) vloc[b2] = a.sb_r;
        double dlane = a.sb_i;
        (void)vrow;
#else
#pragma unroll
        for (int b2 = 0; b2 < LT; ++b2) vloc[b2] = vrow[b2];
        double dlane = vrow[LT];
#endif
#pragma unroll
        for (int b2 = 0; b2 < LGT; ++b2)
            if (!(tid >> b2 & 1u)) dlane += vloc[b2];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double d = dg[r] + dlane;
#pragma unroll
            for (int b2 = LGT; b2 < LT; ++b2)
                if (!(r >> (b2 - LGT) & 1)) d += vloc[b2];
            du[r] = d;
        }
    };
    double du[R];
    if constexpr (REC) interaction_diagonal(du);  // the finish stage needs it too

    if (a.has_p) {
        const double* __restrict__ cf = a.coef_fin + bt * a.coef_bstride;
        for (int g = 0; g < GA; ++g) {
            const uint32_t mask = a.fin_mask[g];
            if (!mask) continue;
            double2 ts[R], ds[R];
#ifndef RYDIFF_ABLATE_COMPUTE
            if (!FAST && (a.cond >> g & 1u)) partner_sums<LT, LGT, CPLX, false, true>(tile, uu, mask, tid, ts, ds);
            else partner_sums<LT, LGT, CPLX>(tile, uu, mask, tid, ts, ds);
#else
            for (int r = 0; r < R; ++r) { ts[r] = uu[r]; ds[r] = uu[r]; }
#endif
#ifdef RYDIFF_ABLATE_COEF
            const double cr = a.sg_r, ci = a.sg_i;
#else
            const double cr = cf[g], ci = cf[a.ga + g];
#endif
            // c*s1 + conj(c)*s0 = cr*(s1+s0) + i*ci*(s1-s0);  k1 = beta*cr, k2 = beta*i*ci
            const double k1r = a.fb_r * cr, k1i = a.fb_i * cr;
            const double k2r = -a.fb_i * ci, k2i = a.fb_r * ci;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r].x += k1r * ts[r].x - k1i * ts[r].y;
                acc[r].y += k1r * ts[r].y + k1i * ts[r].x;
                if (CPLX) {
                    acc[r].x += k2r * ds[r].x - k2i * ds[r].y;
                    acc[r].y += k2r * ds[r].y + k2i * ds[r].x;
                }
            }
            if (BWD && exact_fin) {
                // <F mu, x> with F Hermitian: z1 = sum conj(ts) x, z2 = sum conj(ds) x; dL/dRe c = Re(beta z1), dL/dIm c = Im(beta z2)
                // (CPLX = false in the adjoint: real coefficients and a caller that only wants dL/dRe c — no signed sums at all)
                double z1r = 0.0, z1i = 0.0, z2r = 0.0, z2i = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    z1r += ts[r].x * xf[r].x + ts[r].y * xf[r].y;
                    z1i += ts[r].x * xf[r].y - ts[r].y * xf[r].x;
                    if (CPLX) {
                        z2r += ds[r].x * xf[r].x + ds[r].y * xf[r].y;
                        z2i += ds[r].x * xf[r].y - ds[r].y * xf[r].x;
                    }
                }
                if (CPLX) park2<NW>(a.cb_fin_r * z1r - a.cb_fin_i * z1i, a.cb_fin_r * z2i + a.cb_fin_i * z2r, red, 2 * g, 2 * g + 1);
                else park1<NW>(a.cb_fin_r * z1r - a.cb_fin_i * z1i, red, 2 * g);
            }
        }
        if (a.completes && a.sh_bits) {  // rank-qubit flips: partner slabs (forward: of v_{j-1}; adjoint: of the cotangent)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r].x += remacc[r].x;
                acc[r].y += remacc[r].y;
            }
            if (BWD && exact_fin) {
                // exact drive gradients of the rank qubits: <F_k mu, x> with (F_k mu)(x) = the partner slab's mu at the same local
                // index; plain sum z1 and signed sum z2 (sign + where the own rank bit is set), as for the tile bits above.  The
                // partner values are re-read (L2 hits: requested a moment ago) rather than kept, so that the un-sharded adjoint —
                // the headline path — pays no registers for this.
                for (int k = 0; k < a.sh_bits; ++k) {
                    const int g = a.sh_grp[k];
                    if (g < 0) continue;
                    const double2* __restrict__ src = a.sh_self ? a.u + size_t(unsigned(a.b_first) + (bl ^ (1u << k))) * a.dim : a.sh_rem[k] + size_t(bl) * a.dim;
                    const double sgn = (rank >> k & 1u) ? 1.0 : -1.0;
                    double z1r = 0.0, z1i = 0.0;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double2 pv = src[xg[r]];
                        z1r += pv.x * xf[r].x + pv.y * xf[r].y;
                        z1i += pv.x * xf[r].y - pv.y * xf[r].x;
                    }
                    if (CPLX) park2<NW>(a.cb_fin_r * z1r - a.cb_fin_i * z1i, sgn * (a.cb_fin_r * z1i + a.cb_fin_i * z1r), red, 2 * g, 2 * g + 1);
                    else park1<NW>(a.cb_fin_r * z1r - a.cb_fin_i * z1i, red, 2 * g);
                }
            }
        }
        if constexpr (REC) {
            if (a.completes) {  // acc = mu' (before any injected cotangent), uu = mu, xf = the factor's input
                // weights of d(x) in the gradient, Re(beta conj(mu) x): detuning group + U_ij accumulator
                double sgd = 0.0, zr = 0.0;
                const double cdet = GD ? cf[2] : 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double pr = a.cb_fin_r * uu[r].x + a.cb_fin_i * uu[r].y, pi = a.cb_fin_i * uu[r].x - a.cb_fin_r * uu[r].y;
                    const double w = pr * xf[r].x - pi * xf[r].y;
                    if (a.wtot) unsafeAtomicAdd(a.wtot + (a.sh_bits ? boff : 0) + xg[r], w);
                    const double cnt = GD ? double(a.dcnt[0] - popc_i((xg[r] | rank_hi) & a.dmask[0])) : 0.0;
                    sgd += w * cnt;
                    if (!exact_fin) {  // Re <mu' - (gamma~ + beta~ d) mu, x>
                        double d = du[r] + cdet * cnt;
                        for (int g = 1; g < a.gd; ++g) d += cf[2 + g] * double(a.dcnt[g] - popc_i((xg[r] | rank_hi) & a.dmask[g]));
                        const double dr = a.fg_r + a.fb_r * d, di = a.fg_i + a.fb_i * d;
                        const double wx = acc[r].x - (dr * uu[r].x - di * uu[r].y), wy = acc[r].y - (dr * uu[r].y + di * uu[r].x);
                        zr += wx * xf[r].x + wy * xf[r].y;
                    }
                }
                if (GD) park1<NW>(sgd, red, 2 * a.ga);
                for (int g = 1; g < a.gd; ++g) {  // further detuning groups next to the one global drive (local detuning channels)
                    double sg = 0.0;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double pr = a.cb_fin_r * uu[r].x + a.cb_fin_i * uu[r].y, pi = a.cb_fin_i * uu[r].x - a.cb_fin_r * uu[r].y;
                        sg += (pr * xf[r].x - pi * xf[r].y) * double(a.dcnt[g] - popc_i((xg[r] | rank_hi) & a.dmask[g]));
                    }
                    park1<NW>(sg, red, 2 * a.ga + g);
                }
                if (!exact_fin) park1<NW>(zr / cf[0], red, 0);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = uu[r];
    }
    if (BWD && a.has_p && (a.inj_gexp || a.inj_gstate)) {  // wave-uniform: the completed cotangent sits at a save point
        if (a.inj_gexp) {
            bool any = false;
            for (int o = 0; o < a.inj_n_obs; ++o) any |= a.inj_gexp[o * a.inj_ostride + bt] != 0.0;
            if (any) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    double wsum = 0.0;
                    for (int o = 0; o < a.inj_n_obs; ++o)
                        wsum += a.inj_gexp[o * a.inj_ostride + bt] * a.inj_obs[size_t(o) * a.obs_ostride + bt * a.obs_bstride + xg[r]];
                    acc[r].x += 2.0 * wsum * xf[r].x;
                    acc[r].y += 2.0 * wsum * xf[r].y;
                }
            }
        }
        if (a.inj_gstate) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double2 gs = stream_load(a.inj_gstate + boff + xg[r]);
                acc[r].x += gs.x;
                acc[r].y += gs.y;
            }
        }
    }
    if (BWD && !XS_EARLY && a.has_q && (!REC || exact_sta)) {  // issued here (not at the top) to stay inside the register budget of 1024-thread tiles
#pragma unroll
        for (int r = 0; r < R; ++r) xs[r] = stream_load(a.x_sta + boff + xg[r]);
    }
    RYDIFF_TL(2);
    if (a.write_v) {
#pragma unroll
        for (int r = 0; r < R; ++r) st(a.v_out + boff + xg[r], acc[r]);
    }
    RYDIFF_TL(3);
    if (!BWD && a.obs) {  // <v|O|v> for diagonal observables, straight from the registers that hold v
        for (int o = 0; o < a.n_obs; ++o) {
            double e = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) e += a.obs[size_t(o) * a.obs_ostride + bt * a.obs_bstride + xg[r]] * (acc[r].x * acc[r].x + acc[r].y * acc[r].y);
            wg_atomic_add<NT>(e, a.expect_slot + o * a.exp_ostride + bt, red);
        }
    }
    if (!a.has_q) {
        if (BWD) flush_gradients(ge_fin, ge_sta);
        return;
    }

#ifndef RYDIFF_ABLATE_SYNC
    __syncthreads();  // all partner reads of u are done
#pragma unroll
    for (int r = 0; r < R; ++r) tile[unsigned(r) * NT + tid] = acc[r];
    __syncthreads();
#endif
    RYDIFF_TL(4);

    const double* __restrict__ cf = a.coef_sta + bt * a.coef_bstride;
    if constexpr (!REC) interaction_diagonal(du);
    double rr[R];  // Re(beta conj(mu) x): weight of d(x) in the gradient (REC: taken by the launch that completes the factor)
    if (BWD && !REC) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double pr = a.cb_sta_r * acc[r].x + a.cb_sta_i * acc[r].y, pi = a.cb_sta_i * acc[r].x - a.cb_sta_r * acc[r].y;
            rr[r] = pr * xs[r].x - pi * xs[r].y;
            if (a.wtot) unsafeAtomicAdd(a.wtot + (a.sh_bits ? boff : 0) + xg[r], rr[r]);
        }
        for (int g = 0; g < a.gd; ++g) {
            double sgd = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) sgd += rr[r] * double(a.dcnt[g] - popc_i((xg[r] | rank_hi) & a.dmask[g]));
            park1<NW>(sgd, red, 2 * a.ga + g);
        }
    }
    double2 q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double d = du[r];
        if (FAST) {  // one global drive: the first detuning group straight-line, further ones (local detuning channels) in a uniform loop
            if (GD) d += cf[2] * double(a.dcnt[0] - popc_i((xg[r] | rank_hi) & a.dmask[0]));
            for (int g = 1; g < a.gd; ++g) d += cf[2 + g] * double(a.dcnt[g] - popc_i((xg[r] | rank_hi) & a.dmask[g]));
        } else {
            for (int g = 0; g < a.gd; ++g) d += cf[2 * a.ga + g] * double(a.dcnt[g] - popc_i((xg[r] | rank_hi) & a.dmask[g]));
        }
        const double dr = a.sg_r + a.sb_r * d, di = a.sg_i + a.sb_i * d;
        q[r].x = dr * acc[r].x - di * acc[r].y;
        q[r].y = dr * acc[r].y + di * acc[r].x;
    }
    for (int g = 0; g < GA; ++g) {
        const uint32_t mask = a.sta_mask[g];
        if (!FAST && !mask) continue;
        double2 ts[R], ds[R];
#ifndef RYDIFF_ABLATE_COMPUTE
        if (!FAST && (a.cond >> g & 1u)) partner_sums<LT, LGT, CPLX, false, true>(tile, acc, mask, tid, ts, ds);
        else partner_sums<LT, LGT, CPLX, FAST>(tile, acc, mask, tid, ts, ds);
#else
        for (int r = 0; r < R; ++r) { ts[r] = acc[r]; ds[r] = acc[r]; }
#endif
#ifdef RYDIFF_ABLATE_COEF
        const double cr = a.sg_r, ci = a.sg_i;
#else
        const double cr = cf[g], ci = cf[a.ga + g];
#endif
        const double k1r = a.sb_r * cr, k1i = a.sb_i * cr;
        const double k2r = -a.sb_i * ci, k2i = a.sb_r * ci;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            q[r].x += k1r * ts[r].x - k1i * ts[r].y;
            q[r].y += k1r * ts[r].y + k1i * ts[r].x;
            if (CPLX) {
                q[r].x += k2r * ds[r].x - k2i * ds[r].y;
                q[r].y += k2r * ds[r].y + k2i * ds[r].x;
            }
        }
        if (BWD && exact_sta) {
            double z1r = 0.0, z1i = 0.0, z2r = 0.0, z2i = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                z1r += ts[r].x * xs[r].x + ts[r].y * xs[r].y;
                z1i += ts[r].x * xs[r].y - ts[r].y * xs[r].x;
                if (CPLX) {
                    z2r += ds[r].x * xs[r].x + ds[r].y * xs[r].y;
                    z2i += ds[r].x * xs[r].y - ds[r].y * xs[r].x;
                }
            }
            if (CPLX) park2<NW>(a.cb_sta_r * z1r - a.cb_sta_i * z1i, a.cb_sta_r * z2i + a.cb_sta_i * z2r, red, 2 * a.ga + a.gd + 2 * g,
                                2 * a.ga + a.gd + 2 * g + 1);
            else park1<NW>(a.cb_sta_r * z1r - a.cb_sta_i * z1i, red, 2 * a.ga + a.gd + 2 * g);
        }
    }
    RYDIFF_TL(5);
#pragma unroll
    for (int r = 0; r < R; ++r) st(a.q_out + boff + xg[r], q[r]);
    if (BWD) flush_gradients(ge_fin, ge_sta);
    RYDIFF_TL(6);
#ifdef RYDIFF_TIMELINE
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the stores of this wave have been acknowledged
    RYDIFF_TL(7);
#endif
}

// ---- wide tiles: 2^(LGT+3) amplitudes per 1024-thread workgroup, processed in two register halves --------------------------
// Same launch contract as k_chain (ChainArgs; forward and adjoint passes).  A tile of 2^13 amplitudes (128 KiB of the 160 KiB LDS)
// gives the second tile layout runs of 512 / 256 / 128 / 64 bytes at 21 / 22 / 23 / 24 qubits where 2^12-amplitude tiles have
// 128 / 64 / 32 / 16 — so two layouts (2R+2W per factor) reach 24 qubits.  k_chain instantiated with 8 amplitudes per thread keeps
// u, the accumulator, both partner sums and the output of ALL eight in registers and spills (48-328 VGPRs:
// profiles/r03_tile_size_and_line_sharing.txt).  Here only the accumulator (the vector the finish stage completes) stays in
// registers for all eight; u lives in LDS only (own element re-read where the adjoint needs it), and partner sums, tape values,
// diagonal and output exist for FOUR amplitudes at a time.  In the finish stage every partner comes from LDS, in the start stage
// the three register bits are register renaming as in k_chain.  No trajectory-per-XCD placement (registers of this size are never
// L2-resident).  The arithmetic per amplitude is that of k_chain, statement by statement.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// RH: amplitudes per thread that are worked on together (R / RH passes over the register file: halves or quarters)
template <int LT, bool CPLX, bool BWD, bool FAST, int RH = (1 << (LT - 10)) / 2>
__global__ __launch_bounds__(1024) void k_chain_wide(ChainArgs a) {
    constexpr int LGT = 10, NT = 1 << LGT, R = 1 << (LT - LGT), NP = R / RH, NW = NT / 64;
    static_assert(R >= 2 && RH >= 1 && NP * RH == R, "R amplitudes per thread in NP parts of RH");
    constexpr bool REC = BWD && FAST && !CPLX;
    const int GA = FAST ? 1 : a.ga;
    const int GD = FAST ? (a.gd ? 1 : 0) : a.gd;
    extern __shared__ __attribute__((aligned(16))) double2 tile[];
    double* red = reinterpret_cast<double*>(tile + (size_t(1) << LT));
    const int n_slots = BWD ? 4 * a.ga + a.gd : 0;
    if (BWD) {
        for (int s = int(threadIdx.x); s < n_slots * NW; s += NT) red[s] = 0.0;  // published by the barrier after the tile write
    }
    const unsigned tid = threadIdx.x;
    unsigned t = blockIdx.x;
    if (a.tile_swz) {
        const unsigned w = blockIdx.x, G = 1u << a.tile_swz;
        t = (w & ~(8u * G - 1u)) | ((w & 7u) << a.tile_swz) | ((w >> 3) & (G - 1u));
    }
    const unsigned bl = blockIdx.y;
    const unsigned bt = unsigned(a.b_first) + bl;
    const size_t boff = size_t(bt) * a.dim;
    const unsigned rank = unsigned(a.sh_rank_first) + bl;
    const unsigned rank_hi = a.sh_bits ? (rank << a.sh_nl) : 0u;
    const unsigned t_glob = a.sh_bits ? (t + (rank << (a.sh_nl - LT))) : t;
    const unsigned lomask = (1u << a.lo) - 1u;
    const int midlow = a.hs - a.lo;
    const unsigned xbase = ((t & ((1u << midlow) - 1u)) << a.lo) | ((t >> midlow) << (a.hs + a.hb));
    auto xg_of = [&](int r) -> unsigned {
        const unsigned i = unsigned(r) * NT + tid;
        return xbase | (i & lomask) | ((i >> a.lo) << a.hs);
    };
    double* ge_fin = nullptr;
    double* ge_sta = nullptr;
    if (BWD) {
        const long goff = bt * a.ge_bstride + (blockIdx.x % kGradReplicas) * a.ge_rstride;
        ge_fin = a.ge_fin + goff;
        ge_sta = a.ge_sta + goff;
    }
    auto flush_gradients = [&]() {
        __syncthreads();
        for (int s = int(threadIdx.x); s < n_slots; s += NT) {
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += red[s * NW + w];
            double* dst;
            if (s < 2 * a.ga) dst = ge_fin + ((s & 1) ? a.ga : 0) + (s >> 1);
            else if (s < 2 * a.ga + a.gd) dst = (REC ? ge_fin : ge_sta) + 2 * a.ga + (s - 2 * a.ga);
            else dst = ge_sta + (((s - 2 * a.ga - a.gd) & 1) ? a.ga : 0) + ((s - 2 * a.ga - a.gd) >> 1);
            if (sum != 0.0) unsafeAtomicAdd(dst, sum);
        }
    };
    // interaction diagonal of amplitude (rf, tid): tile-local table + remote part of this tile + cross terms of the tile bits in |r>
    const double* __restrict__ vrow = a.vr + size_t(t_glob) * 16;
    auto lane_diag = [&]() -> double {  // (computed where it is used: two registers less across the finish stage)
        double dlane = vrow[LT];
#pragma unroll
        for (int b2 = 0; b2 < LGT; ++b2)
            if (!(tid >> b2 & 1u)) dlane += vrow[b2];
        return dlane;
    };
    auto diag_of = [&](int rf, double dlane) -> double {
        double d = a.utt[unsigned(rf) * NT + tid] + dlane;
#pragma unroll
        for (int b2 = LGT; b2 < LT; ++b2)
            if (!(rf >> (b2 - LGT) & 1)) d += vrow[b2];
        return d;
    };
    bool exact_fin = true, exact_sta = true;
    if constexpr (REC) {
        exact_fin = !(fabs(a.coef_fin[bt * a.coef_bstride]) * sqrt(a.fb_r * a.fb_r + a.fb_i * a.fb_i) >= kRecMinBetaC);
        exact_sta = !(fabs(a.coef_sta[bt * a.coef_bstride]) * sqrt(a.sb_r * a.sb_r + a.sb_i * a.sb_i) >= kRecMinBetaC);
    }

    double2 acc[R];
    {
        double2 uu[RH];
#pragma unroll
        for (int h = 0; h < NP; ++h) {  // u goes straight to LDS, RH amplitudes at a time
#pragma unroll
            for (int r = 0; r < RH; ++r) uu[r] = stream_load(a.u + boff + xg_of(h * RH + r));
#pragma unroll
            for (int r = 0; r < RH; ++r) tile[unsigned(h * RH + r) * NT + tid] = uu[r];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = stream_load(a.p + boff + xg_of(r));  // (the host passes p = u for a chain's first launch)
    __syncthreads();

    bool inj_any = false;
    if (BWD && a.has_p && a.inj_gexp) {
        for (int o = 0; o < a.inj_n_obs; ++o) inj_any |= a.inj_gexp[o * a.inj_ostride + bt] != 0.0;
    }
    auto finish_half = [&](auto hc) {
        constexpr int h = decltype(hc)::value;
        const double* __restrict__ cf = a.coef_fin + bt * a.coef_bstride;
        double2 xf[RH];
        if (BWD) {
#pragma unroll
            for (int r = 0; r < RH; ++r) xf[r] = stream_load(a.x_fin + boff + xg_of(h * RH + r));
        }
        if (a.has_p) {
            for (int g = 0; g < GA; ++g) {
                const uint32_t mask = a.fin_mask[g];
                if (!mask) continue;
                double2 ts[RH], ds[RH];
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    ts[r] = make_double2(0.0, 0.0);
                    ds[r] = make_double2(0.0, 0.0);
                }
#pragma unroll
                for (int b = 0; b < LT; ++b) {
                    if (mask >> b & 1u) {  // wave-uniform
#pragma unroll
                        for (int r = 0; r < RH; ++r) {
                            const unsigned i = unsigned(h * RH + r) * NT + tid;
                            const double2 q = tile[i ^ (1u << b)];
                            ts[r].x += q.x;
                            ts[r].y += q.y;
                            if (CPLX) {
                                const double sgn = (i >> b & 1u) ? 1.0 : -1.0;
                                ds[r].x = fma(sgn, q.x, ds[r].x);
                                ds[r].y = fma(sgn, q.y, ds[r].y);
                            }
                        }
                    }
                }
                const double cr = cf[g], ci = cf[a.ga + g];
                const double k1r = a.fb_r * cr, k1i = a.fb_i * cr;
                const double k2r = -a.fb_i * ci, k2i = a.fb_r * ci;
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    double2& A = acc[h * RH + r];
                    A.x += k1r * ts[r].x - k1i * ts[r].y;
                    A.y += k1r * ts[r].y + k1i * ts[r].x;
                    if (CPLX) {
                        A.x += k2r * ds[r].x - k2i * ds[r].y;
                        A.y += k2r * ds[r].y + k2i * ds[r].x;
                    }
                }
                if (BWD && exact_fin) {
                    double z1r = 0.0, z1i = 0.0, z2r = 0.0, z2i = 0.0;
#pragma unroll
                    for (int r = 0; r < RH; ++r) {
                        z1r += ts[r].x * xf[r].x + ts[r].y * xf[r].y;
                        z1i += ts[r].x * xf[r].y - ts[r].y * xf[r].x;
                        if (CPLX) {
                            z2r += ds[r].x * xf[r].x + ds[r].y * xf[r].y;
                            z2i += ds[r].x * xf[r].y - ds[r].y * xf[r].x;
                        }
                    }
                    if (CPLX) park2<NW>(a.cb_fin_r * z1r - a.cb_fin_i * z1i, a.cb_fin_r * z2i + a.cb_fin_i * z2r, red, 2 * g, 2 * g + 1);
                    else park1<NW>(a.cb_fin_r * z1r - a.cb_fin_i * z1i, red, 2 * g);
                }
            }
            if (a.completes && a.sh_bits) {  // rank-qubit flips: the partner slabs' complete vector at the same local index
                for (int k = 0; k < a.sh_bits; ++k) {
                    const int g = a.sh_grp[k];
                    if (g < 0) continue;
                    const double cr = cf[g], ci = (rank >> k & 1u) ? cf[a.ga + g] : -cf[a.ga + g];  // row g: c, row r: conj c
                    const double kr = a.fb_r * cr - a.fb_i * ci, ki = a.fb_r * ci + a.fb_i * cr;
                    const double2* __restrict__ src = a.sh_self ? a.u + size_t(unsigned(a.b_first) + (bl ^ (1u << k))) * a.dim : a.sh_rem[k] + size_t(bl) * a.dim;
                    const double sgn = (rank >> k & 1u) ? 1.0 : -1.0;
                    double z1r = 0.0, z1i = 0.0;
#pragma unroll
                    for (int r = 0; r < RH; ++r) {
                        const double2 pv = src[xg_of(h * RH + r)];
                        acc[h * RH + r].x += kr * pv.x - ki * pv.y;
                        acc[h * RH + r].y += kr * pv.y + ki * pv.x;
                        if (BWD) {
                            z1r += pv.x * xf[r].x + pv.y * xf[r].y;
                            z1i += pv.x * xf[r].y - pv.y * xf[r].x;
                        }
                    }
                    if (BWD && exact_fin) {
                        if (CPLX) park2<NW>(a.cb_fin_r * z1r - a.cb_fin_i * z1i, sgn * (a.cb_fin_r * z1i + a.cb_fin_i * z1r), red, 2 * g, 2 * g + 1);
                        else park1<NW>(a.cb_fin_r * z1r - a.cb_fin_i * z1i, red, 2 * g);
                    }
                }
            }
            if constexpr (REC) {
                if (a.completes) {  // acc = mu' (before any injected cotangent), tile = mu, xf = the factor's input
                    double sgd = 0.0, zr = 0.0;
                    const double cdet = GD ? cf[2] : 0.0;
                    const double dlane = exact_fin ? 0.0 : lane_diag();
#pragma unroll
                    for (int r = 0; r < RH; ++r) {
                        const int rf = h * RH + r;
                        const double2 mu = tile[unsigned(rf) * NT + tid];
                        const unsigned x = xg_of(rf);
                        const double pr = a.cb_fin_r * mu.x + a.cb_fin_i * mu.y, pi = a.cb_fin_i * mu.x - a.cb_fin_r * mu.y;
                        const double w = pr * xf[r].x - pi * xf[r].y;
                        if (a.wtot) unsafeAtomicAdd(a.wtot + (a.sh_bits ? boff : 0) + x, w);
                        const double cnt = GD ? double(a.dcnt[0] - popc_i((x | rank_hi) & a.dmask[0])) : 0.0;
                        sgd += w * cnt;
                        if (!exact_fin) {  // Re <mu' - (gamma~ + beta~ d) mu, x>
                            double d = diag_of(rf, dlane) + cdet * cnt;
                            for (int g = 1; g < a.gd; ++g) d += cf[2 + g] * double(a.dcnt[g] - popc_i((x | rank_hi) & a.dmask[g]));
                            const double dr = a.fg_r + a.fb_r * d, di = a.fg_i + a.fb_i * d;
                            const double wx = acc[rf].x - (dr * mu.x - di * mu.y), wy = acc[rf].y - (dr * mu.y + di * mu.x);
                            zr += wx * xf[r].x + wy * xf[r].y;
                        }
                    }
                    if (GD) park1<NW>(sgd, red, 2 * a.ga);
                    for (int g = 1; g < a.gd; ++g) {  // further detuning groups next to the one global drive
                        double sg = 0.0;
#pragma unroll
                        for (int r = 0; r < RH; ++r) {
                            const int rf = h * RH + r;
                            const double2 mu = tile[unsigned(rf) * NT + tid];
                            const double pr = a.cb_fin_r * mu.x + a.cb_fin_i * mu.y, pi = a.cb_fin_i * mu.x - a.cb_fin_r * mu.y;
                            sg += (pr * xf[r].x - pi * xf[r].y) * double(a.dcnt[g] - popc_i((xg_of(rf) | rank_hi) & a.dmask[g]));
                        }
                        park1<NW>(sg, red, 2 * a.ga + g);
                    }
                    if (!exact_fin) park1<NW>(zr / cf[0], red, 0);
                }
            }
        }
        if (BWD && a.has_p && (a.inj_gexp || a.inj_gstate)) {  // the completed cotangent sits at a save point: fused injection
            if (inj_any) {
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    double wsum = 0.0;
                    for (int o = 0; o < a.inj_n_obs; ++o)
                        wsum += a.inj_gexp[o * a.inj_ostride + bt] * a.inj_obs[size_t(o) * a.obs_ostride + bt * a.obs_bstride + xg_of(h * RH + r)];
                    acc[h * RH + r].x += 2.0 * wsum * xf[r].x;
                    acc[h * RH + r].y += 2.0 * wsum * xf[r].y;
                }
            }
            if (a.inj_gstate) {
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    const double2 gs = stream_load(a.inj_gstate + boff + xg_of(h * RH + r));
                    acc[h * RH + r].x += gs.x;
                    acc[h * RH + r].y += gs.y;
                }
            }
        }
        if (a.write_v) {
#pragma unroll
            for (int r = 0; r < RH; ++r) stream_store(a.v_out + boff + xg_of(h * RH + r), acc[h * RH + r]);
        }
    };
    static_for<0, NP>(finish_half);

    if (!BWD && a.obs) {  // <v|O|v> for diagonal observables, straight from the registers that hold v
        for (int o = 0; o < a.n_obs; ++o) {
            double e = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) e += a.obs[size_t(o) * a.obs_ostride + bt * a.obs_bstride + xg_of(r)] * (acc[r].x * acc[r].x + acc[r].y * acc[r].y);
            wg_atomic_add<NT>(e, a.expect_slot + o * a.exp_ostride + bt, red);
        }
    }
    if (!a.has_q) {
        if (BWD) flush_gradients();
        return;
    }

    __syncthreads();  // all partner reads of u are done
#pragma unroll
    for (int r = 0; r < R; ++r) tile[unsigned(r) * NT + tid] = acc[r];
    __syncthreads();

    auto start_half = [&](auto hc) {
        constexpr int h = decltype(hc)::value;
        const double* __restrict__ cf = a.coef_sta + bt * a.coef_bstride;
        const bool need_xs = BWD && (!REC || exact_sta);
        double2 xs[RH];
        if (need_xs) {
#pragma unroll
            for (int r = 0; r < RH; ++r) xs[r] = stream_load(a.x_sta + boff + xg_of(h * RH + r));
        }
        double2 q[RH];
        const double dlane = lane_diag();
#pragma unroll
        for (int r = 0; r < RH; ++r) {
            const int rf = h * RH + r;
            double d = diag_of(rf, dlane);
            const unsigned x = xg_of(rf) | rank_hi;
            if (FAST) {
                if (GD) d += cf[2] * double(a.dcnt[0] - popc_i(x & a.dmask[0]));
                for (int g = 1; g < a.gd; ++g) d += cf[2 + g] * double(a.dcnt[g] - popc_i(x & a.dmask[g]));
            } else {
                for (int g = 0; g < a.gd; ++g) d += cf[2 * a.ga + g] * double(a.dcnt[g] - popc_i(x & a.dmask[g]));
            }
            const double dr = a.sg_r + a.sb_r * d, di = a.sg_i + a.sb_i * d;
            q[r].x = dr * acc[rf].x - di * acc[rf].y;
            q[r].y = dr * acc[rf].y + di * acc[rf].x;
        }
        if (BWD && !REC) {  // Re(beta conj(mu) x): weight of d(x) in the gradient (REC: taken by the launch that completes the factor)
            double rr[RH];
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                const int rf = h * RH + r;
                const double pr = a.cb_sta_r * acc[rf].x + a.cb_sta_i * acc[rf].y, pi = a.cb_sta_i * acc[rf].x - a.cb_sta_r * acc[rf].y;
                rr[r] = pr * xs[r].x - pi * xs[r].y;
                if (a.wtot) unsafeAtomicAdd(a.wtot + (a.sh_bits ? boff : 0) + xg_of(rf), rr[r]);
            }
            for (int g = 0; g < a.gd; ++g) {
                double sgd = 0.0;
#pragma unroll
                for (int r = 0; r < RH; ++r) sgd += rr[r] * double(a.dcnt[g] - popc_i((xg_of(h * RH + r) | rank_hi) & a.dmask[g]));
                park1<NW>(sgd, red, 2 * a.ga + g);
            }
        }
        for (int g = 0; g < GA; ++g) {
            const uint32_t mask = a.sta_mask[g];
            if (!FAST && !mask) continue;
            double2 ts[RH], ds[RH];
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                ts[r] = make_double2(0.0, 0.0);
                ds[r] = make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int b = 0; b < LT; ++b) {
                if (FAST || (mask >> b & 1u)) {
#pragma unroll
                    for (int r = 0; r < RH; ++r) {
                        const int rf = h * RH + r;
                        const unsigned i = unsigned(rf) * NT + tid;
                        double2 pv;
                        if (b < LGT) pv = tile[i ^ (1u << b)];
                        else pv = acc[rf ^ (1 << (b < LGT ? 0 : b - LGT))];
                        ts[r].x += pv.x;
                        ts[r].y += pv.y;
                        if (CPLX) {
                            const double sgn = (i >> b & 1u) ? 1.0 : -1.0;
                            ds[r].x = fma(sgn, pv.x, ds[r].x);
                            ds[r].y = fma(sgn, pv.y, ds[r].y);
                        }
                    }
                }
            }
            const double cr = cf[g], ci = cf[a.ga + g];
            const double k1r = a.sb_r * cr, k1i = a.sb_i * cr;
            const double k2r = -a.sb_i * ci, k2i = a.sb_r * ci;
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                q[r].x += k1r * ts[r].x - k1i * ts[r].y;
                q[r].y += k1r * ts[r].y + k1i * ts[r].x;
                if (CPLX) {
                    q[r].x += k2r * ds[r].x - k2i * ds[r].y;
                    q[r].y += k2r * ds[r].y + k2i * ds[r].x;
                }
            }
            if (BWD && exact_sta) {
                double z1r = 0.0, z1i = 0.0, z2r = 0.0, z2i = 0.0;
#pragma unroll
                for (int r = 0; r < RH; ++r) {
                    z1r += ts[r].x * xs[r].x + ts[r].y * xs[r].y;
                    z1i += ts[r].x * xs[r].y - ts[r].y * xs[r].x;
                    if (CPLX) {
                        z2r += ds[r].x * xs[r].x + ds[r].y * xs[r].y;
                        z2i += ds[r].x * xs[r].y - ds[r].y * xs[r].x;
                    }
                }
                if (CPLX) park2<NW>(a.cb_sta_r * z1r - a.cb_sta_i * z1i, a.cb_sta_r * z2i + a.cb_sta_i * z2r, red, 2 * a.ga + a.gd + 2 * g,
                                    2 * a.ga + a.gd + 2 * g + 1);
                else park1<NW>(a.cb_sta_r * z1r - a.cb_sta_i * z1i, red, 2 * a.ga + a.gd + 2 * g);
            }
        }
#pragma unroll
        for (int r = 0; r < RH; ++r) stream_store(a.q_out + boff + xg_of(h * RH + r), q[r]);
    };
    static_for<0, NP>(start_half);
    if (BWD) flush_gradients();
}
