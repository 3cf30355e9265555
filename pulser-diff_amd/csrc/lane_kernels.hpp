// lane_kernels.hpp — registers up to 6 qubits: the whole state is ONE amplitude per lane of ONE wave (included by rydiff.hip
// after persist_kernels.hpp; same arguments, same factor table, same results as k_persist / k_persist_bwd).
//
// The reference's own workloads live here (its tests and notebooks use 2-4 qubits, BASELINE config 0).  At this size a
// factor is a few dozen arithmetic instructions, so the time of the persistent LDS kernels was all latency: LDS round trips
// for the tile (write, barrier, partner reads) and, worse, one serialised LDS read + wait per scalar of the factor record.
// Here nothing in the per-factor loop touches memory:
//   * partners come through DPP lane exchanges (quad permutes / row rotates; ds_bpermute only for bits 4, 5),
//   * lane t of the wave holds the scalars (gamma, beta, coefficient record) of factor f0 + t of the current chunk of 64 factors in its own
//     registers and the loop broadcasts them with v_readlane (scalar registers, no wait counters),
//   * the next chunk's records are loaded while the current chunk runs (factor entries at the chunk's start, their
//     coefficient records half a chunk later, when the stage indices have arrived),
//   * wave reductions (expectation values, gradient records) are DPP butterflies over the 2^N active lanes only.
// (The register records pay off for ONE wave only: with 2..16 waves per workgroup every wave repeats the ~20 broadcasts that
// one staged LDS row serves, and k_persist got slower — N = 12: 2.89 -> 3.27 us per factor — so it keeps its LDS staging.)
#pragma once

constexpr int kLaneMaxQubits = 6;
constexpr int kLaneChunk = 64;

__device__ __forceinline__ double bcast_lane(double x, int lane /* wave-uniform */) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

template <int B>
__device__ __forceinline__ double lane_xor_f64(double x) {
    const int lo = lane_xor_i32<B>(__double2loint(x)), hi = lane_xor_i32<B>(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

// sum over the 2^LT active lanes (butterfly: every active lane ends up with the total)
template <int LT>
__device__ __forceinline__ double lanes_sum(double x) {
    if constexpr (LT > 0) x += lane_xor_f64<0>(x);
    if constexpr (LT > 1) x += lane_xor_f64<1>(x);
    if constexpr (LT > 2) x += lane_xor_f64<2>(x);
    if constexpr (LT > 3) x += lane_xor_f64<3>(x);
    if constexpr (LT > 4) x += lane_xor_f64<4>(x);
    if constexpr (LT > 5) x += lane_xor_f64<5>(x);
    return x;
}

// Dense two-qubit terms (master equation on the doubled register: <= 3 physical qubits here, so <= 3 pair terms).  The 4x4
// block of term t couples x with the three lanes x ^ mb, x ^ ma, x ^ (ma | mb); each lane keeps ITS row of the (constant)
// block in registers, permuted so that entry delta multiplies the amplitude of lane x ^ D(delta).
constexpr int kLanePairMax = 3;

struct LanePairs {
    int n;
    unsigned dl[kLanePairMax];     // relative flips the block has at all (PairArgs.dl; uniform)
    int d[kLanePairMax][3];        // lane offsets D(1) = mb, D(2) = ma, D(3) = ma | mb
    double2 c[kLanePairMax][4];    // row of the block: c[t][delta] = T_t[4 * own + (delta ^ own)]

    __device__ __forceinline__ void load(const PairArgs& pa, int which, unsigned lane) {
        n = pa.n < kLanePairMax ? pa.n : kLanePairMax;
#pragma unroll
        for (int t = 0; t < kLanePairMax; ++t) {
            const uint32_t ma = t < n ? pa.ma[t] : 0u, mb = t < n ? pa.mb[t] : 0u;
            const int own = ((lane & ma) ? 2 : 0) | ((lane & mb) ? 1 : 0);
            d[t][0] = int(mb);
            d[t][1] = int(ma);
            d[t][2] = int(ma | mb);
            dl[t] = t < n ? pa.dl[t] : 0u;
#pragma unroll
            for (int dl = 0; dl < 4; ++dl)
                c[t][dl] = t < n ? pa.tab[(t * 2 + which) * 16 + own * 4 + (dl ^ own)] : make_double2(0.0, 0.0);
        }
    }
    // sum_t sum_s T_t[4 * own + s] * v[x with the pair's bits set to s]
    __device__ __forceinline__ double2 apply(const double2& v, unsigned lane) const {
        double2 acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int t = 0; t < kLanePairMax; ++t) {
            if (t < n) {  // wave-uniform
                acc.x += c[t][0].x * v.x - c[t][0].y * v.y;
                acc.y += c[t][0].x * v.y + c[t][0].y * v.x;
#pragma unroll
                for (int dl = 1; dl < 4; ++dl) {
                    if (!(this->dl[t] >> dl & 1u)) continue;  // uniform: this relative flip does not occur in the block
                    const int src = int(lane) ^ d[t][dl - 1];
                    const double qx = __shfl(v.x, src, 64), qy = __shfl(v.y, src, 64);
                    acc.x += c[t][dl].x * qx - c[t][dl].y * qy;
                    acc.y += c[t][dl].x * qy + c[t][dl].y * qx;
                }
            }
        }
        return acc;
    }
};

// scalars of one factor as the loop needs them: gamma, beta and beta times every coefficient of the factor's exponential
struct LaneRec {
    double gr, gi, br, bi;
    double cr[kPersistGroups], ci[kPersistGroups], cd[kPersistGroups];  // raw coefficients (Re c, Im c, detuning)
    int stage, save, step_first;
};

struct LaneRecLoader {
    const PersistFactor* factors;
    const double* coef_b;
    int NC, ga, gd, n_factors;
    PersistFactor pf;  // in flight / loaded entry of this lane

    __device__ __forceinline__ void issue_factor(int f0) {
        int idx = f0 + int(threadIdx.x);
        idx = idx < n_factors ? idx : n_factors - 1;
        pf = factors[idx];
    }
    __device__ __forceinline__ void issue_coef(LaneRec& r) const {
        const double* __restrict__ src = coef_b + size_t(pf.stage) * NC;
#pragma unroll
        for (int g = 0; g < kPersistGroups; ++g) {
            r.cr[g] = g < ga ? src[g] : 0.0;
            r.ci[g] = g < ga ? src[ga + g] : 0.0;
            r.cd[g] = g < gd ? src[2 * ga + g] : 0.0;
        }
        r.gr = pf.gr;
        r.gi = pf.gi;
        r.br = pf.br;
        r.bi = pf.bi;
        r.stage = pf.stage;
        r.save = pf.save_index;
        r.step_first = pf.step_first;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// forward: every factor of every time step, one wave per trajectory
// ---------------------------------------------------------------------------------------------------------------------
// FAST: one amplitude group that drives every qubit (a global channel) and at most one detuning group — the usual
// sequence: no group loops, no mask tests, no predicated coefficient broadcasts in the factor loop.
// GLMAX: group slots the generic (non-FAST) instantiation loops over: 2 when there are at most two amplitude and two detuning
// groups (a doubled register of the master equation, one local channel next to the global one), else kPersistGroups.
template <int LT, bool CPLX, bool FAST, int GLMAX = kPersistGroups>
__global__ __launch_bounds__(64) void k_lanes_fwd(PersistArgs a) {
    constexpr int GL = FAST ? 1 : GLMAX;
    constexpr int NT = 1 << LT;
    const unsigned lane = threadIdx.x;
    const bool active = lane < NT;
    const int b = blockIdx.x;
    const size_t boff = size_t(b) * a.dim;
    double2 v = make_double2(0.0, 0.0);
    double ud = 0.0, ob0 = 0.0;
    double cnt[kPersistGroups];
#pragma unroll
    for (int g = 0; g < kPersistGroups; ++g) cnt[g] = g < a.gd ? double(a.dcnt[g] - popc_i(lane & a.dmask[g])) : 0.0;
    if (active) {
        v = a.psi0[boff + lane];
        ud = a.udiag[lane];
        if (a.n_obs > 0) ob0 = a.obs[lane];
    }
    LanePairs pairs;
    pairs.load(a.pair, 0, lane);
    LaneRecLoader ld{a.factors, a.coef + size_t(b) * a.coef_bstride, a.NC, a.ga, a.gd, a.n_factors, {}};
    LaneRec cur, nxt;
    ld.issue_factor(0);
    ld.issue_coef(cur);
    nxt = cur;
    for (int f0 = 0; f0 < a.n_factors; f0 += kLaneChunk) {
        const bool has_next = f0 + kLaneChunk < a.n_factors;
        const int count = has_next ? kLaneChunk : a.n_factors - f0;
        if (has_next) ld.issue_factor(f0 + kLaneChunk);
        for (int fs = 0; fs < count; ++fs) {
            if (fs == kLaneChunk / 2 && has_next) ld.issue_coef(nxt);  // the factor entries issued at fs = 0 have arrived
            const double gr = bcast_lane(cur.gr, fs), gi = bcast_lane(cur.gi, fs);
            const double br = bcast_lane(cur.br, fs), bi = bcast_lane(cur.bi, fs);
            // diagonal: gamma + beta * (U(x) + sum_g c_det[g] * cnt_g(x))
            double d = ud;
#pragma unroll
            for (int g = 0; g < GL; ++g)
                if (FAST || g < a.gd) d = fma(bcast_lane(cur.cd[g], fs), cnt[g], d);
            const double dr = fma(br, d, gr), di = fma(bi, d, gi);
            double2 q;
            q.x = dr * v.x - di * v.y;
            q.y = dr * v.y + di * v.x;
#pragma unroll
            for (int g = 0; g < GL; ++g) {
                if (FAST || g < a.ga) {
                    double2 ts, ds;
                    do { if (!FAST && (a.cond >> g & 1u)) partner_sums_lanes<LT, CPLX, false, true>(v, a.amask[g], lane, ts, ds); else partner_sums_lanes<LT, CPLX, FAST>(v, a.amask[g], lane, ts, ds); } while (0);
                    const double cr = bcast_lane(cur.cr[g], fs);
                    const double k1r = br * cr, k1i = bi * cr;
                    q.x += k1r * ts.x - k1i * ts.y;
                    q.y += k1r * ts.y + k1i * ts.x;
                    if (CPLX) {
                        const double ci = bcast_lane(cur.ci[g], fs);
                        const double k2r = -bi * ci, k2i = br * ci;
                        q.x += k2r * ds.x - k2i * ds.y;
                        q.y += k2r * ds.y + k2i * ds.x;
                    }
                }
            }
            if (!FAST && pairs.n) {  // beta * (dense two-qubit terms)
                const double2 pv = pairs.apply(v, lane);
                q.x += br * pv.x - bi * pv.y;
                q.y += br * pv.y + bi * pv.x;
            }
            v = q;
            if (a.tape_all && active) a.tape_all[(size_t(f0 + fs + 1) * a.B + b) * a.dim + lane] = v;  // full tape: entry g + 1 = output of factor g
            const int save = __builtin_amdgcn_readlane(cur.save, fs);
            if (save) {
                if (a.states && active) a.states[(size_t(save) * a.B + b) * a.dim + lane] = v;
                for (int o = 0; o < a.n_obs; ++o) {
                    const double w = o == 0 ? ob0 : (active ? a.obs[size_t(o) * a.dim + lane] : 0.0);
                    const double e = lanes_sum<LT>(w * (v.x * v.x + v.y * v.y));
                    if (lane == 0) a.expect[(size_t(o) * a.n_tsave + save) * a.B + b] = e;  // single writer: deterministic
                }
            }
        }
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// adjoint sweep: the whole reverse pass of one trajectory in one wave (same contract as k_persist_bwd)
//   per tsave interval, last to first: inject the cotangent of the interval's end point, recompute the interval's factor
//   inputs (each lane parks ITS amplitude in LDS and reads the same slot back: no barrier), walk the factors backwards
//   with the gradient contractions fused, reduce every finished exponential's gradient record over the active lanes.
//   Everything the NEXT interval needs from global memory (its start state, the cotangent weights of its end point) is
//   requested while the current interval computes.
// ---------------------------------------------------------------------------------------------------------------------
template <int LT, bool CPLX, bool FAST, int GLMAX = kPersistGroups>
__global__ __launch_bounds__(64) void k_lanes_bwd(PersistBwdArgs a) {
    constexpr int GL = FAST ? 1 : GLMAX;
    constexpr int NT = 1 << LT;
    __shared__ __attribute__((aligned(16))) double2 park[(kLaneChunk - 1) * 64];
    const unsigned lane = threadIdx.x;
    const bool active = lane < NT;
    const int b = blockIdx.x;
    const size_t boff = size_t(b) * a.dim;
    const size_t sv = size_t(a.B) * a.dim;
    const unsigned xl = active ? lane : 0u;  // inactive lanes read element 0 and are zeroed below (their amplitudes stay 0)
    const double live = active ? 1.0 : 0.0;
    double cnt[kPersistGroups];
#pragma unroll
    for (int g = 0; g < kPersistGroups; ++g) cnt[g] = g < a.gd ? double(a.dcnt[g] - popc_i(lane & a.dmask[g])) : 0.0;
    const double ud = active ? a.udiag[lane] : 0.0;
    auto state_at = [&](int k) -> double2 {
        const double2 s = a.tape[size_t(k) * sv + boff + xl];
        return make_double2(live * s.x, live * s.y);
    };
    // cotangent injected at save point k: grad_states[k] + 2 * (sum_o grad_expect[o][k] obs_o(x)) * psi_k(x).  The loads are
    // only REQUESTED here (no branch on a loaded value); weight() turns them into the factor of psi_k when they are consumed.
    const double ob0 = (a.gexp && a.n_obs > 0 && active) ? a.obs[lane] : 0.0;
    struct Inject {
        double2 gs;
        double g0;
        int flag, k;
    };
    auto inject_terms = [&](int k) -> Inject {
        Inject r{make_double2(0.0, 0.0), 0.0, 0, k};
        if (k < 0) return r;
        if (a.gstate) {
            const double2 s = a.gstate[size_t(k) * sv + boff + xl];
            r.gs = make_double2(live * s.x, live * s.y);
        }
        if (a.gexp) {
            r.flag = a.gflags ? a.gflags[k] : 1;
            r.g0 = a.gexp[size_t(k) * a.B + b];
        }
        return r;
    };
    auto weight = [&](const Inject& r) -> double {
        if (!a.gexp || r.flag == 0) return 0.0;
        double wsum = r.g0 * ob0;
        for (int o = 1; o < a.n_obs; ++o) wsum += a.gexp[(size_t(o) * a.n_tsave + r.k) * a.B + b] * a.obs[size_t(o) * a.dim + xl];
        return 2.0 * live * wsum;
    };
    const int n_save = a.factors[a.n_factors - 1].save_index;  // = T
    double2 mu = make_double2(0.0, 0.0);
    double wt = 0.0;
    double2 xend = state_at(n_save), xnext = state_at(n_save - 1);
    Inject inj = inject_terms(n_save);
    double acc_re[kPersistGroups], acc_im[kPersistGroups], acc_det[kPersistGroups], acc_tau = 0.0;
#pragma unroll
    for (int g = 0; g < kPersistGroups; ++g) acc_re[g] = acc_im[g] = acc_det[g] = 0.0;

    LanePairs pairs, pairs_adj;  // the blocks and their conjugate transposes (the generator need not be Hermitian)
    pairs.load(a.pair, 0, lane);
    pairs_adj.load(a.pair, 1, lane);
    LaneRecLoader ld{a.factors, a.coef + size_t(b) * a.coef_bstride, a.NC, a.ga, a.gd, a.n_factors, {}};
    LaneRec cur;
    int w0 = a.n_factors, w1 = -1;  // window [w0, w1] of factor indices held by lanes 0 .. w1 - w0
    int fend = a.n_factors - 1;
    while (fend >= 0) {
        bool have = fend <= w1 && fend >= w0;
        if (have) have = __builtin_amdgcn_readlane(cur.step_first, fend - w0) >= w0;
        if (!have) {  // the interval ending at fend is not (fully) held: take the kLaneChunk factors ending at fend
            w1 = fend;
            w0 = fend - (kLaneChunk - 1) > 0 ? fend - (kLaneChunk - 1) : 0;
            ld.issue_factor(w0);
            ld.issue_coef(cur);
        }
        const int k1 = __builtin_amdgcn_readlane(cur.save, fend - w0);  // this interval ends at tsave[k1]; xend = state there
        const int fbeg = __builtin_amdgcn_readlane(cur.step_first, fend - w0);
        const int M = fend - fbeg + 1;
        // cotangent of the end point (requested one interval ago)
        {
            const double w = weight(inj);
            mu.x += inj.gs.x + w * xend.x;
            mu.y += inj.gs.y + w * xend.y;
        }
        const double2 x0 = xnext;

        // one forward factor on a register-resident vector
        auto apply_forward = [&](const double2& v, int fs) -> double2 {
            const double gr = bcast_lane(cur.gr, fs), gi = bcast_lane(cur.gi, fs);
            const double br = bcast_lane(cur.br, fs), bi = bcast_lane(cur.bi, fs);
            double d = ud;
#pragma unroll
            for (int g = 0; g < GL; ++g)
                if (FAST || g < a.gd) d = fma(bcast_lane(cur.cd[g], fs), cnt[g], d);
            const double dr = fma(br, d, gr), di = fma(bi, d, gi);
            double2 q;
            q.x = dr * v.x - di * v.y;
            q.y = dr * v.y + di * v.x;
#pragma unroll
            for (int g = 0; g < GL; ++g) {
                if (FAST || g < a.ga) {
                    double2 ts, ds;
                    do { if (!FAST && (a.cond >> g & 1u)) partner_sums_lanes<LT, CPLX, false, true>(v, a.amask[g], lane, ts, ds); else partner_sums_lanes<LT, CPLX, FAST>(v, a.amask[g], lane, ts, ds); } while (0);
                    const double cr = bcast_lane(cur.cr[g], fs);
                    const double k1r = br * cr, k1i = bi * cr;
                    q.x += k1r * ts.x - k1i * ts.y;
                    q.y += k1r * ts.y + k1i * ts.x;
                    if (CPLX) {
                        const double ci = bcast_lane(cur.ci[g], fs);
                        const double k2r = -bi * ci, k2i = br * ci;
                        q.x += k2r * ds.x - k2i * ds.y;
                        q.y += k2r * ds.y + k2i * ds.x;
                    }
                }
            }
            if (!FAST && pairs.n) {
                const double2 pv = pairs.apply(v, lane);
                q.x += br * pv.x - bi * pv.y;
                q.y += br * pv.y + bi * pv.x;
            }
            return q;
        };
        // ---- factor inputs x_1 .. x_{M-1}: x_{i+1} goes to park slot i
        {
            double2 v = x0;
            for (int i = 0; i + 1 < M; ++i) {
                v = apply_forward(v, fbeg + i - w0);
                park[i * 64 + lane] = v;
            }
        }
        // requests for the next interval (one earlier): its start state and the cotangent terms of its end point tsave[k1-1].
        // Issued HERE, after the last use of a loaded value in this interval (x0 in the recompute loop): the wait counter is
        // in order and the compiler waits for everything outstanding, so a request issued before that use would be waited for
        // at once; from here it has the whole adjoint loop to land.
        if (k1 >= 2) xnext = state_at(k1 - 2);
        inj = inject_terms(k1 - 1);
        // ---- adjoint of the interval's factors, last to first
        double2 xin = M > 1 ? park[(M - 2) * 64 + lane] : x0;  // input of factor M-1
        for (int i = M - 1; i >= 0; --i) {
            const int fs = fbeg + i - w0;
            const double2 v = xin;
            if (i >= 1) xin = i >= 2 ? park[(i - 2) * 64 + lane] : x0;  // input of factor i-1, requested one factor ahead
            const int stage = __builtin_amdgcn_readlane(cur.stage, fs);
            const bool stage_end = (i == M - 1) || (__builtin_amdgcn_readlane(cur.stage, fs + 1) != stage);
            const bool stage_begin = (i == 0) || (__builtin_amdgcn_readlane(cur.stage, fs - 1) != stage);
            const double gr = bcast_lane(cur.gr, fs), gi = bcast_lane(cur.gi, fs);
            const double br = bcast_lane(cur.br, fs), bi = bcast_lane(cur.bi, fs);
            double d = ud;
#pragma unroll
            for (int g = 0; g < GL; ++g)
                if (FAST || g < a.gd) d = fma(bcast_lane(cur.cd[g], fs), cnt[g], d);
            double2 hm = make_double2(d * mu.x, d * mu.y);  // H mu
            const double pr = br * mu.x + bi * mu.y, pi = bi * mu.x - br * mu.y;
            const double rr = pr * v.x - pi * v.y;  // Re(beta conj(mu) x)
            wt += rr;
#pragma unroll
            for (int g = 0; g < GL; ++g)
                if (FAST || g < a.gd) acc_det[g] = fma(rr, cnt[g], acc_det[g]);
#pragma unroll
            for (int g = 0; g < GL; ++g) {
                if (FAST || g < a.ga) {
                    double2 ts, ds;
                    do { if (!FAST && (a.cond >> g & 1u)) partner_sums_lanes<LT, true, false, true>(mu, a.amask[g], lane, ts, ds); else partner_sums_lanes<LT, true, FAST>(mu, a.amask[g], lane, ts, ds); } while (0);
                    const double cr = bcast_lane(cur.cr[g], fs), ci = bcast_lane(cur.ci[g], fs);
                    // (F_g mu) = cr * ts + i * ci * ds
                    hm.x += cr * ts.x - ci * ds.y;
                    hm.y += cr * ts.y + ci * ds.x;
                    const double z1r = ts.x * v.x + ts.y * v.y, z1i = ts.x * v.y - ts.y * v.x;
                    const double z2r = ds.x * v.x + ds.y * v.y, z2i = ds.x * v.y - ds.y * v.x;
                    acc_re[g] += br * z1r - bi * z1i;
                    acc_im[g] += br * z2i + bi * z2r;
                }
            }
            if (!FAST && pairs.n) {  // the generator is M = K + T with K Hermitian: M^dagger mu = K mu + T^dagger mu
                const double2 pv = pairs_adj.apply(mu, lane);
                hm.x += pv.x;
                hm.y += pv.y;
            }
            if (stage_end && a.want_tau) {  // dL/dtau = Im<mu, M x_out> = Im<M^dagger mu, x_out>
                const double2 o = (i == M - 1) ? xend : park[i * 64 + lane];
                acc_tau += hm.x * o.y - hm.y * o.x;
            }
            // mu <- conj(gamma) mu + conj(beta) H mu
            const double nx = gr * mu.x + gi * mu.y + br * hm.x + bi * hm.y;
            const double ny = gr * mu.y - gi * mu.x + br * hm.y - bi * hm.x;
            mu = make_double2(nx, ny);
            if (stage_begin) {  // the exponential is complete: reduce its gradient record over the active lanes
                double* rec = a.ge + size_t(b) * a.ge_bstride + size_t(stage) * a.ge_sstride + size_t(b % kGradReplicas) * (a.NC + 1);
#pragma unroll
                for (int g = 0; g < GL; ++g) {
                    if (FAST || g < a.ga) {
                        const double s1 = lanes_sum<LT>(acc_re[g]), s2 = lanes_sum<LT>(acc_im[g]);
                        if (lane == 0) {
                            unsafeAtomicAdd(rec + g, s1);
                            unsafeAtomicAdd(rec + a.ga + g, s2);
                        }
                        acc_re[g] = acc_im[g] = 0.0;
                    }
                    if (g < a.gd) {
                        const double s3 = lanes_sum<LT>(acc_det[g]);
                        if (lane == 0) unsafeAtomicAdd(rec + 2 * a.ga + g, s3);
                        acc_det[g] = 0.0;
                    }
                }
                if (a.want_tau) {
                    const double s4 = lanes_sum<LT>(acc_tau);
                    if (lane == 0) unsafeAtomicAdd(rec + a.NC, s4);
                    acc_tau = 0.0;
                }
            }
        }
        xend = x0;  // the next interval (one earlier) ends where this one started
        fend = fbeg - 1;
    }
    // cotangent of the initial point
    {
        const double w = weight(inj);
        mu.x += inj.gs.x + w * xend.x;
        mu.y += inj.gs.y + w * xend.y;
    }
    if (active) {
        a.mu_out[boff + lane] = mu;
        if (a.wtot) unsafeAtomicAdd(a.wtot + lane, wt);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// adjoint sweep on the FULL tape (PersistBwdArgs::tape_full): the forward sweep kept the output of every factor (entry f of
// the tape = input of factor f, entry n_factors = final state), so nothing is recomputed and the whole reverse pass is ONE
// descending walk over the factors: the inputs stream in from global memory three factors ahead (they are read exactly once,
// in reverse order), the state at an interval's end point is simply the input of the factor processed just before.
// Same contract and arithmetic as k_lanes_bwd.
// ---------------------------------------------------------------------------------------------------------------------
template <int LT, bool CPLX, bool FAST, int GLMAX = kPersistGroups>
__global__ __launch_bounds__(64) void k_lanes_bwd_tape(PersistBwdArgs a) {
    constexpr int GL = FAST ? 1 : GLMAX;
    constexpr int NT = 1 << LT;
    const unsigned lane = threadIdx.x;
    const bool active = lane < NT;
    const int b = blockIdx.x;
    const size_t boff = size_t(b) * a.dim;
    const size_t sv = size_t(a.B) * a.dim;
    const unsigned xl = active ? lane : 0u;
    const double live = active ? 1.0 : 0.0;
    double cnt[kPersistGroups];
#pragma unroll
    for (int g = 0; g < kPersistGroups; ++g) cnt[g] = g < a.gd ? double(a.dcnt[g] - popc_i(lane & a.dmask[g])) : 0.0;
    const double ud = active ? a.udiag[lane] : 0.0;
    auto entry = [&](int f) -> double2 {
        const double2 s = a.tape[size_t(f < 0 ? 0 : f) * sv + boff + xl];
        return make_double2(live * s.x, live * s.y);
    };
    const double ob0 = (a.gexp && a.n_obs > 0 && active) ? a.obs[lane] : 0.0;
    struct Inject {
        double2 gs;
        double g0;
        int flag, k;
    };
    auto inject_terms = [&](int k) -> Inject {
        Inject r{make_double2(0.0, 0.0), 0.0, 0, k};
        if (k < 0) return r;
        if (a.gstate) {
            const double2 s = a.gstate[size_t(k) * sv + boff + xl];
            r.gs = make_double2(live * s.x, live * s.y);
        }
        if (a.gexp) {
            r.flag = a.gflags ? a.gflags[k] : 1;
            r.g0 = a.gexp[size_t(k) * a.B + b];
        }
        return r;
    };
    auto weight = [&](const Inject& r) -> double {
        if (!a.gexp || r.flag == 0) return 0.0;
        double wsum = r.g0 * ob0;
        for (int o = 1; o < a.n_obs; ++o) wsum += a.gexp[(size_t(o) * a.n_tsave + r.k) * a.B + b] * a.obs[size_t(o) * a.dim + xl];
        return 2.0 * live * wsum;
    };
    const int n_save = a.factors[a.n_factors - 1].save_index;  // = T
    double2 mu = make_double2(0.0, 0.0);
    double wt = 0.0;
    // vprev: output of the factor about to be processed (= input of the one processed before); xa, xb, xc: the next inputs
    double2 vprev = entry(a.n_factors), xa = entry(a.n_factors - 1), xb = entry(a.n_factors - 2), xc = entry(a.n_factors - 3);
    Inject inj = inject_terms(n_save);
    double acc_re[kPersistGroups], acc_im[kPersistGroups], acc_det[kPersistGroups], acc_tau = 0.0;
#pragma unroll
    for (int g = 0; g < kPersistGroups; ++g) acc_re[g] = acc_im[g] = acc_det[g] = 0.0;
    LanePairs pairs_adj;  // conjugate transposes of the dense two-qubit blocks
    pairs_adj.load(a.pair, 1, lane);
    LaneRecLoader ld{a.factors, a.coef + size_t(b) * a.coef_bstride, a.NC, a.ga, a.gd, a.n_factors, {}};
    LaneRec cur;
    int w0 = a.n_factors, w1 = -1;  // window [w0, w1] of factor indices held by lanes 0 .. w1 - w0
    for (int f = a.n_factors - 1; f >= 0; --f) {
        if (f < w0) {  // take the kLaneChunk factors ending at f
            w1 = f;
            w0 = f - (kLaneChunk - 1) > 0 ? f - (kLaneChunk - 1) : 0;
            ld.issue_factor(w0);
            ld.issue_coef(cur);
        }
        const int fs = f - w0;
        const int save = __builtin_amdgcn_readlane(cur.save, fs);
        if (save) {  // factor f ends the interval of save point `save`: vprev is the state there; add its cotangent
            const double w = weight(inj);
            mu.x += inj.gs.x + w * vprev.x;
            mu.y += inj.gs.y + w * vprev.y;
            inj = inject_terms(save - 1);  // requested now, consumed at the end of the next interval (or after the loop)
        }
        const double2 v = xa;  // input of factor f
        xa = xb;
        xb = xc;
        xc = entry(f - 3);
        const int stage = __builtin_amdgcn_readlane(cur.stage, fs);
        const bool stage_end = save != 0 || (fs + 1 <= w1 - w0 ? __builtin_amdgcn_readlane(cur.stage, fs + 1) != stage
                                                               : a.factors[f + 1].stage != stage);
        const bool stage_begin = f == 0 || (fs >= 1 ? __builtin_amdgcn_readlane(cur.stage, fs - 1) != stage : a.factors[f - 1].stage != stage);
        const double gr = bcast_lane(cur.gr, fs), gi = bcast_lane(cur.gi, fs);
        const double br = bcast_lane(cur.br, fs), bi = bcast_lane(cur.bi, fs);
        double d = ud;
#pragma unroll
        for (int g = 0; g < GL; ++g)
            if (FAST || g < a.gd) d = fma(bcast_lane(cur.cd[g], fs), cnt[g], d);
        double2 hm = make_double2(d * mu.x, d * mu.y);  // H mu
        const double pr = br * mu.x + bi * mu.y, pi = bi * mu.x - br * mu.y;
        const double rr = pr * v.x - pi * v.y;  // Re(beta conj(mu) x)
        wt += rr;
#pragma unroll
        for (int g = 0; g < GL; ++g)
            if (FAST || g < a.gd) acc_det[g] = fma(rr, cnt[g], acc_det[g]);
#pragma unroll
        for (int g = 0; g < GL; ++g) {
            if (FAST || g < a.ga) {
                double2 ts, ds;
                do { if (!FAST && (a.cond >> g & 1u)) partner_sums_lanes<LT, true, false, true>(mu, a.amask[g], lane, ts, ds); else partner_sums_lanes<LT, true, FAST>(mu, a.amask[g], lane, ts, ds); } while (0);
                const double cr = bcast_lane(cur.cr[g], fs), ci = bcast_lane(cur.ci[g], fs);
                hm.x += cr * ts.x - ci * ds.y;
                hm.y += cr * ts.y + ci * ds.x;
                const double z1r = ts.x * v.x + ts.y * v.y, z1i = ts.x * v.y - ts.y * v.x;
                const double z2r = ds.x * v.x + ds.y * v.y, z2i = ds.x * v.y - ds.y * v.x;
                acc_re[g] += br * z1r - bi * z1i;
                acc_im[g] += br * z2i + bi * z2r;
            }
        }
        if (!FAST && pairs_adj.n) {
            const double2 pv = pairs_adj.apply(mu, lane);
            hm.x += pv.x;
            hm.y += pv.y;
        }
        if (stage_end && a.want_tau) acc_tau += hm.x * vprev.y - hm.y * vprev.x;  // dL/dtau = Im<M^dagger mu, x_out>, x_out = vprev
        const double nx = gr * mu.x + gi * mu.y + br * hm.x + bi * hm.y;
        const double ny = gr * mu.y - gi * mu.x + br * hm.y - bi * hm.x;
        mu = make_double2(nx, ny);
        if (stage_begin) {
            double* rec = a.ge + size_t(b) * a.ge_bstride + size_t(stage) * a.ge_sstride + size_t(b % kGradReplicas) * (a.NC + 1);
#pragma unroll
            for (int g = 0; g < GL; ++g) {
                if (FAST || g < a.ga) {
                    const double s1 = lanes_sum<LT>(acc_re[g]), s2 = lanes_sum<LT>(acc_im[g]);
                    if (lane == 0) {
                        unsafeAtomicAdd(rec + g, s1);
                        unsafeAtomicAdd(rec + a.ga + g, s2);
                    }
                    acc_re[g] = acc_im[g] = 0.0;
                }
                if (g < a.gd) {
                    const double s3 = lanes_sum<LT>(acc_det[g]);
                    if (lane == 0) unsafeAtomicAdd(rec + 2 * a.ga + g, s3);
                    acc_det[g] = 0.0;
                }
            }
            if (a.want_tau) {
                const double s4 = lanes_sum<LT>(acc_tau);
                if (lane == 0) unsafeAtomicAdd(rec + a.NC, s4);
                acc_tau = 0.0;
            }
        }
        vprev = v;
    }
    {   // cotangent of the initial point (vprev = entry 0 = psi0)
        const double w = weight(inj);
        mu.x += inj.gs.x + w * vprev.x;
        mu.y += inj.gs.y + w * vprev.y;
    }
    if (active) {
        a.mu_out[boff + lane] = mu;
        if (a.wtot) unsafeAtomicAdd(a.wtot + lane, wt);
    }
}
