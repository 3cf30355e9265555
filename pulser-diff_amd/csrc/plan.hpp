// Host-side planning: validates a RydProblem, groups qubits that share coefficients, builds the list
// of exponential "stages" (one per tsave interval for KRYLOV_SE) and carves the device workspace.
//
// Reference semantics restated here:
//   interpolation indices / weights      pulser_diff/hamiltonian.py:532-542
//   KRYLOV_SE right-endpoint H freezing  SURVEY.md section 8 a-4 (pinned by notebook outputs KA-2..4)
//   term -> qubit maps                   pulser_diff/hamiltonian.py:419-452, backend.py:102-112
#pragma once
#include <cmath>
#include <cstdint>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rydiff.h"

namespace rydiff {

constexpr int kMaxGroups = RYDIFF_MAX_QUBITS;  // at most one coefficient group per qubit
constexpr double kRhoCap = 6.0;                // per-exponential spectral radius*tau cap (sub-steps above it)
constexpr size_t kAlign = 256;

struct Stage {
    double tau;        // duration of this exponential
    int step;          // tsave interval it belongs to
    int nsub;          // sub-steps (set once the spectral width is known)
    int idx[4];        // sample indices entering the coefficient combination
    double w[4];       // their weights
    double dwdt[4];    // d w / d (interpolation time)
    int tn[2];         // tsave indices the interpolation time depends on (-1: none) ...
    double tnw[2];     // ... with d(time)/d(tsave[tn[i]]) = tnw[i]
    int t_hi, t_lo;    // tau = tau_scale * (P_hi - P_lo); P = tsave[index] or a fixed sample-grid point (index -1)
    double tau_scale;
};

struct Groups {
    int n = 0;
    uint32_t amp_index_mask[kMaxGroups];  // bits of the AMPLITUDE index (qubit j <-> bit N-1-j)
    uint64_t members[kMaxGroups];         // which terms contribute to this group's coefficient
    int count[kMaxGroups];                // qubits in the group
    uint32_t flagged = 0;                 // groups whose member terms carry the flag (amp: conditioned flips, det: ones-counting)
    int nq[kMaxGroups] = {0};             // detuning groups: qubits in the group (count is 0 for a ones-counting group)
};

struct Plan {
    int N = 0;       // qubits of the whole register
    int NL = 0;      // qubits that index a vector of this call: N, or N - shard_bits for a state-sharded run (slabs)
    int shard_bits = 0, rank_first = 0;
    bool shard_self = false;  // every rank's slab is part of this call (partners are read in place)
    size_t dim = 0;  // 2^NL
    int B = 1, Bc = 1, T = 0, n_samples = 0, Ka = 0, Kd = 0, n_obs = 0, solver = 0;
    double dt = 0.0, tol = 1e-13, ode_tol = 1e-9;
    Groups ga, gd;
    int NC = 0;  // doubles per (trajectory, stage) coefficient record: c_re[Ga], c_im[Ga], dcoef[Gd]
    std::vector<Stage> stages;
    std::vector<int> step_begin;  // stages of step k are [step_begin[k], step_begin[k+1])
    std::vector<double> tsave;
    // dense two-qubit terms (master equation on the doubled register): index-bit masks, tables [p][fwd | adjoint][16] (re, im)
    int n_pair = 0;
    uint32_t pair_ma[RYDIFF_MAX_PAIR_TERMS] = {0}, pair_mb[RYDIFF_MAX_PAIR_TERMS] = {0};
    std::vector<double> pair_tab;
    double pair_radius = 0.0;  // Gershgorin radius of the pair terms (sum over terms of the largest absolute row sum)
    size_t off_pair = 0;

    // workspace offsets (bytes)
    size_t off_meta_idx = 0, off_coef = 0, off_stats = 0, off_udiag = 0, off_buf0 = 0, off_buf1 = 0;
    size_t off_tape = 0, off_chain = 0, off_ge = 0, off_wtot = 0, off_members = 0, off_meta2 = 0, off_pp0 = 0, off_pp1 = 0, off_split = 0, off_ptable = 0, ptable_bytes = 0;
    size_t split_off_doubles[4] = {0, 0, 0, 0};  // where the split-diagonal table set of tile size 2^(10 + i) starts inside off_split
    size_t off_pm_begin = 0, off_pm_first = 0, off_pm_tau = 0, off_pm_nsub = 0;  // inputs of the on-device factor table build
    size_t state_bytes = 0;  // B * dim * 16
    size_t total_fwd = 0;
    int chain_slots = 0;
    int tape_mode = 0;
};

inline size_t align_up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

inline bool build_groups(int N, int n_terms, const uint32_t* masks, Groups& g, std::string& err, uint64_t term_flags = 0) {
    g.n = 0;
    g.flagged = 0;
    if (n_terms > RYDIFF_MAX_TERMS) {
        err = "too many terms (max " + std::to_string(RYDIFF_MAX_TERMS) + ")";
        return false;
    }
    for (int j = 0; j < N; ++j) {
        uint64_t sig = 0;
        for (int k = 0; k < n_terms; ++k)
            if (masks[k] >> j & 1u) sig |= (1ull << k);
        if (!sig) continue;
        int found = -1;
        for (int q = 0; q < g.n; ++q)
            if (g.members[q] == sig) found = q;
        if (found < 0) {
            found = g.n++;
            g.members[found] = sig;
            g.amp_index_mask[found] = 0;
            g.count[found] = 0;
        }
        g.amp_index_mask[found] |= (1u << (N - 1 - j));
        g.count[found] += 1;
        if (sig & term_flags) {
            if ((sig & term_flags) != sig) {
                err = "terms that address qubit " + std::to_string(j) + " disagree on amp_conditioned_terms / det_ones_terms";
                return false;
            }
            g.flagged |= 1u << found;
        }
    }
    for (int k = 0; k < n_terms; ++k) {
        if (masks[k] == 0 || (N < 32 && (masks[k] >> N) != 0)) {
            err = "term mask " + std::to_string(k) + " is empty or addresses a qubit >= n_qubits";
            return false;
        }
    }
    return true;
}

// `width`: half spectral width of the generator when it is already known (<= 0: not yet) — the continuous solver's
// sub-step shrinks with it.
inline bool build_plan(const RydProblem* p, Plan& pl, std::string& err, double width = -1.0) {
    if (!p) {
        err = "null problem";
        return false;
    }
    if (p->n_qubits < 1 || p->n_qubits > RYDIFF_MAX_QUBITS) {
        err = "n_qubits out of range";
        return false;
    }
    if (p->batch < 1 || (p->coeff_batch != 1 && p->coeff_batch != p->batch)) {
        err = "coeff_batch must be 1 or batch";
        return false;
    }
    if (p->batch > 65535) {  // the trajectory index is the y dimension of every launch grid
        err = "batch must be <= 65535 (split the columns / trajectories into several calls)";
        return false;
    }
    if (p->n_tsave < 2 || !p->tsave) {
        err = "need at least two evaluation times";
        return false;
    }
    if (p->n_amp_terms < 0 || p->n_det_terms < 0) {
        err = "negative term count";
        return false;
    }
    if ((p->n_amp_terms + p->n_det_terms) > 0 && (p->n_samples < 2 || !(p->dt > 0.0))) {
        err = "coefficient tables need n_samples >= 2 and dt > 0";
        return false;
    }
    if ((p->n_amp_terms > 0 && (!p->amp_tables || !p->amp_masks)) || (p->n_det_terms > 0 && (!p->det_tables || !p->det_masks))) {
        err = "missing coefficient tables or masks";
        return false;
    }
    if (p->n_qubits > 1 && !p->u_pairs) {
        err = "u_pairs is required for more than one qubit";
        return false;
    }
    if (p->n_obs < 0 || (p->n_obs > 0 && !p->obs_diag)) {
        err = "obs_diag missing";
        return false;
    }
    if (p->solver != RYDIFF_SOLVER_KRYLOV_SE && p->solver != RYDIFF_SOLVER_DP5_SE) {
        err = "unknown solver";
        return false;
    }
    if (p->n_pair_terms < 0 || p->n_pair_terms > RYDIFF_MAX_PAIR_TERMS || (p->n_pair_terms > 0 && (!p->pair_qubits || !p->pair_tables))) {
        err = "bad pair terms";
        return false;
    }
    pl.n_pair = p->n_pair_terms;
    pl.pair_tab.assign(size_t(pl.n_pair) * 64, 0.0);
    pl.pair_radius = 0.0;
    for (int t = 0; t < pl.n_pair; ++t) {
        const uint32_t qa = p->pair_qubits[2 * t], qb = p->pair_qubits[2 * t + 1];
        if (qa >= uint32_t(p->n_qubits) || qb >= uint32_t(p->n_qubits) || qa == qb) {
            err = "pair term addresses qubits outside the register";
            return false;
        }
        pl.pair_ma[t] = 1u << (p->n_qubits - 1 - qa);
        pl.pair_mb[t] = 1u << (p->n_qubits - 1 - qb);
        const double* src = p->pair_tables + size_t(t) * 32;
        double* fwd = pl.pair_tab.data() + size_t(t) * 64;
        double* adj = fwd + 32;
        double worst = 0.0;
        for (int r = 0; r < 4; ++r) {
            double row = 0.0;
            for (int c = 0; c < 4; ++c) {
                const double re = src[2 * (4 * r + c)], im = src[2 * (4 * r + c) + 1];
                fwd[2 * (4 * r + c)] = re;
                fwd[2 * (4 * r + c) + 1] = im;
                adj[2 * (4 * c + r)] = re;  // conjugate transpose
                adj[2 * (4 * c + r) + 1] = -im;
                row += std::sqrt(re * re + im * im);
            }
            worst = std::max(worst, row);
        }
        pl.pair_radius += worst;
    }
    pl.N = p->n_qubits;
    pl.shard_bits = p->shard_bits;
    if (pl.shard_bits < 0 || pl.shard_bits > 6 || pl.shard_bits >= pl.N) {
        err = "shard_bits must be in [0, min(6, n_qubits - 1)]";
        return false;
    }
    pl.NL = pl.N - pl.shard_bits;
    pl.rank_first = p->shard_rank_first;
    if (pl.shard_bits) {
        const int world = 1 << pl.shard_bits;
        if (p->coeff_batch != 1 || p->n_pair_terms != 0) {
            err = "state-sharded runs take shared coefficient tables (coeff_batch = 1) and no pair terms";
            return false;
        }
        if (pl.rank_first < 0 || pl.rank_first + p->batch > world) {
            err = "shard_rank_first + batch exceeds 2^shard_bits";
            return false;
        }
        pl.shard_self = p->batch == world;
        if (!pl.shard_self && (!p->shard_recv || !p->shard_exchange || p->batch != 1)) {
            err = "state-sharded run with ranks elsewhere: one slab per call (batch = 1), shard_recv and shard_exchange required";
            return false;
        }
    }
    pl.dim = size_t(1) << pl.NL;
    pl.B = p->batch;
    pl.Bc = p->coeff_batch;
    pl.T = p->n_tsave - 1;
    pl.n_samples = p->n_samples;
    pl.Ka = p->n_amp_terms;
    pl.Kd = p->n_det_terms;
    pl.n_obs = p->n_obs;
    pl.dt = p->dt;
    pl.solver = p->solver;
    if (p->solver == RYDIFF_SOLVER_KRYLOV_SE) {
        pl.tol = (p->tol > 0.0) ? p->tol : 1e-13;
    } else {
        // continuous-time solver: `tol` is the target accuracy of the solution; the exponentials are kept well below it
        pl.ode_tol = (p->tol > 0.0) ? p->tol : 1e-9;
        pl.tol = std::min(1e-13, std::max(pl.ode_tol * 1e-3, 1e-15));
    }
    pl.tsave.assign(p->tsave, p->tsave + p->n_tsave);
    for (int k = 0; k < pl.T; ++k) {
        if (!(pl.tsave[k + 1] > pl.tsave[k]) || !std::isfinite(pl.tsave[k + 1])) {
            err = "tsave must be finite and strictly increasing";
            return false;
        }
    }
    if (!build_groups(pl.N, pl.Ka, p->amp_masks, pl.ga, err, p->amp_conditioned_terms)) return false;
    if (!build_groups(pl.N, pl.Kd, p->det_masks, pl.gd, err, p->det_ones_terms)) return false;
    if (pl.ga.flagged && (pl.N & 1)) {
        err = "conditioned flips pair the qubits (2i, 2i+1): n_qubits must be even";
        return false;
    }
    if ((pl.ga.flagged || pl.gd.flagged) && (p->shard_bits || p->n_pair_terms)) {
        err = "conditioned flips / ones-counting terms: not implemented together with state sharding or pair terms";
        return false;
    }
    // ones-counting detuning groups contribute dcoef * (0 - #ones): the kernels' (count - popcount) with count = 0
    for (int g = 0; g < pl.gd.n; ++g) {
        pl.gd.nq[g] = pl.gd.count[g];
        if (pl.gd.flagged >> g & 1u) pl.gd.count[g] = 0;
    }
    pl.NC = 2 * pl.ga.n + pl.gd.n;

    pl.stages.clear();
    pl.step_begin.assign(pl.T + 1, 0);
    const int n = pl.n_samples;
    auto node = [&](double t, int tnode, Stage& s, double weight, int slot) {
        // hamiltonian.py:532-533 / :538,542
        int i1 = 0, i2 = 0;
        double frac = 0.0;
        if (pl.Ka + pl.Kd > 0) {
            double q = std::floor(t / pl.dt);
            double lim = double(n - 2);
            double qq = q < lim ? q : lim;
            i1 = int(qq) > 0 ? int(qq) : 0;
            i2 = (i1 + 1 < n - 2) ? i1 + 1 : n - 2;
            if (i2 < 0) i2 = 0;
            frac = (t - i1 * pl.dt) / pl.dt;
        }
        s.idx[slot] = i1;
        s.idx[slot + 1] = i2;
        s.w[slot] = weight * (1.0 - frac);
        s.w[slot + 1] = weight * frac;
        s.dwdt[slot] = -weight / pl.dt;
        s.dwdt[slot + 1] = weight / pl.dt;
        (void)tnode;
    };
    // continuous-time solver: largest sub-step of the 4th-order commutator-free Magnus scheme.  Calibrated on the
    // reference's own workloads (tests/test_gpu_dp5.py): 2.5 ns keeps the global error below ~2e-10; error ~ h^4.
    // The leading error term carries two powers of the generator's size ([A,[A,A']]), so beyond the calibrated range
    // (half width ~1e2 rad/us) the step shrinks like width^-1/2 — this is what keeps strongly dissipative master-equation
    // runs (large pair terms) and strongly interacting registers at the same accuracy.
    constexpr double kMagnusWidthRef = 128.0;
    const double h_max = 2.5e-3 * std::pow(std::min(std::max(pl.ode_tol, 1e-14), 1e-4) / 1e-10, 0.25) *
                         (width > kMagnusWidthRef ? std::sqrt(kMagnusWidthRef / width) : 1.0);
    for (int k = 0; k < pl.T; ++k) {
        pl.step_begin[k] = int(pl.stages.size());
        if (pl.solver == RYDIFF_SOLVER_KRYLOV_SE) {
            Stage s{};
            s.step = k;
            s.tau = pl.tsave[k + 1] - pl.tsave[k];
            s.nsub = 1;
            s.tn[0] = k + 1;
            s.tn[1] = -1;
            s.tnw[0] = 1.0;
            s.tnw[1] = 0.0;
            s.t_hi = k + 1;
            s.t_lo = k;
            s.tau_scale = 1.0;
            node(pl.tsave[k + 1], k + 1, s, 1.0, 0);
            s.idx[2] = s.idx[3] = 0;
            s.w[2] = s.w[3] = 0.0;
            s.dwdt[2] = s.dwdt[3] = 0.0;
            pl.stages.push_back(s);
        } else {
            // DP5_SE semantics = the continuous-time solution.  H(t) is piecewise LINEAR in t between sample points
            // (hamiltonian.py:532-542), so the interval is cut at the sample grid and every linear piece is advanced by
            // CF4 Magnus steps: for linear H the two exponentials are exp(-i h/2 H(t0+5h/6)) exp(-i h/2 H(t0+h/6)).
            const double a = pl.tsave[k], b = pl.tsave[k + 1];
            std::vector<double> pts{a};
            if (pl.Ka + pl.Kd > 0) {
                long i = long(std::floor(a / pl.dt)) + 1;
                while (i <= long(n) - 2 && double(i) * pl.dt < b - 1e-13) {
                    if (double(i) * pl.dt > a + 1e-13) pts.push_back(double(i) * pl.dt);
                    ++i;
                }
            }
            pts.push_back(b);
            const int np = int(pts.size()) - 1;
            for (int q = 0; q < np; ++q) {
                const double p0 = pts[q], p1 = pts[q + 1], hf = p1 - p0;
                const int lo_owner = (q == 0) ? k : -1, hi_owner = (q == np - 1) ? k + 1 : -1;
                int S = std::max(1, int(std::ceil(hf / h_max - 1e-9)));
                if (p->dp5_piece_refine && n >= 2) {  // caller's hint: pieces across which the coefficient tables jump
                    const long si = std::min(std::max(long(std::floor(0.5 * (p0 + p1) / pl.dt)), 0L), long(n) - 2);
                    S *= std::max(1, int(p->dp5_piece_refine[si]));
                }
                for (int sub = 0; sub < S; ++sub)
                    for (double theta : {1.0 / 6.0, 5.0 / 6.0}) {
                        Stage s{};
                        s.step = k;
                        const double mu = (sub + theta) / S;
                        s.tau = hf / (2.0 * S);
                        s.nsub = 1;
                        s.tn[0] = lo_owner;
                        s.tn[1] = hi_owner;
                        s.tnw[0] = 1.0 - mu;
                        s.tnw[1] = mu;
                        s.t_hi = hi_owner;
                        s.t_lo = lo_owner;
                        s.tau_scale = 1.0 / (2.0 * S);
                        node(p0 + mu * hf, -1, s, 1.0, 0);
                        s.idx[2] = s.idx[3] = 0;
                        s.w[2] = s.w[3] = 0.0;
                        s.dwdt[2] = s.dwdt[3] = 0.0;
                        pl.stages.push_back(s);
                    }
            }
        }
    }
    pl.step_begin[pl.T] = int(pl.stages.size());
    pl.state_bytes = size_t(pl.B) * pl.dim * 16;
    return true;
}

// Carve the workspace. `chain_slots` = number of intermediate state buffers the backward recompute needs.
// tape_mode: 0 none | 1 one state per tsave | 2 FULL: the output of every factor pass (no recompute in the adjoint sweep;
// needs total_factors+1 states of HBM — e.g. 156 GiB for N=20, T=1000, which an MI355X's 288 GB holds) | 3 PARTIAL: one state per
// tsave + the intermediate factor outputs of the trailing intervals (`tape_entries` states in all, counted by the caller)
inline size_t carve(Plan& pl, int tape_mode, bool need_backward, int chain_slots, int64_t total_factors = 0, int64_t tape_entries = 0) {
    // factor table of the persistent small-N kernel: 40 bytes per factor pass
    pl.ptable_bytes = (pl.N <= 12 && !pl.shard_bits) ? size_t(total_factors) * 48 + 64 : 0;  // sizeof(PersistFactor)
    const size_t E = pl.stages.size();
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes);
        return o;
    };
    pl.off_stats = take(64 * sizeof(double));
    pl.off_meta_idx = take(E * 24);  // StageDev records (rydiff.hip)
    pl.off_members = take(2 * kMaxGroups * sizeof(uint64_t));
    pl.off_coef = take(size_t(pl.Bc) * E * std::max(pl.NC, 1) * sizeof(double));
    pl.off_udiag = take(pl.dim * sizeof(double) * (pl.shard_bits ? size_t(pl.B) : 1));  // sharded: one table per slab
    pl.off_buf0 = take(pl.state_bytes);
    pl.off_buf1 = take(pl.state_bytes);
    pl.off_pp0 = take(pl.state_bytes);  // partial vectors of the chained passes
    pl.off_pp1 = take(pl.state_bytes);
    // split interaction diagonal for the tile layouts: utt[3][2^LT] + vr[3][tiles][16] per tile size; one set per tile size
    // (LT = 10 .. 13: the forward and the adjoint chains pick their tile size independently)
    {   // (sharded runs index the table by the GLOBAL tile: 2^(N-LT) rows)
        const size_t gdim = size_t(1) << pl.N;
        size_t tot = 0;
        for (int lt = 10; lt <= 13; ++lt) {  // tile sizes 2^10 .. 2^13 amplitudes, up to three tile layouts each
            pl.split_off_doubles[lt - 10] = tot;
            tot += 3 * ((size_t(1) << lt) + ((gdim >> lt) ? (gdim >> lt) : 1) * 16);
        }
        pl.off_split = take(tot * sizeof(double));
    }
    pl.off_ptable = take(pl.ptable_bytes);
    if (pl.ptable_bytes) {
        pl.off_pm_begin = take(size_t(pl.T + 1) * sizeof(int32_t));
        pl.off_pm_first = take(size_t(pl.T + 1) * sizeof(int32_t));
        pl.off_pm_tau = take(E * sizeof(double));
        pl.off_pm_nsub = take(E * sizeof(int32_t));
    }
    pl.off_pair = take(size_t(pl.n_pair) * 64 * sizeof(double));
    pl.total_fwd = off;
    pl.tape_mode = tape_mode;
    if (tape_mode == 2) pl.off_tape = take(size_t(total_factors + 1) * pl.state_bytes);
    else if (tape_mode == 3) pl.off_tape = take(size_t(tape_entries) * pl.state_bytes);
    else pl.off_tape = tape_mode ? take(size_t(pl.T + 1) * pl.state_bytes) : 0;
    pl.chain_slots = chain_slots;
    if (need_backward) {
        pl.off_chain = take(size_t(chain_slots > 0 ? chain_slots : 1) * pl.state_bytes);
        pl.off_ge = take(size_t(pl.Bc) * E * 64 /* kGradReplicas */ * (pl.NC + 1) * sizeof(double));
        pl.off_wtot = take(pl.dim * (pl.shard_bits ? size_t(pl.B) : 1) * sizeof(double));  // sharded: one weight slab per rank of the call
        pl.off_meta2 = take(std::max(E * 40, size_t(pl.T + 1) * sizeof(int32_t)));  // StageBwdDev records, or the save-point flags of the one-launch adjoint
    }
    return off;
}

}  // namespace rydiff
