/*
 * rydiff.h — C ABI of the MI355X-native differentiable Rydberg time-evolution library.
 *
 * This is the drop-in boundary for ONE hot path of pasqal-io/pulser-diff: the solver seam
 *
 *     pulser_diff/backend.py:488-494      result = sesolve(H=H_t, psi0, tsave, solver, options)
 *
 * together with the Hamiltonian closure it calls on every sub-step
 *
 *     pulser_diff/hamiltonian.py:499-548  build_ham_tensor(qobj_list) -> H_t(t)
 *
 * and the observable / autograd legs that hang off it
 *
 *     pulser_diff/utils.py:68-86          expect(obs, states)        (ket branch :79-81)
 *     pulser_diff/derivative.py:40,76     torch.autograd.grad(f, x, v, retain_graph=True)
 *
 * The reference hands the solver an opaque Python callable H_t that re-assembles a sparse
 * 2^N x 2^N matrix per call.  This library replaces seam + closure by a STRUCTURED problem:
 * the coefficient arrays the reference captures in `build_ham_tensor` (amp_values /
 * det_values, hamiltonian.py:507-520), the qubits each operator acts on, the pair
 * interaction strengths (hamiltonian.py:343, :536), dt and n_samples (hamiltonian.py:523-524),
 * tsave and psi0 (backend.py:490-491).  H is never materialised.
 *
 * Conventions (fixed by the reference):
 *   - basis r=0, g=1 per qubit; qubit 0 is the MOST significant bit of the amplitude index
 *     (hamiltonian.py:299, utils.py:127-129);  n_j(x) = 1 - bit_j(x).
 *   - (H psi)[x] = [sum_{i<j} U_ij n_i n_j + sum_terms 2*det_k(t) * sum_{j in mask_k} n_j] psi[x]
 *                  + sum_terms sum_{j in mask_k} ( amp_k(t) if bit_j(x)=1 else conj(amp_k(t)) ) psi[x ^ m_j]
 *     where amp_k = 0.5*Omega*exp(-i*phi) and det_k = -0.5*delta are the reference's own
 *     coefficient arrays (hamiltonian.py:420-423, 439-442) and the factor 2 is its
 *     `ham_mat + ham_mat.adjoint()` on a diagonal operator (hamiltonian.py:539-540).
 *   - coefficient at time t: linear interpolation between samples i1 = max(min(floor(t/dt), n-2), 0)
 *     and i2 = min(i1+1, n-2) (hamiltonian.py:532-542).
 *   - states are complex128, laid out (n_tsave, batch, 2^N) with the amplitude index fastest;
 *     the reference's (n_t, dim, B) (simresults.py:398-401) is the permuted view.
 *
 * Ownership: every pointer marked DEVICE is device memory owned by the caller (torch);
 * pointers marked HOST are host memory.  The library allocates nothing on the device: the
 * caller passes a workspace sized by rydiff_plan().  All work is enqueued on `stream`.
 * Functions return 0 on success, a negative RYDIFF_E* code otherwise; rydiff_last_error()
 * gives the message (thread-local).  The library never aborts.
 *
 * Threads and streams: the library keeps NO mutable process state besides the thread-local error string (and a
 * mutex-protected cache of polynomial designs, which only saves host time).  Every choice that steers a call — including
 * the kernel family (RydProblem.kernel_variant) — travels in the RydProblem, so concurrent calls from different threads on
 * different streams with distinct workspaces are independent.  rydiff_forward / rydiff_backward given a RydPlanInfo are
 * fully ASYNCHRONOUS: they only enqueue work (host-side metadata reaches the device as kernel arguments, never through
 * pageable-memory copies) and never synchronise the stream.  rydiff_plan is the one call that waits: the spectral bounds it
 * computes on the device decide the polynomial degree, i.e. how many launches the host has to enqueue.
 */
#ifndef RYDIFF_H
#define RYDIFF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RYDIFF_MAX_QUBITS 30
#define RYDIFF_MAX_PAIR_TERMS 28
#define RYDIFF_MAX_TERMS 64

enum { RYDIFF_OK = 0, RYDIFF_EINVAL = -1, RYDIFF_EWORKSPACE = -2, RYDIFF_EHIP = -3, RYDIFF_ENOTIMPL = -4 };

/* SolverType member names mirror pyqtorch.utils.SolverType as used at backend.py:434,487. */
enum { RYDIFF_SOLVER_KRYLOV_SE = 0, RYDIFF_SOLVER_DP5_SE = 1 };

typedef struct RydProblem {
    int32_t n_qubits;      /* N, 1..RYDIFF_MAX_QUBITS */
    int32_t batch;         /* B: number of state columns / trajectories (backend.py:266-280: psi0 is (dim, B)) */
    int32_t coeff_batch;   /* 1: all trajectories share the coefficient tables; B: one table set per trajectory */
    int32_t n_samples;     /* n: length of every coefficient array (hamiltonian.py:524) */
    double dt;             /* 0.001 / sampling_rate, in us (hamiltonian.py:523) */

    int32_t n_amp_terms;   /* off-diagonal terms ("amp_matrices", hamiltonian.py:517-520) */
    int32_t n_det_terms;   /* diagonal terms ("det_matrices", hamiltonian.py:513-516) */
    const uint32_t* amp_masks;  /* HOST [n_amp_terms]; bit j set = term acts on qubit j (global = all qubits, backend.py:102-112) */
    const uint32_t* det_masks;  /* HOST [n_det_terms] */
    const void* amp_tables;     /* DEVICE complex128 [coeff_batch][n_amp_terms][n_samples] = 0.5*amp*exp(-i*phase) */
    const double* det_tables;   /* DEVICE float64    [coeff_batch][n_det_terms][n_samples] = -0.5*det */
    const double* u_pairs;      /* DEVICE float64 [N(N-1)/2], U_ij = C6/r_ij^6 in itertools.combinations order (hamiltonian.py:385) */

    int32_t n_tsave;       /* number of evaluation times (backend.py:364-373), >= 2 */
    const double* tsave;   /* HOST float64 [n_tsave], strictly increasing, in us */

    int32_t solver;        /* RYDIFF_SOLVER_* */
    double tol;            /* per-exponential truncation target (<=0: default 1e-13) */

    int32_t n_obs;         /* diagonal observables evaluated at every tsave (utils.py:79-81 for diagonal O) */
    const double* obs_diag;/* DEVICE float64 [n_obs][2^N] */

    /* Optional dense two-qubit terms of the generator M in d/dt v = -i M(t) v — how the master-equation path
     * (backend.py:495-509, SolverType.DP5_ME) runs on this library: rho is the state of a DOUBLED register (row qubits,
     * column qubits), the commutator is an ordinary structured "Hamiltonian" on it, and every collapse operator acting
     * on qubit j contributes a constant 4x4 block on the pair (row qubit j, column qubit j):
     *   (M v)[x] += sum_{s=0..3} T_p[4*own + s] * v[x with the bits of qubits (a_p, b_p) set to s],
     * own / s = 2*bit(a_p) + bit(b_p).  M need not be Hermitian.  Problems with pair terms run on the persistent (N <= 12)
     * or direct kernels. */
    int32_t n_pair_terms;          /* 0..RYDIFF_MAX_PAIR_TERMS */
    const uint32_t* pair_qubits;   /* HOST [n_pair_terms][2]: (a_p, b_p), a_p != b_p */
    const double* pair_tables;     /* HOST complex128 as (re, im) [n_pair_terms][16] */

    /* rydiff_backward only.  != 0: the caller uses only the REAL part of g_amp — its amplitude tables were built from
     * real-valued data (a drive without phase; the reference's 0.5*amp*exp(-1j*phase), hamiltonian.py:420, with phase 0).
     * When in addition every table entry is real (RydPlanInfo.flags bit 0 clear) the chained adjoint passes skip the signed
     * partner sums and the contractions for dL/dIm(amp); the imaginary parts of g_amp are then not computed (zero). */
    int32_t real_amp_grad;

    /* Kernel family, 0 = automatic (what production callers pass).  Non-zero values force one implementation so that
     * parity tests can compare the families with each other and with the oracle, and tuning scripts can time them:
     *   1 direct (one amplitude per thread, global partner loads)       9 direct, always the generic kernels
     *   2 / 3 / 4 LDS-tiled chained passes with 512 / 256 / 1024 threads per tile (13 <= N <= 28; two tile layouts up to
     *             N = 22, three from N = 23)
     *   7 automatic, but three tile layouts wherever they are legal (21 <= N <= 28)
     *   8 automatic, but the LDS-tile persistent kernels also up to 6 qubits (instead of the one-wave lane kernels)
     *  10 chained passes with trajectory-per-XCD placement forced (L2-resident trajectories)
     *  11 automatic, but TWO tile layouts up to 24 qubits (32- / 16-byte runs in the second layout at 23 / 24 qubits)
     *  12 automatic, but tiles in plain workgroup order (no line-sharing swizzle where a layout's runs are shorter than 128 bytes)
     *  13 automatic, but 2^12-amplitude tiles everywhere (automatic takes 2^13-amplitude "wide" tiles — k_chain_wide, DESIGN.md
     *     section 3 — with two layouts at 21..24 qubits and with three at 29 and 30, where variant 13 falls back to the direct kernels)
     *  14 chained passes with wide tiles wherever they are legal (14 <= N <= 30; three layouts from 25)
     *  15 / 16 chained passes with tiles of 2^11 / 2^10 amplitudes where two layouts of them are legal (12 <= N <= 20 / 11 <= N <= 18;
     *     automatic takes the 2^11 tiles around 2^19 amplitudes in flight: 256 tiles, one per CU)
     * "automatic" takes the one-launch sweeps up to 12 qubits, the direct kernels while few tiles are in flight
     * (B * 2^N <= 2^18, with gradients 2^19) and the chained passes beyond.  Results do not depend on the variant beyond rounding. */
    int32_t kernel_variant;

    /* STATE-SHARDED runs (BASELINE config 5; the reference has no counterpart: it keeps the whole
     * state in one process).  shard_bits = g > 0: the top g qubits (qubits 0..g-1 = the top g bits of the amplitude index)
     * select the RANK; every rank owns a contiguous slab of 2^(N-g) amplitudes.  n_qubits, the masks and u_pairs describe the
     * WHOLE register; `batch` = number of ranks whose slabs are part of THIS call, rank ids shard_rank_first ... + batch - 1,
     * and psi0 / states_out / obs_diag are laid out per slab: [batch][2^(N-g)] (obs_diag: [n_obs][batch][2^(N-g)]).
     * expect_out [n_obs][n_tsave][batch] receives every slab's PARTIAL sum (the caller adds them / all-reduces them).
     * One factor pass = the local pass on the N-g slab qubits (same kernels, diagonal evaluated at the global index) plus
     * beta * (c or conj c) * the partner rank's slab for each of the g rank qubits (partner of rank bit k: rank ^ (1 << k)):
     *   - batch == 2^g (every rank in this call, e.g. all of them on one device): partners are read in place;
     *   - otherwise shard_recv[k] (HOST array of g DEVICE buffers of 2^(N-g) amplitudes) must hold the partner's current slab
     *     whenever a pass needs it, and shard_exchange says when: it is called (on the calling thread) with phase 0 right
     *     after the launch that produced the slab `src` (nbytes) which the partners need next — post the sends of `src` and
     *     the receives into shard_recv[] there, ordered after the work enqueued on `stream` so far — and with phase 1 before
     *     the first launch that reads shard_recv[] — make `stream` wait for those receives there.  The library runs the whole
     *     trajectory (every step, every factor) in ONE call and never touches the transport itself (torch.distributed /
     *     RCCL stay with the caller).  A non-zero return aborts the run with RYDIFF_EHIP.
     * GRADIENTS (round 3): rydiff_forward with need_tape (the slabs' trajectory in the workspace tape) followed by rydiff_backward runs
     * the whole reverse sweep natively as well.  The cotangent slabs take the SAME exchange (shard_exchange is called for them exactly as
     * for the state slabs, phase 0 / phase 1), the drive gradients of the rank qubits are contracted with the partner slabs inside the
     * completing launch, grad_expect is [n_obs][n_tsave][batch] (the cotangent of every slab's partial sum: normally the same number for
     * all slabs) and g_amp / g_det / g_u receive this call's PARTIAL sums (the caller adds them over the ranks: one all-reduce of the
     * tiny arrays); g_psi0 is [batch][2^(N-g)].  g_tsave is not available (RYDIFF_ENOTIMPL).
     * No pair terms, 1 <= N-g, N <= RYDIFF_MAX_QUBITS. */
    int32_t shard_bits;
    int32_t shard_rank_first;
    void* const* shard_recv;
    int (*shard_exchange)(void* user, int phase, const void* src, size_t nbytes);
    void* shard_user;

    /* rydiff_forward: != 0: states_out is [1][B][2^N] and receives only the state at the LAST evaluation time (a 24-qubit
     * trajectory would otherwise need n_tsave x 256 MiB).  Launch-per-factor kernels only (more than 12 qubits, or a
     * state-sharded run); not together with need_tape. */
    int32_t final_state_only;

    /* Three-level registers (the reference's basis "all": r, g, h, hamiltonian.py:306-310) run as TWO qubits per atom — atom i =
     * qubits (2i, 2i+1) = (a_i, b_i) with r = (0,1), g = (1,1), h = (1,0); the code (0,0) is never populated — with two
     * generalisations of the terms above, both bit masks over TERM indices (bit k = term k):
     *   amp_conditioned_terms: the flip of qubit j by term k acts only on amplitudes whose SIBLING qubit (j ^ 1) is in state 1:
     *       g <-> r flips a_i where b_i = 1, g <-> h flips b_i where a_i = 1, and neither leaves the three valid codes;
     *   det_ones_terms: term k weights the qubits of its mask that are in state 1 with MINUS its coefficient, i.e. contributes
     *       2*det_k(t) * (0 - #ones) instead of 2*det_k(t) * (#zeros): on the valid codes n_g = a_i + b_i - 1, so the reference's
     *       detuning on sigma_gg (hamiltonian.py:413) is an ordinary term on the a qubits plus a ones-counting term on the b qubits.
     * All terms that address one qubit must agree on these flags; n_qubits must be even when any term is conditioned.  Such
     * problems run on the one-launch kernels up to 12 qubits, beyond that on the generic one-amplitude-per-thread kernels and — with
     * enough tiles in flight, up to 20 qubits — on the chained passes over 2^12-amplitude tiles (forward, adjoint, every gradient). */
    uint64_t amp_conditioned_terms;
    uint64_t det_ones_terms;

    /* RYDIFF_SOLVER_DP5_SE only, optional (NULL: none).  HOST uint8 [n_samples - 1]: entry i multiplies the number of Magnus
     * sub-steps taken on the linear piece between samples i and i + 1 (0 and 1: unchanged).  The sub-step is sized from the
     * generator's width; the error constant also carries dH/dt, so a piece across which a table JUMPS (the edge of a constant
     * pulse inside one sample interval) wants a finer step than the smooth pieces — a caller that builds the tables on the host
     * knows where that is (pulser-diff_amd/hamiltonian.py: piece_refinement) and the library does not have to read them back. */
    const uint8_t* dp5_piece_refine;

    /* need_tape = 3 (PARTIAL tape) only: the number of TRAILING tsave intervals whose factor outputs are all kept in the workspace
     * tape (1 .. n_tsave - 1); the earlier intervals keep their save-point states only and are recomputed by the adjoint sweep.  The
     * caller sizes it to the HBM that is free (rydiff_plan reports the workspace for the value given): what the full tape of
     * need_tape = 2 does for a run that fits, this does for the part of a run that fits. */
    int32_t tape_steps;
} RydProblem;

/* Result of rydiff_plan(): everything that depends on the VALUES in the coefficient tables. */
typedef struct RydPlanInfo {
    double spectral_lo, spectral_hi; /* rigorous bounds on the spectrum of H(t) over the whole run (Gershgorin) */
    double rho_design;               /* max over exponentials of tau * (hi-lo)/2 after sub-stepping */
    int32_t degree;                  /* polynomial degree = matrix-free H applications per (sub-)exponential */
    int32_t n_stages;                /* exponentials in the run (KRYLOV_SE: one per tsave interval) */
    int32_t max_step_factors;        /* largest number of factor passes inside one tsave interval */
    int32_t flags;                   /* bit 0: some flip coefficient has a non-zero imaginary part */
    int64_t total_factors;           /* factor passes of one forward run = H applications per trajectory */
    size_t workspace_bytes;          /* device workspace needed by forward/backward for the flags given to rydiff_plan */
    int32_t tape_mode;               /* the tape the workspace was sized for: 0 none, 1 one state per tsave, 2 full, 3 partial (one state per
                                        tsave + every factor output of the last RydProblem.tape_steps intervals); need_tape = 2 / 3 are granted
                                        from 12 / 13 qubits on and without pair terms, otherwise downgraded to 1 */
    int32_t kernel_family;           /* which forward kernels the problem will run on (reporting only): 0 one-wave lane sweep
                                        (<= 6 qubits), 1 one-workgroup persistent sweep (<= 12 qubits), 2 direct launch per factor,
                                        3 chained LDS-tile launch per factor */
    char kernel_fwd[80];             /* reporting only: the instantiation the forward factor passes of THIS problem run on, as a profiler
                                        prints it, e.g. "k_chain<12,10,false,false,true,false>" (tile bits, log2 threads, complex tables,
                                        adjoint, loop-free, L2-resident) — bench.py names its roofline kernel from here */
    char kernel_bwd[80];             /* the same for the adjoint factor passes (empty when need_backward was 0) */
} RydPlanInfo;

#define RYDIFF_PLAN_SCRATCH_BYTES 1024

/* Inspect the coefficient tables (one tiny kernel + ONE stream synchronisation — the only one in the library), bound the spectrum,
 * choose sub-steps and polynomial degree, and size the workspace.
 *   need_tape      1: reserve room for the (n_tsave, B, 2^N) trajectory inside the workspace (caller passes
 *                        states_out == NULL but wants gradients);
 *                  2: FULL tape — the output of every factor pass is kept ((total_factors+1) states), so the adjoint
 *                        sweep recomputes nothing.  Sized for 288 GB of HBM: 156 GiB at N=20, T=1000.  Granted from 12
 *                        qubits on and without pair terms (below, the adjoint sweep is one launch and recomputes on chip);
 *                        otherwise falls back to 1.  RydPlanInfo.tape_mode reports what was granted.
 *                  3: PARTIAL tape — one state per tsave plus the output of every factor pass of the LAST RydProblem.tape_steps
 *                        tsave intervals: the adjoint sweep recomputes the earlier intervals only.  For runs whose full tape does
 *                        not fit (22 qubits x 600 steps, a batch of two 20-qubit trajectories x 1000 steps ...): the caller sizes
 *                        tape_steps to the free HBM.  Granted from 13 qubits on, without pair terms, not for state-sharded runs;
 *                        otherwise falls back to 1.
 *   need_backward  != 0: reserve the backward-sweep buffers too
 *   scratch        DEVICE, >= RYDIFF_PLAN_SCRATCH_BYTES */
int rydiff_plan(const RydProblem* p, int need_tape, int need_backward, void* scratch, void* stream, RydPlanInfo* info);

/* Forward: replaces sesolve(H_t, psi0, tsave, solver, options).states (backend.py:488-494,513-521)
 * and SimulationResults.expect for diagonal observables (simresults.py:81-129).
 *   info        HOST: result of rydiff_plan for the SAME table values (the call is then asynchronous: it enqueues and
 *               returns), or NULL to plan internally (one synchronisation; workspace must then be large enough, see
 *               RYDIFF_EWORKSPACE)
 *   psi0        DEVICE complex128 [B][2^N]
 *   states_out  DEVICE complex128 [n_tsave][B][2^N], or NULL (trajectory kept in the workspace tape if need_tape).
 *               With need_tape = 2 / 3 AND states_out the factor outputs go to the (granted) workspace tape and the states at the
 *               save points are copied out of it — stored states plus a later gradient without (or with less) recomputation.
 *   expect_out  DEVICE float64 [n_obs][n_tsave][B], or NULL */
int rydiff_forward(const RydProblem* p, const RydPlanInfo* info, const void* psi0, void* states_out, double* expect_out,
                   void* workspace, size_t workspace_bytes, int need_tape, void* stream);

/* Backward (vector-Jacobian product): replaces torch autograd's tape through every solver
 * sub-step (derivative.py:40,76).  Cotangents use torch's convention for complex tensors
 * (grad = dL/dRe + i dL/dIm).  The workspace must be the one the forward call used when the
 * trajectory lives in its tape (states == NULL).
 *   states       DEVICE: the states_out of the forward call, or NULL to use the workspace tape (with need_tape = 2 / 3 the
 *                granted workspace tape is used even when states is given)
 *   grad_states  DEVICE complex128 [n_tsave][B][2^N] or NULL
 *   grad_expect  DEVICE float64 [n_obs][n_tsave][B] or NULL
 *   g_amp        DEVICE complex128 [coeff_batch][n_amp_terms][n_samples] or NULL   (overwritten)
 *   g_det        DEVICE float64    [coeff_batch][n_det_terms][n_samples] or NULL   (overwritten)
 *   g_u          DEVICE float64 [N(N-1)/2] or NULL  (dist_grad, backend.py:456-460 / hamiltonian.py:341-344)
 *   g_tsave      DEVICE float64 [n_tsave] or NULL   (time_grad, backend.py:453-455)
 *   g_psi0       DEVICE complex128 [B][2^N] or NULL */
int rydiff_backward(const RydProblem* p, const RydPlanInfo* info, const void* states, const void* grad_states,
                    const double* grad_expect, void* g_amp, double* g_det, double* g_u, double* g_tsave, void* g_psi0,
                    void* workspace, size_t workspace_bytes, int need_tape, void* stream);

/* One matrix-free application y = H(coefficients) x on DEVICE buffers, for get_hamiltonian-style
 * checks (backend.py:401-427) and micro-benchmarks.  c_amp: HOST complex (re,im) per amp term,
 * c_det: HOST value per det term (already interpolated, reference units as above). */
int rydiff_apply_hamiltonian(const RydProblem* p, const double* c_amp_reim, const double* c_det,
                             const void* x, void* y, void* workspace, size_t workspace_bytes, void* stream);

/* One factor pass  y = gamma*x + beta*H x + sum_k rc_k * remote_k  on DEVICE buffers, coefficients given per term on the
 * HOST like rydiff_apply_hamiltonian.  `remote` (HOST array of n_remote DEVICE pointers, each [B][2^N]) carries the vectors
 * of other GPUs in a state-sharded run: the flip terms of the qubits that select the GPU (pulser-diff_amd/sharded.py;
 * SURVEY.md section 8e, BASELINE config 5).  reuse_diag != 0 skips rebuilding the interaction diagonal kept at the start
 * of `workspace` (>= 8 * 2^N bytes) by a previous call with the same u_pairs. */
int rydiff_apply_factor(const RydProblem* p, const double* c_amp_reim, const double* c_det, const double* gamma_reim,
                        const double* beta_reim, const void* x, void* y, int n_remote, const void* const* remote,
                        const double* remote_coef_reim, int reuse_diag, void* workspace, size_t workspace_bytes, void* stream);

/* Host-only helper (no GPU needed): design the product-form polynomial used for exp(-i*rho*x), x in [-1,1].
 * Writes the degree to *degree and roots (re,im interleaved) to roots_reim[2*max_degree]; returns the
 * measured max |p(x) - exp(-i rho x)| on a test grid in *max_err.  Exposed so the host logic is testable on CPU. */
int rydiff_design_polynomial(double rho, double tol, int max_degree, int* degree, double* roots_reim,
                             double* p0_reim, double* max_err);


const char* rydiff_last_error(void);
const char* rydiff_version(void);

/* sizeof(RydProblem) / sizeof(RydPlanInfo) as this library was compiled: lets a foreign-language binding (ctypes, cgo, JNI)
 * assert that its mirror of the two structs has the same layout before it passes one in. */
size_t rydiff_sizeof_problem(void);
size_t rydiff_sizeof_plan_info(void);

#ifdef __cplusplus
}
#endif
#endif /* RYDIFF_H */
