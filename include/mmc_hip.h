/*
 * mmc_hip.h -- C ABI of libmmc_hip.so: the MI355X (gfx950) implementation of the per-move energy
 * hot path of BradenDKelly/MetropolisMonteCarlo.
 *
 * The reference has no FFI layer: its boundary is the set of Julia generic functions that
 * `Loop()` (Ewald/main.jl:460-696) and `potential()` (Ewald/energy.jl:946-1032) call.  Each entry
 * point below names the reference method it replaces (file:line into /root/reference).  The Julia
 * methods of the same names that `ccall` these symbols are in
 * metropolismontecarlo_amd/julia/MMCHip.jl and INTEGRATION.md; the Python mirror used by the
 * tests is metropolismontecarlo_amd/api.py.
 *
 * Conventions (what a `ccall` passes):
 *   - every array is a Julia Vector of bits types, borrowed for the duration of the call only:
 *       Vector{SVector{3,Float64}}  -> const double*  (3 doubles per element, x y z)
 *       Vector{Float64}             -> const double*
 *       Vector{Int64}               -> const int64_t* (atom ranges / types are 1-BASED, inclusive)
 *       Vector{SVector{3,Int32}}    -> int32_t*       (3 per element)
 *       Vector{ComplexF64}          -> double*        (re, im interleaved)
 *       Matrix{Float64}             -> const double*  (column-major n_types x n_types)
 *   - molecule indices (`i`, `chosenOne`) are 1-BASED as in the reference;
 *   - every function returns an int32 status (MMC_OK == 0); results come back through
 *     out-pointers; the library never calls exit() (the reference does, energy.jl:426-428);
 *   - a reference `@assert` becomes MMC_ERR_ASSERT; mmc_last_error() has the text;
 *   - one context (or batch) is not thread-safe, like the reference's mutable EWALD; different
 *     contexts may be driven from different threads and run on different HIP streams.
 * All arithmetic is fp64.  Energies are in K, lengths in Angstrom, charges in e.
 */
#ifndef MMC_HIP_H
#define MMC_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    MMC_OK = 0,
    MMC_ERR_ARG = 1,      /* null pointer, bad size, index out of range */
    MMC_ERR_ASSERT = 2,   /* a reference @assert would have thrown */
    MMC_ERR_HIP = 3,      /* a HIP runtime call failed (no device, OOM, launch failure) */
    MMC_ERR_STATE = 4,    /* call order: e.g. RecipMove before PrepareEwaldVariables */
    MMC_ERR_UNSUPPORTED = 5
};

typedef struct mmc_ctx mmc_ctx;     /* one system (one Markov chain) resident on one GPU */
typedef struct mmc_batch mmc_batch; /* R independent replicas of one system on one GPU   */

/* Totals written by the total-energy drivers: the fields of `Properties`
 * (Ewald/auxillary.jl:37-45) that potential() fills, plus the four terms it println()s. */
typedef struct {
    double energy, virial, coulomb;
    double lj, real, recip, self;
    int32_t n_overlap; /* molecules for which EwaldReal returned (0.0, true) */
    int32_t _pad;
} mmc_totals;

const char *mmc_last_error(void); /* thread-local text of the last non-OK status */
const char *mmc_version(void);
int32_t mmc_device_count(int32_t *count);

/* ---- context ------------------------------------------------------------------------------- */
/* `hip_stream`: a hipStream_t to launch on (e.g. torch.cuda.current_stream().cuda_stream), or
 * NULL to let the context own a non-blocking stream. */
int32_t mmc_ctx_create(int32_t device, void *hip_stream, mmc_ctx **out);
int32_t mmc_ctx_destroy(mmc_ctx *ctx);
int32_t mmc_ctx_synchronize(mmc_ctx *ctx);

/* Mirror moa/soa/vdwTable on the device.  Replaces nothing in the reference -- it is the price of
 * a device: the fields are exactly those LJ_poly_dU / EwaldReal read (Ewald/energy.jl:216-229,
 * Ewald/ewalds.jl:305-308): moa.COM, moa.firstAtom, moa.lastAtom, soa.coords, soa.atype,
 * soa.charge, vdwTable.eps_ij, vdwTable.sig_ij (Ewald/structs.jl:337-347), box. */
int32_t mmc_upload_system(mmc_ctx *ctx, int64_t n_mol, int64_t n_atoms, const double *com,
                          const int64_t *first_atom, const int64_t *last_atom,
                          const double *coords, const int64_t *atype, const double *charge,
                          int64_t n_types, const double *eps, const double *sig, double box);
/* Loop() writes the moved molecule into moa.COM[i] / soa.coords[first:last] before it calls the
 * energy functions (Ewald/main.jl:527,552) and restores them on rejection (:623-624); this is the
 * device-side counterpart of those two assignments.  `atoms`: 3*(last-first+1) doubles.
 * The context compares with its host mirror of the coordinates: an unchanged molecule costs nothing,
 * a changed one travels with the next evaluation instead of a launch of its own. */
int32_t mmc_set_molecule(mmc_ctx *ctx, int64_t i, const double *com, const double *atoms);
/* Re-send every centre of mass and atom position (same topology): the whole-array form of the
 * assignments above, for callers that changed more than one molecule on the host.  `com` may be
 * NULL to re-send the atoms only (all that RecipLong reads). */
int32_t mmc_update_system(mmc_ctx *ctx, const double *com, const double *coords);
int32_t mmc_download_system(mmc_ctx *ctx, double *com, double *coords);
/* The device part of an NPT volume move.  The reference only specifies it in prose
 * (Ewald/volumeChange.jl:59-80): centres of mass scale by new_box/box, atoms translate rigidly
 * with their molecule; then everything that depends on the box is rebuilt -- kappa (alpha/L,
 * Ewald/main.jl:290-291), kxyz/cfac (PrepareEwaldVariables, Ewald/ewalds.jl:45-103), zeroed
 * sumQExp arrays.  Follow with mmc_potential_ewald for the energy at the new volume
 * (volumeChange.jl:91-111). */
int32_t mmc_volume_change(mmc_ctx *ctx, double new_box, double new_kappa);
/* One NPT volume move without a host round trip (Ewald/volumeChange.jl:59-147, `MC_vol`).
 * mmc_volume_trial: copies aside ON THE DEVICE everything the move rewrites (coordinates in their
 * three layouts, fixed-point centres of mass, S(k), k-vectors, erfc table; one launch), then does
 * mmc_volume_change + mmc_potential_ewald: `tot` is the energy at the new volume (:91-111).
 * mmc_volume_accept (:132-147): nothing left to do.  mmc_volume_reject: the copy back (one
 * launch): coordinates, tables and S(k) are the pre-move ones bit for bit. */
int32_t mmc_volume_trial(mmc_ctx *ctx, double new_box, double new_kappa, double lj_rcut,
                         double qq_rcut, mmc_totals *tot);
int32_t mmc_volume_accept(mmc_ctx *ctx);
int32_t mmc_volume_reject(mmc_ctx *ctx);

/* PrepareEwaldVariables(ewald, boxSize)                       Ewald/ewalds.jl:45-103
 * Builds kxyz/cfac on the device, zeroes sumQExpOld/New.  k_sq_max != 27 -> MMC_ERR_ASSERT (:49).
 * `factor` is EWALD.factor (Ewald/constants.jl:24-28). */
int32_t mmc_prepare_ewald(mmc_ctx *ctx, double kappa, int64_t nk, int64_t k_sq_max, double box,
                          double factor, int64_t *nkvecs);
int32_t mmc_get_kvectors(mmc_ctx *ctx, int32_t *kxyz /* [NKVECS][3] */, double *cfac);
/* EWALD.sumQExpOld / sumQExpNew (Ewald/ewalds.jl:16-17); either pointer may be NULL. */
int32_t mmc_get_sumqexp(mmc_ctx *ctx, double *sum_old, double *sum_new);
int32_t mmc_set_sumqexp(mmc_ctx *ctx, const double *sum_old, const double *sum_new);

/* How the per-molecule calls are served.  Loop() calls LJ_poly_dU(i) and EwaldShort(i) in pairs on
 * an unchanged system (Ewald/main.jl:491+501, :557+566).  The context evaluates BOTH terms in one
 * command whichever is asked for first (the cutoff of the other term is assumed to be the one last
 * used for it, else the same) and answers the second call from that result without touching the
 * device -- valid as long as molecule, cutoffs and coordinates (a version number of the host
 * mirror) are unchanged.  For systems of identical 3-site molecules with an EWALD the evaluations
 * run on a persistent kernel (csrc/mmc_ctxsrv.hpp) that the host talks to through a command block
 * in pinned memory: no launch, no stream synchronisation per call; every wait is bounded, and the
 * kernel is stopped before anything else runs on the context (and after ~1 s without a call).
 * Other systems pay one launch per evaluation.  Results of the two paths agree to ~1e-13
 * relative (summation order; the server takes erfc(kappa r)/r from the table of the batch
 * kernels).
 *
 * LJ_poly_dU(i, moa, soa, vdwTable, r_cut, box)               Ewald/energy.jl:209-290
 * (and the legacy LJ_poly_dU(i, system::Requirements)          Ewald/energy.jl:126-206)
 * -> (4*pot, 24*vir/3). */
int32_t mmc_lj_poly_du(mmc_ctx *ctx, int64_t i, double r_cut, double *pot, double *vir);

/* EwaldReal(chosenOne, moa, soa, ewald, r_cut, box)           Ewald/ewalds.jl:293-376, ovr = 0.5
 * EwaldReal(qq_r, qq_q, kappa, box, thisMol_thisAtom, i, sys) Ewald/ewalds.jl:205-289, ovr = 1.0
 * -> (pot, overlap); pot WITHOUT factor; overlap -> pot = 0.0 (:359-360). */
int32_t mmc_ewald_real(mmc_ctx *ctx, int64_t i, double r_cut, double ovr, double *pot,
                       int32_t *overlap);

/* EwaldShort(i, moa, soa, sim_props, ewald, box)              Ewald/ewalds.jl:892-910
 * -> (e = EwaldReal*factor, e/3, overlap) with r_cut = sim_props.qq_rcut. */
int32_t mmc_ewald_short(mmc_ctx *ctx, int64_t i, double qq_rcut, double *e, double *v,
                        int32_t *overlap);

/* CoulombReal(qq_r, qq_q, box, chosenOne, system)             Ewald/energy.jl:618-711
 * bare Coulomb: COM gate r_cut + (r_cut*0.25+5), ovr = 1, atomic cutoff r2 < r_cut^2;
 * r_cut != 10.0 -> MMC_ERR_ASSERT (:648). */
int32_t mmc_coulomb_real(mmc_ctx *ctx, int64_t i, double r_cut, double *pot, int32_t *overlap);

/* RecipLong(ewald, r, qq_q, box)                              Ewald/ewalds.jl:538-604
 * (legacy RecipLong(system, ewald, r, qq_q)                    Ewald/ewalds.jl:465-534)
 * Full structure factor of the uploaded atoms; writes BOTH sumQExpOld and sumQExpNew (:600-601);
 * energy WITHOUT factor (:603). */
int32_t mmc_recip_long(mmc_ctx *ctx, double *energy);

/* RecipMove(box, ewalds, r_old, r_new, qq_q)                  Ewald/ewalds.jl:718-826
 * n != 3, k_sq_max != 27 or nk != 5 -> MMC_ERR_ASSERT (:740-743).  sumQExpNew += dS in place
 * (:805-814); returns energy*factor (:825). */
int32_t mmc_recip_move(mmc_ctx *ctx, const double *r_old, const double *r_new, const double *q,
                       int64_t n, double *d_energy);

/* ewald.sumQExpOld = copy(ewald.sumQExpNew)                   Ewald/main.jl:621
 * (on the device the two arrays are names of buffers: a commit or rollback renames, nothing is copied) */
int32_t mmc_recip_commit(mmc_ctx *ctx);
/* ewald.sumQExpNew = copy(ewald.sumQExpOld)                   Ewald/main.jl:628 */
int32_t mmc_recip_rollback(mmc_ctx *ctx);

/* ---- the same calls with the CALLER'S OWN ARRAYS: what a Julia method forwards --------------------
 * One ccall per reference call, nothing to keep in step by hand.  `com` / `coords` are moa.COM and
 * soa.coords (legacy: system.rm and qq_r / system.ra) as they are NOW, whole arrays.  Loop() changes
 * one molecule between calls and may have restored the one of the previous call (main.jl:527,552,
 * 623-624): the context looks at molecule i and at the molecule of its previous mmc_call_*, compares
 * them with its mirror, and sends what changed along with the evaluation.  After any other edit of
 * the arrays call mmc_update_system.
 *
 * mmc_call_recip_move takes ewalds.sumQExpOld / sumQExpNew as they are now.  Loop() rebinds them to
 * copies on every move (main.jl:621,628), so the context finds out which of its device buffers
 * each array is by CONTENT, against pinned host copies of those buffers (two 5.4 KB comparisons
 * on the host; arrays it has never seen are uploaded).  When the moved molecule was evaluated by
 * mmc_call_lj_poly_du / mmc_call_ewald_short while the device still held its old coordinates,
 * RecipMove was computed in that same command: the call then only checks r_old / r_new / q and the
 * arrays and copies sumQExpNew out.  sum_new is updated in place like the reference's
 * (ewalds.jl:805-814); d_energy has the factor applied (:825). */
int32_t mmc_call_lj_poly_du(mmc_ctx *ctx, int64_t i, const double *com, const double *coords,
                            double r_cut, double *pot, double *vir);
int32_t mmc_call_ewald_real(mmc_ctx *ctx, int64_t i, const double *com, const double *coords,
                            double r_cut, double ovr, double *pot, int32_t *overlap);
int32_t mmc_call_ewald_short(mmc_ctx *ctx, int64_t i, const double *com, const double *coords,
                             double qq_rcut, double *e, double *v, int32_t *overlap);
int32_t mmc_call_recip_move(mmc_ctx *ctx, const double *r_old, const double *r_new,
                            const double *q, int64_t n, const double *sum_old, double *sum_new,
                            double *d_energy);
/* Counters of the context: out[0..9] = commands answered by the persistent kernel, launches of it,
 * commands that had to be repeated on a fresh one, per-molecule calls answered from the cached
 * evaluation, RecipMoves answered from the speculative sum, speculative sums that went unused,
 * evaluations by ordinary launch, 1 if the persistent kernel is running, evaluations answered by
 * the look-ahead (the command that evaluates a moved molecule i also has molecule i + 1 -- the next
 * of Loop()'s sweep, main.jl:490 -- evaluated by a second set of workgroups; valid if nothing
 * changes before it is asked for, i.e. the move was accepted), look-aheads posted. */
int32_t mmc_ctx_stats(mmc_ctx *ctx, int64_t out[10]);
/* Test hook / measurement: average round trip in microseconds of n empty commands through the
 * running persistent kernel -- the floor under every served call. */
int32_t mmc_ctx_ping(mmc_ctx *ctx, int64_t n, double *us_avg);
/* "server": 1 (default) = use the persistent kernel when it applies, 0 = a launch per evaluation */
int32_t mmc_ctx_set_option(mmc_ctx *ctx, const char *key, int64_t value);

/* EwaldSelf(ewald, qq_q)                                      Ewald/ewalds.jl:829-833 (factor in) */
int32_t mmc_ewald_self(mmc_ctx *ctx, double *self_energy);

/* potential(moa, soa, tot, ewalds, vdwTable, sim_props, "ewald")   Ewald/energy.jl:946-1032
 * lj_rcut = sim_props.LJ_rcut, qq_rcut = sim_props.qq_rcut.  Takes everything from the context
 * (the reference reads globals `ewald`, `totProps`, :994 -- identical at its only call site). */
int32_t mmc_potential_ewald(mmc_ctx *ctx, double lj_rcut, double qq_rcut, mmc_totals *tot);
/* potential(moa, soa, tot, ewald, vdwTable, sim_props)  ("Wolf")  Ewald/energy.jl:864-943 */
int32_t mmc_potential_wolf(mmc_ctx *ctx, double lj_rcut, double qq_rcut, mmc_totals *tot);

/* The five hot-path calls of one Loop() iteration in ONE launch   Ewald/main.jl:491-593
 * (2x LJ_poly_dU, 2x EwaldShort, RecipMove) for molecule i moved to com_new/atoms_new.
 * d[0] = E_new_LJ - E_old_LJ, d[1] = real new - old (factor in), d[2] = deltaRecip (0 when
 * overlap, :580-590), d[3] = virial new - old + deltaRecip/3 (:600-601).  The device state is
 * left OLD; follow with mmc_accept_move (:598-621) or mmc_reject_move (:622-629). */
int32_t mmc_trial_move(mmc_ctx *ctx, int64_t i, const double *com_new, const double *atoms_new,
                       double lj_rcut, double qq_rcut, double d[4], int32_t *overlap);
int32_t mmc_accept_move(mmc_ctx *ctx);
int32_t mmc_reject_move(mmc_ctx *ctx);

/* ---- single-precision tolerance study (BASELINE.json configs[4]: Wolf vs Ewald, fp32 vs fp64) ---
 * Not a reference interface: the reference is fp64 only.  The same terms as mmc_lj_poly_du /
 * mmc_ewald_real / mmc_recip_long / mmc_recip_move, evaluated from coordinates rounded to fp32 with
 * fp32 arithmetic; mixed = 0: fp32 accumulators, 1: fp64 accumulators.
 * total: out = { LJ energy, LJ virial, real-space Coulomb (factor in), reciprocal (factor in),
 *                number of overlapping molecules, 0 } -- the summed terms of potential()
 *        (energy.jl:946-1032); add mmc_ewald_self for the Ewald total or the Wolf constants of
 *        mmc_potential_wolf (.self) for the Wolf total.  Also builds the fp32 structure factor.
 * move:  d = { dLJ, dReal, dRecip } of mmc_trial_move's d[0..2] in fp32 for molecule i (1-based)
 *        moved to com_new / atoms_new; the device state is not touched. */
int32_t mmc_study_f32_total(mmc_ctx *ctx, double lj_rcut, double qq_rcut, int32_t mixed,
                            double out[6]);
int32_t mmc_study_f32_move(mmc_ctx *ctx, int64_t i, const double *com_new, const double *atoms_new,
                           double lj_rcut, double qq_rcut, int32_t mixed, double d[3],
                           int32_t *overlap);

/* ---- replica batch: R independent NVT chains of the same system on one GPU ------------------ */
typedef struct {
    int32_t mol;         /* 1-based molecule index of this replica's trial move */
    int32_t accept_prev; /* 1: the PREVIOUS proposal of this replica was accepted -> commit it
                            (Ewald/main.jl:598-621) before evaluating; 0: discard it (:622-629) */
    double com_new[3];   /* moa.COM[i] after the move            (Ewald/main.jl:527) */
    double atoms_new[9]; /* soa.coords[first:last] after the move (Ewald/main.jl:552) */
} mmc_move;

typedef struct {
    double d_lj;    /* partial_new_e - partial_old_e, LJ part     (Ewald/main.jl:491,557) */
    double d_real;  /* EwaldShort new - old, factor applied       (Ewald/main.jl:501,566) */
    double d_recip; /* RecipMove, factor applied; 0 when overlap  (Ewald/main.jl:580-590) */
    double d_vir;   /* virial new - old + d_recip/3               (Ewald/main.jl:600-601) */
    int32_t overlap;
    int32_t _pad;
} mmc_move_result;

/* All replicas start from the given configuration (3 atoms per molecule required: RecipMove's
 * `@assert n == 3`, Ewald/ewalds.jl:740).  Ewald tables are prepared as mmc_prepare_ewald does;
 * call mmc_batch_recip_long once before the first mmc_batch_eval. */
int32_t mmc_batch_create(int32_t device, void *hip_stream, int64_t n_replicas, int64_t n_mol,
                         const double *com, const double *coords, const int64_t *atype,
                         const double *charge, int64_t n_types, const double *eps,
                         const double *sig, double box, double kappa, int64_t nk,
                         int64_t k_sq_max, double factor, double lj_rcut, double qq_rcut,
                         mmc_batch **out);
int32_t mmc_batch_destroy(mmc_batch *b);
int32_t mmc_batch_set_replica(mmc_batch *b, int64_t r, const double *com, const double *coords);
int32_t mmc_batch_get_replica(mmc_batch *b, int64_t r, double *com, double *coords,
                              double *sum_old);
/* RecipLong for every replica; energies[r] WITHOUT factor. */
int32_t mmc_batch_recip_long(mmc_batch *b, double *energies);
/* potential(..., "ewald") for every replica. */
int32_t mmc_batch_potential_ewald(mmc_batch *b, mmc_totals *tot);
/* mmc_volume_change for every replica of the batch (they share one box). */
int32_t mmc_batch_volume_change(mmc_batch *b, double new_box, double new_kappa);
/* An NPT volume move of a ONE-replica batch without a host round trip -- the batch-side twin of
 * mmc_volume_trial / accept / reject, so that the trial moves of an NPT chain run on the batch's
 * fast paths (the move server: BASELINE configs[3], 10 000 molecules) and its volume moves on
 * the same state (the reference's only statement of the move: the docstring
 * Ewald/volumeChange.jl:59-147).  Trial: everything the move rewrites is copied aside on the device,
 * the system rescaled about the centres of mass (:62-80), the tables rebuilt for new_kappa
 * (ewalds.jl:45-103) and the total energy at the new volume evaluated (:91-111).  Accept (:132-147):
 * nothing to do.  Reject: the copy back, bit for bit.  A batch has ONE box: independent NPT
 * replicas are one batch each (MMC_ERR_UNSUPPORTED for more than one replica). */
int32_t mmc_batch_volume_trial(mmc_batch *b, double new_box, double new_kappa, mmc_totals *tot);
int32_t mmc_batch_volume_accept(mmc_batch *b);
int32_t mmc_batch_volume_reject(mmc_batch *b);
/* One trial move per replica, one launch: moves[r] -> results[r].  moves[r].accept_prev settles
 * the replica's previous proposal first.  Synchronous. */
int32_t mmc_batch_eval(mmc_batch *b, const mmc_move *moves, mmc_move_result *results);
/* Workgroups per replica-move used by mmc_batch_eval (1..32; 1 = one workgroup does the whole
 * move, >1 = the molecule range is split and the last workgroup does the reciprocal part). */
int32_t mmc_batch_set_parts(mmc_batch *b, int32_t n_parts);
/* Tuning switches (no effect on results beyond summation order):
 *   "parts"            as mmc_batch_set_parts
 *   "kernel"           2 = one wavefront per trial move, persistent workgroups, erfc(kappa r)/r
 *                      table; 1 = one workgroup per trial move (LDS-tiled, same table);
 *                      3 = 2 for launches of at least 16 moves per compute unit, else 1 (default
 *                      when every molecule has the same atom types and charges); 0 = generic;
 *                      4 = the latency form (k_move_eval_lat): "parts" (4, 8, ... 128) are waves,
 *                      four to a workgroup, each pair part's molecules resident in the wave, three
 *                      lanes to a neighbour, the reciprocal sum split over 1..3 waves; the form the
 *                      move server takes for few replicas -- same chains bit for bit as that server
 *   "wave_wgs"         workgroups of a kernel-2 launch (0 = 5 per compute unit: what its 96 VGPRs and
 *                      31 KB of LDS let be resident)
 *   "inject_torn"      N > 0: the native driver corrupts its first N copies of result records
 *                      before checking them, as a torn PCIe write would (test hook: the check must
 *                      refuse them and read again; mmc_run_stats.torn_records counts them)
 *   "zero_copy_moves"  1 = the kernel reads proposals from pinned host memory instead of an
 *                      H2D copy on the stream (lower latency for one replica, default 0)
 *   "device_moves"     1 = mmc_batch_run / mmc_batch_run_chains generate the trial moves on the
 *                      device (counter-based Philox4x32-10 keyed by seed + replica and the step
 *                      number; same move distributions as the host generator, a different random
 *                      stream): only one flag byte per replica and step crosses PCIe and no host
 *                      mirror of the coordinates is kept.  Default 0.
 *   "accept_on_device" who makes the Metropolis decision of a step of mmc_batch_run / _run_chains:
 *                      1 = the move kernel itself, where it can (device_moves, one part per move,
 *                      the wave kernel, no step-size adaptation): Metropolis(dU / T) && !overlap with
 *                      the same Philox uniform the host would take, its S-buffer bit and accept flag
 *                      kept in device memory for the next launch's commit, the decision sent along
 *                      in the result record (bit 31 of the stamp word) for the host's bookkeeping --
 *                      the same chains bit for bit, and a launch does not wait for the host's pass
 *                      over the previous launch's records.  0 = the host (its threads read every
 *                      record, decide, and write a flag byte per replica before the next launch:
 *                      ~50 ns per record and thread, hidden behind the other group's kernel while
 *                      there are about 8 threads per 32768-move launch).  -1 (default) = 1 where
 *                      a launch can take several steps ("steps_per_launch"), or where a host thread
 *                      would have more than 4096 records per launch to decide; else 0.
 *   "steps_per_launch" where the move kernel decides (above) and the caller asks for energies and
 *                      counts only (mmc_batch_run; not _run_chains, not the "trace_steps" hook), ONE
 *                      launch takes every replica of a group through this many consecutive steps
 *                      -- the same wave commits, evaluates and decides step after step, what one
 *                      step wrote and read is in the caches for the next -- and sends one record
 *                      per replica and launch: the sum of dU over its accepted steps and bit masks
 *                      (accepted, overlap, kind of move).  1, 2, 4, 8 or 16; 0 (default) = 8.  Same
 *                      chains bit for bit (counts, coordinates, S(k)); the running energies differ
 *                      from one step per launch by the order of a sum.  mmc_run_stats.launches
 *                      counts the launches.  1.43e8 against 1.31e8 moves/s (61440 chains of 750
 *                      molecules, one run of bench.py).  Launches this long (1.7 ms) want groups whose size is a multiple
 *                      of 5 * 4 * (compute units) replicas -- 5120 on MI355X: every wavefront the
 *                      kernel keeps resident then takes the same number of replicas.
 *   "image_by_molecule" -1 (default) = the wave kernel takes the minimum image of an atom pair with
 *                      the image of its molecule's centre of mass where that is the reference's
 *                      vector1D bit for bit: moves made on the device (rigid), and
 *                      gate + 2 r_mol < box / 2 (r_mol: the largest atom-to-centre distance of
 *                      anything uploaded).  0 = never: the per-pair minimum image.
 *   "persistent"       the move server for small batches (device_moves = 1, kernel != 0, no
 *                      orientations, at most one replica per compute unit): mmc_batch_run /
 *                      mmc_batch_run_chains launch ONE kernel per call, a workgroup per replica
 *                      whose waves are the parts of the move; per step the host posts one 8-byte
 *                      control word per replica in pinned memory (sequence number, result stamp,
 *                      accept bit of the previous step) and reads one 64-byte result record -- no
 *                      launch, no copy.  The chain is bit-identical to the launch-per-step driver
 *                      with n_parts = the server's waves (ceil(molecules / 64) + 1, at most 8 -- 5 for
 *                      a single replica -- or mmc_run_params.n_parts when > 1).  Every device-side wait is bounded (3 s):
 *                      a host that stops talking gets MMC_ERR_HIP from the run, not a hung GPU.
 *                      -1 (default) = use it for up to 256 replicas (one per compute unit) when it applies, 0 = never,
 *                      1 = insist (MMC_ERR_UNSUPPORTED from the run when it cannot be used)
 *   "trace_steps"      test hook: see mmc_batch_get_trace
 *   "server_wgs"       workgroups per replica of the move server: -1 (default) = 4 while each has
 *                      a compute unit to itself (up to 64 replicas), else 2 -- or, for a system too large for that, as many as it takes for
 *                      every pair wave to hold at most 128 molecules (10 000 molecules: 21) --
 *                      while R x workgroups does not exceed the compute units; 0 = one workgroup
 *                      per replica (k_move_server_wave); 2..32 = that many (k_move_server_lat: each
 *                      workgroup polls the replica's control word itself and keeps its own copy
 *                      of its molecules)
 *   "server_stall_ms"  test hook: the driver sleeps this long before posting the control words of
 *                      step 2 (the server's bounded wait must end the run with MMC_ERR_HIP)
 *   "server_seq_offset" test hook: the control words' sequence numbers (24 bits, compared modulo
 *                      2^24 on both sides) start at this offset instead of 0
 * A run that fails half-way (a server that timed out, a HIP error) leaves accepted moves, S-buffer
 * parity and the caller's energies in doubt: the batch then returns MMC_ERR_STATE from
 * mmc_batch_run / run_chains / eval until EVERY replica has been set again
 * (mmc_batch_set_replica); recompute the energies (mmc_batch_potential_ewald) after that. */
int32_t mmc_batch_set_option(mmc_batch *b, const char *key, int64_t value);
/* The fast kernel's approximation of erfc(kappa r)/r (ewalds.jl:367) evaluated at n values of
 * r^2 in (0, 256): lets a test bound its error against an exact evaluation. */
int32_t mmc_batch_qq_table(mmc_batch *b, const double *r2, int64_t n, double *out);
/* Radial-distribution histogram over every replica of the batch: the intent of Ewald/gr.jl
 * `makeRDF` with each replica as one frame.  One site per molecule -- site >= 0: that atom slot
 * (0 = the oxygens of water), site < 0: the centre of mass (gr.jl's cm mode); all pairs i < j,
 * gr.jl's minimum image (:75-80), bin = ceil(r / dr), dr = box / 2 / numbins (:5,87), counted
 * when bin <= numbins.  hist[0 .. numbins] (numbins + 1 counters). */
int32_t mmc_batch_rdf(mmc_batch *b, int32_t site, int32_t numbins, uint64_t *hist);
/* The reference's own move generation for device-side proposals ("device_moves"): orientations are
 * unit quaternions `totProps.quat[i]` and the atoms of a moved molecule are rebuilt from its
 * body-fixed sites, ra[a] = COM + MATMUL(q_to_a(ei), db[a]) (Ewald/main.jl:516-549;
 * quaternions.jl:11-50 q_to_a, :93-120 rotate_quaternion, :158-182 random_rotate_quaternion;
 * auxillary.jl:154-159 MATMUL).  quat: [n_mol][4] (w, x, y, z), given to every replica; db: [3][3]
 * body-fixed coordinates of the three sites.  mode 1 = q_to_a exactly as the reference has it,
 * including its element (2,3) `2*(q[2]*q[4] + q[1]*q[2])` (quaternions.jl:43; the Allen & Tildesley
 * original reads q[3]*q[4] + q[1]*q[2], so the reference's matrix is not orthogonal); mode 2 = the
 * Allen & Tildesley matrix; mode 0 = back to the default (the current atoms are translated /
 * rotated rigidly about the centre of mass, no quaternions kept).  A quaternion whose squared
 * norm is off by more than 1e-6 returns MMC_ERR_ASSERT (the reference prints and exit()s,
 * quaternions.jl:20-25).  An accepted move commits its quaternion (`totProps.quat[i] = ei`,
 * main.jl:619).  The coordinates given at creation are the caller's: the reference builds them
 * from the same quaternions (MakeAtomArrays, Ewald/setup.jl:447-537). */
int32_t mmc_batch_set_orientations(mmc_batch *b, const double *quat, const double *db,
                                   int32_t mode);
int32_t mmc_batch_get_orientations(mmc_batch *b, int64_t r, double *quat);
/* Result hand-off check (test hook): 1 if the 64-byte move-result record at `part_out_64` carries
 * launch stamp `stamp` and a matching checksum, else 0. */
int32_t mmc_part_validate(const void *part_out_64, uint32_t stamp);
/* Copy the raw 64-byte result record of (replica r, part) of the last mmc_batch_eval and the
 * stamp of that launch (test hook for the hand-off check). */
int32_t mmc_batch_peek_part(mmc_batch *b, int64_t r, int32_t part, void *out64, uint32_t *stamp);
/* Test hook: with option "trace_steps" = N the native driver records, for the first N steps of a
 * run and every replica, dU = d_lj + d_real + d_recip (main.jl:593) and the decision -- bit 0
 * accepted, bit 1 overlap, bit 2 the move was a rotation.  delta, flags: [R][N]. */
int32_t mmc_batch_get_trace(mmc_batch *b, double *delta, uint8_t *flags);
/* Settle the last outstanding proposals without evaluating new ones. */
int32_t mmc_batch_settle(mmc_batch *b, const int32_t *accept);

/* Native host driver: the sequential accept/reject of Loop() (Ewald/main.jl:487-644) for every
 * replica, in C++ on the host, around mmc_batch_eval's kernel.  Sweeps molecules in order like
 * the reference (`for i = 1:numbers.molecules`, :490). */
typedef struct {
    double temperature;  /* K                                (Ewald/main.jl:62)  */
    double dr_max;       /* translation box width, Angstrom  (Ewald/main.jl:118) */
    double dphi_max;     /* max rotation angle, rad          (Ewald/main.jl:73)  */
    uint64_t seed;       /* key of the run's random streams: chain r draws from the stream
                            (seed, replica0 + r) -- see "Random streams" below */
    int64_t n_steps;     /* trial moves per replica to run */
    int32_t n_groups;    /* replica groups pipelined on separate streams (>=1) */
    int32_t n_parts;     /* workgroups per replica-move (0 = choose) */
    int32_t time_kernels;/* N > 0: bracket every Nth launch of a group with HIP events, the
                            (N/2 + 1)th of each window of N (stats.kernel_ms over
                            stats.timed_launches); 0: none */
    int32_t n_threads;   /* host threads sharing the groups (0 or 1 = the calling thread only) */
    int32_t n_streams;   /* HIP streams the groups are spread over; 0 = choose (one per group with
                            host proposals; two with "device_moves": the launches of the groups
                            overlap -- stats.kernel_ms then sums SPANS of launches that share the
                            GPU, not costs; 1 = every launch alone on the GPU) */
    int32_t _pad;
    uint64_t replica0;   /* global index of this batch's replica 0 (a rank that owns chains
                            [g0, g0 + R) of a larger ensemble passes g0): the RANDOM DRAWS of a
                            chain depend on (seed, global index) only.  Its energies also depend,
                            in the last bits, on the order its dU terms are summed in, which the
                            batch picks from its own size (kernel by launch size, parts per move,
                            move server up to 256 replicas): shardings that use the same
                            "kernel", n_parts and "persistent" / "server_wgs" on every rank
                            reproduce one another bit for bit; others agree to ~1e-13 per move
                            and may part ways at a Metropolis comparison eventually */
} mmc_run_params;

/* Random streams.  Device-side proposals ("device_moves"): every draw is Philox4x32-10 with key =
 * seed (64 bit) and counter = (step number (64 bit), slot, global replica index), so streams of
 * different (seed, replica) pairs never coincide -- seeds that differ by less than the replica
 * count do NOT alias.  The step number continues across calls on the same batch (the batch counts
 * the steps it has run), so repeated runs with one seed do not replay their draws; the molecule
 * of step s of a call is still s mod n_mol, as Loop() restarts its sweep (Ewald/main.jl:490).
 * Host-side proposals: one xoshiro256++ stream per chain seeded from a hash of (seed, global
 * replica index, steps already run). */

typedef struct {
    int64_t moves, launches;
    int64_t trans_attempt, trans_accept, rot_attempt, rot_accept, overlaps;
    double wall_ms;      /* host wall clock over the run */
    double kernel_ms;    /* sum of HIP-event durations of the move kernel (time_kernels); with more
                            than one stream launches overlap and a duration is the launch's span */
    double energy_sum;   /* sum over replicas of the running total energy at the end */
    int64_t timed_launches; /* launches that contributed to kernel_ms */
    int64_t torn_records;   /* result records that carried the launch stamp but failed their
                               checksum when first read (re-read until whole) */
    int64_t server_steps;   /* steps that ran on the persistent move server (option "persistent"):
                               `launches` then counts control-word posts, not kernel launches */
    int64_t device_decisions; /* moves whose accept decision the move kernel made itself (option
                               "accept_on_device"): the host did their bookkeeping only */
} mmc_run_stats;

/* An NPT chain of a one-replica batch: n_sweeps times { moves_per_sweep trial moves (mmc_batch_run:
 * Loop(), Ewald/main.jl:487-644), then ONE volume move } -- Ewald/volumeChange.jl:59-147:
 *   vol_new = vol_old + (rand() - 0.5) * vmax                         :59
 *   test = exp(-beta (P dV - N ln(vol_new / vol_old) / beta + dE))    :129-130
 *   accepted if rand() < test                                         :132
 * with beta = 1 / temperature (energies are in K, so the pressure is in K / A^3), kappa = alpha /
 * L_new (Ewald/main.jl:290-291) and dE from the full recompute at the new volume.  The host decides;
 * the two uniforms of a volume move are Philox draws of the chain's own stream (seed, replica0)
 * at the step count it has reached, slot MMC_SLOT_VOLUME.  A move to a box below 2 r_cut is
 * rejected outright.  energy: in/out, the running total of the replica. */
#define MMC_SLOT_VOLUME 0x40000000u /* Philox slot of a volume move's two uniforms */
typedef struct {
    double pressure;         /* K / A^3 */
    double vmax;             /* A^3: dV is uniform in +- vmax / 2 */
    double alpha;            /* kappa * L (5.6 in Ewald/main.jl:290) */
    int64_t n_sweeps;
    int64_t moves_per_sweep; /* 0 = one per molecule (Ewald/main.jl:490) */
} mmc_npt_params;
typedef struct {
    int64_t vol_attempt, vol_accept;
    double box;              /* at the end */
    double volume_sum;       /* sum over the sweeps of the volume after each volume move */
    double volume_ms;        /* host wall clock spent in the volume moves */
} mmc_npt_stats;
int32_t mmc_batch_run_npt(mmc_batch *b, const mmc_run_params *p, const mmc_npt_params *q,
                          double *energy, mmc_run_stats *stats, mmc_npt_stats *npt_stats);

/* The driver's counter-based generator, exposed for known-answer tests and for callers that
 * want to re-derive a chain's draws: Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11). */
int32_t mmc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* energies: in/out running total energy per replica (R doubles), as `total.energy` (:599). */
int32_t mmc_batch_run(mmc_batch *b, const mmc_run_params *p, double *energies,
                      mmc_run_stats *stats);

/* Per-chain bookkeeping of Loop(): running totals, block-average accumulators and the two
 * step-size controllers.  One per replica; persists across mmc_batch_run_chains calls. */
typedef struct {
    double dr_max, dphi_max;       /* totProps.dr_max / totProps.dphi_max  (Ewald/main.jl:523,536) */
    double energy, virial;         /* total.energy, total.virial           (Ewald/main.jl:599-601) */
    double avg_energy, avg_virial; /* averages.energy / .virial: the running total added after
                                      EVERY trial move, accepted or not    (Ewald/main.jl:606-625) */
    int64_t steps_taken;           /* totProps.totalStepsTaken             (Ewald/main.jl:641) */
    int64_t overlaps;              /* ovr_count                            (Ewald/main.jl:595-597) */
    /* trans_moves / rot_moves :: Moves (Ewald/structs.jl), fields as used by Adjust! */
    int64_t trans_naccepp, trans_attempp, trans_naccept, trans_attempt;
    int64_t rot_naccepp, rot_attempp, rot_naccept, rot_attempt;
    double trans_set_value, rot_set_value; /* target acceptance ratios (Moves.set_value) */
} mmc_chain;

/* As mmc_batch_run, with every chain carrying its own step sizes and bookkeeping: p->dr_max and
 * p->dphi_max are ignored, chains[r].energy replaces energies[r].  After the trial move of the
 * last molecule of a sweep, Adjust!(trans_moves, box) and Adjust_rot!(rot_moves, box) run for the
 * chain exactly as at the end of Loop()'s inner loop (Ewald/main.jl:645-651, adjust.jl:1-83);
 * a controller that saw no attempt since its last call is left alone (the reference would divide
 * 0/0 there).  adjust = 0 keeps the step sizes fixed. */
int32_t mmc_batch_run_chains(mmc_batch *b, const mmc_run_params *p, mmc_chain *chains,
                             int32_t adjust, mmc_run_stats *stats);

/* The status line Loop() prints after every block (Ewald/main.jl:667-679), from one chain's record:
 *   "Block: %4d, Energy: %8.2f, Ratio trans: %4.2f, dr_max: %4.2f, Ratio rot: %4.2f, dphi_max: %4.2f,
 *    instant energy: %8.2f, overlap count: %4d, pressure: %8.2f"
 * with Energy = averages.energy / totalStepsTaken / n_mol, the two acceptance ratios
 * naccept / attempt, instant energy = total.energy / n_mol, and
 * pressure = ideal_term + total.virial / box^3: the reference hard-codes ideal_term = 4.60453
 * (main.jl:677); pass rho * T for auxillary.jl:121-123's Pressure(vir, rho, T, vol).  A ratio with
 * no attempt prints NaN, as Julia's 0/0 does.  Writes at most `len` bytes including the
 * terminator; returns MMC_ERR_ARG if the line does not fit. */
int32_t mmc_chain_block_line(const mmc_chain *chain, int64_t block, int64_t n_mol, double box,
                             double ideal_term, char *buf, int64_t len);

/* ---- the one collective of a sharded run: RCCL over xGMI ---------------------
 * Replicas shard over GPUs with no data-path collective; what is reduced, once per block, is a
 * handful of observables (sums of energies and acceptance counters, the maximum of the elapsed
 * time).  For hosts without torch (the reference's language is Julia): one process per GPU,
 *   rank 0:    mmc_dist_unique_id(id), then id goes to the other ranks by whatever the host has (a
 *              file, a socket, MPI);
 *   all ranks: mmc_dist_init(rank, world, id, device, &d)      -- ncclCommInitRank, collective;
 *              mmc_dist_reduce(d, sums, n_sum, maxima, n_max)  -- in place: two ncclAllReduce
 *              (ncclSum / ncclMax, fp64) on the communicator's own stream;
 *              mmc_dist_destroy(d).
 * librccl.so is loaded at the first of these calls (MMC_ERR_UNSUPPORTED if it is not there); the
 * library itself links libamdhip64 only. */
typedef struct mmc_dist mmc_dist;
int32_t mmc_dist_unique_id(uint8_t id[128]);
int32_t mmc_dist_init(int32_t rank, int32_t world, const uint8_t id[128], int32_t device,
                      mmc_dist **out);
int32_t mmc_dist_reduce(mmc_dist *d, double *sums, int64_t n_sum, double *maxima, int64_t n_max);
int32_t mmc_dist_destroy(mmc_dist *d);

#ifdef __cplusplus
}
#endif
#endif
