/*
 * mmc_oracle.h -- CPU restatement (plain C, fp64, single thread) of the per-move energy
 * hot path of BradenDKelly/MetropolisMonteCarlo (Julia).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may build, load or call anything under oracle/.  The product
 * (metropolismontecarlo_amd/ + libmmc_hip.so) never links or imports it.
 *
 * Parity status: the reference is Julia and no `julia` exists in the build image, so the
 * reference itself cannot be run.  The restatement follows the cited lines statement by
 * statement (same comparisons, same loop and summation order) and is pinned by
 *   - NIST SPC/E known answers for the four bundled configurations (E_fourier, E_self pin
 *     RecipLong/EwaldSelf/PrepareEwaldVariables; E_real pins the erfc pair term through the
 *     atom-cutoff EwaldReal of Ewald/ewald.jl:124-169; E_disp pins the LJ pair term through
 *     the monatomic LJ_dU of Ewald/energy.jl:294-340 applied to the oxygens),
 *   - the reference's own analytic tests (Ewald/tests.jl:8-82, :127-161; Monatomic
 *     mainMonatomic.jl:292-325),
 *   - an independent numpy/scipy statement of the same lines (oracle/numpy_check.py).
 * The molecular-cutoff variants of LJ_poly_dU / EwaldReal (COM gate, +100 slack) have no stored
 * expected value anywhere in the reference: for those the oracle is "parity unpinned" beyond the
 * three anchors above.
 *
 * All indices crossing this interface are 1-based and inclusive, exactly as the Julia arrays
 * (moa.firstAtom / moa.lastAtom, soa.atype), so a fixture dumped from Julia can be fed as is.
 * Coordinates are arrays of SVector{3,Float64}: 3 doubles per entry, x y z interleaved.
 * Tables (eps, sig) are Julia Matrix{Float64}: column-major n_types x n_types.
 */
#ifndef MMC_ORACLE_H
#define MMC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Ewald/constants.jl:24-28 */
double orc_factor(void);

/* Ewald/ewalds.jl:30-38 == Ewald/boundaries.jl:8-14 */
double orc_vector1D(double c1, double c2, double box);

/* Ewald/ewalds.jl:45-103.  Pass kxyz/cfac == NULL to only count.  Returns NKVECS, or -1 when the
 * reference's `@assert k_sq_max == 27` (:49) would throw. */
int64_t orc_prepare_ewald(double kappa, int64_t nk, int64_t k_sq_max, double box,
                          int32_t *kxyz /* [NKVECS][3] */, double *cfac /* [NKVECS] */);

/* Ewald/energy.jl:209-290 (moa/soa form; the legacy Requirements form :126-206 is the same
 * arithmetic with `a = 1:3` hard-coded). */
void orc_lj_poly_du(int64_t i, int64_t n_mol, const double *com, const int64_t *first_atom,
                    const int64_t *last_atom, const double *coords, const int64_t *atype,
                    int64_t n_types, const double *eps, const double *sig, double r_cut,
                    double box, double *pot, double *vir);

/* Ewald/ewalds.jl:293-376 (ovr = 0.5) and legacy :205-289 (ovr = 1.0).  No `factor`. */
void orc_ewald_real(int64_t chosen, int64_t n_mol, const double *com, const int64_t *first_atom,
                    const int64_t *last_atom, const double *coords, const double *charge,
                    double kappa, double r_cut, double box, double ovr, double *pot,
                    int32_t *overlap);

/* Ewald/ewalds.jl:892-910: e = EwaldReal * factor, v = e / 3. */
void orc_ewald_short(int64_t chosen, int64_t n_mol, const double *com, const int64_t *first_atom,
                     const int64_t *last_atom, const double *coords, const double *charge,
                     double kappa, double qq_rcut, double box, double factor, double *e,
                     double *v, int32_t *overlap);

/* Ewald/energy.jl:618-711 (bare Coulomb; diameter = r_cut*0.25+5, ovr = 1, atomic cutoff).
 * Returns -1 when `@assert r_cut == 10.0` (:648) would throw, else 0. */
int32_t orc_coulomb_real(int64_t chosen, int64_t n_mol, const double *com,
                         const int64_t *first_atom, const int64_t *last_atom,
                         const double *coords, const double *charge, double r_cut, double box,
                         double *pot, int32_t *overlap);

/* Ewald/ewald.jl:124-169: older atom-cutoff EwaldReal (no COM gate, r2 < r_cut^2).  Used only to
 * pin the erfc pair arithmetic on NIST E_real. */
double orc_ewald_real_atomcut(int64_t chosen, const int64_t *first_atom,
                              const int64_t *last_atom, int64_t n_atoms, const double *coords,
                              const double *charge, double kappa, double r_cut, double box);

/* Ewald/ewalds.jl:538-604.  sum_old/sum_new: NKVECS complex (re,im interleaved), both written
 * (:600-601).  Returns energy WITHOUT factor (:603). */
double orc_recip_long(int64_t nk, int64_t nkvecs, const int32_t *kxyz, const double *cfac,
                      int64_t n, const double *coords, const double *charge, double box,
                      double *sum_old, double *sum_new);

/* Ewald/ewalds.jl:718-826.  Mutates sum_new in place (:805-814); returns energy*factor (:825).
 * status: 0 ok, -1 when one of the asserts at :740-743 (n==3, k_sq_max==27, nk==5) would throw. */
int32_t orc_recip_move(double box, int64_t nk, int64_t k_sq_max, int64_t nkvecs,
                       const int32_t *kxyz, const double *cfac, const double *sum_old,
                       double *sum_new, const double *r_old, const double *r_new,
                       const double *q, int64_t n, double factor, double *d_energy);

/* Ewald/ewalds.jl:829-833 (factor applied). */
double orc_ewald_self(double kappa, double factor, int64_t n, const double *charge);

/* Properties fields written by the total-energy drivers (Ewald/auxillary.jl:37-45). */
typedef struct {
    double energy, virial, coulomb;
    double lj, real, recip, self; /* the values the reference println()s */
    int32_t n_overlap;            /* how many times "overlap after EwaldReal" would print */
} orc_totals;

/* Ewald/energy.jl:946-1032 ("ewald").  sum_old/sum_new as in orc_recip_long. */
void orc_potential_ewald(int64_t n_mol, int64_t n_atoms, const double *com,
                         const int64_t *first_atom, const int64_t *last_atom,
                         const double *coords, const int64_t *atype, const double *charge,
                         int64_t n_types, const double *eps, const double *sig, double lj_rcut,
                         double qq_rcut, double box, double kappa, int64_t nk, int64_t nkvecs,
                         const int32_t *kxyz, const double *cfac, double factor,
                         double *sum_old, double *sum_new, orc_totals *tot);

/* Ewald/energy.jl:864-943 (the reference's "Wolf" total).  The O(N^2) charge loop at :924-930
 * is executed literally when literal_prefactor != 0, otherwise as -(sum q)^2 * erfc/r_cut. */
void orc_potential_wolf(int64_t n_mol, int64_t n_atoms, const double *com,
                        const int64_t *first_atom, const int64_t *last_atom,
                        const double *coords, const int64_t *atype, const double *charge,
                        int64_t n_types, const double *eps, const double *sig, double lj_rcut,
                        double qq_rcut, double box, double kappa, double factor,
                        int32_t literal_prefactor, orc_totals *tot);

/* Monatomic/mainMonatomic.jl:227-272 == Ewald/energy.jl:294-340. */
void orc_lj_du_monatomic(int64_t i, int64_t n, const double *r, const double *eps,
                         const double *sig, double r_cut, double box, double *pot, double *vir);

/* Monatomic/mainMonatomic.jl:274-289 == Ewald/energy.jl:343-364: double count then halve. */
void orc_potential_monatomic(int64_t n, const double *r, const double *eps, const double *sig,
                             double r_cut, double box, double *energy, double *virial);

/* One trial move exactly as Loop() sequences the hot path (Ewald/main.jl:491-629), for molecule
 * i whose NEW centre of mass and atoms are given; the system arrays hold the OLD state and are
 * restored before return (the caller decides commit).  d[0]=E_new_LJ-E_old_LJ, d[1]=real new-old
 * (factor applied), d[2]=deltaRecip (factor applied, 0 when overlap, main.jl:580-590),
 * d[3]=virial new-old incl. recip/3; *overlap = overlap1||overlap2.  sum_new is left mutated as
 * RecipMove leaves it; commit/rollback are the caller's (main.jl:621,628). */
int32_t orc_trial_move(int64_t i, int64_t n_mol, double *com, const int64_t *first_atom,
                       const int64_t *last_atom, double *coords, const int64_t *atype,
                       const double *charge, int64_t n_types, const double *eps,
                       const double *sig, double lj_rcut, double qq_rcut, double box,
                       double kappa, int64_t nk, int64_t k_sq_max, int64_t nkvecs,
                       const int32_t *kxyz, const double *cfac, double factor,
                       const double *sum_old, double *sum_new, const double *com_new,
                       const double *atoms_new, double d[4], int32_t *overlap);

/* Timing helper for bench.py's cpu_baseline legs (not part of the restatement): n_threads
 * independent copies of the system, each running orc_trial_move on successive molecules with
 * small rigid translations (every move rejected) for `seconds`.  Returns the total number of
 * trial moves, *elapsed = the slowest thread's wall time. */
int64_t orc_bench_trial_moves(int64_t n_mol, int64_t n_atoms, const double *com,
                              const int64_t *first_atom, const int64_t *last_atom,
                              const double *coords, const int64_t *atype, const double *charge,
                              int64_t n_types, const double *eps, const double *sig,
                              double lj_rcut, double qq_rcut, double box, double kappa, int64_t nk,
                              int64_t k_sq_max, int64_t nkvecs, const int32_t *kxyz,
                              const double *cfac, double factor, const double *sum_old,
                              double dr_max, uint64_t seed, int32_t n_threads, double seconds,
                              double *elapsed);

#ifdef __cplusplus
}
#endif
#endif
