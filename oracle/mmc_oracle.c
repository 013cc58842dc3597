/*
 * mmc_oracle.c -- see mmc_oracle.h.  TEST INFRASTRUCTURE ONLY (checker, never the product).
 *
 * Written from the cited lines of /root/reference (Julia); nothing here is translated
 * mechanically -- Julia's SVector/@set/OffsetArray machinery is replaced by plain scalars and
 * flat arrays -- but every comparison operator, constant, loop nesting and summation order is
 * the reference's.  Build with -ffp-contract=off so that no FMA is formed (Julia does not
 * contract a*b+c either).
 */
#include "mmc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.141592653589793 /* Julia's Float64(pi) */

/* Ewald/constants.jl:24-28: factor = e1^2 / eps0 / 4 / pi / kb1 with eps0 *= 1e-10 */
double orc_factor(void)
{
    double kb1 = 1.3806488e-23;
    double eps01 = 8.854187817e-12;
    eps01 *= 1e-10;
    double e1 = 1.602176565e-19;
    return e1 * e1 / eps01 / 4 / ORC_PI / kb1;
}

/* Ewald/ewalds.jl:30-38 */
double orc_vector1D(double c1, double c2, double box)
{
    if (c1 < c2)
        return (c2 - c1) < (c1 - c2 + box) ? (c2 - c1) : (c2 - c1 - box);
    else
        return (c1 - c2) < (c2 - c1 + box) ? (c2 - c1) : (c2 - c1 + box);
}

/* Ewald/ewalds.jl:45-103 */
int64_t orc_prepare_ewald(double kappa, int64_t nk, int64_t k_sq_max, double box, int32_t *kxyz,
                          double *cfac)
{
    if (k_sq_max != 27) /* :49 */
        return -1;
    double b = 1.0 / 4.0 / kappa / kappa / box / box; /* :52 */
    double twopi = 2.0 * ORC_PI;
    double twopi_sq = twopi * twopi; /* twopi^2, :54 */
    int64_t n = 0;
    for (int64_t kx = 0; kx <= nk; kx++)           /* :71 */
        for (int64_t ky = -nk; ky <= nk; ky++)     /* :72 */
            for (int64_t kz = -nk; kz <= nk; kz++) /* :73 */
            {
                int64_t k_sq = kx * kx + ky * ky + kz * kz;
                if ((k_sq < k_sq_max) && (k_sq != 0)) { /* :76, strict */
                    if (kxyz) {
                        kxyz[3 * n + 0] = (int32_t)kx;
                        kxyz[3 * n + 1] = (int32_t)ky;
                        kxyz[3 * n + 2] = (int32_t)kz;
                    }
                    if (cfac) {
                        double kr_sq = twopi_sq * (double)k_sq;           /* :79 */
                        double c = twopi * exp(-b * kr_sq) / kr_sq / box; /* :80 */
                        if (kx > 0)
                            c = c * 2.0; /* :81-83 */
                        cfac[n] = c;
                    }
                    n++;
                }
            }
    return n;
}

/* Ewald/energy.jl:209-290 */
void orc_lj_poly_du(int64_t i, int64_t n_mol, const double *com, const int64_t *first_atom,
                    const int64_t *last_atom, const double *coords, const int64_t *atype,
                    int64_t n_types, const double *eps, const double *sig, double r_cut,
                    double box, double *pot_out, double *vir_out)
{
    const double *ri = com + 3 * (i - 1);
    int64_t startAtom = first_atom[i - 1], endAtom = last_atom[i - 1];
    double diameter = 0;                       /* :232 */
    double rm_cut_box = (r_cut + diameter);    /* :233 */
    double rm_cut_box_sq = rm_cut_box * rm_cut_box;
    double r_cut_sq = r_cut * r_cut;
    double pot = 0.0, vir = 0.0;

    for (int64_t j = 1; j <= n_mol; j++) { /* :242 */
        if (j == i)
            continue;
        const double *rj = com + 3 * (j - 1);
        double rij[3];
        for (int k = 0; k < 3; k++)
            rij[k] = orc_vector1D(ri[k], rj[k], box); /* :248-250 */
        double rij2 = rij[0] * rij[0] + rij[1] * rij[1] + rij[2] * rij[2];
        if (rij2 < rm_cut_box_sq) {                              /* :254 */
            for (int64_t a = 1; a <= (endAtom - startAtom + 1); a++) { /* :257 */
                const double *ra = coords + 3 * (startAtom - 1 + a - 1);
                int64_t ta = atype[startAtom - 1 + a - 1];
                for (int64_t b = first_atom[j - 1]; b <= last_atom[j - 1]; b++) { /* :260 */
                    const double *rb = coords + 3 * (b - 1);
                    double rab[3];
                    for (int k = 0; k < 3; k++)
                        rab[k] = orc_vector1D(ra[k], rb[k], box);
                    double rab2 = rab[0] * rab[0] + rab[1] * rab[1] + rab[2] * rab[2];
                    int64_t tb = atype[b - 1];
                    double e = eps[(ta - 1) + (tb - 1) * n_types];   /* :269 */
                    if (rab2 < (r_cut_sq + 100) && e > 0.001) {       /* :270 */
                        double s = sig[(ta - 1) + (tb - 1) * n_types];
                        double s2 = s * s / rab2;   /* :275 */
                        double s6 = s2 * s2 * s2;   /* ^3 */
                        double s12 = s6 * s6;       /* ^2 */
                        pot += e * (s12 - s6);
                        double virab = e * (2.0 * s12 - s6);
                        double f0 = rab[0] * virab * s2, f1 = rab[1] * virab * s2,
                               f2 = rab[2] * virab * s2; /* :280 */
                        vir += rij[0] * f0 + rij[1] * f1 + rij[2] * f2; /* :281 */
                    }
                }
            }
        }
    }
    *pot_out = pot * 4;          /* :289 */
    *vir_out = vir * 24 / 3.0;
}

/* Ewald/ewalds.jl:293-376 / :205-289 */
void orc_ewald_real(int64_t chosen, int64_t n_mol, const double *com, const int64_t *first_atom,
                    const int64_t *last_atom, const double *coords, const double *charge,
                    double kappa, double r_cut, double box, double ovr, double *pot_out,
                    int32_t *overlap)
{
    const double *ri = com + 3 * (chosen - 1);
    int64_t start_a = first_atom[chosen - 1], end_a = last_atom[chosen - 1];
    double diameter = 0; /* :312 */
    double rm_cut_box = (r_cut + diameter);
    double rm_cut_box_sq = rm_cut_box * rm_cut_box;
    double r_cut_sq = r_cut * r_cut;
    double pot = 0.0;
    *overlap = 0;
    for (int64_t j = 1; j <= n_mol; j++) { /* :329 */
        if (j == chosen)
            continue;
        const double *rj = com + 3 * (j - 1);
        double rij[3];
        for (int k = 0; k < 3; k++)
            rij[k] = orc_vector1D(ri[k], rj[k], box);
        double rij2 = rij[0] * rij[0] + rij[1] * rij[1] + rij[2] * rij[2];
        if (rij2 < rm_cut_box_sq) { /* :340 */
            for (int64_t a = start_a; a <= end_a; a++) {
                const double *ra = coords + 3 * (a - 1);
                for (int64_t b = first_atom[j - 1]; b <= last_atom[j - 1]; b++) {
                    const double *rb = coords + 3 * (b - 1);
                    double rab[3];
                    for (int k = 0; k < 3; k++)
                        rab[k] = orc_vector1D(ra[k], rb[k], box);
                    double rab2 = rab[0] * rab[0] + rab[1] * rab[1] + rab[2] * rab[2];
                    if ((rab2 < ovr) && (charge[a - 1] * charge[b - 1] < 0)) { /* :359 */
                        *pot_out = 0.0;
                        *overlap = 1;
                        return; /* :360 */
                    } else if (rab2 < (r_cut_sq + 100)) { /* :362 */
                        double rab_mag = sqrt(rab2);
                        pot += charge[a - 1] * charge[b - 1] * erfc(kappa * rab_mag) / rab_mag;
                    } else {
                        pot += 0.0;
                    }
                }
            }
        }
    }
    *pot_out = pot;
}

/* Ewald/ewalds.jl:892-910 */
void orc_ewald_short(int64_t chosen, int64_t n_mol, const double *com, const int64_t *first_atom,
                     const int64_t *last_atom, const double *coords, const double *charge,
                     double kappa, double qq_rcut, double box, double factor, double *e,
                     double *v, int32_t *overlap)
{
    double partial_e = 0.0, partial_v = 0.0, realEwald;
    orc_ewald_real(chosen, n_mol, com, first_atom, last_atom, coords, charge, kappa, qq_rcut, box,
                   0.5, &realEwald, overlap);
    realEwald *= factor;
    partial_e += realEwald;
    partial_v += (realEwald / 3);
    *e = partial_e;
    *v = partial_v;
}

/* Ewald/energy.jl:618-711 */
int32_t orc_coulomb_real(int64_t chosen, int64_t n_mol, const double *com,
                         const int64_t *first_atom, const int64_t *last_atom,
                         const double *coords, const double *charge, double r_cut, double box,
                         double *pot_out, int32_t *overlap)
{
    const double *ri = com + 3 * (chosen - 1);
    int64_t start_a = first_atom[chosen - 1], end_a = last_atom[chosen - 1];
    double diameter = r_cut * 0.25 + 5.0; /* :642 */
    double rm_cut_box = (r_cut + diameter);
    double rm_cut_box_sq = rm_cut_box * rm_cut_box;
    double r_cut_sq = r_cut * r_cut;
    if (r_cut != 10.0) /* :648 */
        return -1;
    double pot = 0.0, ovr = 1.0;
    *overlap = 0;
    for (int64_t j = 1; j <= n_mol; j++) {
        if (j == chosen)
            continue;
        const double *rj = com + 3 * (j - 1);
        double rij[3];
        for (int k = 0; k < 3; k++)
            rij[k] = orc_vector1D(ri[k], rj[k], box);
        double rij2 = rij[0] * rij[0] + rij[1] * rij[1] + rij[2] * rij[2];
        if (rij2 < rm_cut_box_sq) {
            for (int64_t a = start_a; a <= end_a; a++) {
                const double *ra = coords + 3 * (a - 1);
                for (int64_t b = first_atom[j - 1]; b <= last_atom[j - 1]; b++) {
                    const double *rb = coords + 3 * (b - 1);
                    double rab[3];
                    for (int k = 0; k < 3; k++)
                        rab[k] = orc_vector1D(ra[k], rb[k], box);
                    double rab2 = rab[0] * rab[0] + rab[1] * rab[1] + rab[2] * rab[2];
                    if ((rab2 < ovr) && (charge[a - 1] * charge[b - 1] < 0)) { /* :695 */
                        *pot_out = 0.0;
                        *overlap = 1;
                        return 0;
                    } else if (rab2 < r_cut_sq) { /* :699 */
                        pot += charge[a - 1] * charge[b - 1] / sqrt(rab2);
                    } else {
                        pot += 0.0;
                    }
                }
            }
        }
    }
    *pot_out = pot;
    return 0;
}

/* Ewald/ewald.jl:124-169 */
double orc_ewald_real_atomcut(int64_t chosen, const int64_t *first_atom,
                              const int64_t *last_atom, int64_t n_atoms, const double *coords,
                              const double *charge, double kappa, double r_cut, double box)
{
    int64_t start_i = first_atom[chosen - 1], end_i = last_atom[chosen - 1];
    double r_cut_sq = r_cut * r_cut;
    double pot = 0.0;
    for (int64_t i = start_i; i <= end_i; i++) {
        const double *ri = coords + 3 * (i - 1);
        for (int64_t j = 1; j <= n_atoms; j++) {
            if (j >= start_i && j <= end_i)
                continue; /* :151-153 */
            const double *rj = coords + 3 * (j - 1);
            double rij[3];
            for (int k = 0; k < 3; k++)
                rij[k] = orc_vector1D(ri[k], rj[k], box);
            double r2 = rij[0] * rij[0] + rij[1] * rij[1] + rij[2] * rij[2];
            if (r2 < r_cut_sq) { /* :160 */
                double r = sqrt(r2);
                pot += charge[i - 1] * charge[j - 1] * erfc(kappa * r) / r;
            } else {
                pot += 0.0;
            }
        }
    }
    return pot;
}

/* --- complex helpers with Julia's plain (unfused) arithmetic ---------------------------------- */
typedef struct { double re, im; } cplx;
static inline cplx c_mul(cplx a, cplx b)
{
    cplx r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re };
    return r;
}
static inline cplx c_conj(cplx a) { cplx r = { a.re, -a.im }; return r; }
static inline cplx c_rmul(double s, cplx a) { cplx r = { s * a.re, s * a.im }; return r; }

/* Phase tables e^{i 2 pi k x / L} for n atoms: the recurrence of ewalds.jl:558-585 (identical in
 * RecipMove :753-796).  ex is [n][nk+1] (k = 0..nk); ey, ez are [n][2nk+1] (k = -nk..nk). */
static void phase_tables(int64_t n, int64_t nk, const double *r, double L, cplx *ex, cplx *ey,
                         cplx *ez)
{
    double twopi = 2.0 * ORC_PI;
    int64_t wy = 2 * nk + 1, wx = nk + 1;
    for (int64_t j = 0; j < n; j++) {
        cplx one = { 1.0, 0.0 };
        ex[j * wx + 0] = one;
        ey[j * wy + nk] = one;
        ez[j * wy + nk] = one;
        double ax = twopi * (r[3 * j + 0]) / L, ay = twopi * (r[3 * j + 1]) / L,
               az = twopi * (r[3 * j + 2]) / L;
        cplx x1 = { cos(ax), sin(ax) }, y1 = { cos(ay), sin(ay) }, z1 = { cos(az), sin(az) };
        ex[j * wx + 1] = x1;
        ey[j * wy + nk + 1] = y1;
        ez[j * wy + nk + 1] = z1;
        ey[j * wy + nk - 1] = c_conj(y1);
        ez[j * wy + nk - 1] = c_conj(z1);
    }
    for (int64_t k = 2; k <= nk; k++)
        for (int64_t j = 0; j < n; j++) {
            ex[j * wx + k] = c_mul(ex[j * wx + k - 1], ex[j * wx + 1]);
            ey[j * wy + nk + k] = c_mul(ey[j * wy + nk + k - 1], ey[j * wy + nk + 1]);
            ez[j * wy + nk + k] = c_mul(ez[j * wy + nk + k - 1], ez[j * wy + nk + 1]);
            ey[j * wy + nk - k] = c_conj(ey[j * wy + nk + k]);
            ez[j * wy + nk - k] = c_conj(ez[j * wy + nk + k]);
        }
}

/* Ewald/ewalds.jl:538-604 */
double orc_recip_long(int64_t nk, int64_t nkvecs, const int32_t *kxyz, const double *cfac,
                      int64_t n, const double *coords, const double *charge, double box,
                      double *sum_old, double *sum_new)
{
    int64_t wy = 2 * nk + 1, wx = nk + 1;
    cplx *ex = (cplx *)malloc(sizeof(cplx) * n * wx); /* :554-556 allocates per call too */
    cplx *ey = (cplx *)malloc(sizeof(cplx) * n * wy);
    cplx *ez = (cplx *)malloc(sizeof(cplx) * n * wy);
    phase_tables(n, nk, coords, box, ex, ey, ez);
    double energy = 0.0;
    for (int64_t i = 0; i < nkvecs; i++) { /* :589 */
        cplx term = { 0.0, 0.0 };
        int64_t kx = kxyz[3 * i], ky = kxyz[3 * i + 1], kz = kxyz[3 * i + 2];
        for (int64_t l = 0; l < n; l++) { /* :591-597: ((q*ex)*ey)*ez */
            cplx t = c_rmul(charge[l], ex[l * wx + kx]);
            t = c_mul(t, ey[l * wy + nk + ky]);
            t = c_mul(t, ez[l * wy + nk + kz]);
            term.re += t.re;
            term.im += t.im;
        }
        /* real(conj(term)*term) = re*re - (-im)*im */
        energy += cfac[i] * (term.re * term.re - (-term.im) * term.im); /* :599 */
        sum_new[2 * i] = term.re;
        sum_new[2 * i + 1] = term.im;
        sum_old[2 * i] = term.re;
        sum_old[2 * i + 1] = term.im;
    }
    free(ex);
    free(ey);
    free(ez);
    return energy;
}

/* Ewald/ewalds.jl:718-826 */
int32_t orc_recip_move(double box, int64_t nk, int64_t k_sq_max, int64_t nkvecs,
                       const int32_t *kxyz, const double *cfac, const double *sum_old,
                       double *sum_new, const double *r_old, const double *r_new,
                       const double *q, int64_t n, double factor, double *d_energy)
{
    if (n != 3 || k_sq_max != 27 || nk != 5) /* :740-743 */
        return -1;
    cplx exn[3 * 6], eyn[3 * 11], ezn[3 * 11], exo[3 * 6], eyo[3 * 11], ezo[3 * 11];
    int64_t wy = 2 * nk + 1, wx = nk + 1;
    phase_tables(n, nk, r_new, box, exn, eyn, ezn);
    phase_tables(n, nk, r_old, box, exo, eyo, ezo);
    double energy = 0.0;
    for (int64_t i = 0; i < nkvecs; i++) { /* :803 */
        int64_t kx = kxyz[3 * i], ky = kxyz[3 * i + 1], kz = kxyz[3 * i + 2];
        for (int64_t l = 0; l < n; l++) { /* :804-814 */
            cplx tn = c_mul(c_mul(exn[l * wx + kx], eyn[l * wy + nk + ky]), ezn[l * wy + nk + kz]);
            cplx to = c_mul(c_mul(exo[l * wx + kx], eyo[l * wy + nk + ky]), ezo[l * wy + nk + kz]);
            cplx d = { tn.re - to.re, tn.im - to.im };
            d = c_rmul(q[l], d);
            sum_new[2 * i] += d.re;
            sum_new[2 * i + 1] += d.im;
        }
        double nr = sum_new[2 * i], ni = sum_new[2 * i + 1];
        double orr = sum_old[2 * i], oi = sum_old[2 * i + 1];
        energy += cfac[i] * ((nr * nr - (-ni) * ni) - (orr * orr - (-oi) * oi)); /* :817-821 */
    }
    *d_energy = energy * factor; /* :825 */
    return 0;
}

/* Ewald/ewalds.jl:829-833 */
double orc_ewald_self(double kappa, double factor, int64_t n, const double *charge)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; i++)
        s += charge[i] * charge[i];
    return -kappa * s / sqrt(ORC_PI) * factor;
}

/* Ewald/energy.jl:946-1032 */
void orc_potential_ewald(int64_t n_mol, int64_t n_atoms, const double *com,
                         const int64_t *first_atom, const int64_t *last_atom,
                         const double *coords, const int64_t *atype, const double *charge,
                         int64_t n_types, const double *eps, const double *sig, double lj_rcut,
                         double qq_rcut, double box, double kappa, int64_t nk, int64_t nkvecs,
                         const int32_t *kxyz, const double *cfac, double factor,
                         double *sum_old, double *sum_new, orc_totals *tot)
{
    memset(tot, 0, sizeof(*tot));
    double ener, vir, LJ = 0.0;
    for (int64_t i = 1; i <= n_mol; i++) { /* :972-977 */
        orc_lj_poly_du(i, n_mol, com, first_atom, last_atom, coords, atype, n_types, eps, sig,
                       lj_rcut, box, &ener, &vir);
        tot->energy += ener;
        tot->virial += vir;
        LJ += ener;
    }
    tot->energy = tot->energy / 2; /* :978-980 */
    tot->virial = tot->virial / 2;
    LJ = LJ / 2;
    tot->lj = LJ;

    double totReal = 0.0;
    for (int64_t i = 1; i <= n_mol; i++) { /* :991-1000 */
        int32_t overlap;
        orc_ewald_real(i, n_mol, com, first_atom, last_atom, coords, charge, kappa, qq_rcut, box,
                       0.5, &ener, &overlap);
        totReal += ener;
        if (overlap)
            tot->n_overlap++;
    }
    totReal *= factor / 2; /* :1001 */
    tot->energy += totReal;
    tot->coulomb += totReal;
    tot->virial += totReal / 3.0;
    tot->real = totReal;

    double recipEnergy =
        orc_recip_long(nk, nkvecs, kxyz, cfac, n_atoms, coords, charge, box, sum_old, sum_new);
    recipEnergy *= factor; /* :1009 */
    tot->energy += recipEnergy;
    tot->coulomb += recipEnergy;
    tot->virial += recipEnergy / 3.0;
    tot->recip = recipEnergy;

    double selfEnergy = orc_ewald_self(kappa, factor, n_atoms, charge); /* :1017 */
    tot->energy += selfEnergy;
    tot->coulomb += selfEnergy;
    tot->virial += selfEnergy / 3.0;
    tot->self = selfEnergy;
}

/* Ewald/energy.jl:864-943 */
void orc_potential_wolf(int64_t n_mol, int64_t n_atoms, const double *com,
                        const int64_t *first_atom, const int64_t *last_atom,
                        const double *coords, const int64_t *atype, const double *charge,
                        int64_t n_types, const double *eps, const double *sig, double lj_rcut,
                        double qq_rcut, double box, double kappa, double factor,
                        int32_t literal_prefactor, orc_totals *tot)
{
    memset(tot, 0, sizeof(*tot));
    double ener, vir, LJ = 0.0;
    for (int64_t i = 1; i <= n_mol; i++) { /* :889-894 */
        orc_lj_poly_du(i, n_mol, com, first_atom, last_atom, coords, atype, n_types, eps, sig,
                       lj_rcut, box, &ener, &vir);
        tot->energy += ener;
        tot->virial += vir;
        LJ += ener;
    }
    tot->energy = tot->energy / 2;
    tot->virial = tot->virial / 2;
    tot->lj = LJ / 2;

    double totReal = 0.0;
    for (int64_t i = 1; i <= n_mol; i++) { /* :908-917 */
        int32_t overlap;
        orc_ewald_real(i, n_mol, com, first_atom, last_atom, coords, charge, kappa, qq_rcut, box,
                       0.5, &ener, &overlap);
        totReal += ener;
        if (overlap)
            tot->n_overlap++;
    }
    totReal *= factor / 2; /* :918 */
    tot->energy += totReal;
    tot->coulomb += totReal; /* NB no virial contribution here (:919-920) */
    tot->real = totReal;

    double r_cut = lj_rcut; /* :875 -- the Wolf terms use sim_props.LJ_rcut */
    double prefactor = 0.0;
    if (literal_prefactor) {
        for (int64_t i = 0; i < n_atoms; i++) /* :925-929 */
            for (int64_t j = 0; j < n_atoms; j++)
                prefactor += charge[i] * charge[j] * erfc(kappa * r_cut) / r_cut;
    } else {
        double sq = 0.0;
        for (int64_t i = 0; i < n_atoms; i++)
            sq += charge[i];
        prefactor = sq * sq * erfc(kappa * r_cut) / r_cut;
    }
    prefactor *= -1; /* :930 */
    double qdot = 0.0;
    for (int64_t i = 0; i < n_atoms; i++)
        qdot += charge[i] * charge[i];
    double prefactor2 = (erfc(kappa * r_cut) / 2 / r_cut + kappa / sqrt(ORC_PI)) * qdot; /* :931 */
    tot->energy += (prefactor - prefactor2) * factor;
    tot->coulomb += (prefactor - prefactor2) * factor;
    tot->self = (prefactor - prefactor2) * factor;
}

/* Monatomic/mainMonatomic.jl:227-272 */
void orc_lj_du_monatomic(int64_t i, int64_t n, const double *r, const double *eps,
                         const double *sig, double r_cut, double box, double *pot_out,
                         double *vir_out)
{
    double rcut_sq = r_cut * r_cut;
    double pot = 0.0, vir = 0.0;
    const double *ri = r + 3 * (i - 1);
    for (int64_t j = 1; j <= n; j++) {
        if (j == i)
            continue;
        const double *rj = r + 3 * (j - 1);
        double d[3];
        for (int k = 0; k < 3; k++)
            d[k] = orc_vector1D(ri[k], rj[k], box);
        double rij_sq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        if (rij_sq > rcut_sq) { /* :249: r2 == rc2 is INCLUDED */
            pot += 0.0;
            vir += 0.0;
        } else {
            double sr2 = sig[j - 1] * sig[j - 1] / rij_sq;
            double sr6 = sr2 * sr2 * sr2;
            double sr12 = sr6 * sr6;
            pot += eps[j - 1] * (sr12 - sr6);
            vir += eps[j - 1] * (2 * sr12 - sr6);
        }
    }
    *pot_out = pot * 4.0;
    *vir_out = vir * 24.0 / 3.0;
}

/* Monatomic/mainMonatomic.jl:274-289 */
void orc_potential_monatomic(int64_t n, const double *r, const double *eps, const double *sig,
                             double r_cut, double box, double *energy, double *virial)
{
    double e = 0.0, v = 0.0, ener, vir;
    for (int64_t i = 1; i <= n; i++) {
        orc_lj_du_monatomic(i, n, r, eps, sig, r_cut, box, &ener, &vir);
        e += ener;
        v += vir;
    }
    *energy = e / 2;
    *virial = v / 2;
}

/* Ewald/main.jl:491-629, hot-path calls only */
int32_t orc_trial_move(int64_t i, int64_t n_mol, double *com, const int64_t *first_atom,
                       const int64_t *last_atom, double *coords, const int64_t *atype,
                       const double *charge, int64_t n_types, const double *eps,
                       const double *sig, double lj_rcut, double qq_rcut, double box,
                       double kappa, int64_t nk, int64_t k_sq_max, int64_t nkvecs,
                       const int32_t *kxyz, const double *cfac, double factor,
                       const double *sum_old, double *sum_new, const double *com_new,
                       const double *atoms_new, double d[4], int32_t *overlap)
{
    int64_t fa = first_atom[i - 1], la = last_atom[i - 1], na = la - fa + 1;
    if (na > 16)
        return -2;
    double old_e, old_v, e, v, new_e, new_v;
    int32_t o1, o2;
    double lj_old, lj_new, qq_old, qq_new;
    orc_lj_poly_du(i, n_mol, com, first_atom, last_atom, coords, atype, n_types, eps, sig, lj_rcut,
                   box, &old_e, &old_v); /* :491 */
    lj_old = old_e;
    orc_ewald_short(i, n_mol, com, first_atom, last_atom, coords, charge, kappa, qq_rcut, box,
                    factor, &e, &v, &o1); /* :501-502 */
    old_v += v;
    qq_old = e;

    double rm_old[3], ra_old[3 * 16]; /* :514-515 */
    memcpy(rm_old, com + 3 * (i - 1), sizeof(rm_old));
    memcpy(ra_old, coords + 3 * (fa - 1), sizeof(double) * 3 * na);
    memcpy(com + 3 * (i - 1), com_new, sizeof(rm_old));                 /* :527 */
    memcpy(coords + 3 * (fa - 1), atoms_new, sizeof(double) * 3 * na);  /* :552 */

    orc_lj_poly_du(i, n_mol, com, first_atom, last_atom, coords, atype, n_types, eps, sig, lj_rcut,
                   box, &new_e, &new_v); /* :557 */
    lj_new = new_e;
    orc_ewald_short(i, n_mol, com, first_atom, last_atom, coords, charge, kappa, qq_rcut, box,
                    factor, &e, &v, &o2); /* :566-567 */
    new_v += v;
    qq_new = e;

    *overlap = (o1 || o2); /* :574-578 */
    double deltaRecip = 0.0;
    int32_t st = 0;
    if (!*overlap) /* :580 */
        st = orc_recip_move(box, nk, k_sq_max, nkvecs, kxyz, cfac, sum_old, sum_new, ra_old,
                            atoms_new, charge + (fa - 1), na, factor, &deltaRecip);
    d[0] = lj_new - lj_old;
    d[1] = qq_new - qq_old;
    d[2] = deltaRecip;
    d[3] = (new_v - old_v) + deltaRecip / 3; /* :600-601 */

    memcpy(com + 3 * (i - 1), rm_old, sizeof(rm_old)); /* caller commits */
    memcpy(coords + 3 * (fa - 1), ra_old, sizeof(double) * 3 * na);
    return st;
}

/* ---- timing helper for bench.py's cpu_baseline (not part of the restatement) ------------------
 * n_threads independent copies of the system; each thread does Loop()'s hot-path calls
 * (orc_trial_move: 2x LJ_poly_dU + 2x EwaldShort + RecipMove) for successive molecules with small
 * rigid translations, rejecting every move (main.jl:628), until `seconds` have passed.  Returns
 * the total number of trial moves; *elapsed = wall time of the slowest thread. */
#include <pthread.h>
#include <stdlib.h>
#include <time.h>

typedef struct {
    int64_t n_mol, n_atoms, n_types, nk, k_sq_max, nkvecs, done;
    const double *com, *coords, *charge, *eps, *sig, *cfac, *sum_old;
    const int64_t *first_atom, *last_atom, *atype;
    const int32_t *kxyz;
    double lj_rcut, qq_rcut, box, kappa, factor, dr_max, seconds, elapsed;
    uint64_t seed;
} bench_arg;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *bench_thread(void *p)
{
    bench_arg *a = (bench_arg *)p;
    double *com = malloc(sizeof(double) * 3 * a->n_mol);
    double *coords = malloc(sizeof(double) * 3 * a->n_atoms);
    double *s_new = malloc(sizeof(double) * 2 * a->nkvecs);
    memcpy(com, a->com, sizeof(double) * 3 * a->n_mol);
    memcpy(coords, a->coords, sizeof(double) * 3 * a->n_atoms);
    uint64_t x = a->seed * 0x9e3779b97f4a7c15ULL + 1;
    const double t0 = now_s();
    int64_t n = 0;
    for (;;) {
        const int64_t i = n % a->n_mol + 1, fa = a->first_atom[i - 1], na = a->last_atom[i - 1] - fa + 1;
        double cn[3], an[3 * 16], d[4], dv[3];
        int32_t ov;
        for (int k = 0; k < 3; k++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17; /* xorshift64 */
            dv[k] = ((double)(x >> 11) * 0x1.0p-53 - 0.5) * a->dr_max;
            cn[k] = com[3 * (i - 1) + k] + dv[k];
        }
        for (int64_t q = 0; q < na && q < 16; q++)
            for (int k = 0; k < 3; k++)
                an[3 * q + k] = coords[3 * (fa - 1 + q) + k] + dv[k];
        memcpy(s_new, a->sum_old, sizeof(double) * 2 * a->nkvecs);
        orc_trial_move(i, a->n_mol, com, a->first_atom, a->last_atom, coords, a->atype, a->charge,
                       a->n_types, a->eps, a->sig, a->lj_rcut, a->qq_rcut, a->box, a->kappa, a->nk,
                       a->k_sq_max, a->nkvecs, a->kxyz, a->cfac, a->factor, a->sum_old, s_new, cn,
                       an, d, &ov);
        n++;
        if ((n & 15) == 0 && now_s() - t0 > a->seconds)
            break;
    }
    a->elapsed = now_s() - t0;
    a->done = n;
    free(com); free(coords); free(s_new);
    return NULL;
}

int64_t orc_bench_trial_moves(int64_t n_mol, int64_t n_atoms, const double *com,
                              const int64_t *first_atom, const int64_t *last_atom,
                              const double *coords, const int64_t *atype, const double *charge,
                              int64_t n_types, const double *eps, const double *sig,
                              double lj_rcut, double qq_rcut, double box, double kappa, int64_t nk,
                              int64_t k_sq_max, int64_t nkvecs, const int32_t *kxyz,
                              const double *cfac, double factor, const double *sum_old,
                              double dr_max, uint64_t seed, int32_t n_threads, double seconds,
                              double *elapsed)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    bench_arg *args = calloc((size_t)n_threads, sizeof(bench_arg));
    pthread_t *th = calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; t++) {
        bench_arg *a = &args[t];
        a->n_mol = n_mol; a->n_atoms = n_atoms; a->n_types = n_types; a->nk = nk;
        a->k_sq_max = k_sq_max; a->nkvecs = nkvecs;
        a->com = com; a->coords = coords; a->charge = charge; a->eps = eps; a->sig = sig;
        a->cfac = cfac; a->sum_old = sum_old; a->first_atom = first_atom; a->last_atom = last_atom;
        a->atype = atype; a->kxyz = kxyz;
        a->lj_rcut = lj_rcut; a->qq_rcut = qq_rcut; a->box = box; a->kappa = kappa;
        a->factor = factor; a->dr_max = dr_max; a->seconds = seconds; a->seed = seed + (uint64_t)t;
        pthread_create(&th[t], NULL, bench_thread, a);
    }
    int64_t total = 0;
    double worst = 0.0;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        total += args[t].done;
        if (args[t].elapsed > worst) worst = args[t].elapsed;
    }
    if (elapsed) *elapsed = worst;
    free(args); free(th);
    return total;
}
