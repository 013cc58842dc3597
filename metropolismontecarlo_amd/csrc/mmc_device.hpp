// mmc_device.hpp -- device-side building blocks shared by every kernel of libmmc_hip.so.
//
// gfx950 (CDNA4) only: 64-lane wavefronts, 256-thread workgroups (one wave per SIMD), LDS for the
// neighbour list, DPP/shuffle wave reductions.  No MFMA: nothing on this path is a dense
// contraction (pairwise erfc/LJ terms and a 3-factor phase product).
//
// The arithmetic of each pair term follows the reference statement by statement (comparisons,
// constants, order of operations inside a term); only the ORDER OF SUMMATION over pairs differs
// (tree reduction instead of the reference's sequential loop), which is worth ~1e-13 relative.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MMC_BLOCK 256
#define MMC_WAVES (MMC_BLOCK / 64)
#define MMC_LIST_CAP 2048 // neighbour-list slots per chunk (8 KiB of LDS)
#define MMC_MAX_ATOMS 16  // atoms in the CHOSEN molecule (neighbours are unbounded)
#define MMC_NKTAB 11      // phase-table width for nk = 5: k = -5..5
#define MMC_NK_STRIDE 352 // >= 337 k-vectors, 16-element aligned: entries per k-indexed device array

// One replica's state as the kernels see it.  Structure-of-arrays so that lane j reading
// molecule j (or atom j) is a unit-stride, fully coalesced HBM/L2 access.
struct SysView {
    const double *comx, *comy, *comz; // [n_mol]   moa.COM
    const double *ax, *ay, *az;       // [n_atoms] soa.coords
    const int32_t *first0;            // [n_mol]   moa.firstAtom - 1
    const int32_t *cnt;               // [n_mol]   lastAtom - firstAtom + 1
    const int32_t *atype;             // [n_atoms] soa.atype - 1
    const double *charge;             // [n_atoms] soa.charge
    const double *eps, *sig;          // [n_types^2] column-major Tables
    int32_t n_mol, n_atoms, n_types;
    double box;
};

// The chosen molecule in up to two states (0 = old, 1 = new), staged in LDS.
struct Chosen {
    int32_t i0, na;
    int32_t type[MMC_MAX_ATOMS];
    double q[MMC_MAX_ATOMS];
    double com[2][3];
    double at[2][MMC_MAX_ATOMS][3];
};

// An accepted-but-not-yet-written move of this replica: every reader substitutes it for
// molecule `mol` (so no inter-workgroup ordering is needed inside the launch that commits it).
struct Pending {
    int32_t mol; // 0-based, -1 = none
    int32_t _pad;
    double com[3];
    double at[MMC_MAX_ATOMS][3];
};

struct PairParams {
    double lj_gate_sq;  // (r_cut + diameter)^2, LJ          energy.jl:233-234
    double qq_gate_sq;  // (r_cut + diameter)^2, Coulomb     ewalds.jl:313-314
    double lj_slack_sq; // r_cut^2 + 100                     energy.jl:270
    double qq_slack_sq; // r_cut^2 + 100 (Ewald) or r_cut^2 (bare, energy.jl:699)
    double ovr;         // 0.5 (ewalds.jl:327), 1.0 (legacy :240, bare :652)
    double kappa;
};

struct PairAcc {
    double lj_pot, lj_vir, qq_pot;
    int32_t ovl;
};

// Ewald/ewalds.jl:30-38 (== boundaries.jl:8-14): minimum image by comparison, not by rounding.
// Branch-free but bit-identical to the reference's two-branch form:
//   c1 <  c2 (d > 0):  (c2-c1) < (c1-c2+box) ? d : d - box      c1-c2 == -d exactly, so the test is
//   c1 >= c2 (d <= 0): (c1-c2) < (c2-c1+box) ? d : d + box      |d| < fl(box - |d|) in both branches.
// * The test equals |d| < box/2 EXACTLY: box/2 is representable and rounding is monotone, so
//   |d| < box/2 => fl(box-|d|) >= box/2 > |d|, |d| > box/2 => fl(box-|d|) <= box/2 < |d|, and at
//   |d| == box/2 both are false.
// * The wrapped value d -+ box is fma(m, -box, d) with m = 0 or +-1 (m*box exact, one rounding,
//   and d + 0*(-box) == d): no 64-bit select.  5 VALU (sub, cmp, bfi, cndmask, fma) instead of 12.
struct BoxConsts {
    double half, neg, box; // box/2, -box and box, kept in registers across the pair loops
};
__device__ __forceinline__ BoxConsts box_consts(double box)
{
    BoxConsts b;
    b.half = 0.5 * box;
    b.neg = -box;
    b.box = box;
    return b;
}
__device__ __forceinline__ double vector1D(double c1, double c2, const BoxConsts &bc)
{
    const double d = c2 - c1;
    // m = 0 or copysign(1, d): one v_bfi_b32 and one v_cndmask_b32 on the high dword (the low
    // dwords of 0.0 and 1.0 are both zero)
    const double m = (fabs(d) < bc.half) ? 0.0 : copysign(1.0, d);
    return fma(m, bc.neg, d);
}
// The MAGNITUDE of the minimum image, for callers that only square it (an r^2):
//     |vector1D(c1, c2)| == min(|d|, fl(box - |d|))     bit for bit,
// because the reference keeps d when |d| < fl(box - |d|) and otherwise returns d -+ box, whose
// magnitude is fl(box - |d|) (negation is exact; at a tie both candidates are the same number).
// 3 VALU (sub, sub, min) instead of 6; a component beyond the box (|d| > box, atoms carried
// outside by their molecule) gives box - |d| < 0, whose square is again the reference's.
__device__ __forceinline__ double vector1D_abs(double c1, double c2, const BoxConsts &bc)
{
    const double d = fabs(c2 - c1);
    return fmin(d, bc.box - d);
}
__device__ __forceinline__ double vector1D(double c1, double c2, double box)
{
    return vector1D(c1, c2, box_consts(box));
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ int lanes_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// Lane masks straight from the compare's SGPR pair.  HIP's __ballot / __any go through an integer
// (v_cndmask 0/1, v_cmp_ne): two VALU instructions per use, 8 % of an atom-pair evaluation.
__device__ __forceinline__ unsigned long long wave_ballot(bool p)
{
    return __builtin_amdgcn_ballot_w64(p);
}
__device__ __forceinline__ bool wave_any(bool p)
{
    return __builtin_amdgcn_ballot_w64(p) != 0ULL;
}

// sin and cos of a moderate angle (here 2 pi c / L with c within a few box lengths): quadrant
// reduction by a two-part pi/2 (Cody-Waite: n * pio2_1 is exact for n < 2^20, so the reduction
// holds its 1e-16 absolute accuracy up to |x| ~ 8e5) and the fdlibm kernel polynomials on
// |r| <= pi/4 (< 1 ulp each) -- the accuracy class of the libm sin/cos the reference calls, in
// ~45 instructions and 16 registers instead of ocml's ~200 / 60 (whose Payne-Hanek path for huge
// arguments is what costs).  An atom more than 1e5 box lengths outside its box is first folded
// back by whole periods, at the accuracy such a coordinate has left.
// A double constant materialised in scalar registers where it is used (two s_mov_b32): the
// persistent wave kernels sit at their register limits, and LLVM otherwise hoists the constants of
// this polynomial out of the unit loop into VGPRs that it then spills to scratch -- every reload a
// trip to memory in the middle of a dependent chain (11 of them per unit in round 2's build).
__device__ __forceinline__ double scalar_const(double c)
{
    asm volatile("" : "+s"(c));
    return c;
}
__device__ __forceinline__ void sincos_moderate(double x, double &sn, double &cs)
{
    if (!(fabs(x) < 8.0e5))
        x = fma(-6.283185307179586, rint(x * 0.15915494309189535), x);
    const double fn = rint(x * 6.36619772367581382433e-01);          // x * 2/pi
    double r = fma(-fn, 1.57079632673412561417e+00, x);              // pio2_1 (33 bits)
    r = fma(-fn, 6.07710050650619224932e-11, r);                     // pio2_1t
    const int n = (int)fn;
    const double z = r * r;
    double ps = scalar_const(1.58969099521155010221e-10);             // S6
    ps = fma(ps, z, scalar_const(-2.50507602534068634195e-08));
    ps = fma(ps, z, scalar_const(2.75573137070700676789e-06));
    ps = fma(ps, z, scalar_const(-1.98412698298579493134e-04));
    ps = fma(ps, z, scalar_const(8.33333333332248946124e-03));
    ps = fma(ps, z, scalar_const(-1.66666666666666324348e-01));       // S1
    const double s = fma(z * r, ps, r);
    double pc = scalar_const(-1.13596475577881948265e-11);            // C6
    pc = fma(pc, z, scalar_const(2.08757232129817482790e-09));
    pc = fma(pc, z, scalar_const(-2.75573143513906633035e-07));
    pc = fma(pc, z, scalar_const(2.48015872894767294178e-05));
    pc = fma(pc, z, scalar_const(-1.38888888888741095749e-03));
    pc = fma(pc, z, scalar_const(4.16666666666666019037e-02));        // C1
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const double a = (n & 1) ? c : s, b = (n & 1) ? s : c;
    sn = (n & 2) ? -a : a;
    cs = ((n + 1) & 2) ? -b : b;
}

// lane `src`'s value in every lane (ds_bpermute; no lane id involved, unlike __shfl, whose
// __lane_id() LLVM hoists out of a persistent loop and keeps in a register for the kernel's life)
__device__ __forceinline__ double wave_pick(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v; // lane 0 holds the sum
}

// ---- DPP forms (no LDS crossbar): for the latency kernels, where a ds_bpermute tree of six sums is
// a visible part of a unit's critical path.  row_shr:N = lane i reads lane i - N of its row of 16
// lanes, 0 beyond the start of the row.
template <int N> __device__ __forceinline__ int dpp_row_shr(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xf, 0xf, true);
}
template <int N> __device__ __forceinline__ double dpp_row_shr_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)dpp_row_shr<N>((int)b), hi = (unsigned)dpp_row_shr<N>((int)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// Sum over the 64 lanes, the total in EVERY lane (wave-uniform): an inclusive scan inside each row
// of 16 (1, 2, 4, 8), then the four row totals added in row order.  Fixed order -> reproducible.
__device__ __forceinline__ double wave_sum_rows(double v)
{
    v += dpp_row_shr_f64<1>(v);
    v += dpp_row_shr_f64<2>(v);
    v += dpp_row_shr_f64<4>(v);
    v += dpp_row_shr_f64<8>(v);
    const long long b = __double_as_longlong(v);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)b, 16 * k + 15);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), 16 * k + 15);
        r[k] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    return ((r[0] + r[1]) + r[2]) + r[3];
}

// N such sums at once, step by step across all of them: the six sums of a unit are independent
// chains of ~15 dependent instructions each, and written one after the other that is how they were
// scheduled (0.47 us of a latency unit's 2.4); in lockstep they overlap.  Same operations, same
// order within each sum: the same bits as wave_sum_rows.
template <int N> __device__ __forceinline__ void wave_sum_rows_n(double (&v)[N])
{
#pragma unroll
    for (int k = 0; k < N; k++) v[k] += dpp_row_shr_f64<1>(v[k]);
#pragma unroll
    for (int k = 0; k < N; k++) v[k] += dpp_row_shr_f64<2>(v[k]);
#pragma unroll
    for (int k = 0; k < N; k++) v[k] += dpp_row_shr_f64<4>(v[k]);
#pragma unroll
    for (int k = 0; k < N; k++) v[k] += dpp_row_shr_f64<8>(v[k]);
    double r[N][4];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const long long b = __double_as_longlong(v[k]);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)b, 16 * q + 15);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), 16 * q + 15);
            r[k][q] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        }
    }
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = r[k][0] + r[k][1];
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = v[k] + r[k][2];
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = v[k] + r[k][3];
}

// Sum NV values over the 256-thread workgroup; thread 0 gets the totals in out[].
// red: LDS scratch of NV * MMC_WAVES doubles.  Fixed order -> bitwise reproducible.
template <int NV>
__device__ __forceinline__ void block_sum(const double (&v)[NV], double *red, double (&out)[NV])
{
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double s = wave_sum(v[k]);
        if (lane_id() == 0)
            red[k * MMC_WAVES + wave_id()] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < MMC_WAVES; w++)
                s += red[k * MMC_WAVES + w];
            out[k] = s;
        }
    }
    __syncthreads();
}

// The same sum for many values at once, through an LDS transpose instead of NV separate 64-lane
// shuffle trees: every thread parks its NV partials, then NV*32 threads each add 8 of them
// (unit-stride, conflict-free) and a 32-lane tree finishes one value per half-wave.  About 4x
// fewer LDS operations than block_sum for NV = 7.  buf: NV*256 doubles of LDS; res: NV doubles of
// LDS (thread 0 reads the totals from it after the call).  `flag` (0/1 per thread) is OR-reduced
// over the workgroup into *flag_out for free.  Fixed order -> bitwise reproducible.
template <int NV>
__device__ __forceinline__ void block_sum_wide(const double (&v)[NV], double *buf, double *res,
                                               int flag, int32_t *wflag)
{
#pragma unroll
    for (int k = 0; k < NV; k++)
        buf[k * MMC_BLOCK + threadIdx.x] = v[k];
    const unsigned long long b0 = __ballot((flag & 1) != 0), b1 = __ballot((flag & 2) != 0);
    if (lane_id() == 0)
        wflag[wave_id()] = (b0 != 0ULL ? 1 : 0) | (b1 != 0ULL ? 2 : 0);
    __syncthreads();
    const int t = threadIdx.x;
    if (t < NV * 32) { // NV <= 8
        const int k = t >> 5, c = t & 31;
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < MMC_BLOCK / 32; i++)
            s += buf[k * MMC_BLOCK + i * 32 + c];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1)
            s += __shfl_down(s, off, 32);
        if (c == 0)
            res[k] = s;
    }
    __syncthreads();
}

// STYLE 0: erfc-damped Coulomb (EwaldReal); STYLE 1: bare Coulomb (CoulombReal).
//
// Scan molecules [j_begin, j_end) of replica `s` against the chosen molecule in NS states.
//   phase A  one lane per molecule j: COM minimum image against each state, gate test
//            (energy.jl:248-254, ewalds.jl:334-340); survivors are compacted into an LDS list,
//            each wave owning a contiguous j-range and a contiguous list segment so the list is
//            in ascending j whatever the wave timing (deterministic summation order);
//   phase B  one lane per (neighbour, atom a of the chosen molecule): loop over the neighbour's
//            atoms b, both states share the loads of b (energy.jl:257-285, ewalds.jl:343-372).
// acc[] accumulates per thread; the caller reduces.
template <int NS, bool LJ, bool QQ, int STYLE, bool PEND>
__device__ __forceinline__ void pair_scan(const SysView &s, const Chosen *ch, const Pending *pd,
                                          int j_begin, int j_end, const PairParams &pp,
                                          int32_t *list, int32_t *wcnt, PairAcc (&acc)[NS])
{
    const double box = s.box;
    const int i0 = ch->i0, na = ch->na;
    const int pend = PEND ? pd->mol : -1;
    const double gate = fmax(LJ ? pp.lj_gate_sq : 0.0, QQ ? pp.qq_gate_sq : 0.0);

    for (int jb = j_begin; jb < j_end; jb += MMC_LIST_CAP) {
        const int je = min(jb + MMC_LIST_CAP, j_end);
        const int len = je - jb;
        const int seg = ((len + MMC_WAVES * 64 - 1) / (MMC_WAVES * 64)) * 64; // per-wave slots
        const int w = wave_id();
        const int wj0 = jb + w * seg, wj1 = min(wj0 + seg, je);

        // ---- phase A ----
        int count = 0; // wave-uniform
        for (int base = wj0; base < wj1; base += 64) {
            const int j = base + lane_id();
            bool keep = false;
            if (j < wj1 && j != i0) {
                double cx, cy, cz;
                if (PEND && j == pend) {
                    cx = pd->com[0]; cy = pd->com[1]; cz = pd->com[2];
                } else {
                    cx = s.comx[j]; cy = s.comy[j]; cz = s.comz[j];
                }
#pragma unroll
                for (int st = 0; st < NS; st++) {
                    double dx = vector1D(ch->com[st][0], cx, box);
                    double dy = vector1D(ch->com[st][1], cy, box);
                    double dz = vector1D(ch->com[st][2], cz, box);
                    double r2 = dx * dx + dy * dy + dz * dz;
                    keep = keep || (r2 < gate);
                }
            }
            unsigned long long m = __ballot(keep);
            if (keep)
                list[w * seg + count + lanes_below(m)] = j;
            count += __popcll(m);
        }
        if (lane_id() == 0)
            wcnt[w] = count;
        __syncthreads();

        // ---- phase B ----
        int c0 = wcnt[0], c1 = wcnt[1], c2 = wcnt[2], c3 = wcnt[3];
        const int total = c0 + c1 + c2 + c3;
        for (int g = threadIdx.x; g < total * na; g += MMC_BLOCK) {
            const int n = g / na, a = g - n * na;
            int slot;
            if (n < c0) slot = n;
            else if (n < c0 + c1) slot = seg + (n - c0);
            else if (n < c0 + c1 + c2) slot = 2 * seg + (n - c0 - c1);
            else slot = 3 * seg + (n - c0 - c1 - c2);
            const int j = list[slot];
            const bool jp = PEND && (j == pend);

            double cx, cy, cz;
            if (jp) { cx = pd->com[0]; cy = pd->com[1]; cz = pd->com[2]; }
            else { cx = s.comx[j]; cy = s.comy[j]; cz = s.comz[j]; }
            double rij[NS][3];
            bool lj_on[NS], qq_on[NS];
#pragma unroll
            for (int st = 0; st < NS; st++) {
                rij[st][0] = vector1D(ch->com[st][0], cx, box);
                rij[st][1] = vector1D(ch->com[st][1], cy, box);
                rij[st][2] = vector1D(ch->com[st][2], cz, box);
                double r2 = rij[st][0] * rij[st][0] + rij[st][1] * rij[st][1] +
                            rij[st][2] * rij[st][2];
                lj_on[st] = LJ && (r2 < pp.lj_gate_sq);
                qq_on[st] = QQ && (r2 < pp.qq_gate_sq);
            }
            const int fb = s.first0[j], nb = s.cnt[j];
            const int ta = ch->type[a];
            const double qa = ch->q[a];
            for (int b = 0; b < nb; b++) {
                double bx, by, bz;
                if (jp) { bx = pd->at[b][0]; by = pd->at[b][1]; bz = pd->at[b][2]; }
                else { bx = s.ax[fb + b]; by = s.ay[fb + b]; bz = s.az[fb + b]; }
                const int tb = s.atype[fb + b];
                const double qb = QQ ? s.charge[fb + b] : 0.0;
                double e = 0.0, sg = 0.0;
                if (LJ) {
                    e = s.eps[ta + tb * s.n_types];
                    sg = s.sig[ta + tb * s.n_types];
                }
#pragma unroll
                for (int st = 0; st < NS; st++) {
                    double rx = vector1D(ch->at[st][a][0], bx, box);
                    double ry = vector1D(ch->at[st][a][1], by, box);
                    double rz = vector1D(ch->at[st][a][2], bz, box);
                    double rab2 = rx * rx + ry * ry + rz * rz;
                    if (QQ && qq_on[st]) {
                        if ((rab2 < pp.ovr) && (qa * qb < 0)) {
                            acc[st].ovl = 1;
                        } else if (rab2 < pp.qq_slack_sq) {
                            if (STYLE == 0) {
                                double rab_mag = sqrt(rab2);
                                acc[st].qq_pot += qa * qb * erfc(pp.kappa * rab_mag) / rab_mag;
                            } else {
                                acc[st].qq_pot += qa * qb / sqrt(rab2);
                            }
                        }
                    }
                    if (LJ && lj_on[st]) {
                        if (rab2 < pp.lj_slack_sq && e > 0.001) {
                            double s2 = sg * sg / rab2;
                            double s6 = s2 * s2 * s2;
                            double s12 = s6 * s6;
                            acc[st].lj_pot += e * (s12 - s6);
                            double virab = e * (2.0 * s12 - s6);
                            double f0 = rx * virab * s2, f1 = ry * virab * s2,
                                   f2 = rz * virab * s2;
                            acc[st].lj_vir += rij[st][0] * f0 + rij[st][1] * f1 +
                                              rij[st][2] * f2;
                        }
                    }
                }
            }
        }
        __syncthreads(); // list is reused by the next chunk
    }
}

// ---- complex helpers (plain, unfused: the reference's Complex{Float64} arithmetic) -------------
struct cplx { double re, im; };
__device__ __forceinline__ cplx c_mul(cplx a, cplx b)
{
    cplx r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}
// The same product with one rounding less per component (4 instructions instead of 6), for the
// k-vector loop of the throughput kernel: no decision hangs on these values, and the sums stay
// within a few 1e-16 relative of the unfused ones (the reference's tolerance is 1e-6).
__device__ __forceinline__ cplx c_mul_fused(cplx a, cplx b)
{
    cplx r;
    r.re = fma(a.re, b.re, -(a.im * b.im));
    r.im = fma(a.re, b.im, a.im * b.re);
    return r;
}
__device__ __forceinline__ cplx c_conj(cplx a) { cplx r = { a.re, -a.im }; return r; }
__device__ __forceinline__ cplx c_rmul(double q, cplx a) { cplx r = { q * a.re, q * a.im }; return r; }

#define MMC_TWOPI (2.0 * 3.141592653589793)

// One row of the e^{i 2 pi k x / L} table, k = -5..5, by the recurrence of ewalds.jl:564-585
// (same in RecipMove :762-796): cos/sin for k = 1, products above, conjugates below.
__device__ __forceinline__ void phase_row(double x, double L, cplx *row /* [MMC_NKTAB] */)
{
    double ang = MMC_TWOPI * x / L;
    double sn, cs;
    sincos(ang, &sn, &cs);
    cplx e1 = { cs, sn };
    cplx one = { 1.0, 0.0 };
    row[5] = one;
    row[6] = e1;
    row[4] = c_conj(e1);
    cplx p = e1;
#pragma unroll
    for (int k = 2; k <= 5; k++) {
        p = c_mul(p, e1);
        row[5 + k] = p;
        row[5 - k] = c_conj(p);
    }
}
