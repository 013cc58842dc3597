// mmc_total.hpp -- K2/K3 for homogeneous 3-atom systems: the total energy
// `potential(moa, soa, tot, ewalds, vdwTable, sim_props, "ewald")` (Ewald/energy.jl:946-1032)
// with each molecule pair visited ONCE.
//
// The reference sums LJ_poly_dU(i) and EwaldReal(i) over all i and halves (energy.jl:972-1001):
// every pair twice.  Pair terms and the COM gate are symmetric in (i, j), so
//     sum_i E_i / 2 = sum_{i<j} e_ij
// holds term by term; only the summation order changes (~1e-15 relative).  The one asymmetric
// piece is the overlap sentinel (EwaldReal returns 0.0 for a molecule that overlaps,
// ewalds.jl:359-360): the kernel counts overlapping atom pairs and the host falls back to the
// per-molecule kernels for a replica that has any, so the quirk is reproduced exactly.
//
//   k_total_pairs   grid (tile pairs I<=J, R): two 64-molecule record tiles in LDS, COM gate of
//                   the 64x64 molecule pairs -> compacted list -> one lane per (pair, a, b) with
//                   the erfc(kappa r)/r table of mmc_fast.hpp; LJ pass for eps > 0.001 pairs.
//   k_total_sum     per-replica sum of the tile-pair partials, fixed order.
//   k_atom_phases   cos/sin(2 pi x/L) for every atom once (6 doubles), instead of once per
//                   (kx, ky) column as k_recip_long does.
//   k_recip_long_ph k_recip_long reading those phases.
#pragma once
#include "mmc_fast.hpp"

#define MMC_TM 64 // molecules per tile

struct TotalPart {
    double lj_pot, lj_vir, qq;
    int32_t n_ovl, _pad;
};

struct TotalShared {
    alignas(16) double ti[MMC_TM * MMC_REC]; // 6 KB each; ti doubles as reduction scratch
    alignas(16) double tj[MMC_TM * MMC_REC];
    alignas(16) double qtab[MMC_QQ_TABLE_DOUBLES];
    double red[4];
    double qq9[9], ljp_eps[9], ljp_sig[9];
    uint16_t list[MMC_TM * MMC_TM];
    int32_t ljp_ab[9];
    int32_t wcnt[MMC_WAVES];
};
static_assert(2 * MMC_TM * MMC_REC >= 4 * MMC_BLOCK, "tiles double as reduction scratch");

__global__ __launch_bounds__(MMC_BLOCK) void k_total_pairs(BatchView bv, const double *rec,
                                                           const double *qq_tab, FastConsts fc,
                                                           PairParams pp, const int16_t *tile_pairs,
                                                           int n_pairs, TotalPart *out)
{
    __shared__ __align__(16) TotalShared sm;
    const int r = blockIdx.y, tp = blockIdx.x, tid = threadIdx.x;
    const int n_mol = bv.n_mol;
    const double box = bv.box;
    const int ti_idx = tile_pairs[2 * tp], tj_idx = tile_pairs[2 * tp + 1];
    const bool diag = ti_idx == tj_idx;
    const int i0 = ti_idx * MMC_TM, j0 = tj_idx * MMC_TM;
    const int ni = min(MMC_TM, n_mol - i0), nj = min(MMC_TM, n_mol - j0);
    const double *myrec = rec + (int64_t)r * n_mol * MMC_RSTRIDE;

    // tiles are runs of consecutive records (96 of every 128 bytes): 16-byte copies
    for (int g = tid; g < ni * 6; g += MMC_BLOCK)
        *reinterpret_cast<double2 *>(&sm.ti[2 * g]) = *reinterpret_cast<const double2 *>(
            myrec + (int64_t)(i0 + g / 6) * MMC_RSTRIDE + 2 * (g % 6));
    for (int g = tid; g < nj * 6; g += MMC_BLOCK)
        *reinterpret_cast<double2 *>(&sm.tj[2 * g]) = *reinterpret_cast<const double2 *>(
            myrec + (int64_t)(j0 + g / 6) * MMC_RSTRIDE + 2 * (g % 6));
    for (int k = tid; k < MMC_QQ_TABLE_DOUBLES; k += MMC_BLOCK)
        sm.qtab[k] = qq_tab[k];
    if (tid < 9) {
        sm.qq9[tid] = fc.qq9[tid];
        sm.ljp_eps[tid] = fc.ljp_eps[tid];
        sm.ljp_sig[tid] = fc.ljp_sig[tid];
        sm.ljp_ab[tid] = fc.ljp_ab[tid];
    }
    __syncthreads();

    // ---- COM gate of the tile's molecule pairs (energy.jl:248-254, ewalds.jl:334-340) ----
    const int w = wave_id();
    const int per_wave = MMC_TM * MMC_TM / MMC_WAVES; // 1024 pairs, 16 rows of 64
    int count = 0;
    for (int it = 0; it < per_wave / 64; it++) {
        const int p = w * per_wave + it * 64 + lane_id();
        const int ii = p >> 6, jj = p & 63; // ii is wave-uniform: its COM is an LDS broadcast
        int f = 0;
        if (ii < ni && jj < nj && (!diag || ii < jj)) {
            const double dx = vector1D(sm.ti[ii * MMC_REC + 9], sm.tj[jj * MMC_REC + 9], box);
            const double dy = vector1D(sm.ti[ii * MMC_REC + 10], sm.tj[jj * MMC_REC + 10], box);
            const double dz = vector1D(sm.ti[ii * MMC_REC + 11], sm.tj[jj * MMC_REC + 11], box);
            const double r2 = dx * dx + dy * dy + dz * dz;
            f = ((r2 < pp.lj_gate_sq) ? 1 : 0) | ((r2 < pp.qq_gate_sq) ? 2 : 0);
        }
        const unsigned long long m = __ballot(f != 0);
        if (f)
            sm.list[w * per_wave + count + lanes_below(m)] = (uint16_t)(p | (f << 12));
        count += __popcll(m);
    }
    if (lane_id() == 0)
        sm.wcnt[w] = count;
    __syncthreads();
    const int c0 = sm.wcnt[0], c1 = sm.wcnt[1], c2 = sm.wcnt[2], c3 = sm.wcnt[3];
    const int total = c0 + c1 + c2 + c3;
    auto entry = [&](int pos) {
        int slot;
        if (pos < c0) slot = pos;
        else if (pos < c0 + c1) slot = per_wave + (pos - c0);
        else if (pos < c0 + c1 + c2) slot = 2 * per_wave + (pos - c0 - c1);
        else slot = 3 * per_wave + (pos - c0 - c1 - c2);
        return (int)sm.list[slot];
    };

    double a_lj = 0.0, a_v = 0.0, a_q = 0.0;
    int n_ovl = 0;
    // ---- Coulomb pass: one lane per (molecule pair, a, b) (ewalds.jl:343-372), predicated like
    // the move kernel's (a skipped term is added as +0.0) with incremental (pair, a, b) indices ----
    const BoxConsts bc = box_consts(box);
    {
        int n = tid / 9, ab = tid - 9 * n; // g = 9 n + ab; g += 256 = 9 * 28 + 4
        for (int g = tid; g < total * 9; g += MMC_BLOCK) {
            const int a = (ab * 11) >> 5, b = ab - 3 * a;
            const int e = entry(n);
            const int ii = (e >> 6) & 63, jj = e & 63;
            const double *pa = &sm.ti[ii * MMC_REC + 3 * a], *pb = &sm.tj[jj * MMC_REC + 3 * b];
            const double rx = vector1D(pa[0], pb[0], bc);
            const double ry = vector1D(pa[1], pb[1], bc);
            const double rz = vector1D(pa[2], pb[2], bc);
            const double rab2 = rx * rx + ry * ry + rz * rz;
            const double qq = sm.qq9[ab];
            const bool gq = (e & (2 << 12)) != 0;
            const bool ov = gq && (rab2 < pp.ovr) && (qq < 0);
            const bool in = gq && !ov && (rab2 < pp.qq_slack_sq);
            double ev = qq_table_eval_clamped(sm.qtab, rab2);
            if (__any(in && rab2 < MMC_QQ_UMIN)) {
                if (rab2 < MMC_QQ_UMIN) ev = qq_pair(sm.qtab, rab2, pp.kappa);
            }
            a_q += in ? qq * ev : 0.0;
            n_ovl |= ov ? 1 : 0;
            n += 28;
            ab += 4;
            if (ab >= 9) { ab -= 9; n += 1; }
        }
    }
    // ---- LJ pass: atom pairs with eps > 0.001 (energy.jl:257-285) ----
    const int n_ljp = fc.n_ljp;
    for (int g = MMC_BLOCK - 1 - tid; g < total * n_ljp; g += MMC_BLOCK) { // last thread first:
        int n = g, p = 0;                                  // the waves the Coulomb tail leaves idle
        if (n_ljp != 1) {
            n = g / n_ljp;
            p = g - n * n_ljp;
        }
        const int e = entry(n);
        if (e & (1 << 12)) {
            const int ii = (e >> 6) & 63, jj = e & 63;
            const int ab = sm.ljp_ab[p];
            const int a = ab / 3, b = ab - 3 * a;
            const double *ri = &sm.ti[ii * MMC_REC], *rj = &sm.tj[jj * MMC_REC];
            const double rx = vector1D(ri[3 * a], rj[3 * b], box);
            const double ry = vector1D(ri[3 * a + 1], rj[3 * b + 1], box);
            const double rz = vector1D(ri[3 * a + 2], rj[3 * b + 2], box);
            const double rab2 = rx * rx + ry * ry + rz * rz;
            if (rab2 < pp.lj_slack_sq) {
                const double eps = sm.ljp_eps[p], sg = sm.ljp_sig[p];
                const double cx = vector1D(ri[9], rj[9], box);
                const double cy = vector1D(ri[10], rj[10], box);
                const double cz = vector1D(ri[11], rj[11], box);
                const double s2 = sg * sg / rab2;
                const double s6 = s2 * s2 * s2;
                const double s12 = s6 * s6;
                const double virab = eps * (2.0 * s12 - s6);
                const double f0 = rx * virab * s2, f1 = ry * virab * s2, f2 = rz * virab * s2;
                a_lj += eps * (s12 - s6);
                a_v += cx * f0 + cy * f1 + cz * f2;
            }
        }
    }
    __syncthreads(); // tiles are read no more: reuse them as reduction scratch
    double v[3] = { a_lj, a_v, a_q };
    block_sum_wide<3>(v, sm.ti, sm.red, n_ovl, sm.wcnt);
    if (tid == 0) {
        TotalPart o;
        o.lj_pot = sm.red[0]; o.lj_vir = sm.red[1]; o.qq = sm.red[2];
        o.n_ovl = sm.wcnt[0] | sm.wcnt[1] | sm.wcnt[2] | sm.wcnt[3];
        o._pad = 0;
        out[(int64_t)r * n_pairs + tp] = o;
    }
}

// per replica: sum the tile-pair partials in index order.  out: TotalsRaw in the reference's
// normalisation (sum_i 4 pot_i etc., which the host halves): pairs counted once -> x2.
__global__ void k_total_sum(const TotalPart *parts, int n_pairs, int n_rep, TotalsRaw *out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rep)
        return;
    double lj = 0.0, vir = 0.0, qq = 0.0;
    int ov = 0;
    const TotalPart *p = parts + (int64_t)r * n_pairs;
    for (int k = 0; k < n_pairs; k++) {
        lj += p[k].lj_pot; vir += p[k].lj_vir; qq += p[k].qq; ov |= p[k].n_ovl;
    }
    TotalsRaw t;
    t.lj_e = 2.0 * (lj * 4);          // energy.jl:289, both directions
    t.lj_v = 2.0 * (vir * 24 / 3.0);
    t.qq = 2.0 * qq;
    t.n_ovl = ov; t._pad = 0;
    out[r] = t;
}

// Volume move, coordinate part (volumeChange.jl:62-80, the reference's NPT specification): centres
// of mass scale by f = L_new / L_old, atoms translate rigidly with their molecule.
// grid (ceil(n_mol/256), R).
__global__ void k_rescale(BatchView bv, double *rec, double f)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (j >= bv.n_mol)
        return;
    const int64_t m = r * bv.mol_stride + j;
    double *c[3] = { bv.comx + m, bv.comy + m, bv.comz + m };
    double d[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double old = *c[k], nw = old * f;
        d[k] = nw - old;
        *c[k] = nw;
    }
    const int fa = bv.first0[j], na = bv.cnt[j];
    for (int a = 0; a < na; a++) {
        const int64_t o = r * bv.atom_stride + fa + a;
        bv.ax[o] += d[0]; bv.ay[o] += d[1]; bv.az[o] += d[2];
    }
    if (rec) {
        double *o = rec + ((int64_t)r * bv.n_mol + j) * MMC_RSTRIDE;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const int64_t s = r * bv.atom_stride + fa + a;
            o[3 * a] = bv.ax[s]; o[3 * a + 1] = bv.ay[s]; o[3 * a + 2] = bv.az[s];
        }
        o[9] = *c[0]; o[10] = *c[1]; o[11] = *c[2];
        comq_store(bv, r, j, 0, o[9]); comq_store(bv, r, j, 1, o[10]); comq_store(bv, r, j, 2, o[11]);
    }
}

// cos/sin of 2 pi c / L for the three coordinates of every atom (ewalds.jl:564-569), once.
// ph[r][l][0..5] = cos x, sin x, cos y, sin y, cos z, sin z
__global__ void k_atom_phases(BatchView bv, double *ph)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (l >= bv.n_atoms)
        return;
    const double L = bv.box;
    const int64_t a = r * bv.atom_stride + l;
    double *o = ph + ((int64_t)r * bv.n_atoms + l) * 6;
    double sn, cs;
    sincos_moderate(MMC_TWOPI * bv.ax[a] / L, sn, cs); o[0] = cs; o[1] = sn;
    sincos_moderate(MMC_TWOPI * bv.ay[a] / L, sn, cs); o[2] = cs; o[3] = sn;
    sincos_moderate(MMC_TWOPI * bv.az[a] / L, sn, cs); o[4] = cs; o[5] = sn;
}

// RecipLong for large batches with the whole replica in LDS: one workgroup of RL_WAVES waves per
// replica builds the six phase factors of every atom (k_atom_phases' arithmetic) and its charge in
// LDS -- 56 bytes per atom, 126 KB at 2250 atoms, which MI355X's 160 KB per compute unit holds --
// and every wave then walks all atoms once per (kx, ky) column it takes from a shared queue
// (columns in descending order of work, so the waves finish together), exactly as a 64-thread
// workgroup of k_recip_long_ph<64> does: same per-lane sums, same wave reduction, same S(k) bit
// for bit.  The kz of a column are the symmetric range 5 - M .. 5 + M (k^2 < k_sq_max), so the
// inner loop is compiled for each M instead of testing every kz of every atom; and the phase
// array never exists in memory.
#define RL_WAVES 16
struct RecipOrder {
    uint8_t col[6 * MMC_NKTAB]; // column indices kx * 11 + ky + 5, most work first
    int32_t n;                  // columns that have any k-vector
};

template <int M>
__device__ __forceinline__ void recip_column_lds(const int16_t *col, double *s0, double *s1, const double *ph,
                                             const double *qv, int n_atoms, int kx, int ky, int lane)
{   // (plain pointers, not the BatchView: a struct passed by reference to a function that is not
    // inlined is copied to scratch by every thread)
    const int aky = ky < 0 ? -ky : ky;
    double acc[2 * (2 * M + 1)];
#pragma unroll
    for (int k = 0; k < 2 * (2 * M + 1); k++)
        acc[k] = 0.0;
    for (int l = lane; l < n_atoms; l += 64) {
        const double q = qv[l];
        const double2 px = *reinterpret_cast<const double2 *>(ph + 6 * l),
                      py = *reinterpret_cast<const double2 *>(ph + 6 * l + 2),
                      pz = *reinterpret_cast<const double2 *>(ph + 6 * l + 4);
        const cplx x1 = { px.x, px.y }, y1 = { py.x, py.y }, z1 = { pz.x, pz.y };
        cplx ex = { 1.0, 0.0 }, ey = { 1.0, 0.0 };
        if (kx > 0) {
            ex = x1;
            for (int k = 2; k <= kx; k++)
                ex = c_mul(ex, x1);
        }
        if (aky > 0) {
            ey = y1;
            for (int k = 2; k <= aky; k++)
                ey = c_mul(ey, y1);
            if (ky < 0)
                ey = c_conj(ey);
        }
        // kz = -M .. M at index kz + M: the powers of e^{i 2 pi z / L} by the reference's recurrence
        // (ewalds.jl:575-585), each used as it is made -- the products are the ones an ez[] table
        // would give, without holding the table in registers
        const cplx one = { 1.0, 0.0 };
        const cplx qxy = c_mul(c_rmul(q, ex), ey);
        {
            const cplx t = c_mul(qxy, one);
            acc[2 * M] += t.re;
            acc[2 * M + 1] += t.im;
        }
        cplx p = one;
#pragma unroll
        for (int k = 1; k <= M; k++) {
            p = k == 1 ? z1 : c_mul(p, z1);
            const cplx tp = c_mul(qxy, p), tm = c_mul(qxy, c_conj(p));
            acc[2 * (M + k)] += tp.re;
            acc[2 * (M + k) + 1] += tp.im;
            acc[2 * (M - k)] += tm.re;
            acc[2 * (M - k) + 1] += tm.im;
        }
    }
    double tot[2 * (2 * M + 1)];
#pragma unroll
    for (int k = 0; k < 2 * (2 * M + 1); k++)
        tot[k] = wave_sum(acc[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 2 * M + 1; k++) {
            const int idx = col[5 - M + k];
            if (idx >= 0) {
                s0[2 * idx] = tot[2 * k]; s0[2 * idx + 1] = tot[2 * k + 1];
                s1[2 * idx] = tot[2 * k]; s1[2 * idx + 1] = tot[2 * k + 1];
            }
        }
    }
}

__global__ __launch_bounds__(RL_WAVES * 64) void k_recip_long_lds(BatchView bv, RecipOrder order)
{
    extern __shared__ __align__(16) double rl_lds[];
    __shared__ int next_col;
    const int n_atoms = bv.n_atoms;
    double *ph = rl_lds;               // [n_atoms][6] cos x, sin x, cos y, sin y, cos z, sin z
    double *qv = rl_lds + 6 * n_atoms; // [n_atoms]
    const int r = blockIdx.x;
    const double L = bv.box;
    if (threadIdx.x == 0)
        next_col = 0;
    for (int l = threadIdx.x; l < n_atoms; l += RL_WAVES * 64) {
        const int64_t a = r * bv.atom_stride + l;
        double sn, cs;
        sincos_moderate(MMC_TWOPI * bv.ax[a] / L, sn, cs); ph[6 * l] = cs; ph[6 * l + 1] = sn;
        sincos_moderate(MMC_TWOPI * bv.ay[a] / L, sn, cs); ph[6 * l + 2] = cs; ph[6 * l + 3] = sn;
        sincos_moderate(MMC_TWOPI * bv.az[a] / L, sn, cs); ph[6 * l + 4] = cs; ph[6 * l + 5] = sn;
        qv[l] = bv.charge[l];
    }
    __syncthreads();
    const int lane0 = threadIdx.x & 63;
    for (;;) {
        int lane = lane0;
        asm volatile("" : "+v"(lane)); // keep lane-derived addresses of the inlined column bodies out of LICM
        int slot = 0;
        if (lane == 0)
            slot = atomicAdd(&next_col, 1);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= order.n)
            break;
        const int c = order.col[slot];
        const int kx = c / MMC_NKTAB, ky = c % MMC_NKTAB - 5;
        // the kz of this column: 5 - m .. 5 + m, and within it exactly those with an index
        const int16_t *col = bv.kmap + c * MMC_NKTAB;
        int m = 0;
        for (int k = 0; k <= 5; k++)
            if (col[5 + k] >= 0 || col[5 - k] >= 0)
                m = k;
        double *s0 = s_buf(bv, r, 0), *s1 = s_buf(bv, r, 1);
        switch (m) {
        case 0: // only kz = 0
        case 1: recip_column_lds<1>(col, s0, s1, ph, qv, n_atoms, kx, ky, lane); break;
        case 2: recip_column_lds<2>(col, s0, s1, ph, qv, n_atoms, kx, ky, lane); break;
        case 3: recip_column_lds<3>(col, s0, s1, ph, qv, n_atoms, kx, ky, lane); break;
        case 4: recip_column_lds<4>(col, s0, s1, ph, qv, n_atoms, kx, ky, lane); break;
        default: recip_column_lds<5>(col, s0, s1, ph, qv, n_atoms, kx, ky, lane); break;
        }
    }
}

// k_recip_long with the phases read instead of recomputed per (kx, ky) column.
// grid (66 columns, R, n_chunks): with n_chunks > 1 (few replicas: the 66 workgroups of one
// system would leave most of the chip idle and walk all atoms serially) chunk c sums atoms
// [c * chunk_len, (c + 1) * chunk_len) into spart[r][c][k] and k_recip_finish adds the chunks in
// index order; with one chunk the sums go straight to S.
// BLOCK = 256 (few replicas: short critical path) or 64 (many replicas: one wave walks all atoms
// of its column, so the 22-value reduction -- a third of the 256-thread form's instructions --
// is paid once per column instead of four times, and needs no LDS pass).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_recip_long_ph(BatchView bv, const double *ph,
                                                         double *spart, int chunk_len)
{
    __shared__ double red[2 * MMC_NKTAB * MMC_WAVES];
    const int r = blockIdx.y;
    const int kx = blockIdx.x / MMC_NKTAB, ky = blockIdx.x % MMC_NKTAB - 5;
    const int aky = ky < 0 ? -ky : ky;
    // which kz of this (kx, ky) column survive 0 < k^2 < k_sq_max (ewalds.jl:57-66): 16 of the 66
    // columns have none, the others 3..11 -- skip what is not there (workgroup-uniform bounds)
    const int16_t *col = bv.kmap + (kx * MMC_NKTAB + (ky + 5)) * MMC_NKTAB;
    int k_lo = MMC_NKTAB, k_hi = -1;
    for (int k = 0; k < MMC_NKTAB; k++)
        if (col[k] >= 0) {
            k_lo = min(k_lo, k);
            k_hi = max(k_hi, k);
        }
    if (k_hi < 0)
        return;
    const double *myph = ph + (int64_t)r * bv.n_atoms * 6;
    double acc[2 * MMC_NKTAB];
#pragma unroll
    for (int k = 0; k < 2 * MMC_NKTAB; k++)
        acc[k] = 0.0;
    const int l_begin = blockIdx.z * chunk_len, l_end = min(l_begin + chunk_len, (int)bv.n_atoms);
    for (int l = l_begin + threadIdx.x; l < l_end; l += BLOCK) {
        const double q = bv.charge[l];
        const double2 px = *reinterpret_cast<const double2 *>(myph + 6 * l),
                      py = *reinterpret_cast<const double2 *>(myph + 6 * l + 2),
                      pz = *reinterpret_cast<const double2 *>(myph + 6 * l + 4);
        const cplx x1 = { px.x, px.y }, y1 = { py.x, py.y }, z1 = { pz.x, pz.y };
        cplx ex = { 1.0, 0.0 }, ey = { 1.0, 0.0 };
        if (kx > 0) {
            ex = x1;
            for (int k = 2; k <= kx; k++)
                ex = c_mul(ex, x1);
        }
        if (aky > 0) {
            ey = y1;
            for (int k = 2; k <= aky; k++)
                ey = c_mul(ey, y1);
            if (ky < 0)
                ey = c_conj(ey);
        }
        cplx ez[MMC_NKTAB];
        const cplx one = { 1.0, 0.0 };
        ez[5] = one; ez[6] = z1; ez[4] = c_conj(z1);
        cplx p = z1;
#pragma unroll
        for (int k = 2; k <= 5; k++) {
            p = c_mul(p, z1);
            ez[5 + k] = p;
            ez[5 - k] = c_conj(p);
        }
        const cplx qxy = c_mul(c_rmul(q, ex), ey);
#pragma unroll
        for (int k = 0; k < MMC_NKTAB; k++) {
            if (k >= k_lo && k <= k_hi) {
                const cplx t = c_mul(qxy, ez[k]);
                acc[2 * k] += t.re;
                acc[2 * k + 1] += t.im;
            }
        }
    }
    double tot[2 * MMC_NKTAB];
    if (BLOCK == 64) {
#pragma unroll
        for (int k = 0; k < 2 * MMC_NKTAB; k++)
            tot[k] = (k >= 2 * k_lo && k <= 2 * k_hi + 1) ? wave_sum(acc[k]) : 0.0;
    } else {
        block_sum<2 * MMC_NKTAB>(acc, red, tot);
    }
    if (threadIdx.x == 0) {
        double *s0 = s_buf(bv, r, 0), *s1 = s_buf(bv, r, 1);
        if (gridDim.z > 1) // a partial sum: both targets are this chunk's row of the scratch
            s0 = s1 = spart + ((int64_t)r * gridDim.z + blockIdx.z) * bv.nk_stride * 2;
#pragma unroll
        for (int k = 0; k < MMC_NKTAB; k++) {
            const int idx = col[k];
            if (idx >= 0) {
                s0[2 * idx] = tot[2 * k]; s0[2 * idx + 1] = tot[2 * k + 1];
                s1[2 * idx] = tot[2 * k]; s1[2 * idx + 1] = tot[2 * k + 1];
            }
        }
    }
}

// S(k) = sum of the chunk partials in chunk order (both arrays get `term`, ewalds.jl:600-601).
__global__ void k_recip_finish(BatchView bv, const double *spart, int n_chunks)
{
    const int r = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 2 * bv.nkvecs)
        return;
    double s = 0.0;
    for (int c = 0; c < n_chunks; c++)
        s += spart[((int64_t)r * n_chunks + c) * bv.nk_stride * 2 + k];
    s_buf(bv, r, 0)[k] = s;
    s_buf(bv, r, 1)[k] = s;
}
