// mmc_kernels.hpp -- the HIP kernels of libmmc_hip.so (gfx950).
//
//   k_mol_energy     K1/K2  one chosen molecule vs all others: LJ_poly_dU + EwaldReal/CoulombReal
//   k_totals         K2     per-replica sums of k_mol_energy's per-molecule results
//   k_recip_long     K3     S(k) = sum_l q_l e^{ik.r_l}: one workgroup per (kx, ky) column
//   k_recip_energy   K3     sum_k cfac_k |S(k)|^2
//   k_recip_move     K4     dS(k) for the moved atoms of ONE system (context API)
//   k_move_eval      K1+K4  one trial move per replica, fused: 2x LJ_poly_dU + 2x EwaldShort +
//                           RecipMove, plus the commit of the replica's previous accepted move
//   k_set_molecule / k_settle / k_charge_sums / k_copy_s   small state updates
#pragma once
#include "mmc_device.hpp"

// Device state of R replicas (R = 1 for a context).  Coordinates are SoA per replica.
struct BatchView {
    double *comx, *comy, *comz; // [R][mol_stride]
    double *ax, *ay, *az;       // [R][atom_stride]
    int64_t mol_stride, atom_stride;
    const int32_t *first0, *cnt, *atype; // shared topology
    const double *charge, *eps, *sig;
    int32_t n_mol, n_atoms, n_types;
    double box;
    // Ewald
    double *S;           // [R][2][nk_stride] complex (re, im): two structure-factor buffers
    int64_t nk_stride;   // in complex elements
    const int32_t *kxyz; // [nkvecs][3]
    const double *cfac;  // [nkvecs]
    const int16_t *kmap; // [(nk+1)*(2nk+1)*(2nk+1)] -> k index or -1
    int32_t nkvecs, nk;
    double kappa, factor;
    // 16-bit fixed-point copy of the centres of mass (homogeneous systems only, else NULL):
    // q = floor(frac(x / L) * 2^16) per axis, the prefilter stream of the COM scans of
    // k_move_eval_wave / k_total_wave (6 bytes per molecule instead of 24).  Per replica
    // 3 * cq_stride codes: (x, y) pairs of molecule j at [2 j], [2 j + 1] -- one 32-bit load, one
    // packed 16-bit subtract -- then the z codes from [2 * cq_stride].  A 16-bit difference wraps
    // exactly as the minimum image does; see com_quant() for the bound that makes the prefilter a
    // superset of the reference's gate.  Every writer of a centre of mass keeps it in step through
    // comq_store().
    uint16_t *comq;
    int64_t cq_stride;
    // orientation quaternions [R][n_mol][4] (totProps.quat, Ewald/main.jl:527,535,619) or NULL:
    // only kept when the caller asked for the reference's quaternion move generation
    // (mmc_batch_set_orientations); committed together with the coordinates of an accepted move
    double *quat;
};

// Box-fraction fixed point of one coordinate: floor(frac(x / L) * 65536) mod 65536.  For two
// coordinates a, b with minimum-image difference d (|d| <= L/2) the wrapped signed 16-bit
// difference D of their codes satisfies |D| <= |d| / u + 1, u = L / 65536 (a difference of two
// floors is off by less than 1; the fp64 rounding of x / L adds < 1e-10).  Hence
// sqrt(Dx^2 + Dy^2 + Dz^2) <= r / u + sqrt(3): com_quant_gate() is a threshold no pair inside the
// gate can exceed.  (The reference's vector1D wraps once; its distance is never below the true
// minimum image, so a pair it accepts is accepted here too.)
__host__ __device__ inline uint16_t com_quant(double x, double inv_box)
{
    double y = x * inv_box;
    y -= floor(y);
    return (uint16_t)((uint32_t)(int64_t)(y * 65536.0) & 0xffffu);
}

// the prefilter threshold on Dx^2 + Dy^2 + Dz^2 (unsigned 32-bit) for a gate of gate_sq A^2
__host__ __device__ inline uint32_t com_quant_gate(double gate_sq, double box)
{
    const double t = sqrt(gate_sq) * 65536.0 / box + 2.6; // sqrt(3) and half a unit of slack per axis
    const double t2 = t * t;
    return t2 >= 4294967295.0 ? 0xffffffffu : (uint32_t)t2 + 1u; // (a box below ~2.3 gates: all pass)
}

// One prefilter test: codes (x | y << 16, z) of a molecule against those of a chosen centre of
// mass; returns Dx^2 + Dy^2 + Dz^2 (5 instructions: v_pk_sub_i16, v_sub, v_mul_i32_i24 on the
// sign-extended half, v_dot2c_i32_i16).
typedef short mmc_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short mmc_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t com_quant_dist2(uint32_t xy, uint32_t z, uint32_t cxy, uint32_t cz)
{
    const mmc_u16x2 d = __builtin_bit_cast(mmc_u16x2, xy) - __builtin_bit_cast(mmc_u16x2, cxy);
    const int dz = (int16_t)(uint16_t)(z - cz);
    const mmc_s16x2 ds = __builtin_bit_cast(mmc_s16x2, d);
    return (uint32_t)__builtin_amdgcn_sdot2(ds, ds, dz * dz, false);
}

__device__ __forceinline__ void comq_store(const BatchView &b, int r, int j, int d, double v, double inv_box)
{
    if (b.comq)
        b.comq[(int64_t)r * 3 * b.cq_stride + (d < 2 ? 2 * j + d : 2 * b.cq_stride + j)] =
            com_quant(v, inv_box);
}
__device__ __forceinline__ void comq_store(const BatchView &b, int r, int j, int d, double v)
{
    comq_store(b, r, j, d, v, 1.0 / b.box);
}

__device__ __forceinline__ SysView sys_view(const BatchView &b, int r)
{
    SysView s;
    s.comx = b.comx + r * b.mol_stride;
    s.comy = b.comy + r * b.mol_stride;
    s.comz = b.comz + r * b.mol_stride;
    s.ax = b.ax + r * b.atom_stride;
    s.ay = b.ay + r * b.atom_stride;
    s.az = b.az + r * b.atom_stride;
    s.first0 = b.first0; s.cnt = b.cnt; s.atype = b.atype;
    s.charge = b.charge; s.eps = b.eps; s.sig = b.sig;
    s.n_mol = b.n_mol; s.n_atoms = b.n_atoms; s.n_types = b.n_types;
    s.box = b.box;
    return s;
}

__device__ __forceinline__ double *s_buf(const BatchView &b, int r, int which)
{
    return b.S + ((int64_t)r * 2 + which) * b.nk_stride * 2;
}

struct MolE {
    double lj_pot, lj_vir, qq_pot;
    int32_t ovl, _pad;
};

// ---- K1/K2: chosen molecule i = i_base + blockIdx.x of replica blockIdx.y vs everyone ---------
template <bool LJ, bool QQ, int STYLE>
__global__ __launch_bounds__(MMC_BLOCK) void k_mol_energy(BatchView bv, int i_base, PairParams pp,
                                                          MolE *out, int out_stride)
{
    __shared__ Chosen ch;
    __shared__ int32_t list[MMC_LIST_CAP];
    __shared__ int32_t wcnt[MMC_WAVES];
    __shared__ double red[3 * MMC_WAVES];
    const int r = blockIdx.y, i0 = i_base + blockIdx.x;
    const SysView s = sys_view(bv, r);
    const int na = min(s.cnt[i0], MMC_MAX_ATOMS);
    if (threadIdx.x < na) {
        const int a = s.first0[i0] + threadIdx.x;
        ch.at[0][threadIdx.x][0] = s.ax[a];
        ch.at[0][threadIdx.x][1] = s.ay[a];
        ch.at[0][threadIdx.x][2] = s.az[a];
        ch.q[threadIdx.x] = s.charge[a];
        ch.type[threadIdx.x] = s.atype[a];
    }
    if (threadIdx.x == 0) {
        ch.i0 = i0; ch.na = na;
        ch.com[0][0] = s.comx[i0]; ch.com[0][1] = s.comy[i0]; ch.com[0][2] = s.comz[i0];
    }
    __syncthreads();
    PairAcc acc[1] = { { 0.0, 0.0, 0.0, 0 } };
    pair_scan<1, LJ, QQ, STYLE, false>(s, &ch, nullptr, 0, s.n_mol, pp, list, wcnt, acc);
    double v[3] = { acc[0].lj_pot, acc[0].lj_vir, acc[0].qq_pot }, tot[3];
    const int ovl = __syncthreads_or(acc[0].ovl);
    block_sum<3>(v, red, tot);
    if (threadIdx.x == 0) {
        MolE m;
        m.lj_pot = tot[0]; m.lj_vir = tot[1];
        m.qq_pot = ovl ? 0.0 : tot[2]; // ewalds.jl:359-360: `return 0.0, true`
        m.ovl = ovl; m._pad = 0;
        out[(int64_t)r * out_stride + blockIdx.x] = m;
    }
}

struct TotalsRaw {
    double lj_e, lj_v, qq; // sum_i 4*pot_i ; sum_i 24*vir_i/3 ; sum_i EwaldReal_i
    int32_t n_ovl, _pad;
};

// energy.jl:972-977 and :991-1000: the per-molecule loop sums (halving is done by the host).
__global__ __launch_bounds__(MMC_BLOCK) void k_totals(const MolE *per_mol, int n_mol, int stride,
                                                      TotalsRaw *out)
{
    __shared__ double red[4 * MMC_WAVES];
    const int r = blockIdx.x;
    double v[4] = { 0, 0, 0, 0 }, tot[4];
    for (int i = threadIdx.x; i < n_mol; i += MMC_BLOCK) {
        const MolE m = per_mol[(int64_t)r * stride + i];
        v[0] += m.lj_pot * 4;        // energy.jl:289
        v[1] += m.lj_vir * 24 / 3.0;
        v[2] += m.qq_pot;
        v[3] += (double)m.ovl;
    }
    block_sum<4>(v, red, tot);
    if (threadIdx.x == 0) {
        TotalsRaw t;
        t.lj_e = tot[0]; t.lj_v = tot[1]; t.qq = tot[2];
        t.n_ovl = (int)tot[3]; t._pad = 0;
        out[r] = t;
    }
}

// ---- K3: full structure factor -----------------------------------------------------------------
// grid (nk+1)*(2nk+1) x R.  The workgroup owns one (kx, ky) and accumulates all 2nk+1 kz at once:
// per atom 3 sincos, the reference's power recurrence, and 11 complex multiply-adds that share
// q*e^{ikx x}*e^{iky y} (ewalds.jl:589-597).  Atoms are read unit-stride.  nk == 5 only.
__global__ __launch_bounds__(MMC_BLOCK) void k_recip_long(BatchView bv)
{
    __shared__ double red[2 * MMC_NKTAB * MMC_WAVES];
    const int r = blockIdx.y;
    const int kx = blockIdx.x / MMC_NKTAB, ky = blockIdx.x % MMC_NKTAB - 5;
    const SysView s = sys_view(bv, r);
    const double L = s.box;
    double acc[2 * MMC_NKTAB];
#pragma unroll
    for (int k = 0; k < 2 * MMC_NKTAB; k++)
        acc[k] = 0.0;
    const int aky = ky < 0 ? -ky : ky;
    // a column without any kz inside 0 < k^2 < k_sq_max (16 of the 66) has nothing to sum
    const int16_t *col = bv.kmap + (kx * MMC_NKTAB + (ky + 5)) * MMC_NKTAB;
    bool any = false;
    for (int k = 0; k < MMC_NKTAB; k++)
        any = any || col[k] >= 0;
    if (!any)
        return;
    for (int l = threadIdx.x; l < s.n_atoms; l += MMC_BLOCK) {
        const double q = s.charge[l];
        double sn, cs;
        sincos(MMC_TWOPI * s.ax[l] / L, &sn, &cs);
        const cplx x1 = { cs, sn };
        sincos(MMC_TWOPI * s.ay[l] / L, &sn, &cs);
        const cplx y1 = { cs, sn };
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
        phase_row(s.az[l], L, ez);
        const cplx qxy = c_mul(c_rmul(q, ex), ey); // (q*eikx)*eiky
#pragma unroll
        for (int k = 0; k < MMC_NKTAB; k++) {
            const cplx t = c_mul(qxy, ez[k]);
            acc[2 * k] += t.re;
            acc[2 * k + 1] += t.im;
        }
    }
    double tot[2 * MMC_NKTAB];
    block_sum<2 * MMC_NKTAB>(acc, red, tot);
    if (threadIdx.x == 0) {
        double *s0 = s_buf(bv, r, 0), *s1 = s_buf(bv, r, 1);
#pragma unroll
        for (int k = 0; k < MMC_NKTAB; k++) {
            const int idx = bv.kmap[(kx * MMC_NKTAB + (ky + 5)) * MMC_NKTAB + k];
            if (idx >= 0) { // ewalds.jl:600-601: both arrays get `term`
                s0[2 * idx] = tot[2 * k]; s0[2 * idx + 1] = tot[2 * k + 1];
                s1[2 * idx] = tot[2 * k]; s1[2 * idx + 1] = tot[2 * k + 1];
            }
        }
    }
}

// ewalds.jl:599: energy += cfac[i] * real(conj(term) * term), no factor.
__global__ __launch_bounds__(MMC_BLOCK) void k_recip_energy(BatchView bv, int which, double *out)
{
    __shared__ double red[MMC_WAVES];
    const int r = blockIdx.x;
    const double *S = s_buf(bv, r, which);
    double v[1] = { 0.0 }, tot[1];
    for (int k = threadIdx.x; k < bv.nkvecs; k += MMC_BLOCK) {
        const double re = S[2 * k], im = S[2 * k + 1];
        v[0] += bv.cfac[k] * (re * re - (-im) * im);
    }
    block_sum<1>(v, red, tot);
    if (threadIdx.x == 0)
        out[r] = tot[0];
}

// ---- K4 (context API): RecipMove for one system, explicit old/new arrays -----------------------
struct RecipMoveArgs {
    double r_old[9], r_new[9], q[3];
};

// ewalds.jl:718-826.  S(a_old) = sumQExpOld, S(a_new) = sumQExpNew; S(a_dst) = S(a_new) + dS
// (a_dst == a_new: in place, :805-814), energy against S(a_old) (:817-821).
__global__ __launch_bounds__(MMC_BLOCK) void k_recip_move(BatchView bv, RecipMoveArgs a,
                                                          double *out, int a_old, int a_new, int a_dst)
{
    __shared__ cplx tab[2][3][3][MMC_NKTAB];
    __shared__ double red[MMC_WAVES];
    if (threadIdx.x < 18) {
        const int st = threadIdx.x / 9, l = (threadIdx.x % 9) / 3, d = threadIdx.x % 3;
        const double x = st ? a.r_new[3 * l + d] : a.r_old[3 * l + d];
        phase_row(x, bv.box, tab[st][l][d]);
    }
    __syncthreads();
    const double *So = s_buf(bv, 0, a_old), *Si = s_buf(bv, 0, a_new);
    double *Sn = s_buf(bv, 0, a_dst);
    double v[1] = { 0.0 }, tot[1];
    for (int k = threadIdx.x; k < bv.nkvecs; k += MMC_BLOCK) {
        const int kx = bv.kxyz[3 * k], ky = bv.kxyz[3 * k + 1], kz = bv.kxyz[3 * k + 2];
        double nr = Si[2 * k], ni = Si[2 * k + 1];
#pragma unroll
        for (int l = 0; l < 3; l++) {
            const cplx tn = c_mul(c_mul(tab[1][l][0][5 + kx], tab[1][l][1][5 + ky]),
                                  tab[1][l][2][5 + kz]);
            const cplx to = c_mul(c_mul(tab[0][l][0][5 + kx], tab[0][l][1][5 + ky]),
                                  tab[0][l][2][5 + kz]);
            nr += a.q[l] * (tn.re - to.re);
            ni += a.q[l] * (tn.im - to.im);
        }
        Sn[2 * k] = nr; Sn[2 * k + 1] = ni;
        const double orr = So[2 * k], oi = So[2 * k + 1];
        v[0] += bv.cfac[k] * ((nr * nr - (-ni) * ni) - (orr * orr - (-oi) * oi));
    }
    block_sum<1>(v, red, tot);
    if (threadIdx.x == 0)
        out[0] = tot[0];
}

// ---- K1+K4 fused, batched: one trial move per replica ------------------------------------------
// The device form of mmc_move (include/mmc_hip.h).  `flags`: bit 0 = commit this replica's
// previous proposal, bit 1 = which S buffer is current.  com_old/atoms_old: the chosen molecule's
// current state as the host mirror holds it (the values the device holds too) -- k_move_eval_fast
// takes the old state from here so that nothing in it depends on a second memory round trip;
// k_move_eval ignores them and reads the device arrays.
struct MoveRec {
    int32_t mol, flags;
    double com_new[3];
    double atoms_new[9];
    double com_old[3];
    double atoms_old[9];
    double q_new[4]; // proposed orientation `ei` (main.jl:528,532) when quaternions are kept
                     // (k_propose), else zeros: a record with |q_new| = 0 commits no quaternion
};

// commit `totProps.quat[i] = ei` (main.jl:619): lane t in [0, 4) writes component t
__device__ __forceinline__ void quat_commit(const BatchView &b, int r, int mol0, int t, double v,
                                            bool valid)
{
    if (b.quat && valid)
        b.quat[((int64_t)r * b.n_mol + mol0) * 4 + t] = v;
}
__device__ __forceinline__ bool quat_valid(const double *q)
{
    return q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] > 0.25;
}

struct PartOut {
    double lj_pot[2], lj_vir[2], qq_pot[2]; // [old, new] raw sums of this workgroup's j-range
    double recip;                           // sum_k cfac (|S_new|^2 - |S_old|^2), no factor
    int32_t ovl[2];                         // bit 0: overlap of the old / new state; above it
                                            // ovl[0] carries the record's checksum and ovl[1] the
                                            // launch stamp (store_part, part_checksum)
};
static_assert(sizeof(PartOut) == 64, "PartOut is one 64-byte line");

// A PartOut leaves the wave as one store instruction of 4 lanes x 16 B into pinned host memory.
// The driver stamps every launch with its number and polls ovl[1]; nothing (neither HIP nor PCIe)
// promises that the four pieces land together or in order, so the record also carries a 31-bit
// checksum of its seven sums and the stamp in ovl[0]: the host consumes a record only when the
// stamp is the expected one AND the checksum of the bytes it copied out matches -- a torn record
// (new stamp, some sums still from the previous launch) fails the checksum and is polled again.
// No fence, no event, no stream synchronisation, and the host starts on the first results while
// the launch still runs.  `w8`: the record as 8 words in LDS, complete before the call.
#define MMC_STAMP_MASK 0x3fffffffu
// WRITE_THROUGH = false: plain stores, visible to the host when the L2 writes them back -- at the
// latest when the kernel ends (a launch-per-step driver needs no more; a write-through store would
// keep the wave's next s_waitcnt vmcnt waiting for a PCIe round trip).  true: system-scope stores
// (sc0 sc1), for a persistent kernel whose results the host waits for while it keeps running: a
// plain store can sit in the L2 for as long as the kernel lives.
template <bool WRITE_THROUGH = false>
__device__ inline void store_part(PartOut *dst, const double *w8, int tid)
{
    if (tid < 4) {
        double2 *p = reinterpret_cast<double2 *>(dst) + tid;
        if (WRITE_THROUGH) {
            typedef double part_piece __attribute__((ext_vector_type(2)));
            part_piece v;
            v.x = w8[2 * tid];
            v.y = w8[2 * tid + 1];
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
        } else {
            *p = make_double2(w8[2 * tid], w8[2 * tid + 1]);
        }
    }
}

// 31-bit checksum of a PartOut's seven sums and its launch stamp: every word is hashed with its
// position and the stamp on its own, the seven hashes are ADDED and the sum is folded -- so seven
// lanes can hash in parallel (part_checksum_lanes) where one lane walking a serial chain over the
// 14 dwords cost ~110 dependent instructions per record.  A record whose pieces come from two
// launches has at least one word whose hash belongs to another (value, stamp): the sum differs
// except with probability 2^-31.  The same function on the device (writer) and the host (reader).
__host__ __device__ inline uint32_t part_word_hash(uint32_t lo, uint32_t hi, int k, unsigned stamp)
{
    uint32_t h = (lo ^ (0x9E3779B1u * (uint32_t)(k + 1))) * 0x85EBCA6Bu;
    h ^= h >> 13;
    h = (h ^ hi ^ ((stamp & MMC_STAMP_MASK) * 0xC2B2AE35u)) * 0x27D4EB2Fu;
    h ^= h >> 15;
    return h;
}
__host__ __device__ inline uint32_t part_checksum_fold(uint32_t c)
{
    c ^= c >> 16;
    return c & 0x7fffffffu;
}
__host__ __device__ inline uint32_t part_checksum(const double *w7, unsigned stamp)
{
    uint32_t c = 0;
    for (int k = 0; k < 7; k++) {
        unsigned long long b;
        __builtin_memcpy(&b, &w7[k], 8);
        c += part_word_hash((uint32_t)b, (uint32_t)(b >> 32), k, stamp);
    }
    return part_checksum_fold(c);
}
// The same by the lanes of a wave: lane k < 7 holds word k in `word`; every lane gets the checksum.
__device__ __forceinline__ uint32_t part_checksum_lanes(double word, int lane, unsigned stamp)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(word);
    int h = lane < 7 ? (int)part_word_hash((uint32_t)b, (uint32_t)(b >> 32), lane, stamp) : 0;
    h += dpp_row_shr<1>(h);
    h += dpp_row_shr<2>(h);
    h += dpp_row_shr<4>(h);
    return part_checksum_fold((uint32_t)__builtin_amdgcn_readlane(h, 7));
}

// word 7 of the record: ovl[0] = checksum << 1 | overlap_old, ovl[1] = stamp << 1 | overlap_new
// (bit 31 of ovl[1]: the Metropolis decision, where the kernel made it -- DecideConsts below)
__device__ inline double pack_ovl(int o0, int o1, unsigned stamp, uint32_t csum, int accept = 0)
{
    const unsigned lo = (csum << 1) | (unsigned)(o0 & 1),
                   hi = ((unsigned)(accept & 1) << 31) | ((stamp & MMC_STAMP_MASK) << 1) | (unsigned)(o1 & 1);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__host__ __device__ inline uint32_t part_stamp_of(int32_t ovl1) { return ((uint32_t)ovl1 >> 1) & MMC_STAMP_MASK; }
__host__ __device__ inline int part_accept_of(int32_t ovl1) { return (int)((uint32_t)ovl1 >> 31); }

// dU of a move from the seven sums of its ONE result record, in mmc_combine_parts' arithmetic
// (mmc_system.inc; main.jl:574-593): the number Metropolis() is called with.  On the host and on
// the device, -ffp-contract=off on both: the same bits.
__host__ __device__ inline double mmc_move_delta(const double *w7, bool ov_old, bool ov_new, double factor)
{
    const double e0 = (0.0 + w7[0]) * 4, e1 = (0.0 + w7[1]) * 4;       // energy.jl:289
    double q0 = ov_old ? 0.0 : 0.0 + w7[4], q1 = ov_new ? 0.0 : 0.0 + w7[5]; // ewalds.jl:359-360
    q0 *= factor; q1 *= factor;                                         // ewalds.jl:905
    const double d_recip = (ov_old || ov_new) ? 0.0 : (0.0 + w7[6]) * factor; // main.jl:580-590
    return ((e1 - e0) + (q1 - q0)) + d_recip;                           // main.jl:593
}

// The accept decision made by the kernel that evaluated the move (launches of one part per move,
// device-side proposals): Metropolis(dU / T) && !overlap (main.jl:598, auxillary.jl:106-114) with
// the step's uniform from the chain's Philox counter (mmc_propose.hpp) -- what Driver::decide does
// on the host, without the trip there and back between two steps of a chain.  The kernel rewrites
// the replica's flag byte (accept | S-buffer << 1) for the NEXT launch's commit and sends the
// decision to the host in its result record, which the host uses for its bookkeeping as it is.
struct MoveRec;
struct DecideConsts {
    double temperature, factor;
    uint64_t seed;
    uint32_t replica0;
    int32_t ring_slots;     // several steps per launch: the ring of device-made move records,
    const MoveRec *ring;    // [ring_slots][ring_stride], by step % ring_slots
    int64_t ring_stride;
    uint8_t *flags; // [R]
};
// Several steps of a chain in ONE launch (the same wave takes the replica through them: what it
// wrote and read in one step is in the caches for the next): the launch sends ONE record per
// replica, a PartOut whose words are
//   [0] the sum of dU over the accepted steps, added up in step order
//   [1] bit k: step k accepted; bit 16 + k: step k saw an overlap; bit 32 + k: step k was a rotation
//       (an integer in a double's bits)
//   [2..6] 0;  word 7 as ever (stamp, checksum; bit 31: the LAST step's decision)
#define MMC_STEPS_PER_LAUNCH_MAX 16

// grid (n_parts, R).  n_parts == 1: the workgroup scans all molecules and then does the
// reciprocal part.  n_parts > 1: parts 0..n_parts-2 split the molecule range, the last part does
// the reciprocal part -- this is how a small replica count still fills 256 CUs.
// `rec`: the record array of homogeneous systems (kept in step by the commit) or NULL.
// `flagv`: when not NULL the per-replica flag byte comes from here instead of the record header
// (device-generated records are written before the accept decision of the previous step exists).
__global__ __launch_bounds__(MMC_BLOCK) void k_move_eval(BatchView bv, const MoveRec *cur,
                                                         const MoveRec *prev, PartOut *out,
                                                         int n_parts, PairParams pp, int r_base,
                                                         double *rec, const uint8_t *flagv,
                                                         unsigned stamp)
{
    __shared__ __align__(16) double pobuf[8];
    __shared__ Chosen ch;
    __shared__ Pending pd;
    __shared__ int32_t list[MMC_LIST_CAP];
    __shared__ int32_t wcnt[MMC_WAVES];
    __shared__ double red[6 * MMC_WAVES];
    __shared__ cplx tab[2][3][3][MMC_NKTAB];

    const int r = r_base + blockIdx.y, part = blockIdx.x, tid = threadIdx.x;
    const SysView s = sys_view(bv, r);
    const MoveRec *mv = cur + r;
    const int flags = flagv ? (int)flagv[r] : mv->flags;
    const bool commit = prev && (flags & 1);
    const int scur = (flags >> 1) & 1;
    const int i0 = mv->mol - 1;
    const int pend = commit ? prev[r].mol - 1 : -1;

    if (tid == 0) {
        pd.mol = pend;
        ch.i0 = i0; ch.na = 3;
    }
    if (commit && tid < 12) {
        if (tid < 3) pd.com[tid] = prev[r].com_new[tid];
        else pd.at[(tid - 3) / 3][(tid - 3) % 3] = prev[r].atoms_new[tid - 3];
    }
    __syncthreads();
    // the chosen molecule: state 0 = old (device state, or the pending commit), 1 = proposal
    if (tid < 3) {
        const int a = s.first0[i0] + tid;
        if (i0 == pend) {
            ch.at[0][tid][0] = pd.at[tid][0]; ch.at[0][tid][1] = pd.at[tid][1];
            ch.at[0][tid][2] = pd.at[tid][2];
            ch.com[0][tid] = pd.com[tid];
        } else {
            ch.at[0][tid][0] = s.ax[a]; ch.at[0][tid][1] = s.ay[a]; ch.at[0][tid][2] = s.az[a];
            ch.com[0][tid] = (tid == 0 ? s.comx[i0] : tid == 1 ? s.comy[i0] : s.comz[i0]);
        }
        ch.at[1][tid][0] = mv->atoms_new[3 * tid];
        ch.at[1][tid][1] = mv->atoms_new[3 * tid + 1];
        ch.at[1][tid][2] = mv->atoms_new[3 * tid + 2];
        ch.com[1][tid] = mv->com_new[tid];
        ch.q[tid] = s.charge[a];
        ch.type[tid] = s.atype[a];
    }
    // commit of the previous accepted move (main.jl:598-621): one workgroup writes it; every
    // reader in this launch substitutes `pd` for that molecule, so no ordering is required.
    if (part == 0 && commit && tid < 12) {
        double *w;
        const int fa = s.first0[pend];
        if (tid < 3) {
            w = (tid == 0 ? bv.comx : tid == 1 ? bv.comy : bv.comz) + r * bv.mol_stride + pend;
            *w = prev[r].com_new[tid];
        } else {
            const int a = (tid - 3) / 3, d = (tid - 3) % 3;
            w = (d == 0 ? bv.ax : d == 1 ? bv.ay : bv.az) + r * bv.atom_stride + fa + a;
            *w = prev[r].atoms_new[tid - 3];
        }
        if (rec) // [atoms(9), com(3)] record of the molecule (mmc_fast.hpp)
            rec[((int64_t)r * bv.n_mol + pend) * 16 + (tid < 3 ? 9 + tid : tid - 3)] = *w;
        if (tid < 3)
            comq_store(bv, r, pend, tid, *w);
    }
    if (part == 0 && commit && tid >= 12 && tid < 16)
        quat_commit(bv, r, pend, tid - 12, prev[r].q_new[tid - 12], quat_valid(prev[r].q_new));
    __syncthreads();

    const bool do_pairs = (n_parts == 1) || (part < n_parts - 1);
    const bool do_recip = (n_parts == 1) || (part == n_parts - 1);
    PartOut po;
    po.lj_pot[0] = po.lj_pot[1] = po.lj_vir[0] = po.lj_vir[1] = 0.0;
    po.qq_pot[0] = po.qq_pot[1] = po.recip = 0.0;
    po.ovl[0] = po.ovl[1] = 0;

    if (do_pairs) {
        const int np = (n_parts == 1) ? 1 : n_parts - 1;
        const int len = (s.n_mol + np - 1) / np;
        const int j0 = part * len, j1 = min(j0 + len, s.n_mol);
        PairAcc acc[2] = { { 0.0, 0.0, 0.0, 0 }, { 0.0, 0.0, 0.0, 0 } };
        pair_scan<2, true, true, 0, true>(s, &ch, &pd, j0, j1, pp, list, wcnt, acc);
        double v[6] = { acc[0].lj_pot, acc[1].lj_pot, acc[0].lj_vir,
                        acc[1].lj_vir, acc[0].qq_pot, acc[1].qq_pot }, tot[6];
        const int o0 = __syncthreads_or(acc[0].ovl), o1 = __syncthreads_or(acc[1].ovl);
        block_sum<6>(v, red, tot);
        if (tid == 0) {
            po.lj_pot[0] = tot[0]; po.lj_pot[1] = tot[1];
            po.lj_vir[0] = tot[2]; po.lj_vir[1] = tot[3];
            po.qq_pot[0] = tot[4]; po.qq_pot[1] = tot[5];
            po.ovl[0] = o0; po.ovl[1] = o1;
        }
    }
    if (do_recip) {
        if (tid < 18) {
            const int st = tid / 9, l = (tid % 9) / 3, d = tid % 3;
            phase_row(ch.at[st][l][d], s.box, tab[st][l][d]);
        }
        __syncthreads();
        const double *So = s_buf(bv, r, scur);
        double *Sn = s_buf(bv, r, scur ^ 1);
        double v[1] = { 0.0 }, tot[1];
        for (int k = tid; k < bv.nkvecs; k += MMC_BLOCK) {
            const int kx = bv.kxyz[3 * k], ky = bv.kxyz[3 * k + 1], kz = bv.kxyz[3 * k + 2];
            const double orr = So[2 * k], oi = So[2 * k + 1];
            double nr = orr, ni = oi;
#pragma unroll
            for (int l = 0; l < 3; l++) {
                const cplx tn = c_mul(c_mul(tab[1][l][0][5 + kx], tab[1][l][1][5 + ky]),
                                      tab[1][l][2][5 + kz]);
                const cplx to = c_mul(c_mul(tab[0][l][0][5 + kx], tab[0][l][1][5 + ky]),
                                      tab[0][l][2][5 + kz]);
                nr += ch.q[l] * (tn.re - to.re);
                ni += ch.q[l] * (tn.im - to.im);
            }
            Sn[2 * k] = nr; Sn[2 * k + 1] = ni;
            v[0] += bv.cfac[k] * ((nr * nr - (-ni) * ni) - (orr * orr - (-oi) * oi));
        }
        block_sum<1>(v, red, tot);
        if (tid == 0)
            po.recip = tot[0];
    }
    if (tid == 0) {
        pobuf[0] = po.lj_pot[0]; pobuf[1] = po.lj_pot[1];
        pobuf[2] = po.lj_vir[0]; pobuf[3] = po.lj_vir[1];
        pobuf[4] = po.qq_pot[0]; pobuf[5] = po.qq_pot[1];
        pobuf[6] = po.recip;
        pobuf[7] = pack_ovl(po.ovl[0], po.ovl[1], stamp, part_checksum(pobuf, stamp));
    }
    __syncthreads();
    store_part(out + (int64_t)r * n_parts + part, pobuf, tid);
}

// ---- small state updates -----------------------------------------------------------------------
struct SetMolArgs {
    int32_t r, i0, na, _pad;
    double com[3];
    double at[MMC_MAX_ATOMS][3];
};

// rec: the record array of homogeneous systems (12 doubles at a stride of 16, mmc_fast.hpp) or NULL.
__global__ void k_set_molecule(BatchView bv, SetMolArgs a, double *rec)
{
    const int t = threadIdx.x;
    if (t < 3)
        (t == 0 ? bv.comx : t == 1 ? bv.comy : bv.comz)[a.r * bv.mol_stride + a.i0] = a.com[t];
    if (t < a.na) {
        const int64_t o = a.r * bv.atom_stride + bv.first0[a.i0] + t;
        bv.ax[o] = a.at[t][0]; bv.ay[o] = a.at[t][1]; bv.az[o] = a.at[t][2];
    }
    if (rec && t < 12) {
        double *o = rec + ((int64_t)a.r * bv.n_mol + a.i0) * 16;
        o[t] = t < 9 ? a.at[t / 3][t % 3] : a.com[t - 9];
    }
    if (t < 3)
        comq_store(bv, a.r, a.i0, t, a.com[t]);
}

// Commit the outstanding proposal of every replica whose accept flag is set (main.jl:598-621),
// without evaluating a new one.  grid R, block 64.
// (16 lanes per replica, 16 replicas per block of 256: a block per replica was 32768 blocks of
// which 48 lanes each did nothing -- 100 us of a run's fixed cost)
__global__ void k_settle(BatchView bv, const MoveRec *prev, const int32_t *accept, int r_base, int nr)
{
    const int q = blockIdx.x * 16 + (threadIdx.x >> 4), t = threadIdx.x & 15;
    const int r = r_base + q;
    if (q >= nr || !accept[r])
        return;
    const int m = prev[r].mol - 1;
    if (t >= 12) {
        quat_commit(bv, r, m, t - 12, prev[r].q_new[t - 12], quat_valid(prev[r].q_new));
        return;
    }
    if (t < 3) {
        (t == 0 ? bv.comx : t == 1 ? bv.comy : bv.comz)[r * bv.mol_stride + m] =
            prev[r].com_new[t];
    } else {
        const int a = (t - 3) / 3, d = (t - 3) % 3;
        (d == 0 ? bv.ax : d == 1 ? bv.ay : bv.az)[r * bv.atom_stride + bv.first0[m] + a] =
            prev[r].atoms_new[t - 3];
    }
}

// sum q and sum q^2 (EwaldSelf ewalds.jl:832; Wolf terms energy.jl:924-932)
__global__ __launch_bounds__(MMC_BLOCK) void k_charge_sums(const double *q, int n, double *out)
{
    __shared__ double red[2 * MMC_WAVES];
    double v[2] = { 0.0, 0.0 }, tot[2];
    for (int i = threadIdx.x; i < n; i += MMC_BLOCK) {
        v[0] += q[i];
        v[1] += q[i] * q[i];
    }
    block_sum<2>(v, red, tot);
    if (threadIdx.x == 0) {
        out[0] = tot[0];
        out[1] = tot[1];
    }
}

// S(dst) = S(src) for replica 0: main.jl:621 (commit: old <- new) / :628 (rollback: new <- old)
__global__ void k_copy_s(BatchView bv, int dst, int src)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < 2 * bv.nkvecs)
        s_buf(bv, 0, dst)[k] = s_buf(bv, 0, src)[k];
}

// Copy up to 12 device arrays in one launch (the snapshot a volume move takes of everything it
// rewrites, and its restoration on rejection): segment q holds n[q] 32-bit words.
#define MMC_SNAP_SEGS 12
struct SnapSegs {
    uint32_t *dst[MMC_SNAP_SEGS];
    const uint32_t *src[MMC_SNAP_SEGS];
    int64_t n[MMC_SNAP_SEGS];
    int count;
};
__global__ void k_copy_segments(SnapSegs g)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int q = 0; q < g.count; q++)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < g.n[q]; i += stride)
            g.dst[q][i] = g.src[q][i];
}

// AoS (Julia Vector{SVector{3,Float64}}) -> SoA for one replica; n elements.
__global__ void k_aos_to_soa(const double *aos, double *x, double *y, double *z, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        x[i] = aos[3 * i]; y[i] = aos[3 * i + 1]; z[i] = aos[3 * i + 2];
    }
}
__global__ void k_soa_to_aos(const double *x, const double *y, const double *z, double *aos,
                             int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        aos[3 * i] = x[i]; aos[3 * i + 1] = y[i]; aos[3 * i + 2] = z[i];
    }
}

// ---- radial distribution histogram (the intent of Ewald/gr.jl `makeRDF`) -------------------------
// One site per molecule (site >= 0: atom slot of the molecule, e.g. 0 = the oxygens; site < 0: the
// centre of mass, gr.jl's cm mode).  All pairs i < j of one replica, gr.jl's minimum image
// (:75-80: strict < -side/2 -> + side, > side/2 -> - side), bin = ceil(r / dr) with
// dr = side / 2 / numbins (:5,87), counted when bin <= numbins (:88-90).  hist[0 .. numbins]
// (bin 0 only for coincident sites) is accumulated over the replicas of the launch.
__global__ __launch_bounds__(MMC_BLOCK) void k_rdf(BatchView bv, int site, int numbins,
                                                   unsigned long long *hist)
{
    extern __shared__ unsigned int sh_hist[];
    const int r = blockIdx.x, n = bv.n_mol;
    for (int k = threadIdx.x; k <= numbins; k += MMC_BLOCK)
        sh_hist[k] = 0u;
    __syncthreads();
    const double side = bv.box, sideh = side / 2.0, dr = sideh / numbins;
    auto load = [&](int j, double &x, double &y, double &z) {
        if (site < 0) {
            const int64_t m = r * bv.mol_stride + j;
            x = bv.comx[m]; y = bv.comy[m]; z = bv.comz[m];
        } else {
            const int64_t a = r * bv.atom_stride + bv.first0[j] + site;
            x = bv.ax[a]; y = bv.ay[a]; z = bv.az[a];
        }
    };
    for (int i = 0; i < n - 1; i++) {
        double xi, yi, zi;
        load(i, xi, yi, zi);
        for (int j = i + 1 + threadIdx.x; j < n; j += MMC_BLOCK) {
            double xj, yj, zj;
            load(j, xj, yj, zj);
            double xx = xi - xj, yy = yi - yj, zz = zi - zj;
            if (xx < -sideh) xx = xx + side;
            if (xx > sideh) xx = xx - side;
            if (yy < -sideh) yy = yy + side;
            if (yy > sideh) yy = yy - side;
            if (zz < -sideh) zz = zz + side;
            if (zz > sideh) zz = zz - side;
            const double rij = sqrt(xx * xx + yy * yy + zz * zz);
            const double b = ceil(rij / dr);
            if (b <= (double)numbins)
                atomicAdd(&sh_hist[(int)b], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= numbins; k += MMC_BLOCK)
        if (sh_hist[k])
            atomicAdd(&hist[k], (unsigned long long)sh_hist[k]);
}
