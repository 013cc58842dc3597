// mmc_propose.hpp -- trial-move generation on the device.
//
// The reference draws its proposals from Julia's global RNG inside Loop() (Ewald/main.jl:516-552:
// `chose_move = rand()`, random_translate_vector, random_rotate_quaternion), which it never seeds,
// so there is no stream to reproduce; what is reproduced is the distribution of each move:
//   translation  auxillary.jl:94-103   COM + (zeta - 1/2) * dr_max per axis, then PBC()
//                boundaries.jl:16-26   strict > box / < 0 wrap; the atoms follow rigidly
//   rotation     quaternions.jl:52-74  random axis by rejection from the cube
//                quaternions.jl:158-182 angle uniform in +-dphi_max about that axis
// Randomness is counter based (Philox4x32-10, Salmon et al. SC'11): every draw is a pure function
// of (seed, global replica index, step, slot) -- key = seed, counter = (step, slot, replica) -- so
// the host can re-derive the move kind and take the Metropolis uniform of the same step without
// any state, whatever the grouping of the replicas, and streams of different (seed, replica)
// pairs never coincide.
#pragma once
#include <stdint.h>

#include "mmc_kernels.hpp"

struct Philox {
    uint32_t v[4];
};

__host__ __device__ inline Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int round = 0; round < 10; round++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox{ { c0, c1, c2, c3 } };
}

// Slots of one (chain, step): each slot yields two uniforms in [0, 1).
enum { MMC_SLOT_KIND = 0, MMC_SLOT_MOVE = 1, MMC_SLOT_METROPOLIS = 2, MMC_SLOT_AXIS = 3 };
// (the axis slots count up from 3; the volume move of an NPT chain draws from MMC_SLOT_VOLUME =
// 0x40000000, include/mmc_hip.h)

struct Uniform2 {
    double a, b;
};

// One chain's stream: the run's seed and the chain's GLOBAL replica index.
struct ChainKey {
    uint64_t seed;
    uint32_t replica;
};

__host__ __device__ inline Uniform2 mmc_draw(ChainKey ck, uint64_t step, uint32_t slot)
{
    const Philox x = philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), slot, ck.replica,
                                   (uint32_t)ck.seed, (uint32_t)(ck.seed >> 32));
    Uniform2 u;
    u.a = (double)((((uint64_t)x.v[0] << 32) | x.v[1]) >> 11) * 0x1.0p-53;
    u.b = (double)((((uint64_t)x.v[2] << 32) | x.v[3]) >> 11) * 0x1.0p-53;
    return u;
}

// main.jl:516-517: `chose_move = rand()`, translation when < probability_of_move["translation"]
__host__ __device__ inline int mmc_move_kind(ChainKey chain_seed, uint64_t step)
{
    return mmc_draw(chain_seed, step, MMC_SLOT_KIND).a < 0.5 ? 0 : 1;
}

__host__ __device__ inline double mmc_metropolis_uniform(ChainKey chain_seed, uint64_t step)
{
    return mmc_draw(chain_seed, step, MMC_SLOT_METROPOLIS).a;
}

__device__ inline double pbc_wrap(double x, double box) // boundaries.jl:16-26
{
    if (x > box) x -= box;
    if (x < 0) x += box;
    return x;
}

// The move as a rigid transformation of the chosen molecule: drawn once (propose_xform), applied
// to each atom (apply_xform).  Shared by k_propose and the fused path of k_move_eval_fast -- there
// lanes 0..2 apply it to one atom each -- so both produce the same doubles.
struct MoveXform {
    int kind;          // 0 translation, 1 rotation
    double com_new[3];
    double d[3];       // translation: com_new - com (the atoms follow the wrapped COM rigidly)
    double Rm[3][3];   // rotation about the COM
};

__device__ inline MoveXform propose_xform(ChainKey cs, uint64_t step, double box, double dr_max,
                                          double dphi_max, const double *com)
{
    MoveXform x;
    const Uniform2 u0 = mmc_draw(cs, step, MMC_SLOT_KIND);
    const Uniform2 u1 = mmc_draw(cs, step, MMC_SLOT_MOVE);
    if (u0.a < 0.5) {
        x.kind = 0;
        const double zeta[3] = { u0.b - 0.5, u1.a - 0.5, u1.b - 0.5 };
#pragma unroll
        for (int k = 0; k < 3; k++) {
            x.com_new[k] = pbc_wrap(com[k] + zeta[k] * dr_max, box);
            x.d[k] = x.com_new[k] - com[k];
            x.Rm[k][0] = x.Rm[k][1] = x.Rm[k][2] = 0.0;
        }
    } else {
        x.kind = 1;
        double e[3], norm;
        uint32_t slot = MMC_SLOT_AXIS;
        do {
            const Uniform2 p = mmc_draw(cs, step, slot), q = mmc_draw(cs, step, slot + 1);
            slot += 2;
            e[0] = 2.0 * p.a - 1.0; e[1] = 2.0 * p.b - 1.0; e[2] = 2.0 * q.a - 1.0;
            norm = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
        } while (!(norm < 1.0) || norm == 0.0);
        const double inv = 1.0 / sqrt(norm);
#pragma unroll
        for (int k = 0; k < 3; k++) e[k] *= inv;
        const double angle = (2.0 * u1.a - 1.0) * dphi_max;
        double s, c;
        sincos(angle, &s, &c);
        const double tt = 1.0 - c;
        x.Rm[0][0] = tt * e[0] * e[0] + c;        x.Rm[0][1] = tt * e[0] * e[1] - s * e[2];
        x.Rm[0][2] = tt * e[0] * e[2] + s * e[1]; x.Rm[1][0] = tt * e[0] * e[1] + s * e[2];
        x.Rm[1][1] = tt * e[1] * e[1] + c;        x.Rm[1][2] = tt * e[1] * e[2] - s * e[0];
        x.Rm[2][0] = tt * e[0] * e[2] - s * e[1]; x.Rm[2][1] = tt * e[1] * e[2] + s * e[0];
        x.Rm[2][2] = tt * e[2] * e[2] + c;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            x.com_new[k] = com[k];
            x.d[k] = 0.0;
        }
    }
    return x;
}

__device__ inline void apply_xform(const MoveXform &x, const double *com, const double *at,
                                   double *at_new)
{
    if (x.kind == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) at_new[k] = at[k] + x.d[k];
    } else {
        const double o[3] = { at[0] - com[0], at[1] - com[1], at[2] - com[2] };
#pragma unroll
        for (int k = 0; k < 3; k++)
            at_new[k] = com[k] + x.Rm[k][0] * o[0] + x.Rm[k][1] * o[1] + x.Rm[k][2] * o[2];
    }
}

// ---- the reference's own move generation: quaternions + body-fixed sites -------------------------
// (selected by mmc_batch_set_orientations; Ewald/main.jl:516-549)
//   q_to_a              quaternions.jl:11-50.  `faithful`: element (2,3) as the reference has it,
//                       2*(q2*q4 + q1*q2) -- a typo for 2*(q3*q4 + q1*q2), SURVEY quirk Q12, which
//                       makes the matrix non-orthogonal; !faithful: the Allen & Tildesley matrix.
//   quatmul             quaternions.jl:76-91
//   rotate_quaternion   quaternions.jl:93-120   rot = (cos(angle/2), sin(angle/2) * axis), rot * old
//   MATMUL              auxillary.jl:154-159    (dot(db, a[:,1]), dot(db, a[:,2]), dot(db, a[:,3]))
__host__ __device__ inline void mmc_q_to_a(const double *q, int faithful, double a[3][3])
{
    const double q1 = q[0], q2 = q[1], q3 = q[2], q4 = q[3];
    a[0][0] = q1 * q1 + q2 * q2 - q3 * q3 - q4 * q4;
    a[0][1] = 2 * (q2 * q3 + q1 * q4);
    a[0][2] = 2 * (q2 * q4 - q1 * q3);
    a[1][0] = 2 * (q2 * q3 - q1 * q4);
    a[1][1] = q1 * q1 - q2 * q2 + q3 * q3 - q4 * q4;
    a[1][2] = faithful ? 2 * (q2 * q4 + q1 * q2) : 2 * (q3 * q4 + q1 * q2);
    a[2][0] = 2 * (q2 * q4 + q1 * q3);
    a[2][1] = 2 * (q3 * q4 - q1 * q2);
    a[2][2] = q1 * q1 - q2 * q2 - q3 * q3 + q4 * q4;
}

__host__ __device__ inline void mmc_quatmul(const double *a, const double *b, double *c)
{
    c[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    c[1] = a[1] * b[0] + a[0] * b[1] - a[3] * b[2] + a[2] * b[3];
    c[2] = a[2] * b[0] + a[3] * b[1] + a[0] * b[2] - a[1] * b[3];
    c[3] = a[3] * b[0] - a[2] * b[1] + a[1] * b[2] + a[0] * b[3];
}

// ra[a] = COM + MATMUL(ai, db[a]) for the three sites (main.jl:543-549)
__host__ __device__ inline void mmc_space_fixed(const double *com, const double a[3][3],
                                                const double *db /* [3][3] */, double *at_new)
{
    for (int s = 0; s < 3; s++)
        for (int k = 0; k < 3; k++)
            at_new[3 * s + k] = com[k] + (db[3 * s] * a[0][k] + db[3 * s + 1] * a[1][k]
                                          + db[3 * s + 2] * a[2][k]);
}

// n bytes from mapped pinned host memory into device memory (the per-step flag bytes).
__global__ void k_fetch_bytes(uint8_t *dst, const uint8_t *src, int n)
{
    const int i = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 3 < n && ((reinterpret_cast<uintptr_t>(src + i) | reinterpret_cast<uintptr_t>(dst + i)) & 3) == 0)
        *reinterpret_cast<uint32_t *>(dst + i) = *reinterpret_cast<const uint32_t *>(src + i);
    else
        for (int k = i; k < n && k < i + 4; k++)
            dst[k] = src[k];
}

// What a launch of k_propose needs.
struct GenArgs {
    const double2 *steps; // [R] {dr_max, dphi_max} of the chain
    uint64_t seed;        // chain r draws from ChainKey{seed, replica0 + r} at counter
    uint64_t replica0;    // rng_off + step
    int64_t rng_off;
    int64_t step0;        // first step generated by this launch
    int n_gen;            // steps generated: step0 .. step0 + n_gen - 1
    int ring;             // slots of the record ring; step s lives in slot s % ring
    int64_t ring_stride;  // records per slot (= replicas of the batch)
    const uint8_t *flag0; // [R] flag byte of step0 (only consulted when n_mol == 1, see below)
    int quat_mode;        // 0: rigid transformation of the current atoms; 1 / 2: the reference's
                          // quaternion route with the faithful / the Allen-Tildesley q_to_a
    double db[9];         // body-fixed sites db[a] of the molecule (quat_mode != 0)
};

// One thread per (replica, step): read the chosen molecule's current state, draw the move and
// write the move record k_move_eval* consumes into ring slot (step % ring).  Step s moves
// molecule s % n_mol (main.jl:490), so the records of up to n_mol - 1 consecutive steps can be
// generated before any of them has been decided: the accept decisions in between only touch
// OTHER molecules, and the one decision still pending at generation time (step0 - 1) concerns
// molecule (step0 - 1) % n_mol, which none of the generated steps moves.  The host guarantees
// n_gen <= max(1, n_mol - 1).  With n_mol == 1 (n_gen == 1) the pending proposal is the same
// molecule and is substituted here like the move kernel does.  The flag byte (accept of the
// previous step, S buffer) is therefore NOT part of these records: the move kernel takes it from
// its `flagv` argument.
__global__ void k_propose(BatchView bv, const double *rec, MoveRec *ring, GenArgs ga, int r_base,
                          int nr, int has_prev)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nr * ga.n_gen)
        return;
    const int k = t / nr, r = r_base + (t - k * nr);
    const int64_t step = ga.step0 + k;
    const int i0 = (int)(step % bv.n_mol);
    MoveRec *cur = ring + (int64_t)(step % ga.ring) * ga.ring_stride + r;
    double com[3], at[9];
    bool sub = false;
    if (bv.n_mol == 1 && has_prev && (ga.flag0[r] & 1)) { // the same molecule again
        const MoveRec *prev = ring + (int64_t)((step + ga.ring - 1) % ga.ring) * ga.ring_stride + r;
        for (int q = 0; q < 3; q++) com[q] = prev->com_new[q];
        for (int q = 0; q < 9; q++) at[q] = prev->atoms_new[q];
        sub = true;
    }
    if (!sub && rec) {
        const double *src = rec + ((int64_t)r * bv.n_mol + i0) * 16; // MMC_RSTRIDE
        for (int q = 0; q < 9; q++) at[q] = src[q];
        for (int q = 0; q < 3; q++) com[q] = src[9 + q];
    } else if (!sub) {
        const int64_t m0 = r * bv.mol_stride + i0, a0 = r * bv.atom_stride + bv.first0[i0];
        com[0] = bv.comx[m0]; com[1] = bv.comy[m0]; com[2] = bv.comz[m0];
        for (int a = 0; a < 3; a++) {
            at[3 * a] = bv.ax[a0 + a]; at[3 * a + 1] = bv.ay[a0 + a]; at[3 * a + 2] = bv.az[a0 + a];
        }
    }
    const double2 sz = ga.steps[r];
    MoveRec m;
    m.mol = i0 + 1;
    m.flags = 0;
    for (int q = 0; q < 3; q++) m.com_old[q] = com[q];
    for (int q = 0; q < 9; q++) m.atoms_old[q] = at[q];
    const ChainKey ck{ ga.seed, (uint32_t)(ga.replica0 + (uint64_t)r) };
    const uint64_t rs = (uint64_t)(ga.rng_off + step);
    if (ga.quat_mode && bv.quat) {
        // the reference's route (main.jl:516-549): the orientation is a quaternion, the atoms are
        // rebuilt from the body-fixed sites on every move
        double q_old[4], q_new[4], A[3][3];
        const double *qsrc = bv.quat + ((int64_t)r * bv.n_mol + i0) * 4;
        if (sub) {
            const MoveRec *prev = ring + (int64_t)((step + ga.ring - 1) % ga.ring) * ga.ring_stride + r;
            if (quat_valid(prev->q_new))
                qsrc = prev->q_new;
        }
        for (int q = 0; q < 4; q++) q_old[q] = qsrc[q];
        const Uniform2 u0 = mmc_draw(ck, rs, MMC_SLOT_KIND);
        const Uniform2 u1 = mmc_draw(ck, rs, MMC_SLOT_MOVE);
        m.flags = u0.a < 0.5 ? 0 : 256; // (bit 8: the kind of the move, for a kernel that keeps the counts)
        if (u0.a < 0.5) { // translation: random_translate_vector + PBC, ei = quat[i] (:519-528)
            const double zeta[3] = { u0.b - 0.5, u1.a - 0.5, u1.b - 0.5 };
            for (int k = 0; k < 3; k++) m.com_new[k] = pbc_wrap(com[k] + zeta[k] * sz.x, bv.box);
            for (int q = 0; q < 4; q++) q_new[q] = q_old[q];
        } else {          // rotation: random_rotate_quaternion (:529-536)
            double e[3], norm;
            uint32_t slot = MMC_SLOT_AXIS;
            do { // random_vector (quaternions.jl:52-74)
                const Uniform2 p = mmc_draw(ck, rs, slot), q = mmc_draw(ck, rs, slot + 1);
                slot += 2;
                e[0] = 2.0 * p.a - 1.0; e[1] = 2.0 * p.b - 1.0; e[2] = 2.0 * q.a - 1.0;
                norm = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
            } while (!(norm < 1.0) || norm == 0.0);
            const double sq = sqrt(norm);
            for (int k = 0; k < 3; k++) e[k] = e[k] / sq;           // e ./ sqrt(norm)
            const double angle = (2.0 * u1.a - 1.0) * sz.y;          // quaternions.jl:176
            double sh, ch;
            sincos(0.5 * angle, &sh, &ch);
            const double rot[4] = { ch, sh * e[0], sh * e[1], sh * e[2] }; // :113-114
            mmc_quatmul(rot, q_old, q_new);                          // :116
            for (int k = 0; k < 3; k++) m.com_new[k] = com[k];
        }
        mmc_q_to_a(q_new, ga.quat_mode == 1, A);
        mmc_space_fixed(m.com_new, A, ga.db, m.atoms_new);
        for (int q = 0; q < 4; q++) m.q_new[q] = q_new[q];
    } else {
        const MoveXform x = propose_xform(ck, rs, bv.box, sz.x, sz.y, com);
        m.flags = x.kind << 8;
        for (int q = 0; q < 3; q++) m.com_new[q] = x.com_new[q];
        for (int a = 0; a < 3; a++)
            apply_xform(x, com, &at[3 * a], &m.atoms_new[3 * a]);
        for (int q = 0; q < 4; q++) m.q_new[q] = 0.0;
    }
    *cur = m;
}
