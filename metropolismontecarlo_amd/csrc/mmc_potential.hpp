// mmc_potential.hpp -- k_potential_one: potential(moa, soa, tot, ewalds, vdwTable, sim_props, "ewald")
// (Ewald/energy.jl:946-1032) for ONE system in ONE launch.
//
// What a Julia caller of `potential` (or of RecipLong) sees is a latency: round 2 spent 81 us at 750
// molecules on five small kernels, two device-to-host copies and a stream synchronisation -- the
// kernels themselves run for ~15 us.  Here one grid holds
//   * the pair workgroups: the body of k_total_wave with short units -- a wave per (molecule i,
//     optionally a range of molecules j > i) -- one partial per unit;
//   * the reciprocal workgroups: one per ((kx, ky) column that holds a k-vector, chunk of atoms),
//     each atom's three sincos (k_atom_phases' arithmetic) and the power recurrences of
//     ewalds.jl:575-585 in registers, the 2 * 11 sums of the column by block_sum, one partial per
//     (chunk, k);
//   * an epilogue in whichever workgroup finishes LAST: the unit partials added in
//     k_total_reduce's order, the chunk partials in chunk order into both S buffers
//     (ewalds.jl:600-601), sum_k cfac |S|^2 in k_recip_energy's order, and the four numbers
//     written straight into pinned host memory behind a stamp.
// The host polls the stamp: no copy, no stream synchronisation.
//
// Partials cross XCDs inside one launch, and the XCDs' L2s are not coherent with each other.
// Measured on the way (750 molecules, 450 reciprocal workgroups alone): __threadfence() before a
// ticket makes every workgroup write back its XCD's whole L2 -- 29 us; device-scope stores / loads
// for the partials and ONE ticket counter -- 23 us, the 638 read-modify-writes of one address from
// eight XCDs are serialised at the memory side; the epilogue as a second launch -- 13 + 9 us.  Now:
// device-scope stores and loads, a workgroup draws its ticket only after its stores have been
// acknowledged (s_waitcnt vmcnt(0)), and the ticket has two levels (POT_BUCKETS counters, then one).
#pragma once
#include "mmc_wave.hpp"

struct PotOneOut { // one 64-byte line of pinned host memory
    double lj_e, lj_v, qq, recip_e; // TotalsRaw's three sums; sum_k cfac |S(k)|^2 (no factor)
    int32_t n_ovl;
    uint32_t stamp;
    double _pad[3];
};
static_assert(sizeof(PotOneOut) == 64, "PotOneOut is one line");

struct PotOneArgs {
    TotalPart *tparts;  // [n_units]
    double *spart;      // [n_chunks][2 * nk_stride]
    unsigned *counter;  // device; zero between launches
    PotOneOut *out;     // pinned host
    int n_units, pair_wgs, j_chunk;   // pair part (pair_wgs == 0: none): units of (molecule, j-range)
    int n_cols, n_chunks, chunk_len;  // reciprocal part (n_cols == 0: none)
    unsigned stamp;
};

#define POT_MAX_CHUNKS 64
#define POT_CHUNK_ATOMS 768 // atoms per reciprocal workgroup (256: 30.9 us per potential() at 750 molecules, 768: 25.6)
#define POT_BUCKETS 32 // first-level tickets (a.counter[1 ..]); a.counter[0] is the second level

// The epilogue, by the workgroup that drew the last ticket: the unit partials in k_total_reduce's
// order, the chunk partials in chunk order into both S buffers, sum_k cfac |S|^2 in
// k_recip_energy's order; the results go straight into pinned host memory behind a stamp.
__device__ __forceinline__ void potential_finish(const BatchView &bv, const PotOneArgs &a, double *red, int tid)
{
    double lj_e = 0.0, lj_v = 0.0, qq = 0.0, recip_e = 0.0;
    int n_ovl = 0;
    if (a.pair_wgs > 0) { // k_total_reduce
        double v[4] = { 0, 0, 0, 0 }, tot[4];
        const double *p = reinterpret_cast<const double *>(a.tparts); // device-scope loads
        for (int k = tid; k < a.n_units; k += MMC_BLOCK) {
            v[0] += __hip_atomic_load(p + 4 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[1] += __hip_atomic_load(p + 4 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[2] += __hip_atomic_load(p + 4 * k + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[3] += (double)(int)__hip_atomic_load(reinterpret_cast<const long long *>(p + 4 * k + 3),
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        block_sum<4>(v, red, tot);
        lj_e = 2.0 * (tot[0] * 4);
        lj_v = 2.0 * (tot[1] * 24 / 3.0);
        qq = 2.0 * tot[2];
        n_ovl = tot[3] > 0 ? 1 : 0;
    }
    if (a.n_cols > 0) { // k_recip_finish + k_recip_energy
        double *s0 = s_buf(bv, 0, 0), *s1 = s_buf(bv, 0, 1);
        const double *sp = a.spart;
        double v[1] = { 0.0 }, tot[1];
        for (int k = tid; k < bv.nkvecs; k += MMC_BLOCK) {
            double re = 0.0, im = 0.0;
            for (int cz = 0; cz < a.n_chunks; cz++) {
                re += __hip_atomic_load(sp + (int64_t)cz * bv.nk_stride * 2 + 2 * k, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
                im += __hip_atomic_load(sp + (int64_t)cz * bv.nk_stride * 2 + 2 * k + 1, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
            }
            s0[2 * k] = re; s0[2 * k + 1] = im; // both arrays get `term` (ewalds.jl:600-601)
            s1[2 * k] = re; s1[2 * k + 1] = im;
            v[0] += bv.cfac[k] * (re * re - (-im) * im); // ewalds.jl:599
        }
        block_sum<1>(v, red, tot);
        recip_e = tot[0];
    }
    if (tid == 0) {
        PotOneOut *o = a.out;
        o->lj_e = lj_e; o->lj_v = lj_v; o->qq = qq; o->recip_e = recip_e;
        o->n_ovl = n_ovl;
        __hip_atomic_store(&o->stamp, a.stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <bool IMG>
__global__ __launch_bounds__(MMC_BLOCK) void k_potential_one(
    BatchView bv, const double *__restrict__ rec, const double *__restrict__ qq_tab, FastConsts fc,
    PairParams pp, RecipOrder order, PotOneArgs a)
{
    __shared__ __align__(16) TotalWaveShared sm;
    __shared__ double red[2 * MMC_NKTAB * MMC_WAVES];
    __shared__ int is_last;
    const int tid = threadIdx.x;

    if ((int)blockIdx.x < a.pair_wgs) {
        // ---- pair part: units blockIdx.x * 4 + wave, ... (paired = 0: one molecule per unit) ----
        total_wave_body<true, IMG>(sm, bv, rec, qq_tab, fc, pp, a.tparts, a.n_units, a.n_units, 0,
                              (int)blockIdx.x, a.pair_wgs, a.j_chunk);
    } else {
        // ---- reciprocal part: column ci, atom chunk cz ----
        const int idx = (int)blockIdx.x - a.pair_wgs;
        const int ci = idx / a.n_chunks, cz = idx - ci * a.n_chunks;
        const int c = order.col[ci];
        const int kx = c / MMC_NKTAB, ky = c % MMC_NKTAB - 5;
        const int aky = ky < 0 ? -ky : ky;
        const int16_t *col = bv.kmap + c * MMC_NKTAB;
        int k_lo = MMC_NKTAB, k_hi = -1;
        for (int k = 0; k < MMC_NKTAB; k++)
            if (col[k] >= 0) {
                k_lo = min(k_lo, k);
                k_hi = max(k_hi, k);
            }
        const double L = bv.box;
        double acc[2 * MMC_NKTAB];
#pragma unroll
        for (int k = 0; k < 2 * MMC_NKTAB; k++)
            acc[k] = 0.0;
        const int l_begin = cz * a.chunk_len, l_end = min(l_begin + a.chunk_len, (int)bv.n_atoms);
        for (int l = l_begin + tid; l < l_end; l += MMC_BLOCK) {
            const double q = bv.charge[l];
            double sn, cs;
            sincos_moderate(MMC_TWOPI * bv.ax[l] / L, sn, cs);
            const cplx x1 = { cs, sn };
            sincos_moderate(MMC_TWOPI * bv.ay[l] / L, sn, cs);
            const cplx y1 = { cs, sn };
            sincos_moderate(MMC_TWOPI * bv.az[l] / L, sn, cs);
            const cplx z1 = { cs, sn };
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
            const cplx qxy = c_mul(c_rmul(q, ex), ey); // (q*eikx)*eiky, ewalds.jl:589-597
#pragma unroll
            for (int k = 0; k < MMC_NKTAB; k++) {
                if (k >= k_lo && k <= k_hi) {
                    const cplx t = c_mul(qxy, ez[k]);
                    acc[2 * k] += t.re;
                    acc[2 * k + 1] += t.im;
                }
            }
        }
        // the sums of the column's valid kz over the workgroup: DPP row scans per wave
        // (wave_sum_rows: the total in every lane), the four waves through LDS in wave order
        const int wv = tid >> 6, ln = tid & 63;
#pragma unroll
        for (int k = 0; k < 2 * MMC_NKTAB; k++) {
            if (k >= 2 * k_lo && k <= 2 * k_hi + 1) { // (workgroup-uniform)
                const double sw = wave_sum_rows(acc[k]);
                if (ln == 0)
                    red[k * MMC_WAVES + wv] = sw;
            }
        }
        __syncthreads();
        if (tid < 2 * MMC_NKTAB && tid >= 2 * k_lo && tid <= 2 * k_hi + 1) {
            const int ki = col[tid >> 1];
            if (ki >= 0) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < MMC_WAVES; w++)
                    t += red[tid * MMC_WAVES + w];
                __hip_atomic_store(a.spart + (int64_t)cz * bv.nk_stride * 2 + 2 * ki + (tid & 1), t,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

    // ---- who is last?  A two-level ticket: POT_BUCKETS counters taken by ~n/POT_BUCKETS workgroups
    // each, and one taken by the last of every bucket -- tickets on ONE address from eight XCDs are
    // serialised at the memory side (638 of them: 20 us) ----
    // Ordering, spelled out (the advisor asked why these are relaxed atomics): every partial is written
    // with an agent-scope atomic store and read by potential_finish with agent-scope atomic loads, i.e.
    // both bypass the non-coherent part of the XCD's L2; the s_waitcnt below makes this wave's stores
    // ACKNOWLEDGED by the memory side before its workgroup can draw a ticket (the barrier after it
    // extends that to every wave), and the ticket counters are read-modify-writes at the same memory
    // side: whoever draws the last ticket draws it after every partial has landed.  A release/acquire
    // pair at agent scope on the ticket would say the same to the compiler -- and compiles to a
    // write-back of the XCD's whole L2 per workgroup (measured: 29 us for the 450 reciprocal
    // workgroups alone, against 15 us for the whole kernel as written).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's partial stores are acknowledged
    __syncthreads();
    if (tid == 0) {
        const unsigned nb = min((unsigned)POT_BUCKETS, gridDim.x), bk = blockIdx.x % nb;
        const unsigned in_bucket = gridDim.x / nb + (bk < gridDim.x % nb ? 1u : 0u);
        int last = 0;
        if (__hip_atomic_fetch_add(a.counter + 1 + bk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
            == in_bucket - 1) {
            __hip_atomic_store(a.counter + 1 + bk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb - 1) {
                __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        is_last = last;
    }
    __syncthreads();
    if (!is_last)
        return;
    potential_finish(bv, a, red, tid);
}

