// mmc_wave.hpp -- k_move_eval_wave: the per-move kernel (K1+K4) with ONE WAVEFRONT PER TRIAL MOVE.
//
// Why (profiles/round1_default_pmc_summary.json): the workgroup-per-move form
// (k_move_eval_fast) spent 1542 VALU instructions in each of its four waves, 58 % of its wave
// cycles parked at barriers / s_waitcnt, and ran at 0.64 of the algorithmic roofline.  Three
// structural costs: (1) the chosen molecule's coordinates were lane-varying LDS reads (a, b and
// the state differ per lane), (2) every pass dealt its items to 256 lanes, so the LJ pass ran at
// 46 % and the reciprocal pass at 66 % lane utilisation and one wave carried a nearly empty fifth
// Coulomb iteration, (3) eight barriers per move serialised the four waves.
//
// Here a wave owns a whole (replica, part) unit and never synchronises with another wave:
//   * the chosen molecule (old and new state, 24 doubles) lives in SGPRs: every lane works on a
//     different NEIGHBOUR, all lanes on the same atom pair (a, b) and state, so chosen atoms,
//     q_a q_b and the LJ parameters are scalar operands of the VALU instructions;
//   * lane n holds neighbour n's whole 96-byte record in registers (six 16-byte loads straight
//     from HBM/L2, no LDS staging) and runs the 9 atom pairs x 2 states over it; the LJ term of a
//     pair with eps > 0.001 reuses the Coulomb term's minimum-image vector and r^2;
//   * the COM scan runs on 16-bit box fractions of the centres of mass (mmc_kernels.hpp:
//     com_quant; the wrapped integer difference is the minimum image) and only decides which
//     records are gathered -- the reference's fp64 comparison on the record decides the rest;
//   * the COM scan, the neighbour loop and the reciprocal loop each run at >= 90 % lane
//     utilisation (750 / 768, 117 / 128, 337 / 384);
//   * workgroups are persistent (grid = a few per CU, each wave loops over units), so the 14 KB
//     erfc table is copied to LDS once per workgroup, not once per move, and launch tails vanish;
//   * latency is hidden by the other resident waves, which are at unrelated points of their own
//     moves -- no barrier ever lines them up.
// Branch decisions (gates, overlap, slack) are the reference's comparisons on unfused arithmetic,
// as in every other kernel here; only the summation order differs from k_move_eval_fast.
#pragma once
#include "mmc_fast.hpp"
#include "mmc_propose.hpp"

#ifndef WV_WAVES
#define WV_WAVES 4   // waves (= units in flight) per workgroup
#endif
#ifndef WV_OCC
#define WV_OCC 5     // waves per SIMD k_move_eval_wave is compiled for: 96 VGPRs, and 31 KB of LDS per
#endif               // workgroup of four waves (five workgroups per compute unit)
#ifndef WV_MWAVES
#define WV_MWAVES WV_WAVES // waves per workgroup of k_move_eval_wave (they share one copy of the erfc table)
#endif
#ifndef WV_LIST
#define WV_LIST 640  // neighbour-list slots per wave; the scan empties it when fewer than 64 WV_PF are free
#endif
#ifndef WV_PF
#define WV_PF 6      // 64-molecule blocks of the COM scan in flight ahead of the one being tested
#endif
// Bytes readable past the end of BatchView::comq: a scan prefetches WV_PF blocks ahead of the block
// it tests, WV_PF blocks per trip, from the last molecule of the last replica at worst.
#define MMC_CQ_PAD (4 * 64 * (2 * WV_PF + 2))
#ifndef WV_TPF
#define WV_TPF 3     // ... in k_total_wave, whose scans are 6 blocks long on average (4: spills at 96 VGPRs)
#endif

// Neighbour lists hold 16-bit molecule indices (half the LDS: with it five workgroups fit a compute
// unit); the host takes these kernels for at most MMC_WAVE_MAX_MOL molecules.
typedef uint16_t wv_list_t;
#define MMC_WAVE_MAX_MOL 65535
template <int NW> struct WaveSharedT {
    alignas(16) double qtab[MMC_QQ_TABLE_DOUBLES];
    cplx ptab[NW][2][3][3][MMC_NKTAB]; // phase tables of the 3 moved atoms, old and new
    wv_list_t list[NW][WV_LIST];
    alignas(16) double pvw[NW][12];    // pending commit of the unit's replica, record layout
    alignas(16) double outw[NW][8];    // the PartOut being assembled
};
typedef WaveSharedT<WV_WAVES> WaveShared;

// Diagnostic build (-DWV_STAMPS, scripts/stamps.sh): lane 0 of every unit records the shader clock
// at the boundaries of its phases into a device array the host reads back (mmc_debug_stamps).
// Compiled out of the product: the stamps and their s_waitcnt would slow what they measure.
#ifdef WV_STAMPS
#define WV_NSTAMP 8
__device__ unsigned long long g_wv_stamps[65536 * WV_NSTAMP];
#define WV_STAMP(i)                                                                              \
    do {                                                                                         \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                              \
        if (lane == 0 && unit < 65536)                                                           \
            g_wv_stamps[unit * WV_NSTAMP + (i)] = __builtin_amdgcn_s_memtime();                  \
    } while (0)
#else
#define WV_STAMP(i)
#endif

// Orders this wave's own LDS traffic (a wave's DS instructions execute in order; this only stops
// the compiler from moving accesses across the point).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Six sums over the wave at once, ADDED to out[0..5] by lanes 0..5 (out and scratch: this wave's
// LDS; scratch 24 doubles).  Bit for bit wave_sum_rows of each -- the same additions in the same
// order -- for a third of the vector instructions (~55 instead of ~150 per unit of the move kernel):
//   * the first step packs two sums into one register: even lanes keep a_i + a_(i+1), odd lanes
//     b_i + b_(i-1) (a quad swap of what the lane does not keep), so the three row_shr steps that
//     follow -- 2, 4, 8 stay within a parity -- run on three registers instead of six, and lanes 14
//     and 15 of a row end with the row totals of a and b that lane 15 would have held;
//   * the four row totals of each sum go through LDS to the lane that adds them, in row order,
//     instead of through eight v_readlane per sum into every lane.
__device__ __forceinline__ double dpp_quad_swap_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xf, 0xf, true),
                   hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0xB1, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ void wave_sum6_add(double v0, double v1, double v2, double v3, double v4,
                                              double v5, double *out, double *scratch, int lane)
{
    const bool odd = (lane & 1) != 0;
    double p0 = (odd ? v1 : v0) + dpp_quad_swap_f64(odd ? v0 : v1);
    double p1 = (odd ? v3 : v2) + dpp_quad_swap_f64(odd ? v2 : v3);
    double p2 = (odd ? v5 : v4) + dpp_quad_swap_f64(odd ? v4 : v5);
    p0 += dpp_row_shr_f64<2>(p0); p1 += dpp_row_shr_f64<2>(p1); p2 += dpp_row_shr_f64<2>(p2);
    p0 += dpp_row_shr_f64<4>(p0); p1 += dpp_row_shr_f64<4>(p1); p2 += dpp_row_shr_f64<4>(p2);
    p0 += dpp_row_shr_f64<8>(p0); p1 += dpp_row_shr_f64<8>(p1); p2 += dpp_row_shr_f64<8>(p2);
    if ((lane & 14) == 14) { // lanes 14 and 15 of each row: the row totals of (v0, v1), (v2, v3), (v4, v5)
        double *dst = scratch + 4 * (lane & 1) + (lane >> 4);
        dst[0] = p0; dst[8] = p1; dst[16] = p2;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 6) {
        const double *r = scratch + 4 * lane;
        out[lane] += ((r[0] + r[1]) + r[2]) + r[3];
    }
}
// (The same packing for the three sums of k_total_wave measured 12 % SLOWER per evaluation, round 4:
// that kernel is bound by vector issue with its sums in flight beside the next unit's scan.)
// One sum, the total STORED to *out by lane 0 (scratch: 4 doubles): wave_sum_rows' bits.
__device__ __forceinline__ void wave_sum1_store(double v, double *out, double *scratch, int lane)
{
    v += dpp_row_shr_f64<1>(v);
    v += dpp_row_shr_f64<2>(v);
    v += dpp_row_shr_f64<4>(v);
    v += dpp_row_shr_f64<8>(v);
    if ((lane & 15) == 15)
        scratch[lane >> 4] = v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0)
        *out = ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
}

__device__ __forceinline__ int lane_i32(int v, int src)
{
    return __builtin_amdgcn_readlane(v, src);
}

// a value every lane holds alike, moved to scalar registers for good (through asm: the compiler
// cannot fall back on the vector copy, which it would keep alive -- and spill -- beside this one)
__device__ __forceinline__ double uniform_f64(double v)
{
    const long long b = __double_as_longlong(v);
    unsigned lo, hi;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(lo) : "v"((int)b));
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(hi) : "v"((int)(b >> 32)));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// lane `src`'s double as a wave-uniform value (two v_readlane_b32 -> an SGPR pair)
__device__ __forceinline__ double lane_f64(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)b, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// phase_row (mmc_device.hpp) on sincos_moderate: the recurrence of ewalds.jl:564-585 unchanged.
__device__ __forceinline__ void phase_row_moderate(double x, double L, cplx *row /* [MMC_NKTAB] */)
{
    double sn, cs;
    sincos_moderate(MMC_TWOPI * x / L, sn, cs);
    const cplx e1 = { cs, sn }, one = { 1.0, 0.0 };
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

// The kernel's own Metropolis decision (DecideConsts, mmc_kernels.hpp): every lane computes it from
// the unit's seven sums in LDS; lane 0 rewrites the replica's flag byte.  Returns accept (0 / 1).
__device__ __forceinline__ int wave_decide(const double *outw, bool ov_old, bool ov_new,
                                           const DecideConsts *__restrict__ dc, long long step, int r,
                                           int scur, int lane, double *delta_out = nullptr, bool store_flags = true)
{
    const double delta = mmc_move_delta(outw, ov_old, ov_new, dc->factor);
    if (delta_out)
        *delta_out = delta;
    const double x = delta / dc->temperature;
    const double u = mmc_metropolis_uniform(ChainKey{ dc->seed, dc->replica0 + (uint32_t)r }, (uint64_t)step);
    // exp(-x) > u, decided without the exponential where 1 - x <= exp(-x) <= 1 / (1 + x) (x >= 0)
    // already says which (the margins are far above the rounding of the two bounds): every lane of
    // the wave holds the same x, so these are branches, and exp() -- ~50 fp64 instructions that
    // cost the SIMD the same for one lane as for 64 -- runs for about a third of the moves.
    bool met;
    if (x < 0.0) met = true;
    else if (1.0 - x > u + 1e-9) met = true;
    else if ((1.0 + x) * u > 1.0 + 1e-9) met = false;
    else met = exp(-x) > u;
    const int acc = (met && !(ov_old || ov_new)) ? 1 : 0;
    if (store_flags && lane == 0)
        dc->flags[r] = (uint8_t)(acc | ((scur ^ acc) << 1));
    return acc;
}

// grid: any number of workgroups of WV_WAVES waves; wave w of workgroup g handles units
// g * WV_WAVES + w, + gridDim.x * WV_WAVES, ...  Unit u = (replica r_base + u / n_parts,
// part u % n_parts); part semantics as k_move_eval (the last part of n_parts > 1 does the
// reciprocal part, the others split the molecule range).
// SUBST = false (launches with n_parts == 1 only): the wave that writes a pending commit to memory
// is the only reader of that replica in the launch -- it waits for its own stores (same compute
// unit, same L1) instead of substituting the pending words in every gather and scan block.
// IMG = true: the minimum image of an atom pair from the image of its molecule (WV_IMG in
// mmc_wave_unit.inc; the launch site checks the condition).
// MULTI = true: several steps of the chain per launch (n_sub; the kernel decides) -- its own
// instantiation, so that the one-step form compiles exactly as it did.
template <bool SUBST, bool IMG, bool MULTI = false>
__global__ __launch_bounds__(WV_MWAVES * 64) __attribute__((amdgpu_waves_per_eu(WV_OCC, WV_OCC))) void k_move_eval_wave(
    BatchView bv, double *rec, const double *__restrict__ qq_tab,
    const int32_t *__restrict__ kpack, FastConsts fc, const MoveRec *__restrict__ cur,
    const MoveRec *__restrict__ prev, PartOut *out, int n_parts, PairParams pp, int r_base,
    int n_units, const uint8_t *flagv, unsigned stamp, const DecideConsts *__restrict__ dc, long long dec_step,
    int n_sub_arg, int slot0)
{
    const int n_sub = MULTI ? n_sub_arg : 1;
    __shared__ __align__(16) WaveSharedT<WV_MWAVES> sm;
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int k = tid; k < MMC_QQ_TABLE_DOUBLES; k += WV_MWAVES * 64)
        sm.qtab[k] = qq_tab[k];
    __syncthreads(); // the only workgroup barrier: from here on the waves are independent

    const int n_mol = bv.n_mol, nkv = bv.nkvecs;
    const double box = bv.box;
    const BoxConsts bc = box_consts(box);
    const int np = (n_parts == 1) ? 1 : n_parts - 1;
    const int plen = (n_mol + np - 1) / np;
    const bool same_gate = pp.lj_gate_sq == pp.qq_gate_sq;
    // prefilter of the COM scan: 16-bit box fractions (com_quant, mmc_kernels.hpp)
    // (scalars, opaquely: the VGPR copies of values computed before the loop were held for the
    // kernel's life and spilled to scratch)
    const double inv_box = uniform_f64(1.0 / box);
    uint32_t gate_q;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(gate_q) : "v"(com_quant_gate(fmax(pp.lj_gate_sq, pp.qq_gate_sq), box)));
    wv_list_t *const list = sm.list[wv];
    const double *const pvw = sm.pvw[wv];

    for (int unit = blockIdx.x * WV_MWAVES + wv; unit < n_units; unit += gridDim.x * WV_MWAVES) {
        // `lane` is made opaque once per unit: without this LLVM hoists every lane-derived address
        // and shuffle index of the body out of the persistent loop and holds them in registers
        // for the kernel's whole lifetime (180 VGPRs instead of 128).  (lane0 itself is spilled
        // to scratch and reloaded once per unit, long before its use.  Computing the lane id
        // anew per unit in asm instead -- no range information for the compiler -- ran 4 % slower.)
        int lane = lane0;
        if (!MULTI)
            asm volatile("" : "+v"(lane));
        int rl = unit, part = 0;
        if (n_parts != 1) {
            rl = unit / n_parts;
            part = unit - rl * n_parts;
        }
        const int r = r_base + rl;
        const bool do_pairs = (n_parts == 1) || (part < n_parts - 1);
        const bool do_recip = (n_parts == 1) || (part == n_parts - 1);
        const int j_begin = do_pairs ? min(part * plen, n_mol) : 0;
        const int j_end = do_pairs ? min(j_begin + plen, n_mol) : 0;
        double *const myrec = rec + (int64_t)r * n_mol * MMC_RSTRIDE;

        // n_sub > 1 (the kernel decides, one part per move): this wave takes the replica through n_sub
        // consecutive steps -- commit, evaluate, decide, and again -- and sends one record for all
        // of them (DecideConsts).  What a step wrote and read is in the L2 / the Infinity Cache for
        // the next: the same memory accesses without the arithmetic take 0.72 of the time at four
        // steps per launch (scripts/gather_bw.hip).  Carried from step to step in scalars:
        int flags_carry = -1;            // accept | S-buffer bit << 1 after the step before
        double e_sum = 0.0;              // dU of the accepted steps, in step order
        unsigned acc_mask = 0, ovl_mask = 0, kind_mask = 0;
      for (int sub = 0; sub < n_sub; sub++) {
        if (MULTI) { // (opaque once per STEP here: see above; the steps are a loop inside the unit's)
            lane = lane0;
            asm volatile("" : "+v"(lane));
        }
        const MoveRec *cur_s = cur, *prev_s = prev;
        if (n_sub > 1) {
            const int sl = (slot0 + sub) % dc->ring_slots, slp = (slot0 + sub + dc->ring_slots - 1) % dc->ring_slots;
            cur_s = dc->ring + (int64_t)sl * dc->ring_stride;
            if (sub > 0)
                prev_s = dc->ring + (int64_t)slp * dc->ring_stride;
        }
        WV_STAMP(0); // unit start
        // ---- the move record: one load instruction (lane t holds word t), then scalars ----
        const double *mvp = reinterpret_cast<const double *>(cur_s + r);
        double w = 0.0;
        if (lane < MV_Q_NEW)
            w = mvp[lane];
        WV_STAMP(1); // move record arrived
        const int gflag = flags_carry >= 0 ? flags_carry : (flagv ? __builtin_amdgcn_readfirstlane((int)flagv[r]) : -1);
        const long long hdr = __double_as_longlong(w);
        const int i0 = lane_i32((int)hdr, 0) - 1;
        const int flags = gflag >= 0 ? gflag : lane_i32((int)(hdr >> 32), 0);
        const bool commit = prev_s && (flags & 1);
        const int scur = (flags >> 1) & 1;

        // pending commit of the previous accepted move (main.jl:598-621): written by part 0;
        // every reader of this launch substitutes the pending words for that molecule
        int pend = -1;
        if (commit) {
            const double *pvp = reinterpret_cast<const double *>(prev_s + r);
            double pw = 0.0;
            if (lane < 9) pw = pvp[MV_AT_NEW + lane];
            else if (lane < 12) pw = pvp[MV_COM_NEW + lane - 9];
            else if (lane == 12) pw = pvp[0];
            else if (lane < 17) pw = pvp[MV_Q_NEW + lane - 13];
            pend = lane_i32((int)__double_as_longlong(pw), 12) - 1;
            if (lane < 12)
                sm.pvw[wv][lane] = pw;
            if (part == 0 && lane < 12) {
                myrec[(int64_t)pend * MMC_RSTRIDE + lane] = pw;
                if (lane < 9) {
                    const int a = lane / 3, d = lane % 3;
                    (d == 0 ? bv.ax : d == 1 ? bv.ay : bv.az)[r * bv.atom_stride + 3 * pend + a] = pw;
                } else {
                    const int d = lane - 9;
                    (d == 0 ? bv.comx : d == 1 ? bv.comy : bv.comz)[r * bv.mol_stride + pend] = pw;
                    comq_store(bv, r, pend, d, pw, inv_box);
                }
            }
            if (bv.quat) { // totProps.quat[i] = ei (main.jl:619)
                const double q0 = wave_pick(pw, 13), q1 = wave_pick(pw, 14),
                             q2 = wave_pick(pw, 15), q3 = wave_pick(pw, 16);
                if (part == 0 && lane >= 13 && lane < 17)
                    quat_commit(bv, r, pend, lane - 13, pw, q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3 > 0.25);
            }
            wave_sync();
        }

        // (expanded where they are used: as variables they would be live across the whole unit)
#define WV_CQ_BASE (bv.comq + (int64_t)r * 3 * bv.cq_stride)
#define WV_PART_DST (out + (int64_t)r * n_parts + part)
        // the decision, and with several steps per launch the bookkeeping between them: the last
        // step replaces the seven sums in outw by the launch's record (mmc_kernels.hpp)
        auto decide_here = [&](bool o0, bool o1) -> int {
            if (!dc)
                return 0;
            if (n_sub <= 1)
                return wave_decide(sm.outw[wv], o0, o1, dc, dec_step, r, scur, lane);
            double delta;
            const bool last = sub + 1 == n_sub;
            const int acc = wave_decide(sm.outw[wv], o0, o1, dc, dec_step + sub, r, scur, lane, &delta, last);
            if (acc) {
                e_sum += delta;
                acc_mask |= 1u << sub;
            }
            if (o0 || o1)
                ovl_mask |= 1u << sub;
            kind_mask |= (unsigned)((lane_i32((int)(hdr >> 32), 0) >> 8) & 1) << sub; // (k_propose's note in the record)
            flags_carry = acc | ((scur ^ acc) << 1);
            if (last) {
                wave_sync();
                if (lane < 7)
                    sm.outw[wv][lane] = lane == 0 ? e_sum
                                 : lane == 1 ? __longlong_as_double((long long)(acc_mask | (ovl_mask << 16)) | ((long long)kind_mask << 32)) : 0.0;
                wave_sync();
            }
            return acc;
        };
#define WV_DECIDE(o0, o1) decide_here(o0, o1)
#define WV_RECORD_OVL(o) (n_sub > 1 ? false : (o))
#define WV_STORE_IF (sub + 1 == n_sub)
// (a launch of several steps is long -- 1.7 ms -- and the driver times the next group's launch by this
// launch's records: they must reach the host as the units end, not with the kernel's last write-back)
#define WV_STORE_SYSTEM MULTI
#define WV_ZERO opaque_f64(0.0)
#define WV_SUBST SUBST
#define WV_IMG IMG
#ifndef WV_XY_MOVE
#define WV_XY_MOVE 1
#endif
#define WV_XY WV_XY_MOVE
#define WV_PASSES 2 // (105-111 VGPRs unconstrained instead of 137-151: what makes WV_OCC = 5 possible)
    // (n_parts == 1: every unit has the reciprocal part, and the commit's stores, issued before the
    // phase tables were computed, have mostly landed when those are done)
#define WV_AFTER_PHASE_TABLES                                                                    \
    if (!SUBST && commit)                                                                        \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* this wave's commit is in memory */
#include "mmc_wave_unit.inc"
#undef WV_AFTER_PHASE_TABLES
#undef WV_IMG
#undef WV_PASSES
#undef WV_XY
#undef WV_SUBST
#undef WV_ZERO
#undef WV_CQ_BASE
#undef WV_PART_DST
#undef WV_DECIDE
#undef WV_RECORD_OVL
#undef WV_STORE_IF
#undef WV_STORE_SYSTEM
        if (n_sub > 1) // what this step stored (S_new, soon the commit) is read by the next
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } // sub
    }
}

// =================================================================================================
// k_total_wave -- the pair part of potential(..., "ewald") (Ewald/energy.jl:972-1001) on the same
// scheme as k_move_eval_wave: a wave per unit, lane per neighbour, the chosen molecule in scalar
// registers, fixed-point prefilter + exact fp64 gate, erfc table.
//
// Every molecule pair is visited once (i < j; see mmc_total.hpp for why that equals the
// reference's "twice, then halve" up to summation order, and how the overlap sentinel is kept).
// Unit u of a replica takes molecules u and n_mol - 1 - u: molecule i scans j > i, so pairing the
// two ends gives every unit the same scan length.  Per replica ceil(n_mol / 2) partial sums are
// written; k_total_reduce adds them in index order.
//
// Against k_total_pairs (64 x 64 tiles): 2.7 k VALU instructions in each of 4 waves per tile pair
// and 713 KB of HBM traffic per evaluation (every tile re-read by 12 workgroups) become ~1.3 k per
// wave-unit and one gather per neighbour (profiles/README.md, round 2).
struct TotalWaveShared {
    alignas(16) double qtab[MMC_QQ_TABLE_DOUBLES];
    int32_t list[WV_WAVES][WV_LIST];
};

#define WV_TOTAL_WAVES_PER_SIMD 5 // 87 VGPRs: the single-state pair loop needs fewer than the move kernel
// The body: workgroup `wg` of `n_wgs` takes units wg * WV_WAVES + wave, + n_wgs * WV_WAVES, ...
// (k_total_wave: the whole grid; k_potential_one: the pair workgroups of its grid).
// COHERENT: the unit partials are written with device-scope stores (they bypass the non-coherent
// part of the XCD's L2), for a reader in the SAME launch on another XCD (k_potential_one's last
// workgroup) -- a release fence instead would write back the whole L2 once per workgroup.
// IMG: the minimum image of an atom pair from its molecule's (as WV_IMG in mmc_wave_unit.inc: bit
// for bit vector1D where gate + 2 r_mol < box / 2, which the host checks).
template <bool COHERENT, bool IMG>
__device__ __forceinline__ void total_wave_body(
    TotalWaveShared &sm, const BatchView &bv, const double *__restrict__ rec,
    const double *__restrict__ qq_tab, const FastConsts &fc, const PairParams &pp, TotalPart *out,
    int units_per_rep, int n_units, int paired, int wg, int n_wgs, int j_chunk = 0)
{   // j_chunk > 0 (one system, latency): unit u = (molecule u / n_ch, chunk u % n_ch) scans only the
    // j > i of molecules [chunk * j_chunk, (chunk + 1) * j_chunk) -- many short units instead of one
    // wave walking a whole row; units_per_rep then counts those units
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int k = tid; k < MMC_QQ_TABLE_DOUBLES; k += WV_WAVES * 64)
        sm.qtab[k] = qq_tab[k];
    __syncthreads();

    const int n_mol = bv.n_mol;
    const double box = bv.box;
    const BoxConsts bc = box_consts(box);
    const bool same_gate = pp.lj_gate_sq == pp.qq_gate_sq;
    const double inv_box = 1.0 / box;
    const uint32_t gate_q = com_quant_gate(fmax(pp.lj_gate_sq, pp.qq_gate_sq), box);
    int32_t *const list = sm.list[wv];

    for (int unit = wg * WV_WAVES + wv; unit < n_units; unit += n_wgs * WV_WAVES) {
        int lane = lane0;
        asm volatile("" : "+v"(lane)); // see k_move_eval_wave
        const int r = unit / units_per_rep, u = unit - r * units_per_rep;
        const double *myrec = rec + (int64_t)r * n_mol * MMC_RSTRIDE;
        const uint16_t *cq0 = bv.comq + (int64_t)r * 3 * bv.cq_stride;
        const uint32_t *sxy = reinterpret_cast<const uint32_t *>(cq0);
        const uint16_t *sz = cq0 + 2 * bv.cq_stride;
        double a_lj = 0, a_v = 0, a_q = 0;
        unsigned long long ovm = 0; // lanes that saw an overlap

        // The unit's two molecules as lane-distributed records (lane t = word t); the neighbours
        // of both go into ONE list, tagged with the molecule they belong to, so that the pair
        // loop runs on full batches whatever the split between the two (molecule u has nearly all
        // of its neighbours above it, molecule n_mol - 1 - u nearly none).
        // (paired == 0: one molecule per unit -- twice the waves with uneven work, for the latency
        // of a single system's evaluation)
        const int n_ch = j_chunk > 0 ? (n_mol + j_chunk - 1) / j_chunk : 1;
        const int j_lo = j_chunk > 0 ? (u % n_ch) * j_chunk : 0;
        const int j_hi = j_chunk > 0 ? min(j_lo + j_chunk, n_mol) : n_mol;
        const int iA = j_chunk > 0 ? u / n_ch : u, iB = n_mol - 1 - u;
        const bool hasB = paired && iB > iA; // the middle molecule of an odd count stands alone
        // lane t < 12: word t of molecule A's record; lane 16 + t: word t of molecule B's
        double wAB = 0.0;
        if (lane < MMC_REC)
            wAB = myrec[(int64_t)iA * MMC_RSTRIDE + lane];
        else if (hasB && lane >= 16 && lane < 16 + MMC_REC)
            wAB = myrec[(int64_t)iB * MMC_RSTRIDE + lane - 16];

        auto process = [&](int cnt) {
            wave_sync();
            for (int n0 = 0; n0 < cnt; n0 += 64) {
                const int n = n0 + lane;
                const bool act = n < cnt;
                const int ent = act ? list[n] : 0;
                const int j = ent & ((1 << 27) - 1);
                const bool isB = (ent >> 27) != 0;
                double t[MMC_REC];
                const double2 *src = reinterpret_cast<const double2 *>(myrec + (int64_t)j * MMC_RSTRIDE);
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const double2 v = src[q];
                    t[2 * q] = v.x;
                    t[2 * q + 1] = v.y;
                }
                // this lane's own molecule: its words sit in lanes 0.. (A) or 16.. (B) of wAB -- one
                // ds_bpermute pair per word with a per-lane source (two v_readlane pairs and a
                // 64-bit select were 6 vector instructions per word, 72 per round of 64 neighbours)
                const int own0 = isB ? 16 : 0;
                auto mine = [&](int word) { return wave_pick(wAB, own0 + word); };
                const double ccx = mine(9), ccy = mine(10), ccz = mine(11);
                // the gate, exactly (energy.jl:248-254, ewalds.jl:334-340)
                double m[3] = { 0, 0, 0 }; // (IMG) the image of this neighbour's molecule: 0 or +-1 per axis
                auto min1 = [&](int d, double a, double b) {
                    if constexpr (IMG)
                        return fma(m[d], bc.neg, b - a); // == vector1D(a, b, bc), signed
                    else
                        return vector1D_abs(a, b, bc);
                };
                if constexpr (IMG) {
                    const double dx = t[9] - ccx, dy = t[10] - ccy, dz = t[11] - ccz;
                    m[0] = (fabs(dx) < bc.half) ? 0.0 : copysign(1.0, dx);
                    m[1] = (fabs(dy) < bc.half) ? 0.0 : copysign(1.0, dy);
                    m[2] = (fabs(dz) < bc.half) ? 0.0 : copysign(1.0, dz);
                }
                const double x0 = min1(0, ccx, t[9]), y0 = min1(1, ccy, t[10]), z0 = min1(2, ccz, t[11]);
                const double c0 = x0 * x0 + y0 * y0 + z0 * z0;
                const bool g0 = act && (c0 < pp.qq_gate_sq);
                const bool l0 = same_gate ? g0 : (act && (c0 < pp.lj_gate_sq));
                const unsigned long long gm0 = wave_ballot(g0);
                auto pair_ab = [&](int ab, double ax, double ay, double az, double bx, double by,
                                   double bz) {
                    const double qq = fc.qq9[ab];
                    const bool qneg = (fc.qneg_mask >> ab) & 1; // uniform
                    const double px = min1(0, ax, bx), py = min1(1, ay, by), pz = min1(2, az, bz);
                    const double u0 = px * px + py * py + pz * pz;
                    // scalar mask arithmetic and the table's row of zeros, as in mmc_wave_unit.inc
                    // (ewalds.jl:359, :362)
                    const unsigned long long cm0 = wave_ballot(u0 < pp.ovr) & gm0;
                    // (IMG also promises (gate + 2 r_mol)^2 < r_cut^2 + 100: no atom pair of a gated
                    // molecule pair can fail the slack tests, as in mmc_wave_unit.inc)
                    const unsigned long long im0 = (IMG ? ~0ULL : wave_ballot(u0 < pp.qq_slack_sq)) & (qneg ? gm0 & ~cm0 : gm0);
                    double e0 = qq_table_eval_lanes(sm.qtab, u0, im0);
                    if (qneg) {
                        ovm |= cm0;
                    } else if (cm0 != 0ULL) {
                        if (((im0 >> lane) & 1) && u0 < MMC_QQ_UMIN) e0 = qq_pair_cold(u0, pp.kappa);
                    }
                    a_q = fma(e0, qq, a_q);
                    const double eps = fc.eps9[ab], sg = fc.sig9[ab];
                    if ((fc.lj_mask >> ab) & 1) { // uniform (energy.jl:270: eps > 0.001)
                        if (l0 && (IMG || u0 < pp.lj_slack_sq)) {
                            const double s2 = sg * sg / u0;
                            const double s6 = s2 * s2 * s2;
                            const double s12 = s6 * s6;
                            const double virab = eps * (2.0 * s12 - s6);
                            const double f0 = (IMG ? px : vector1D(ax, bx, bc)) * virab * s2,
                                         f1 = (IMG ? py : vector1D(ay, by, bc)) * virab * s2,
                                         f2 = (IMG ? pz : vector1D(az, bz, bc)) * virab * s2;
                            a_lj += eps * (s12 - s6);
                            a_v += (IMG ? x0 : vector1D(ccx, t[9], bc)) * f0 + (IMG ? y0 : vector1D(ccy, t[10], bc)) * f1
                                   + (IMG ? z0 : vector1D(ccz, t[11], bc)) * f2;
                        }
                    }
                };
#pragma unroll 1
                for (int a = 0; a < 3; a++) {
                    const double ax = mine(3 * a), ay = mine(3 * a + 1), az = mine(3 * a + 2);
                    pair_ab(3 * a, ax, ay, az, t[0], t[1], t[2]);
                    __builtin_amdgcn_sched_barrier(0);
                    pair_ab(3 * a + 1, ax, ay, az, t[3], t[4], t[5]);
                    __builtin_amdgcn_sched_barrier(0);
                    pair_ab(3 * a + 2, ax, ay, az, t[6], t[7], t[8]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            wave_sync();
        };

        int cnt = 0;
        for (int half = 0; half < (hasB ? 2 : 1); half++) {
            const int i0 = half == 0 ? iA : iB;
            const int woff = half == 0 ? 0 : 16;
            uint32_t cqxy, cqz; // this molecule's codes, x | y << 16 and z
            {
                const int t = lane & 15;
                const int myq = (t >= 9 && t < 12) ? (int)com_quant(wAB, inv_box) : 0;
                // (woff is wave-uniform, but not a constant: the lane index of v_readlane in an SGPR)
                cqxy = (uint32_t)lane_i32(myq, woff + 9) | (uint32_t)lane_i32(myq, woff + 10) << 16;
                cqz = (uint32_t)lane_i32(myq, woff + 11);
            }
            // scan j > i0: 64-molecule blocks from the aligned block that holds i0 + 1, WV_TPF of
            // them per trip with no branch around a load (see mmc_wave_unit.inc).  Addressing costs no
            // vector instruction (scalar block pointers + lane offset + immediate) and nothing is
            // clamped: the code arrays end in MMC_CQ_PAD readable bytes and what lies outside
            // [first, j_hi) is masked by ONE unsigned compare of the lane's offset from `first`
            // (this kernel is bound by vector issue, 0.81 busy: the three compares, two clamps and
            // address arithmetic this replaces were 8 of a block's 25 vector instructions).
            const int first = max(i0 + 1, j_lo);
            int base = first & ~63;
            const char *pxy = reinterpret_cast<const char *>(sxy) + 4 * (int64_t)base;
            const char *pz = reinterpret_cast<const char *>(sz) + 2 * (int64_t)base;
            const uint32_t ul4 = 4u * (uint32_t)lane, ul2 = 2u * (uint32_t)lane;
            uint32_t fxy[WV_TPF], fz[WV_TPF];
#pragma unroll
            for (int b = 0; b < WV_TPF; b++) {
                fxy[b] = *reinterpret_cast<const uint32_t *>(pxy + 256 * b + ul4);
                fz[b] = *reinterpret_cast<const uint16_t *>(pz + 128 * b + ul2);
            }
            while (base < j_hi) {
#pragma unroll
                for (int b = 0; b < WV_TPF; b++) {
                    const int j = base + lane;
                    const uint32_t xy = fxy[b], z = fz[b];
                    fxy[b] = *reinterpret_cast<const uint32_t *>(pxy + 256 * (b + WV_TPF) + ul4);
                    fz[b] = *reinterpret_cast<const uint16_t *>(pz + 128 * (b + WV_TPF) + ul2);
                    // first <= j < j_hi  <=>  (unsigned)(j - first) < (unsigned)(j_hi - first)
                    const bool keep = (com_quant_dist2(xy, z, cqxy, cqz) < gate_q)
                                      && ((uint32_t)(j - first) < (uint32_t)(j_hi - first));
                    const unsigned long long m = wave_ballot(keep);
                    if (keep)
                        list[cnt + lanes_below(m)] = j | (half << 27);
                    cnt += __popcll(m);
                    base += 64;
                }
                pxy += 256 * WV_TPF;
                pz += 128 * WV_TPF;
                if (cnt > WV_LIST - 64 * WV_TPF) { // no room for another trip: empty the list
                    process(cnt);
                    cnt = 0;
                }
            }
        }
        if (cnt)
            process(cnt);
        const double s0 = wave_sum_rows(a_lj), s1 = wave_sum_rows(a_v), s2 = wave_sum_rows(a_q);
        const bool any_ovl = ovm != 0ULL;
        if (lane == 0) {
            TotalPart o;
            o.lj_pot = s0; o.lj_vir = s1; o.qq = s2;
            o.n_ovl = any_ovl ? 1 : 0;
            o._pad = 0;
            if (COHERENT) {
                double *w = reinterpret_cast<double *>(out + unit);
                __hip_atomic_store(w, o.lj_pot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(w + 1, o.lj_vir, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(w + 2, o.qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<long long *>(w + 3), (long long)o.n_ovl,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                out[unit] = o;
            }
        }
    }
}

template <bool IMG>
__global__ __launch_bounds__(WV_WAVES * 64, WV_TOTAL_WAVES_PER_SIMD) void k_total_wave(
    BatchView bv, const double *__restrict__ rec, const double *__restrict__ qq_tab, FastConsts fc,
    PairParams pp, TotalPart *out, int units_per_rep, int n_units, int paired)
{
    __shared__ __align__(16) TotalWaveShared sm;
    total_wave_body<false, IMG>(sm, bv, rec, qq_tab, fc, pp, out, units_per_rep, n_units, paired,
                           (int)blockIdx.x, (int)gridDim.x);
}

// Per replica: the unit partials added in index order by one workgroup (fixed order: bitwise
// reproducible), in the reference's normalisation (sum_i 4 pot_i etc., pairs counted once -> x2,
// which the host halves again; energy.jl:289, :978-980).
__global__ __launch_bounds__(MMC_BLOCK) void k_total_reduce(const TotalPart *parts, int n_parts,
                                                             TotalsRaw *out)
{
    __shared__ double red[4 * MMC_WAVES];
    const int r = blockIdx.x;
    const TotalPart *p = parts + (int64_t)r * n_parts;
    double v[4] = { 0, 0, 0, 0 }, tot[4];
    for (int k = threadIdx.x; k < n_parts; k += MMC_BLOCK) {
        v[0] += p[k].lj_pot; v[1] += p[k].lj_vir; v[2] += p[k].qq; v[3] += (double)p[k].n_ovl;
    }
    block_sum<4>(v, red, tot);
    if (threadIdx.x == 0) {
        TotalsRaw t;
        t.lj_e = 2.0 * (tot[0] * 4);
        t.lj_v = 2.0 * (tot[1] * 24 / 3.0);
        t.qq = 2.0 * tot[2];
        t.n_ovl = tot[3] > 0 ? 1 : 0;
        t._pad = 0;
        out[r] = t;
    }
}

// =================================================================================================
// k_move_server_wave -- the wave-per-unit scheme as a PERSISTENT kernel for small replica counts
// (BASELINE configs[1]: one chain, configs[2]: 32 chains per GPU), where a step is latency: launch
// + kernel + result poll = 17 us with a launch per step.  Here the kernel is launched once per run,
// ONE WORKGROUP PER REPLICA: wave w < n_parts - 1 owns pair part w, the last wave the reciprocal
// part (the part split of k_move_eval_wave, so a chain is bit-identical to the launch-per-step
// driver with the same n_parts).  Per step
//   1. every wave draws the proposal of the step itself (the generator of k_propose),
//      speculatively while the host is still deciding the previous step (legal when n_mol > 1:
//      step s moves molecule s mod n_mol, which the pending decision does not touch);
//   2. wave 0 polls the replica's 8-byte control word in pinned host memory (sequence number,
//      launch stamp, accept bit of the previous step, S-buffer bit, "new step sizes", "quit"),
//      applies the accepted move to the batch's arrays -- no other workgroup ever touches this
//      replica's state, and the waves of a workgroup share one L1, so a workgroup barrier is the
//      whole visibility protocol -- and publishes the word in LDS;
//   3. the waves evaluate their parts (mmc_wave_unit.inc, the body of k_move_eval_wave) into LDS;
//      wave 0 adds the parts in the order mmc_combine_parts does and stores ONE 64-byte record
//      with stamp and checksum: the host consumes it exactly as the single part of a launch.
// Accept/reject stays on the host.  Every wait is bounded: when wave 0 sees no new control word
// for SRV_TIMEOUT_TICKS (3 s of the 100 MHz real-time counter) it raises *timeout_flag and the
// whole workgroup exits, so the grid drains whatever happens to the host.  Workgroups never wait
// for each other, so residency is not a correctness condition (one workgroup of 8 waves per
// compute unit fits; the kernel is built for 2 waves per SIMD).
#define SRV_WAVES 8
struct ServerArgs {
    const unsigned long long *ctrl; // pinned host [R]: seq << 40 | (stamp & MMC_STAMP_MASK) << 8 | flags
    const double2 *steps;           // pinned host [R]: {dr_max, dphi_max}
    uint64_t seed, replica0;
    int64_t rng_off;
    int32_t *timeout_flag;          // pinned host: [0] raised, [1..4] who waited for what
    uint64_t seq_off;               // added to the step's sequence number on both sides (test hook:
                                    // start a run just below the 2^24 wrap of the control word)
};
#define SRV_ACCEPT 1u
#define SRV_SCUR 2u
#define SRV_STEPS 4u
#define SRV_QUIT 8u
#define SRV_TIMEOUT_TICKS 300000000ULL
#define SRV_GAVE_UP (~0ULL)

// grid = replicas, block = 64 * n_parts threads (2 <= n_parts <= SRV_WAVES)
__global__ __launch_bounds__(SRV_WAVES * 64, 2) void k_move_server_wave(
    BatchView bv, double *rec, const double *__restrict__ qq_tab,
    const int32_t *__restrict__ kpack, FastConsts fc, PartOut *out, int n_parts, PairParams pp,
    ServerArgs sa)
{
    __shared__ __align__(16) WaveSharedT<SRV_WAVES> sm;
    __shared__ __align__(16) double mvw_all[SRV_WAVES][32];
    __shared__ __align__(16) double comb[8];
    __shared__ unsigned long long ctl;
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int k = tid; k < MMC_QQ_TABLE_DOUBLES; k += blockDim.x)
        sm.qtab[k] = qq_tab[k];
    __syncthreads();

    const int n_mol = bv.n_mol, nkv = bv.nkvecs;
    const double box = bv.box;
    const BoxConsts bc = box_consts(box);
    const int np = n_parts - 1;
    const int plen = (n_mol + np - 1) / np;
    const bool same_gate = pp.lj_gate_sq == pp.qq_gate_sq;
    const double inv_box = 1.0 / box;
    const uint32_t gate_q = com_quant_gate(fmax(pp.lj_gate_sq, pp.qq_gate_sq), box);
    wv_list_t *const list = sm.list[wv];
    const double *const pvw = sm.pvw[wv];
    double *const mvw = mvw_all[wv];

    const int r = blockIdx.x, part = wv;
    const int unit = r * n_parts + part;
    (void)unit;
    const bool do_pairs = part < n_parts - 1;
    const bool do_recip = part == n_parts - 1;
    const int j_begin = do_pairs ? min(part * plen, n_mol) : 0;
    const int j_end = do_pairs ? min(j_begin + plen, n_mol) : 0;
    double *const myrec = rec + (int64_t)r * n_mol * MMC_RSTRIDE;
    const uint16_t *const cq_base = bv.comq + (int64_t)r * 3 * bv.cq_stride;
    PartOut *const part_dst = out + r;
    // host memory that changes while the kernel runs: system-scope loads (a plain load may be
    // served from a cache line fetched at the start of the run)
    auto load_steps = [&]() {
        const double *p = reinterpret_cast<const double *>(sa.steps + r);
        return make_double2(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM),
                            __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    };
    double2 sz = load_steps();
    const ChainKey ck{ sa.seed, (uint32_t)(sa.replica0 + (uint64_t)r) };

    // the move record of step s as a lane-distributed register (lane t = word t), drawn from the
    // chosen molecule's record: k_propose's rigid generator
    // (the molecule of a step is step mod n_mol, main.jl:490: kept as running 32-bit counters -- a
    // 64-bit modulo is ~150 scalar instructions, and there were three of them per step)
    auto make_proposal = [&](int64_t s, int i0p) {
        double cur = 0.0;
        if (lane < MMC_REC)
            cur = myrec[(int64_t)i0p * MMC_RSTRIDE + lane];
        double com[3], at[9];
#pragma unroll
        for (int q = 0; q < 9; q++) at[q] = lane_f64(cur, q);
#pragma unroll
        for (int q = 0; q < 3; q++) com[q] = lane_f64(cur, 9 + q);
        const MoveXform x = propose_xform(ck, (uint64_t)(sa.rng_off + s), box, sz.x, sz.y, com);
        if (lane == 0) {
            MoveRec m;
            m.mol = i0p + 1;
            m.flags = 0;
#pragma unroll
            for (int q = 0; q < 3; q++) { m.com_new[q] = x.com_new[q]; m.com_old[q] = com[q]; }
#pragma unroll
            for (int q = 0; q < 9; q++) m.atoms_old[q] = at[q];
#pragma unroll
            for (int a = 0; a < 3; a++)
                apply_xform(x, com, &at[3 * a], &m.atoms_new[3 * a]);
#pragma unroll
            for (int q = 0; q < 4; q++) m.q_new[q] = 0.0;
            const double *mw = reinterpret_cast<const double *>(&m);
#pragma unroll
            for (int q = 0; q < MV_WORDS; q++) mvw[q] = mw[q];
        }
        wave_sync();
        double wnew = 0.0;
        if (lane < MV_WORDS)
            wnew = mvw[lane];
        wave_sync();
        return wnew;
    };

    int mol_cur = 0; // step mod n_mol
    double w = make_proposal(0, 0), pw = 0.0;
#ifdef SRV_PROFILE
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, p0, p1, p2, p3, p4, p5;
#define SRV_T(x) x = __builtin_amdgcn_s_memrealtime()
#else
#define SRV_T(x)
#endif
    for (int64_t step = 0;; step++) {
        asm volatile("" : "+v"(lane)); // as in k_move_eval_wave: keep lane-derived values out of LICM
        SRV_T(p0);
        // ---- wave 0: wait for the host's word of this step (bounded), commit, publish ----
        if (wv == 0) {
            unsigned long long c = 0;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned spins = 1;; spins++) {
                unsigned long long v = 0;
                if (lane == 0)
                    v = __hip_atomic_load(sa.ctrl + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                c = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32)
                    | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
                // (the word carries the sequence number modulo 2^24: compare modulo 2^24)
                if ((c >> 40) == (((unsigned long long)(step + 1) + sa.seq_off) & 0xffffffULL) || (c & SRV_QUIT))
                    break; // (a host that gives up early posts "quit" with whatever sequence number)
                if ((spins & 63) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > SRV_TIMEOUT_TICKS) {
                    if (lane == 0) { // what this workgroup was waiting for, for the host's message
                        sa.timeout_flag[1] = r;
                        sa.timeout_flag[2] = (int32_t)step;
                        sa.timeout_flag[3] = (int32_t)(c >> 40);
                        sa.timeout_flag[4] = (int32_t)c;
                        __hip_atomic_store(sa.timeout_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    c = SRV_GAVE_UP;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            SRV_T(p1);
            // the previous step was accepted: commit it (main.jl:598-621)
            if (c != SRV_GAVE_UP && (c & SRV_ACCEPT) && step > 0) {
                const int pm = mol_cur == 0 ? n_mol - 1 : mol_cur - 1; // (step - 1) mod n_mol
                int word = -1; // record word this lane's piece of the previous proposal goes to
                if (lane >= MV_AT_NEW && lane < MV_AT_NEW + 9) word = lane - MV_AT_NEW;
                else if (lane >= MV_COM_NEW && lane < MV_COM_NEW + 3) word = 9 + lane - MV_COM_NEW;
                if (word >= 0) {
                    myrec[(int64_t)pm * MMC_RSTRIDE + word] = pw;
                    if (word < 9) {
                        const int a = word / 3, d = word % 3;
                        (d == 0 ? bv.ax : d == 1 ? bv.ay : bv.az)[r * bv.atom_stride + 3 * pm + a] = pw;
                    } else {
                        const int d = word - 9;
                        (d == 0 ? bv.comx : d == 1 ? bv.comy : bv.comz)[r * bv.mol_stride + pm] = pw;
                        comq_store(bv, r, pm, d, pw);
                    }
                }
            }
            if (lane == 0)
                ctl = c;
        }
        __syncthreads(); // the word is published, wave 0's commit stores are complete
        const unsigned long long c =
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ctl >> 32)) << 32)
            | (unsigned)__builtin_amdgcn_readfirstlane((int)ctl);
        if (c == SRV_GAVE_UP)
            return;
        const unsigned flags = (unsigned)(c & 0xffu);
        const unsigned stamp = (unsigned)((c >> 8) & MMC_STAMP_MASK);
        SRV_T(p2);
#ifdef SRV_PROFILE
        if ((flags & SRV_QUIT) && lane == 0 && r == 0)
            sa.timeout_flag[8 + wv] = (int32_t)pt[2];
#endif
        if (flags & SRV_QUIT)
            break;
        if (flags & SRV_STEPS)
            sz = load_steps();
        if (n_mol == 1 || (flags & SRV_STEPS)) // the speculative proposal is out of date
            w = make_proposal(step, mol_cur);
        const int i0 = mol_cur;
        const int scur = (flags & SRV_SCUR) ? 1 : 0;
        const int pend = -1;
#define WV_UNIT_NO_STORE
#define WV_CQ_BASE cq_base
#include "mmc_wave_unit.inc"
#undef WV_CQ_BASE
#undef WV_UNIT_NO_STORE
        SRV_T(p3);
        __syncthreads(); // every part's sums are in sm.outw
        SRV_T(p4);
        if (wv == 0) { // mmc_combine_parts' order: ((0 + part 0) + part 1) + ...
            double v = 0.0;
            long long ob = 0;
            for (int q = 0; q < n_parts; q++) {
                if (lane < 7)
                    v += sm.outw[q][lane];
                ob |= __double_as_longlong(sm.outw[q][7]);
            }
            if (lane < 7)
                comb[lane] = v;
            wave_sync();
            if (lane == 0)
                comb[7] = pack_ovl((int)(ob & 1), (int)((ob >> 1) & 1), stamp, part_checksum(comb, stamp));
            wave_sync();
            store_part<true>(part_dst, comb, lane);
        }
        SRV_T(p5);
        pw = w;
        if (n_mol > 1)
            w = make_proposal(step + 1, mol_cur + 1 == n_mol ? 0 : mol_cur + 1); // while the host decides this step
#ifdef SRV_PROFILE
        pt[0] += p1 - p0; pt[1] += p2 - p1; pt[2] += p3 - p2; pt[3] += p4 - p3; pt[4] += p5 - p4;
        pt[5] += __builtin_amdgcn_s_memrealtime() - p5;
#endif
        mol_cur = mol_cur + 1 == n_mol ? 0 : mol_cur + 1;
    }
}
