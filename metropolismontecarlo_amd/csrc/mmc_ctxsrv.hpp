// mmc_ctxsrv.hpp -- k_ctx_server: the per-call surface of ONE system (mmc_ctx) served by a
// persistent kernel.
//
// Why.  Loop() (Ewald/main.jl:487-644) calls LJ_poly_dU, EwaldShort (x2 each) and RecipMove once per
// trial move, synchronously, from a host that draws the move in between.  With a kernel launch and a
// stream synchronisation per call a call costs 18-20 us on this machine whatever the kernel does
// (round 2: 210 us per Loop() body through the Python mirror, against 97 us for the CPU port).  The
// move server of the replica batch (k_move_server_wave) showed that a resident kernel answers in
// well under half of that.  This is the same scheme behind the context API:
//
//   * the host posts a COMMAND BLOCK in pinned memory (which molecule, its coordinates as the
//     caller's arrays hold them, up to CS_MAX_PEND molecules whose device copy is stale, which
//     terms are wanted): eight 64-byte lines, each closed by a tag that repeats the sequence
//     number, the head word written last;
//   * every workgroup of the server reads the whole block with one load per lane (system scope)
//     until head and tags carry the number it waits for, writes the stale molecules' records
//     ITSELF -- each workgroup's stores go through its own XCD's L2, so no cross-workgroup
//     visibility protocol is needed; they all write the same bytes -- and evaluates its share of
//     the parts: k_ctx_server_lat with the latency body (mmc_wave_lat.inc: resident molecule
//     ranges, one 64-byte record per workgroup), k_ctx_server with the unit body of
//     k_move_eval_wave (one record per wave) where a range does not fit a wave's storage;
//   * the records (stamp + checksum, write-through) land in pinned memory; the host combines them
//     in index order.
// One command evaluates LJ and real-space Coulomb TOGETHER (the host caches the pair for the
// second of LJ_poly_dU / EwaldShort), for one state or for two (mmc_trial_move), and -- when the
// molecule's old coordinates are known -- RecipMove as well, speculatively: S_new goes to a free
// structure-factor buffer and to its pinned host mirror, so that the RecipMove call that follows is
// a comparison of arrays on the host.  A second set of workgroups (look-ahead) evaluates, with the
// command of a moved molecule i, molecule i + 1 of Loop()'s sweep as the system then stands: if the
// move is kept, the next LJ_poly_dU / EwaldShort pair is answered without a command.
//
// Every wait is bounded: a workgroup that sees no command for CS_IDLE_TICKS raises its flag and
// exits; the host never posts to a server it has not heard from for a third of that time without
// relaunching it, and relaunches (replaying the idempotent command) whenever a reply is late.
#pragma once
#include "mmc_wave.hpp"
#include "mmc_lat.hpp"

#define CS_MAX_PEND 2
#define CS_EVAL 1u   // pair part: LJ + real-space Coulomb of molecule `mol`
#define CS_TWO 2u    // ... in both states of the record (old and new); else the new slot only
#define CS_RECIP 4u  // reciprocal part: S[s_dst] = S[s_base] + dS(old -> new), dE against S[s_base]
#define CS_QUIT 8u
#define CS_NEXT 16u  // the look-ahead workgroups evaluate molecule `mol2` (one state, its device record) as well
// record slots of k_ctx_server_lat with G workgroups per role: results out[g], the look-ahead
// workgroups' out[G + g], the S-mirror sums of workgroups with reciprocal waves out[2 G + g]
#define CS_OUT_AHEAD(G, g) ((G) + (g))
#define CS_OUT_SSUM(G, g) (2 * (G) + (g))
#define CS_OUT_RECORDS (3 * MMC_CTX_MAX_WGS)
#define MMC_CTX_MAX_WGS 32
#define CS_IDLE_TICKS 100000000ULL // 1 s of the 100 MHz real-time counter

// The command block: 64 words of 8 bytes = eight 64-byte lines; the LAST word of every line is a
// TAG that repeats the sequence number.  The host writes the data words, then the tags, then the
// head (its stores become visible in that order); a server workgroup reads the whole block with
// ONE load per lane and accepts it when the head and all eight tags carry the number it waits for
// -- a line whose tag is current holds current data -- so a command costs one PCIe read round
// trip, not a poll plus a fetch.  Logical word L sits at physical word CS_PHYS(L):
//   [0]      head: seq << 16 | flags
//   [1]      lo: mol (0-based)   hi: n_pend | mol2 << 4   (mol2: CS_NEXT)
//   [2]      lo: s_base | s_dst << 8              hi: launch stamp of the results
//   [3..27]  the first 25 words of a MoveRec: mol + 1, com_new, atoms_new, com_old, atoms_old
//   [28]     lo: pend_mol[0]  hi: pend_mol[1]    (0-based)
//   [29..40] pend_rec[0]: atoms (9) + com (3)
//   [41..52] pend_rec[1]
#define CS_PHYS(L) ((L) + (L) / 7)
#define CS_W_HEAD 0
#define CS_W_MOL 1
#define CS_W_SBUF 2
#define CS_W_MV 3
#define CS_W_PMOL 28
#define CS_W_PREC 29
#define CS_WORDS_LOGICAL (CS_W_PREC + 12 * CS_MAX_PEND)
static_assert(CS_PHYS(CS_WORDS_LOGICAL - 1) < 63, "the command block is eight lines");
static_assert(MV_Q_NEW == 25, "the block carries a MoveRec up to q_new");

struct CtxSrvArgs {
    const unsigned long long *cmd; // pinned host, 64 words
    double *s_mirror;              // pinned host [4][2 * nk_stride]: copies of the S buffers the server writes
    int32_t *state;                // pinned host: [0] a workgroup gave up, [1 + g] workgroup g exited at seq
    unsigned long long seq0;       // first sequence number this launch answers
};

// 64-bit sum over the lanes (integer; the order does not matter)
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned lo = (unsigned)__shfl_down((int)(unsigned)v, off, 64);
        const unsigned hi = (unsigned)__shfl_down((int)(unsigned)(v >> 32), off, 64);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

// grid = G workgroups of wpg waves; n_parts = G * wpg; part = g * wpg + wave.
__global__ __launch_bounds__(SRV_WAVES * 64, 2) void k_ctx_server(
    BatchView bv, double *rec, const double *__restrict__ qq_tab,
    const int32_t *__restrict__ kpack, FastConsts fc, PartOut *out, int n_parts, PairParams pp,
    CtxSrvArgs sa)
{
    __shared__ __align__(16) WaveSharedT<SRV_WAVES> sm;
    __shared__ __align__(16) unsigned long long cmdw[2][64];
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

    const int r = 0;
    const int part = blockIdx.x * (blockDim.x >> 6) + wv;
    const int unit = part;
    (void)unit;
    const int j_begin = part < np ? min(part * plen, n_mol) : 0;
    const int j_end = part < np ? min(j_begin + plen, n_mol) : 0;
    double *const myrec = rec;
    const uint16_t *const cq_base = bv.comq;
    PartOut *const part_dst = out + part;

#ifdef CS_PROFILE // diagnostic build: 10 ns ticks per phase, summed over the commands (state[16 + 8 wave + k])
    unsigned long long cp[6] = { 0, 0, 0, 0, 0, 0 }, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#define CS_T(x) x = __builtin_amdgcn_s_memrealtime()
#else
#define CS_T(x)
#endif
    for (unsigned long long seq = sa.seq0;; seq++) {
        asm volatile("" : "+v"(lane)); // keep lane-derived values out of LICM (see k_move_eval_wave)
        CS_T(c0);
        unsigned long long *const cw = cmdw[seq & 1];
        // ---- wave 0: wait for the command (bounded), copy the block to LDS, refresh stale molecules ----
        if (wv == 0) {
            unsigned long long v = 0, head = 0;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool gave_up = false;
            for (unsigned spins = 1;; spins++) {
                v = __hip_atomic_load(sa.cmd + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                head = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32)
                       | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
                if (head & CS_QUIT)
                    break;
                const bool mine_ok = (lane & 7) != 7 || v == seq; // tags: the sequence number
                if ((head >> 16) == seq && wave_ballot(mine_ok) == ~0ULL)
                    break;
                if ((spins & 63) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > CS_IDLE_TICKS) {
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (gave_up || (head & CS_QUIT)) {
                if (lane == 0) {
                    __hip_atomic_store(sa.state + 1 + blockIdx.x, gave_up ? 1 : 2, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
                    if (gave_up)
                        __hip_atomic_store(sa.state, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    cw[CS_W_HEAD] = CS_QUIT;
                }
            } else {
                cw[lane] = v;
                wave_sync();
                // molecules whose device copy is stale: record, SoA arrays, fixed-point COM
                const int n_pend = __builtin_amdgcn_readfirstlane((int)(cw[CS_PHYS(CS_W_MOL)] >> 32));
                const unsigned long long pmw = cw[CS_PHYS(CS_W_PMOL)];
                for (int p = 0; p < n_pend && p < CS_MAX_PEND; p++) {
                    const int pm = __builtin_amdgcn_readfirstlane(p == 0 ? (int)pmw : (int)(pmw >> 32));
                    if (lane < 12) {
                        const int word = lane;
                        const double pw = __longlong_as_double((long long)cw[CS_PHYS(CS_W_PREC + 12 * p + word)]);
                        myrec[(int64_t)pm * MMC_RSTRIDE + word] = pw;
                        if (word < 9) {
                            const int a = word / 3, d = word % 3;
                            (d == 0 ? bv.ax : d == 1 ? bv.ay : bv.az)[3 * pm + a] = pw;
                        } else {
                            const int d = word - 9;
                            (d == 0 ? bv.comx : d == 1 ? bv.comy : bv.comz)[pm] = pw;
                            comq_store(bv, 0, pm, d, pw);
                        }
                    }
                }
            }
        }
        CS_T(c1);
        __syncthreads(); // the block is in LDS, wave 0's stores are complete
        CS_T(c2);
        const unsigned long long head = cw[CS_W_HEAD];
        const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane((int)head) & 0xffffu;
#ifdef CS_PROFILE
        if ((flags & CS_QUIT) && lane == 0 && blockIdx.x == 0)
            for (int q = 0; q < 6; q++)
                sa.state[16 + 8 * wv + q] = (int32_t)cp[q];
#endif
        if (flags & CS_QUIT)
            return;
        const unsigned long long wmol = cw[CS_PHYS(CS_W_MOL)], wsb = cw[CS_PHYS(CS_W_SBUF)];
        const int i0 = __builtin_amdgcn_readfirstlane((int)wmol);
        const int sbits = __builtin_amdgcn_readfirstlane((int)wsb);
        const unsigned stamp = (unsigned)__builtin_amdgcn_readfirstlane((int)(wsb >> 32)) & MMC_STAMP_MASK;
        const int s_base = sbits & 0xff, s_dst = (sbits >> 8) & 0xff;
        double w = 0.0; // the move record, lane t = word t
        if (lane < MV_Q_NEW)
            w = __longlong_as_double((long long)cw[CS_PHYS(CS_W_MV + lane)]);
        const bool do_pairs = (part < np) && (flags & CS_EVAL);
        const bool do_recip = (part == np) && (flags & CS_RECIP);
        const int pend = -1, scur = s_base & 1; // (scur: unused, the buffers are named by macros)
        (void)scur;
        double *const s_mirror = sa.s_mirror + (int64_t)s_dst * bv.nk_stride * 2;
#define WV_CQ_BASE cq_base
#define WV_PART_DST part_dst
#define WV_STORE_WT
#define WV_S_BASE s_buf(bv, 0, s_base)
#define WV_S_DST s_buf(bv, 0, s_dst)
#define WV_S_MIRROR s_mirror
        if (flags & CS_TWO) {
#define WV_NS 2
#include "mmc_wave_unit.inc"
#undef WV_NS
        } else {
#define WV_NS 1
#include "mmc_wave_unit.inc"
#undef WV_NS
        }
#ifdef CS_PROFILE
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        CS_T(c3);
        if (flags & CS_EVAL) { cp[0] += c1 - c0; cp[1] += c2 - c1; cp[2] += c3 - c2; cp[3] += 1; }
#endif
#undef WV_S_MIRROR
#undef WV_S_DST
#undef WV_S_BASE
#undef WV_STORE_WT
#undef WV_PART_DST
#undef WV_CQ_BASE
    }
}

// =================================================================================================
// k_ctx_server_lat: the same protocol on the latency unit body (mmc_wave_lat.inc, mmc_lat.hpp):
// G workgroups of four waves, each wave a part with its molecule range resident; one combined
// record per workgroup (out[g]), and -- for commands with a reciprocal part -- the integrity sum of
// the mirrored S_new of the workgroup's reciprocal waves in a second record (out[CS_OUT_SSUM(G, g)]).
// =================================================================================================
__global__ __launch_bounds__(LAT_WAVES * 64, 2) void k_ctx_server_lat(
    BatchView bv, double *rec, const double *__restrict__ qq_tab,
    const int32_t *__restrict__ kpack, FastConsts fc, PartOut *out, int n_parts, PairParams pp,
    CtxSrvArgs sa)
{
    __shared__ __align__(16) LatShared ls;
    __shared__ __align__(16) unsigned long long cmdw[2][64];
    const int tid = threadIdx.x, lane0 = tid & 63;
    int lane = lane0;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int k = tid; k < MMC_QQ_TABLE_DOUBLES; k += LAT_WAVES * 64)
        ls.qtab[k] = qq_tab[k];
    // workgroups [0, G): the command's evaluation; [G, 2 G) (when launched): LOOK-AHEAD -- the same
    // part plan, evaluating the molecule the host expects to be asked about next (CS_NEXT) in the
    // state this command leaves behind.  Their records (out[CS_OUT_AHEAD(G, g)]) are collected by the host
    // when and if that call comes; nobody waits for them.
    const int G = n_parts / LAT_WAVES;
    const int r = 0, role = (int)blockIdx.x / G, g = (int)blockIdx.x - role * G;
    double *const myrec = rec;
    const uint16_t *const cq_base = bv.comq;
    LAT_WAVE_SETUP(g * LAT_WAVES + wv, myrec, cq_base)
    __syncthreads();
    const bool wg_has_recip = role == 0 && (g + 1) * LAT_WAVES > plan.np;

    for (unsigned long long seq = sa.seq0;; seq++) {
        asm volatile("" : "+v"(lane));
        unsigned long long *const cw = cmdw[seq & 1];
        if (wv == 0) { // the command block: one load per lane, head + eight tags (see k_ctx_server)
            unsigned long long v = 0, head = 0;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool gave_up = false;
            for (unsigned spins = 1;; spins++) {
                v = __hip_atomic_load(sa.cmd + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                head = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32)
                       | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
                if (head & CS_QUIT)
                    break;
                const bool mine_ok = (lane & 7) != 7 || v == seq;
                if ((head >> 16) == seq && wave_ballot(mine_ok) == ~0ULL)
                    break;
                if ((spins & 63) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > CS_IDLE_TICKS) {
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (gave_up || (head & CS_QUIT)) {
                if (lane == 0) {
                    __hip_atomic_store(sa.state + 1 + blockIdx.x, gave_up ? 1 : 2, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
                    if (gave_up)
                        __hip_atomic_store(sa.state, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    cw[CS_W_HEAD] = CS_QUIT;
                }
            } else {
                cw[lane] = v;
            }
        }
        __syncthreads(); // the block is in LDS (and every wave is done with the previous command's LDS)
        const unsigned long long head = cw[CS_W_HEAD];
        const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane((int)head) & 0xffffu;
        if (flags & CS_QUIT)
            return;
        // molecules whose device copy is stale: every wave refreshes its resident range, one wave
        // of the workgroup writes the global copies
        {
            const int n_pend = __builtin_amdgcn_readfirstlane((int)(cw[CS_PHYS(CS_W_MOL)] >> 32)) & 15;
            const unsigned long long pmw = cw[CS_PHYS(CS_W_PMOL)];
            for (int p = 0; p < n_pend && p < CS_MAX_PEND; p++) {
                const int pm = __builtin_amdgcn_readfirstlane(p == 0 ? (int)pmw : (int)(pmw >> 32));
                const double val = lane < 12
                    ? __longlong_as_double((long long)cw[CS_PHYS(CS_W_PREC + 12 * p + lane)]) : 0.0;
                LAT_REFRESH(pm, val);
                if (wv == 0)
                    LAT_COMMIT_GLOBAL(0, pm, val, myrec);
            }
        }
        if (role == 1) { // ---- look-ahead: molecule mol2, one state, no reciprocal part ----
            if (!(flags & CS_NEXT))
                continue;
            const int i0 = (int)((unsigned)__builtin_amdgcn_readfirstlane((int)(cw[CS_PHYS(CS_W_MOL)] >> 32)) >> 4);
            const unsigned stamp = (unsigned)__builtin_amdgcn_readfirstlane((int)(cw[CS_PHYS(CS_W_SBUF)] >> 32))
                                   & MMC_STAMP_MASK;
            // its record as the device holds it (this workgroup's own commit stores above came
            // first; the host never names a molecule here whose device copy is stale)
            double cur = 0.0;
            if (lane < MMC_REC)
                cur = myrec[(int64_t)i0 * MMC_RSTRIDE + lane];
            const double w = __shfl(cur, (lane >= MV_AT_NEW && lane < MV_AT_NEW + 9) ? lane - MV_AT_NEW
                                         : (lane >= MV_COM_NEW && lane < MV_COM_NEW + 3) ? 9 + lane - MV_COM_NEW : 0, 64);
            unsigned long long s_sum_out = 0;
            (void)s_sum_out;
            {
                const bool part_pairs = do_pairs;
                const bool do_pairs = part_pairs, do_recip = false;
#define WV_S_BASE s_buf(bv, 0, 0)
#define WV_S_DST s_buf(bv, 0, 0)
#define WV_NS 1
#include "mmc_wave_lat.inc"
#undef WV_NS
#undef WV_S_DST
#undef WV_S_BASE
            }
            __syncthreads();
            if (wv == 0)
                lat_store_combined<true>(ls, out + CS_OUT_AHEAD(G, g), lane, stamp);
            continue;
        }
        const unsigned long long wmol = cw[CS_PHYS(CS_W_MOL)], wsb = cw[CS_PHYS(CS_W_SBUF)];
        const int i0 = __builtin_amdgcn_readfirstlane((int)wmol);
        const int sbits = __builtin_amdgcn_readfirstlane((int)wsb);
        const unsigned stamp = (unsigned)__builtin_amdgcn_readfirstlane((int)(wsb >> 32)) & MMC_STAMP_MASK;
        const int s_base = sbits & 0xff, s_dst = (sbits >> 8) & 0xff;
        double w = 0.0; // the move record, lane t = word t
        if (lane < MV_Q_NEW)
            w = __longlong_as_double((long long)cw[CS_PHYS(CS_W_MV + lane)]);
        double *const s_mirror = sa.s_mirror + (int64_t)s_dst * bv.nk_stride * 2;
        unsigned long long s_sum_out = 0;
        {
            const bool part_pairs = do_pairs, part_recip = do_recip;
            const bool do_pairs = part_pairs && (flags & CS_EVAL);   // (shadow: what THIS command wants)
            const bool do_recip = part_recip && (flags & CS_RECIP);
#define WV_S_BASE s_buf(bv, 0, s_base)
#define WV_S_DST s_buf(bv, 0, s_dst)
#define WV_S_MIRROR s_mirror
            if (flags & CS_TWO) {
#define WV_NS 2
#include "mmc_wave_lat.inc"
#undef WV_NS
            } else {
#define WV_NS 1
#include "mmc_wave_lat.inc"
#undef WV_NS
            }
#undef WV_S_MIRROR
#undef WV_S_DST
#undef WV_S_BASE
        }
        if (lane == 0)
            ls.mvw[wv][0] = __longlong_as_double((long long)s_sum_out);
        __syncthreads(); // every wave's sums are in ls.outw
        if (wv == 0) {
            lat_store_combined<true>(ls, out + g, lane, stamp);
            if (wg_has_recip && (flags & CS_RECIP)) {
                unsigned long long ss = 0;
#pragma unroll
                for (int q = 0; q < LAT_WAVES; q++)
                    ss += (unsigned long long)__double_as_longlong(ls.mvw[q][0]);
                const double word = lane == 0 ? __longlong_as_double((long long)ss) : 0.0;
                const uint32_t csum = part_checksum_lanes(word, lane, stamp);
                if (lane < 7)
                    ls.comb[lane] = word;
                if (lane == 7)
                    ls.comb[7] = pack_ovl(0, 0, stamp, csum);
                wave_sync();
                store_part<true>(out + CS_OUT_SSUM(G, g), ls.comb, lane);
                wave_sync();
            }
        }
    }
}
