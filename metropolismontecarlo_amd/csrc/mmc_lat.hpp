// mmc_lat.hpp -- the latency kernels of the per-move path: the unit body of mmc_wave_lat.inc behind
//   k_move_eval_lat     one launch per step (grid: G workgroups x replicas), and
//   k_move_server_lat   the persistent move server with G WORKGROUPS PER REPLICA.
// Both split a move into n_parts = 4 G parts, one wave each: np pair parts (a fixed molecule range
// of at most LAT_MAXMOL each, resident in the wave's registers and LDS) and nr reciprocal parts (a
// share of the k-iterations each); the four waves of a workgroup are added in wave order into ONE
// 64-byte record per workgroup, the host adds the G records in workgroup order -- the same sums in
// the same order from either kernel, so a chain is bit-identical between them.
//
// Several workgroups per replica need no hand-off between them: every workgroup polls the replica's
// control word itself, draws the (counter-based, hence identical) proposal itself, and applies an
// accepted move to its own copy of its molecules AND to global memory itself -- each workgroup's
// stores go through its own L2, all write the same bytes -- so what one workgroup reads it has
// written.  (Round 2 polled from 13 workgroups per replica x 32 replicas and saturated PCIe reads;
// this form is taken for few replicas only, see batch_lat_shape.)
#pragma once
#include "mmc_wave.hpp"

#define LAT_WAVES 4     // waves per workgroup (256 threads): one per SIMD
#define LAT_MAXMOL 128  // molecules of a pair part: two 64-lane blocks of resident codes
#define LAT_KIT 6       // k-iterations (64 vectors each) a reciprocal part may have: all 337 vectors
#define LAT_MAX_PARTS 128 // waves per replica at most (32 workgroups: 125 pair parts x 128 = 16000 molecules)

struct LatShared {
    alignas(16) double qtab[MMC_QQ_TABLE_DOUBLES];
    alignas(16) double lrec[LAT_WAVES][LAT_MAXMOL * MMC_REC]; // the waves' resident records
    cplx ptab[LAT_WAVES][2][3][3][MMC_NKTAB];
    uint16_t list[LAT_WAVES][LAT_MAXMOL];
    alignas(16) double outw[LAT_WAVES][8];
    alignas(16) double comb[8];
    alignas(16) double mvw[LAT_WAVES][32];
    unsigned long long ctl;
};

// how n_parts = 4 G parts are shared between the pair term and the reciprocal sum
struct LatPlan {
    int n_parts, np, nr, plen, n_it;
};
__host__ __device__ inline LatPlan lat_plan(int n_parts, int n_mol, int nkv)
{
    LatPlan p;
    p.n_parts = n_parts;
    p.nr = n_parts >= 12 ? 3 : n_parts >= 6 ? 2 : 1;
    p.np = n_parts - p.nr;
    p.plen = (n_mol + p.np - 1) / p.np;
    p.n_it = (nkv + 63) >> 6;
    return p;
}
// the latency kernels apply when every pair part's range fits a wave's resident storage
__host__ __device__ inline bool lat_applies(int n_parts, int n_mol, int nkv)
{
    if (n_parts < 4 || n_parts % LAT_WAVES != 0 || n_parts > LAT_MAX_PARTS)
        return false;
    const LatPlan p = lat_plan(n_parts, n_mol, nkv);
    return p.plen <= LAT_MAXMOL && p.n_it <= LAT_KIT && p.np >= 1;
}

// ---- per-wave set-up shared by the kernels: constants of the lane, part ranges, resident state ----
#define LAT_WAVE_SETUP(part_expr, rec_ptr, cq_ptr)                                                 \
    const int n_mol = bv.n_mol, nkv = bv.nkvecs;                                                   \
    const double box = bv.box;                                                                     \
    const BoxConsts bc = box_consts(box);                                                          \
    const bool same_gate = pp.lj_gate_sq == pp.qq_gate_sq;                                         \
    const double inv_box = 1.0 / box;                                                              \
    const uint32_t gate_q = com_quant_gate(fmax(pp.lj_gate_sq, pp.qq_gate_sq), box);               \
    const LatPlan plan = lat_plan(n_parts, n_mol, nkv);                                            \
    const int part = (part_expr);                                                                  \
    const bool do_pairs = part < plan.np;                                                          \
    const bool do_recip = !do_pairs;                                                               \
    const int jb = do_pairs ? min(part * plan.plen, n_mol) : 0;                                    \
    const int jn = do_pairs ? min(plan.plen, n_mol - jb) : 0;                                      \
    const int kit0 = do_recip ? (part - plan.np) * plan.n_it / plan.nr : 0;                        \
    const int kit1 = do_recip ? (part - plan.np + 1) * plan.n_it / plan.nr : 0;                    \
    const int l3 = lane0 / 3, lb = lane0 - 3 * l3;                                                 \
    double qrow[3], erow[3], srow[3];                                                              \
    bool any_eps[3];                                                                               \
    _Pragma("unroll") for (int a = 0; a < 3; a++) {                                                \
        qrow[a] = lb == 0 ? fc.qq9[3 * a] : lb == 1 ? fc.qq9[3 * a + 1] : fc.qq9[3 * a + 2];       \
        erow[a] = lb == 0 ? fc.eps9[3 * a] : lb == 1 ? fc.eps9[3 * a + 1] : fc.eps9[3 * a + 2];    \
        srow[a] = lb == 0 ? fc.sig9[3 * a] : lb == 1 ? fc.sig9[3 * a + 1] : fc.sig9[3 * a + 2];    \
        any_eps[a] = fc.eps9[3 * a] > 0.001 || fc.eps9[3 * a + 1] > 0.001                          \
                     || fc.eps9[3 * a + 2] > 0.001;                                                \
    }                                                                                              \
    double *const lrec = ls.lrec[wv];                                                              \
    uint32_t rq_xy[2] = { 0, 0 }, rq_z[2] = { 0, 0 };                                              \
    {                                                                                              \
        const uint32_t *sxy_ = reinterpret_cast<const uint32_t *>(cq_ptr);                         \
        const uint16_t *sz_ = (cq_ptr) + 2 * bv.cq_stride;                                         \
        _Pragma("unroll") for (int blk = 0; blk < LAT_MAXMOL / 64; blk++) {                        \
            const int jl = blk * 64 + lane0;                                                       \
            if (jl < jn) {                                                                         \
                rq_xy[blk] = sxy_[jb + jl];                                                        \
                rq_z[blk] = sz_[jb + jl];                                                          \
            }                                                                                      \
        }                                                                                          \
        for (int g = lane0; g < jn * 6; g += 64) {                                                 \
            const int m_ = g / 6, q_ = g - 6 * m_;                                                 \
            *reinterpret_cast<double2 *>(lrec + m_ * MMC_REC + 2 * q_) =                           \
                *reinterpret_cast<const double2 *>((rec_ptr) + (int64_t)(jb + m_) * MMC_RSTRIDE + 2 * q_); \
        }                                                                                          \
    }

// A molecule changes (a committed move): the wave refreshes its resident copy if the molecule is in
// its range.  `val`: lane t < 12 holds word t of the new record (atoms 9, com 3).  pm is uniform.
#define LAT_REFRESH(pm, val)                                                                       \
    do {                                                                                           \
        const int jl_ = (pm) - jb;                                                                 \
        if (jl_ >= 0 && jl_ < jn) {                                                                \
            if (lane < 12)                                                                         \
                lrec[jl_ * MMC_REC + lane] = (val);                                                \
            const int myq_ = (lane >= 9 && lane < 12) ? (int)com_quant((val), inv_box) : 0;        \
            const uint32_t nxy_ = (uint32_t)lane_i32(myq_, 9) | (uint32_t)lane_i32(myq_, 10) << 16; \
            const uint32_t nz_ = (uint32_t)lane_i32(myq_, 11);                                     \
            if (lane == (jl_ & 63)) {                                                              \
                if (jl_ < 64) { rq_xy[0] = nxy_; rq_z[0] = nz_; }                                  \
                else { rq_xy[1] = nxy_; rq_z[1] = nz_; }                                           \
            }                                                                                      \
        }                                                                                          \
    } while (0)

// global copies of a committed molecule (record, SoA arrays, fixed-point COM): lane t < 12 = word t
#define LAT_COMMIT_GLOBAL(r, pm, val, recbase)                                                     \
    do {                                                                                           \
        if (lane < 12) {                                                                           \
            (recbase)[(int64_t)(pm) * MMC_RSTRIDE + lane] = (val);                                 \
            if (lane < 9) {                                                                        \
                const int a_ = lane / 3, d_ = lane % 3;                                            \
                (d_ == 0 ? bv.ax : d_ == 1 ? bv.ay : bv.az)[(r) * bv.atom_stride + 3 * (pm) + a_] = (val); \
            } else {                                                                               \
                const int d_ = lane - 9;                                                           \
                (d_ == 0 ? bv.comx : d_ == 1 ? bv.comy : bv.comz)[(r) * bv.mol_stride + (pm)] = (val); \
                comq_store(bv, (r), (pm), d_, (val), inv_box);                                     \
            }                                                                                      \
        }                                                                                          \
    } while (0)

// the four waves of the workgroup -> one record: sums in wave order, overlap bits or-ed, stamp +
// checksum; 4 lanes x 16 B.  Call after a workgroup barrier, by wave 0.
template <bool WRITE_THROUGH>
__device__ __forceinline__ void lat_store_combined(LatShared &ls, PartOut *dst, int lane, unsigned stamp)
{
    double v = 0.0;
    long long ob = 0;
#pragma unroll
    for (int q = 0; q < LAT_WAVES; q++) {
        if (lane < 7)
            v += ls.outw[q][lane];
        ob |= __double_as_longlong(ls.outw[q][7]);
    }
    const uint32_t csum = part_checksum_lanes(v, lane, stamp);
    if (lane < 7)
        ls.comb[lane] = v;
    if (lane == 7)
        ls.comb[7] = pack_ovl((int)(ob & 1), (int)((ob >> 1) & 1), stamp, csum);
    wave_sync();
    store_part<WRITE_THROUGH>(dst, ls.comb, lane);
    wave_sync();
}

// =================================================================================================
// k_move_eval_lat: one launch per step.  grid (G, replicas of the launch), block 256.
// =================================================================================================
__global__ __launch_bounds__(LAT_WAVES * 64, 2) void k_move_eval_lat(
    BatchView bv, double *rec, const double *__restrict__ qq_tab,
    const int32_t *__restrict__ kpack, FastConsts fc, const MoveRec *__restrict__ cur,
    const MoveRec *__restrict__ prev, PartOut *out, int n_parts, PairParams pp, int r_base,
    const uint8_t *__restrict__ flagv, unsigned stamp)
{
    __shared__ __align__(16) LatShared ls;
    const int tid = threadIdx.x, lane0 = tid & 63;
    int lane = lane0;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int k = tid; k < MMC_QQ_TABLE_DOUBLES; k += LAT_WAVES * 64)
        ls.qtab[k] = qq_tab[k];
    const int r = r_base + blockIdx.y, g = blockIdx.x;
    double *const myrec = rec + (int64_t)r * bv.n_mol * MMC_RSTRIDE;
    const uint16_t *const cq_base = bv.comq + (int64_t)r * 3 * bv.cq_stride;
    LAT_WAVE_SETUP(g * LAT_WAVES + wv, myrec, cq_base)

    // the move record and the pending commit of the previous accepted move (main.jl:598-621)
    const double *mvp = reinterpret_cast<const double *>(cur + r);
    double w = 0.0;
    if (lane < MV_Q_NEW)
        w = mvp[lane];
    const int gflag = flagv ? __builtin_amdgcn_readfirstlane((int)flagv[r]) : -1;
    const long long hdr = __double_as_longlong(w);
    const int i0 = lane_i32((int)hdr, 0) - 1;
    const int flags = gflag >= 0 ? gflag : lane_i32((int)(hdr >> 32), 0);
    const bool commit = prev && (flags & 1);
    const int scur = (flags >> 1) & 1;
    if (commit) {
        const double *pvp = reinterpret_cast<const double *>(prev + r);
        double pw = 0.0; // lane t < 12: word t of the committed record; lane 12: its header
        if (lane < 9) pw = pvp[MV_AT_NEW + lane];
        else if (lane < 12) pw = pvp[MV_COM_NEW + lane - 9];
        else if (lane == 12) pw = pvp[0];
        else if (lane < 17) pw = pvp[MV_Q_NEW + lane - 13];
        const int pm = lane_i32((int)__double_as_longlong(pw), 12) - 1;
        // every workgroup brings its own copy up to date whatever memory held when it loaded it
        // (another workgroup may or may not have written the commit yet)
        wave_sync(); // (lrec was written by this wave's own loads above)
        LAT_REFRESH(pm, pw);
        if (g == 0 && wv == 0) {
            LAT_COMMIT_GLOBAL(r, pm, pw, myrec);
            if (bv.quat) { // totProps.quat[i] = ei (main.jl:619)
                const double q0 = __shfl(pw, 13, 64), q1 = __shfl(pw, 14, 64),
                             q2 = __shfl(pw, 15, 64), q3 = __shfl(pw, 16, 64);
                if (lane >= 13 && lane < 17)
                    quat_commit(bv, r, pm, lane - 13, pw, q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3 > 0.25);
            }
        }
    }
    __syncthreads(); // qtab
#define WV_NS 2
#define WV_S_BASE s_buf(bv, r, scur)
#define WV_S_DST s_buf(bv, r, scur ^ 1)
#include "mmc_wave_lat.inc"
#undef WV_S_DST
#undef WV_S_BASE
#undef WV_NS
    __syncthreads();
    if (wv == 0)
        lat_store_combined<false>(ls, out + (int64_t)r * (n_parts / LAT_WAVES) + g, lane, stamp);
}

// Diagnostic build (-DLAT_PROFILE, scripts/lat_profile.py): every wave of replica 0 sums the
// 10 ns ticks it spends per phase of a step -- waiting for the word, commit, unit body, barrier +
// record, next proposal -- read back with mmc_debug_lat_profile.  Compiled out of the product.
#ifdef LAT_PROFILE
__device__ unsigned long long g_lat_prof[64][8];
#define LAT_TICK(k)                                                                              \
    do {                                                                                         \
        const unsigned long long t_now = __builtin_amdgcn_s_memrealtime();                       \
        lat_acc[k] += t_now - lat_t;                                                             \
        lat_t = t_now;                                                                           \
    } while (0)
#else
#define LAT_TICK(k)
#endif

// =================================================================================================
// k_move_server_lat: the persistent move server, G workgroups per replica.
// grid = R * G (workgroup b: replica b / G, group b % G), block 256.  Protocol of
// k_move_server_wave (control word per replica and step, bounded waits, quit), with every
// workgroup of a replica reading the word itself.
// =================================================================================================
__global__ __launch_bounds__(LAT_WAVES * 64, 2) void k_move_server_lat(
    BatchView bv, double *rec, const double *__restrict__ qq_tab,
    const int32_t *__restrict__ kpack, FastConsts fc, PartOut *out, int n_parts, PairParams pp,
    ServerArgs sa)
{
    __shared__ __align__(16) LatShared ls;
    const int tid = threadIdx.x, lane0 = tid & 63;
    int lane = lane0;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int k = tid; k < MMC_QQ_TABLE_DOUBLES; k += LAT_WAVES * 64)
        ls.qtab[k] = qq_tab[k];
    const int G = n_parts / LAT_WAVES;
    const int r = blockIdx.x / G, g = blockIdx.x - r * G;
    double *const myrec = rec + (int64_t)r * bv.n_mol * MMC_RSTRIDE;
    const uint16_t *const cq_base = bv.comq + (int64_t)r * 3 * bv.cq_stride;
    LAT_WAVE_SETUP(g * LAT_WAVES + wv, myrec, cq_base)
    __syncthreads();
    double *const mvw = ls.mvw[wv];
    PartOut *const part_dst = out + (int64_t)r * G + g;
    auto load_steps = [&]() {
        const double *p = reinterpret_cast<const double *>(sa.steps + r);
        return make_double2(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM),
                            __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    };
    double2 sz = load_steps();
    const ChainKey ck{ sa.seed, (uint32_t)(sa.replica0 + (uint64_t)r) };

    // the move record of step s (lane t = word t), drawn from the chosen molecule's record in
    // global memory -- which this workgroup keeps current itself (LAT_COMMIT_GLOBAL below)
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
#ifdef LAT_PROFILE
    unsigned long long lat_acc[6] = { 0, 0, 0, 0, 0, 0 }, lat_t = __builtin_amdgcn_s_memrealtime();
    unsigned long long sub_acc[4] = { 0, 0, 0, 0 };
#endif
    for (int64_t step = 0;; step++) {
        asm volatile("" : "+v"(lane)); // keep lane-derived values out of LICM (see k_move_eval_wave)
        // ---- wave 0: the host's word of this step (bounded wait), published through LDS ----
        if (wv == 0) {
            unsigned long long c = 0;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned spins = 1;; spins++) {
                unsigned long long v = 0;
                if (lane == 0)
                    v = __hip_atomic_load(sa.ctrl + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                c = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32)
                    | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
                if ((c >> 40) == (((unsigned long long)(step + 1) + sa.seq_off) & 0xffffffULL) || (c & SRV_QUIT))
                    break;
                if ((spins & 63) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > SRV_TIMEOUT_TICKS) {
                    if (lane == 0) {
                        sa.timeout_flag[1] = r;
                        sa.timeout_flag[2] = (int32_t)step;
                        sa.timeout_flag[3] = (int32_t)(c >> 40);
                        sa.timeout_flag[4] = (int32_t)c;
                        __hip_atomic_store(sa.timeout_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    c = SRV_GAVE_UP;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0)
                ls.ctl = c;
        }
        __syncthreads(); // the word is published (and every wave is done with the previous step's LDS)
        LAT_TICK(0); // waited for the word
        const unsigned long long c =
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ls.ctl >> 32)) << 32)
            | (unsigned)__builtin_amdgcn_readfirstlane((int)ls.ctl);
        if (c == SRV_GAVE_UP)
            return;
        const unsigned flags = (unsigned)(c & 0xffu);
        const unsigned stamp = (unsigned)((c >> 8) & MMC_STAMP_MASK);
        // the previous step was accepted (main.jl:598-621): every wave refreshes its own copy, one
        // wave per workgroup writes the global copies (this workgroup's L2 then holds them)
        // (the global copies are written after this step's record has left: only later proposals
        // and whoever reads the state after the run need them)
        const bool commit_prev = (flags & SRV_ACCEPT) && step > 0;
        const int pm = mol_cur == 0 ? n_mol - 1 : mol_cur - 1; // (step - 1) mod n_mol
        double commit_val = 0.0;
        if (commit_prev) {
            commit_val = __shfl(pw, lane < 9 ? MV_AT_NEW + lane : MV_COM_NEW + (lane < 12 ? lane - 9 : 0), 64);
            LAT_REFRESH(pm, commit_val);
        }
        // (... at once where no record follows -- a quit -- and where the next proposal reads the
        // very molecule: two molecules, every wave reads the record for itself after the barrier)
        const bool commit_early = (flags & SRV_QUIT) || n_mol <= 2;
        if (commit_prev && wv == 0 && commit_early)
            LAT_COMMIT_GLOBAL(r, pm, commit_val, myrec);
        if (flags & SRV_QUIT)
            break;
        if (flags & SRV_STEPS)
            sz = load_steps();
        if (n_mol == 1 || (flags & SRV_STEPS)) // the speculative proposal is out of date
            w = make_proposal(step, mol_cur);
        LAT_TICK(1); // commit of the previous step
        const int i0 = mol_cur;
        const int scur = (flags & SRV_SCUR) ? 1 : 0;
#define WV_NS 2
#define WV_S_BASE s_buf(bv, r, scur)
#define WV_S_DST s_buf(bv, r, scur ^ 1)
#ifdef LAT_PROFILE
        unsigned long long sub_t = __builtin_amdgcn_s_memrealtime();
#define LAT_SUBTICK(k)                                                                           \
    do {                                                                                         \
        const unsigned long long t_now = __builtin_amdgcn_s_memrealtime();                       \
        sub_acc[k] += t_now - sub_t;                                                             \
        sub_t = t_now;                                                                           \
    } while (0)
#endif
#include "mmc_wave_lat.inc"
#ifdef LAT_PROFILE
#undef LAT_SUBTICK
#endif
#undef WV_S_DST
#undef WV_S_BASE
#undef WV_NS
        LAT_TICK(2); // unit body
        __syncthreads(); // every wave's sums are in ls.outw
        LAT_TICK(3); // waited for the slowest wave
        if (wv == 0) {
            lat_store_combined<true>(ls, part_dst, lane, stamp);
            if (commit_prev && !commit_early)
                LAT_COMMIT_GLOBAL(r, pm, commit_val, myrec);
        }
        LAT_TICK(4); // record
        pw = w;
        if (n_mol > 1) {
            // while the host decides this step.  (The record read here is another molecule's than
            // the one wave 0 has just committed, unless there are only two: see commit_early.)
            w = make_proposal(step + 1, mol_cur + 1 == n_mol ? 0 : mol_cur + 1);
        }
        LAT_TICK(5); // next proposal
        mol_cur = mol_cur + 1 == n_mol ? 0 : mol_cur + 1;
    }
#ifdef LAT_PROFILE
    if (r == 0 && lane == 0)
        for (int k = 0; k < 6; k++)
            g_lat_prof[g * LAT_WAVES + wv][k] = lat_acc[k];
    if (r == 0 && lane == 0) { // two of the unit's sub-phases share the last two slots
        g_lat_prof[g * LAT_WAVES + wv][6] = (sub_acc[0] + sub_acc[1]) | (sub_acc[2] << 32);
        g_lat_prof[g * LAT_WAVES + wv][7] = sub_acc[3];
    }
#endif
}
