// mmc_host.hpp -- host-side state shared by the context, batch and engine translation units.
#pragma once
#include "../../include/mmc_hip.h"
#include "mmc_total.hpp"
#include "mmc_wave.hpp"
#include "mmc_lat.hpp"
#include "mmc_potential.hpp"
#include <string>
#include <vector>

void mmc_set_error(const char *fmt, ...);

#define MMC_HIP(call)                                                                            \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            mmc_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,      \
                          __LINE__);                                                             \
            return MMC_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

#define MMC_REQUIRE(cond, code, ...)                                                             \
    do {                                                                                         \
        if (!(cond)) {                                                                           \
            mmc_set_error(__VA_ARGS__);                                                          \
            return (code);                                                                       \
        }                                                                                        \
    } while (0)

#define MMC_TRY(expr)                                                                            \
    do {                                                                                         \
        int32_t st__ = (expr);                                                                   \
        if (st__ != MMC_OK)                                                                      \
            return st__;                                                                         \
    } while (0)

#define MMC_MAX_PARTS 32 // result records (units) per trial move
static_assert(LAT_MAX_PARTS / LAT_WAVES <= MMC_MAX_PARTS, "one record per workgroup of the latency kernels");

// Device state of R replicas of one system + Ewald tables.  R = 1 for a context.
struct DeviceSystem {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t R = 0, n_mol = 0, n_atoms = 0, n_types = 0;
    double box = 0.0;
    bool uploaded = false, ewald_ready = false, uniform3 = false;
    // fast paths (mmc_fast.hpp, mmc_total.hpp): systems whose molecules are all copies of one
    // 3-atom molecule get per-molecule records, launch constants and the erfc table
    bool homogeneous = false;
    // largest distance of an atom from its molecule's centre of mass over everything uploaded
    // (rigid moves keep it); +inf once unknown.  Decides whether an atom pair's minimum image may
    // be taken with its molecule's (WV_IMG, mmc_wave_unit.inc).
    double r_mol_max = 0.0;
    void note_shape(const double *com, const double *coords); // one replica's host arrays
    bool image_by_molecule(double gate_sq) const;
    bool pairs_inside_slack(double gate_sq, double slack_sq) const;
    double *rec = nullptr;    // [R][n_mol][MMC_RSTRIDE], only when homogeneous
    FastConsts fc{};          // launch constants of the fast kernels
    double *qq_tab = nullptr; // [MMC_QQ_NINT][MMC_QQ_NCOEF] for the prepared kappa
    int32_t *kpack = nullptr; // [MMC_NK_STRIDE + MMC_KCOLS] packed k-vectors and their (kx, ky) columns (k_pack_kvec)
    int16_t *tile_pairs = nullptr; // [n_tile_pairs][2], I <= J (k_total_pairs)
    int n_tile_pairs = 0;
    TotalPart *d_tparts = nullptr; // [R][n_tile_pairs], lazily
    double *d_phase = nullptr;     // [R][n_atoms][6], lazily (k_atom_phases)
    double *d_spart = nullptr;     // [R][16][2 * MMC_NK_STRIDE] chunk partials of S(k), lazily
    bool sums_valid = false;       // c_sq / c_sq2 = sum q, sum q^2 of the uploaded charges
    bool recip_lds_ready = false;  // k_recip_long_lds has been granted its dynamic LDS size
    int64_t lds_per_block = 0;     // hipDeviceAttributeMaxSharedMemoryPerBlock of the device (init)
    bool recip_lds_refused = false; // the device would not grant the LDS k_recip_long_lds asks for
    RecipOrder recip_order;        // its (kx, ky) columns, most work first (prepare_ewald)
    double c_sq = 0.0, c_sq2 = 0.0;
    bool fast_table_ok(double qq_rcut) const; // the erfc table covers this cutoff
    int64_t nk = 0, k_sq_max = 0;
    int64_t nkvecs = 0;     // k-vectors the device holds (= bv.nkvecs)
    int64_t nkvecs_ref = 0; // ... and the reference's NKVECS (ewalds.jl:90): the same unless half_k
    bool half_k = false;    // keep one of each conjugate pair of the kx = 0 plane (k_kvec_setup); set before prepare_ewald
    std::vector<int32_t> ref_to_dev; // [nkvecs_ref] device index, or ~index of the conjugate partner
    void expand_S(const double *dev_order, double *ref_order) const;
    BatchView bv{};
    std::vector<void *> allocs; // hipMalloc'ed
    std::vector<int32_t> h_first0, h_cnt;
    std::vector<double> h_charge;
    // scratch
    MolE *d_permol = nullptr;      // [R][n_mol]
    void *d_scr = nullptr;         // 64 B per replica: reduced results before the D2H copy
    void *h_res = nullptr;         // pinned, mapped: small per-call results (4 KiB)
    void *d_res = nullptr;         // device alias of h_res
    double *d_stage = nullptr;     // device staging for AoS uploads/downloads (3*n_atoms)
    std::vector<double> h_stage;

    int32_t init(int dev, void *hip_stream, int64_t replicas);
    void release();
    int32_t dmalloc(void **p, size_t bytes);
    int32_t upload(int64_t n_mol_, int64_t n_atoms_, const double *com,
                   const int64_t *first_atom, const int64_t *last_atom, const double *coords,
                   const int64_t *atype, const double *charge, int64_t n_types_,
                   const double *eps, const double *sig, double box_);
    int32_t set_replica(int64_t r, const double *com, const double *coords);
    int32_t get_replica(int64_t r, double *com, double *coords);
    int32_t broadcast_replica0();
    int32_t prepare_ewald(double kappa, int64_t nk_, int64_t k_sq_max_, double box_,
                          double factor, int64_t *nkvecs_out);
    int32_t sync();
    // per-molecule energies for molecules [i_base, i_base + n_sel) of every replica
    int32_t mol_energy(int i_base, int n_sel, bool lj, bool qq, int style, const PairParams &pp,
                       MolE *out, int out_stride);
    int32_t recip_long_all(double *energies_host /* [R] */);
    int32_t recip_long_enqueue(double *energies_host /* [R] */);
    int32_t pair_totals_enqueue(double lj_rcut, double qq_rcut, std::vector<TotalsRaw> &ht, bool *fast);
    int32_t pair_totals_finish(double lj_rcut, double qq_rcut, std::vector<TotalsRaw> &ht, bool fast);
    int32_t totals_ewald(double lj_rcut, double qq_rcut, mmc_totals *tot /* [R] */);
    // one system, one launch (mmc_potential.hpp): pair totals and / or structure factor + energy
    TotalPart *d_tparts1 = nullptr; // [units] of the one-system launch
    double *d_spart1 = nullptr;   // [POT_MAX_CHUNKS][2 * MMC_NK_STRIDE] chunk partials of S(k)
    unsigned *d_ticket = nullptr; // the last-workgroup ticket, zero between launches
    unsigned pot_stamp = 0;
    int32_t potential_one(double lj_rcut, double qq_rcut, bool pairs, bool recip, PotOneOut *res);
    // sum_i 4 pot_i, sum_i 8 vir_i, sum_i EwaldReal_i and the overlap count, per replica
    int32_t pair_totals(double lj_rcut, double qq_rcut, std::vector<TotalsRaw> &ht);
    int32_t charge_sums(double *sum_q, double *sum_q2);
    // volume move: rescale every replica to `new_box` and rebuild the Ewald tables (K6)
    int32_t volume_change(double new_box, double new_kappa);
    // everything a volume move rewrites, kept on the device so that a rejection is a copy back
    // (one launch) and an acceptance is nothing: coordinates in their three layouts, the
    // fixed-point centres of mass, S(k), the Ewald tables; and the scalars that go with them
    struct {
        bool valid = false;
        uint32_t *buf = nullptr; // one allocation, the segments back to back
        SnapSegs to_snap{}, from_snap{};
        double box = 0, kappa = 0;
        int64_t nkvecs = 0;
        RecipOrder recip_order{};
    } snap;
    int32_t snapshot_take();
    int32_t snapshot_restore();
};

PairParams mmc_pair_params(double lj_rcut, double qq_rcut, double diameter, double ovr,
                           double kappa, bool bare);

bool part_copy_checked(const PartOut *src, unsigned want_stamp, PartOut *dst);
void mmc_combine_parts(const PartOut *parts, int n_parts, double factor, mmc_move_result *res);

// ---- command memory of the persistent servers ---------------------------------------------------
// Device memory the HOST writes (fine-grained, through the large PCIe BAR): a server kernel polling
// it reads local memory, where polling pinned host memory is a PCIe read per look (measured, one
// 8-byte word: 2.02 us per host -> device -> host round trip instead of 2.58; a 512-byte block:
// 2.1 instead of 2.45).  NULL where the device has no large BAR (or MMC_NO_BAR is set): the
// callers keep their pinned host buffers for that case.  The host must never READ this memory.
#if defined(__x86_64__)
#include <immintrin.h>
static inline void mmc_bar_flush() { _mm_sfence(); } // push the write-combining buffers out
#else
static inline void mmc_bar_flush() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#endif
static inline void mmc_cpu_relax()
{
#if defined(__x86_64__)
    _mm_pause();
#endif
}
static inline void *mmc_bar_alloc(size_t bytes)
{
    if (getenv("MMC_NO_BAR"))
        return nullptr;
    int dev = 0, large = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, dev) != hipSuccess || !large)
        return nullptr;
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes < 4096 ? 4096 : bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
