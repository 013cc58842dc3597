// gather_bw.hip -- what HBM delivers for the move kernel's ACCESS PATTERN, without its arithmetic.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/gather_bw scripts/gather_bw.hip && build/gather_bw
//
// k_move_eval_wave moves ~32 KB per trial move: 125 neighbour records gathered as 128-byte lines
// scattered over the replica's 96 KB record array (16 KB), the replica's S(k) -- the 293 k-vectors a
// batch keeps of the reference's 337 (k_kvec_setup) -- read (4.7 KB) and written to its other
// buffer (4.7 KB), 4.5 KB of fixed-point centres of mass streamed, a move record in and a result out.  Half of the bytes are 128-byte gathers: a stream copy's 6.29 TB/s
// (MI355X_MICROARCH.md) is not the ceiling for that mix.  This program replays the pattern with the
// same launch shape (1280 workgroups of 4 waves, persistent, one wave per "move", replicas of the
// same sizes laid out the same way) and reports bytes / time:
//   1. gathers only           random 128-B lines (6 x 16 B per lane, like the kernel's record loads)
//   2. the kernel's mix       scan stream + gather + S read + S write per unit
//   3. streams only           the same bytes with the gathers replaced by a contiguous read
// Nothing here computes an energy; every loaded value is folded into a checksum so that no load is
// dropped.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                 \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                              \
            exit(1);                                                                             \
        }                                                                                        \
    } while (0)

constexpr int N_MOL = 750, N_NEIGH = 125, NK = 352 /* stride */, NK_USED = 293;
constexpr int64_t REC_DOUBLES = 16;                  // 128-byte records
constexpr int64_t REP_REC = (int64_t)N_MOL * REC_DOUBLES; // doubles of records per replica
constexpr int64_t REP_S = 2 * 2 * NK;                // two S buffers of NK complex doubles
constexpr int64_t REP_CQ = 3 * 768;                  // 16-bit codes (x, y interleaved, then z), padded

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// mode bit 0: scan stream, bit 1: gather (bit 4: ... as a contiguous read instead), bit 2: S read,
// bit 3: S write, bit 5: codes by wide loads, bit 6: nontemporal S stores, bit 7: nontemporal loads
// (the mode is a template parameter: a run-time test of it around a load makes the compiler wait for
// every outstanding load at each use -- the "gathers only" case ran 97 instead of 81 us that way)
template <int mode>
__global__ __launch_bounds__(256) void k_pattern(const double *rec, double *S, const uint16_t *cq,
                                                 int n_units, unsigned salt, double *sink)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc = 0.0;
    for (int unit = blockIdx.x * 4 + wv; unit < n_units; unit += gridDim.x * 4) {
        const double *myrec = rec + (int64_t)unit * REP_REC;
        double *myS = S + (int64_t)unit * REP_S;
        if (mode & 32) { // (experiment) the same 4.5 KB of codes with 16-byte and 8-byte loads per lane
            const uint4 *xy4 = reinterpret_cast<const uint4 *>(cq + (int64_t)unit * REP_CQ);
            const uint2 *z4 = reinterpret_cast<const uint2 *>(cq + (int64_t)unit * REP_CQ + 2 * 768);
#pragma unroll
            for (int b = 0; b < 3; b++) {
                const uint4 v = xy4[64 * b + lane];
                const uint2 w = z4[64 * b + lane];
                acc += (double)(v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y);
            }
        } else
        if (mode & 1) { // the COM scan: 4 B (x, y) + 2 B (z) per molecule, 12 blocks of 64
            const uint32_t *xy = reinterpret_cast<const uint32_t *>(cq + (int64_t)unit * REP_CQ);
            const uint16_t *z = cq + (int64_t)unit * REP_CQ + 2 * 768;
#pragma unroll
            for (int b = 0; b < 12; b++)
                acc += (double)(xy[64 * b + lane] ^ z[64 * b + lane]);
        }
        if (mode & 2) { // two rounds of 64 neighbours (125 of them), six 16-byte loads per record
            for (int round = 0; round < 2; round++) {
                const int n = 64 * round + lane;
                int j = (mode & 16) ? n : (int)(hash32((unsigned)unit * 131u + (unsigned)n + salt) % N_MOL);
                if (n >= N_NEIGH)
                    j = (mode & 16) ? 0 : j; // (the kernel's idle lanes load molecule 0)
                const double2 *src = reinterpret_cast<const double2 *>(myrec + (int64_t)j * REC_DOUBLES);
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    double2 v;
                    if (mode & 128) {
                        v.x = __builtin_nontemporal_load(&src[q].x);
                        v.y = __builtin_nontemporal_load(&src[q].y);
                    } else {
                        v = src[q];
                    }
                    acc += v.x + v.y;
                }
            }
        }
        if (mode & (4 | 8)) { // S_old read / S_new written: NK_USED complex doubles, lane per k
            for (int k = lane; k < NK_USED; k += 64) {
                double2 v = make_double2(1.0, 2.0);
                if (mode & 4) {
                    if (mode & 128) {
                        v.x = __builtin_nontemporal_load(myS + 2 * k);
                        v.y = __builtin_nontemporal_load(myS + 2 * k + 1);
                    } else {
                        v = *reinterpret_cast<const double2 *>(myS + 2 * k);
                    }
                }
                if ((mode & 8) && (mode & 64)) {
                    __builtin_nontemporal_store(v.x + 1.0, myS + 2 * NK + 2 * k);
                    __builtin_nontemporal_store(v.y, myS + 2 * NK + 2 * k + 1);
                } else if (mode & 8)
                    *reinterpret_cast<double2 *>(myS + 2 * NK + 2 * k) = make_double2(v.x + 1.0, v.y);
                else
                    acc += v.x + v.y;
            }
        }
    }
    if (acc == 1.2345e300)
        sink[0] = acc;
}

// The same mix with K consecutive "steps" per unit by the same wave (what a kernel that took the
// accept decision itself could do): step k reads the S buffer step k-1 wrote, the same codes, and 125
// lines of the same 96 KB of records (another random draw: two molecules' neighbour sets share ~17 %).
// How much of that is then served by the L2 / the 256 MiB Infinity Cache instead of HBM?
template <int K>
__global__ __launch_bounds__(256) void k_pattern_steps(const double *rec, double *S, const uint16_t *cq,
                                                       int n_units, unsigned salt, double *sink)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc = 0.0;
    for (int unit = blockIdx.x * 4 + wv; unit < n_units; unit += gridDim.x * 4) {
        const double *myrec = rec + (int64_t)unit * REP_REC;
        double *myS = S + (int64_t)unit * REP_S;
        for (int step = 0; step < K; step++) {
            const uint32_t *xy = reinterpret_cast<const uint32_t *>(cq + (int64_t)unit * REP_CQ);
            const uint16_t *z = cq + (int64_t)unit * REP_CQ + 2 * 768;
#pragma unroll
            for (int b = 0; b < 12; b++)
                acc += (double)(xy[64 * b + lane] ^ z[64 * b + lane]);
            for (int round = 0; round < 2; round++) {
                const int n = 64 * round + lane;
                int j = (int)(hash32((unsigned)unit * 131u + (unsigned)n + salt + 7919u * step) % N_MOL);
                if (n >= N_NEIGH)
                    j = 0;
                const double2 *src = reinterpret_cast<const double2 *>(myrec + (int64_t)j * REC_DOUBLES);
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const double2 v = src[q];
                    acc += v.x + v.y;
                }
            }
            const int cur = step & 1;
            for (int k = lane; k < NK_USED; k += 64) {
                double2 v = *reinterpret_cast<const double2 *>(myS + 2 * NK * cur + 2 * k);
                *reinterpret_cast<double2 *>(myS + 2 * NK * (cur ^ 1) + 2 * k) = make_double2(v.x + 1.0, v.y);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (acc == 1.2345e300)
        sink[0] = acc;
}

int main(int argc, char **argv)
{
    const int n_units = argc > 1 ? atoi(argv[1]) : 32768;
    const int reps = argc > 2 ? atoi(argv[2]) : 40;
    double *rec, *S, *sink;
    uint16_t *cq;
    // two "groups" of replicas, used alternately, like the bench (nothing stays in a cache between launches)
    CHECK(hipMalloc(&rec, sizeof(double) * REP_REC * n_units * 2));
    CHECK(hipMalloc(&S, sizeof(double) * REP_S * n_units * 2));
    CHECK(hipMalloc(&cq, sizeof(uint16_t) * REP_CQ * n_units * 2 + 4096));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(rec, 0, sizeof(double) * REP_REC * n_units * 2));
    CHECK(hipMemset(S, 0, sizeof(double) * REP_S * n_units * 2));
    CHECK(hipMemset(cq, 0, sizeof(uint16_t) * REP_CQ * n_units * 2 + 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct Case { const char *name; int mode; double bytes_per_unit; };
    const double gather_b = 125.0 * 128.0, scan_b = 750.0 * 6.0, s_b = NK_USED * 16.0;
    const Case cases[] = {
        { "gathers only (125 random 128-B lines per unit)", 2, gather_b },
        { "the move kernel's mix: scan + gather + S read + S write", 1 | 2 | 4 | 8, scan_b + gather_b + 2 * s_b },
        { "the same bytes, gathers as a contiguous read", 1 | 2 | 16 | 4 | 8, scan_b + gather_b + 2 * s_b },
        { "scan + S read + S write only", 1 | 4 | 8, scan_b + 2 * s_b },
        { "(experiment) the mix with the codes read by 16- and 8-byte loads", 32 | 2 | 4 | 8, scan_b + gather_b + 2 * s_b },
        { "(experiment) scan only, 4- and 2-byte loads", 1, scan_b },
        { "(experiment) scan only, 16- and 8-byte loads", 32, scan_b },
        { "(experiment) S read + S write only", 4 | 8, 2 * s_b },
        { "(experiment) the mix, S written with nontemporal stores", 1 | 2 | 4 | 8 | 64, scan_b + gather_b + 2 * s_b },
        { "(experiment) the mix, nontemporal loads (gather, S) and stores", 1 | 2 | 4 | 8 | 64 | 128, scan_b + gather_b + 2 * s_b },
        { "(experiment) S read + S write only, nontemporal both", 4 | 8 | 64 | 128, 2 * s_b },
    };
    printf("{\"units_per_launch\": %d, \"launches_timed\": %d, \"cases\": [", n_units, reps);
    bool first = true;
    auto launch = [&](int mode, int grp, unsigned salt) {
        const double *r_ = rec + (int64_t)grp * REP_REC * n_units;
        double *s_ = S + (int64_t)grp * REP_S * n_units;
        const uint16_t *c_ = cq + (int64_t)grp * REP_CQ * n_units;
#define MODE_CASE(M) case M: k_pattern<M><<<1280, 256>>>(r_, s_, c_, n_units, salt, sink); break;
        switch (mode) {
            MODE_CASE(2) MODE_CASE(1 | 2 | 4 | 8) MODE_CASE(1 | 2 | 16 | 4 | 8) MODE_CASE(1 | 4 | 8)
            MODE_CASE(32 | 2 | 4 | 8) MODE_CASE(1) MODE_CASE(32) MODE_CASE(4 | 8)
            MODE_CASE(1 | 2 | 4 | 8 | 64) MODE_CASE(1 | 2 | 4 | 8 | 64 | 128) MODE_CASE(4 | 8 | 64 | 128)
        default: fprintf(stderr, "mode %d not instantiated\n", mode); exit(1);
        }
#undef MODE_CASE
    };
    for (const Case &c : cases) {
        for (int w = 0; w < 4; w++) // warm-up, both groups
            launch(c.mode, w & 1, 17u * w);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++)
            launch(c.mode, i & 1, 1000u + 31u * i);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / reps, tbs = c.bytes_per_unit * n_units / (us * 1e-6) / 1e12;
        printf("%s{\"pattern\": \"%s\", \"bytes_per_unit\": %.0f, \"us_per_launch\": %.1f, \"TB_per_s\": %.3f}",
               first ? "" : ", ", c.name, c.bytes_per_unit, us, tbs);
        first = false;
    }
    printf("], \"steps_per_unit\": [");
    {   // K steps per unit by the same wave: time per unit-STEP
        const double bytes = scan_b + gather_b + 2 * s_b;
        auto run = [&](int K, auto kern) {
            for (int w = 0; w < 2; w++)
                kern(w & 1, 17u * w);
            CHECK(hipDeviceSynchronize());
            const int rr = reps / K > 4 ? reps / K : 4;
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < rr; i++)
                kern(i & 1, 1000u + 31u * i);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / rr / K;
            printf("%s{\"K\": %d, \"us_per_unit_steps_of_one_launch\": %.1f, \"TB_per_s_of_the_pattern_bytes\": %.3f}", K == 1 ? "" : ", ", K, us,
                   bytes * n_units / (us * 1e-6) / 1e12);
        };
#define STEPS_CASE(KK) run(KK, [&](int grp, unsigned salt) { k_pattern_steps<KK><<<1280, 256>>>(rec + (int64_t)grp * REP_REC * n_units, S + (int64_t)grp * REP_S * n_units, cq + (int64_t)grp * REP_CQ * n_units, n_units, salt, sink); });
        STEPS_CASE(1) STEPS_CASE(2) STEPS_CASE(4) STEPS_CASE(8) STEPS_CASE(16)
#undef STEPS_CASE
    }
    printf("]}\n");
    return 0;
}
