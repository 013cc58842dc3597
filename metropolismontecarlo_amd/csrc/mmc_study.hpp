// mmc_study.hpp -- single-precision restatement of the hot path for the tolerance study of
// BASELINE.json configs[4] / SURVEY.md section 8(d) "cfg5" (Wolf vs Ewald, fp32 vs fp64).
//
// The reference is fp64 throughout; nothing here is on the product path.  These kernels evaluate
// the same terms -- LJ_poly_dU (energy.jl:209-290), EwaldReal (ewalds.jl:293-376), RecipLong
// (ewalds.jl:538-604), RecipMove (ewalds.jl:718-826) -- from coordinates ROUNDED TO fp32, with fp32
// arithmetic and ACC = float ("fp32") or double ("mixed") accumulators, so that a caller can put a
// number on what single precision would cost: totals and per-move dU against the fp64 path.
// Simple decomposition (workgroup per molecule / per k-vector); speed is not the point.
#pragma once
#include "mmc_kernels.hpp"

struct StudyParams {
    float lj_gate_sq, qq_gate_sq, lj_slack_sq, qq_slack_sq, ovr, kappa, box;
};

struct StudyMolOut {
    double lj_pot, lj_vir, qq_pot;
    int32_t ovl, _pad;
};

// fp32 copies of the coordinates of replica 0: at[3][n_atoms], com[3][n_mol]
__global__ void k_study_to_f32(BatchView bv, float *at, float *com)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < bv.n_atoms) {
        at[i] = (float)bv.ax[i];
        at[bv.n_atoms + i] = (float)bv.ay[i];
        at[2 * bv.n_atoms + i] = (float)bv.az[i];
    }
    if (i < bv.n_mol) {
        com[i] = (float)bv.comx[i];
        com[bv.n_mol + i] = (float)bv.comy[i];
        com[2 * bv.n_mol + i] = (float)bv.comz[i];
    }
}

__device__ __forceinline__ float vector1D_f(float c1, float c2, float box) // ewalds.jl:30-38
{
    if (c1 < c2) {
        const float d = c2 - c1;
        return d < box - d ? d : d - box;
    }
    const float d = c1 - c2;
    return d < box - d ? -d : -d + box;
}

template <typename ACC>
__device__ __forceinline__ ACC study_block_sum(ACC v, ACC *sh)
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = MMC_BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const ACC r = sh[0];
    __syncthreads();
    return r;
}

// Workgroup b evaluates molecule i = i_first + b against all others.  `chosen` (12 floats:
// atoms x0 y0 z0 .. z2, COM) replaces molecule i's own coordinates when not NULL (trial state;
// 3-atom molecules only).
template <typename ACC>
__global__ __launch_bounds__(MMC_BLOCK) void k_study_mol(BatchView bv, const float *at,
                                                         const float *com, int i_first,
                                                         const float *chosen, StudyParams sp,
                                                         StudyMolOut *out)
{
    __shared__ ACC sh[MMC_BLOCK];
    __shared__ int sh_ovl;
    const int i = i_first + blockIdx.x, n_mol = bv.n_mol, n_at = bv.n_atoms;
    const int fi = bv.first0[i], ni = bv.cnt[i];
    if (threadIdx.x == 0)
        sh_ovl = 0;
    __syncthreads();
    float ci[3];
    for (int d = 0; d < 3; d++)
        ci[d] = chosen ? chosen[9 + d] : com[d * n_mol + i];
    ACC lj = 0, vir = 0, qq = 0;
    int ovl = 0;
    for (int j = threadIdx.x; j < n_mol; j += MMC_BLOCK) {
        if (j == i)
            continue;
        const float rx = vector1D_f(ci[0], com[j], sp.box);
        const float ry = vector1D_f(ci[1], com[n_mol + j], sp.box);
        const float rz = vector1D_f(ci[2], com[2 * n_mol + j], sp.box);
        const float rij2 = rx * rx + ry * ry + rz * rz;
        const bool g_lj = rij2 < sp.lj_gate_sq, g_qq = rij2 < sp.qq_gate_sq; // energy.jl:254
        if (!g_lj && !g_qq)
            continue;
        const int fj = bv.first0[j], nj = bv.cnt[j];
        for (int a = 0; a < ni; a++) {
            float pa[3];
            for (int d = 0; d < 3; d++)
                pa[d] = chosen ? chosen[3 * a + d] : at[d * n_at + fi + a];
            const int ta = bv.atype[fi + a];
            const float qa = (float)bv.charge[fi + a];
            for (int b = 0; b < nj; b++) {
                const float dx = vector1D_f(pa[0], at[fj + b], sp.box);
                const float dy = vector1D_f(pa[1], at[n_at + fj + b], sp.box);
                const float dz = vector1D_f(pa[2], at[2 * n_at + fj + b], sp.box);
                const float r2 = dx * dx + dy * dy + dz * dz;
                const int tb = bv.atype[fj + b];
                if (g_lj) {
                    const float e = (float)bv.eps[ta + tb * bv.n_types];
                    if (r2 < sp.lj_slack_sq && e > 0.001f) { // energy.jl:270
                        const float sg = (float)bv.sig[ta + tb * bv.n_types];
                        const float s2 = sg * sg / r2, s6 = s2 * s2 * s2, s12 = s6 * s6;
                        const float virab = e * (2.0f * s12 - s6);
                        lj += (ACC)(e * (s12 - s6));
                        vir += (ACC)(rx * (dx * virab * s2) + ry * (dy * virab * s2) +
                                     rz * (dz * virab * s2));
                    }
                }
                if (g_qq) {
                    const float qab = qa * (float)bv.charge[fj + b];
                    if (r2 < sp.ovr && qab < 0.0f) // ewalds.jl:359
                        ovl = 1;
                    else if (r2 < sp.qq_slack_sq) { // ewalds.jl:362-367
                        const float r = sqrtf(r2);
                        qq += (ACC)(qab * erfcf(sp.kappa * r) / r);
                    }
                }
            }
        }
    }
    if (ovl)
        sh_ovl = 1;
    const ACC t_lj = study_block_sum<ACC>(lj, sh), t_vir = study_block_sum<ACC>(vir, sh),
              t_qq = study_block_sum<ACC>(qq, sh);
    if (threadIdx.x == 0) {
        StudyMolOut o;
        o.lj_pot = (double)t_lj;
        o.lj_vir = (double)t_vir;
        o.qq_pot = (double)t_qq;
        o.ovl = sh_ovl;
        o._pad = 0;
        out[blockIdx.x] = o;
    }
}

// RecipLong in fp32: workgroup per k-vector.  S[k] -> Sf (float2), energy term cfac |S|^2 -> ek.
template <typename ACC>
__global__ __launch_bounds__(MMC_BLOCK) void k_study_recip_long(BatchView bv, const float *at,
                                                                float box, float2 *Sf, double *ek)
{
    __shared__ ACC sh[MMC_BLOCK];
    const int k = blockIdx.x, n_at = bv.n_atoms;
    const float tw = 6.2831853071795864769f / box;
    const float kx = tw * (float)bv.kxyz[3 * k], ky = tw * (float)bv.kxyz[3 * k + 1],
                kz = tw * (float)bv.kxyz[3 * k + 2];
    ACC re = 0, im = 0;
    for (int l = threadIdx.x; l < n_at; l += MMC_BLOCK) {
        float s, c;
        sincosf(kx * at[l] + ky * at[n_at + l] + kz * at[2 * n_at + l], &s, &c);
        const float q = (float)bv.charge[l];
        re += (ACC)(q * c);
        im += (ACC)(q * s);
    }
    const ACC tr = study_block_sum<ACC>(re, sh), ti = study_block_sum<ACC>(im, sh);
    if (threadIdx.x == 0) {
        const float fr = (float)tr, fi = (float)ti;
        Sf[k] = make_float2(fr, fi);
        ek[k] = (double)((float)bv.cfac[k] * (fr * fr + fi * fi));
    }
}

// RecipMove in fp32 for one 3-atom molecule: thread per k-vector; S itself is not modified.
// old3 / new3: 9 floats each; q3: the three charges.
template <typename ACC>
__global__ __launch_bounds__(MMC_BLOCK) void k_study_recip_move(BatchView bv, const float2 *Sf,
                                                                const float *old3,
                                                                const float *new3, int i0,
                                                                float box, double *out)
{
    __shared__ ACC sh[MMC_BLOCK];
    const int fi = bv.first0[i0];
    const float tw = 6.2831853071795864769f / box;
    ACC e = 0;
    for (int k = threadIdx.x; k < bv.nkvecs; k += MMC_BLOCK) {
        const float kx = tw * (float)bv.kxyz[3 * k], ky = tw * (float)bv.kxyz[3 * k + 1],
                    kz = tw * (float)bv.kxyz[3 * k + 2];
        float dr = 0.f, di = 0.f;
        for (int l = 0; l < 3; l++) {
            float sn, cn, so, co;
            sincosf(kx * new3[3 * l] + ky * new3[3 * l + 1] + kz * new3[3 * l + 2], &sn, &cn);
            sincosf(kx * old3[3 * l] + ky * old3[3 * l + 1] + kz * old3[3 * l + 2], &so, &co);
            const float q = (float)bv.charge[fi + l];
            dr += q * (cn - co); // ewalds.jl:803-814
            di += q * (sn - so);
        }
        const float2 s = Sf[k];
        const float nr = s.x + dr, ni = s.y + di;
        e += (ACC)((float)bv.cfac[k] * ((nr * nr + ni * ni) - (s.x * s.x + s.y * s.y)));
    }
    const ACC t = study_block_sum<ACC>(e, sh);
    if (threadIdx.x == 0)
        out[0] = (double)t;
}
