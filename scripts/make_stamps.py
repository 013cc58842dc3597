#!/usr/bin/env python
"""Developer tool: build scratch/libmmc_stamps.so, a copy of the library whose k_move_eval_fast
records wall_clock64() (10 ns ticks) at phase boundaries for one workgroup.  Not part of the product."""
import os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "metropolismontecarlo_amd", "csrc"), os.path.join(ROOT, "scratch", "csrc")
shutil.rmtree(dst, ignore_errors=True); shutil.copytree(src, dst)
p = os.path.join(dst, "mmc_fast.hpp"); s = open(p).read()
def rep(a, b, cnt=1):
    global s
    assert a in s, a
    s = s.replace(a, b, cnt)
rep("struct FastShared {", "__device__ long long g_stamps[32];\n#define STAMP(i) do { if (blockIdx.x==0 && blockIdx.y==7 && threadIdx.x==0) g_stamps[i] = wall_clock64(); } while(0)\nstruct FastShared {")
rep("    // ================= trip 1:", "    STAMP(0);\n    // ================= trip 1:")
rep("    __syncthreads();\n\n    const int2 hdr", "    STAMP(1);\n    __syncthreads();\n    STAMP(2);\n\n    const int2 hdr")
rep("    double a_lj0 = 0, a_lj1 = 0", "    STAMP(3);\n    double a_lj0 = 0, a_lj1 = 0")
rep("            if (lane_id() == 0)\n                sm.wcnt[w] = count;\n            __syncthreads();", "            STAMP(4);\n            if (lane_id() == 0)\n                sm.wcnt[w] = count;\n            __syncthreads();\n            STAMP(5);")
rep("                __syncthreads();\n                // ---- Coulomb pass", "                STAMP(6);\n                __syncthreads();\n                STAMP(7);\n                // ---- Coulomb pass")
rep("                // ---- LJ pass: only atom pairs", "                STAMP(8);\n                // ---- LJ pass: only atom pairs")
rep("                __syncthreads(); // the tile and the list are reused", "                STAMP(9);\n                __syncthreads(); // the tile and the list are reused\n                STAMP(10);")
rep("    if (do_recip) {\n        __syncthreads(); // ptab", "    STAMP(11);\n    if (do_recip) {\n        __syncthreads(); // ptab")
rep("    // one transpose-reduction", "    STAMP(12);\n    // one transpose-reduction")
rep("    if (tid == 0) {\n        const int of", "    STAMP(13);\n    if (tid == 0) {\n        const int of")
rep("        out[(int64_t)r * n_parts + part] = po;\n    }\n}", "        out[(int64_t)r * n_parts + part] = po;\n    }\n    STAMP(14);\n}")
open(p, "w").write(s)
b = os.path.join(dst, "mmc_batch.inc")
open(b, "a").write('''
extern "C" int32_t mmc_dbg_get_stamps(long long *out)
{
    MMC_HIP(hipDeviceSynchronize());
    MMC_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(long long) * 32));
    return MMC_OK;
}
''')
h = os.path.join(dst, "mmc_host.hpp")
txt = open(h).read().replace('#include "../../include/mmc_hip.h"', '#include "mmc_hip.h"')
open(h, "w").write(txt)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-o",
                       os.path.join(ROOT, "scratch", "libmmc_stamps.so"), os.path.join(dst, "mmc_hip.hip")])
print("built scratch/libmmc_stamps.so")
