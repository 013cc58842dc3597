// mmc_hip.hip -- single translation unit of libmmc_hip.so (hipcc --offload-arch=gfx950).
// C ABI: include/mmc_hip.h.  No torch types, no CPU fallback: without a HIP device every entry
// point that needs one returns MMC_ERR_HIP.
#include "mmc_host.hpp"
#include "mmc_propose.hpp"

#include "mmc_system.inc"
#include "mmc_ctx.inc"
#include "mmc_batch.inc"
#include "mmc_engine.inc"
#include "mmc_dist.inc"
