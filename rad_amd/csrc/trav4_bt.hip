// trav4_bt.hip — trav4_kernel (traverse4.inc) with the bucket table (the default): adjacency rows of up to 16 slots, the
// WIDE form (17..64 slots), and the SLOT forms of both (heavy state per resident row, round 4).
#include "traverse_dev.h"
#include "traverse4.inc"

#define RH_T4_CASES(K, GRID, ST, P)                                                          \
    switch (lpr) {                                                                           \
        case 1: hipLaunchKernelGGL((K(1)), dim3(GRID), dim3(64), 0, ST, P); break;           \
        case 2: hipLaunchKernelGGL((K(2)), dim3(GRID), dim3(64), 0, ST, P); break;           \
        case 4: hipLaunchKernelGGL((K(4)), dim3(GRID), dim3(64), 0, ST, P); break;           \
        case 8: hipLaunchKernelGGL((K(8)), dim3(GRID), dim3(64), 0, ST, P); break;           \
        default: hipLaunchKernelGGL((K(16)), dim3(GRID), dim3(64), 0, ST, P); break;         \
    }
#define RH_KB(LPR) trav4_kernel<LPR, false, false, true, false, false>
#define RH_KBW(LPR) trav4_kernel<LPR, false, false, true, true, false>
#define RH_KBS(LPR) trav4_kernel<LPR, false, false, true, false, true>
#define RH_KBWS(LPR) trav4_kernel<LPR, false, false, true, true, true>

int rh_trav4_launch_bucket(bool wide, bool slot, int lpr, uint32_t grid, hipStream_t st, const TravParams &P) {
    if (wide && slot) { RH_T4_CASES(RH_KBWS, grid, st, P) }
    else if (wide) { RH_T4_CASES(RH_KBW, grid, st, P) }
    else if (slot) { RH_T4_CASES(RH_KBS, grid, st, P) }
    else { RH_T4_CASES(RH_KB, grid, st, P) }
    RH_HIP(hipGetLastError());
    return RADHIP_OK;
}
