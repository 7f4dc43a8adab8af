// trav4_gt.hip — trav4_kernel (traverse4.inc) with the grouped table (RADHIP_TABLE=group; needs the index's graph-locality
// layout): adjacency rows of up to 16 slots, the WIDE form, and the SLOT form of the narrow one.
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
#define RH_KG(LPR) trav4_kernel<LPR, true, false, false, false, false>
#define RH_KGW(LPR) trav4_kernel<LPR, true, false, false, true, false>
#define RH_KGS(LPR) trav4_kernel<LPR, true, false, false, false, true>
#define RH_KGWS(LPR) trav4_kernel<LPR, true, false, false, true, true>

int rh_trav4_launch_grouped(bool wide, bool slot, int lpr, uint32_t grid, hipStream_t st, const TravParams &P) {
    if (wide && slot) { RH_T4_CASES(RH_KGWS, grid, st, P) }
    else if (wide) { RH_T4_CASES(RH_KGW, grid, st, P) }
    else if (slot) { RH_T4_CASES(RH_KGS, grid, st, P) }
    else { RH_T4_CASES(RH_KG, grid, st, P) }
    RH_HIP(hipGetLastError());
    return RADHIP_OK;
}
