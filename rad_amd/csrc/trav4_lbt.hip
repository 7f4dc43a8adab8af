// trav4_lbt.hip — trav4_kernel (traverse4.inc) with the LOCALITY-HASHED bucket table (RADHIP_TABLE=local; needs the index's
// graph-locality layout): the bucket table's entries and code, a node's home bucket from its layout id.  Narrow and WIDE rows,
// state per traversal and per resident row.
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
#define RH_KL(LPR) trav4_kernel<LPR, true, false, true, false, false>
#define RH_KLW(LPR) trav4_kernel<LPR, true, false, true, true, false>
#define RH_KLS(LPR) trav4_kernel<LPR, true, false, true, false, true>
#define RH_KLWS(LPR) trav4_kernel<LPR, true, false, true, true, true>

int rh_trav4_launch_local(bool wide, bool slot, int lpr, uint32_t grid, hipStream_t st, const TravParams &P) {
    if (wide && slot) { RH_T4_CASES(RH_KLWS, grid, st, P) }
    else if (wide) { RH_T4_CASES(RH_KLW, grid, st, P) }
    else if (slot) { RH_T4_CASES(RH_KLS, grid, st, P) }
    else { RH_T4_CASES(RH_KL, grid, st, P) }
    RH_HIP(hipGetLastError());
    return RADHIP_OK;
}
