// rows_tile.h — shared by topk.hip and index.hip (not by the traversal kernels: their build id does not move with this file)
#pragma once
#include "common.h"

// ---- 1024- and 2048-bit rows (LPR = 8 / 16 chunks of 16 B), ONE ROW PER LANE (round 4: topk.hip topk_rows_kernel, index.hip
// scan_rows_kernel) ---------------------------------------------------------------------------------------------------------
// With a row across eight lanes every count goes through an 8 x 8 transpose-and-sum (252 of the ~1100 VALU instructions a
// 64-row tile costs at 8 queries: more issue time than HBM needs to deliver the tile) and every lane keeps its own chunk of
// every query in registers.  Here a wavefront loads its tile with the same coalesced 16-B-per-lane loads, turns it over
// through LDS in two halves of 32 rows (rows 16 * (LPR + 1) bytes apart: the 16-B reads of 16 lanes cover all 64 banks) and each lane counts
// a whole row; the queries are the same in every lane, so they are scalar operands, and nothing crosses lanes.
#define RH_ROWS_TR_VEC(LPR) (32 * ((LPR) + 1))   // uint4 of LDS per wavefront: 32 rows (half a tile) of LPR chunks + 1 of padding
typedef const uint32_t __attribute__((address_space(4))) *rh_cptr;
// popcount(x) + acc in ONE instruction (the compiler prefers trees of v_add3 over the accumulating form)
__device__ __forceinline__ uint32_t rh_bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
    return d;
}
// the wavefront's coalesced loads of tile `tile` (lane (grp, chunk) = (lane / LPR, lane % LPR): chunk `chunk` of rows
// u * (64 / LPR) + grp, u < LPR)
template <int LPR>
__device__ __forceinline__ void rh_rows_load(const uint4 *__restrict__ fp, uint64_t first, uint64_t count, uint64_t tile, uint32_t lane, uint4 (&nv)[LPR]) {
    constexpr int RPL = 64 / LPR;
    const uint32_t chunk = lane % LPR, grp = lane / LPR;
    const uint64_t r0 = tile * 64;
    if (r0 + 64 <= count) {   // (wave-uniform: every tile but the last)
        const uint4 *base = fp + (first + r0 + grp) * LPR + chunk;
#pragma unroll
        for (int u = 0; u < LPR; ++u) nv[u] = base[u * 64];
    } else {
#pragma unroll
        for (int u = 0; u < LPR; ++u) {
            const uint64_t r = r0 + (uint64_t)u * RPL + grp;
            nv[u] = make_uint4(0, 0, 0, 0);
            if (r < count) nv[u] = fp[(first + r) * LPR + chunk];
        }
    }
}
// nv (as loaded) -> v = the LPR chunks of row `lane` of the tile, through tr[RH_ROWS_TR_VEC(LPR)]; nv is free afterwards
template <int LPR>
__device__ __forceinline__ void rh_rows_turn(const uint4 (&nv)[LPR], uint4 *tr, uint32_t lane, uint4 (&v)[LPR]) {
    constexpr int RPL = 64 / LPR, HU = LPR / 2;   // rows per load, loads per half tile
    const uint32_t chunk = lane % LPR, grp = lane / LPR;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int u = 0; u < HU; ++u) tr[(u * RPL + grp) * (LPR + 1) + chunk] = nv[h * HU + u];
        RH_WAVE_SYNC();
        if ((int)(lane >> 5) == h) {
#pragma unroll
            for (int c = 0; c < LPR; ++c) v[c] = tr[(lane & 31u) * (LPR + 1) + c];
        }
        RH_WAVE_SYNC();   // (the half tile is written again right away)
    }
}
// rp = popcount of the lane's row, a[i] = popcount(row & query i); qd = [NQ][4 * LPR] query words in global memory
template <int LPR, int NQ>
__device__ __forceinline__ void rh_rows_count(const uint4 (&v)[LPR], const uint32_t *qd, uint32_t &rp, uint32_t (&a)[NQ]) {
    rp = 0;
#pragma unroll
    for (int i = 0; i < NQ; ++i) a[i] = 0;
#pragma unroll
    for (int c = 0; c < LPR; ++c) rp = rh_bcnt_acc(v[c].x, rh_bcnt_acc(v[c].y, rh_bcnt_acc(v[c].z, rh_bcnt_acc(v[c].w, rp))));
#pragma unroll
    for (int cb = 0; cb < LPR; cb += 4) {
        // the same address in every lane: scalar loads from the constant address space (64 B = four chunks of a query at a
        // time) — behind an offset the compiler cannot see through (0, made wave-uniform again), or it keeps all the query
        // words of the pass in scalar registers across the tile loop and spills them into lanes (314 v_readlane per tile)
        uint32_t off = 0;
        asm volatile("" : "+v"(off));
        const rh_cptr qc = (rh_cptr)(uintptr_t)qd + __builtin_amdgcn_readfirstlane(off) + cb * 4;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const rh_cptr q = qc + i * (4 * LPR);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const uint4 &x = v[cb + cc];
                a[i] = rh_bcnt_acc(x.x & q[cc * 4], rh_bcnt_acc(x.y & q[cc * 4 + 1], rh_bcnt_acc(x.z & q[cc * 4 + 2], rh_bcnt_acc(x.w & q[cc * 4 + 3], a[i]))));
            }
        }
    }
}
