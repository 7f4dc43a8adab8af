/* locsim.c — experiment helper (not product code): graph-locality renumbering candidates and the
 * line statistics of a group-bitmap visited table, on the CPU.  Built by scripts/locality_sim.py. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NO_SLOT 0xFFFFFFFFu

/* BFS-block order: blocks of G nodes grown breadth-first from a seed; the next seed is the oldest
 * node left in the queue of the block that just filled up (so consecutive blocks are adjacent).
 * lid[ext] = position in that order. */
void bfs_block_order(const uint32_t *adj0, uint64_t n, uint32_t cap0, uint32_t G, uint32_t *lid) {
    uint8_t *claimed = calloc(n, 1);
    uint32_t *queue = malloc((size_t)n * 4);
    uint64_t next_lid = 0, scan = 0;
    uint64_t qh = 0, qt = 0;      /* queue of claimed-but-not-expanded nodes (global FIFO) */
    while (next_lid < n) {
        if (qh == qt) {           /* component exhausted: next unclaimed node in slot order */
            while (claimed[scan]) scan++;
            claimed[scan] = 1;
            lid[scan] = (uint32_t)next_lid++;
            queue[qt++] = (uint32_t)scan;
        }
        const uint32_t u = queue[qh++];
        const uint32_t *row = adj0 + (uint64_t)u * cap0;
        for (uint32_t j = 0; j < cap0; ++j) {
            const uint32_t v = row[j];
            if (v == NO_SLOT) break;
            if (!claimed[v]) { claimed[v] = 1; lid[v] = (uint32_t)next_lid++; queue[qt++] = v; }
        }
    }
    (void)G;
    free(claimed);
    free(queue);
}

/* Block-local BFS: a block is grown from its seed breadth-first until it holds G nodes; nodes that were
 * reached but not claimed stay unclaimed.  The next seed is the first reached-but-unclaimed node of the
 * finished block (else the next unclaimed slot). */
void block_grow_order(const uint32_t *adj0, uint64_t n, uint32_t cap0, uint32_t G, uint32_t *lid) {
    uint8_t *claimed = calloc(n, 1);
    uint32_t *blk = malloc((size_t)G * 4);
    uint32_t *cand = malloc((size_t)n * 4);   /* stack of seed candidates */
    uint64_t ncand = 0;
    uint64_t next_lid = 0, scan = 0;
    while (next_lid < n) {
        uint32_t seed = NO_SLOT;
        while (ncand) { const uint32_t c = cand[--ncand]; if (!claimed[c]) { seed = c; break; } }
        if (seed == NO_SLOT) { while (claimed[scan]) scan++; seed = (uint32_t)scan; }
        uint32_t nb = 0, head = 0;
        claimed[seed] = 1; blk[nb++] = seed;
        while (head < nb && nb < G) {
            const uint32_t u = blk[head++];
            const uint32_t *row = adj0 + (uint64_t)u * cap0;
            for (uint32_t j = 0; j < cap0 && nb < G; ++j) {
                const uint32_t v = row[j];
                if (v == NO_SLOT) break;
                if (!claimed[v]) { claimed[v] = 1; blk[nb++] = v; }
            }
        }
        /* seed candidates: unclaimed neighbours of the block's unexpanded tail */
        for (uint32_t i = head; i < nb; ++i) {
            const uint32_t *row = adj0 + (uint64_t)blk[i] * cap0;
            for (uint32_t j = 0; j < cap0; ++j) {
                const uint32_t v = row[j];
                if (v == NO_SLOT) break;
                if (!claimed[v] && ncand < n) cand[ncand++] = v;
            }
        }
        /* pad the block to G ids so that blocks are aligned to groups */
        for (uint32_t i = 0; i < nb; ++i) lid[blk[i]] = (uint32_t)(next_lid + i);
        next_lid += nb;
    }
    free(claimed); free(blk); free(cand);
}

/* distinct groups (lid / G) among the valid entries of each listed row; returns the sum */
uint64_t rows_distinct_groups(const uint32_t *adj0, uint32_t cap0, const uint32_t *lid, uint32_t G,
                              const uint32_t *nodes, uint64_t n_nodes) {
    uint64_t tot = 0;
    uint32_t g[64];
    for (uint64_t i = 0; i < n_nodes; ++i) {
        const uint32_t *row = adj0 + (uint64_t)nodes[i] * cap0;
        uint32_t k = 0;
        for (uint32_t j = 0; j < cap0; ++j) {
            if (row[j] == NO_SLOT) break;
            const uint32_t gg = lid[row[j]] / G;
            uint32_t t = 0;
            for (; t < k; ++t) if (g[t] == gg) break;
            if (t == k) g[k++] = gg;
        }
        tot += k;
    }
    return tot;
}

/* depth-first order: visit the first (nearest) unvisited neighbour first; a node gets its id when it is
 * first reached.  Explicit stack of (node, next neighbour index). */
void dfs_order(const uint32_t *adj0, uint64_t n, uint32_t cap0, uint32_t G, uint32_t *lid) {
    uint8_t *seen = calloc(n, 1);
    uint32_t *st_node = malloc((size_t)n * 4);
    uint8_t *st_pos = malloc((size_t)n);
    uint64_t next_lid = 0, scan = 0, sp = 0;
    (void)G;
    while (next_lid < n) {
        if (sp == 0) {
            while (seen[scan]) scan++;
            seen[scan] = 1; lid[scan] = (uint32_t)next_lid++;
            st_node[0] = (uint32_t)scan; st_pos[0] = 0; sp = 1;
        }
        const uint32_t u = st_node[sp - 1];
        const uint32_t *row = adj0 + (uint64_t)u * cap0;
        uint32_t j = st_pos[sp - 1];
        uint32_t v = NO_SLOT;
        for (; j < cap0; ++j) {
            if (row[j] == NO_SLOT) { j = cap0; break; }
            if (!seen[row[j]]) { v = row[j]; ++j; break; }
        }
        if (v == NO_SLOT) { sp--; continue; }
        st_pos[sp - 1] = (uint8_t)j;
        seen[v] = 1; lid[v] = (uint32_t)next_lid++;
        st_node[sp] = v; st_pos[sp] = 0; sp++;
    }
    free(seen); free(st_node); free(st_pos);
}

/* block_grow_order restricted to the first K entries of every row (the nearest neighbours) */
void block_grow_order_k(const uint32_t *adj0, uint64_t n, uint32_t cap0, uint32_t G, uint32_t K, uint32_t *lid) {
    uint8_t *claimed = calloc(n, 1);
    uint32_t *blk = malloc((size_t)G * 4);
    uint32_t *cand = malloc((size_t)n * 4);
    uint64_t ncand = 0;
    uint64_t next_lid = 0, scan = 0;
    if (K > cap0) K = cap0;
    while (next_lid < n) {
        uint32_t seed = NO_SLOT;
        while (ncand) { const uint32_t c = cand[--ncand]; if (!claimed[c]) { seed = c; break; } }
        if (seed == NO_SLOT) { while (claimed[scan]) scan++; seed = (uint32_t)scan; }
        uint32_t nb = 0, head = 0;
        claimed[seed] = 1; blk[nb++] = seed;
        while (head < nb && nb < G) {
            const uint32_t u = blk[head++];
            const uint32_t *row = adj0 + (uint64_t)u * cap0;
            for (uint32_t j = 0; j < K && nb < G; ++j) {
                const uint32_t v = row[j];
                if (v == NO_SLOT) break;
                if (!claimed[v]) { claimed[v] = 1; blk[nb++] = v; }
            }
        }
        for (uint32_t i = nb; i-- > head;) {
            const uint32_t *row = adj0 + (uint64_t)blk[i] * cap0;
            for (uint32_t j = K; j-- > 0;) {
                const uint32_t v = row[j];
                if (v == NO_SLOT) continue;
                if (!claimed[v] && ncand < n) cand[ncand++] = v;
            }
        }
        for (uint32_t i = 0; i < nb; ++i) lid[blk[i]] = (uint32_t)(next_lid + i);
        next_lid += nb;
    }
    free(claimed); free(blk); free(cand);
}

/* greedy graph growing: a block takes, one at a time, the unclaimed node with the most edges from
 * (and to) the block so far; ties go to the node that entered the frontier first.  Bucket queue by
 * count with lazy deletion.  radj (reverse adjacency, CSR) is built here. */
int g_dirs = 2;
void set_dirs(int d) { g_dirs = d; }
void greedy_grow_order(const uint32_t *adj0, uint64_t n, uint32_t cap0, uint32_t G, uint32_t *lid) {
    /* reverse CSR */
    uint32_t *rdeg = calloc(n + 1, 4);
    for (uint64_t u = 0; u < n; ++u)
        for (uint32_t j = 0; j < cap0; ++j) { uint32_t v = adj0[u * cap0 + j]; if (v == NO_SLOT) break; rdeg[v + 1]++; }
    for (uint64_t i = 0; i < n; ++i) rdeg[i + 1] += rdeg[i];
    uint32_t *radj = malloc((size_t)rdeg[n] * 4 + 4);
    uint32_t *fill = malloc((size_t)n * 4);
    memcpy(fill, rdeg, (size_t)n * 4);
    for (uint64_t u = 0; u < n; ++u)
        for (uint32_t j = 0; j < cap0; ++j) { uint32_t v = adj0[u * cap0 + j]; if (v == NO_SLOT) break; radj[fill[v]++] = (uint32_t)u; }
    free(fill);
    uint8_t *claimed = calloc(n, 1);
    uint16_t *cnt = calloc(n, 2);            /* edges between v and the current block */
    uint32_t *stamp = calloc(n, 4);          /* block number that cnt[v] belongs to */
#define MAXC 64
    uint32_t *bucket[MAXC + 1]; uint32_t bn[MAXC + 1], bcap[MAXC + 1], bh[MAXC + 1];
    for (int c = 0; c <= MAXC; ++c) { bcap[c] = 1024; bucket[c] = malloc(bcap[c] * 4); bn[c] = 0; bh[c] = 0; }
    uint32_t *carry = malloc((size_t)n * 4); uint64_t ncarry = 0;   /* frontier left over: seeds for later blocks */
    uint64_t next_lid = 0, scan = 0;
    uint32_t blockno = 0;
    while (next_lid < n) {
        blockno++;
        for (int c = 0; c <= MAXC; ++c) { bn[c] = 0; bh[c] = 0; }
        uint32_t seed = NO_SLOT;
        while (ncarry) { const uint32_t c = carry[--ncarry]; if (!claimed[c]) { seed = c; break; } }
        if (seed == NO_SLOT) { while (claimed[scan]) scan++; seed = (uint32_t)scan; }
        uint32_t nb = 0;
        uint32_t cur = seed;
        int top = 0;
        for (;;) {
            claimed[cur] = 1; lid[cur] = (uint32_t)next_lid++; nb++;
            if (nb >= G) break;
            /* bump the counts of cur's out- and in-neighbours */
            for (int dir = 0; dir < g_dirs; ++dir) {
                const uint32_t *lst; uint32_t len;
                if (dir == 0) { lst = adj0 + (uint64_t)cur * cap0; len = cap0; }
                else { lst = radj + rdeg[cur]; len = rdeg[cur + 1] - rdeg[cur]; }
                for (uint32_t j = 0; j < len; ++j) {
                    const uint32_t v = lst[j];
                    if (v == NO_SLOT) break;
                    if (claimed[v]) continue;
                    if (stamp[v] != blockno) { stamp[v] = blockno; cnt[v] = 0; }
                    int c = ++cnt[v]; if (c > MAXC) c = MAXC;
                    if (bn[c] == bcap[c]) { bcap[c] *= 2; bucket[c] = realloc(bucket[c], (size_t)bcap[c] * 4); }
                    bucket[c][bn[c]++] = v;
                    if (c > top) top = c;
                }
            }
            /* pop the best live candidate */
            cur = NO_SLOT;
            while (top > 0) {
                while (bh[top] < bn[top]) {
                    const uint32_t v = bucket[top][bh[top]++];
                    int c = cnt[v] > MAXC ? MAXC : cnt[v];
                    if (!claimed[v] && stamp[v] == blockno && c == top) { cur = v; break; }
                }
                if (cur != NO_SLOT) break;
                top--;
            }
            if (cur == NO_SLOT) break;   /* component exhausted */
        }
        /* leftover frontier nodes (best first) become seed candidates */
        for (int c = 1; c <= MAXC; ++c)
            for (uint32_t i = bh[c]; i < bn[c]; ++i) {
                const uint32_t v = bucket[c][i];
                if (!claimed[v] && ncarry < n) carry[ncarry++] = v;
            }
    }
    for (int c = 0; c <= MAXC; ++c) free(bucket[c]);
    free(claimed); free(cnt); free(stamp); free(carry); free(rdeg); free(radj);
}
