// keys.hip — key <-> slot map of the index, behind the C ABI (host memory only, no device call).
//
// The reference's index is keyed: add(keys, fps) (README.md:58), get_neighbors / get_top_level_nodes
// return flat [slot, key, ...] lists (rad/hnsw_service.py:222,229; mock tests/test_redis_auth.py:37-43),
// get_node_ids_from_keys(keys) maps back (examples/DUDEZ_example.ipynb:408) and the SQLite join is on
// the key (rad/hnsw_service.py:271).  At 100M+ rows that map cannot be a Python dict: it lives here
// as the key of every slot plus, built on the first lookup, the slots ordered by key.
#include "common.h"

#include <algorithm>
#include <numeric>

static uint64_t key_count_needed(const radhip_index *idx) {
    uint64_t n = idx->has_graph ? idx->g_n : 0;
    if (idx->has_vectors && idx->n > n) n = idx->n;
    return n;
}
// slots the caller never gave a key keep the identity key (what Index.add(None, fps) assigns)
static void keys_extend(radhip_index *idx, uint64_t n) {
    const uint64_t old = idx->h_keys.size();
    if (n <= old) return;
    idx->h_keys.resize(n);
    for (uint64_t i = old; i < n; ++i) idx->h_keys[i] = i;
    idx->key_order_valid = false;
}

extern "C" int radhip_index_set_keys(radhip_index_t *idx, uint64_t first_slot, const uint64_t *keys, uint64_t n) {
    if (!idx || (!keys && n)) RH_FAIL(RADHIP_E_INVALID, "null argument");
    if (first_slot + n >= 0xFFFFFFFFull) RH_FAIL(RADHIP_E_RANGE, "slots out of range");
    std::lock_guard<std::mutex> lk(idx->mu);
    try {
        keys_extend(idx, first_slot + n);
    } catch (...) { RH_FAIL(RADHIP_E_NOMEM, "out of host memory for %llu keys", (unsigned long long)(first_slot + n)); }
    if (n) memcpy(idx->h_keys.data() + first_slot, keys, n * 8);
    idx->key_order_valid = false;
    return RADHIP_OK;
}

extern "C" int radhip_keys_from_slots(const radhip_index_t *cidx, const uint32_t *slots, uint64_t n, uint64_t *out_keys) {
    radhip_index *idx = const_cast<radhip_index *>(cidx);
    if (!idx || ((!slots || !out_keys) && n)) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    const uint64_t size = std::max<uint64_t>(key_count_needed(idx), idx->h_keys.size());
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t s = slots[i];
        if (s >= size) RH_FAIL(RADHIP_E_RANGE, "slot %u out of range (size %llu)", s, (unsigned long long)size);
        out_keys[i] = s < idx->h_keys.size() ? idx->h_keys[s] : (uint64_t)s;
    }
    return RADHIP_OK;
}

extern "C" int radhip_index_read_keys(const radhip_index_t *cidx, uint64_t first, uint64_t count, uint64_t *out_keys) {
    radhip_index *idx = const_cast<radhip_index *>(cidx);
    if (!idx || (!out_keys && count)) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    const uint64_t size = std::max<uint64_t>(key_count_needed(idx), idx->h_keys.size());
    if (first + count > size) RH_FAIL(RADHIP_E_RANGE, "slots [%llu, %llu) out of range (size %llu)", (unsigned long long)first,
                                      (unsigned long long)(first + count), (unsigned long long)size);
    for (uint64_t i = 0; i < count; ++i) out_keys[i] = first + i < idx->h_keys.size() ? idx->h_keys[first + i] : first + i;
    return RADHIP_OK;
}

extern "C" int radhip_slots_from_keys(const radhip_index_t *cidx, const uint64_t *keys, uint64_t n, uint32_t *out_slots,
                                      uint64_t *out_missing) {
    radhip_index *idx = const_cast<radhip_index *>(cidx);
    if (!idx || ((!keys || !out_slots) && n)) RH_FAIL(RADHIP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    try {
        keys_extend(idx, key_count_needed(idx));
        if (!idx->key_order_valid) {
            const uint64_t m = idx->h_keys.size();
            idx->h_key_order.resize(m);
            std::iota(idx->h_key_order.begin(), idx->h_key_order.end(), 0u);
            const uint64_t *hk = idx->h_keys.data();
            bool sorted = true;   // the common case (keys ascending with the slots) needs no sort
            for (uint64_t i = 1; i < m && sorted; ++i) sorted = hk[i - 1] <= hk[i];
            if (!sorted)
                std::sort(idx->h_key_order.begin(), idx->h_key_order.end(),
                          [hk](uint32_t a, uint32_t b) { return hk[a] != hk[b] ? hk[a] < hk[b] : a < b; });
            idx->key_order_valid = true;
        }
    } catch (...) { RH_FAIL(RADHIP_E_NOMEM, "out of host memory for the key order"); }
    const uint64_t *hk = idx->h_keys.data();
    const uint32_t *ord = idx->h_key_order.data();
    const uint64_t m = idx->h_key_order.size();
    uint64_t missing = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t k = keys[i];
        uint64_t lo = 0, hi = m;
        while (lo < hi) {   // first slot (in key order) whose key is >= k: with duplicate keys the lowest slot wins
            const uint64_t mid = (lo + hi) >> 1;
            if (hk[ord[mid]] < k) lo = mid + 1; else hi = mid;
        }
        if (lo < m && hk[ord[lo]] == k) out_slots[i] = ord[lo];
        else { out_slots[i] = RADHIP_NO_SLOT; missing++; }
    }
    if (out_missing) *out_missing = missing;
    return RADHIP_OK;
}

// [slot, key, slot, key, ...]: the list shape of the reference's index.get_neighbors
extern "C" int radhip_get_neighbors_keyed(const radhip_index_t *cidx, uint32_t slot, int32_t level, uint64_t *out_pairs,
                                          uint32_t cap_pairs, uint32_t *out_n) {
    uint32_t s[64];
    uint32_t k = 0;
    RH_TRY(radhip_get_neighbors(cidx, slot, level, s, 64, &k));
    if (out_n) *out_n = k;
    if (!out_pairs) return RADHIP_OK;
    const uint32_t w = k < cap_pairs ? k : cap_pairs;
    uint64_t keys[64];
    RH_TRY(radhip_keys_from_slots(cidx, s, w, keys));
    for (uint32_t i = 0; i < w; ++i) { out_pairs[2 * i] = s[i]; out_pairs[2 * i + 1] = keys[i]; }
    return RADHIP_OK;
}
