// phl_reftable.hip -- PHL_BUILD_REFERENCE_TABLE: reproduce the OBSERVABLE behaviour of the reference's hash
// table across its doublings (crf/lattice/lite/permutohedral.h:29-169).
//
// What is observable.  The reference numbers vertices in insertion order and finds them through an open-
// addressing table that starts at 2^15 slots and doubles when `filled >= capacity/2 - 1` (:62).  lookup()
// computes h = hash(k) % capacity BEFORE lookupOffset() may grow() the table (:101-103, :59-62), and then
// keeps probing from that stale h in the doubled table.  The ONE key in flight at a doubling is therefore
// probed from the wrong place whenever hash % (2 cap) != hash % cap: it is not found there, so splat()
// (create = true) appends a SECOND vertex with the same key, filed where later lookups (which hash with the
// new capacity) cannot see it until the next grow() re-files every entry.  Effects on the result: M is
// larger by one per such doubling, the contributions of that key are split over two vertices, and blur()'s
// neighbour lookups see only one of them.  0.3-4 % of output rows move by more than 1e-4
// (tests/golden/PIN_REPORT.json).
//
// The default build of this library implements the algorithm without that defect (every key has one
// vertex).  With PHL_BUILD_REFERENCE_TABLE the build additionally replays the reference's table on the host
// and re-labels the device lattice so that vertex numbering, duplicate vertices, the per-pixel vertex each
// lookup resolved to, and the vertices blur can see are EXACTLY the reference's -- the filter is then
// bit-identical to the reference's CPU path at any size (tests: growth_*.npz, outputs of the reference
// engine itself).
//
// How.  Everything outside the doublings is history-independent: a key that has one properly filed entry
// always resolves to it.  So the replay is event driven over the CLEAN first-touch vertex list the device
// build produced (M keys + the candidate index e = pixel*(d+1)+remainder of each first touch):
//   * creations in first-touch order keep the simulated table (same hash, same linear probe, same
//     slot-order re-filing in grow(), :122-155) in the reference's state at every moment;
//   * when `filled` reaches capacity/2 - 1 the NEXT lookup -- candidate e+1, whose key is fetched from the
//     device -- grows the table and probes from its stale slot;
//   * a key with an unreachable or duplicated entry is tracked: which vertex its lookups resolve to can
//     only change at a doubling or at its first lookup after one (then fixed: entries never move inside an
//     epoch and a probe path never contains an empty slot), so each such key gets a short list of
//     (from candidate index, vertex) segments.  At most two extra vertices per doubling.
// The device then re-labels replay[].vid (one pass), uploads the new key list, and builds its own lookup
// table from the vertices the reference's final table can reach.
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <queue>
#include <unordered_map>
#include <vector>

#include "phl_device_utils.h"

namespace {

// The simulated table.  Same hash (:109-116, size_t arithmetic), same linear probe (:65-90), same slot-order
// re-filing in grow() (:122-155).  Host-speed details that do not change what it computes (C3, M = 297 k, five
// doublings: 35 ms -> a few ms): the capacity is a power of two (2^15 doubling), so `% capacity` is a mask and the
// low 32 bits of a hash decide the slot at any capacity; a slot is 8 bytes {low half of the key's hash, vertex + 1}
// with 0 = empty, so a probe compares keys (a random access into the key array) only on a hash match, grow()
// re-files straight from the old slots without touching keys, and clearing is a memset; the two slot arrays
// (current / next) are kept per thread across builds -- most of the 35 ms were page faults on fresh 16 MB
// vectors -- and both loops prefetch their target slots ahead.
struct ref_sim {
    typedef uint64_t slot_t;
    static slot_t mk(uint64_t hfull, int32_t v) { return ((uint64_t)(uint32_t)(v + 1) << 32) | (uint32_t)hfull; }
    static int32_t vertex(slot_t s) { return (int32_t)(s >> 32) - 1; }
    int d = 0;
    uint64_t cap = (uint64_t)1 << 15;   // :36
    std::vector<slot_t> *cur, *nxt;
    slot_t *slots;
    std::vector<int16_t> keys;          // [F][d], insertion order
    int64_t F = 0;

    explicit ref_sim(int d_) : d(d_)
    {
        static thread_local std::vector<slot_t> pool[2];
        cur = &pool[0];
        nxt = &pool[1];
        if (cur->size() < cap) cur->resize((size_t)cap);
        slots = cur->data();
        memset(slots, 0, sizeof(slot_t) * (size_t)cap);
    }
    uint64_t hash(const int16_t *k) const    // :109-116, size_t arithmetic
    {
        uint64_t h = 0;
        for (int i = 0; i < d; i++) {
            h += (uint64_t)(int64_t)k[i];
            h *= 2531011;
        }
        return h;
    }
    bool need_grow() const { return (uint64_t)F >= cap / 2 - 1; }   // :62
    void prefetch(uint64_t hfull) const
    {
        if (!frozen) __builtin_prefetch(&slots[(size_t)(hfull & (cap - 1))], 1, 1);
    }

    // ---- the LAST doubling is not carried out --------------------------------------------------------------------
    // Re-filing the table at the last doubling of a build moves more entries than all earlier doublings together, and
    // nothing after it needs the table itself: no further doubling will read its slot order.  What lookups in the
    // re-filed table return follows from the OLD table (kept as it is, `slots` with `cap_old` slots):
    //  * entries are re-filed in ascending old slot index, all entries of one key from the same new home, so of a
    //    key's old entries the one with the smallest old slot index comes first on every probe path from that home
    //    and is what lookups find; a key's old entries lie on the probe paths from its home under cap_old (filed or
    //    re-filed properly) and under cap_old / 2 (the one entry a stale probe left there at the previous doubling);
    //  * a probe that starts from a STALE slot (the key in flight at this doubling, hash bit log2(cap_old) set) walks
    //    cap_old slots before it could meet the key's entries; fewer than cap_old / 2 slots are occupied, so it ends
    //    on an empty slot: not found, and a vertex it creates there is unreachable for every later lookup;
    //  * vertices created afterwards from their proper home are found by later lookups of their key, behind the
    //    key's old entries.
    bool frozen = false;
    uint64_t cap_old = 0;
    std::vector<std::pair<int32_t, bool>> post;      // vertices created after the freeze by tracked-key paths: (id, reachable)
    void freeze()
    {
        frozen = true;
        cap_old = cap;
        cap *= 2;
    }
    int32_t new_vertex(const int16_t *key, bool reachable, bool record)
    {
        keys.insert(keys.end(), key, key + d);
        if (record) post.push_back({(int32_t)F, reachable});
        return (int32_t)(F++);
    }
    // what a lookup from the key's proper home returns in the re-filed table, or -1
    int32_t frozen_find(const int16_t *key, uint64_t hfull) const
    {
        int64_t best_slot = -1;
        int32_t best = -1;
        const uint64_t homes[2] = {hfull & (cap_old - 1), hfull & (cap_old / 2 - 1)};
        for (int t = 0; t < (homes[0] == homes[1] ? 1 : 2); t++) {
            uint64_t h = homes[t];
            for (uint64_t steps = 0; steps < cap_old && slots[h]; steps++, h = (h + 1) & (cap_old - 1)) {
                const slot_t sl = slots[h];
                if ((uint32_t)sl == (uint32_t)hfull && memcmp(keys.data() + (size_t)vertex(sl) * d, key, sizeof(int16_t) * d) == 0 &&
                    (best_slot < 0 || (int64_t)h < best_slot)) {
                    best_slot = (int64_t)h;
                    best = vertex(sl);
                }
            }
        }
        if (best >= 0) return best;
        for (const auto &pv : post)
            if (pv.second && memcmp(keys.data() + (size_t)pv.first * d, key, sizeof(int16_t) * d) == 0) return pv.first;
        return -1;
    }
    void grow()                                                      // :122-155, entries re-filed in old slot order
    {
        const slot_t *old = slots;
        const uint64_t oldcap = cap;
        cap *= 2;
        if (nxt->size() < cap) nxt->resize((size_t)cap);
        std::swap(cur, nxt);
        slots = cur->data();
        memset(slots, 0, sizeof(slot_t) * (size_t)cap);
        const uint64_t mask = cap - 1;
        constexpr int B = 32;
        slot_t pend[B];
        int np = 0;
        auto flush = [&]() {
            for (int k = 0; k < np; k++) {
                uint64_t h = pend[k] & mask;
                while (slots[h]) h = (h + 1) & mask;
                slots[h] = pend[k];
            }
            np = 0;
        };
        for (uint64_t i = 0; i < oldcap; i++) {
            if (!old[i]) continue;
            __builtin_prefetch(&slots[(size_t)(old[i] & mask)], 1, 1);
            pend[np] = old[i];
            if (++np == B) flush();                                  // same order as the one-by-one loop
        }
        flush();
    }
    // linear probe from slot h (:65-90); returns the vertex or -1.  hfull = the key's hash.
    int32_t probe(const int16_t *key, uint64_t hfull, uint64_t h, bool create, bool *created)
    {
        *created = false;
        if (frozen) {
            const bool proper = h == (hfull & (cap - 1));
            const int32_t found = proper ? frozen_find(key, hfull) : -1;
            if (found >= 0 || !create) return found;
            *created = true;
            return new_vertex(key, proper, true);
        }
        const uint64_t mask = cap - 1;
        for (;;) {
            const slot_t s = slots[h];
            if (!s) {
                if (!create) return -1;
                keys.insert(keys.end(), key, key + d);
                slots[h] = mk(hfull, (int32_t)F);
                *created = true;
                return (int32_t)(F++);
            }
            if ((uint32_t)s == (uint32_t)hfull && memcmp(keys.data() + (size_t)vertex(s) * d, key, sizeof(int16_t) * d) == 0)
                return vertex(s);
            h = (h + 1) & mask;
        }
    }
    int32_t lookup(const int16_t *key, bool create, bool *created)
    {
        const uint64_t hf = hash(key);
        return probe(key, hf, hf & (cap - 1), create, created);
    }
    // the creation loop's lookup: a clean first touch, i.e. a key the table has never seen
    int32_t lookup_hashed(const int16_t *key, uint64_t hf, bool create, bool *created)
    {
        if (frozen && create) {
            *created = true;
            return new_vertex(key, true, false);
        }
        return probe(key, hf, hf & (cap - 1), create, created);
    }
};

struct dup_key {
    int clean;                                       // clean vertex id of the key
    std::vector<std::pair<int32_t, int32_t>> seg;    // (from candidate index, reference vertex), ascending
    int64_t pending_e = -1;                          // scheduled "first lookup since it became unreachable"
};

}  // namespace

// Host replay.  keys_clean [M][d] and efirst [M] (candidate index of each clean vertex's first touch,
// ascending) come from the device build; `q` answers the two questions the replay has about individual
// candidates.  See the file header.
int phl_reference_table_sim(const int16_t *keys_clean, const int32_t *efirst, int64_t M, int d, int64_t N,
                            phl_reftable_query &q, phl_reftable_result &out)
{
    const auto t_begin = std::chrono::steady_clock::now();
    double t_query = 0, t_grow = 0, t_hash = 0, t_loop = 0;
    ref_sim sim(d);
    sim.keys.reserve((size_t)(M + 64) * d);
    // hashes of the clean keys up front (a streaming pass), so that the creation loop can prefetch its slots
    std::vector<uint64_t> hclean((size_t)M);
    for (int64_t v = 0; v < M; v++) hclean[(size_t)v] = sim.hash(keys_clean + (size_t)v * d);
    constexpr int64_t AHEAD = 12;
    t_hash = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    std::vector<int32_t> primary((size_t)M, -1);
    std::vector<dup_key> dups;
    std::unordered_map<int, int> dup_index;
    using ev = std::pair<int64_t, int>;              // (candidate index, index into dups)
    std::priority_queue<ev, std::vector<ev>, std::greater<ev>> pending;
    const int64_t INF = (int64_t)1 << 62;

    auto key_of = [&](int clean) { return keys_clean + (size_t)clean * d; };
    auto dup_for = [&](int clean) -> int {
        auto it = dup_index.find(clean);
        if (it != dup_index.end()) return it->second;
        dup_key D;
        D.clean = clean;
        if (primary[clean] >= 0) D.seg.push_back({efirst[clean], primary[clean]});
        dups.push_back(D);
        dup_index[clean] = (int)dups.size() - 1;
        return (int)dups.size() - 1;
    };

    int64_t vi = 0, t_last = -1;
    for (;;) {
        if (sim.need_grow()) {
            // `filled` reached capacity/2 - 1 at the creation at t_last: the very next lookup grows the table
            const int64_t eg = t_last + 1;
            if (eg >= N) break;                       // that lookup is blur's first one: below
            const bool first_touch = vi < M && efirst[vi] == eg;
            const auto tq0 = std::chrono::steady_clock::now();
            int K = first_touch ? (int)vi : q.vid_at(eg);
            t_query += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq0).count();
            if (K < 0 || K >= M) return PHL_ERR_INVALID;
            const int16_t *key = key_of(K);
            const uint64_t hf = hclean[(size_t)K];
            const uint64_t h_old = hf % sim.cap;                // :102, computed before grow()
            const auto tg0 = std::chrono::steady_clock::now();
            // the last doubling of this build?  (the next one comes at cap - 1 vertices; duplicates add a few to M)
            const char *env_sf = getenv("PHL_REPLAY_SKIP_FINAL");       // per call: the CPU test runs both forms
            const bool skip_final = !(env_sf && atoi(env_sf) == 0);
            if (sim.frozen) return PHL_ERR_INVALID;                      // cannot happen: M + 128 < cap - 1 was checked
            if (skip_final && (uint64_t)M + 128 < sim.cap - 1) sim.freeze();
            else sim.grow();
            t_grow += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tg0).count();
            const uint64_t h_new = hf % sim.cap;
            bool created = false;
            const int32_t r = sim.probe(key, hf, h_old, true, &created);
            if (first_touch) {
                primary[K] = r;
                vi++;
            }
            if ((created && h_new != h_old) || dup_index.count(K)) {
                const int di = dup_for(K);                       // base segment: (first touch, primary)
                if (!(first_touch && dups[di].seg.size() == 1 && dups[di].seg[0].second == r))
                    dups[di].seg.push_back({(int32_t)eg, r});     // this one candidate resolved to r
                if (dups[di].pending_e == eg) dups[di].pending_e = -1;
            }
            while (!pending.empty() && pending.top().first == eg) pending.pop();   // it was this lookup
            // every tracked key: what do lookups resolve to in the re-filed table?
            for (size_t i = 0; i < dups.size(); i++) {
                bool c;
                const int32_t rr = sim.lookup(key_of(dups[i].clean), false, &c);
                if (rr >= 0) {
                    dups[i].seg.push_back({(int32_t)(eg + 1), rr});
                } else if (dups[i].pending_e <= eg) {
                    // only an unreachable copy exists: the key's next lookup will append another vertex
                    const auto tq1 = std::chrono::steady_clock::now();
                    const int64_t en = q.next_occurrence(dups[i].clean, eg);
                    t_query += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq1).count();
                    dups[i].pending_e = en;
                    if (en >= 0) pending.push({en, (int)i});
                }
            }
            t_last = eg;
            continue;
        }
        const int64_t e_c = vi < M ? (int64_t)efirst[vi] : INF;
        while (!pending.empty() && dups[pending.top().second].pending_e != pending.top().first) pending.pop();   // stale
        const int64_t e_p = pending.empty() ? INF : pending.top().first;
        if (e_c == INF && e_p == INF) break;
        bool created = false;
        if (e_c <= e_p) {
            if (vi + AHEAD < M) sim.prefetch(hclean[(size_t)(vi + AHEAD)]);
            const int32_t r = sim.lookup_hashed(key_of((int)vi), hclean[(size_t)vi], true, &created);
            if (!created) return PHL_ERR_INVALID;     // a clean first touch must be new to the table
            primary[vi] = r;
            vi++;
            t_last = e_c;
        } else {
            const int di = pending.top().second;
            pending.pop();
            const int32_t r = sim.lookup(key_of(dups[di].clean), true, &created);
            dups[di].seg.push_back({(int32_t)e_p, r});
            dups[di].pending_e = -1;
            t_last = e_p;
        }
    }

    t_loop = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    // blur()'s lookups (create = false) grow the table too when splat left it at the threshold (:62 has no
    // `create` test); the first neighbour lookup -- axis 0, vertex 0, key+1 with coordinate 0 at key-d
    // (:504-509) -- is then the one probed from a stale slot.
    out.blur_grow = false;
    out.blur_first_nbr = -1;
    if (sim.F > 0 && sim.need_grow()) {
        std::vector<int16_t> n1(d);
        for (int i = 0; i < d; i++) n1[i] = (int16_t)(sim.keys[i] + 1);
        n1[0] = (int16_t)(sim.keys[0] - d);
        const uint64_t hf = sim.hash(n1.data());
        const uint64_t h_old = hf % sim.cap;
        sim.grow();
        bool c;
        out.blur_first_nbr = sim.probe(n1.data(), hf, h_old, false, &c);
        out.blur_grow = true;
    }

    out.M_ref = sim.F;
    out.remap.swap(primary);
    out.dup_clean.clear();
    out.dup_ptr.assign(1, 0);
    out.seg_e.clear();
    out.seg_id.clear();
    out.hidden.clear();
    for (size_t i = 0; i < dups.size(); i++) {
        const dup_key &D = dups[i];
        bool c;
        const int32_t visible = sim.lookup(key_of(D.clean), false, &c);
        std::vector<int32_t> ids;
        for (auto &s : D.seg) ids.push_back(s.second);
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        for (int32_t id : ids)
            if (id != visible) out.hidden.push_back(id);
        out.remap[D.clean] = -(int32_t)(i + 1);
        out.dup_clean.push_back(D.clean);
        for (auto &s : D.seg) {
            out.seg_e.push_back(s.first);
            out.seg_id.push_back(s.second);
        }
        out.dup_ptr.push_back((int32_t)out.seg_e.size());
    }
    std::sort(out.hidden.begin(), out.hidden.end());
    out.keys.swap(sim.keys);                // (after the final lookups above, which compare keys)
    static const bool dbg = getenv("PHL_DEBUG") != nullptr;
    if (dbg)
        fprintf(stderr, "[phl] reference-table replay: M %lld -> %lld, %zu tracked keys, hashes by %.2f, loop done by %.2f (grow %.2f), total %.2f ms (of which %.2f ms in candidate queries)\n",
                (long long)M, (long long)out.M_ref, dups.size(), t_hash, t_loop, t_grow,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), t_query);
    return PHL_OK;
}

// ---- analytic replay -------------------------------------------------------------------------------------------------
// The simulation above spends its time keeping a table nobody looks at: what the build needs from it is (1) WHEN the
// doublings happen, (2) whether the key in flight at a doubling is probed from a stale slot, (3) for the handful of
// keys that ever got a second vertex, which of their vertices a lookup finds in each epoch.  All three follow from
// counting and from the ORDER of a key's entries, without slot positions, as long as no entry of any epoch's table
// wraps past the table's last slot (then "smaller slot index" and "earlier on the probe path" are the same thing):
//  * the vertex count after candidate t is (#clean first touches <= t) + (#extra creations <= t); the doubling of
//    capacity c is triggered by the lookup after the creation that brings the count to c/2 - 1 (:62);
//  * the key in flight is hashed with c before and 2c after (:101-103): a stale start iff bit log2(c) of its hash is
//    set; from a stale start the probe meets none of the key's entries (they lie c slots further on, fewer than c/2
//    slots are occupied) and appends a vertex there, unreachable for lookups until the next doubling;
//  * grow() re-files in ascending slot order (:122-155).  Entries of one key are re-filed from one home, so they keep
//    their order; the stale one of the previous doubling lies c/2 slots BELOW the key's home, hence is re-filed first
//    and from then on is what lookups find; vertices created later in an epoch are filed behind the existing ones.
// "No entry wraps" is an occupancy property, independent of the insertion order: q.no_wrap() evaluates it for the
// table at the end of every epoch (the device does it with one histogram and one ordered reduction).  If it does not
// hold, or the splat ends exactly at a threshold (the doubling inside blur()), 1 is returned and the caller simulates.
static int reference_table_fast_impl(const int16_t *keys_clean, const int32_t *efirst, int64_t M, int d, int64_t N,
                                     phl_reftable_query &q, phl_reftable_result &out, bool compact);
int phl_reference_table_fast(const int16_t *keys_clean, const int32_t *efirst, int64_t M, int d, int64_t N,
                             phl_reftable_query &q, phl_reftable_result &out, bool compact)
{
    const int rc = reference_table_fast_impl(keys_clean, efirst, M, d, N, q, out, compact);
    // the occupancy questions were answered behind the replay's back: nothing of it holds if one came back "no"
    if (q.probe_paths_all_ok() != 1) {
        if (getenv("PHL_DEBUG")) fprintf(stderr, "[phl] analytic replay: a tracked key's probe path wraps\n");
        return 1;
    }
    return rc;
}

static int reference_table_fast_impl(const int16_t *keys_clean, const int32_t *efirst, int64_t M, int d, int64_t N,
                                     phl_reftable_query &q, phl_reftable_result &out, bool compact)
{
    struct tracked {
        int clean;
        std::vector<int32_t> P;                          // reachable vertices in probe order
        int32_t S = -1;                                  // vertex filed from a stale slot in the current epoch
        int64_t pending_e = -1;
        std::vector<std::pair<int32_t, int32_t>> seg;    // (from candidate, vertex)
    };
    std::vector<tracked> tk;
    std::unordered_map<int, int> tindex;
    struct extra_t { int64_t e; int clean; int32_t id; };
    std::vector<extra_t> extras;                         // creations that are not clean first touches, ascending e
    auto hash_of = [&](int clean) {
        if (!keys_clean) return q.key_hash(clean);       // keys left on the device: asked for (a handful of vertices)
        uint64_t h = 0;
        const int16_t *k = keys_clean + (size_t)clean * d;
        for (int i = 0; i < d; i++) { h += (uint64_t)(int64_t)k[i]; h *= 2531011; }
        return h;
    };
    {
        // The questions of an undisturbed replay are known in advance: at the doubling behind the T-th creation the
        // candidate right after that creation's first touch, for up to three extra creations before it.  One round trip.
        std::vector<int64_t> pc;
        std::vector<int32_t> pk;
        for (uint64_t c = (uint64_t)1 << 15; (int64_t)(c / 2 - 1) <= M + 8 && pc.size() < 28; c <<= 1)
            for (int x = 0; x < 4; x++) {
                const int64_t idx = (int64_t)(c / 2 - 1) - 1 - x;
                if (idx < 0 || idx >= M) continue;
                const int64_t e = (int64_t)efirst[idx] + 1;
                if (e >= N) continue;
                pc.push_back(e);
                pk.push_back(idx + 1 < M && efirst[idx + 1] == e ? (int32_t)(idx + 1) : -1);
            }
        if (!pc.empty()) q.prefetch(pc, pk);
    }
    auto clean_before = [&](int64_t e) { return (int64_t)(std::lower_bound(efirst, efirst + M, (int32_t)std::min<int64_t>(e, 0x7FFFFFFF)) - efirst); };
    // reference id of clean vertex v: v plus the extra creations before its first touch
    auto ref_id = [&](int v) {
        int64_t k = 0;
        for (const extra_t &x : extras) k += x.e < efirst[v];
        return (int32_t)(v + k);
    };
    auto track = [&](int clean, bool exists) -> int {
        auto it = tindex.find(clean);
        if (it != tindex.end()) return it->second;
        tracked t;
        t.clean = clean;
        if (exists) {
            const int32_t id = ref_id(clean);
            t.seg.push_back({efirst[clean], id});
            t.P.push_back(id);
        }
        tk.push_back(t);
        tindex[clean] = (int)tk.size() - 1;
        return (int)tk.size() - 1;
    };
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    double t_check = 0, t_query = 0;
    uint64_t cap = (uint64_t)1 << 15;
    int64_t vi = 0;                                       // clean first touches consumed
    int64_t F = 0;                                        // vertices so far
    for (;;) {
        // creations up to the next doubling: clean first touches in candidate order, pending creations in between
        const int64_t T = (int64_t)(cap / 2 - 1);
        int64_t t_last = -1;
        bool reached = F >= T;
        while (!reached) {
            int pi = -1;
            for (size_t i = 0; i < tk.size(); i++)
                if (tk[i].pending_e >= 0 && (pi < 0 || tk[i].pending_e < tk[(size_t)pi].pending_e)) pi = (int)i;
            const int64_t need = T - F;
            const int64_t e_clean = vi + need - 1 < M ? (int64_t)efirst[vi + need - 1] : ((int64_t)1 << 62);
            if (pi < 0 || e_clean < tk[(size_t)pi].pending_e) {
                if (vi + need - 1 >= M) break;            // the splat ends before the next doubling
                vi += need;
                F += need;
                t_last = e_clean;
                reached = true;
            } else {
                const int64_t ep = tk[(size_t)pi].pending_e;
                const int64_t k = clean_before(ep) - vi;  // clean first touches before the pending creation
                vi += k;
                F += k;
                tracked &t = tk[(size_t)pi];
                const int32_t r = (int32_t)F++;
                extras.push_back({ep, t.clean, r});
                t.P.assign(1, r);
                t.seg.push_back({(int32_t)ep, r});
                t.pending_e = -1;
                if (F >= T) { t_last = ep; reached = true; }
            }
        }
        if (!reached) break;
        const int64_t eg = t_last + 1;
        if (eg >= N) { if (getenv("PHL_DEBUG")) fprintf(stderr, "[phl] analytic replay: doubling inside blur\n"); return 1; }   // simulate
        {
            // the order of a key's entries is read off their probe path: that needs the path not to wrap past the
            // table's last slot -- asked for every key that has more than one entry
            std::vector<int32_t> ex, st, chk;
            for (const extra_t &x : extras) ex.push_back(x.clean);
            for (const tracked &t : tk) {
                if (t.S >= 0) st.push_back(t.clean);      // filed from the home under cap / 2, not from its own
                if (t.P.size() + (t.S >= 0 ? 1 : 0) >= 2) chk.push_back(t.clean);
            }
            const auto tc0 = std::chrono::steady_clock::now();
            if (!chk.empty()) q.probe_paths_submit(vi, ex, st, cap, chk);
            t_check += ms_since(tc0);
        }
        const bool first_touch = vi < M && efirst[vi] == eg;
        const auto tq0 = std::chrono::steady_clock::now();
        const int K = first_touch ? (int)vi : q.vid_at(eg);
        t_query += ms_since(tq0);
        if (K < 0 || K >= M) return PHL_ERR_INVALID;
        const uint64_t hf = hash_of(K);
        const bool stale = (hf & cap) != 0;
        cap *= 2;
        for (tracked &t : tk)                             // re-filing: the stale entry of the last epoch comes first now
            if (t.S >= 0) {
                t.P.insert(t.P.begin(), t.S);
                t.S = -1;
                t.pending_e = -1;                         // ... and is found by the key's next lookup: nothing to append
            }
        if (!stale) {
            if (first_touch) {                            // an ordinary creation
                vi++;
                F++;
            } else if (tindex.count(K)) {
                tracked &t = tk[(size_t)tindex[K]];
                int32_t r;
                if (!t.P.empty()) r = t.P[0];
                else {                                    // nothing reachable: this lookup appends a vertex
                    r = (int32_t)F++;
                    extras.push_back({eg, K, r});
                    t.P.assign(1, r);
                }
                t.seg.push_back({(int32_t)eg, r});
                if (t.pending_e == eg) t.pending_e = -1;
            }
        } else {
            const int32_t r = (int32_t)F++;
            const int ti = track(K, !first_touch);
            tracked &t = tk[(size_t)ti];
            if (first_touch) {
                vi++;                                     // the clean vertex itself, filed from the stale slot
                t.seg.push_back({(int32_t)eg, r});
            } else {
                extras.push_back({eg, K, r});
                t.seg.push_back({(int32_t)eg, r});
            }
            t.S = r;
            if (t.pending_e == eg) t.pending_e = -1;
        }
        for (tracked &t : tk) {
            if (!t.P.empty()) {
                t.seg.push_back({(int32_t)(eg + 1), t.P[0]});
            } else if (t.pending_e <= eg) {
                const auto tq1 = std::chrono::steady_clock::now();
                t.pending_e = q.next_occurrence(t.clean, eg);
                t_query += ms_since(tq1);
            }
        }
    }
    // creations after the last doubling
    for (;;) {
        int pi = -1;
        for (size_t i = 0; i < tk.size(); i++)
            if (tk[i].pending_e >= 0 && (pi < 0 || tk[i].pending_e < tk[(size_t)pi].pending_e)) pi = (int)i;
        if (pi < 0) break;
        tracked &t = tk[(size_t)pi];
        const int64_t ep = t.pending_e;
        const int64_t k = clean_before(ep) - vi;
        vi += k;
        F += k;
        const int32_t r = (int32_t)F++;
        extras.push_back({ep, t.clean, r});
        t.P.assign(1, r);
        t.seg.push_back({(int32_t)ep, r});
        t.pending_e = -1;
    }
    F += M - vi;
    if ((uint64_t)F >= cap / 2 - 1) { if (getenv("PHL_DEBUG")) fprintf(stderr, "[phl] analytic replay: splat ends at a threshold\n"); return 1; }   // blur() would double the table: simulate

    // ---- results in the simulation's format ----
    out.M_ref = F;
    out.blur_grow = false;
    out.blur_first_nbr = -1;
    std::sort(extras.begin(), extras.end(), [](const extra_t &a, const extra_t &b) { return a.e < b.e; });
    out.compact = compact;
    out.ex_id.clear();
    out.ex_clean.clear();
    out.keys.clear();
    out.remap.clear();
    if (compact) {
        // reference order = clean first touches and extra creations merged by candidate index: all the caller needs
        // is where the extras sit
        for (size_t xi = 0; xi < extras.size(); xi++) {
            if (extras[xi].id != clean_before(extras[xi].e) + (int64_t)xi) return PHL_ERR_INVALID;
            out.ex_id.push_back(extras[xi].id);
            out.ex_clean.push_back(extras[xi].clean);
        }
        if (F != M + (int64_t)extras.size()) return PHL_ERR_INVALID;
    } else {
        out.keys.resize((size_t)F * d);
        out.remap.resize((size_t)M);
        // whole runs of clean vertices between two extras are copied (keys) and numbered (remap) in one go
        int64_t v = 0, id = 0;
        for (size_t xi = 0; xi <= extras.size(); xi++) {
            const int64_t vend = xi < extras.size() ? clean_before(extras[xi].e) : M;     // clean vertices before this extra
            if (vend > v) {
                memcpy(out.keys.data() + (size_t)id * d, keys_clean + (size_t)v * d, sizeof(int16_t) * (size_t)(vend - v) * d);
                int32_t *rm = out.remap.data() + v;
                const int32_t off = (int32_t)(id - v);
                for (int64_t k = 0; k < vend - v; k++) rm[k] = (int32_t)(v + k) + off;
                id += vend - v;
                v = vend;
            }
            if (xi < extras.size()) {
                if (extras[xi].id != id) return PHL_ERR_INVALID;
                memcpy(out.keys.data() + (size_t)id * d, keys_clean + (size_t)extras[xi].clean * d, sizeof(int16_t) * d);
                id++;
            }
        }
        if (id != F || v != M) return PHL_ERR_INVALID;
    }
    out.dup_clean.clear();
    out.dup_ptr.assign(1, 0);
    out.seg_e.clear();
    out.seg_id.clear();
    out.hidden.clear();
    for (size_t i = 0; i < tk.size(); i++) {
        const tracked &t = tk[i];
        const int32_t visible = t.P.empty() ? -1 : t.P[0];
        std::vector<int32_t> ids;
        for (auto &sg : t.seg) ids.push_back(sg.second);
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        for (int32_t id : ids)
            if (id != visible) out.hidden.push_back(id);
        if (!compact) out.remap[(size_t)t.clean] = -(int32_t)(i + 1);
        out.dup_clean.push_back(t.clean);
        for (auto &sg : t.seg) {
            out.seg_e.push_back(sg.first);
            out.seg_id.push_back(sg.second);
        }
        out.dup_ptr.push_back((int32_t)out.seg_e.size());
    }
    std::sort(out.hidden.begin(), out.hidden.end());
    if (getenv("PHL_DEBUG"))
        fprintf(stderr, "[phl] analytic replay: M %lld -> %lld, %zu tracked keys, %zu extra creations; %.2f ms (occupancy checks %.2f, candidate queries %.2f)\n",
                (long long)M, (long long)F, tk.size(), extras.size(), ms_since(t_begin), t_check, t_query);
    return PHL_OK;
}

// keys / remap of a compact result, on the host (tests; more extras than the device kernels take)
void phl_reftable_expand(const int16_t *keys_clean, int64_t M, int d, phl_reftable_result &R)
{
    if (!R.compact) return;
    const size_t nx = R.ex_id.size();
    R.keys.resize((size_t)R.M_ref * d);
    R.remap.resize((size_t)M);
    size_t j = 0;
    for (int64_t r = 0; r < R.M_ref; r++) {
        int64_t src;
        if (j < nx && R.ex_id[j] == r) src = R.ex_clean[j++];
        else src = r - (int64_t)j;
        memcpy(R.keys.data() + (size_t)r * d, keys_clean + (size_t)src * d, sizeof(int16_t) * d);
    }
    j = 0;
    for (int64_t v = 0; v < M; v++) {
        while (j < nx && (int64_t)R.ex_id[j] - (int64_t)j <= v) j++;     // extras created before clean vertex v
        R.remap[(size_t)v] = (int32_t)(v + (int64_t)j);
    }
    for (size_t i = 0; i < R.dup_clean.size(); i++) R.remap[(size_t)R.dup_clean[i]] = -(int32_t)(i + 1);
    R.compact = false;
}

namespace {

// ---- device side ---------------------------------------------------------------------------------
// candidates are known by their slot in the build's key table (phl_build_device: bt_slot_of / bt_table); a vertex id
// is not written per candidate before the very end (k_final_vid)
__global__ void k_vid_at(const int *__restrict__ table, const int *__restrict__ slot_of, int e, int *__restrict__ out)
{
    *out = -(table[slot_of[e]] + 1);
}

// several questions in one launch: the clean vertex of a candidate (or a vertex the host knows already) and the hash of
// its key (permutohedral.h:109-116), written where the host reads them
struct vid_hash_q {
    int n;
    int cand[32], known[32];
};
__global__ void k_vid_hash_many(const int *__restrict__ table, const int *__restrict__ slot_of, const int16_t *__restrict__ vkeys,
                                int d, vid_hash_q qs, int *__restrict__ out /* [n][3]: vertex, hash lo, hash hi */)
{
    const int i = threadIdx.x;
    if (i >= qs.n) return;
    const int v = qs.known[i] >= 0 ? qs.known[i] : -(table[slot_of[qs.cand[i]]] + 1);
    unsigned long long h = 0;
    if (v >= 0)
        for (int k = 0; k < d; k++) {
            h += (unsigned long long)(long long)vkeys[(size_t)v * d + k];
            h *= 2531011ull;
        }
    out[3 * i] = v;
    out[3 * i + 1] = (int)(unsigned)(h & 0xFFFFFFFFull);
    out[3 * i + 2] = (int)(unsigned)(h >> 32);
}

// The questions of an undisturbed replay, asked before the host has seen anything: slot (j, x) = the doubling behind the
// (2^(14+j) - 1)-th creation with x extra creations before it -> {candidate right after that creation's first touch, its
// clean vertex, hash of that vertex's key} (candidate -1: no such doubling).  The host reads them with the first touches.
constexpr int SPEC_J = 8, SPEC_X = 4;
__global__ void k_spec_many(const int *__restrict__ efirst, int M, int N, const int *__restrict__ table,
                            const int *__restrict__ slot_of, const int16_t *__restrict__ vkeys, int d, int *__restrict__ out /* [J*X][4] */)
{
    const int i = threadIdx.x;
    if (i >= SPEC_J * SPEC_X) return;
    const int j = i / SPEC_X, x = i % SPEC_X;
    const long long T = ((long long)1 << (14 + j)) - 1;
    const long long idx = T - 1 - x;
    int e = -1, v = -1;
    unsigned long long h = 0;
    if (T <= (long long)M + 8 && idx >= 0 && idx < M) {
        const long long ee = (long long)efirst[idx] + 1;
        if (ee < N) {
            e = (int)ee;
            v = (idx + 1 < M && efirst[idx + 1] == e) ? (int)(idx + 1) : -(table[slot_of[e]] + 1);
            if (v >= 0)
                for (int k = 0; k < d; k++) {
                    h += (unsigned long long)(long long)vkeys[(size_t)v * d + k];
                    h *= 2531011ull;
                }
        }
    }
    out[4 * i] = e;
    out[4 * i + 1] = v;
    out[4 * i + 2] = (int)(unsigned)(h & 0xFFFFFFFFull);
    out[4 * i + 3] = (int)(unsigned)(h >> 32);
}

// first candidate after `after` with the same key as candidate e_first (= the same table slot)
__global__ __launch_bounds__(256) void k_next_occurrence(const int *__restrict__ slot_of, int N, int e_first, int after,
                                                         int *__restrict__ out)
{
    const int slot = slot_of[e_first];
    int best = 0x7FFFFFFF;
    for (int e = after + 1 + blockIdx.x * blockDim.x + threadIdx.x; e < N; e += gridDim.x * blockDim.x)
        if (slot_of[e] == slot && e < best) best = e;
    for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
    if ((threadIdx.x & 63) == 0 && best != 0x7FFFFFFF) atomicMin(out, best);
}

// first-touch candidate of every REFERENCE vertex (the home cell of the locality renumbering is read from it):
// a vertex with one key keeps its clean first touch; the few vertices of duplicated keys get theirs from the host
__global__ __launch_bounds__(256) void k_vfirst_ref(const int *__restrict__ remap, const int *__restrict__ efirst, int M,
                                                    int *__restrict__ vfirst)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= M) return;
    const int r = remap[v];
    if (r >= 0) vfirst[r] = efirst[v];
}

// Compact replay result -> remap / keys on the device: the reference order is the clean order with a few extra
// creations inserted (ex.id ascending; extra k repeats clean vertex ex.clean[k]).
constexpr int PHL_MAX_EXTRAS = 96;
struct extras_t {
    int n;
    int32_t id[PHL_MAX_EXTRAS], clean[PHL_MAX_EXTRAS];
};

__global__ __launch_bounds__(256) void k_ref_remap(int M, extras_t ex, int *__restrict__ remap)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= M) return;
    int j = 0;
    while (j < ex.n && ex.id[j] - j <= v) j++;            // extras created before clean vertex v
    remap[v] = v + j;
}

__global__ __launch_bounds__(256) void k_ref_keys(const int16_t *__restrict__ vkeys, int d, int M_ref, extras_t ex,
                                                  int16_t *__restrict__ out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M_ref) return;
    int j = 0;
    while (j < ex.n && ex.id[j] < r) j++;
    const int src = (j < ex.n && ex.id[j] == r) ? ex.clean[j] : r - j;
    for (int i = 0; i < d; i++) out[(int64_t)r * d + i] = vkeys[(int64_t)src * d + i];
}

__global__ void k_set_pairs(const int *__restrict__ idx, const int *__restrict__ val, int k, int *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) out[idx[i]] = val[i];
}

// ---- occupancy of the reference's table, for the analytic replay's one assumption ---------------------------------
// hist[x] = entries whose home is slot x (hash as permutohedral.h:109-116, size_t arithmetic; capacity a power of two)
// (k_home_hist_add, below cl_homes)

// Is there an empty slot in [home, cap) for every key to check?  Linear probing fills slots from the left:
// carry(x+1) = max(0, carry(x) + hist[x] - 1) entries arrive at slot x+1 still looking for a slot, and slot x is empty
// iff carry(x) + hist[x] == 0.  The carry at slot 0 is what wraps around; two sweeps settle it.  ONE workgroup: every
// thread folds its contiguous piece of the table into a map c -> max(m, c + s); the 1024 maps are chained (twice for
// the wrap, once more for every piece's incoming carry); then every thread walks its piece with its real carry and
// the workgroup keeps the LAST empty slot of the table: a key passes iff its home is not behind it (k_cluster_verdict).
// (Used above 2^22 slots; below, k_cluster_fold / k_cluster_walk spread the same computation over the chip.)
// (hist is read 16 bytes at a time, eight loads in flight: the walk is latency-bound otherwise -- 276 us at 2^19 slots.)
__global__ __launch_bounds__(1024) void k_cluster_check(const int *__restrict__ hist, uint32_t cap, int *__restrict__ last_empty_out)
{
    __shared__ int sm[1024], ss[1024], cin[1024], gm[64], gs[64], gc[64];
    __shared__ int last_empty;
    const uint32_t piece = cap / 1024u;                    // cap >= 2^15: a multiple of 32
    const uint32_t x0 = threadIdx.x * piece;
    const int4 *__restrict__ hp = reinterpret_cast<const int4 *>(hist + x0);
    const uint32_t nq = piece / 4u;
    int m = INT_MIN / 4, s = 0;                            // identity map
    for (uint32_t q = 0; q < nq; q += 8) {
        int4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = hp[q + j];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int a[4] = {v[j].x - 1, v[j].y - 1, v[j].z - 1, v[j].w - 1};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                m = max(0, m + a[k]);
                s += a[k];
            }
        }
    }
    sm[threadIdx.x] = m;
    ss[threadIdx.x] = s;
    if (threadIdx.x == 0) last_empty = -1;
    __syncthreads();
    // chain the 1024 maps in two levels (maps compose: (m2, s2) after (m1, s1) = (max(m2, m1 + s2), s1 + s2)):
    // 64 threads fold 16 maps each, one thread chains the 64 results, the 64 threads hand every piece its carry
    const int t0 = (int)threadIdx.x * 16;
    if (threadIdx.x < 64) {
        int a = INT_MIN / 4, b = 0;
        for (int j = 0; j < 16; j++) {
            a = max(sm[t0 + j], a + ss[t0 + j]);
            b += ss[t0 + j];
        }
        gm[threadIdx.x] = a;
        gs[threadIdx.x] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
        for (int pass = 0; pass < 2; pass++)
            for (int t = 0; t < 64; t++) c = max(gm[t], c + gs[t]);
        for (int t = 0; t < 64; t++) {                     // c = the carry that wraps into slot 0
            gc[t] = c;
            c = max(gm[t], c + gs[t]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        int c = gc[threadIdx.x];
        for (int j = 0; j < 16; j++) {
            cin[t0 + j] = c;
            c = max(sm[t0 + j], c + ss[t0 + j]);
        }
    }
    __syncthreads();
    int c = cin[threadIdx.x];
    int last = -1;
    for (uint32_t q = 0; q < nq; q += 8) {
        int4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = hp[q + j];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int a[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (c + a[k] == 0) last = (int)(x0 + (q + (uint32_t)j) * 4u + (uint32_t)k);
                c = max(0, c + a[k] - 1);
            }
        }
    }
    if (last >= 0) atomicMax(&last_empty, last);
    __syncthreads();
    if (threadIdx.x == 0) *last_empty_out = last_empty + 1;      // (+1, 0 = no empty slot: as k_cluster_walk)
}

// The same question with the table spread over many workgroups (cap <= 2^22): k_cluster_fold reduces every 4096-slot
// piece to one map, k_cluster_walk chains the pieces' maps (a few hundred at most, by one thread), repeats the fold inside
// its piece with the real incoming carry and keeps the table's last empty slot (+1, 0 = none) in *last_empty.
constexpr int CL_PIECE = 4096;           // 256 threads x 16 slots

struct cl_map { int m, s; };             // c -> max(m, c + s)

__device__ __forceinline__ cl_map cl_fold16(const int *__restrict__ hist, uint32_t x0)
{
    const int4 *__restrict__ hp = reinterpret_cast<const int4 *>(hist + x0);
    int4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = hp[j];
    cl_map r = {INT_MIN / 4, 0};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int a[4] = {v[j].x - 1, v[j].y - 1, v[j].z - 1, v[j].w - 1};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            r.m = max(0, r.m + a[k]);
            r.s += a[k];
        }
    }
    return r;
}

// the 256 thread maps of a workgroup, chained in two levels; returns (in sm/ss) nothing, fills gm/gs[16] group maps
__device__ __forceinline__ void cl_group_maps(const cl_map mine, int *sm, int *ss, int *gm, int *gs)
{
    sm[threadIdx.x] = mine.m;
    ss[threadIdx.x] = mine.s;
    __syncthreads();
    if (threadIdx.x < 16) {
        int a = INT_MIN / 4, b = 0;
        for (int j = 0; j < 16; j++) {
            a = max(sm[threadIdx.x * 16 + j], a + ss[threadIdx.x * 16 + j]);
            b += ss[threadIdx.x * 16 + j];
        }
        gm[threadIdx.x] = a;
        gs[threadIdx.x] = b;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_cluster_fold(const int *__restrict__ hist, int2 *__restrict__ piece_map)
{
    __shared__ int sm[256], ss[256], gm[16], gs[16];
    const cl_map mine = cl_fold16(hist, (uint32_t)blockIdx.x * CL_PIECE + threadIdx.x * 16u);
    cl_group_maps(mine, sm, ss, gm, gs);
    if (threadIdx.x == 0) {
        int a = INT_MIN / 4, b = 0;
        for (int j = 0; j < 16; j++) {
            a = max(gm[j], a + gs[j]);
            b += gs[j];
        }
        piece_map[blockIdx.x] = make_int2(a, b);
    }
}

__global__ __launch_bounds__(256) void k_cluster_walk(const int *__restrict__ hist, const int2 *__restrict__ piece_map, int npieces,
                                                      int *__restrict__ last_empty)
{
    __shared__ int sm[256], ss[256], gm[16], gs[16], gc[16], cin[256];
    __shared__ int piece_carry;
    const uint32_t x0 = (uint32_t)blockIdx.x * CL_PIECE + threadIdx.x * 16u;
    const cl_map mine = cl_fold16(hist, x0);
    __shared__ int2 pm[1024];                              // npieces <= 2^22 / CL_PIECE
    for (int t = threadIdx.x; t < npieces; t += 256) pm[t] = piece_map[t];
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;                                         // the carry that wraps into slot 0: two sweeps settle it
        for (int pass = 0; pass < 2; pass++)
            for (int t = 0; t < npieces; t++) c = max(pm[t].x, c + pm[t].y);
        for (int t = 0; t < (int)blockIdx.x; t++) c = max(pm[t].x, c + pm[t].y);
        piece_carry = c;
    }
    cl_group_maps(mine, sm, ss, gm, gs);                   // (its barriers publish piece_carry as well)
    if (threadIdx.x == 0) {
        int c = piece_carry;
        for (int j = 0; j < 16; j++) {
            gc[j] = c;
            c = max(gm[j], c + gs[j]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        int c = gc[threadIdx.x];
        for (int j = 0; j < 16; j++) {
            cin[threadIdx.x * 16 + j] = c;
            c = max(sm[threadIdx.x * 16 + j], c + ss[threadIdx.x * 16 + j]);
        }
    }
    __syncthreads();
    int c = cin[threadIdx.x];
    int last = 0;
    const int4 *__restrict__ hp = reinterpret_cast<const int4 *>(hist + x0);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int4 v = hp[j];
        const int a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (c + a[k] == 0) last = (int)(x0 + (uint32_t)(4 * j + k)) + 1;
            c = max(0, c + a[k] - 1);
        }
    }
    for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o));
    if ((threadIdx.x & 63) == 0 && last > 0) atomicMax(last_empty, last);
}

// homes of the entries filed in addition to the clean vertices' own, and of the keys asked about (kernel arguments)
struct cl_homes {
    int n_add, n_chk;
    int add[128], chk[64];
};

// The table's entries by home slot, and the extra homes of a question, in one launch (both only add into hist): the LAST
// workgroup adds the extra homes
__global__ __launch_bounds__(256) void k_home_hist_add(const int16_t *__restrict__ vkeys, int64_t count, int d, uint32_t mask,
                                                       cl_homes hm, int *__restrict__ hist)
{
    if (blockIdx.x == gridDim.x - 1) {
        if ((int)threadIdx.x < hm.n_add) atomicAdd(&hist[hm.add[threadIdx.x]], 1);     // (n_add <= 128)
        return;
    }
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= count) return;
    unsigned long long h = 0;
    for (int i = 0; i < d; i++) {
        h += (unsigned long long)(long long)vkeys[v * d + i];
        h *= 2531011ull;
    }
    atomicAdd(&hist[(uint32_t)h & mask], 1);
}

// verdict of one question: every asked key's home at or before the last empty slot
__global__ void k_cluster_verdict(int *__restrict__ last_empty, cl_homes h, int *__restrict__ verdict)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int le = *last_empty - 1;
    int ok = 1;
    for (int k = 0; k < h.n_chk; k++) ok &= (h.chk[k] <= le) ? 1 : 0;
    *verdict = ok;
    *last_empty = 0;            // (for the next question on this stream: saves its memset launch)
}

struct device_query : phl_reftable_query {
    const int16_t *vkeys_dev = nullptr;      // clean keys, clean order (device)
    const int16_t *keys_host = nullptr;      // the same on the host, or null: hashes are fetched (key_hash)
    int d = 0;
    int *hist = nullptr;                     // [max capacity] device
    int *small = nullptr;                    // [256] device ints: homes in, results out
    bool last_empty_zeroed = false;
    std::unordered_map<int, uint64_t> hcache;            // clean vertex -> hash of its key
    std::unordered_map<int64_t, int> vcache;             // candidate -> clean vertex
    int *answers = nullptr;                              // [32][3] ints the device can write and the host can read, or null
    uint64_t hash_of(int clean)
    {
        if (!keys_host) return key_hash(clean);
        uint64_t h = 0;
        for (int i = 0; i < d; i++) { h += (uint64_t)(int64_t)keys_host[(size_t)clean * d + i]; h *= 2531011; }
        return h;
    }
    // one launch, one synchronisation for up to 32 questions
    bool ask(const vid_hash_q &qs)
    {
        int host[96];
        int *dst = answers ? answers : small;
        hipLaunchKernelGGL(k_vid_hash_many, dim3(1), dim3(32), 0, st, table, slot_of, vkeys_dev, d, qs, dst);
        hipError_t r = hipGetLastError();
        const int *res = answers;
        if (r == hipSuccess && !answers) {
            r = hipMemcpyAsync(host, small, sizeof(int) * 3 * (size_t)qs.n, hipMemcpyDeviceToHost, st);
            res = host;
        }
        if (r == hipSuccess) r = hipStreamSynchronize(st);
        if (r != hipSuccess) { err = r; return false; }
        for (int i = 0; i < qs.n; i++) {
            const int v = res[3 * i];
            if (qs.known[i] < 0) vcache[qs.cand[i]] = v;
            if (v >= 0) hcache[v] = (uint64_t)(unsigned)res[3 * i + 1] | ((uint64_t)(unsigned)res[3 * i + 2] << 32);
        }
        return true;
    }
    void prefetch(const std::vector<int64_t> &cands, const std::vector<int32_t> &known) override
    {
        if (!small || !table || !slot_of) return;
        vid_hash_q qs;
        qs.n = 0;
        for (size_t i = 0; i < cands.size() && qs.n < 32; i++) {
            if (known[i] >= 0 ? hcache.count(known[i]) != 0 : vcache.count(cands[i]) != 0) continue;     // answered already
            qs.cand[qs.n] = (int)cands[i];
            qs.known[qs.n] = known[i];
            qs.n++;
        }
        if (qs.n) (void)ask(qs);
    }
    uint64_t key_hash(int clean) override
    {
        auto it = hcache.find(clean);
        if (it != hcache.end()) return it->second;
        vid_hash_q qs;
        qs.n = 1;
        qs.cand[0] = 0;
        qs.known[0] = clean;
        if (!small || !ask(qs)) return 0;
        return hcache[clean];
    }
    // answers land in `verdicts` (pinned host memory where there is some, else device memory read back at the end)
    int *verdicts = nullptr;                 // [MAX_Q]
    bool verdicts_on_host = false;
    int2 *piece_map = nullptr;               // [max capacity / CL_PIECE] device
    int n_submitted = 0;
    bool cannot_tell = false;
    static constexpr int MAX_Q = 32;
    void probe_paths_submit(int64_t n_clean, const std::vector<int32_t> &extra_clean, const std::vector<int32_t> &stale_clean,
                            uint64_t cap, const std::vector<int32_t> &check) override
    {
        if (!hist || !small || !verdicts || cap > ((uint64_t)1 << 27) || check.size() > 64 ||
            extra_clean.size() + stale_clean.size() > 128 || n_submitted >= MAX_Q) {
            cannot_tell = true;
            return;
        }
        cl_homes h;
        h.n_add = 0;
        for (int32_t v : extra_clean) h.add[h.n_add++] = (int)(hash_of(v) & (cap - 1));
        for (int32_t v : stale_clean) h.add[h.n_add++] = (int)(hash_of(v) & (cap / 2 - 1));
        h.n_chk = 0;
        for (int32_t v : check) h.chk[h.n_chk++] = (int)(hash_of(v) & (cap - 1));
        int *const last_empty = small + 200; // one device int (behind the 96 ints `ask` may use); zeroed once, then by k_cluster_verdict
        hipError_t r = hipMemsetAsync(hist, 0, sizeof(int) * (size_t)cap, st);
        if (r == hipSuccess && !last_empty_zeroed) {
            r = hipMemsetAsync(last_empty, 0, sizeof(int), st);
            last_empty_zeroed = (r == hipSuccess);
        }
        if (r == hipSuccess) {
            if (n_clean > 0 || h.n_add > 0)
                hipLaunchKernelGGL(k_home_hist_add, dim3((unsigned)((n_clean + 255) / 256 + 1)), dim3(256), 0, st, vkeys_dev, n_clean, d,
                                   (uint32_t)(cap - 1), h, hist);
            if (piece_map && cap <= ((uint64_t)1 << 22)) {
                const int np = (int)(cap / CL_PIECE);
                hipLaunchKernelGGL(k_cluster_fold, dim3(np), dim3(256), 0, st, hist, piece_map);
                hipLaunchKernelGGL(k_cluster_walk, dim3(np), dim3(256), 0, st, hist, piece_map, np, last_empty);
            } else {
                hipLaunchKernelGGL(k_cluster_check, dim3(1), dim3(1024), 0, st, hist, (uint32_t)cap, last_empty);
            }
            hipLaunchKernelGGL(k_cluster_verdict, dim3(1), dim3(64), 0, st, last_empty, h, verdicts + n_submitted);
            r = hipGetLastError();
        }
        if (r != hipSuccess) { err = r; cannot_tell = true; return; }
        n_submitted++;
    }
    int probe_paths_all_ok() override
    {
        if (cannot_tell) return 0;
        if (n_submitted == 0) return 1;
        int host[MAX_Q];
        const int *v = verdicts;
        hipError_t r = hipSuccess;
        if (!verdicts_on_host) {
            r = hipMemcpyAsync(host, verdicts, sizeof(int) * (size_t)n_submitted, hipMemcpyDeviceToHost, st);
            v = host;
        }
        if (r == hipSuccess) r = hipStreamSynchronize(st);
        if (r != hipSuccess) { err = r; return 0; }
        int ok = 1;
        for (int i = 0; i < n_submitted; i++) ok &= v[i] == 1 ? 1 : 0;
        n_submitted = 0;
        return ok;
    }
    const int *slot_of = nullptr, *table = nullptr;      // the build's key table (bt_*)
    const int32_t *efirst_host = nullptr;                // first-touch candidate of every clean vertex
    int N;
    int *scratch;            // one device int
    hipStream_t st;
    hipError_t err = hipSuccess;
    int *mailbox = nullptr;                              // pinned host int (null: answers come back through pageable memory)
    int vid_at(int64_t e) override
    {
        {
            auto it = vcache.find(e);
            if (it != vcache.end()) return it->second;
        }
        if (!keys_host && small) {           // its key's hash is the next thing the replay wants: both in one round trip
            vid_hash_q qs;
            qs.n = 1;
            qs.cand[0] = (int)e;
            qs.known[0] = -1;
            return ask(qs) ? vcache[e] : -1;
        }
        int v = -1;
        int *dst = mailbox ? mailbox : &v;
        hipLaunchKernelGGL(k_vid_at, dim3(1), dim3(1), 0, st, table, slot_of, (int)e, scratch);
        hipError_t r = hipGetLastError();
        if (r == hipSuccess) r = hipMemcpyAsync(dst, scratch, sizeof(int), hipMemcpyDeviceToHost, st);
        if (r == hipSuccess) r = hipStreamSynchronize(st);
        if (r != hipSuccess) { err = r; return -1; }
        return *dst;
    }
    int64_t next_occurrence(int vid, int64_t after) override
    {
        int v = 0x7FFFFFFF;
        int *dst = mailbox ? mailbox : &v;
        hipError_t r = hipMemsetD32Async((hipDeviceptr_t)scratch, 0x7FFFFFFF, 1, st);
        if (r == hipSuccess) {
            hipLaunchKernelGGL(k_next_occurrence, dim3(1024), dim3(256), 0, st, slot_of, N, (int)efirst_host[vid], (int)after, scratch);
            r = hipGetLastError();
            if (r == hipSuccess) r = hipMemcpyAsync(dst, scratch, sizeof(int), hipMemcpyDeviceToHost, st);
        }
        if (r == hipSuccess) r = hipStreamSynchronize(st);
        if (r != hipSuccess) { err = r; return -1; }
        return *dst == 0x7FFFFFFF ? -1 : *dst;
    }
};

}  // namespace

// Called by phl_build_device between the clean numbering and the neighbour tables.  lat->vfirst holds every clean
// vertex's first-touch candidate; the candidates' clean vertices are read through lat->bt_slot_of / bt_table.
namespace {
uint64_t replay_cap_max(int64_t M)
{
    uint64_t cap_max = (uint64_t)1 << 15;
    while (cap_max / 2 - 1 <= (uint64_t)M + 128) cap_max <<= 1;
    return cap_max;
}
}  // namespace

size_t phl_reftable_scratch_bytes(int64_t M)
{
    const uint64_t cap_max = replay_cap_max(M);
    return (size_t)(cap_max * sizeof(int) + (cap_max / CL_PIECE + 1) * sizeof(int2) + 32 * 1024);
}

int phl_apply_reference_table(phl_lattice *lat, hipStream_t st, void *arena, size_t arena_bytes,
                              int (*under_replay)(void *), void *under_replay_arg)
{
    const int d = lat->d;
    const int64_t M = lat->M;
    const int N = (int)lat->N;
    lat->n_hidden = 0;
    lat->nbr00_override = -2;
    if (M < (1 << 14) - 1) return PHL_OK;            // the reference's table never doubles: nothing to reproduce
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    static const bool dbg = getenv("PHL_DEBUG") != nullptr;
    arena_pool tmp(arena, arena_bytes);
    int *scratch;
    const int *efirst_dev = lat->vfirst;
    if (!efirst_dev) { phl_set_error("reference-table replay: first touches missing"); return PHL_ERR_INVALID; }
    PHL_HIP(tmp.get(&scratch, 1));
    // (pinned staging where there is some: a 4 MB copy into pageable memory goes through a bounce buffer)
    // The first touches come to the host (the replay searches them); the KEYS stay on the device unless the simulation or
    // a long list of extra creations needs them: the analytic replay wants the hash of a handful of keys only and asks
    // for those (device_query::key_hash / prefetch) -- 3 of the 4.2 MB this copy used to move at C3.
    std::vector<int16_t> keys_pageable;
    std::vector<int32_t> efirst_pageable;
    int16_t *keys = nullptr;
    int32_t *efirst = (int32_t *)phl_pinned_alloc(sizeof(int32_t) * (size_t)M);
    if (!efirst) { efirst_pageable.resize((size_t)M); efirst = efirst_pageable.data(); }
    const char *envf = getenv("PHL_REPLAY_FAST");
    const bool try_fast = !(envf && atoi(envf) == 0);
    auto fetch_keys = [&]() -> hipError_t {      // (blocking; vkeys still holds the clean keys until the result is applied)
        if (keys) return hipSuccess;
        keys = (int16_t *)phl_pinned_alloc(sizeof(int16_t) * (size_t)M * d);
        if (!keys) { keys_pageable.resize((size_t)M * d); keys = keys_pageable.data(); }
        hipError_t r = hipMemcpyAsync(keys, lat->vkeys, sizeof(int16_t) * (size_t)M * d, hipMemcpyDeviceToHost, st);
        if (r == hipSuccess) r = hipStreamSynchronize(st);
        return r;
    };
    if (!try_fast) {
        keys = (int16_t *)phl_pinned_alloc(sizeof(int16_t) * (size_t)M * d);
        if (!keys) { keys_pageable.resize((size_t)M * d); keys = keys_pageable.data(); }
        PHL_HIP(hipMemcpyAsync(keys, lat->vkeys, sizeof(int16_t) * (size_t)M * d, hipMemcpyDeviceToHost, st));
    }
    // the replay's likely questions, answered on the device while the first touches travel (k_spec_many)
    int *spec = (try_fast && lat->bt_slot_of && lat->bt_table) ? (int *)phl_pinned_alloc(sizeof(int) * 4 * SPEC_J * SPEC_X) : nullptr;
    if (spec) {
        hipLaunchKernelGGL(k_spec_many, dim3(1), dim3(64), 0, st, efirst_dev, (int)M, N, lat->bt_table, lat->bt_slot_of, lat->vkeys, d, spec);
        PHL_HIP(hipGetLastError());
    }
    PHL_HIP(hipMemcpyAsync(efirst, efirst_dev, sizeof(int32_t) * (size_t)M, hipMemcpyDeviceToHost, st));
    if (under_replay) {
        // wait for the copies only: what the caller launches now runs while the host replays
        hipEvent_t copied;
        PHL_HIP(hipEventCreateWithFlags(&copied, hipEventDisableTiming));
        hipError_t er = hipEventRecord(copied, st);
        int rcu = PHL_OK;
        if (er == hipSuccess) rcu = under_replay(under_replay_arg);
        if (er == hipSuccess) er = hipEventSynchronize(copied);
        (void)hipEventDestroy(copied);
        if (er != hipSuccess) return phl_hip_fail(er, "reference-table replay: copies", __FILE__, __LINE__);
        if (rcu) return rcu;
    } else {
        PHL_HIP(hipStreamSynchronize(st));
    }
    if (dbg) fprintf(stderr, "[phl] reference table: first touches%s on the host after %.2f ms\n", keys ? " + keys" : "", since());

    device_query q;
    if (!lat->bt_slot_of || !lat->bt_table) { phl_set_error("reference-table replay: build tables missing"); return PHL_ERR_INVALID; }
    q.slot_of = lat->bt_slot_of;
    q.table = lat->bt_table;
    q.efirst_host = efirst;
    q.mailbox = (int *)phl_pinned_alloc(sizeof(int));
    q.N = N;
    q.scratch = scratch;
    q.st = st;
    q.vkeys_dev = lat->vkeys;
    q.keys_host = keys;
    q.d = d;
    q.answers = (int *)phl_pinned_alloc(sizeof(int) * 96);
    if (spec)
        for (int i = 0; i < SPEC_J * SPEC_X; i++) {
            const int e = spec[4 * i], v = spec[4 * i + 1];
            if (e < 0) continue;
            q.vcache[e] = v;
            if (v >= 0) q.hcache[v] = (uint64_t)(unsigned)spec[4 * i + 2] | ((uint64_t)(unsigned)spec[4 * i + 3] << 32);
        }
    phl_reftable_result R;
    // the analytic replay first (counting, no table: phl_reference_table_fast); the simulation where it does not apply
    int rc = 1;
    if (try_fast) {
        const uint64_t cap_max = replay_cap_max(M);
        q.verdicts = (int *)phl_pinned_alloc(sizeof(int) * device_query::MAX_Q);
        q.verdicts_on_host = q.verdicts != nullptr;
        if (tmp.get(&q.hist, (size_t)cap_max) == hipSuccess && tmp.get(&q.small, 256) == hipSuccess &&
            tmp.get(&q.piece_map, (size_t)(cap_max / CL_PIECE) + 1) == hipSuccess &&
            (q.verdicts || tmp.get(&q.verdicts, (size_t)device_query::MAX_Q) == hipSuccess))
            rc = phl_reference_table_fast(keys, efirst, M, d, N, q, R, true);
        else
            (void)hipGetLastError();
        if (dbg) fprintf(stderr, "[phl] reference table: analytic replay %s after %.2f ms\n", rc == 0 ? "done" : (rc == 1 ? "not applicable" : "failed"), since());
        if (rc == 1) R = phl_reftable_result();
    }
    if (rc == 1) {
        PHL_HIP(fetch_keys());
        q.keys_host = keys;
        rc = phl_reference_table_sim(keys, efirst, M, d, N, q, R);
    }
    if (q.err != hipSuccess) return phl_hip_fail(q.err, "reference-table device query", __FILE__, __LINE__);
    if (rc) { phl_set_error("reference-table replay failed (inconsistent first-touch list)"); return rc; }
    if ((int)R.hidden.size() > PHL_MAX_HIDDEN) { phl_set_error("reference-table replay: too many duplicate vertices"); return PHL_ERR_UNSUPPORTED; }

    if (R.compact && R.ex_id.size() > (size_t)PHL_MAX_EXTRAS) {
        PHL_HIP(fetch_keys());
        phl_reftable_expand(keys, M, d, R);
    }
    if (R.M_ref != M || !R.dup_clean.empty()) {
        int *remap_dev;
        extras_t ex;
        ex.n = 0;
        if (R.compact) {
            ex.n = (int)R.ex_id.size();
            for (int k = 0; k < ex.n; k++) { ex.id[k] = R.ex_id[(size_t)k]; ex.clean[k] = R.ex_clean[(size_t)k]; }
        }
        // first touches in the reference's numbering (phl_build_device would otherwise have to find every vertex's home
        // cell with an atomicMin over all N candidates: 1.1 ms at C3)
        std::vector<int32_t> dv_id, dv_e;
        for (size_t k = 0; k + 1 < R.dup_ptr.size(); k++)
            for (int32_t sidx = R.dup_ptr[k]; sidx < R.dup_ptr[k + 1]; sidx++) {
                size_t j = 0;
                for (; j < dv_id.size(); j++)
                    if (dv_id[j] == R.seg_id[(size_t)sidx]) break;
                if (j == dv_id.size()) { dv_id.push_back(R.seg_id[(size_t)sidx]); dv_e.push_back(R.seg_e[(size_t)sidx]); }
                else if (R.seg_e[(size_t)sidx] < dv_e[j]) dv_e[j] = R.seg_e[(size_t)sidx];
            }
        for (int32_t &e : dv_e) e = e < 0 ? 0 : (e >= N ? N - 1 : e);
        // The replay's short lists go up in ONE copy into ONE block (each separate copy is a ~5 us launch of its own on
        // both sides): [dup_ptr | seg_e | seg_id] are kept until the candidates' vertex ids are written (lat->bt_dup_ptr
        // owns the block, bt_seg_e / bt_seg_id point into it: k_final_vid), the rest is read by the launches below.
        // (tracked keys: -(k+1), resolved per candidate by k_final_vid)
        const size_t n_dp = R.dup_ptr.size(), n_sg = R.seg_e.size(), n_dc = R.compact ? R.dup_clean.size() : 0, n_dv = dv_id.size();
        const size_t o_se = n_dp, o_si = o_se + n_sg + 1, o_dc = o_si + n_sg + 1, o_dval = o_dc + n_dc, o_dvid = o_dval + n_dc,
                     o_dve = o_dvid + n_dv, total = o_dve + n_dv + 1;
        std::vector<int32_t> pack(total, 0);
        std::copy(R.dup_ptr.begin(), R.dup_ptr.end(), pack.begin());
        std::copy(R.seg_e.begin(), R.seg_e.end(), pack.begin() + (ptrdiff_t)o_se);
        std::copy(R.seg_id.begin(), R.seg_id.end(), pack.begin() + (ptrdiff_t)o_si);
        for (size_t k = 0; k < n_dc; k++) { pack[o_dc + k] = R.dup_clean[k]; pack[o_dval + k] = -(int32_t)(k + 1); }
        std::copy(dv_id.begin(), dv_id.end(), pack.begin() + (ptrdiff_t)o_dvid);
        std::copy(dv_e.begin(), dv_e.end(), pack.begin() + (ptrdiff_t)o_dve);
        // (kept until the candidates' vertex ids are written: lat->bt_*, k_final_vid)
        PHL_HIP(phl_dev_malloc((void **)&lat->bt_remap, sizeof(int) * (size_t)M));
        PHL_HIP(phl_dev_malloc((void **)&lat->bt_dup_ptr, sizeof(int) * total));
        lat->bt_seg_e = lat->bt_dup_ptr + o_se;
        lat->bt_seg_id = lat->bt_dup_ptr + o_si;
        remap_dev = lat->bt_remap;
        int *const blk = lat->bt_dup_ptr;
        // The guard: on ANY early return it synchronises the stream first (the copies of the host vectors above may still
        // be queued; it is destroyed before them) and releases what the lattice does not own yet.
        struct apply_guard {
            hipStream_t st;
            int16_t *vkeys_new = nullptr;
            int *vfirst_clean = nullptr;
            bool done = false;
            ~apply_guard()
            {
                if (done) return;
                (void)hipStreamSynchronize(st);
                (void)hipGetLastError();
                if (vkeys_new) (void)phl_dev_free(vkeys_new);
                if (vfirst_clean) (void)phl_dev_free(vfirst_clean);
            }
        } guard{st};
        PHL_HIP(hipMemcpyAsync(blk, pack.data(), sizeof(int32_t) * total, hipMemcpyHostToDevice, st));
        if (R.compact) {
            hipLaunchKernelGGL(k_ref_remap, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, (int)M, ex, remap_dev);
            if (n_dc)
                hipLaunchKernelGGL(k_set_pairs, dim3((unsigned)((n_dc + 63) / 64)), dim3(64), 0, st, blk + o_dc, blk + o_dval, (int)n_dc, remap_dev);
            PHL_HIP(hipGetLastError());
        } else {
            PHL_HIP(hipMemcpyAsync(remap_dev, R.remap.data(), sizeof(int) * (size_t)M, hipMemcpyHostToDevice, st));
        }
        int16_t *vkeys_new;
        PHL_HIP(phl_dev_malloc((void **)&vkeys_new, sizeof(int16_t) * (size_t)R.M_ref * d));
        guard.vkeys_new = vkeys_new;
        if (R.compact)
            hipLaunchKernelGGL(k_ref_keys, dim3((unsigned)((R.M_ref + 255) / 256)), dim3(256), 0, st, lat->vkeys, d, (int)R.M_ref, ex, vkeys_new);
        else
            PHL_HIP(hipMemcpyAsync(vkeys_new, R.keys.data(), sizeof(int16_t) * (size_t)R.M_ref * d, hipMemcpyHostToDevice, st));
        int *vfirst_clean = lat->vfirst;              // (= efirst_dev: read by the launch below, released behind the sync)
        lat->vfirst = nullptr;
        guard.vfirst_clean = vfirst_clean;
        PHL_HIP(phl_dev_malloc((void **)&lat->vfirst, sizeof(int) * (size_t)R.M_ref));
        PHL_HIP(hipMemsetAsync(lat->vfirst, 0, sizeof(int) * (size_t)R.M_ref, st));
        hipLaunchKernelGGL(k_vfirst_ref, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, remap_dev, efirst_dev, (int)M, lat->vfirst);
        if (n_dv)
            hipLaunchKernelGGL(k_set_pairs, dim3((unsigned)((n_dv + 63) / 64)), dim3(64), 0, st, blk + o_dvid, blk + o_dve, (int)n_dv, lat->vfirst);
        PHL_HIP(hipGetLastError());
        PHL_HIP(hipStreamSynchronize(st));            // host vectors and pool temporaries die on return
        guard.done = true;                            // from here on the lattice owns vkeys_new; vfirst_clean goes now
        (void)phl_dev_free(vfirst_clean);
        (void)phl_dev_free(lat->vkeys);
        lat->vkeys = vkeys_new;
        lat->M = R.M_ref;
        lat->vfirst_valid_for_M = R.M_ref;
    }
    if (dbg) fprintf(stderr, "[phl] reference table: applied after %.2f ms\n", since());
    lat->n_hidden = (int)R.hidden.size();
    for (int i = 0; i < lat->n_hidden; i++) lat->hidden[i] = R.hidden[i];
    if (R.blur_grow) lat->nbr00_override = R.blur_first_nbr;
    return PHL_OK;
}

// CPU-only entry for tests: the same replay with the candidates' clean vertex ids in a host array.
namespace {
struct host_query : phl_reftable_query {
    const int32_t *cand;
    int64_t N;
    const int16_t *keys = nullptr;
    int d = 0;
    uint64_t hash_of(int clean) const
    {
        uint64_t h = 0;
        for (int i = 0; i < d; i++) { h += (uint64_t)(int64_t)keys[(size_t)clean * d + i]; h *= 2531011; }
        return h;
    }
    int all_ok = 1;
    int probe_paths_all_ok() override
    {
        const int r = all_ok;
        all_ok = 1;
        return r;
    }
    void probe_paths_submit(int64_t n_clean, const std::vector<int32_t> &extra_clean, const std::vector<int32_t> &stale_clean,
                            uint64_t cap, const std::vector<int32_t> &check) override
    {
        all_ok &= answer(n_clean, extra_clean, stale_clean, cap, check);
    }
    int answer(int64_t n_clean, const std::vector<int32_t> &extra_clean, const std::vector<int32_t> &stale_clean,
               uint64_t cap, const std::vector<int32_t> &check)
    {
        std::vector<int32_t> hist((size_t)cap, 0);
        for (int64_t v = 0; v < n_clean; v++) hist[(size_t)(hash_of((int)v) & (cap - 1))]++;
        for (int32_t v : extra_clean) hist[(size_t)(hash_of(v) & (cap - 1))]++;
        for (int32_t v : stale_clean) hist[(size_t)(hash_of(v) & (cap / 2 - 1))]++;
        // carry[x] = entries that arrive at slot x from the left still looking for a slot; two passes settle the wrap
        int64_t carry = 0;
        for (int pass = 0; pass < 2; pass++)
            for (uint64_t x = 0; x < cap; x++) carry = std::max<int64_t>(0, carry + hist[(size_t)x] - 1);
        std::vector<uint8_t> empty((size_t)cap);
        for (uint64_t x = 0; x < cap; x++) {
            empty[(size_t)x] = (carry + hist[(size_t)x]) == 0;
            carry = std::max<int64_t>(0, carry + hist[(size_t)x] - 1);
        }
        for (int32_t v : check) {
            bool ok = false;
            for (uint64_t x = hash_of(v) & (cap - 1); x < cap && !ok; x++) ok = empty[(size_t)x];
            if (!ok) return 0;
        }
        return 1;
    }
    int vid_at(int64_t e) override { return cand[e]; }
    int64_t next_occurrence(int vid, int64_t after) override
    {
        for (int64_t e = after + 1; e < N; e++)
            if (cand[e] == vid) return e;
        return -1;
    }
};
}  // namespace

extern "C" int phl_debug_reference_table(const int16_t *keys_clean, const int32_t *cand_vid, int64_t M, int d, int64_t N,
                                         int16_t *keys_ref_out, int64_t keys_ref_cap, int64_t *M_ref_out,
                                         int32_t *cand_ref_vid_out, int32_t *hidden_out, int hidden_cap, int *n_hidden_out,
                                         int *blur_first_nbr_out)
{
    if (!keys_clean || !cand_vid || M < 0 || d < 1 || N < 0 || !M_ref_out) { phl_set_error("phl_debug_reference_table: bad arguments"); return PHL_ERR_INVALID; }
    std::vector<int32_t> efirst((size_t)M, -1);
    for (int64_t e = N - 1; e >= 0; e--) efirst[cand_vid[e]] = (int32_t)e;
    host_query q;
    q.cand = cand_vid;
    q.N = N;
    q.keys = keys_clean;
    q.d = d;
    phl_reftable_result R;
    // PHL_REPLAY_FAST: 1 = analytic replay where it applies (else the simulation), 2 = analytic only (error if it does
    // not apply), unset / 0 = the simulation
    const char *envf = getenv("PHL_REPLAY_FAST");
    const int mode = envf ? atoi(envf) : 0;
    int rc = 1;
    if (mode >= 1) {
        rc = phl_reference_table_fast(keys_clean, efirst.data(), M, d, N, q, R, true);
        if (rc == 0) phl_reftable_expand(keys_clean, M, d, R);
    }
    if (rc == 1) {
        if (mode == 2) { phl_set_error("analytic replay not applicable"); return PHL_ERR_UNSUPPORTED; }
        R = phl_reftable_result();
        rc = phl_reference_table_sim(keys_clean, efirst.data(), M, d, N, q, R);
    }
    if (rc) return rc;
    *M_ref_out = R.M_ref;
    if (R.M_ref > keys_ref_cap || (int)R.hidden.size() > hidden_cap) { phl_set_error("phl_debug_reference_table: output too small"); return PHL_ERR_INVALID; }
    memcpy(keys_ref_out, R.keys.data(), sizeof(int16_t) * (size_t)R.M_ref * d);
    for (int64_t e = 0; e < N; e++) {
        int r = R.remap[cand_vid[e]];
        if (r < 0) {
            const int k = -r - 1;
            int id = R.seg_id[R.dup_ptr[k]];
            for (int s = R.dup_ptr[k]; s < R.dup_ptr[k + 1] && R.seg_e[s] <= e; s++) id = R.seg_id[s];
            r = id;
        }
        cand_ref_vid_out[e] = r;
    }
    *n_hidden_out = (int)R.hidden.size();
    for (size_t i = 0; i < R.hidden.size(); i++) hidden_out[i] = R.hidden[i];
    *blur_first_nbr_out = R.blur_grow ? R.blur_first_nbr : -2;
    return PHL_OK;
}

// Test entry: the occupancy check of the analytic replay (probe_paths_do_not_wrap) on the device (k_home_hist_add,
// k_add_homes, k_cluster_check) or on the host, for a caller-made key set.  result_out: 1 = no probe path wraps.
extern "C" int phl_debug_probe_paths(const int16_t *keys_clean, int64_t n_clean, int d, const int32_t *extra_clean, int n_extra,
                                     const int32_t *stale_clean, int n_stale, uint64_t cap, const int32_t *check, int n_check,
                                     int on_device, int *result_out)
{
    if (!keys_clean || n_clean < 0 || d < 1 || !result_out || cap < ((uint64_t)1 << 15) || (cap & (cap - 1)) || n_check < 0) {
        phl_set_error("phl_debug_probe_paths: bad arguments");
        return PHL_ERR_INVALID;
    }
    const std::vector<int32_t> ex(extra_clean, extra_clean + n_extra), sc(stale_clean, stale_clean + n_stale), ck(check, check + n_check);
    if (!on_device) {
        host_query q;
        q.cand = nullptr;
        q.N = 0;
        q.keys = keys_clean;
        q.d = d;
        *result_out = q.probe_paths_do_not_wrap(n_clean, ex, sc, cap, ck);
        return PHL_OK;
    }
    temp_pool tmp;
    device_query q;
    int16_t *kd;
    PHL_HIP(tmp.get(&kd, (size_t)n_clean * d + 1));
    PHL_HIP(tmp.get(&q.hist, (size_t)cap));
    PHL_HIP(tmp.get(&q.small, 256));
    PHL_HIP(tmp.get(&q.piece_map, (size_t)(cap / CL_PIECE) + 1));
    PHL_HIP(tmp.get(&q.verdicts, (size_t)device_query::MAX_Q));
    PHL_HIP(hipMemcpy(kd, keys_clean, sizeof(int16_t) * (size_t)n_clean * d, hipMemcpyHostToDevice));

    q.N = 0;
    q.scratch = nullptr;
    q.st = nullptr;
    q.vkeys_dev = kd;
    q.keys_host = keys_clean;
    q.d = d;
    *result_out = q.probe_paths_do_not_wrap(n_clean, ex, sc, cap, ck);
    if (q.err != hipSuccess) return phl_hip_fail(q.err, "phl_debug_probe_paths", __FILE__, __LINE__);
    return PHL_OK;
}
