// sharded_two_contexts.cpp -- a sharded run driven through include/gpe.h alone: no Python, no torch, no launcher.
//
// Several gpe_ctx in ONE process, one host thread each, as the ranks of a decomposition (gpe_shard_layout_build /
// gpe_shard_setup / gpe_shard_run_scheduled): 30 steps with Morton re-sorts at steps 0 and 17, gravity pushing the
// cloud across the cuts so that particles migrate and the ghost bands are busy.  The union of the ranks' particles
// must be bit-identical to the single-context run of the same scene (State::update, /root/reference/src/state.rs:115-131;
// the re-sort is particle_sort.rs:58-69 made global: the order keys a rank downloads are the particles' indices in
// the single-context arrays after ITS re-sorts).
//
//   transport "group"      the library's local group (gpe_local_group_*: hipMemcpyAsync between the contexts' segments,
//                          event-ordered; reductions by a kernel)
//   transport "callbacks"  gpe_shard_set_collectives with this file's own all-reduce / all-to-all, staged through host
//                          memory with gpe_buffer_download / gpe_buffer_upload: what a host with its own message
//                          layer (MPI, sockets) would plug in
//
// usage: sharded_two_contexts [--list] [ranks=2] [particles=40000]
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gpe.h"

namespace {

struct Scene {
    float world_w, world_h;
    std::vector<float> pos, rad;       // interleaved x, y
};

Scene make_scene(uint64_t n)
{
    Scene s;
    s.world_w = 420.0f; s.world_h = 300.0f;
    s.pos.resize(2 * n); s.rad.assign(n, 0.5f);
    uint64_t x = 0x9E3779B97F4A7C15ull;                                 // SplitMix64
    auto next = [&]() {
        x += 0x9E3779B97F4A7C15ull;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    for (uint64_t i = 0; i < n; ++i) {
        s.pos[2 * i] = (float)((next() >> 40) * (1.0 / 16777216.0)) * s.world_w;
        s.pos[2 * i + 1] = (float)((next() >> 40) * (1.0 / 16777216.0)) * s.world_h;
    }
    return s;
}

gpe_ctx *make_ctx(const Scene &s, float gx, float gy)
{
    gpe_config cfg;
    gpe_config_default(&cfg);
    cfg.world_width = s.world_w; cfg.world_height = s.world_h;
    cfg.gravity_x = gx; cfg.gravity_y = gy;
    cfg.mode = GPE_MODE_NATIVE;
    gpe_ctx *c = nullptr;
    if (gpe_create(&cfg, &c) != GPE_OK) {
        std::fprintf(stderr, "gpe_create: %s\n", gpe_last_error(nullptr));
        return nullptr;
    }
    return c;
}

#define CHECK(ctx, expr)                                                                          \
    do {                                                                                          \
        gpe_status _s = (expr);                                                                   \
        if (_s != GPE_OK) {                                                                       \
            std::fprintf(stderr, "%s -> %d: %s\n", #expr, (int)_s, gpe_last_error(ctx));          \
            return false;                                                                         \
        }                                                                                         \
    } while (0)

// ---- the caller's own collectives: a rendezvous of the rank threads, data staged through host memory ---------------
struct HostWorld {
    uint32_t ws = 0;
    std::mutex mu;
    std::condition_variable cv;
    uint32_t arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    std::vector<std::vector<uint32_t>> stage;          // per rank: what it contributes / sends
    std::vector<const uint64_t *> send_off, send_cnt;
    bool meet()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const uint64_t gen = generation;
        if (++arrived == ws) { arrived = 0; ++generation; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return generation != gen || broken; });
        return !broken;
    }
    void abort()
    {
        std::lock_guard<std::mutex> lk(mu);
        broken = true;
        cv.notify_all();
    }
};
struct HostRank { HostWorld *w; gpe_ctx *ctx; uint32_t rank; };

int32_t cb_all_reduce(void *user, uint32_t *d_buf, uint64_t count, uint32_t op, void *)
{
    HostRank *r = (HostRank *)user;
    HostWorld *w = r->w;
    std::vector<uint32_t> &mine = w->stage[r->rank];
    mine.resize(count);
    if (gpe_buffer_download(r->ctx, d_buf, mine.data(), count * 4) != GPE_OK) { w->abort(); return 1; }
    if (!w->meet()) return 1;
    std::vector<uint32_t> out(count);
    for (uint64_t i = 0; i < count; ++i) {
        uint32_t v = w->stage[0][i];
        for (uint32_t p = 1; p < w->ws; ++p) v = op == GPE_REDUCE_MAX ? (v > w->stage[p][i] ? v : w->stage[p][i]) : v + w->stage[p][i];
        out[i] = v;
    }
    if (!w->meet()) return 1;                                            // everybody has read everybody's contribution
    if (gpe_buffer_upload(r->ctx, d_buf, out.data(), count * 4) != GPE_OK) { w->abort(); return 1; }
    return 0;
}

int32_t cb_all_to_all(void *user, const uint32_t *d_send, const uint64_t *send_off, const uint64_t *send_cnt, uint32_t *d_recv,
                      const uint64_t *recv_off, const uint64_t *recv_cnt, void *)
{
    HostRank *r = (HostRank *)user;
    HostWorld *w = r->w;
    uint64_t words = 0;
    for (uint32_t p = 0; p < w->ws; ++p) words = send_off[p] + send_cnt[p] > words ? send_off[p] + send_cnt[p] : words;
    std::vector<uint32_t> &mine = w->stage[r->rank];
    mine.resize(words ? words : 1);
    if (words && gpe_buffer_download(r->ctx, d_send, mine.data(), words * 4) != GPE_OK) { w->abort(); return 1; }
    w->send_off[r->rank] = send_off; w->send_cnt[r->rank] = send_cnt;
    if (!w->meet()) return 1;
    for (uint32_t p = 0; p < w->ws; ++p) {
        if (w->send_cnt[p][r->rank] != recv_cnt[p]) { w->abort(); return 1; }
        if (recv_cnt[p] == 0) continue;
        if (gpe_buffer_upload(r->ctx, d_recv + recv_off[p], w->stage[p].data() + w->send_off[p][r->rank], recv_cnt[p] * 4) != GPE_OK) {
            w->abort();
            return 1;
        }
    }
    if (!w->meet()) return 1;
    return 0;
}

struct RankResult {
    bool ok = false;
    std::vector<uint32_t> gid;
    std::vector<float> pos, prev;
    gpe_shard_stats stats;
};

bool run_rank(const Scene &s, const gpe_shard_layout &L, const std::vector<uint8_t> &owner, uint32_t rank, float gx, float gy,
              gpe_local_group *group, HostWorld *hw, uint64_t steps, uint64_t resort_every, float dt, RankResult *out)
{
    gpe_ctx *c = make_ctx(s, gx, gy);
    if (!c) { if (group) gpe_local_group_abort(group); if (hw) hw->abort(); return false; }
    std::vector<float> p, r;
    std::vector<uint32_t> g;
    const uint64_t n = s.rad.size();
    for (uint64_t i = 0; i < n; ++i)
        if (owner[i] == rank) { p.push_back(s.pos[2 * i]); p.push_back(s.pos[2 * i + 1]); r.push_back(s.rad[i]); g.push_back((uint32_t)i); }
    HostRank me{hw, c, rank};
    auto body = [&]() -> bool {
        CHECK(c, gpe_shard_set_particles(c, p.data(), nullptr, r.data(), g.data(), g.size(), 0));
        if (group) CHECK(c, gpe_local_group_join(c, group, rank));
        else {
            gpe_shard_collectives coll;
            std::memset(&coll, 0, sizeof(coll));
            coll.struct_size = sizeof(coll);
            coll.user = &me;
            coll.all_reduce_u32 = cb_all_reduce;
            coll.all_to_all_u32 = cb_all_to_all;
            CHECK(c, gpe_shard_set_collectives(c, &coll));
        }
        CHECK(c, gpe_shard_setup(c, &L, rank, 1.0f));
        CHECK(c, gpe_shard_run_scheduled(c, dt, steps, resort_every, 1));
        uint64_t cap = 0;
        CHECK(c, gpe_capacity(c, &cap));
        out->gid.resize(cap); out->pos.resize(2 * cap); out->prev.resize(2 * cap);
        uint64_t no = 0;
        CHECK(c, gpe_shard_download_owned(c, out->gid.data(), out->pos.data(), out->prev.data(), cap, &no));
        out->gid.resize(no); out->pos.resize(2 * no); out->prev.resize(2 * no);
        out->stats.struct_size = sizeof(out->stats);
        CHECK(c, gpe_shard_get_stats(c, &out->stats));
        CHECK(c, gpe_sync(c));
        return true;
    };
    out->ok = body();
    if (!out->ok) { if (group) gpe_local_group_abort(group); if (hw) hw->abort(); }
    gpe_destroy(c);
    return out->ok;
}

bool run_case(const char *transport, uint32_t ws, uint64_t n, float gx, float gy)
{
    const uint64_t steps = 30, resort_every = 17;
    const float dt = 0.05f;
    const Scene s = make_scene(n);
    // single context
    gpe_ctx *ref = make_ctx(s, gx, gy);
    if (!ref) return false;
    CHECK(ref, gpe_set_particles(ref, s.pos.data(), nullptr, s.rad.data(), n));
    CHECK(ref, gpe_run(ref, dt, steps, resort_every, 1));
    std::vector<float> want_pos(2 * n), want_prev(2 * n);
    CHECK(ref, gpe_download(ref, GPE_POS, want_pos.data(), 8 * n));
    CHECK(ref, gpe_download(ref, GPE_PREV, want_prev.data(), 8 * n));
    gpe_destroy(ref);
    // the ranks
    gpe_shard_layout L;
    CHECK(nullptr, gpe_shard_layout_build(s.world_w, s.world_h, gpe_compute_cell_size(0.5f), ws, 0, 0, nullptr, nullptr, &L));
    std::vector<uint8_t> owner(n);
    CHECK(nullptr, gpe_shard_layout_owner_of(&L, s.pos.data(), n, owner.data()));
    gpe_local_group *group = nullptr;
    HostWorld hw;
    const bool use_group = std::string(transport) == "group";
    if (use_group) CHECK(nullptr, gpe_local_group_create(ws, &group));
    else { hw.ws = ws; hw.stage.resize(ws); hw.send_off.resize(ws); hw.send_cnt.resize(ws); }
    std::vector<RankResult> res(ws);
    std::vector<std::thread> th;
    for (uint32_t r = 0; r < ws; ++r)
        th.emplace_back([&, r] { run_rank(s, L, owner, r, gx, gy, group, use_group ? nullptr : &hw, steps, resort_every, dt, &res[r]); });
    for (auto &t : th) t.join();
    if (group) gpe_local_group_destroy(group);
    std::vector<uint8_t> seen(n, 0);
    uint64_t total = 0, moved = 0, bad = 0;
    for (uint32_t r = 0; r < ws; ++r) {
        if (!res[r].ok) { std::fprintf(stderr, "rank %u failed\n", r); return false; }
        uint64_t started = 0;
        for (uint64_t i = 0; i < n; ++i) started += owner[i] == r;
        moved += res[r].gid.size() > started ? res[r].gid.size() - started : started - res[r].gid.size();
        for (size_t k = 0; k < res[r].gid.size(); ++k) {
            const uint32_t g = res[r].gid[k];
            if (g >= n || seen[g]) { std::fprintf(stderr, "rank %u: order key %u out of range or owned twice\n", r, g); return false; }
            seen[g] = 1;
            ++total;
            if (std::memcmp(&res[r].pos[2 * k], &want_pos[2 * g], 8) != 0 || std::memcmp(&res[r].prev[2 * k], &want_prev[2 * g], 8) != 0) ++bad;
        }
        if (res[r].stats.resorts != 2 || res[r].stats.steps != steps) {
            std::fprintf(stderr, "rank %u: %llu re-sorts, %llu steps\n", r, (unsigned long long)res[r].stats.resorts,
                         (unsigned long long)res[r].stats.steps);
            return false;
        }
    }
    if (total != n) { std::fprintf(stderr, "%llu of %llu particles owned\n", (unsigned long long)total, (unsigned long long)n); return false; }
    if (bad) { std::fprintf(stderr, "%llu particles differ from the single-context run\n", (unsigned long long)bad); return false; }
    if (gx != 0.0f && moved == 0) { std::fprintf(stderr, "no particle changed rank: the case tests nothing\n"); return false; }
    std::printf("  %u ranks over %s, %llu particles, %llu steps: bit-identical to one context (ownership changed by %llu)\n", ws, transport,
                (unsigned long long)n, (unsigned long long)steps, (unsigned long long)moved);
    return true;
}

}  // namespace

int main(int argc, char **argv)
{
    const char *names[] = {"two_contexts_local_group_equal_one_context", "two_contexts_caller_collectives_equal_one_context",
                           "four_contexts_local_group_equal_one_context"};
    if (argc > 1 && std::string(argv[1]) == "--list") {
        for (const char *n : names) std::printf("%s\n", n);
        return 0;
    }
    const uint64_t n = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 40000;
    int failed = 0;
    struct { const char *name, *transport; uint32_t ws; float gx, gy; } cases[] = {
        {names[0], "group", 2, 40.0f, 0.0f}, {names[1], "callbacks", 2, 40.0f, 0.0f}, {names[2], "group", 4, 25.0f, -30.0f}};
    for (auto &c : cases) {
        const bool ok = run_case(c.transport, c.ws, c.ws == 4 ? n * 3 / 2 : n, c.gx, c.gy);
        std::printf("test %s ... %s\n", c.name, ok ? "ok" : "FAILED");
        failed += !ok;
    }
    return failed ? 1 : 0;
}
