// CPU-only harness for lmdeploy_amd/csrc/scheduler.h (pure host logic) under AddressSanitizer + UndefinedBehaviorSanitizer: a seeded random stream of
// submit / admit / token / logprob-record / cancel / forget calls -- the calls the engine makes (engine_serve.hip) -- with the scheduler's invariants
// checked after every call.  Built and run by tests/test_scheduler.py::test_scheduler_under_sanitizers (g++ -fsanitize=address,undefined).
#include "../../lmdeploy_amd/csrc/scheduler.h"

#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

using namespace tmk;

#define CHECK(c)                                                                 \
    do {                                                                         \
        if (!(c)) {                                                              \
            std::fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #c); \
            std::exit(1);                                                        \
        }                                                                        \
    } while (0)

int main(int argc, char** argv)
{
    const unsigned seed = argc > 1 ? (unsigned)std::atoi(argv[1]) : 1u;
    std::mt19937   rng(seed);
    const int      B = 5, NBLK = 40, SESSION = 512;
    BatchScheduler s(B, NBLK, SESSION, 64);
    std::set<int64_t> live;     // ids not yet forgotten
    long tokens = 0, records = 0;
    for (int it = 0; it < 20000; ++it) {
        const int op = (int)(rng() % 100);
        if (op < 25) {  // submit (sometimes invalid / too long / can never fit)
            const int        n = (int)(rng() % 300), mx = (int)(rng() % 260);
            std::vector<int> ids(n > 0 ? n : 1, 7);
            int64_t          id = -1;
            const int        rc = s.submit(ids.data(), n, mx, (rng() & 1) ? 3 : -1, &id);
            CHECK(rc == 0 || rc == 1 || rc == 6 || rc == 11);
            if (rc == 0) {
                live.insert(id);
                if (rng() % 3 == 0) {
                    CHECK(s.set_logprobs(id, 1 + (int)(rng() % 6)) == 0);
                }
                if (rng() % 4 == 0) {
                    const int st[2] = {5, 9};
                    CHECK(s.set_stop_ids(id, st, 2) == 0);
                }
            }
        }
        else if (op < 45) {  // admission
            for (const SchedAdmit& a : s.admit(1 + (int)(rng() % 600))) {
                const SchedRequest* r = s.find(a.id);
                CHECK(r && r->running && r->slot == a.slot && s.slot_request(a.slot) == a.id);
                CHECK((int)r->blocks.size() == s.blocks_for((int)r->prompt.size() + r->max_new));
                CHECK(s.set_logprobs(a.id, 2) == 1);  // too late: the request runs
            }
        }
        else if (op < 85) {  // one decode step: a record (for the requests that asked) and a token per running slot
            for (int b = 0; b < B; ++b) {
                const int64_t id = s.slot_request(b);
                if (id < 0) {
                    continue;
                }
                float     vals[8];
                int       idx[8];
                const int num = (int)(rng() % 9);  // may exceed the request's lp_n, may be 0
                for (int k = 0; k < 8; ++k) {
                    vals[k] = -0.5f * k, idx[k] = 100 + k;
                }
                const int lp_n = s.find(id)->lp_n;
                s.on_logprobs(b, vals, idx, num, -1.25f);
                records += lp_n > 0;
                const int tok = (int)(rng() % 12);
                const bool fin = s.on_token(b, tok);
                ++tokens;
                const SchedRequest* r = s.find(id);
                CHECK(r != nullptr);
                CHECK((r->status == 7) == fin && (fin ? s.slot_request(b) < 0 : s.slot_request(b) == id));
                CHECK(r->lp_num.size() == (r->lp_n > 0 ? r->out.size() : 0u) && r->lp_sel.size() == r->lp_num.size());
                CHECK(r->lp_vals.size() == r->lp_num.size() * (size_t)r->lp_n && r->lp_idx.size() == r->lp_vals.size());
                if (r->lp_n > 0) {
                    const int n = r->lp_num.back();
                    CHECK(n == (num < r->lp_n ? num : r->lp_n));
                    const size_t base = (r->lp_num.size() - 1) * (size_t)r->lp_n;
                    for (int k = 0; k < r->lp_n; ++k) {
                        CHECK(k < n ? r->lp_idx[base + k] == 100 + k : r->lp_idx[base + k] == -1);
                    }
                }
            }
        }
        else if (op < 92 && !live.empty()) {  // cancel
            auto it2 = live.begin();
            std::advance(it2, rng() % live.size());
            int slot = -2;
            CHECK(s.cancel(*it2, &slot) == 0);
            CHECK(slot == -1 || (slot >= 0 && slot < B && s.slot_request(slot) < 0));
        }
        else if (!live.empty()) {  // forget (only finished / cancelled requests go)
            auto it2 = live.begin();
            std::advance(it2, rng() % live.size());
            const SchedRequest* r   = s.find(*it2);
            const bool          fin = r && r->status != 0;
            CHECK(s.erase(*it2) == fin);
            if (fin) {
                CHECK(s.find(*it2) == nullptr);
                live.erase(it2);
            }
        }
        // invariants: every block is free or owned by exactly one running request; slots and requests agree
        int owned = 0, running = 0;
        for (int64_t id : live) {
            const SchedRequest* r = s.find(id);
            CHECK(r != nullptr);
            owned += (int)r->blocks.size();
            running += r->running;
            CHECK(r->running == (r->slot >= 0) && (!r->running || s.slot_request(r->slot) == id));
            CHECK(r->running ? r->status == 0 : r->blocks.empty());
        }
        CHECK(owned + s.n_free_blocks() == NBLK && running == s.n_active() && s.n_records() == live.size());
    }
    if (argc > 2) {
        s.abort_all(5);
        CHECK(s.n_active() == 0 && s.n_free_blocks() == NBLK);
    }
    std::printf("ok seed %u: %ld tokens, %ld logprob records, %zu requests alive\n", seed, tokens, records, live.size());
    return 0;
}
