// TSan / ASan harness for picha_amd/csrc/lane_pool.h (the leasing policy of the host entry points' resource pool): worker
// threads lease and return lanes concurrently; no lane may ever be held by two leases at once, nothing may be freed on the
// return path, the pool must not create more lanes than callers were ever inside a call at once, and surplus lanes must be
// handed back for destruction only once they have idled.
#include "../../picha_amd/csrc/lane_pool.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>

struct DummyLane {
    std::chrono::steady_clock::time_point idle_since;
    std::atomic<int> holders{0};
    size_t bytes = 1 << 20;
    size_t held_bytes() const { return bytes; }
};

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); exit(1); } } while (0)

int main() {
    using Pool = picha_hip::LanePoolT<DummyLane>;
    Pool pool;
    std::atomic<int> created{0}, destroyed{0}, concurrent{0}, max_concurrent{0};
    auto worker = [&](int iters) {
        for (int i = 0; i < iters; ++i) {
            std::vector<DummyLane *> surplus;
            DummyLane *l = pool.take(&surplus);
            for (DummyLane *d : surplus) { CHECK(d->holders.load() == 0); delete d; ++destroyed; }
            if (!l) { l = new DummyLane; ++created; }
            CHECK(l->holders.fetch_add(1) == 0);   // exclusively ours
            const int c = ++concurrent;
            int m = max_concurrent.load();
            while (c > m && !max_concurrent.compare_exchange_weak(m, c)) {}
            if ((i & 63) == 0) std::this_thread::yield();
            --concurrent;
            CHECK(l->holders.fetch_sub(1) == 1);
            pool.give(l);
        }
    };
    {   // steady state: 24 threads, more than kMinIdle
        std::vector<std::thread> ts;
        for (int t = 0; t < 24; ++t) ts.emplace_back(worker, 4000);
        for (auto &t : ts) t.join();
    }
    CHECK(destroyed.load() == 0);                                  // nothing retired while the lanes are young
    CHECK(created.load() <= 24 && (size_t)created.load() <= pool.peak());
    CHECK(pool.idle_count() == (size_t)created.load());
    {   // a later, smaller load with a zero idle time and a tiny budget: the surplus is handed back, by take(), not by give()
        std::vector<DummyLane *> surplus;
        DummyLane *l = pool.take(&surplus, /*idle_seconds=*/0, /*budget=*/(size_t)2 << 20);
        CHECK(l != nullptr);
        CHECK(surplus.size() + 2 >= (size_t)created.load() - 1);    // everything but ~2 MB of idle lanes
        for (DummyLane *d : surplus) { delete d; ++destroyed; }
        const size_t before = pool.idle_count();
        pool.give(l);
        CHECK(pool.idle_count() == before + 1);                     // give() frees nothing
    }
    {   // (r04) a burst of 40 concurrent callers long ago must not pin 40 lanes for ever: once the peak windows have rolled over (here:
        // windows of zero seconds) the count rule retires every idle lane beyond kMinIdle that has idled long enough — under an
        // unlimited byte budget
        std::vector<DummyLane *> held;
        for (int i = 0; i < 40; ++i) {
            std::vector<DummyLane *> surplus;
            DummyLane *l = pool.take(&surplus);
            for (DummyLane *d : surplus) { delete d; ++destroyed; }
            if (!l) { l = new DummyLane; ++created; }
            held.push_back(l);
        }
        for (DummyLane *l : held) pool.give(l);
        CHECK(pool.idle_count() >= 40);
        for (int k = 0; k < 3; ++k) {   // three takes: the 40-wide window becomes the previous one, then drops out
            std::vector<DummyLane *> surplus;
            DummyLane *l = pool.take(&surplus, /*idle_seconds=*/0, /*budget=*/(size_t)-1, /*peak_window_seconds=*/0);
            for (DummyLane *d : surplus) { delete d; ++destroyed; }
            CHECK(l != nullptr);
            pool.give(l);
        }
        CHECK(pool.idle_count() <= Pool::kMinIdle);
    }
    // tidy up for the leak checker
    for (;;) {
        std::vector<DummyLane *> surplus;
        DummyLane *l = pool.take(&surplus, 0, 0);
        for (DummyLane *d : surplus) delete d;
        if (!l) { pool.cancel(); break; }
        delete l;
        pool.cancel();
    }
    printf("ok: lane pool, %d lanes created for %d concurrent callers, %d retired lazily\n", created.load(), max_concurrent.load(), destroyed.load());
    return 0;
}
