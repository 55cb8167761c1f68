// lane_pool.h — the leasing policy of the host-pointer entry points' per-device resource pool (capi.cpp), separated from the HIP
// calls so that tests/sanitize can run it under TSan / ASan with a dummy lane type.
//
// A "lane" (stream + pinned staging buffer + device arena) is leased for the duration of a call.  The pool keeps as many idle
// lanes as callers were inside a call at once during the last kPeakWindowSeconds .. 2 x that (at least kMinIdle) as long as their
// memory stays under a byte budget: a steady pool of N worker threads never creates or frees a lane after warm-up, and the lanes
// of a burst of callers are retired once the burst is that long ago (r04, ADVICE r03: the peak used to be the all-time one, so the
// count rule could never retire anything and a single burst kept its staging memory for the life of the process).  Lanes beyond that are retired lazily: take()
// hands back the ones that have been idle for kIdleSeconds and are surplus to the peak concurrency or the budget, for the CALLER
// to destroy outside the lock; give() never frees anything (hipFree / hipHostFree are device-wide synchronisations: not on a
// call's return path).
#pragma once

#include <algorithm>
#include <chrono>
#include <cstddef>
#include <mutex>
#include <vector>

namespace picha_hip {

template <typename Lane>  // Lane: size_t held_bytes() const; std::chrono::steady_clock::time_point idle_since;
class LanePoolT {
public:
    static constexpr size_t kMinIdle = 16;
    static constexpr size_t kBudgetBytes = (size_t)4 << 30;
    static constexpr int kIdleSeconds = 10;
    static constexpr int kPeakWindowSeconds = 30;

    // An idle lane (most recently used first) or nullptr if the caller has to create one; *surplus receives lanes to destroy.
    Lane *take(std::vector<Lane *> *surplus, int idle_seconds = kIdleSeconds, size_t budget = kBudgetBytes, int peak_window_seconds = kPeakWindowSeconds) {
        std::lock_guard<std::mutex> lk(mu_);
        Lane *lane = nullptr;
        if (!idle_.empty()) { lane = idle_.back(); idle_.pop_back(); }
        ++in_use_;
        // concurrency peak over the current and the previous window
        const auto now = std::chrono::steady_clock::now();
        if (std::chrono::duration_cast<std::chrono::seconds>(now - window_start_).count() >= peak_window_seconds) {
            peak_prev_ = peak_cur_; peak_cur_ = 0; window_start_ = now;
        }
        peak_cur_ = std::max(peak_cur_, in_use_);
        peak_ = std::max(peak_, in_use_);
        collect(surplus, idle_seconds, budget);
        return lane;
    }
    // The caller could not create the lane take() asked for.
    void cancel() { std::lock_guard<std::mutex> lk(mu_); --in_use_; }
    // End of the lease: the lane goes back to the idle list; nothing is freed here.
    void give(Lane *lane) {
        lane->idle_since = std::chrono::steady_clock::now();
        std::lock_guard<std::mutex> lk(mu_);
        idle_.push_back(lane);
        --in_use_;
    }
    size_t idle_count() { std::lock_guard<std::mutex> lk(mu_); return idle_.size(); }
    size_t peak() { std::lock_guard<std::mutex> lk(mu_); return peak_; }

private:
    void collect(std::vector<Lane *> *out, int idle_seconds, size_t budget) {  // mu_ held
        const auto now = std::chrono::steady_clock::now();
        const size_t keep = std::max(kMinIdle, std::max(peak_cur_, peak_prev_));
        size_t bytes = 0;
        for (Lane *l : idle_) bytes += l->held_bytes();
        size_t i = 0;  // oldest first
        while (i < idle_.size() && (idle_.size() + in_use_ > keep || bytes > budget)) {
            Lane *l = idle_[i];
            if (std::chrono::duration_cast<std::chrono::seconds>(now - l->idle_since).count() < idle_seconds) { ++i; continue; }
            bytes -= l->held_bytes();
            out->push_back(l);
            idle_.erase(idle_.begin() + (long)i);
        }
    }
    std::mutex mu_;
    std::vector<Lane *> idle_;  // most recently used at the back
    size_t in_use_ = 0, peak_ = 0;          // peak_: all-time (diagnostics)
    size_t peak_cur_ = 0, peak_prev_ = 0;   // peaks of the current / previous window: what the pool keeps
    std::chrono::steady_clock::time_point window_start_ = std::chrono::steady_clock::now();
};

}  // namespace picha_hip
