// Named begin/end timer reporting mean [min:max] in ms every `period` calls
// (interface parity with /root/reference/include/inria_wbc/utils/timer.hpp:8-69, used by qp_timer_test.cpp:58-62).
#ifndef IWBC_HIP_TIMER_HPP
#define IWBC_HIP_TIMER_HPP

#include <algorithm>
#include <chrono>
#include <iostream>
#include <map>
#include <string>

namespace inria_wbc {
    namespace utils {
        class Timer {
        public:
            struct info_t {
                int iterations;
                double time, min_time, max_time; // microseconds
            };
            void begin(const std::string& name) { start_[name] = clock_t::now(); }
            void end(const std::string& name)
            {
                const double us = std::chrono::duration<double, std::micro>(clock_t::now() - start_[name]).count();
                auto it = data_.find(name);
                if (it == data_.end())
                    data_[name] = {1, us, us, us};
                else {
                    it->second.iterations += 1;
                    it->second.time += us;
                    it->second.min_time = std::min(us, it->second.min_time);
                    it->second.max_time = std::max(us, it->second.max_time);
                }
            }
            void report(double t, int period = 100) { report(std::cout, t, period, '\t'); }
            // period = -1: print without touching the counter
            void report(std::ostream& os, double t, int period = 100, char sep = '\t')
            {
                if (period != -1 && ++k_ != period + 1) return;
                os << "t:" << t << sep;
                os.precision(3);
                for (const auto& x : data_)
                    os << x.first << ":" << (x.second.time / x.second.iterations) / 1000.0 << "ms"
                       << " [" << x.second.min_time / 1000.0 << ":" << x.second.max_time / 1000.0 << "]" << sep;
                os << std::endl;
                if (period != -1) {
                    k_ = 1;
                    data_.clear();
                }
            }
            int iteration() const { return k_; }
            const info_t& operator[](const std::string& name) const { return data_.at(name); }

        private:
            using clock_t = std::chrono::high_resolution_clock;
            int k_ = 1;
            std::map<std::string, clock_t::time_point> start_;
            std::map<std::string, info_t> data_;
        };
    } // namespace utils
} // namespace inria_wbc
#endif
