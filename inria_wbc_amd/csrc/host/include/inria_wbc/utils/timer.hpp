// Named stopwatch sections with a periodic one-line report "t:<t> <name>:<mean>ms [<min>:<max>]" -- the tool the reference's
// harness prints its solver times with (/root/reference/include/inria_wbc/utils/timer.hpp:8-69, qp_timer_test.cpp:58-62).
// Same calls (begin / end / report / iteration / operator[]); inside, each section is a small accumulator object and the report
// is assembled in one string before it goes out.
#ifndef IWBC_HIP_TIMER_HPP
#define IWBC_HIP_TIMER_HPP

#include <chrono>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>

namespace inria_wbc {
    namespace utils {
        class Timer {
            using stopwatch = std::chrono::steady_clock;

        public:
            // one section's statistics, in microseconds (field names as the reference's info_t)
            struct info_t {
                int iterations = 0;
                double time = 0.0;
                double min_time = std::numeric_limits<double>::infinity();
                double max_time = 0.0;
                void take(double us)
                {
                    ++iterations;
                    time += us;
                    if (us < min_time) min_time = us;
                    if (us > max_time) max_time = us;
                }
                double mean_ms() const { return iterations ? time / iterations * 1e-3 : 0.0; }
            };

            void begin(const std::string& section) { opened_[section] = stopwatch::now(); }
            void end(const std::string& section)
            {
                const auto closed = stopwatch::now();
                auto it = opened_.find(section);
                if (it == opened_.end()) throw std::runtime_error("Timer::end(" + section + ") without begin");
                stats_[section].take(std::chrono::duration<double, std::micro>(closed - it->second).count());
            }

            void report(double t, int period = 100) { report(std::cout, t, period, '\t'); }
            // Every `period` calls: print, then start over.  period == -1: print now and keep counting (a second stream).
            void report(std::ostream& out, double t, int period = 100, char sep = '\t')
            {
                const bool counting = period != -1;
                if (counting && ++calls_ <= period) return;
                std::ostringstream line;
                line << "t:" << t << sep << std::setprecision(3);
                for (const auto& kv : stats_)
                    line << kv.first << ':' << kv.second.mean_ms() << "ms [" << kv.second.min_time * 1e-3 << ':' << kv.second.max_time * 1e-3 << ']' << sep;
                out << line.str() << std::endl;
                if (counting) {
                    calls_ = 1;
                    stats_.clear();
                }
            }
            int iteration() const { return calls_; }
            const info_t& operator[](const std::string& section) const { return stats_.at(section); }

        private:
            int calls_ = 1;
            std::map<std::string, stopwatch::time_point> opened_;
            std::map<std::string, info_t> stats_;
        };
    } // namespace utils
} // namespace inria_wbc
#endif
