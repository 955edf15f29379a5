// Building blocks the behaviours share: a reference stream made of consecutive segments (min-jerk moves and holds), for SE(3)
// tasks (pose 12 + velocity 6 + acceleration 6 per sample) and for 3-vectors (CoM), and a cursor that plays segments tick by
// tick.  Every behaviour of the reference precomputes such streams in its constructor and feeds one sample per tick to the
// controller (cartesian.cpp:28-61,66-92; move_com.cpp:22-60; clapping.cpp:28-51; walk_on_spot.cpp:40-160) -- here once.
#ifndef IWBC_HIP_REFERENCE_STREAMS_HPP
#define IWBC_HIP_REFERENCE_STREAMS_HPP

#include <cmath>
#include <vector>

#include <inria_wbc/controllers/pos_tracker.hpp>
#include <inria_wbc/trajs/trajectory_generator.hpp>

namespace inria_wbc {
    namespace behaviors {
        // pose displaced by a relative translation and rotated by Rz(yaw) Ry(pitch) Rx(roll) on the left (cartesian.cpp:31-41);
        // an empty list leaves that part alone.  Poses are 12 numbers, translation then rotation column-major.
        inline trajs::Vec displaced(const trajs::Vec& pose, const std::vector<double>& rel_pos, const std::vector<double>& rel_rpy)
        {
            trajs::Vec out = pose;
            if (rel_pos.size() == 3)
                for (int k = 0; k < 3; ++k) out[k] += rel_pos[k];
            if (rel_rpy.size() == 3) {
                const double cr = std::cos(rel_rpy[0]), sr = std::sin(rel_rpy[0]), cp = std::cos(rel_rpy[1]), sp = std::sin(rel_rpy[1]),
                             cy = std::cos(rel_rpy[2]), sy = std::sin(rel_rpy[2]);
                const double rot[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                                       -sp, cp * sr, cp * cr};
                for (int col = 0; col < 3; ++col)
                    for (int r = 0; r < 3; ++r) {
                        double v = 0.0;
                        for (int k = 0; k < 3; ++k) v += rot[3 * r + k] * pose[3 + 3 * col + k];
                        out[3 + 3 * col + r] = v;
                    }
            }
            return out;
        }

        // SE(3) stream.  with_derivatives = false reproduces the behaviours that hand poses only to the controller
        // (set_se3_ref(SE3, name) -> to_sample(ref): velocity and acceleration zero).
        class Se3Stream {
        public:
            Se3Stream(double dt, bool with_derivatives) : dt_(dt), deriv_(with_derivatives) {}
            void move(const trajs::Vec& from, const trajs::Vec& to, double duration)
            {
                pose_.push_back(trajs::min_jerk_trajectory_se3(from, to, dt_, duration));
                if (deriv_) {
                    vel_.push_back(trajs::min_jerk_trajectory_se3_d<trajs::d_order::FIRST>(from, to, dt_, duration));
                    acc_.push_back(trajs::min_jerk_trajectory_se3_d<trajs::d_order::SECOND>(from, to, dt_, duration));
                }
            }
            void hold(const trajs::Vec& at, double duration)
            {
                IWBC_ASSERT(!deriv_, "hold() belongs to pose-only streams");
                pose_.emplace_back((size_t)std::floor(duration / dt_), at);
            }
            size_t segments() const { return pose_.size(); }
            size_t length(size_t seg) const { return pose_[seg].size(); }
            const trajs::Vec& pose(size_t seg, size_t k) const { return pose_[seg][std::min(k, pose_[seg].size() - 1)]; }
            controllers::TrajectorySample sample(size_t seg, size_t k) const
            {
                controllers::TrajectorySample s(0);
                s.pos = pose(seg, k);
                if (deriv_) { s.vel = vel_[seg][k]; s.acc = acc_[seg][k]; }
                else { s.vel.assign(6, 0.0); s.acc.assign(6, 0.0); }
                return s;
            }

        private:
            double dt_;
            bool deriv_;
            std::vector<std::vector<trajs::Vec>> pose_, vel_, acc_;
        };

        // 3-vector stream (CoM)
        class Vec3Stream {
        public:
            Vec3Stream(double dt, bool with_derivatives) : dt_(dt), deriv_(with_derivatives) {}
            void move(const trajs::Vec& from, const trajs::Vec& to, double duration)
            {
                pos_.push_back(trajs::min_jerk_trajectory<trajs::d_order::ZERO>(from, to, dt_, duration));
                if (deriv_) {
                    vel_.push_back(trajs::min_jerk_trajectory<trajs::d_order::FIRST>(from, to, dt_, duration));
                    acc_.push_back(trajs::min_jerk_trajectory<trajs::d_order::SECOND>(from, to, dt_, duration));
                }
            }
            void hold(const trajs::Vec& at, double duration) { pos_.emplace_back((size_t)std::floor(duration / dt_), at); }
            size_t segments() const { return pos_.size(); }
            size_t length(size_t seg) const { return pos_[seg].size(); }
            controllers::TrajectorySample sample(size_t seg, size_t k) const
            {
                controllers::TrajectorySample s(3);
                s.pos = pos_[seg][k];
                if (deriv_ && seg < vel_.size()) { s.vel = vel_[seg][k]; s.acc = acc_[seg][k]; }
                return s;
            }

        private:
            double dt_;
            bool deriv_;
            std::vector<std::vector<trajs::Vec>> pos_, vel_, acc_;
        };

        // (segment, tick) position in a stream.  After the last tick of a segment the next segment starts; past the last
        // segment the cursor either wraps (loop) or parks beyond the end, where finished() is true.
        struct SegmentCursor {
            size_t segment = 0, tick = 0;
            bool finished(size_t n_segments) const { return segment >= n_segments; }
            void step(size_t segment_length, size_t n_segments, bool loop)
            {
                if (++tick < segment_length) return;
                tick = 0;
                ++segment;
                if (loop && n_segments) segment %= n_segments;
            }
        };
    } // namespace behaviors
} // namespace inria_wbc
#endif
