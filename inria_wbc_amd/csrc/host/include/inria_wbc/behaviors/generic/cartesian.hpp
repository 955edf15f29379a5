// generic::cartesian -- SE(3) reference streams for the tasks named in BEHAVIOR.task_names (BASELINE configs 1-2:
// /root/reference/src/behaviors/generic/cartesian.cpp:8-94, /root/reference/etc/franka/cartesian_line.yaml): each task goes
// from its current reference to that reference displaced by relative_targets_pos / relative_targets_rpy on a min-jerk path
// (and back when looping); pose, velocity and acceleration of the current sample go to PosTracker::set_se3_ref every tick.
#ifndef IWBC_HIP_CARTESIAN_HPP
#define IWBC_HIP_CARTESIAN_HPP

#include <inria_wbc/behaviors/behavior.hpp>
#include <inria_wbc/behaviors/reference_streams.hpp>

namespace inria_wbc {
    namespace behaviors {
        namespace generic {
            class Cartesian : public Behavior {
            public:
                Cartesian(const controller_ptr_t& controller, const yaml::Node& config) : Behavior(controller, config)
                {
                    tracker_ = std::dynamic_pointer_cast<controllers::PosTracker>(controller_);
                    IWBC_ASSERT(tracker_, "Need a PosTracker for ", name_in_errors());
                    const yaml::Node c = IWBC_CHECK(config["BEHAVIOR"]);
                    const double duration = IWBC_CHECK(c["trajectory_duration"].as<double>());
                    loop_ = IWBC_CHECK(c["loop"].as<bool>());
                    task_names_ = IWBC_CHECK(c["task_names"].as<std::vector<std::string>>());
                    const auto rel_pos = IWBC_CHECK(c["relative_targets_pos"].as<std::vector<std::vector<double>>>());
                    const auto rel_rpy = IWBC_CHECK(c["relative_targets_rpy"].as<std::vector<std::vector<double>>>());
                    if (task_names_.size() != rel_pos.size()) IWBC_ERROR(name_in_errors(), " behavior needs the same number of tasks and targets");
                    behavior_type_ = controllers::behavior_types::FIXED_BASE;
                    controller_->set_behavior_type(behavior_type_);
                    for (size_t i = 0; i < task_names_.size(); ++i) {
                        const trajs::Vec start = tracker_->get_se3_ref(task_names_[i]);
                        const trajs::Vec goal = displaced(start, rel_pos[i], i < rel_rpy.size() ? rel_rpy[i] : std::vector<double>());
                        Se3Stream s(controller_->dt(), true);
                        s.move(start, goal, duration);
                        if (loop_) s.move(goal, start, duration);
                        streams_.push_back(s);
                    }
                }
                void update(const controllers::SensorData& sensor_data = {}) override
                {
                    const size_t n_seg = streams_.empty() ? 0 : streams_[0].segments();
                    if (!cursor_.finished(n_seg))
                        for (size_t i = 0; i < streams_.size(); ++i) apply(i, streams_[i].sample(cursor_.segment, cursor_.tick));
                    controller_->update(sensor_data);
                    if (!cursor_.finished(n_seg)) cursor_.step(streams_[0].length(cursor_.segment), n_seg, loop_);
                }
                std::string behavior_type() const override { return behavior_type_; }

            protected:
                virtual const char* name_in_errors() const { return "cartesian"; }
                virtual void apply(size_t i, const controllers::TrajectorySample& sample) { tracker_->set_se3_ref(sample, task_names_[i]); }

                std::shared_ptr<controllers::PosTracker> tracker_;
                std::vector<std::string> task_names_;
                std::vector<Se3Stream> streams_;
                SegmentCursor cursor_;
                bool loop_ = false;
            };
        } // namespace generic
    } // namespace behaviors
} // namespace inria_wbc
#endif
