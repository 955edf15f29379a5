// humanoid::move_com -- the squat of BASELINE config 4 (/root/reference/src/behaviors/humanoid/move_com.cpp:8-61,
// /root/reference/etc/talos/squat.yaml): precomputes min-jerk CoM position / velocity / acceleration tables for every
// target (and the way back when looping), then feeds one sample per tick to the controller.
#ifndef IWBC_HIP_MOVE_COM_HPP
#define IWBC_HIP_MOVE_COM_HPP

#include <algorithm>

#include <inria_wbc/behaviors/behavior.hpp>
#include <inria_wbc/trajs/trajectory_generator.hpp>

namespace inria_wbc {
    namespace behaviors {
        namespace humanoid {
            class MoveCom : public Behavior {
            public:
                MoveCom(const controller_ptr_t& controller, const yaml::Node& config) : Behavior(controller, config)
                {
                    auto tracker = std::dynamic_pointer_cast<controllers::PosTracker>(controller_);
                    IWBC_ASSERT(tracker, "Need a PosTracker for MovCom");
                    auto c = IWBC_CHECK(config["BEHAVIOR"]);
                    const double trajectory_duration = IWBC_CHECK(c["trajectory_duration"].as<double>());
                    behavior_type_ = this->behavior_type();
                    controller_->set_behavior_type(behavior_type_);
                    loop_ = IWBC_CHECK(c["loop"].as<bool>());
                    auto targets = IWBC_CHECK(c["targets"].as<std::vector<std::vector<double>>>());
                    auto mask = IWBC_CHECK(c["mask"].as<std::string>());
                    auto absolute = IWBC_CHECK(c["absolute"].as<bool>());
                    IWBC_ASSERT(mask.size() == 3, "The mask for the CoM should be 3-dimensional");
                    const trajs::Vec task_init = tracker->get_com_ref();
                    if (loop_) targets.push_back(absolute ? task_init : trajs::Vec{0., 0., 0.});
                    trajs::Vec start = task_init;
                    for (const auto& target : targets) {
                        IWBC_ASSERT(target.size() == 3, "references need to be 3-dimensional");
                        trajs::Vec end = task_init;
                        for (size_t j = 0; j < 3; ++j)
                            if (mask[j] == '1') end[j] = absolute ? target[j] : target[j] + task_init[j];
                        auto p = trajs::min_jerk_trajectory<trajs::d_order::ZERO>(start, end, controller_->dt(), trajectory_duration);
                        auto v = trajs::min_jerk_trajectory<trajs::d_order::FIRST>(start, end, controller_->dt(), trajectory_duration);
                        auto a = trajs::min_jerk_trajectory<trajs::d_order::SECOND>(start, end, controller_->dt(), trajectory_duration);
                        trajectory_.insert(trajectory_.end(), p.begin(), p.end());
                        trajectory_d_.insert(trajectory_d_.end(), v.begin(), v.end());
                        trajectory_dd_.insert(trajectory_dd_.end(), a.begin(), a.end());
                        start = end;
                    }
                }
                void update(const controllers::SensorData& sensor_data = {}) override
                {
                    controllers::TrajectorySample sample_ref(3);
                    sample_ref.pos = trajectory_[time_];
                    sample_ref.vel = trajectory_d_[time_];
                    sample_ref.acc = trajectory_dd_[time_];
                    std::static_pointer_cast<controllers::PosTracker>(controller_)->set_com_ref_tracking(sample_ref);
                    controller_->update(sensor_data);
                    time_++;
                    if (loop_)
                        time_ = time_ % (int)trajectory_.size();
                    else
                        time_ = std::min(time_, (int)trajectory_.size() - 1);
                }
                std::string behavior_type() const override { return controllers::behavior_types::DOUBLE_SUPPORT; }
                void set_time(int tick) { time_ = tick % (int)trajectory_.size(); }
                size_t trajectory_size() const { return trajectory_.size(); }

            private:
                int time_ = 0;
                bool loop_ = false;
                std::vector<trajs::Vec> trajectory_, trajectory_d_, trajectory_dd_;
            };
        } // namespace humanoid
    } // namespace behaviors
} // namespace inria_wbc
#endif
