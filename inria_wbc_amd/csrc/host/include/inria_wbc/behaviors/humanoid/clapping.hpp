// humanoid::clapping -- both hands move motion_size towards each other along y and back, min-jerk, for ever
// (/root/reference/src/behaviors/humanoid/clapping.cpp:8-52, /root/reference/etc/talos/clapping.yaml).  Pose-only references:
// PosTracker::set_se3_ref(SE3, name) goes through to_sample(ref), which leaves the derivatives at zero.
#ifndef IWBC_HIP_CLAPPING_HPP
#define IWBC_HIP_CLAPPING_HPP

#include <inria_wbc/behaviors/behavior.hpp>
#include <inria_wbc/trajs/trajectory_generator.hpp>

namespace inria_wbc {
    namespace behaviors {
        namespace humanoid {
            class Clapping : public Behavior {
            public:
                Clapping(const controller_ptr_t& controller, const yaml::Node& config) : Behavior(controller, config)
                {
                    auto tracker = std::dynamic_pointer_cast<controllers::PosTracker>(controller_);
                    IWBC_ASSERT(tracker, "we need a pos tracker here");
                    const trajs::Vec lh_init = tracker->get_se3_ref("lh"), rh_init = tracker->get_se3_ref("rh");
                    yaml::Node c = IWBC_CHECK(config["BEHAVIOR"]);
                    trajectory_duration_ = IWBC_CHECK(c["trajectory_duration"].as<double>());
                    motion_size_ = IWBC_CHECK(c["motion_size"].as<double>());
                    behavior_type_ = this->behavior_type();
                    controller_->set_behavior_type(behavior_type_);
                    trajs::Vec lh_final = lh_init, rh_final = rh_init;
                    lh_final[1] -= motion_size_;
                    rh_final[1] += motion_size_;
                    const double dt = controller_->dt();
                    lh_trajs_.push_back(trajs::min_jerk_trajectory_se3(lh_init, lh_final, dt, trajectory_duration_));
                    lh_trajs_.push_back(trajs::min_jerk_trajectory_se3(lh_final, lh_init, dt, trajectory_duration_));
                    rh_trajs_.push_back(trajs::min_jerk_trajectory_se3(rh_init, rh_final, dt, trajectory_duration_));
                    rh_trajs_.push_back(trajs::min_jerk_trajectory_se3(rh_final, rh_init, dt, trajectory_duration_));
                }
                void update(const controllers::SensorData& sensor_data = {}) override
                {
                    auto controller = std::static_pointer_cast<controllers::PosTracker>(controller_);
                    controllers::TrajectorySample lh(0), rh(0);
                    lh.pos = lh_trajs_[current_traj_][time_];
                    rh.pos = rh_trajs_[current_traj_][time_];
                    lh.vel.assign(6, 0.0); lh.acc.assign(6, 0.0);
                    rh.vel.assign(6, 0.0); rh.acc.assign(6, 0.0);
                    controller->set_se3_ref(lh, "lh");
                    controller->set_se3_ref(rh, "rh");
                    controller_->update(sensor_data);
                    time_++;
                    if (time_ == lh_trajs_[current_traj_].size()) {
                        time_ = 0;
                        current_traj_ = (current_traj_ + 1) % lh_trajs_.size();
                    }
                }
                std::string behavior_type() const override { return controllers::behavior_types::DOUBLE_SUPPORT; }

            private:
                size_t time_ = 0, current_traj_ = 0;
                double trajectory_duration_ = 1.0, motion_size_ = 0.0;
                std::vector<std::vector<trajs::Vec>> lh_trajs_, rh_trajs_;
            };
        } // namespace humanoid
    } // namespace behaviors
} // namespace inria_wbc
#endif
