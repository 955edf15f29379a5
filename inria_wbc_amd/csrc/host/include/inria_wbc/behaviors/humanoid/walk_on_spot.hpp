// humanoid::walk-on-spot -- the behaviour that changes the QP's size while it runs (SURVEY 3.4;
// /root/reference/src/behaviors/humanoid/walk_on_spot.cpp:8-211, /root/reference/etc/talos/walk_on_spot.yaml): the CoM moves
// over one foot, the other foot's contact is removed (n 74 -> 62, nEq 18 -> 12, nIn 122 -> 105 on Talos), the foot goes up and
// down on a min-jerk path, its contact comes back, and the same on the other side.  The reference wants a
// HumanoidPosTracker only for the cast; nothing of the stabiliser is used here, so a PosTracker does.
#ifndef IWBC_HIP_WALK_ON_SPOT_HPP
#define IWBC_HIP_WALK_ON_SPOT_HPP

#include <inria_wbc/behaviors/behavior.hpp>
#include <inria_wbc/trajs/trajectory_generator.hpp>

namespace inria_wbc {
    namespace behaviors {
        namespace humanoid {
            class WalkOnSpot : public Behavior {
            public:
                enum States { INIT = 0, LIFT_UP_LF, LIFT_DOWN_LF, MOVE_COM_LEFT, LIFT_UP_RF, LIFT_DOWN_RF, MOVE_COM_RIGHT };

                WalkOnSpot(const controller_ptr_t& controller, const yaml::Node& config) : Behavior(controller, config)
                {
                    auto h_controller = std::dynamic_pointer_cast<controllers::PosTracker>(controller_);
                    IWBC_ASSERT(h_controller != NULL, "Walk on spot: the controllers needs to be a PosTracker (or related)!");
                    IWBC_ASSERT(h_controller->has_task("lf"), "Walk on spot: an lf task is required (left foot)");
                    IWBC_ASSERT(h_controller->has_task("rf"), "Walk on spot: an rf task is required (right foot)");
                    IWBC_ASSERT(h_controller->has_task("com"), "Walk: a com task is required");
                    IWBC_ASSERT(h_controller->has_contact("contact_lfoot"), "Walk on spot: a contact_lfoot task is required");
                    IWBC_ASSERT(h_controller->has_contact("contact_rfoot"), "Walk on spot: a contact_rfoot task is required");
                    auto c = IWBC_CHECK(config["BEHAVIOR"]);
                    traj_com_duration_ = IWBC_CHECK(c["traj_com_duration"].as<double>());
                    traj_foot_duration_ = IWBC_CHECK(c["traj_foot_duration"].as<double>());
                    step_height_ = IWBC_CHECK(c["step_height"].as<double>());
                    behavior_type_ = this->behavior_type();
                    controller_->set_behavior_type(behavior_type_);
                    dt_ = controller_->dt();
                    state_ = States::INIT;
                    time_ = 0;
                    _generate_trajectories();
                }

                void update(const controllers::SensorData& sensor_data = {}) override
                {
                    auto controller = std::static_pointer_cast<controllers::PosTracker>(controller_);
                    // add and remove contacts (walk_on_spot.cpp:165-184)
                    if (time_ == 0 && state_ == States::LIFT_UP_LF) {
                        controller->set_behavior_type(controllers::behavior_types::SINGLE_SUPPORT);
                        controller->remove_contact("contact_lfoot");
                    }
                    if (time_ == 0 && state_ == States::LIFT_UP_RF) {
                        controller->set_behavior_type(controllers::behavior_types::SINGLE_SUPPORT);
                        controller->remove_contact("contact_rfoot");
                    }
                    if (time_ == (int)com_trajs_[current_traj_].size() - 1 && state_ == States::LIFT_DOWN_LF) {
                        controller->set_behavior_type(controllers::behavior_types::DOUBLE_SUPPORT);
                        controller->add_contact("contact_lfoot");
                    }
                    if (time_ == (int)com_trajs_[current_traj_].size() - 1 && state_ == States::LIFT_DOWN_RF) {
                        controller->set_behavior_type(controllers::behavior_types::DOUBLE_SUPPORT);
                        controller->add_contact("contact_rfoot");
                    }
                    // the trajectories carry positions only: to_sample_trajectory(traj) leaves the derivatives at zero (:79-84)
                    controllers::TrajectorySample com(3), lf(0), rf(0);
                    com.pos = com_trajs_[current_traj_][time_];
                    lf.pos = lf_trajs_[current_traj_][time_];
                    rf.pos = rf_trajs_[current_traj_][time_];
                    lf.vel.assign(6, 0.0); lf.acc.assign(6, 0.0);
                    rf.vel.assign(6, 0.0); rf.acc.assign(6, 0.0);
                    controller->set_com_ref(com);
                    controller->set_se3_ref(lf, "lf");
                    controller->set_se3_ref(rf, "rf");
                    controller->set_contact_se3_ref(lf.pos, "contact_lfoot");
                    controller->set_contact_se3_ref(rf.pos, "contact_rfoot");
                    controller_->update(sensor_data);
                    time_++;
                    if (time_ == (int)com_trajs_[current_traj_].size()) {
                        time_ = 0;
                        current_traj_ = (current_traj_ + 1) % (int)cycle_.size();
                        if (current_traj_ == 0) current_traj_++; // we skip the init_traj
                        state_ = cycle_[current_traj_];
                    }
                }
                std::string behavior_type() const override { return controllers::behavior_types::DOUBLE_SUPPORT; }
                int state() const { return state_; }

            private:
                void _generate_trajectories()
                {
                    cycle_ = {States::INIT, States::LIFT_UP_LF, States::LIFT_DOWN_LF, States::MOVE_COM_LEFT, States::LIFT_UP_RF, States::LIFT_DOWN_RF,
                              States::MOVE_COM_RIGHT};
                    auto controller = std::static_pointer_cast<controllers::PosTracker>(controller_);
                    auto translate_up = [](trajs::Vec p, double v) { p[2] += v; return p; };
                    const trajs::Vec lf_low = controller->get_se3_ref("lf"), lf_high = translate_up(lf_low, step_height_);
                    const trajs::Vec rf_low = controller->get_se3_ref("rf"), rf_high = translate_up(rf_low, step_height_);
                    // waypoints for the CoM: over lf / rf, same height (walk_on_spot.cpp:62-67)
                    const trajs::Vec com_init = controller->get_com_ref();
                    const trajs::Vec com_lf = {lf_low[0], lf_low[1], com_init[2]}, com_rf = {rf_low[0], rf_low[1], com_init[2]};
                    auto constant = [&](const trajs::Vec& p, double duration) { return std::vector<trajs::Vec>((size_t)std::floor(duration / dt_), p); };
                    auto se3 = [&](const trajs::Vec& a, const trajs::Vec& b) { return trajs::min_jerk_trajectory_se3(a, b, dt_, traj_foot_duration_); };
                    auto com = [&](const trajs::Vec& a, const trajs::Vec& b) { return trajs::min_jerk_trajectory<trajs::d_order::ZERO>(a, b, dt_, traj_com_duration_); };
                    for (auto c : cycle_) {
                        switch (c) {
                        case States::INIT:
                            rf_trajs_.push_back(constant(rf_low, traj_com_duration_)); lf_trajs_.push_back(constant(lf_low, traj_com_duration_));
                            com_trajs_.push_back(com(com_init, com_rf));
                            break;
                        case States::LIFT_UP_LF:
                            rf_trajs_.push_back(constant(rf_low, traj_foot_duration_)); lf_trajs_.push_back(se3(lf_low, lf_high));
                            com_trajs_.push_back(constant(com_rf, traj_foot_duration_));
                            break;
                        case States::LIFT_DOWN_LF:
                            rf_trajs_.push_back(constant(rf_low, traj_foot_duration_)); lf_trajs_.push_back(se3(lf_high, lf_low));
                            com_trajs_.push_back(constant(com_rf, traj_foot_duration_));
                            break;
                        case States::MOVE_COM_LEFT:
                            rf_trajs_.push_back(constant(rf_low, traj_com_duration_)); lf_trajs_.push_back(constant(lf_low, traj_com_duration_));
                            com_trajs_.push_back(com(com_rf, com_lf));
                            break;
                        case States::LIFT_UP_RF:
                            rf_trajs_.push_back(se3(rf_low, rf_high)); lf_trajs_.push_back(constant(lf_low, traj_foot_duration_));
                            com_trajs_.push_back(constant(com_lf, traj_foot_duration_));
                            break;
                        case States::LIFT_DOWN_RF:
                            rf_trajs_.push_back(se3(rf_high, rf_low)); lf_trajs_.push_back(constant(lf_low, traj_foot_duration_));
                            com_trajs_.push_back(constant(com_lf, traj_foot_duration_));
                            break;
                        case States::MOVE_COM_RIGHT:
                            rf_trajs_.push_back(constant(rf_low, traj_com_duration_)); lf_trajs_.push_back(constant(lf_low, traj_com_duration_));
                            com_trajs_.push_back(com(com_lf, com_rf));
                            break;
                        }
                    }
                }

                int time_ = 0, current_traj_ = 0;
                int state_ = States::INIT;
                double dt_ = 0.001, traj_com_duration_ = 1.0, traj_foot_duration_ = 1.0, step_height_ = 0.1;
                std::vector<int> cycle_;
                std::vector<std::vector<trajs::Vec>> lf_trajs_, rf_trajs_, com_trajs_;
            };
        } // namespace humanoid
    } // namespace behaviors
} // namespace inria_wbc
#endif
