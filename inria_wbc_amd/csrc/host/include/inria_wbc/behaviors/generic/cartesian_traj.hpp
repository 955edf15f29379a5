// generic::cartesian_traj -- SE(3) and CoM references replayed from trajectory files
// (/root/reference/src/behaviors/generic/cartesian_traj.cpp:8-52; file format of src/trajs/loader.cpp, read by trajs::Loader):
// every tick, every task of the `refs` file gets sample `time_` (translations scaled by BEHAVIOR.scale), then time runs
// forward to the end of the files and backward to their start.
#ifndef IWBC_HIP_CARTESIAN_TRAJ_HPP
#define IWBC_HIP_CARTESIAN_TRAJ_HPP

#include <iostream>

#include <inria_wbc/behaviors/behavior.hpp>
#include <inria_wbc/trajs/loader.hpp>

namespace inria_wbc {
    namespace behaviors {
        namespace generic {
            class CartesianTraj : public Behavior {
            public:
                CartesianTraj(const controller_ptr_t& controller, const yaml::Node& config) : Behavior(controller, config)
                {
                    auto c = IWBC_CHECK(config["BEHAVIOR"]);
                    loop_ = IWBC_CHECK(c["loop"].as<bool>());
                    auto yaml_traj = IWBC_CHECK(c["trajectories"].as<std::string>());
                    auto traj_yaml_path = yaml_traj.size() && yaml_traj[0] == '/' ? yaml_traj : controller->base_path() + "/" + yaml_traj;
                    traj_loader_ = std::make_shared<trajs::Loader>(traj_yaml_path);
                    scale_ = IWBC_CHECK(c["scale"].as<double>());
                    step_ = 1;
                }
                void update(const controllers::SensorData& sensor_data = {}) override
                {
                    auto controller = std::static_pointer_cast<controllers::PosTracker>(controller_);
                    for (const auto& task : traj_loader_->ref_names()) {
                        const trajs::SE3& ref = traj_loader_->task_ref(task, time_);
                        controllers::TrajectorySample s(0);
                        s.pos.assign(12, 0.0);
                        for (int k = 0; k < 3; ++k) s.pos[k] = ref.translation[k] * scale_;
                        for (int k = 0; k < 9; ++k) s.pos[3 + k] = ref.rotation[k]; // column-major in the file and in SE3ToVector
                        s.vel.assign(6, 0.0); // set_se3_ref(SE3, name) goes through to_sample(ref): zero derivatives (pos_tracker.cpp:221-226)
                        s.acc.assign(6, 0.0);
                        if (controller->verbose()) std::cout << "task:" << task << " : " << s.pos[0] << " " << s.pos[1] << " " << s.pos[2] << std::endl;
                        controller->set_se3_ref(s, task);
                    }
                    if (traj_loader_->has_com_refs()) {
                        const auto& ref = traj_loader_->com_ref(time_);
                        controllers::TrajectorySample s(3);
                        for (int k = 0; k < 3; ++k) s.pos[k] = ref[k] * scale_;
                        controller->set_com_ref_tracking(s);
                    }
                    controller_->update(sensor_data);
                    time_ += step_;
                    if (time_ >= (int)traj_loader_->size() - 1) step_ = -1;
                    else if (time_ <= 0) step_ = 1;
                }
                std::string behavior_type() const override { return controllers::behavior_types::FIXED_BASE; }
                int time() const { return time_; }

            private:
                int time_ = 0, step_ = 1;
                bool loop_ = false;
                double scale_ = 1.0;
                std::shared_ptr<trajs::Loader> traj_loader_;
            };
        } // namespace generic
    } // namespace behaviors
} // namespace inria_wbc
#endif
