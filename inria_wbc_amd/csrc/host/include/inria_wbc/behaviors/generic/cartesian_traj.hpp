// generic::cartesian_traj -- references replayed from trajectory files (/root/reference/src/behaviors/generic/
// cartesian_traj.cpp:8-52; wire format of src/trajs/loader.cpp, read by trajs::Loader): sample k of every file goes to its
// task (translations times BEHAVIOR.scale, CoM too when the file set has one); k sweeps to the end of the files and back.
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
                    tracker_ = std::dynamic_pointer_cast<controllers::PosTracker>(controller_);
                    IWBC_ASSERT(tracker_, "Need a PosTracker for CartesianTraj");
                    const yaml::Node c = IWBC_CHECK(config["BEHAVIOR"]);
                    (void)IWBC_CHECK(c["loop"].as<bool>()); // read and unused, as in the reference: the sweep never stops
                    const auto file = IWBC_CHECK(c["trajectories"].as<std::string>());
                    files_ = std::make_shared<trajs::Loader>(file.size() && file[0] == '/' ? file : controller->base_path() + "/" + file);
                    scale_ = IWBC_CHECK(c["scale"].as<double>());
                }
                void update(const controllers::SensorData& sensor_data = {}) override
                {
                    for (const auto& task : files_->ref_names()) {
                        const trajs::SE3& pose = files_->task_ref(task, k_);
                        controllers::TrajectorySample s(0); // pose only: set_se3_ref(SE3, name) leaves the derivatives at zero
                        s.pos.resize(12);
                        for (int i = 0; i < 3; ++i) s.pos[i] = scale_ * pose.translation[i];
                        std::copy(pose.rotation.begin(), pose.rotation.end(), s.pos.begin() + 3); // column-major on both sides
                        s.vel.assign(6, 0.0);
                        s.acc.assign(6, 0.0);
                        if (tracker_->verbose()) std::cout << "task:" << task << " : " << s.pos[0] << " " << s.pos[1] << " " << s.pos[2] << std::endl;
                        tracker_->set_se3_ref(s, task);
                    }
                    if (files_->has_com_refs()) {
                        const auto com = files_->com_ref(k_);
                        controllers::TrajectorySample s(3);
                        for (int i = 0; i < 3; ++i) s.pos[i] = scale_ * com[i];
                        tracker_->set_com_ref_tracking(s);
                    }
                    controller_->update(sensor_data);
                    k_ += direction_;
                    if (k_ >= (int)files_->size() - 1) direction_ = -1; // turn round at either end
                    else if (k_ <= 0) direction_ = 1;
                }
                std::string behavior_type() const override { return controllers::behavior_types::FIXED_BASE; }
                int time() const { return k_; }

            private:
                std::shared_ptr<controllers::PosTracker> tracker_;
                std::shared_ptr<trajs::Loader> files_;
                double scale_ = 1.0;
                int k_ = 0, direction_ = 1;
            };
        } // namespace generic
    } // namespace behaviors
} // namespace inria_wbc
#endif
