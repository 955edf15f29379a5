// humanoid::move-feet -- generic::cartesian for the feet tasks, with each foot's CONTACT following the same sample, so the
// contact constraint itself moves (/root/reference/src/behaviors/humanoid/move_feet.cpp:8-102, /root/reference/etc/talos/
// move_feet.yaml; PosTracker::set_contact_se3_ref(sample, name), pos_tracker.cpp:240-244).
#ifndef IWBC_HIP_MOVE_FEET_HPP
#define IWBC_HIP_MOVE_FEET_HPP

#include <inria_wbc/behaviors/generic/cartesian.hpp>

namespace inria_wbc {
    namespace behaviors {
        namespace generic {
            class MoveFeet : public Cartesian {
            public:
                MoveFeet(const controller_ptr_t& controller, const yaml::Node& config) : Cartesian(controller, config)
                {
                    contact_names_ = IWBC_CHECK(config["BEHAVIOR"]["contact_names"].as<std::vector<std::string>>());
                    IWBC_ASSERT(contact_names_.size() == task_names_.size(), "MoveFeet behavior needs one contact per task");
                    for (const auto& t : task_names_) IWBC_ASSERT(tracker_->has_task(t), "active_walk: a " + t + " task is required");
                    for (const auto& c : contact_names_) IWBC_ASSERT(tracker_->has_contact(c), "active_walk: a " + c + " task is required");
                    behavior_type_ = controllers::behavior_types::DOUBLE_SUPPORT;
                    controller_->set_behavior_type(behavior_type_);
                }

            protected:
                const char* name_in_errors() const override { return "MoveFeet"; }
                void apply(size_t i, const controllers::TrajectorySample& sample) override
                {
                    tracker_->set_se3_ref(sample, task_names_[i]);
                    tracker_->set_contact_se3_ref(sample, contact_names_[i]);
                }
                std::vector<std::string> contact_names_;
            };
        } // namespace generic
    } // namespace behaviors
} // namespace inria_wbc
#endif
