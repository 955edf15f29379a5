// "humanoid-pos-tracker" and "talos-pos-tracker" (/root/reference/src/controllers/humanoid_pos_tracker.cpp:35,
// /root/reference/src/controllers/talos_pos_tracker.cpp:35): the registration names and constructor signature of the
// reference's two humanoid controllers, so that a configuration file that names them loads here.  Their update() is the
// plain PosTracker tick: what the reference adds around it -- the stabiliser (CoM / ankle / ZMP admittance on force-torque
// and IMU data, humanoid_pos_tracker.cpp:133-260), the CoP estimator and filters, Talos' torque-collision safety
// (talos_pos_tracker.cpp:60-160) -- is host-side signal processing outside the hot path (OUT OF SCOPE)
// and is NOT built.  A configuration that switches the stabiliser on is refused with a message that says so.
#ifndef IWBC_HIP_HUMANOID_POS_TRACKER_HPP
#define IWBC_HIP_HUMANOID_POS_TRACKER_HPP

#include <inria_wbc/controllers/pos_tracker.hpp>

namespace inria_wbc {
    namespace controllers {
        class HumanoidPosTracker : public PosTracker {
        public:
            explicit HumanoidPosTracker(const yaml::Node& config) : PosTracker(config)
            {
                behavior_type_ = behavior_types::FIXED_BASE; // humanoid_pos_tracker.cpp:39
                yaml::Node c = IWBC_CHECK(config["CONTROLLER"]);
                if (c["stabilizer"] && c["stabilizer"]["activated"] && c["stabilizer"]["activated"].as<bool>())
                    IWBC_ERROR("humanoid-pos-tracker: stabilizer.activated is true, but the stabilizer (CoM / ankle / ZMP admittance, CoP "
                               "estimator, sensor filters) is outside the batched hot path and is not part of this build; set "
                               "stabilizer.activated: false");
                if (verbose_) std::cout << "Humanoid pos tracker initialized (plain tick: no stabilizer in this build)" << std::endl;
            }
            // the reference accepts FIXED_BASE / SINGLE_SUPPORT / DOUBLE_SUPPORT only (humanoid_pos_tracker.cpp:116-131)
            void set_behavior_type(const std::string& bt) override
            {
                if (bt != behavior_types::FIXED_BASE && bt != behavior_types::SINGLE_SUPPORT && bt != behavior_types::DOUBLE_SUPPORT)
                    IWBC_ERROR("_stabilizer_configs does not have ", bt);
                Controller::set_behavior_type(bt);
            }
        };

        class TalosPosTracker : public HumanoidPosTracker {
        public:
            explicit TalosPosTracker(const yaml::Node& config) : HumanoidPosTracker(config)
            {
                yaml::Node c = IWBC_CHECK(config["CONTROLLER"]);
                if (c["collision_detection"] && c["collision_detection"]["activated"] && c["collision_detection"]["activated"].as<bool>())
                    IWBC_ERROR("talos-pos-tracker: collision_detection.activated is true, but the torque-collision safety "
                               "(talos_pos_tracker.cpp:60-160) is outside the batched hot path and is not part of this build; set it to false");
                if (verbose_) std::cout << "Talos pos tracker initialized (no torque safety, no torso roll clamp in this build)" << std::endl;
            }
        };
    } // namespace controllers
} // namespace inria_wbc
#endif
