// Behavior base class + factory (interface parity with /root/reference/include/inria_wbc/behaviors/behavior.hpp:9-30).
#ifndef IWBC_HIP_BEHAVIOR_HPP
#define IWBC_HIP_BEHAVIOR_HPP

#include <map>

#include <inria_wbc/controllers/pos_tracker.hpp>
#include <inria_wbc/utils/factory.hpp>

namespace inria_wbc {
    namespace behaviors {
        class Behavior {
        public:
            using controller_ptr_t = std::shared_ptr<inria_wbc::controllers::Controller>;
            Behavior(const controller_ptr_t& controller, const yaml::Node& config) : controller_(controller)
            {
                IWBC_ASSERT(controller, "Invalid controller pointer");
                _customize_tasks(controller, config);
            }
            virtual ~Behavior() {}
            virtual void update(const controllers::SensorData& sensor_data = {}) = 0;
            virtual std::shared_ptr<controllers::Controller> controller() { return controller_; }
            virtual std::string behavior_type() const = 0;

        protected:
            // BEHAVIOR.customize_task_weights: {task: weight} overrides (reference behavior.cpp:7-22, etc/talos/walk.yaml:8-9)
            void _customize_tasks(const controller_ptr_t& controller, const yaml::Node& config)
            {
                yaml::Node c = config["BEHAVIOR"];
                if (!c || !c["customize_task_weights"]) return;
                auto pt = std::dynamic_pointer_cast<controllers::PosTracker>(controller);
                IWBC_ASSERT(pt, "customize_task_weights needs a PosTracker");
                std::map<std::string, double> w;
                for (const auto& kv : c["customize_task_weights"]) w[kv.first] = kv.second.as<double>();
                pt->update_task_weights(w);
            }
            std::shared_ptr<inria_wbc::controllers::Controller> controller_;
            std::string behavior_type_;
        };
        using Factory = utils::Factory<Behavior, Behavior::controller_ptr_t, yaml::Node>;
        template <typename T>
        using Register = Factory::AutoRegister<T>;
    } // namespace behaviors
} // namespace inria_wbc
#endif
