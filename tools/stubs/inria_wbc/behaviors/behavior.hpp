// COMPILE-CHECK STUB (tools/stubs/README.md): declarations only (reference: include/inria_wbc/behaviors/behavior.hpp, utils/factory.hpp).
#pragma once
#include <memory>
#include <string>
#include "inria_wbc/controllers/pos_tracker.hpp"
namespace inria_wbc { namespace behaviors {
struct Behavior { virtual ~Behavior(); virtual void update() = 0; };
struct Factory {
    static Factory& instance();
    std::shared_ptr<Behavior> create(const std::string&, const std::shared_ptr<controllers::PosTracker>&, const YAML::Node&);
};
}} // namespace inria_wbc::behaviors
