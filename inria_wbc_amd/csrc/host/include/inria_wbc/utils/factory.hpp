// String-keyed self-registering factory (interface parity with
// /root/reference/include/inria_wbc/utils/factory.hpp:15-75: instance(), register_creator, create, AutoRegister;
// duplicate name -> warning and first registration wins; unknown name -> exception listing the known names).
#ifndef IWBC_HIP_FACTORY_HPP
#define IWBC_HIP_FACTORY_HPP

#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <typeinfo>

#include <inria_wbc/exceptions.hpp>

namespace inria_wbc {
    namespace utils {
        template <typename T, class... Types>
        class Factory {
        public:
            using ptr_t = std::shared_ptr<T>;
            using creator_t = std::function<ptr_t(const Types&... args)>;

            template <typename B>
            struct AutoRegister {
                explicit AutoRegister(const std::string& name)
                {
                    instance().register_creator(name, [](const Types&... args) { return std::make_shared<B>(args...); });
                }
                AutoRegister(const std::string& name, const creator_t& creator) { instance().register_creator(name, creator); }
            };

            static Factory& instance()
            {
                static Factory f;
                return f;
            }
            void register_creator(const std::string& name, const creator_t& creator)
            {
                if (!creators_.emplace(name, creator).second)
                    std::cout << "Warning : there is already a " << name << " in the factory [" << typeid(*this).name() << "]" << std::endl;
            }
            ptr_t create(const std::string& name, const Types&... args) const
            {
                auto it = creators_.find(name);
                if (it == creators_.end()) {
                    std::string names;
                    for (const auto& kv : creators_) names += "\t" + kv.first + "\n";
                    throw IWBC_EXCEPTION(name, " is not in the factory [", typeid(*this).name(), "]\nThe factory contains:\n", names);
                }
                return it->second(args...);
            }
            bool has(const std::string& name) const { return creators_.count(name) != 0; }
            void print() const
            {
                for (const auto& kv : creators_) std::cout << kv.first << std::endl;
            }

        private:
            Factory() = default;
            std::map<std::string, creator_t> creators_;
        };
    } // namespace utils
} // namespace inria_wbc
#endif
