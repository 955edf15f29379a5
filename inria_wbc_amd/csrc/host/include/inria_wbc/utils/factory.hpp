// String-keyed self-registering factory: the plugin mechanism of the reference
// (/root/reference/include/inria_wbc/utils/factory.hpp:15-75 -- instance(), register_creator(), create(), AutoRegister<B>;
// a duplicate name is refused with a warning and the first registration stays, an unknown name throws and lists what is known).
// Here: a registry kept as a name-sorted vector, looked up by binary search; registration and lookup share one locate().
#ifndef IWBC_HIP_FACTORY_HPP
#define IWBC_HIP_FACTORY_HPP

#include <algorithm>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <typeinfo>
#include <utility>
#include <vector>

#include <inria_wbc/exceptions.hpp>

namespace inria_wbc {
    namespace utils {
        template <typename T, class... Types>
        class Factory {
        public:
            using ptr_t = std::shared_ptr<T>;
            using creator_t = std::function<ptr_t(const Types&... args)>;

            static Factory& instance()
            {
                static Factory the_one;
                return the_one;
            }

            // Static objects of this type are how plugins announce themselves: Register<MyController> r("my-name");
            template <typename B>
            struct AutoRegister {
                explicit AutoRegister(const std::string& key) : AutoRegister(key, &AutoRegister::make) {}
                AutoRegister(const std::string& key, const creator_t& how) { Factory::instance().register_creator(key, how); }
                static ptr_t make(const Types&... args) { return std::make_shared<B>(args...); }
            };

            void register_creator(const std::string& key, const creator_t& how)
            {
                auto at = locate(key);
                if (at != entries_.end() && at->first == key) {
                    std::cout << "Warning : there is already a " << key << " in the factory [" << label() << "]" << std::endl;
                    return; // first wins
                }
                entries_.insert(at, std::make_pair(key, how));
            }

            ptr_t create(const std::string& key, const Types&... args) const
            {
                auto at = const_cast<Factory*>(this)->locate(key);
                if (at == entries_.end() || at->first != key) {
                    std::ostringstream known;
                    for (const auto& e : entries_) known << '\t' << e.first << '\n';
                    throw IWBC_EXCEPTION(key, " is not in the factory [", label(), "]\nThe factory contains:\n", known.str());
                }
                return (at->second)(args...);
            }

            bool has(const std::string& key) const
            {
                auto at = const_cast<Factory*>(this)->locate(key);
                return at != entries_.end() && at->first == key;
            }
            std::vector<std::string> names() const
            {
                std::vector<std::string> out;
                for (const auto& e : entries_) out.push_back(e.first);
                return out;
            }
            void print() const
            {
                for (const auto& n : names()) std::cout << n << std::endl;
            }

        private:
            using entry_t = std::pair<std::string, creator_t>;
            Factory() = default;
            Factory(const Factory&) = delete;
            std::string label() const { return typeid(*this).name(); }
            typename std::vector<entry_t>::iterator locate(const std::string& key)
            {
                return std::lower_bound(entries_.begin(), entries_.end(), key, [](const entry_t& e, const std::string& k) { return e.first < k; });
            }
            std::vector<entry_t> entries_; // sorted by name
        };
    } // namespace utils
} // namespace inria_wbc
#endif
