// Error conventions of the host facade: exceptions derived from std::runtime_error that carry file:line,
// raised through the same macro names the reference uses (interface parity with
// /root/reference/include/inria_wbc/exceptions.hpp:16-41,52-66; own implementation).
#ifndef IWBC_HIP_EXCEPTIONS_HPP
#define IWBC_HIP_EXCEPTIONS_HPP

#include <sstream>
#include <stdexcept>
#include <string>

namespace inria_wbc {
    namespace detail {
        inline void cat_into(std::ostringstream&) {}
        template <typename T, typename... Rest>
        void cat_into(std::ostringstream& os, const T& v, const Rest&... rest)
        {
            os << v;
            cat_into(os, rest...);
        }
        template <typename... Args>
        std::string cat(const Args&... args)
        {
            std::ostringstream os;
            cat_into(os, args...);
            return os.str();
        }
    } // namespace detail

    class Exception : public std::runtime_error {
    public:
        Exception(const char* file, int line, const std::string& msg)
            : std::runtime_error("inria_wbc:: " + msg + "\t[" + file + ":" + std::to_string(line) + "]\n") {}
    };
} // namespace inria_wbc

// usage: throw IWBC_EXCEPTION("error:", 42)
#define IWBC_EXCEPTION(...) ::inria_wbc::Exception(__FILE__, __LINE__, ::inria_wbc::detail::cat(__VA_ARGS__))
// usage: IWBC_ERROR("message ", value)
#define IWBC_ERROR(...)                    \
    do {                                   \
        throw IWBC_EXCEPTION(__VA_ARGS__); \
    } while (0)
// usage: IWBC_ASSERT(x < 3, "we received x=", x)
#define IWBC_ASSERT(cond, ...)                                        \
    do {                                                              \
        if (!(cond)) throw IWBC_EXCEPTION(#cond, " ", ##__VA_ARGS__); \
    } while (0)
// usage: auto v = IWBC_CHECK(node["key"].as<double>());  decorates any runtime_error with the call site
#define IWBC_CHECK(expr)                                                                      \
    [&]() -> decltype(auto) {                                                                 \
        try {                                                                                 \
            return (expr);                                                                    \
        }                                                                                     \
        catch (const std::runtime_error& e_) {                                                \
            throw IWBC_EXCEPTION("[", e_.what(), "] when calling: ", std::string(#expr));     \
        }                                                                                     \
    }()

#endif
