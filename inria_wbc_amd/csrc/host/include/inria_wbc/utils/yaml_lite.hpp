// Minimal reader for the YAML subset inria_wbc's configuration files use (nested block maps, scalars, flow
// sequences like [0, 0, -0.2] or [[0, 0, -0.2]], '#' comments).  yaml-cpp is not available in this build, so the
// facade keeps the reference's key names (CONTROLLER / BEHAVIOR trees, tasks.yaml entries) and reads them with this.
// Node mimics the part of YAML::Node the reference touches: operator[], as<T>(), explicit bool, ordered map
// iteration (pos_tracker.cpp:161-189 relies on file order), in-memory patching (test_all_robots.cpp:160-164).
#ifndef IWBC_HIP_YAML_LITE_HPP
#define IWBC_HIP_YAML_LITE_HPP

#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include <inria_wbc/exceptions.hpp>

namespace inria_wbc {
    namespace yaml {
        class Node {
        public:
            enum Kind { Undefined, Scalar, Map, Sequence };
            Node() : d_(std::make_shared<Data>()) {}
            explicit Node(const std::string& scalar) : d_(std::make_shared<Data>())
            {
                d_->kind = Scalar;
                d_->scalar = scalar;
            }

            Kind kind() const { return d_->kind; }
            bool IsDefined() const { return d_->kind != Undefined; }
            bool IsMap() const { return d_->kind == Map; }
            bool IsSequence() const { return d_->kind == Sequence; }
            bool IsScalar() const { return d_->kind == Scalar; }
            explicit operator bool() const { return IsDefined(); }
            size_t size() const { return IsMap() ? d_->map.size() : (IsSequence() ? d_->seq.size() : 0); }

            // map access; a missing key yields an undefined node (like yaml-cpp on a const node)
            Node operator[](const std::string& key) const
            {
                if (d_->kind == Map)
                    for (const auto& kv : d_->map)
                        if (kv.first == key) return kv.second;
                Node n;
                n.d_->missing_key = key;
                return n;
            }
            Node operator[](const char* key) const { return (*this)[std::string(key)]; }
            Node operator[](size_t i) const
            {
                if (d_->kind != Sequence || i >= d_->seq.size()) throw std::runtime_error("yaml: bad sequence index");
                return d_->seq[i];
            }
            // in-memory patching: node.set("solver", "hip-batched")
            void set(const std::string& key, const Node& value)
            {
                if (d_->kind == Undefined) d_->kind = Map;
                if (d_->kind != Map) throw std::runtime_error("yaml: set() on a non-map node");
                for (auto& kv : d_->map)
                    if (kv.first == key) {
                        kv.second = value;
                        return;
                    }
                d_->map.emplace_back(key, value);
            }
            void set(const std::string& key, const std::string& scalar) { set(key, Node(scalar)); }
            static Node MakeSequence()
            {
                Node n;
                n.d_->kind = Sequence;
                return n;
            }
            static Node MakeMap()
            {
                Node n;
                n.d_->kind = Map;
                return n;
            }
            void push_back(const Node& v)
            {
                if (d_->kind == Undefined) d_->kind = Sequence;
                d_->seq.push_back(v);
            }

            using map_t = std::vector<std::pair<std::string, Node>>;
            map_t::const_iterator begin() const { return d_->map.begin(); }
            map_t::const_iterator end() const { return d_->map.end(); }
            const std::vector<Node>& items() const { return d_->seq; }

            template <typename T>
            T as() const
            {
                if (d_->kind == Undefined)
                    throw std::runtime_error("yaml: bad conversion, key '" + d_->missing_key + "' is not defined");
                return Convert<T>::get(*this);
            }
            const std::string& scalar() const
            {
                if (d_->kind != Scalar) throw std::runtime_error("yaml: node is not a scalar");
                return d_->scalar;
            }

        private:
            struct Data {
                Kind kind = Undefined;
                std::string scalar, missing_key;
                map_t map;
                std::vector<Node> seq;
            };
            std::shared_ptr<Data> d_;

            template <typename T, typename = void>
            struct Convert;
            friend Node Load(const std::string&);
        };

        template <>
        struct Node::Convert<std::string> {
            static std::string get(const Node& n) { return n.scalar(); }
        };
        template <>
        struct Node::Convert<double> {
            static double get(const Node& n)
            {
                size_t pos = 0;
                const std::string& s = n.scalar();
                double v = std::stod(s, &pos);
                if (pos != s.size()) throw std::runtime_error("yaml: '" + s + "' is not a number");
                return v;
            }
        };
        template <>
        struct Node::Convert<float> {
            static float get(const Node& n) { return (float)Convert<double>::get(n); }
        };
        template <>
        struct Node::Convert<int> {
            static int get(const Node& n) { return (int)std::stol(n.scalar()); }
        };
        template <>
        struct Node::Convert<bool> {
            static bool get(const Node& n)
            {
                const std::string& s = n.scalar();
                if (s == "true" || s == "True" || s == "yes" || s == "on") return true;
                if (s == "false" || s == "False" || s == "no" || s == "off") return false;
                throw std::runtime_error("yaml: '" + s + "' is not a boolean");
            }
        };
        template <typename E>
        struct Node::Convert<std::vector<E>> {
            static std::vector<E> get(const Node& n)
            {
                if (!n.IsSequence()) throw std::runtime_error("yaml: node is not a sequence");
                std::vector<E> out;
                for (const auto& it : n.items()) out.push_back(it.as<E>());
                return out;
            }
        };

        namespace detail {
            inline std::string trim(const std::string& s)
            {
                size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
                return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
            }
            inline std::string strip_comment(const std::string& s)
            {
                bool in_s = false, in_d = false;
                for (size_t i = 0; i < s.size(); ++i) {
                    if (s[i] == '\'' && !in_d) in_s = !in_s;
                    if (s[i] == '"' && !in_s) in_d = !in_d;
                    if (s[i] == '#' && !in_s && !in_d && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
                }
                return s;
            }
            inline std::string unquote(const std::string& s)
            {
                if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
                return s;
            }
            // flow value: scalar or [a, b, [c, d]]
            inline Node parse_flow(const std::string& text, size_t& pos)
            {
                while (pos < text.size() && (text[pos] == ' ' || text[pos] == '\t')) ++pos;
                if (pos < text.size() && text[pos] == '[') {
                    Node seq = Node::MakeSequence();
                    ++pos;
                    bool any = false;
                    while (true) {
                        while (pos < text.size() && (text[pos] == ' ' || text[pos] == ',')) ++pos;
                        if (pos >= text.size()) throw std::runtime_error("yaml: unterminated flow sequence: " + text);
                        if (text[pos] == ']') {
                            ++pos;
                            break;
                        }
                        seq.push_back(parse_flow(text, pos));
                        any = true;
                    }
                    (void)any;
                    return seq;
                }
                size_t start = pos;
                while (pos < text.size() && text[pos] != ',' && text[pos] != ']') ++pos;
                return Node(unquote(trim(text.substr(start, pos - start))));
            }
            struct Line {
                int indent;
                std::string key, value;
                bool has_value;
            };
        } // namespace detail

        inline Node Load(const std::string& text)
        {
            using namespace detail;
            std::vector<Line> lines;
            std::istringstream is(text);
            std::string raw;
            while (std::getline(is, raw)) {
                std::string s = strip_comment(raw);
                if (trim(s).empty()) continue;
                int indent = 0;
                while (indent < (int)s.size() && s[indent] == ' ') ++indent;
                std::string body = trim(s);
                size_t colon = std::string::npos;
                int depth = 0;
                for (size_t i = 0; i < body.size(); ++i) {
                    if (body[i] == '[') ++depth;
                    if (body[i] == ']') --depth;
                    if (body[i] == ':' && depth == 0 && (i + 1 == body.size() || body[i + 1] == ' ')) {
                        colon = i;
                        break;
                    }
                }
                if (colon == std::string::npos) throw std::runtime_error("yaml: unsupported line (expected 'key: value'): " + raw);
                Line l;
                l.indent = indent;
                l.key = unquote(trim(body.substr(0, colon)));
                l.value = trim(body.substr(colon + 1));
                l.has_value = !l.value.empty();
                lines.push_back(l);
            }
            Node root;
            root.d_->kind = Node::Map;
            std::vector<std::pair<int, Node>> stack; // (indent of the keys in this map, map node)
            stack.emplace_back(lines.empty() ? 0 : lines[0].indent, root);
            for (size_t i = 0; i < lines.size(); ++i) {
                const Line& l = lines[i];
                while (stack.size() > 1 && l.indent < stack.back().first) stack.pop_back();
                if (l.indent != stack.back().first) throw std::runtime_error("yaml: inconsistent indentation at key '" + l.key + "'");
                if (l.has_value) {
                    size_t pos = 0;
                    stack.back().second.set(l.key, parse_flow(l.value, pos));
                }
                else {
                    Node child;
                    child.d_->kind = Node::Map;
                    stack.back().second.set(l.key, child);
                    if (i + 1 < lines.size() && lines[i + 1].indent > l.indent) stack.emplace_back(lines[i + 1].indent, child);
                }
            }
            return root;
        }

        inline Node LoadFile(const std::string& path)
        {
            std::ifstream f(path);
            if (!f) throw std::runtime_error("yaml: cannot open '" + path + "'");
            std::stringstream ss;
            ss << f.rdbuf();
            return Load(ss.str());
        }
    } // namespace yaml
} // namespace inria_wbc
#endif
