// Reference trajectories from files: the wire format of /root/reference/src/trajs/loader.cpp:11-88.
//   refs:                          one YAML file naming one whitespace-separated text file per task
//     lh: lh.csv                   SE3 task: 12 numbers per sample = translation (3) + rotation (9, COLUMN-major:
//     com: com.csv                            the file stores Eigen's default storage order, loader.cpp:28-29)
//     posture: {posture: q.csv, size: 32}     "com" is 3 numbers per sample, "posture" `size` numbers per sample
// Line breaks carry no meaning (the reference reads the file as one stream of doubles, loader.cpp:17-19); every SE3
// file must hold the same number of samples as the first one (loader.cpp:80).
#ifndef IWBC_HIP_TRAJ_LOADER_HPP
#define IWBC_HIP_TRAJ_LOADER_HPP

#include <array>
#include <fstream>
#include <string>
#include <unordered_map>
#include <vector>

#include <inria_wbc/exceptions.hpp>
#include <inria_wbc/utils/yaml_lite.hpp>

namespace inria_wbc {
    namespace trajs {
        struct SE3 {
            std::array<double, 3> translation{{0.0, 0.0, 0.0}};
            std::array<double, 9> rotation{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; // column-major, as the file and Eigen keep it
            double R(int row, int col) const { return rotation[(size_t)col * 3 + row]; }
        };

        class Loader {
        public:
            explicit Loader(const std::string& yaml_path)
            {
                auto doc = IWBC_CHECK(yaml::LoadFile(yaml_path));
                const size_t slash = yaml_path.find_last_of('/');
                const std::string dir = slash == std::string::npos ? std::string() : yaml_path.substr(0, slash + 1);
                auto refs = IWBC_CHECK(doc["refs"]);
                for (const auto& entry : refs) {
                    const std::string& task = entry.first;
                    if (task == "posture") {
                        // first key of the nested map names the task and its file, `size` the width (loader.cpp:66-73)
                        auto first = entry.second.begin();
                        IWBC_ASSERT(first != entry.second.end(), "refs.posture is empty");
                        const int size = IWBC_CHECK(entry.second["size"].as<int>());
                        ref_names_vec_.push_back(first->first);
                        refs_vec_[first->first] = read_rows(dir + first->second.as<std::string>(), size);
                        IWBC_ASSERT(refs_vec_[first->first].size() == refs_vec_.begin()->second.size(), "wrong number of rows in ",
                                    first->second.as<std::string>());
                    }
                    else if (task == "com") {
                        com_refs_ = read_rows(dir + entry.second.as<std::string>(), 3);
                    }
                    else {
                        const std::string file = entry.second.as<std::string>();
                        std::vector<SE3> out;
                        for (const auto& row : read_rows(dir + file, 12)) {
                            SE3 m;
                            for (int j = 0; j < 3; ++j) m.translation[j] = row[j];
                            for (int j = 0; j < 9; ++j) m.rotation[j] = row[3 + j];
                            out.push_back(m);
                        }
                        ref_names_.push_back(task);
                        refs_[task] = std::move(out);
                        IWBC_ASSERT(refs_[task].size() == refs_.at(ref_names_.front()).size(), "wrong number of rows in ", file);
                    }
                }
            }
            const std::vector<std::string>& ref_names() const { return ref_names_; }
            const std::vector<std::string>& ref_names_vec() const { return ref_names_vec_; }
            const SE3& task_ref(const std::string& name, int k) const { return refs_.at(name)[k]; }
            const std::vector<double>& task_ref_vec(const std::string& name, int k) const { return refs_vec_.at(name)[k]; }
            std::array<double, 3> com_ref(int k) const { return {{com_refs_[k][0], com_refs_[k][1], com_refs_[k][2]}}; }
            bool has_com_refs() const { return !com_refs_.empty(); }
            size_t size() const { return refs_.empty() ? 0 : refs_.at(ref_names_.front()).size(); }
            size_t size_vec() const { return refs_vec_.empty() ? 0 : refs_vec_.at(ref_names_vec_.front()).size(); }

        private:
            static std::vector<std::vector<double>> read_rows(const std::string& path, int cols)
            {
                std::ifstream ifs(path.c_str());
                IWBC_ASSERT(ifs.good(), "Error when loading trajectory:", path);
                std::vector<double> all;
                for (double v; ifs >> v;) all.push_back(v);
                IWBC_ASSERT(cols > 0 && all.size() % (size_t)cols == 0, "incomplete line in ", path);
                std::vector<std::vector<double>> rows(all.size() / cols);
                for (size_t i = 0; i < rows.size(); ++i) rows[i].assign(all.begin() + i * cols, all.begin() + (i + 1) * cols);
                return rows;
            }
            std::vector<std::string> ref_names_, ref_names_vec_;
            std::vector<std::vector<double>> com_refs_;
            std::unordered_map<std::string, std::vector<SE3>> refs_;
            std::unordered_map<std::string, std::vector<std::vector<double>>> refs_vec_;
        };
    } // namespace trajs
} // namespace inria_wbc
#endif
