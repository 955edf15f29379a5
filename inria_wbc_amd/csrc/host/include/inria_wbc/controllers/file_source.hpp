// ProblemSource that replays a batch dumped by tools/dump_batch.py (raw little-endian doubles, one [B][len] block
// per wbcqp_inputs field in header order).  Stands in for pinocchio + task.compute() in the harnesses.
#ifndef IWBC_HIP_FILE_SOURCE_HPP
#define IWBC_HIP_FILE_SOURCE_HPP

#include <cstdio>
#include <cstring>

#include <inria_wbc/controllers/controller.hpp>

namespace inria_wbc {
    namespace controllers {
        class FileSource : public ProblemSource {
        public:
            explicit FileSource(const std::string& path)
            {
                FILE* f = std::fopen(path.c_str(), "rb");
                IWBC_ASSERT(f, "cannot open batch file ", path);
                // header: magic, batch, then one length per field -- eleven fields under the first magic, twelve (+ Acop, the cop task's
                // rows) under the second
                int64_t hdr[14];
                IWBC_ASSERT(std::fread(hdr, sizeof(int64_t), 13, f) == 13, "short header in ", path);
                IWBC_ASSERT(hdr[0] == 0x5742435150ll || hdr[0] == 0x5742435151ll, "bad magic in ", path);
                nf_ = hdr[0] == 0x5742435151ll ? 12 : 11;
                if (nf_ == 12) IWBC_ASSERT(std::fread(hdr + 13, sizeof(int64_t), 1, f) == 1, "short header in ", path);
                batch_ = (int)hdr[1];
                lens_[11] = 0;
                for (int k = 0; k < nf_; ++k) {
                    lens_[k] = (int)hdr[2 + k];
                    data_[k].resize((size_t)batch_ * lens_[k]);
                    if (!data_[k].empty())
                        IWBC_ASSERT(std::fread(data_[k].data(), sizeof(double), data_[k].size(), f) == data_[k].size(), "short read in ", path);
                }
                std::fclose(f);
            }
            int batch() const override { return batch_; }
            void compute(double, const MatrixXd&, const MatrixXd&, const tasks::TaskStack&, const wbcqp_layout& L, TickInputs& in) override
            {
                const int want[12] = {L.len_M, L.len_h, L.len_A, L.len_b1, L.len_Ac, L.len_bc, L.len_blb, L.len_bub, L.len_tlb, L.len_tub, L.len_w, L.len_Acop};
                std::vector<double>* dst[12] = {&in.M, &in.h, &in.A, &in.b1, &in.Ac, &in.bc, &in.blb, &in.bub, &in.tlb, &in.tub, &in.w, &in.Acop};
                for (int k = 0; k < 12; ++k) {
                    IWBC_ASSERT(want[k] == lens_[k], "batch file does not match the task stack (field ", k, ": ", lens_[k], " vs ", want[k], ")");
                    *dst[k] = data_[k];
                }
            }
            void com(MatrixXd& pos, MatrixXd& vel) const override
            {
                pos = MatrixXd(batch_, 3);
                vel = MatrixXd(batch_, 3);
            }

        private:
            int batch_ = 0, nf_ = 11;
            int lens_[12];
            std::vector<double> data_[12];
        };
    } // namespace controllers
} // namespace inria_wbc
#endif
