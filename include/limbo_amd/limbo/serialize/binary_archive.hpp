// limbo/serialize/binary_archive.hpp — one ".bin" file per object: rows, cols as Eigen::Index
// (8 bytes each) followed by the column-major doubles; a list is an `int` count followed by its
// items — the on-disk format of src/limbo/serialize/binary_archive.hpp:63-170.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_SERIALIZE_BINARY_ARCHIVE_HPP
#define LIMBO_SERIALIZE_BINARY_ARCHIVE_HPP
#include <cassert>
#include <filesystem>
#include <fstream>
#include <string>
#include <vector>
#include <Eigen/Core>
namespace limbo {
    namespace serialize {
        class BinaryArchive {
        public:
            BinaryArchive(const std::string& dir_name) : _dir_name(dir_name) {}

            void save(const Eigen::MatrixXd& v, const std::string& object_name) const
            {
                std::filesystem::create_directories(_dir_name);
                std::ofstream out(fname(object_name).c_str(), std::ios::out | std::ios::binary | std::ios::trunc);
                _write(out, v);
            }
            template <typename T>
            void save(const std::vector<T>& v, const std::string& object_name) const
            {
                std::filesystem::create_directories(_dir_name);
                std::ofstream out(fname(object_name).c_str(), std::ios::out | std::ios::binary | std::ios::trunc);
                int size = (int)v.size();
                out.write((const char*)&size, sizeof(int));
                for (auto& x : v)
                    _write(out, x);
            }
            template <typename M>
            void load(M& m, const std::string& object_name) const
            {
                std::ifstream in(fname(object_name).c_str(), std::ios::in | std::ios::binary);
                assert(in.good() && "file not found");
                _read(in, m);
            }
            template <typename V>
            void load(std::vector<V>& m_list, const std::string& object_name) const
            {
                m_list.clear();
                std::ifstream in(fname(object_name).c_str(), std::ios::in | std::ios::binary);
                assert(in.good() && "file not found");
                int size = 0;
                in.read((char*)&size, sizeof(int));
                for (int i = 0; i < size; i++) {
                    V v;
                    _read(in, v);
                    m_list.push_back(v);
                }
                assert(!m_list.empty());
            }
            std::string fname(const std::string& object_name) const { return _dir_name + "/" + object_name + ".bin"; }
            const std::string& directory() const { return _dir_name; }

        protected:
            std::string _dir_name;
            template <class Matrix, class Stream>
            void _write(Stream& out, const Matrix& m) const
            {
                Eigen::Index rows = m.rows(), cols = m.cols();
                out.write((const char*)&rows, sizeof(Eigen::Index));
                out.write((const char*)&cols, sizeof(Eigen::Index));
                out.write((const char*)m.data(), rows * cols * sizeof(double));
            }
            template <class Matrix, class Stream>
            void _read(Stream& in, Matrix& m) const
            {
                Eigen::Index rows = 0, cols = 0;
                in.read((char*)&rows, sizeof(Eigen::Index));
                in.read((char*)&cols, sizeof(Eigen::Index));
                m.resize(rows, cols);
                in.read((char*)m.data(), rows * cols * sizeof(double));
            }
        };
    } // namespace serialize
} // namespace limbo
#endif
