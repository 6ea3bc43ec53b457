// limbo/serialize/text_archive.hpp — one whitespace-separated ".dat" text file per object, one
// matrix row (or one vector of a list) per line, full precision: the on-disk format of
// src/limbo/serialize/text_archive.hpp:63-151, so directories written by stock limbo load here and
// vice versa (row SURVEY.md §8f N3).  std::filesystem instead of boost::filesystem.
// Interface attribution: the template signature / policy shape of this header reproduces, by requirement (drop-in
// for user code), the public interface of resibots/limbo (Copyright Inria, 2015-; CeCILL-C licence, http://www.cecill.info),
// file named above.  The implementation behind the interface is this project's own.
#ifndef LIMBO_SERIALIZE_TEXT_ARCHIVE_HPP
#define LIMBO_SERIALIZE_TEXT_ARCHIVE_HPP
#include <cassert>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <limits>
#include <sstream>
#include <string>
#include <vector>
#include <Eigen/Core>
namespace limbo {
    namespace serialize {
        class TextArchive {
        public:
            TextArchive(const std::string& dir_name) : _dir_name(dir_name) {}

            void save(const Eigen::MatrixXd& v, const std::string& object_name) const
            {
                std::ofstream ofs(_open_for_write(object_name));
                ofs << std::setprecision(std::numeric_limits<double>::max_digits10);
                for (int i = 0; i < (int)v.rows(); ++i) {
                    for (int j = 0; j < (int)v.cols(); ++j)
                        ofs << (j ? " " : "") << v(i, j);
                    ofs << "\n";
                }
            }
            template <typename T>
            void save(const std::vector<T>& v, const std::string& object_name) const
            {
                std::ofstream ofs(_open_for_write(object_name));
                ofs << std::setprecision(std::numeric_limits<double>::max_digits10);
                for (auto& x : v) {
                    for (int j = 0; j < (int)x.size(); ++j)
                        ofs << (j ? " " : "") << x(j);
                    ofs << "\n";
                }
            }
            template <typename M>
            void load(M& m, const std::string& object_name) const
            {
                auto values = _load(object_name);
                m.resize(values.size(), values[0].size());
                for (size_t i = 0; i < values.size(); ++i)
                    for (size_t j = 0; j < values[i].size(); ++j)
                        m(i, j) = values[i][j];
            }
            template <typename V>
            void load(std::vector<V>& m_list, const std::string& object_name) const
            {
                m_list.clear();
                auto values = _load(object_name);
                for (auto& row : values) {
                    V v(row.size());
                    for (size_t j = 0; j < row.size(); ++j)
                        v(j) = row[j];
                    m_list.push_back(v);
                }
                assert(!m_list.empty());
            }
            std::string fname(const std::string& object_name) const { return _dir_name + "/" + object_name + ".dat"; }
            const std::string& directory() const { return _dir_name; }

        protected:
            std::string _dir_name;
            std::string _open_for_write(const std::string& object_name) const
            {
                std::filesystem::create_directories(_dir_name);
                return fname(object_name);
            }
            std::vector<std::vector<double>> _load(const std::string& object_name) const
            {
                std::ifstream ifs(fname(object_name).c_str());
                assert(ifs.good() && "file not found");
                std::vector<std::vector<double>> v;
                std::string line;
                while (std::getline(ifs, line)) {
                    std::stringstream ls(line);
                    std::vector<double> row;
                    std::string cell;
                    while (ls >> cell)
                        row.push_back(std::stod(cell));
                    if (!row.empty())
                        v.push_back(row);
                }
                assert(!v.empty() && "empty file");
                return v;
            }
        };
    } // namespace serialize
} // namespace limbo
#endif
