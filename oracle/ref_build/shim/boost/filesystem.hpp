// Stand-in for the two names of Boost.Filesystem that resibots/limbo's archives use (serialize/text_archive.hpp:129-130,
// binary_archive.hpp:139-140: boost::filesystem::path, create_directories) — TEST INFRASTRUCTURE, over std::filesystem.
// The reference's archives also rely on this header to bring in <fstream> (they declare std::ofstream / std::ifstream
// and include only <iostream> / <sstream> themselves).
#ifndef REFSHIM_BOOST_FILESYSTEM_HPP
#define REFSHIM_BOOST_FILESYSTEM_HPP
#include <filesystem>
#include <fstream>
namespace boost {
    namespace filesystem {
        using path = std::filesystem::path;
        inline bool create_directories(const path& p) { return std::filesystem::create_directories(p); }
    } // namespace filesystem
} // namespace boost
#endif
