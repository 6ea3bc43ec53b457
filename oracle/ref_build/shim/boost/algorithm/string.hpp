// boost::replace_all stand-in (BO_PARAMS macro, src/limbo/tools/macros.hpp:113-114) — TEST INFRASTRUCTURE.
#ifndef REFSHIM_BOOST_STRING
#define REFSHIM_BOOST_STRING
#include <string>
namespace boost {
    inline void replace_all(std::string& s, const std::string& from, const std::string& to)
    {
        if (from.empty())
            return;
        for (size_t p = 0; (p = s.find(from, p)) != std::string::npos; p += to.size())
            s.replace(p, from.size(), to);
    }
} // namespace boost
#endif
