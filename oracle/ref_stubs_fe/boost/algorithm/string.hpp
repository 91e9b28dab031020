// TEST INFRASTRUCTURE — stand-in for <boost/algorithm/string.hpp>: iequals (ASCII case-insensitive comparison)
#ifndef VINS_REF_FE_BOOST_ALGORITHM_STRING_HPP
#define VINS_REF_FE_BOOST_ALGORITHM_STRING_HPP
#include <cctype>
#include <string>
namespace boost {
inline bool iequals(const std::string& a, const std::string& b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i)
        if (std::tolower((unsigned char)a[i]) != std::tolower((unsigned char)b[i])) return false;
    return true;
}
}  // namespace boost
#endif
