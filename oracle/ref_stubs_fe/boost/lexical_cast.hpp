// TEST INFRASTRUCTURE — stand-in for <boost/lexical_cast.hpp>: number -> std::string is all the camera models use
#ifndef VINS_REF_FE_BOOST_LEXICAL_CAST_HPP
#define VINS_REF_FE_BOOST_LEXICAL_CAST_HPP
#include <sstream>
#include <string>
namespace boost {
template <typename T, typename S> inline T lexical_cast(const S& v) {
    std::stringstream ss;
    ss << v;
    T out;
    ss >> out;
    return out;
}
}  // namespace boost
#endif
