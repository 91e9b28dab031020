// TEST INFRASTRUCTURE — stand-in for <boost/shared_ptr.hpp>: the std:: smart pointers under boost's names
#ifndef VINS_REF_FE_BOOST_SHARED_PTR_HPP
#define VINS_REF_FE_BOOST_SHARED_PTR_HPP
#include <memory>
namespace boost {
using std::shared_ptr;
using std::make_shared;
using std::dynamic_pointer_cast;
using std::static_pointer_cast;
using std::const_pointer_cast;
}  // namespace boost
#endif
