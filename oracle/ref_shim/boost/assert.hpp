// TEST INFRASTRUCTURE: Boost is absent; BOOST_ASSERT as <cassert>'s assert for the boost::span the reference vendors.
#pragma once
#include <cassert>
#define BOOST_ASSERT(expr) assert(expr)
#define BOOST_ASSERT_MSG(expr, msg) assert((expr) && (msg))
