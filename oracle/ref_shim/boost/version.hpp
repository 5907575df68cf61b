// TEST INFRASTRUCTURE: Boost is absent; reporting an old version makes aligator/utils/make_span.hpp use the
// boost::span the reference vendors under aligator/third-party/boost/core/.
#pragma once
#define BOOST_VERSION 0
