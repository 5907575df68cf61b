// TEST INFRASTRUCTURE (oracle/ref_shim): aligator/utils/logger.hpp keeps its columns in a boost::unordered_map
#pragma once
#include <unordered_map>
#include <vector>
#include <string>
namespace boost {
template <class K, class V, class... R> using unordered_map = std::unordered_map<K, V>;
}
