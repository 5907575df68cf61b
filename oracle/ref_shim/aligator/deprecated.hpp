// TEST INFRASTRUCTURE: stand-in for the generated deprecation header (see config.hpp here).
#pragma once
#define ALIGATOR_DEPRECATED [[deprecated]]
#define ALIGATOR_DEPRECATED_MESSAGE(msg) [[deprecated(msg)]]
#define ALIGATOR_DEPRECATED_HEADER(msg)
