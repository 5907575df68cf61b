// TEST INFRASTRUCTURE: stand-in for the generated Tracy switch header with profiling off (ALIGATOR_TRACY_ENABLE=OFF).
#pragma once
#define ALIGATOR_TRACY_ZONE_SCOPED
#define ALIGATOR_TRACY_ZONE_SCOPED_N(name)
#define ALIGATOR_TRACY_ZONE_NAMED(var, active)
#define ALIGATOR_TRACY_ZONE_NAMED_N(var, name, active)
