// Force-included (-include) when compiling the unmodified reference sources:
// common/Scene.hpp:269 calls isnan()/isinf() unqualified, which libstdc++'s
// <cmath> declares only in namespace std.  No behaviour is added or replaced.
#pragma once
#include <cmath>
using std::isinf;
using std::isnan;
