// ref_shim.cpp -- C entry points around the part of the REFERENCE that compiles
// from its own sources with g++ alone (TEST INFRASTRUCTURE ONLY).
//
// Only upkie/cpp/utils/low_pass_filter.h (and the two exception headers it
// includes) is self-contained: every other C++ file on or next to the path needs
// palimpsest, spdlog, Eigen or Bullet, none of which is in this image, so the
// observers (WheelContact.cpp, FloorContact.cpp, ...) and BulletInterface.cpp
// are treated as unbuildable and restated instead.
//
// The header is included where it lies under /root/reference (see
// oracle/Makefile: -I$(REFERENCE)); nothing is copied into this repository.
#include "upkie/cpp/utils/low_pass_filter.h"

extern "C" {

// upkie::cpp::utils::low_pass_filter (low_pass_filter.h:17-35). *threw is set
// to 1 when the reference throws FilterError (cutoff_period <= 2 dt).
double ref_low_pass_filter(double prev_output, double cutoff_period, double new_input, double dt, int* threw) {
  *threw = 0;
  try {
    return upkie::cpp::utils::low_pass_filter(prev_output, cutoff_period, new_input, dt);
  } catch (const upkie::cpp::exceptions::FilterError&) {
    *threw = 1;
    return 0.0;
  }
}

}  // extern "C"
